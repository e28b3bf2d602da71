// Error plumbing of the C-ABI: thread-local message + version.
#include "common.h"

namespace mmh {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return 1;
}

}  // namespace mmh

extern "C" {
const char* mmh_last_error(void) { return mmh::err_buf(); }
int mmh_version(void) { return 100; }
}
