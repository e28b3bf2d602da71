// Shared host-side helpers for libmmhand_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/mmhand_hip.h"

namespace mmh {

// thread-local error text returned by mmh_last_error()
char* err_buf();
int fail(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("%s: %s", what, hipGetErrorString(e));
    return 0;
}

inline hipStream_t as_stream(mmh_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace mmh

#define MMH_REQUIRE(cond, ...)                      \
    do {                                            \
        if (!(cond)) return mmh::fail(__VA_ARGS__); \
    } while (0)
