// Gradient-bucket all-reduce on RCCL (mmh_allreduce_bucket; replaces apex DistributedDataParallel's
// flattened all-reduce behind models/MMHandModel.py:109-116,381-384).
//
// The library has no link-time dependency on librccl: the process that trains already holds ONE RCCL communicator per
// GPU (torch.distributed's "nccl" backend IS RCCL on ROCm) and the RCCL image that communicator lives in.  mmh_rccl_bind()
// resolves ncclAllReduce from THAT image (dlopen of the path the caller names with RTLD_NOLOAD: never loads one), so the collective below
// runs on the caller's communicator and the caller's stream: no second communicator, no second RCCL in the process.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include "common.h"

namespace {
using allreduce_fn = ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
using errstr_fn = const char* (*)(ncclResult_t);
using count_fn = ncclResult_t (*)(const ncclComm_t, int*);
allreduce_fn p_allreduce = nullptr;
errstr_fn p_errstr = nullptr;
count_fn p_count = nullptr;
}  // namespace

extern "C" {

int mmh_rccl_bind(const char* librccl_path) {
    MMH_REQUIRE(librccl_path && *librccl_path, "mmh_rccl_bind: path of the RCCL library the process uses is required");
    // RTLD_NOLOAD only: the image must ALREADY be mapped (the caller names the one its communicator lives in, from
    // /proc/self/maps).  Loading a second RCCL image and handing it a communicator created by the first would read a
    // foreign struct layout - so a miss is an error and the caller keeps torch.distributed's own all_reduce.
    void* h = dlopen(librccl_path, RTLD_NOW | RTLD_NOLOAD);
    MMH_REQUIRE(h, "mmh_rccl_bind: %s is not loaded in this process (refusing to load a second RCCL image)", librccl_path);
    p_allreduce = reinterpret_cast<allreduce_fn>(dlsym(h, "ncclAllReduce"));
    p_errstr = reinterpret_cast<errstr_fn>(dlsym(h, "ncclGetErrorString"));
    p_count = reinterpret_cast<count_fn>(dlsym(h, "ncclCommCount"));
    MMH_REQUIRE(p_allreduce && p_errstr && p_count, "mmh_rccl_bind: %s does not export ncclAllReduce / ncclGetErrorString / "
                "ncclCommCount", librccl_path);
    return 0;
}

int mmh_rccl_comm_ranks(void* comm) {
    if (!p_count || !comm) return -1;
    int n = -1;
    return p_count(static_cast<ncclComm_t>(comm), &n) == ncclSuccess ? n : -1;
}

int mmh_allreduce_bucket(void* comm, void* buf, int64_t count, int dtype, mmh_stream_t s) {
    MMH_REQUIRE(p_allreduce, "mmh_allreduce_bucket: call mmh_rccl_bind(<path of librccl>) first");
    MMH_REQUIRE(comm && buf && count > 0, "mmh_allreduce_bucket: null communicator / buffer or empty bucket");
    ncclDataType_t t;
    switch (dtype) {
        case MMH_F32: t = ncclFloat32; break;
        case MMH_BF16: t = ncclBfloat16; break;
        case MMH_FP16: t = ncclFloat16; break;
        default: return mmh::fail("mmh_allreduce_bucket: dtype %d", dtype);
    }
    ncclResult_t r = p_allreduce(buf, buf, static_cast<size_t>(count), t, ncclSum, static_cast<ncclComm_t>(comm),
                                 mmh::as_stream(s));
    MMH_REQUIRE(r == ncclSuccess, "mmh_allreduce_bucket: ncclAllReduce: %s", p_errstr(r));
    return 0;
}

}  // extern "C"
