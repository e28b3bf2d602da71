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

// Winograd F(6x6,3x3) transforms (wino6.hip); tiles = B * ceil(H/6) * ceil(W/6), 64 planes
int wino6_weights(const float* w, float* U, int Cin, int Cout, int flip_transpose, hipStream_t st);
int wino6_input(const float* x, float* V, int B, int H, int W, int C, int reflect, int xcd, hipStream_t st);
int wino6_output(const float* M, float* y, const float* bias, int B, int H, int W, int C, int act, float* stats,
                 int fold, hipStream_t st);
int wino6_dy(const float* dy, float* Yh, int B, int H, int W, int C, hipStream_t st);
int wino6_input_dy(const float* dy, float* V, float* Yh, int B, int H, int W, int C, int xcd, int fold,
                   hipStream_t st);
int wino6_dw(const float* dU, float* dw, int Cin, int Cout, int accumulate, hipStream_t st);
extern int g_wino6_vec;
extern int g_lp16_shape;
extern int g_lp16_tap_inner;
extern int g_lp16_dbg;
extern int g_lp16_wgrad_ring;
extern int g_pw_v2;

}  // namespace mmh

#define MMH_REQUIRE(cond, ...)                      \
    do {                                            \
        if (!(cond)) return mmh::fail(__VA_ARGS__); \
    } while (0)
