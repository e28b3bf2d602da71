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

// One element of the norm backward, dx = k0*(dz - k1) - (x - mu)*k2 with dz = keep ? g/(1-p) : 0, as a
// PINNED sequence of operations (the empty asm statements stop the compiler from contracting it with
// its neighbours): norm_bwd_apply_v2 (pointwise.hip) and the backward transform that computes dx on
// the fly (wino6.hip) must round identically.
__device__ __forceinline__ float norm_bwd_elem(float g, bool keep, float dsc, float xv, float mu, float k0, float k1,
                                               float k2) {
    float dz = keep ? g * dsc : 0.f;
    asm volatile("" : "+v"(dz));
    float a = dz - k1;
    asm volatile("" : "+v"(a));
    float t = (xv - mu) * k2;
    asm volatile("" : "+v"(t));
    float o = __builtin_fmaf(k0, a, -t);
    asm volatile("" : "+v"(o));
    return o;
}

// One LDS-DMA instruction (global_load_lds_dwordx4: 1 KiB per wave, lane i -> lds_base + 16 i) as inline asm.
// Why not __builtin_amdgcn_global_load_lds: hipcc (ROCm 7.2) tracks the builtin as a pending LDS store and emits
// `s_waitcnt vmcnt(0)` in front of the next ds_read_b64_tr_b16 it cannot prove disjoint - right behind every issue,
// which drains a multi-stage ring once per step (seen in the ISA of the wgrad kernels; plain ds_read_b128 reads are not
// affected).  The asm form is invisible to that pass: the kernel orders DMA against reads itself (counted vmcnt +
// barrier).  M0 carries the LDS base (must be wave-uniform); kernels that use this must not use M0 otherwise.
__device__ __forceinline__ void lds_dma16(const void* g, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_base) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}

// Epilogue of the MFMA 16x16x32 kernels whose first operand is the weight fragment, 16-bit output: lane (l15 = pixel, g4) holds
// channels 4 g4 .. + 3 of each 16-channel column tile.  The lanes g4 and g4 ^ 1 (16 apart) trade one accumulator of a column
// tile PAIR (v_permlane16_swap: one instruction per register) - afterwards a lane holds EIGHT consecutive channels of its
// pixel, tile `even` from the even lane's side, tile `odd` from the odd lane's - and stores 16 bytes: half the store
// instructions, 64 contiguous bytes per pixel and instruction instead of 32 (the store shapes alone: 3.7 against 5.5 TB/s,
// tools/probes/store_pattern.hip).  v[0..7] on return: the lane's 8 channels, first channel = (g4 odd ? 16 : 0) + 4 (g4 & 2)
// of the 32-channel pair.  Every lane of the wave must call this (the swap is a cross-lane operation).
// (Inline asm: through __builtin_amdgcn_permlane16_swap hipcc 7.2 loses the instruction's second output in unrolled code.)
typedef float mmh_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void pair_swap8(const mmh_f32x4& even, const mmh_f32x4& odd, float* v) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float lo = even[r], hi = odd[r];
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
        v[r] = lo;
        v[4 + r] = hi;
    }
}
template <bool H16>
__device__ __forceinline__ void store8_lp16(char* dst, const float* v) {
    if (H16) {
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)v[e];
        *reinterpret_cast<h8*>(dst) = o;
    } else {
        typedef __bf16 b8 __attribute__((ext_vector_type(8)));
        b8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
        *reinterpret_cast<b8*>(dst) = o;
    }
}

// Partial InstanceNorm statistics of a conv output tile from the epilogue of an MFMA 16x16x32 kernel whose FIRST operand is the
// weight fragment (lane (l15 = pixel, g4): channels 4 g4 + r, r < 4, of each of its NJ 16-channel column tiles, NV pixel
// rows per lane): count / mean / M2 of the wave's 16 NV pixels per channel, in the partial layout mmh_norm_stats_merge
// [_finalize] reduces ([.][3][N]: n, mean, M2).  val(i, j, r) = the value AS STORED (bias added, rounded to 16 bits) of row i,
// column tile j, register r.  Per lane two passes over its NV values per channel slot (4 j + r), then four equal-count Chan
// merges across the 16 pixel lanes as a reduce-scatter (ds_swizzle, xor 8 / 4 / 2 / 1): while a lane still holds more than
// one slot it keeps the half whose index bit matches its lane bit and hands the other half over; with one slot left the two
// partners merge and both keep the result (the lane whose remaining low bits are 0 writes).  sp: the partial's `n` row at
// this wave's first channel + 4 g4.  conv_lp16h2_kernel carries the NV = 8, NJ = 4 instance of the same scheme inline.
#define MMH_SWZ_(val, s) __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, val), 0x1f | ((s) << 10)))
template <int NS, int S>
__device__ __forceinline__ void wave_stats_step(float* m, float* q, int& ns, bool bit, float w) {
    if (ns > 1) {
        const int h = ns / 2;
#pragma unroll
        for (int c = 0; c < NS / 2; ++c) {
            if (c < h) {
                const float km = bit ? m[h + c] : m[c], sm = bit ? m[c] : m[h + c];
                const float kq = bit ? q[h + c] : q[c], sq = bit ? q[c] : q[h + c];
                const float om = MMH_SWZ_(sm, S), oq = MMH_SWZ_(sq, S), dl = om - km;
                q[c] = kq + oq + dl * dl * w;
                m[c] = 0.5f * (km + om);
            }
        }
        ns = h;
    } else {
        const float om = MMH_SWZ_(m[0], S), oq = MMH_SWZ_(q[0], S), dl = om - m[0];
        q[0] = q[0] + oq + dl * dl * w;
        m[0] = 0.5f * (m[0] + om);
    }
}
template <int NV, int NJ, typename F>
__device__ __forceinline__ void wave_tile_stats(F val, int l15, float* sp, int N) {
    constexpr int NS = 4 * NJ;
    float m[NS], q[NS];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v[NV], mean = 0.f, qq = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) { v[i] = val(i, j, r); mean += v[i]; }
            mean *= 1.f / NV;
#pragma unroll
            for (int i = 0; i < NV; ++i) qq = __builtin_fmaf(v[i] - mean, v[i] - mean, qq);
            m[4 * j + r] = mean; q[4 * j + r] = qq;
        }
    int ns = NS;
    wave_stats_step<NS, 8>(m, q, ns, (l15 & 8) != 0, 0.5f * NV);
    wave_stats_step<NS, 4>(m, q, ns, (l15 & 4) != 0, 1.0f * NV);
    wave_stats_step<NS, 2>(m, q, ns, (l15 & 2) != 0, 2.0f * NV);
    wave_stats_step<NS, 1>(m, q, ns, (l15 & 1) != 0, 4.0f * NV);
    // the slot this lane ends with: its lane bits, high to low, over the steps that still split (log2 NS of them)
    constexpr int SPLITS = NS >= 16 ? 4 : (NS >= 8 ? 3 : (NS >= 4 ? 2 : 1));
    const int slot = l15 >> (4 - SPLITS);
    const bool writer = (l15 & ((1 << (4 - SPLITS)) - 1)) == 0;
    if (writer) {
        const int co = (slot >> 2) * 16 + (slot & 3);
        sp[co] = 16.f * NV;
        sp[N + co] = m[0];
        sp[2 * N + co] = q[0];
    }
}

// Winograd F(6x6,3x3) transforms (wino6.hip); tiles = B * ceil(H/6) * ceil(W/6), 64 planes
int wino6_weights(const float* w, float* U, int Cin, int Cout, int flip_transpose, hipStream_t st);
int wino6_weights_multi(const long long* table, int n, long long total_blocks, hipStream_t st);
int wino6_input(const float* x, float* V, int B, int H, int W, int C, int reflect, int xcd, hipStream_t st);
int wino6_output(const float* M, float* y, const float* bias, int B, int H, int W, int C, int act, float* stats,
                 int fold, hipStream_t st);
int wino6_dy(const float* dy, float* Yh, int B, int H, int W, int C, hipStream_t st);
int wino6_input_normact(const float* x, float* V, int B, int H, int W, int C, int reflect, int xcd,
                        const float* scale, const float* shift, int groups, int relu, float drop_p,
                        const uint32_t* drows, hipStream_t st);
int wino6_input_dy_normbwd(const float* g, const float* x, float* V, float* Yh, int B, int H, int W, int C, int xcd,
                           int fold, const float* mean, const float* invstd, const float* gamma, const float* s1,
                           const float* s2, double count, const float* scale, const float* shift,
                           const uint32_t* drows, int groups, int relu, float drop_p, hipStream_t st);
int wino6_input_dy(const float* dy, float* V, float* Yh, int B, int H, int W, int C, int xcd, int fold,
                   hipStream_t st);
int wino6_dw(const float* dU, float* dw, int Cin, int Cout, int accumulate, hipStream_t st);
// 16-bit wgrad of the 3x3 stride-1 stack with all nine taps resident (wgrad_lp16t.hip)
bool wgrad_lp16t_supported(const mmh_conv_desc* d);
int wgrad_lp16t_splits(const mmh_conv_desc* d);
int launch_wgrad_lp16t(const mmh_conv_desc* d, const void* x16, const void* dy16, float* slab, const void* zeros,
                       hipStream_t st);
// fp32 dgrad of the 3x3 stride-2 convs 64 -> 128 / 128 -> 256 with the dy halo resident in LDS (dgrad_s2.hip)
bool dgrad_s2_halo_ok(const mmh_conv_desc* d, int dx_cs, int act);
int launch_dgrad_s2_halo(const mmh_conv_desc* d, const void* dy, const void* w, const void* bias, void* dx, int dx_cs,
                         int act, hipStream_t st);
extern int g_dgrad_s2_halo, g_dgrad_s2_dbg;
// fp32 wgrad of the 3x3 stride-2 convs as a stream down a column strip of dy (wgrad_s2.hip)
bool wgrad_s2_strip_ok(const mmh_conv_desc* d);
size_t wgrad_s2_strip_ws_bytes(const mmh_conv_desc* d);
int launch_wgrad_s2_strip(const mmh_conv_desc* d, const void* x, const void* dy, void* dw, void* ws, size_t ws_bytes,
                          int accumulate, hipStream_t st);
extern int g_wgrad_s2_strip;
// fp32 fprop of the 7x7 stems from an LDS-resident input halo (conv_stem_f32.hip)
bool stem_f32_ok(const mmh_conv_desc* d);
int stem_f32_stats_chunks(const mmh_conv_desc* d);
int launch_stem_f32(const mmh_conv_desc* d, const void* x, const void* w, const void* bias, void* y, int act, float* stats,
                    hipStream_t st);
extern int g_stem_f32, g_stem_f32_dbg, g_stem_f32_levels;
// fp32 Winograd-domain wgrad GEMMs as a three-stage LDS-DMA ring (wino_wgrad_dma.hip)
bool wino_wgrad_dma_ok(int64_t tiles, int Cin, int Cout, int nbatch);
size_t wino_wgrad_dma_ws_bytes(int64_t tiles, int Cin, int Cout, int nbatch);
int launch_wino_wgrad_dma(const float* V, const float* Yh, int64_t tiles, int Cin, int Cout, int nbatch, float* ws, float* dU,
                          hipStream_t st);
extern int g_wino_wgrad_dma;
// dw[b][i] (+)= sum over a batch's split-K slabs, fixed order (slab_reduce.hip)
int launch_slab_reduce(const float* slab, float* dw, int64_t n4_total, int splits, int accumulate, int64_t n4, hipStream_t st);
// 16-bit fprop of the 3x3 stride-2 convs with register-resident weights and an LDS-resident input halo (conv_s2_lp16.hip)
bool conv_s2f_ok(const mmh_conv_desc* d, int mode);
bool conv_s1f_ok(const mmh_conv_desc* d, int mode);     // its stride-1 form: 64 -> 64, zero padding, mode 0 fprop | 1 dgrad
int conv_s2f_stats_chunks(const mmh_conv_desc* d);
int launch_conv_s2f(const mmh_conv_desc* d, const void* x16, const void* w16, const void* bias, void* y, int y_is16, int act,
                    const void* zeros, hipStream_t st, float* stats = nullptr, int mode = 0);
bool conv_s2d_ok(const mmh_conv_desc* d, int mode);
int launch_conv_s2d(const mmh_conv_desc* d, const void* g16, const void* w16, const void* bias, void* dx, int dx_is16, int act,
                    const void* zeros, hipStream_t st);
extern int g_lp16_s2f;
extern int g_lp16_persist;
extern int g_slab_reduce_par;
extern int g_wino6_vec;
extern int g_lp16_shape;
extern int g_lp16_tap_inner;
extern int g_lp16_dbg;
extern int g_lp16_wgrad_ring;
extern int g_lp16_wgrad_s2;     // 1: the stride-2 3x3 wgrads on the nine-tap halo kernel (wgrad_lp16t.hip), 0: flat rows
extern int g_pw_v2;
extern int g_col_chunks, g_row_chunks;   // workgroups per launch the column-reduce / row kernels aim for

}  // namespace mmh

#define MMH_REQUIRE(cond, ...)                      \
    do {                                            \
        if (!(cond)) return mmh::fail(__VA_ARGS__); \
    } while (0)
