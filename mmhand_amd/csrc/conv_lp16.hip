// 16-bit (bf16 / fp16) implicit-GEMM convolution for the 3x3 stride-1 stack, gfx950, second generation.
//
// What differs from conv_igemm_bf16_body (conv_igemm.hip), which reads fp32 activations, converts
// them in registers and stages 128x128x64 tiles through VGPRs with two barriers per k-step:
//   * BOTH operands are 16-bit in HBM with the contraction index contiguous - activations
//     [B][H][W][C] (a 16-bit twin written by the producer or by mmh_cvt_lp16), weights
//     [tap][N][K] (mmh_prep_weights_bf16/_fp16) - so both go global -> LDS by LDS-DMA
//     (global_load_lds_dwordx4, 1 KiB per wave instruction, no VGPR staging, no ds_write);
//   * 256 x 256 x 64 block tile, 512 threads = 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 =
//     4 x 2 MFMA 32x32x16 tiles (128 accumulator VGPRs): 32 B/clk/CU of L2 traffic at the full
//     MFMA rate (the 128^2 tile needs 64);
//   * two LDS stages of 64 KiB (1 workgroup per CU, 2 waves per SIMD), ONE barrier per k-step:
//     the DMA of k-step i+1 is in flight while the MFMAs of k-step i run;
//   * LDS rows are 128 B (64 elements) with an XOR swizzle of the 16-byte chunk index by
//     (row >> 1) & 7: every ds_read_b128 lane group touches 16 distinct (row parity, chunk) bank
//     sets - conflict-free - and, as the DMA writes LDS linearly (lane i -> base + 16 i), the swizzle
//     is applied to the per-lane GLOBAL source address (MI355X guide: swizzle both sides or neither).
// The gather (zero / reflect padding, tap offsets) is folded into the per-lane source address of the
// A-operand DMA; out-of-image taps and the M tail read a zero page.  k order: tap outer, 64-channel
// chunk inner, so a lane recomputes its 4 source pixels once per tap.
//
// GEMM view: M = B*H*W output pixels, N output channels (rows of the weight operand), K = taps * C.
// Requires C % 64 == 0, N % 256 == 0 (the 256 / 512-channel PATBlock and Discriminator trunks).
#include <algorithm>
#include <type_traits>
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int TBM = 256, TBN = 256, TBK = 64;
constexpr int ROWB = TBK * 2;                 // 128 bytes per LDS row
constexpr int STAGE = (TBM + TBN) * ROWB;     // 64 KiB

struct LpConvKP {
    const char* x;          // 16-bit activations, pixel stride cs elements
    const char* w;          // 16-bit weights [tap][N][K]
    const char* zeros;      // >= 128 zero bytes
    float* y;               // fp32 output [M][y_cs] (y16 == nullptr) ...
    char* y16;              // ... or 16-bit output [M][y_cs]
    const float* bias;
    int B, H, W, C, cs;     // input geometry (same spatial size out: stride 1, 'same' padding)
    int N, y_cs;
    int tap_sign;           // source pixel of tap (kh, kw) = output pixel + tap_sign * (kh - 1, kw - 1):
                            // +1 correlation (fprop), -1 flipped filter (dgrad)
    int reflect;            // mirror the source pixel into the image (else zero outside)
    int act, h16;
    int MT, NT;             // row / column tiles
    int tap_inner;          // k order: 1 = (channel chunk, tap), 0 = (tap, channel chunk)
    int dbg;                // timing-only ablation bits (mmh_set_option "lp16_dbg"): results wrong
    float* stats;           // conv_lp16h2_kernel fprop: per (image, half tile, channel) count / mean / M2 of the stored
                            // outputs, [B][chunks][3][N] (mmh_norm_stats_merge layout), or nullptr
    const float* addend;    // conv_lp16h2_kernel, fp32 output: y += addend (same [M][y_cs] layout) - the other gradient
                            // of a tensor with two consumers, added in the dgrad's epilogue instead of by a pass of its own
};

template <bool H16>
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
    if (H16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0,
                                                      0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == MMH_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == MMH_ACT_TANH) return tanhf(v);
    return v;
}

typedef __attribute__((address_space(3))) void* lds_vp;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Four consecutive output channels of one pixel from one lane (the accumulator layout of an MFMA 16x16x32 whose FIRST
// operand is the weight fragment: row = channel 4 g4 + r, column = pixel l15): bias, activation, one 8- or 16-byte store.
template <bool H16>
__device__ __forceinline__ void store4(float* y, char* y16, size_t elem, f32x4 v, const float* bv, int act) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float t = v[r] + bv[r];
        v[r] = act == MMH_ACT_RELU ? (t > 0.f ? t : 0.f) : (act == MMH_ACT_TANH ? tanhf(t) : t);
    }
    if (y16) {
        if (H16) {
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            h4 o = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            *reinterpret_cast<h4*>(y16 + elem * 2) = o;
        } else {
            typedef __bf16 b4 __attribute__((ext_vector_type(4)));
            b4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            *reinterpret_cast<b4*>(y16 + elem * 2) = o;
        }
    } else {
        *reinterpret_cast<f32x4*>(y + elem) = v;
    }
}
template <bool H16>
__device__ __forceinline__ f32x4 mfma16s(bf16x8 a, bf16x8 b, f32x4 c) {
    if (H16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0,
                                                      0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// Same tile, staging and swizzle on the 16x16x32 MFMA (the chip holds a higher clock on this shape:
// MI355X guide, DVFS item 7): wave tile 128 x 64 = 8 x 4 tiles of 16x16, 4 accumulator VGPRs each.
// A / B fragment of lane l: row l & 15, k = 8 (l >> 4) .. +7 of a 32-deep step, i.e. logical chunk
// 4*s32 + (l >> 4); C/D: col = l & 15, row = 4 (l >> 4) + reg.
template <bool H16>
__global__ void __launch_bounds__(512, 2) conv_lp16s_kernel(const LpConvKP p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wr = wave >> 2, wc = wave & 3;
    const int per_xcd = (p.MT * p.NT + 7) / 8;
    const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= p.MT * p.NT) return;
    const int mt = tile / p.NT, nt = tile - mt * p.NT;
    const int m0 = mt * TBM, n0 = nt * TBN;
    const int M = p.B * p.H * p.W;
    int a_pix[4], a_hw[4];
    unsigned b_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (wave * 4 + j) * 8 + (lane >> 3);
        const unsigned q = (unsigned)((lane & 7) ^ ((r >> 1) & 7));
        const int m = m0 + r;
        const int b = m / (p.H * p.W);
        const int rem = m - b * (p.H * p.W);
        const int oh = rem / p.W, ow = rem - oh * p.W;
        a_pix[j] = m < M ? m : -1;
        a_hw[j] = (oh << 16) | ow;
        b_off[j] = (unsigned)(n0 + r) * (unsigned)p.C * 2u + q * 16u;
    }
    const int KC = p.C / TBK;
    const int nk = 9 * KC;
    unsigned a_off[4];
    auto set_tap = [&](int t) {
        const int kh = t / 3, kw = t - 3 * kh;
        const int dh = p.tap_sign * (kh - 1), dw = p.tap_sign * (kw - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int oh = a_hw[j] >> 16, ow = a_hw[j] & 0xffff;
            int ih = oh + dh, iw = ow + dw;
            bool ok = a_pix[j] >= 0;
            if (p.reflect) {
                ih = ih < 0 ? -ih : ih;
                iw = iw < 0 ? -iw : iw;
                ih = ih >= p.H ? 2 * (p.H - 1) - ih : ih;
                iw = iw >= p.W ? 2 * (p.W - 1) - iw : iw;
            } else {
                ok = ok && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            }
            const int r = (wave * 4 + j) * 8 + (lane >> 3);
            const unsigned q = (unsigned)((lane & 7) ^ ((r >> 1) & 7));
            const int src = a_pix[j] + (ih - oh) * p.W + (iw - ow);
            a_off[j] = ok ? (unsigned)src * (unsigned)p.cs * 2u + q * 16u : 0xffffffffu;
        }
    };
    // k order (mmh_set_option "lp16_tap_inner"): tap outer / 64-channel chunk inner by default - a lane
    // recomputes its 4 source pixels once per tap.  Chunk-outer order keeps the nine taps of a chunk in
    // L2 (FETCH_SIZE is 9.2x the input with tap-outer order: each tap streams 8 MiB per XCD through a
    // 4 MiB L2 and is served by the Infinity Cache) but pays the source-pixel arithmetic every k-step:
    // measured 7 % SLOWER (A/B in one process, tools/ab_lp16_shape.py) - the kernel is not fetch-bound.
    auto issue = [&](int ks, int stage) {
        int kc, t;
        if (p.tap_inner) { kc = ks / 9; t = ks - kc * 9; set_tap(t); }
        else { t = ks / KC; kc = ks - t * KC; if (kc == 0) set_tap(t); }
        char* sA = smem + stage * STAGE;
        char* sB = sA + TBM * ROWB;
        const unsigned kb = (unsigned)kc * (TBK * 2);
        const char* wbase = p.w + (size_t)t * p.N * p.C * 2 + kb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const char* g = a_off[j] != 0xffffffffu ? p.x + a_off[j] + kb : p.zeros + (lane & 7) * 16;
            __builtin_amdgcn_global_load_lds(g, (lds_vp)(sA + (wave * 4 + j) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds(wbase + b_off[j], (lds_vp)(sB + (wave * 4 + j) * 1024), 16, 0, 0);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    // rows read by this lane: A wr*128 + i*16 + l15, B wc*64 + j*16 + l15: key (row >> 1) & 7 = (l15 >> 1)
    const unsigned key = (unsigned)(l15 >> 1);
    const unsigned a_base = (unsigned)(wr * 128 + l15) * ROWB;
    const unsigned b_base = (unsigned)(TBM + wc * 64 + l15) * ROWB;

    issue(0, 0);
    for (int ks = 0; ks < nk; ++ks) {
        __builtin_amdgcn_s_waitcnt(0x0070);     // vmcnt(0) lgkmcnt(0): the LDS-DMA of the previous iteration
        __syncthreads();
        if (ks + 1 < nk && !(p.dbg & 1)) issue(ks + 1, (ks + 1) & 1);      // dbg 1: no DMA after the first stage
        const char* st = smem + (ks & 1) * STAGE;
#pragma unroll
        for (int s32 = 0; s32 < 2; ++s32) {
            const unsigned sw = ((unsigned)(4 * s32 + g4) ^ key) << 4;
            bf16x8 af[8], bfr[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const bf16x8*>(st + a_base + sw + i * (16 * ROWB));
#pragma unroll
            for (int j = 0; j < 4; ++j)      // dbg 2: B fragments not read from LDS
                bfr[j] = (p.dbg & 2) ? af[j] : *reinterpret_cast<const bf16x8*>(st + b_base + sw + j * (16 * ROWB));
            __builtin_amdgcn_s_setprio(1);
            if (!(p.dbg & 4))               // dbg 4: no MFMAs
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16s<H16>(af[i], bfr[j], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wr * 128 + i * 16 + 4 * g4 + r;
            if (m >= M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wc * 64 + j * 16 + l15;
                float v = acc[i][j][r];
                if (p.bias) v += p.bias[n];
                v = act_apply(v, p.act);
                if (p.y16) {
                    if (H16) reinterpret_cast<_Float16*>(p.y16)[(size_t)m * p.y_cs + n] = (_Float16)v;
                    else reinterpret_cast<__bf16*>(p.y16)[(size_t)m * p.y_cs + n] = (__bf16)v;
                } else {
                    p.y[(size_t)m * p.y_cs + n] = v;
                }
            }
        }
}


// conv_lp16s_kernel with the LDS fragment reads software-pipelined INTO the MFMA stream: an ablation
// (tools/ablate_lp16.py, 512->512: 628 us = 452 us without the DMA = 201 us of fragment reads + 247 us
// of MFMAs) showed the two phases running back to back - both waves of a SIMD leave the barrier in
// lockstep, read, wait, then multiply.  Here a wave's 32-deep step is: 4 MFMAs on A fragment i, then the
// ds_read that refills fragment i for the NEXT 32-deep step (rolling reuse, no extra A registers; the B
// fragments alternate between two sets), so the reads hide under the MFMAs of the same wave.  The one
// barrier per k-step moves to the middle of the step: by then stage ks is fully read (the DMA of k-step
// ks+2 may overwrite it) and stage ks+1 has had a whole k-step to land.
template <bool H16>
__global__ void __launch_bounds__(512, 2) conv_lp16p_kernel(const LpConvKP p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wr = wave >> 2, wc = wave & 3;
    const int per_xcd = (p.MT * p.NT + 7) / 8;
    const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= p.MT * p.NT) return;
    const int mt = tile / p.NT, nt = tile - mt * p.NT;
    const int m0 = mt * TBM, n0 = nt * TBN;
    const int M = p.B * p.H * p.W;
    int a_pix[4], a_hw[4];
    unsigned b_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (wave * 4 + j) * 8 + (lane >> 3);
        const unsigned q = (unsigned)((lane & 7) ^ ((r >> 1) & 7));
        const int m = m0 + r;
        const int b = m / (p.H * p.W);
        const int rem = m - b * (p.H * p.W);
        const int oh = rem / p.W, ow = rem - oh * p.W;
        a_pix[j] = m < M ? m : -1;
        a_hw[j] = (oh << 16) | ow;
        b_off[j] = (unsigned)(n0 + r) * (unsigned)p.C * 2u + q * 16u;
    }
    const int KC = p.C / TBK;
    const int nk = 9 * KC;
    unsigned a_off[4];
    auto set_tap = [&](int t) {
        const int kh = t / 3, kw = t - 3 * kh;
        const int dh = p.tap_sign * (kh - 1), dw = p.tap_sign * (kw - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int oh = a_hw[j] >> 16, ow = a_hw[j] & 0xffff;
            int ih = oh + dh, iw = ow + dw;
            bool ok = a_pix[j] >= 0;
            if (p.reflect) {
                ih = ih < 0 ? -ih : ih;
                iw = iw < 0 ? -iw : iw;
                ih = ih >= p.H ? 2 * (p.H - 1) - ih : ih;
                iw = iw >= p.W ? 2 * (p.W - 1) - iw : iw;
            } else {
                ok = ok && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            }
            const int r = (wave * 4 + j) * 8 + (lane >> 3);
            const unsigned q = (unsigned)((lane & 7) ^ ((r >> 1) & 7));
            const int src = a_pix[j] + (ih - oh) * p.W + (iw - ow);
            a_off[j] = ok ? (unsigned)src * (unsigned)p.cs * 2u + q * 16u : 0xffffffffu;
        }
    };
    // k order (mmh_set_option "lp16_tap_inner"): tap outer / 64-channel chunk inner by default - a lane
    // recomputes its 4 source pixels once per tap.  Chunk-outer order keeps the nine taps of a chunk in
    // L2 (FETCH_SIZE is 9.2x the input with tap-outer order: each tap streams 8 MiB per XCD through a
    // 4 MiB L2 and is served by the Infinity Cache) but pays the source-pixel arithmetic every k-step:
    // measured 7 % SLOWER (A/B in one process, tools/ab_lp16_shape.py) - the kernel is not fetch-bound.
    auto issue = [&](int ks, int stage) {
        int kc, t;
        if (p.tap_inner) { kc = ks / 9; t = ks - kc * 9; set_tap(t); }
        else { t = ks / KC; kc = ks - t * KC; if (kc == 0) set_tap(t); }
        char* sA = smem + stage * STAGE;
        char* sB = sA + TBM * ROWB;
        const unsigned kb = (unsigned)kc * (TBK * 2);
        const char* wbase = p.w + (size_t)t * p.N * p.C * 2 + kb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const char* g = a_off[j] != 0xffffffffu ? p.x + a_off[j] + kb : p.zeros + (lane & 7) * 16;
            __builtin_amdgcn_global_load_lds(g, (lds_vp)(sA + (wave * 4 + j) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds(wbase + b_off[j], (lds_vp)(sB + (wave * 4 + j) * 1024), 16, 0, 0);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    // rows read by this lane: A wr*128 + i*16 + l15, B wc*64 + j*16 + l15: key (row >> 1) & 7 = (l15 >> 1)
    const unsigned key = (unsigned)(l15 >> 1);
    const unsigned a_base = (unsigned)(wr * 128 + l15) * ROWB;
    const unsigned b_base = (unsigned)(TBM + wc * 64 + l15) * ROWB;

    // byte offsets of this lane's fragments inside a stage for the two 32-deep halves of a k-step
    const unsigned sw0 = ((unsigned)g4 ^ key) << 4, sw1 = ((unsigned)(4 + g4) ^ key) << 4;
    bf16x8 af[8], b0[4], b1[4];
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();                    // (waits for both: the first step has no MFMAs to hide under)
#pragma unroll
    for (int j = 0; j < 4; ++j) b0[j] = *reinterpret_cast<const bf16x8*>(smem + b_base + sw0 + j * (16 * ROWB));
#pragma unroll
    for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const bf16x8*>(smem + a_base + sw0 + i * (16 * ROWB));
    for (int ks = 0; ks < nk; ++ks) {
        const char* st = smem + (ks & 1) * STAGE;
        const char* sn = smem + ((ks + 1) & 1) * STAGE;
        // ---- half 0: multiply (ks, 0) out of af / b0 while the fragments of (ks, 1) stream in
#pragma unroll
        for (int j = 0; j < 4; ++j) b1[j] = *reinterpret_cast<const bf16x8*>(st + b_base + sw1 + j * (16 * ROWB));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16s<H16>(af[i], b0[j], acc[i][j]);
            af[i] = *reinterpret_cast<const bf16x8*>(st + a_base + sw1 + i * (16 * ROWB));
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);      // 4 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
        }
        // stage ks is read, stage ks+1 has landed.  The wait is explicit: the compiler does not count the
        // LDS-DMA of an earlier loop iteration against this barrier (seen in the ISA, and as a race)
        __builtin_amdgcn_s_waitcnt(0x0070);     // vmcnt(0) lgkmcnt(0)
        __syncthreads();
        if (ks + 2 < nk) issue(ks + 2, ks & 1);
        // ---- half 1: multiply (ks, 1) out of af / b1 while the fragments of (ks+1, 0) stream in
        const bool more = ks + 1 < nk;
        if (more) {
#pragma unroll
            for (int j = 0; j < 4; ++j) b0[j] = *reinterpret_cast<const bf16x8*>(sn + b_base + sw0 + j * (16 * ROWB));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16s<H16>(af[i], b1[j], acc[i][j]);
            if (more) af[i] = *reinterpret_cast<const bf16x8*>(sn + a_base + sw0 + i * (16 * ROWB));
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    }

#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wr * 128 + i * 16 + 4 * g4 + r;
            if (m >= M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wc * 64 + j * 16 + l15;
                float v = acc[i][j][r];
                if (p.bias) v += p.bias[n];
                v = act_apply(v, p.act);
                if (p.y16) {
                    if (H16) reinterpret_cast<_Float16*>(p.y16)[(size_t)m * p.y_cs + n] = (_Float16)v;
                    else reinterpret_cast<__bf16*>(p.y16)[(size_t)m * p.y_cs + n] = (__bf16)v;
                } else {
                    p.y[(size_t)m * p.y_cs + n] = v;
                }
            }
        }
}




// ---------------------------------------------------------------------------------------------
// conv_lp16h_kernel: the stride-1 3x3 kernel with the activation tile held in LDS ONCE for all nine
// taps.  The M tile is a 16 x 16 pixel block of one image; per 64-channel chunk its 18 x 18 halo
// (324 pixel rows x 128 B = 40.5 KiB) is brought in by LDS-DMA one whole chunk (nine k-steps) ahead and
// the nine taps read it at shifted rows - the A operand crosses L2 -> LDS once instead of nine times
// (conv_lp16p_kernel: FETCH_SIZE 9.2x the input, and inside the training step, where the input does
// not come from a warm cache, it is the DMA that the k-step waits for).  Only the weights stream per
// k-step (32 KiB, L2 resident).  LDS: 2 halo stages + 2 weight stages = 145 KiB.
//   k order: 64-channel chunk outer, tap inner.  Halo row of output pixel (py, px), tap offset (dh, dw) in
//   {0,1,2}^2: (py + dh) * 18 + px + dw; fprop (dh, dw) = (kh, kw), dgrad (2 - kh, 2 - kw).
//   Rows keep the (row >> 1) & 7 chunk swizzle: a fragment's 16 lanes read 16 consecutive halo rows, which
//   from any starting row pair up as (even, odd) rows with equal keys - conflict-free as before.
// The fragment reads are pipelined into the MFMA stream as in conv_lp16p_kernel.
constexpr int HT = 16;                      // tile edge (output pixels)
constexpr int HW_ = HT + 2;                 // halo edge
constexpr int HROWS = HW_ * HW_;            // 324 halo pixel rows
constexpr int HSTAGE_A = ((HROWS * ROWB + 1023) / 1024) * 1024;     // 41984
constexpr int HSTAGE_B = TBN * ROWB;                                 // 32768
constexpr int HROUNDS = (HROWS + 63) / 64;                           // 6 DMA instructions per wave and chunk

template <bool H16>
__global__ void __launch_bounds__(512, 2) conv_lp16h_kernel(const LpConvKP p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const sAh = smem;                         // [2][HSTAGE_A]
    char* const sBh = smem + 2 * HSTAGE_A;          // [2][HSTAGE_B]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wr = wave >> 2, wc = wave & 3;
    const int per_xcd = (p.MT * p.NT + 7) / 8;
    const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= p.MT * p.NT) return;
    const int mt = tile / p.NT, nt = tile - mt * p.NT;
    const int n0 = nt * TBN;
    const int TX = (p.W + HT - 1) / HT, TY = (p.H + HT - 1) / HT;
    const int b = mt / (TX * TY);
    const int trem = mt - b * (TX * TY);
    const int ty = trem / TX, tx = trem - ty * TX;
    const int oh0 = ty * HT, ow0 = tx * HT;

    // halo DMA roles: round rd covers halo rows rd*64 .. +63; wave w rows rd*64 + w*8 + lane/8
    unsigned a_off[HROUNDS];
#pragma unroll
    for (int rd = 0; rd < HROUNDS; ++rd) {
        const int r = rd * 64 + wave * 8 + (lane >> 3);
        const int hy = r / HW_, hx = r - hy * HW_;
        int ih = oh0 + hy - 1, iw = ow0 + hx - 1;
        bool ok = r < HROWS;
        if (p.reflect) {
            ih = ih < 0 ? -ih : ih;
            iw = iw < 0 ? -iw : iw;
            ih = ih >= p.H ? 2 * (p.H - 1) - ih : ih;
            iw = iw >= p.W ? 2 * (p.W - 1) - iw : iw;
        }
        ok = ok && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
        const unsigned q = (unsigned)((lane & 7) ^ ((r >> 1) & 7));
        a_off[rd] = ok ? (unsigned)((b * p.H + ih) * p.W + iw) * (unsigned)p.cs * 2u + q * 16u
                       : (r < HROWS ? 0xfffffffeu : 0xffffffffu);        // ...fe: zero page, ...ff: no row
    }
    unsigned b_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (wave * 4 + j) * 8 + (lane >> 3);
        const unsigned q = (unsigned)((lane & 7) ^ ((r >> 1) & 7));
        b_off[j] = (unsigned)(n0 + r) * (unsigned)p.C * 2u + q * 16u;
    }
    const int KC = p.C / TBK;
    const int nk = 9 * KC;
    auto issue_halo = [&](int kc) {
        char* sA = sAh + (kc & 1) * HSTAGE_A;
        const unsigned kb = (unsigned)kc * (TBK * 2);
#pragma unroll
        for (int rd = 0; rd < HROUNDS; ++rd) {
            if (a_off[rd] != 0xffffffffu) {
                const char* g = a_off[rd] != 0xfffffffeu ? p.x + a_off[rd] + kb : p.zeros + (lane & 7) * 16;
                __builtin_amdgcn_global_load_lds(g, (lds_vp)(sA + (rd * 8 + wave) * 1024), 16, 0, 0);
            }
        }
    };
    auto issue_w = [&](int ks) {
        const int kc = ks / 9, t = ks - kc * 9;
        char* sB = sBh + (ks & 1) * HSTAGE_B;
        const char* wbase = p.w + ((size_t)t * p.N * p.C + (size_t)kc * TBK) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds(wbase + b_off[j], (lds_vp)(sB + (wave * 4 + j) * 1024), 16, 0, 0);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    // B fragments: rows wc*64 + j*16 + l15 of the weight stage (key (row >> 1) & 7 = l15 >> 1)
    const unsigned bkey = (unsigned)(l15 >> 1);
    const unsigned b_base = (unsigned)(wc * 64 + l15) * ROWB;
    const unsigned bsw0 = ((unsigned)g4 ^ bkey) << 4, bsw1 = ((unsigned)(4 + g4) ^ bkey) << 4;
    // A fragment i of this lane for tap offset (dh, dw): halo row (wr*8 + i + dh) * 18 + l15 + dw
    auto a_addr = [&](const char* sA, int dh, int dw, int i, int half) -> const char* {
        const unsigned hr = (unsigned)((wr * 8 + i + dh) * HW_ + l15 + dw);
        return sA + hr * ROWB + ((((unsigned)(4 * half + g4)) ^ ((hr >> 1) & 7u)) << 4);
    };
    auto tap_dh = [&](int t) { const int kh = t / 3; return p.tap_sign > 0 ? kh : 2 - kh; };
    auto tap_dw = [&](int t) { const int kw = t - 3 * (t / 3); return p.tap_sign > 0 ? kw : 2 - kw; };

    bf16x8 af[8], b0[4], b1[4];
    issue_halo(0);
    issue_w(0);
    if (nk > 1) issue_w(1);
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();
    {
        const int dh = tap_dh(0), dw = tap_dw(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) b0[j] = *reinterpret_cast<const bf16x8*>(sBh + b_base + bsw0 + j * (16 * ROWB));
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const bf16x8*>(a_addr(sAh, dh, dw, i, 0));
    }
    int kc = 0, t = 0;                  // (chunk, tap) of k-step ks
    for (int ks = 0; ks < nk; ++ks) {
        const char* sA = sAh + (kc & 1) * HSTAGE_A;
        const char* sB = sBh + (ks & 1) * HSTAGE_B;
        const int dh = tap_dh(t), dw = tap_dw(t);
        // ---- half 0: multiply (ks, 0) while the fragments of (ks, 1) stream in
#pragma unroll
        for (int j = 0; j < 4; ++j) b1[j] = *reinterpret_cast<const bf16x8*>(sB + b_base + bsw1 + j * (16 * ROWB));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16s<H16>(b0[j], af[i], acc[i][j]);
            af[i] = *reinterpret_cast<const bf16x8*>(a_addr(sA, dh, dw, i, 1));
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        // ---- middle of the k-step: the weight stage ks is read; stage ks+1 (issued one k-step ago) must have
        // landed.  The halo of the next chunk, issued right after the weights at t == 0, may stay in
        // flight across the barrier of t == 1 (it is needed nine k-steps after its issue): the wait then
        // leaves the newest HROUNDS - 1 loads outstanding (vmcnt counts in issue order; every wave issues
        // HROUNDS - 1 or HROUNDS of them)
        if (t == 1 && kc + 1 < KC) __builtin_amdgcn_s_waitcnt(0x0070 | (HROUNDS - 1));      // vmcnt(5) lgkmcnt(0)
        else __builtin_amdgcn_s_waitcnt(0x0070);                                             // vmcnt(0) lgkmcnt(0)
        __syncthreads();
        if (ks + 2 < nk) issue_w(ks + 2);
        if (t == 0 && kc + 1 < KC) issue_halo(kc + 1);
        // ---- half 1: multiply (ks, 1) while the fragments of (ks+1, 0) stream in
        int kc2 = kc, t2 = t + 1;
        if (t2 == 9) { t2 = 0; ++kc2; }
        const bool more = ks + 1 < nk;
        if (more) {
            const char* sBn = sBh + ((ks + 1) & 1) * HSTAGE_B;
#pragma unroll
            for (int j = 0; j < 4; ++j) b0[j] = *reinterpret_cast<const bf16x8*>(sBn + b_base + bsw0 + j * (16 * ROWB));
        }
        {
            const char* sAn = sAh + (kc2 & 1) * HSTAGE_A;
            const int dh2 = tap_dh(t2), dw2 = tap_dw(t2);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16s<H16>(b1[j], af[i], acc[i][j]);
                if (more) af[i] = *reinterpret_cast<const bf16x8*>(a_addr(sAn, dh2, dw2, i, 0));
            }
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        kc = kc2; t = t2;
    }

    // Epilogue.  The MFMAs above take the WEIGHT fragment as their first operand: D[row = output channel][col = pixel],
    // i.e. lane (l15, g4) holds, per (i, j), the four consecutive channels n0 + wc*64 + j*16 + 4*g4 + 0..3 of pixel
    // (oh0 + wr*8 + i, ow0 + l15): one 8-byte (16-bit output) or 16-byte (fp32) store per accumulator instead of four
    // 2- / 4-byte ones - a quarter of the store instructions of the pixel-major layout, whose 128 scalar stores per lane
    // took longer than a filter row of MFMAs.  The bias is read once per lane, not once per element.
    float bv[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[j][r] = p.bias ? p.bias[n0 + wc * 64 + j * 16 + 4 * g4 + r] : 0.f;
    const int ow = ow0 + l15;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int oh = oh0 + wr * 8 + i;
        if (oh >= p.H || ow >= p.W) continue;
        const size_t m = ((size_t)b * p.H + oh) * p.W + ow;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            store4<H16>(p.y, p.y16, m * p.y_cs + (n0 + wc * 64 + j * 16 + 4 * g4), acc[i][j], bv[j], p.act);
    }
}

// ---------------------------------------------------------------------------------------------
// conv_lp16h2_kernel: conv_lp16h_kernel with the fragment ADDRESS arithmetic taken out of the k-loop.
// conv_lp16h_kernel recomputes, per A fragment and k-step, the halo row of the tap and its swizzle key: its loop body
// holds 142 vector-ALU instructions beside 64 MFMAs per wave, and with two waves per SIMD that is MORE vector issue
// time (2 x 142 x 4 cycles) than the MFMAs leave free (8 of every 16 cycles): the MFMA stream waits on address
// arithmetic.  Here
//   * halo row (hy, hx) sits at LDS row hy * 20 + hx (pitch 20: even, so the row's bank parity is hx & 1) and its
//     16-byte chunks are XOR-ed with hx & 6 - a key that does not depend on hy, so the lane's address depends on the
//     tap only through dw: lane_base[dw] + (wr*8 + i + dh) * 2560 - the row part is an IMMEDIATE.  hx & 6 (not
//     (row >> 1) & 7 as in conv_lp16h_kernel): ds_read_b128 serves a wave in groups {lanes 0-3, 12-15, 20-27}, ...
//     (MI355X guide), i.e. 8 rows with chunk c and the 8 rows between them with chunk c ^ 1; with (row >> 1) & 7 those
//     16 slots are distinct only when the fragment starts on an even row (dw = 0, 2) and 4 of 16 lanes collide on an
//     odd start (the 25 M conflict cycles per launch the round-2 profile could not place); hx & 6 is conflict-free
//     for every start (checked by enumeration over the real lane groups);
//   * six lane-constant A addresses (3 dw x 2 halves of the k-step) and two for B; per k-step the stage offset and
//     dh rows are added as scalars (a handful of vector adds instead of ~100) - the taps stay a run-time loop (unrolled
//     nine-fold the compiler keeps every per-tap DMA address alive and spills);
//   * LDS is addressed through 32-bit local pointers (no 64-bit flat address arithmetic).  SIGN = +1 fprop, -1 dgrad.
// Same tile, weights path, pipelining (fragments of the next half k-step requested while the current half multiplies,
// one barrier per k-step in its middle) and epilogue as conv_lp16h_kernel; LDS 2 x 45 KiB + 2 x 32 KiB = 154 KiB.
constexpr int HP2 = 20;                                     // halo pitch
constexpr int HROWS2 = HW_ * HP2;                           // 360 LDS rows per stage
constexpr int HSTAGE_A2 = HROWS2 * ROWB;                    // 46080 = 45 KiB
constexpr int HROUNDS2 = (HROWS2 + 63) / 64;                // 6

typedef const bf16x8 __attribute__((address_space(3))) * lds_frag_p;
__device__ __forceinline__ bf16x8 lds_frag(unsigned addr) { return *reinterpret_cast<lds_frag_p>(addr); }

template <bool H16, int SIGN, bool FOLD>
__global__ void __launch_bounds__(512, 2) conv_lp16h2_kernel(const LpConvKP p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const sAh = smem;                         // [2][HSTAGE_A2]
    char* const sBh = smem + 2 * HSTAGE_A2;         // [2][HSTAGE_B]
    const int tid = threadIdx.x;
    // wave index as a SCALAR: everything derived from it (DMA destinations, row roles) stays in SGPRs
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wr = wave >> 2, wc = wave & 3;
    // Tile lists: XCD x owns the tiles [x per_xcd, (x + 1) per_xcd); its workgroup `slot` takes tile slot, slot + wpx, ...
    // (wpx = workgroups per XCD).  Launched with one workgroup per tile (wpx = per_xcd) this is the one-tile mapping;
    // launched PERSISTENT (wpx = CUs / 8: mmh_set_option("lp16_persist")) a workgroup walks several tiles, each with its
    // own prologue: 3-8 % faster from 512 tiles up (no second wave of workgroup launches behind the first, no ragged last
    // round).  Prefetching the next tile's first stages during the last k-steps was built as well and added nothing to that
    // (145 against 143 us at 256 -> 256), while its live state spilled the reflect-fold variant (148 -> 201 us): not kept.
    // (The reflect-fold variant walks tile lists as well since it has one fold accumulator per wave and its halo offsets in
    // LDS: 250 registers, nothing in scratch; mmh_set_option("lp16_persist", 2) = every variant but that one.)
    const int per_xcd = (p.MT * p.NT + 7) / 8;
    const int wpx = (int)(gridDim.x >> 3);
    const int tile_end = min(((int)(blockIdx.x & 7) + 1) * per_xcd, p.MT * p.NT);
    for (int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3); tile < tile_end; tile += wpx) {
    const int mt = tile / p.NT, nt = tile - mt * p.NT;
    const int n0 = nt * TBN;
    const int TX = (p.W + HT - 1) / HT, TY = (p.H + HT - 1) / HT;
    const int b = mt / (TX * TY);
    const int trem = mt - b * (TX * TY);
    const int ty = trem / TX, tx = trem - ty * TX;
    const int oh0 = ty * HT, ow0 = tx * HT;

    // halo DMA roles: round rd covers LDS rows rd*64 .. +63 (row = hy * 20 + hx); wave w rows rd*64 + w*8 + lane/8.
    // Source offset of LDS row r with the row's swizzle key in the free low bits (the offset is a multiple of 128 bytes):
    // a lane XORs its own chunk (lane & 7) * 16 into it.  0xfffffffe: zero page, 0xffffffff: no such row.
    auto halo_row_off = [&](int r) -> unsigned {
        const int hy = r / HP2, hx = r - hy * HP2;
        int ih = oh0 + hy - 1, iw = ow0 + hx - 1;
        const bool row = r < HROWS2 && hx < HW_;
        if (p.reflect) {
            ih = ih < 0 ? -ih : ih;
            iw = iw < 0 ? -iw : iw;
            ih = ih >= p.H ? 2 * (p.H - 1) - ih : ih;
            iw = iw >= p.W ? 2 * (p.W - 1) - iw : iw;
        }
        const bool ok = row && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
        return ok ? (unsigned)((b * p.H + ih) * p.W + iw) * (unsigned)p.cs * 2u + (unsigned)(hx & 6) * 16u
                  : (row ? 0xfffffffeu : 0xffffffffu);
    };
    // The plain variants hold their six offsets in registers.  The reflect-fold variant is six registers short of keeping
    // its fold accumulators out of scratch (a scratch reload in the k-loop is followed by s_waitcnt vmcnt(0): a drain of the
    // whole LDS-DMA ring in every fold k-step), so it parks the 384 row offsets in LDS behind the stages (1.5 of the 6 KiB
    // the stages leave) and fetches its six per chunk.
    unsigned a_off[FOLD ? 1 : HROUNDS2];
    unsigned* const sOff = reinterpret_cast<unsigned*>(smem + 2 * HSTAGE_A2 + 2 * HSTAGE_B);
    if (FOLD) {
        if (tid < HROUNDS2 * 64) sOff[tid] = halo_row_off(tid);
        __syncthreads();
    } else {
#pragma unroll
        for (int rd = 0; rd < HROUNDS2; ++rd) {
            const unsigned ro = halo_row_off(rd * 64 + wave * 8 + (lane >> 3));
            a_off[rd] = ro >= 0xfffffffeu ? ro : ro ^ ((unsigned)(lane & 7) * 16u);
        }
    }
    // weight DMA: wave w, round j moves rows (w * 4 + j) * 8 + lane / 8 of the [256][64] tile.  The swizzle key (r >> 1) & 7
    // = (4 j + lane / 16) & 7 splits into a lane part and bit 0 of j: ONE lane offset, ^ 64 (chunk ^ 4) for odd j; the row
    // advance of j is a scalar added to the base pointer
    unsigned b_off0;
    {
        const int r = wave * 32 + (lane >> 3);
        b_off0 = (unsigned)(n0 + r) * (unsigned)p.C * 2u + (unsigned)((lane & 7) ^ ((r >> 1) & 7)) * 16u;
    }
    const int KC = p.C / TBK;
    auto issue_halo = [&](int kc) {
        char* sA = sAh + (kc & 1) * HSTAGE_A2;
        const char* xb = p.x + (size_t)kc * (TBK * 2);
#pragma unroll
        for (int rd = 0; rd < HROUNDS2; ++rd) {
            unsigned ao;
            if (FOLD) {         // one offset at a time (the asm keeps the six reads from being gathered in front)
                ao = sOff[rd * 64 + wave * 8 + (lane >> 3)];
                asm volatile("" : "+v"(ao) :: "memory");
                ao = ao >= 0xfffffffeu ? ao : ao ^ ((unsigned)(lane & 7) * 16u);
            } else {
                ao = a_off[FOLD ? 0 : rd];
            }
            if (ao != 0xffffffffu) {
                const char* g = ao != 0xfffffffeu ? xb + ao : p.zeros + (lane & 7) * 16;
                __builtin_amdgcn_global_load_lds(g, (lds_vp)(sA + (rd * 8 + wave) * 1024), 16, 0, 0);
            }
        }
    };
    auto issue_w = [&](int kc, int t) {             // weight tile of (chunk kc, tap t) -> stage (9 kc + t) & 1
        char* sB = sBh + ((kc + t) & 1) * HSTAGE_B;     // 9 kc + t and kc + t have the same parity
        const char* wbase = p.w + ((size_t)t * p.N * p.C + (size_t)kc * TBK) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds(wbase + (size_t)(j * 8) * p.C * 2 + ((j & 1) ? (b_off0 ^ 64u) : b_off0),
                                             (lds_vp)(sB + (wave * 4 + j) * 1024), 16, 0, 0);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    // lane-constant LDS byte addresses (32-bit) of stage 0: A [dw][half], B [half]; the k-step adds the stage offset
    // and dh rows (scalars), the fragment index i an immediate
    const unsigned lds0 = mmh::lds_addr_of(smem);
    // (named scalars, not an array: a select over array elements comes back from the compiler as a run-time indexed
    // load from a SCRATCH copy of the array)
    auto a_lane = [&](int dw, int hf) -> unsigned {
        const unsigned hx = (unsigned)(dw + l15);
        return lds0 + (unsigned)(wr * 8 * HP2) * ROWB + hx * ROWB + ((((unsigned)(4 * hf + g4)) ^ (hx & 6u)) << 4);
    };
    const unsigned aA00 = a_lane(0, 0), aA01 = a_lane(0, 1), aA10 = a_lane(1, 0), aA11 = a_lane(1, 1), aA20 = a_lane(2, 0),
                   aA21 = a_lane(2, 1);
    unsigned aB[2];
    {
        const unsigned bkey = (unsigned)(l15 >> 1);     // weight rows wc*64 + j*16 + l15: key (row >> 1) & 7 = l15 >> 1
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
            aB[hf] = lds0 + 2 * HSTAGE_A2 + (unsigned)(wc * 64 + l15) * ROWB + ((((unsigned)(4 * hf + g4)) ^ bkey) << 4);
    }
    // The half-0 address of tap (dh, dw) in halo stage st: aA00 + [dw == 1] * d1 + [dw == 2] * d2 + st * HSTAGE_A2 +
    // dh * 2560 - two multiply-adds by 0 / 1 scalars instead of a three-way select (which the compiler turns into
    // scalar branches at the head of every k-step); the half-1 address of the same tap is that ^ 64 (every offset
    // added is a multiple of the 128-byte row).  Taps advance by counters, not by t / 3 and t % 3.
    const unsigned d1 = aA10 - aA00, d2 = aA20 - aA00;
    (void)aA01; (void)aA11; (void)aA21;
    auto a_half0 = [&](int dh, int dw, int st) -> unsigned {
        const unsigned m1 = dw == 1 ? 1u : 0u, m2 = dw == 2 ? 1u : 0u;
        return aA00 + m1 * d1 + m2 * d2 + (unsigned)(st * HSTAGE_A2 + dh * (HP2 * ROWB));
    };
    const int nk = 9 * KC;

    // FOLD (dgrad of a ReflectionPad2d(1) conv; H, W multiples of 16, at least two tiles each way): the gradient of the pad
    // ring, folded back onto rows 1 / H-2 and columns 1 / W-2, is computed HERE with the weight fragments the k-step holds
    // anyway - instead of eight border GEMMs + an add kernel per conv (mmh_conv2d_dgrad_border: 46-70 us, 90 times per step).
    //   ring row -1 -> row 1 (top tiles, taps kh = 0): one extra A fragment (halo row of dy row 0), accumulated by the
    //       waves that own tile rows 0-7 and added to acc[1] after the loop; ring row H -> row H-2 (bottom tiles, kh = 2)
    //       likewise by the waves of rows 8-15 into acc[6];
    //   ring column -1 -> column 1 (left tiles, taps kw = 0): the ring values of the tile's 16 ROWS form one MFMA
    //       column block (lane <-> tile row, its fragment read down the halo column of dy column 0); ring column W ->
    //       column W-2 (right tiles, kw = 2) likewise; after the loop the block goes through LDS to the lanes that hold
    //       pixel column 1 / 14;
    //   the four ring corners (one tap each) are single-lane fragments of the row term.
    // A tile has at most one row term and one column term (two tiles each way), and a workgroup two wave rows: ONE fold
    // accumulator FA per wave - the row term on its own wave row, the column term on the other one (on wr = 0 / 1 for left /
    // right when the tile has no row term).  One accumulator, one fragment, one multiply site per half k-step: the earlier
    // build (row term into acc[1] | acc[6] in place, a second accumulator for the column term) made the register allocator
    // copy acc[6] through temporaries in every fold k-step and spill part of the column accumulator - whose reload was
    // followed by s_waitcnt vmcnt(0), a drain of the LDS-DMA ring - and left no room for the tile loop (534 -> 482 us at
    // 512 -> 512, plain dgrad 441: tools/bench_lp16_fold.py).  Cost: 4 MFMAs on 64 in a third of the k-steps of edge tiles.
    const bool fold_on = FOLD && !(p.dbg & 4);
    const bool t_top = fold_on && !(p.dbg & 16) && ty == 0, t_bot = fold_on && !(p.dbg & 16) && ty == TY - 1;
    const bool f_left = fold_on && !(p.dbg & 8) && tx == 0, f_right = fold_on && !(p.dbg & 8) && tx == TX - 1;
    const bool my_row = (t_top && wr == 0) || (t_bot && wr == 1);
    const int col_wr = t_top ? 1 : (t_bot ? 0 : (f_left ? 0 : 1));
    const bool my_col = (f_left || f_right) && wr == col_wr;
    const int fold_kh = t_top ? 0 : 2, fold_kw = f_left ? 0 : 2;    // the taps of this wave's term
    unsigned fold_taps = 0, cnr_taps = 0;       // bit t = 3 kh + kw: this wave multiplies a fold term / the corner term at tap t
    if (FOLD) {
        for (int tt = 0; tt < 9; ++tt) {
            const int kh_ = tt / 3, kw_ = tt - 3 * kh_;
            const bool row = my_row && kh_ == fold_kh;
            if (row || (my_col && kw_ == fold_kw)) fold_taps |= 1u << tt;
            if (row && ((kw_ == 0 && f_left) || (kw_ == 2 && f_right))) cnr_taps |= 1u << tt;
        }
    }
    f32x4 FA[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) FA[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // column term: lane <-> tile row l15: halo row (l15 + dh) * 20 + hx, hx = 1 (left) or 16 (right): both have key hx & 6 == 0
    const unsigned baseT = lds0 + (unsigned)l15 * (HP2 * ROWB) + (unsigned)(f_left ? 1 : 16) * ROWB + ((unsigned)g4 << 4);
    // row term: the tap's halo row 1 (top) / 16 (bottom) at this lane's column: a_cur without its dh rows and wave rows
    const unsigned f_rsel = (unsigned)(((t_top ? 1 : 16) - wr * 8) * (HP2 * ROWB));

    bf16x8 af[8], b0[4], b1[4];
    issue_halo(0);
    issue_w(0, 0);
    issue_w(0, 1);
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();
    int kc = 0, t = 0, kh = 0, kw = 0;                  // (chunk, tap = 3 kh + kw) of k-step ks
    unsigned a_cur = a_half0(SIGN > 0 ? 0 : 2, SIGN > 0 ? 0 : 2, 0);
    {
#pragma unroll
        for (int j = 0; j < 4; ++j) b0[j] = lds_frag(aB[0] + j * (16 * ROWB));
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = lds_frag(a_cur + i * (HP2 * ROWB));
    }
    for (int ks = 0; ks < nk; ++ks) {
        const unsigned sb = (unsigned)(ks & 1) * HSTAGE_B;
        const unsigned a1 = a_cur ^ 64u;                // second half of this tap
        const unsigned bb1 = aB[1] + sb;
        int kc2 = kc, t2 = t + 1, kh2 = kh, kw2 = kw + 1;
        if (kw2 == 3) { kw2 = 0; ++kh2; }
        if (t2 == 9) { t2 = 0; kh2 = 0; ++kc2; }
        const unsigned a0n = a_half0(SIGN > 0 ? kh2 : 2 - kh2, SIGN > 0 ? kw2 : 2 - kw2, kc2 & 1);
        const unsigned bb0n = aB[0] + (HSTAGE_B - sb);
        // fold role of this k-step (wave-uniform): this wave's row term (its kh) or column term (its kw); corner = a row
        // k-step of a left / right tile with kw = 0 / 2.  The fragment is fetched BEFORE the MFMA block of each half so
        // that the LDS latency hides behind it: the tap's halo row 1 / 16 at this lane's column, or the halo column 1 / 16
        // at this lane's row; lane 1 / 14 of the dw = 0 / 2 variant for the corner.
        // (one bit test per k-step outside the fold taps: fold_taps / cnr_taps are this wave's nine-bit tap masks of the tile)
        const bool do_fold = FOLD && ((fold_taps >> t) & 1u);
        const bool do_cnr = FOLD && ((cnr_taps >> t) & 1u);
        const unsigned f_st = (unsigned)((kc & 1) * HSTAGE_A2);
        unsigned f_addr = 0;
        bf16x8 axf;
        if (FOLD && do_fold) {
            const unsigned f_dhb = (unsigned)((2 - kh) * (HP2 * ROWB));
            f_addr = my_row ? a_cur - f_dhb + f_rsel : baseT + f_st + f_dhb;
            axf = lds_frag(f_addr);
        }
        // ---- half 0: multiply (ks, 0) while the fragments of (ks, 1) stream in
#pragma unroll
        for (int j = 0; j < 4; ++j) b1[j] = lds_frag(bb1 + j * (16 * ROWB));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16s<H16>(b0[j], af[i], acc[i][j]);
            af[i] = lds_frag(a1 + i * (HP2 * ROWB));
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if (FOLD && do_fold) {    // the fold term of this half (b0 = this tap's weights); fragment fetched above
#pragma unroll
            for (int j = 0; j < 4; ++j) FA[j] = mfma16s<H16>(b0[j], axf, FA[j]);
            if (do_cnr) {         // once per chunk in the four corner tiles: not worth registers for a prefetch
                axf = lds_frag((kw == 0 ? aA00 : aA20) + f_st + f_rsel);
                if (l15 != (kw == 0 ? 1 : 14) || (p.dbg & 2048)) axf = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int j = 0; j < 4; ++j) FA[j] = mfma16s<H16>(b0[j], axf, FA[j]);
            }
        }
        // ---- middle of the k-step: the weight stage ks is read; stage ks+1 (issued one k-step ago) must have
        // landed.  The halo of the next chunk, issued right after the weights at t == 0, may stay in flight across
        // the barrier of t == 1 (it is needed nine k-steps after its issue): the wait then leaves the newest
        // HROUNDS2 - 1 loads outstanding (every wave issues HROUNDS2 - 1 or HROUNDS2 of them)
        if (t == 1 && kc + 1 < KC) __builtin_amdgcn_s_waitcnt(0x0070 | (HROUNDS2 - 1));
        else __builtin_amdgcn_s_waitcnt(0x0070);
        __syncthreads();
        // The DMA of the next stages: a DMA instruction costs its wave 60-180 cycles of issue time, so the two waves of a SIMD
        // (wr = 0 / 1) issue theirs at different points of half 1 - one wave's issue runs under the other's multiplies
        // (wino_wgrad_dma.hip: 1045 -> 906 us from the same change; mmh_set_option("lp16_dbg") bit 32 = everybody here).
        auto issue_next = [&]() {
            if (ks + 2 < nk && !(p.dbg & 1)) {      // dbg: timing-only ablations (mmh_set_option "lp16_dbg"; results wrong)
                int kc3 = kc, t3 = t + 2;
                if (t3 >= 9) { t3 -= 9; ++kc3; }
                issue_w(kc3, t3);
            }
            if (t == 0 && kc + 1 < KC && !(p.dbg & 2)) issue_halo(kc + 1);
        };
        const bool early = wr == 0 || (p.dbg & 32);
        if (early) issue_next();
        if (FOLD && do_fold) axf = lds_frag(f_addr ^ 64u);      // the fold fragment of half 1
        // ---- half 1: multiply (ks, 1) while the fragments of (ks+1, 0) stream in
        // (after the last k-step these reads fetch fragments nobody uses, from addresses inside the stages: cheaper
        // than a branch around each of them)
#pragma unroll
        for (int j = 0; j < 4; ++j) b0[j] = lds_frag(bb0n + j * (16 * ROWB));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16s<H16>(b1[j], af[i], acc[i][j]);
            af[i] = lds_frag(a0n + i * (HP2 * ROWB));
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if (!early) issue_next();
#pragma unroll
        for (int i = 4; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16s<H16>(b1[j], af[i], acc[i][j]);
            af[i] = lds_frag(a0n + i * (HP2 * ROWB));
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if (FOLD && do_fold) {    // the fold term of this half (b1 = this tap's weights); fragment fetched above
#pragma unroll
            for (int j = 0; j < 4; ++j) FA[j] = mfma16s<H16>(b1[j], axf, FA[j]);
            if (do_cnr) {
                axf = lds_frag(((kw == 0 ? aA00 : aA20) + f_st + f_rsel) ^ 64u);
                if (l15 != (kw == 0 ? 1 : 14) || (p.dbg & 2048)) axf = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int j = 0; j < 4; ++j) FA[j] = mfma16s<H16>(b1[j], axf, FA[j]);
            }
        }
        kc = kc2; t = t2; kh = kh2; kw = kw2; a_cur = a0n;
    }

    if (FOLD && my_row) {               // the row term: same layout as the accumulators of tile row 1 / 14
        if (t_top) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[1][j] += FA[j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[6][j] += FA[j];
        }
    }
    if (FOLD && (f_left || f_right)) {
        // the column term: FA [channel 4 g4 + r of block j][tile row l15] -> LDS X[256 channels][16 rows] -> the lanes
        // that hold pixel column 1 (left) / 14 (right) of each tile row
        float* X = reinterpret_cast<float*>(smem);
        __syncthreads();                    // every wave is done reading the last stages
        if (my_col) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) X[(wc * 64 + j * 16 + 4 * g4 + r) * 16 + l15] = FA[j][r];
        }
        __syncthreads();
        if (l15 == (f_left ? 1 : 14)) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] += X[(wc * 64 + j * 16 + 4 * g4 + r) * 16 + wr * 8 + i];
        }
    }

    float bv[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[j][r] = p.bias ? p.bias[n0 + wc * 64 + j * 16 + 4 * g4 + r] : 0.f;
    const int ow = ow0 + l15;
    // The activation as a compile-time constant per branch (a run-time test per element put the tanh expansion behind every
    // one of the 128 values of a lane: 21 000 instructions of epilogue); the dgrad variants have none.
    auto with_act = [&](auto&& body) {
        if (SIGN > 0 && p.act == MMH_ACT_RELU) body(std::integral_constant<int, MMH_ACT_RELU>{});
        else if (SIGN > 0 && p.act == MMH_ACT_TANH) body(std::integral_constant<int, MMH_ACT_TANH>{});
        else body(std::integral_constant<int, MMH_ACT_NONE>{});
    };
    if (p.y16 && !(p.dbg & 128)) {
        // 16-bit output: 16-byte stores after the lane-pair trade (common.h: pair_swap8) - 16 store instructions per tile
        // instead of 32; 256 -> 256 fprop 150 -> 134 us, the 16-bit step 102.5 -> 100.8 ms (tools/ab_lp16_stores.py;
        // mmh_set_option("lp16_dbg", 128) = 8-byte stores)
        const bool odd = (g4 & 1) != 0;
        const int cb0 = (odd ? 16 : 0) + 4 * (g4 & 2);      // this lane's 8 channels within a 32-channel tile pair
        float bo[2][8];
#pragma unroll
        for (int jp = 0; jp < 2; ++jp)
#pragma unroll
            for (int e = 0; e < 8; ++e) bo[jp][e] = p.bias ? p.bias[n0 + wc * 64 + jp * 32 + cb0 + e] : 0.f;
        with_act([&](auto A) {
            constexpr int ACT = decltype(A)::value;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int oh = oh0 + wr * 8 + i;
                const size_t m = ((size_t)b * p.H + oh) * p.W + ow;
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    float v[8];
                    mmh::pair_swap8(acc[i][2 * jp], acc[i][2 * jp + 1], v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float t = v[e] + bo[jp][e];
                        v[e] = ACT == MMH_ACT_RELU ? (t > 0.f ? t : 0.f) : (ACT == MMH_ACT_TANH ? tanhf(t) : t);
                    }
                    if (oh < p.H && ow < p.W) {
                        mmh::store8_lp16<H16>(p.y16 + (m * p.y_cs + (n0 + wc * 64 + jp * 32 + cb0)) * 2, v);
                    }
                }
            }
        });
    } else if (SIGN < 0 && p.addend) {
        // dx = dgrad + addend, fp32 (mmh_conv3x3_lp16_dgrad_add: no bias, no activation).  The addend of tile row i + 1 is
        // requested while row i is added and stored: written as "load, add, store" per accumulator the compiler put an
        // s_waitcnt vmcnt(0) behind every load - 32 full memory round trips per tile, each also waiting for the store in
        // front of it (256 -> 256: 182 us against 142 for the same dgrad without addend).
        const size_t o00 = (((size_t)b * p.H + (oh0 + wr * 8)) * p.W + ow) * p.y_cs + (size_t)(n0 + wc * 64 + 4 * g4);
        const size_t rstep = (size_t)p.W * p.y_cs;             // one tile row down
        if (FOLD || (oh0 + HT <= p.H && ow0 + HT <= p.W)) {     // a full tile (always, with the fold): three rows in flight
            constexpr int NB = 3;
            f32x4 ad[NB][4];
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) ad[i][j] = *reinterpret_cast<const f32x4*>(p.addend + o00 + i * rstep + j * 16);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] += ad[i % NB][j];
                    *reinterpret_cast<f32x4*>(p.y + o00 + i * rstep + j * 16) = acc[i][j];
                }
                if (i + NB < 8) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        ad[i % NB][j] = *reinterpret_cast<const f32x4*>(p.addend + o00 + (i + NB) * rstep + j * 16);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (oh0 + wr * 8 + i >= p.H || ow >= p.W) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] += *reinterpret_cast<const f32x4*>(p.addend + o00 + i * rstep + j * 16);
                    *reinterpret_cast<f32x4*>(p.y + o00 + i * rstep + j * 16) = acc[i][j];
                }
            }
        }
    } else {
        with_act([&](auto A) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int oh = oh0 + wr * 8 + i;
                if (oh >= p.H || ow >= p.W) continue;
                const size_t m = ((size_t)b * p.H + oh) * p.W + ow;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const size_t elem = m * p.y_cs + (n0 + wc * 64 + j * 16 + 4 * g4);
                    store4<H16>(p.y, p.y16, elem, acc[i][j], bv[j], decltype(A)::value);
                }
            }
        });
    }
    if (SIGN > 0 && !FOLD && p.stats) {
        // The InstanceNorm behind this conv (models/Generator.py:66-77) wants mean and M2 per (image, channel): each wave
        // owns 8 rows x 16 pixels of 64 channels - count / mean / M2 of the values AS STORED (rounded to 16 bits) per
        // wave and channel, merged later (Chan) by mmh_norm_stats_merge[_finalize]: y is not read again for statistics.
        // Lane: 8 values per channel (two passes in registers), then four equal-count Chan merges across the 16 pixel
        // lanes.  Host side guarantees H, W multiples of 16 (no ragged tiles), no activation.
        const int chunks = TX * TY * 2;
        float* sp = p.stats + ((size_t)(b * chunks + (ty * TX + tx) * 2 + wr) * 3) * p.N + n0 + wc * 64 + 4 * g4;
        float mu[16], m2[16];               // channel c = 4 j + r of this lane's 16: mean and M2 of its 8 rows
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v[8], mean = 0.f, q = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float t = acc[i][j][r] + bv[j][r];
                    v[i] = H16 ? (float)(_Float16)t : (float)(__bf16)t;
                    mean += v[i];
                }
                mean *= 0.125f;
#pragma unroll
                for (int i = 0; i < 8; ++i) q = fmaf(v[i] - mean, v[i] - mean, q);
                mu[4 * j + r] = mean; m2[4 * j + r] = q;
            }
        // Four equal-count Chan merges across the 16 pixel lanes as a reduce-scatter: at the step with lane distance s the
        // lane keeps the half of its channels whose index bit matches its own lane bit and hands the other half to its
        // partner (ds_swizzle bit mode: and 0x1f, or 0, xor s) - 8 + 4 + 2 + 1 merges instead of 4 x 16, and lane l15 ends
        // up with channel l15 of the 16.
#define MMH_SWZ(val, s) __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, val), 0x1f | ((s) << 10)))
#define MMH_RS_STEP(NOUT, S, BIT, W, MI, QI, MO, QO)                                            \
    _Pragma("unroll") for (int c = 0; c < NOUT; ++c) {                                         \
        const float km = BIT ? MI[NOUT + c] : MI[c], sm = BIT ? MI[c] : MI[NOUT + c];           \
        const float kq = BIT ? QI[NOUT + c] : QI[c], sq = BIT ? QI[c] : QI[NOUT + c];           \
        const float om = MMH_SWZ(sm, S), oq = MMH_SWZ(sq, S), dl = om - km;                     \
        QO[c] = kq + oq + dl * dl * W; MO[c] = 0.5f * (km + om);                                \
    }
        const bool b3 = (l15 & 8) != 0, b2 = (l15 & 4) != 0, b1 = (l15 & 2) != 0, b0 = (l15 & 1) != 0;
        float ma[8], qa[8], mb[4], qb[4], mc[2], qc[2], md[1], qd[1];
        MMH_RS_STEP(8, 8, b3, 4.f, mu, m2, ma, qa)
        MMH_RS_STEP(4, 4, b2, 8.f, ma, qa, mb, qb)
        MMH_RS_STEP(2, 2, b1, 16.f, mb, qb, mc, qc)
        MMH_RS_STEP(1, 1, b0, 32.f, mc, qc, md, qd)
#undef MMH_RS_STEP
#undef MMH_SWZ
        const int co = (l15 >> 2) * 16 + (l15 & 3);     // channel 4 j + r = l15 of the lane's 16 -> j * 16 + r of the 64
        sp[co] = 128.f;
        sp[p.N + co] = md[0];
        sp[2 * p.N + co] = qd[0];
    }
    __syncthreads();        // the next tile's prologue refills the stages: every wave must be done with this tile's
    }   // tiles of this workgroup
}

// ---------------------------------------------------------------------------------------------
// The same machine for the other 3x3 convolutions of the step: stride 2 (the down-sampling convs),
// their dgrad and ConvTranspose2d (per output-parity class), and 64 / 128 output channels.
//   M-space  m -> (b, mh, mw), mh < MH, mw < MW
//   source   pixel (mh*ss + dh, mw*ss + dw) of the [B][SH][SW][cs] tensor, dh = (ah + sgn*kh) >> dsh for
//            tap kh = kh0 + tstep*th (th < nth); outside the image: zero page (or mirrored, ss == 1)
//   output   pixel (mh*os + oh0, mw*os + ow0) of [B][OH][OW][y_cs]
//   fprop stride s:         MH x MW = Ho x Wo, ss = s, dh = kh - pad, os = 1
//   dgrad stride 1:         dh = pad - kh
//   dgrad stride 2, class (ph, pw) = blockIdx.y: M-space = input pixels (2 mh + ph, 2 mw + pw), taps with
//            kh = (ph + pad) & 1 (mod 2), source (dy) pixel mh + (ph + pad - kh) / 2, os = 2, oh0 = ph
// Tile 256 x TBN x 64, TBN = 256 | 128 | 64 (template): 8 waves as 2 (M) x 4 (N), wave tile 128 x TBN/4.
struct LpGConvKP {
    const char* x;
    const char* w;          // [tap = kh*KW + kw][N][C] 16-bit
    const char* zeros;
    float* y;
    char* y16;
    const float* bias;
    int B, MH, MW;
    int SH, SW, C, cs, ss;
    int ah, aw, sgn, dsh;
    int kh0, kw0, tstep, nth, ntw, KW;
    int classes;            // 1, or 4: stride-2 dgrad, class = blockIdx.y overrides kh0/kw0/nth/ntw/ah/aw/oh0/ow0
    int pad, KH;
    int reflect;
    int OH, OW, os, oh0, ow0, y_cs, N;
    int act, MT, NT;
    int st16;               // 16-byte epilogue stores allowed (y_cs % 8 == 0: the address (pix*y_cs + n)*2 is 16-byte aligned)
};

template <bool H16, int BNT>
__device__ __forceinline__ void conv_lp16g_body(const LpGConvKP& p) {
    constexpr int NB = BNT / 64;               // B-operand DMA instructions per wave and k-step
    constexpr int NJ = BNT / 64;               // 16-column MFMA tiles per wave (wave tile 128 x TBN/4)
    constexpr int GSTAGE = (TBM + BNT) * ROWB;
    // Stages: these convs have short contractions (64 -> 128 stride 2: nine k-steps of 0.2 us of MFMA work each), so a
    // work-group lives on what it keeps in flight: with two stages ONE k-step's 48 KB travels while the other is
    // multiplied - 48 KB per L2 round trip, ~19 GB/s per CU, measured 207 us where HBM alone would need 80.  The
    // 128-column tile (48 KB stages) takes a third stage: two k-steps in flight, counted vmcnt.  (64 columns: two
    // work-groups per CU already; 256 columns: 64 KB stages, no room.)
    constexpr int NST = BNT == 128 ? 3 : 2;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wr = wave >> 2, wc = wave & 3;
    const int per_xcd = (p.MT * p.NT + 7) / 8;
    const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= p.MT * p.NT) return;
    const int mt = tile / p.NT, nt = tile - mt * p.NT;
    const int m0 = mt * TBM, n0 = nt * BNT;
    const int M = p.B * p.MH * p.MW;
    int kh0 = p.kh0, kw0 = p.kw0, nth = p.nth, ntw = p.ntw, ah = p.ah, aw = p.aw, oh0 = p.oh0, ow0 = p.ow0;
    if (p.classes == 4) {
        const int ph = blockIdx.y >> 1, pw = blockIdx.y & 1;
        kh0 = (ph + p.pad) & 1; kw0 = (pw + p.pad) & 1;
        nth = (p.KH - kh0 + 1) / 2; ntw = (p.KW - kw0 + 1) / 2;
        ah = ph + p.pad; aw = pw + p.pad;
        oh0 = ph; ow0 = pw;
    }
    int a_img[4], a_hw[4];          // image base b*SH*SW (or -1: row beyond M) and (mh*ss) << 16 | (mw*ss)
    unsigned b_off[NB];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (wave * 4 + j) * 8 + (lane >> 3);
        const int m = m0 + r;
        const int b = m / (p.MH * p.MW);
        const int rem = m - b * (p.MH * p.MW);
        const int mh = rem / p.MW, mw = rem - mh * p.MW;
        a_img[j] = m < M ? b * p.SH * p.SW : -1;
        a_hw[j] = ((mh * p.ss) << 16) | (mw * p.ss);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int r = (wave * NB + j) * 8 + (lane >> 3);
        const unsigned q = (unsigned)((lane & 7) ^ ((r >> 1) & 7));
        b_off[j] = (unsigned)(n0 + r) * (unsigned)p.C * 2u + q * 16u;
    }
    const int KC = p.C / TBK;
    const int nk = nth * ntw * KC;
    unsigned a_off[4];
    int wtap = 0;
    auto set_tap = [&](int t) {
        const int th = t / ntw, tw = t - th * ntw;
        const int kh = kh0 + p.tstep * th, kw = kw0 + p.tstep * tw;
        const int dh = (ah + p.sgn * kh) >> p.dsh, dw = (aw + p.sgn * kw) >> p.dsh;
        wtap = kh * p.KW + kw;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int ih = (a_hw[j] >> 16) + dh, iw = (a_hw[j] & 0xffff) + dw;
            bool ok = a_img[j] >= 0;
            if (p.reflect) {
                ih = ih < 0 ? -ih : ih;
                iw = iw < 0 ? -iw : iw;
                ih = ih >= p.SH ? 2 * (p.SH - 1) - ih : ih;
                iw = iw >= p.SW ? 2 * (p.SW - 1) - iw : iw;
            } else {
                ok = ok && ih >= 0 && ih < p.SH && iw >= 0 && iw < p.SW;
            }
            const int r = (wave * 4 + j) * 8 + (lane >> 3);
            const unsigned q = (unsigned)((lane & 7) ^ ((r >> 1) & 7));
            const int src = a_img[j] + ih * p.SW + iw;
            a_off[j] = ok ? (unsigned)src * (unsigned)p.cs * 2u + q * 16u : 0xffffffffu;
        }
    };
    auto issue = [&](int ks, int stage) {
        const int t = ks / KC, kc = ks - t * KC;
        if (kc == 0) set_tap(t);
        char* sA = smem + stage * GSTAGE;
        char* sB = sA + TBM * ROWB;
        const unsigned kb = (unsigned)kc * (TBK * 2);
        const char* wbase = p.w + (size_t)wtap * p.N * p.C * 2 + kb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const char* g = a_off[j] != 0xffffffffu ? p.x + a_off[j] + kb : p.zeros + (lane & 7) * 16;
            __builtin_amdgcn_global_load_lds(g, (lds_vp)(sA + (wave * 4 + j) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j)
            __builtin_amdgcn_global_load_lds(wbase + b_off[j], (lds_vp)(sB + (wave * NB + j) * 1024), 16, 0, 0);
    };

    f32x4 acc[8][NJ];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    const unsigned key = (unsigned)(l15 >> 1);
    const unsigned a_base = (unsigned)(wr * 128 + l15) * ROWB;
    const unsigned b_base = (unsigned)(TBM + wc * (BNT / 4) + l15) * ROWB;

    if (nk > 0) issue(0, 0);
    if (NST == 3 && nk > 1) issue(1, 1);
    int stg = 0;                                    // ks % NST
    for (int ks = 0; ks < nk; ++ks) {
        // k-step ks has landed; with three stages the 4 + NB loads of k-step ks + 1 (every wave issues exactly that
        // many) may stay in flight.  The barrier also frees the stage multiplied in the previous iteration.
        if (NST == 3 && ks + 1 < nk) __builtin_amdgcn_s_waitcnt(0x0070 | (4 + NB));
        else __builtin_amdgcn_s_waitcnt(0x0070);
        if (NST == 3) {
            // the bare barrier: __syncthreads() carries a fence in front of which hipcc drains every pending LDS-DMA
            // (s_waitcnt vmcnt(0)) - the second k-step in flight would never be.  The waits above order this wave's DMA
            // and LDS reads (lgkmcnt(0)) against the barrier themselves.
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else {
            __syncthreads();
        }
        if (ks + NST - 1 < nk) issue(ks + NST - 1, stg == 0 ? NST - 1 : stg - 1);
        const char* st = smem + stg * GSTAGE;
        stg = stg + 1 == NST ? 0 : stg + 1;
#pragma unroll
        for (int s32 = 0; s32 < 2; ++s32) {
            const unsigned sw = ((unsigned)(4 * s32 + g4) ^ key) << 4;
            bf16x8 af[8], bfr[NJ];
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const bf16x8*>(st + a_base + sw + i * (16 * ROWB));
#pragma unroll
            for (int j = 0; j < NJ; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(st + b_base + sw + j * (16 * ROWB));
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = mfma16s<H16>(bfr[j], af[i], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // epilogue: the weight fragment is the MFMA's first operand - lane (l15, g4) holds channels 4 g4 .. + 3 of pixel
    // m0 + wr*128 + i*16 + l15 (store4: a quarter of the store instructions; the bias is read once per lane)
    float bv[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[j][r] = p.bias ? p.bias[n0 + wc * (BNT / 4) + j * 16 + 4 * g4 + r] : 0.f;
    if (p.y16 && NJ % 2 == 0 && p.st16) {      // 16-byte stores after the lane-pair trade (common.h: pair_swap8)
        const int cb0 = ((g4 & 1) ? 16 : 0) + 4 * (g4 & 2);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + wr * 128 + i * 16 + l15;
            const int mc = m < M ? m : M - 1;
            const int b = mc / (p.MH * p.MW);
            const int rem = mc - b * (p.MH * p.MW);
            const int mh = rem / p.MW, mw = rem - mh * p.MW;
            const size_t opix = ((size_t)b * p.OH + (mh * p.os + oh0)) * p.OW + (mw * p.os + ow0);
#pragma unroll
            for (int jp = 0; jp < NJ / 2; ++jp) {
                float v[8];
                mmh::pair_swap8(acc[i][2 * jp], acc[i][2 * jp + (NJ > 1 ? 1 : 0)], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float t = v[e] + (p.bias ? p.bias[n0 + wc * (BNT / 4) + jp * 32 + cb0 + e] : 0.f);
                    v[e] = p.act == MMH_ACT_RELU ? (t > 0.f ? t : 0.f) : (p.act == MMH_ACT_TANH ? tanhf(t) : t);
                }
                if (m < M) mmh::store8_lp16<H16>(p.y16 + (opix * p.y_cs + (n0 + wc * (BNT / 4) + jp * 32 + cb0)) * 2, v);
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + wr * 128 + i * 16 + l15;
        if (m >= M) continue;
        const int b = m / (p.MH * p.MW);
        const int rem = m - b * (p.MH * p.MW);
        const int mh = rem / p.MW, mw = rem - mh * p.MW;
        const size_t opix = ((size_t)b * p.OH + (mh * p.os + oh0)) * p.OW + (mw * p.os + ow0);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            store4<H16>(p.y, p.y16, opix * p.y_cs + (n0 + wc * (BNT / 4) + j * 16 + 4 * g4), acc[i][j], bv[j], p.act);
    }
}


// ---------------------------------------------------------------------------------------------
// Flat-K variant for the 7x7 stems (Cin = 3 .. 42): the contraction index runs over (tap, channel)
// flattened, k = tap * C8 + c with the channels padded to C8 (a multiple of 8: one 16-byte chunk is
// 8 channels of one tap), so a 64-deep k-step spans 64 / C8 taps (C8 = 8: eight taps) and every lane
// of the A-operand DMA carries its own tap.  Tile 256 pixels x 64 channels x 64, 8 waves stacked along
// M (wave tile 32 x 64: 2 A and 4 B fragments per 8 MFMAs), two 40 KiB stages.
//   x16p  [B][H][W][C8] 16-bit (mmh_lp16_pad_cvt)      w  [N][Kpad] 16-bit, Kpad = roundup(KH*KW*C8, 64)
struct LpFlatKP {
    const char* x;
    const char* w;
    const char* zeros;
    float* y;
    char* y16;
    const float* bias;
    int B, H, W, C8, KH, KW, pad, reflect;
    int cpt, q8, r8;        // chunks per tap = C8 / 8; 8 / cpt and 8 % cpt (a lane's chunk index advances by 8 per k-step)
    int Kpad, nk;
    int N, y_cs, act, MT, NT;
};

template <bool H16>
__global__ void __launch_bounds__(512, 2) conv_lp16f_kernel(const LpFlatKP p) {
    constexpr int FSTAGE = (TBM + 64) * ROWB;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int per_xcd = (p.MT * p.NT + 7) / 8;
    const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= p.MT * p.NT) return;
    const int mt = tile / p.NT, nt = tile - mt * p.NT;
    const int m0 = mt * TBM, n0 = nt * 64;
    const int M = p.B * p.H * p.W;
    const int taps = p.KH * p.KW;
    // A DMA: 4 rows per lane; per row the lane's logical chunk q (of 8) and its running (tap, chunk-in-tap)
    int a_pix[4], a_hw[4], a_tap[4], a_cc[4], a_kh[4], a_kw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (wave * 4 + j) * 8 + (lane >> 3);
        const int q = (lane & 7) ^ ((r >> 1) & 7);
        const int m = m0 + r;
        const int b = m / (p.H * p.W);
        const int rem = m - b * (p.H * p.W);
        const int oh = rem / p.W, ow = rem - oh * p.W;
        a_pix[j] = m < M ? b * p.H * p.W : -1;
        a_hw[j] = (oh << 16) | ow;
        a_tap[j] = q / p.cpt;
        a_cc[j] = q - a_tap[j] * p.cpt;
        a_kh[j] = a_tap[j] / p.KW;
        a_kw[j] = a_tap[j] - a_kh[j] * p.KW;
    }
    const int br = wave * 8 + (lane >> 3);                              // B DMA: one row per lane
    const unsigned b_off = (unsigned)(n0 + br) * (unsigned)p.Kpad * 2u + (unsigned)((lane & 7) ^ ((br >> 1) & 7)) * 16u;

    auto issue = [&](int ks, int stage) {
        char* sA = smem + stage * FSTAGE;
        char* sB = sA + TBM * ROWB;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int ih = (a_hw[j] >> 16) + a_kh[j] - p.pad, iw = (a_hw[j] & 0xffff) + a_kw[j] - p.pad;
            bool ok = a_pix[j] >= 0 && a_tap[j] < taps;
            if (p.reflect) {
                ih = ih < 0 ? -ih : ih;
                iw = iw < 0 ? -iw : iw;
                ih = ih >= p.H ? 2 * (p.H - 1) - ih : ih;
                iw = iw >= p.W ? 2 * (p.W - 1) - iw : iw;
            } else {
                ok = ok && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            }
            const unsigned off = (unsigned)(a_pix[j] + ih * p.W + iw) * (unsigned)p.C8 * 2u + (unsigned)a_cc[j] * 16u;
            const char* g = ok ? p.x + off : p.zeros + (lane & 7) * 16;
            __builtin_amdgcn_global_load_lds(g, (lds_vp)(sA + (wave * 4 + j) * 1024), 16, 0, 0);
            // next k-step: this lane's flat chunk index advances by 8
            int dt = p.q8;
            a_cc[j] += p.r8;
            if (a_cc[j] >= p.cpt) { a_cc[j] -= p.cpt; ++dt; }
            a_tap[j] += dt;
            a_kw[j] += dt;
            while (a_kw[j] >= p.KW) { a_kw[j] -= p.KW; ++a_kh[j]; }
        }
        __builtin_amdgcn_global_load_lds(p.w + b_off + (unsigned)ks * (TBK * 2), (lds_vp)(sB + wave * 1024), 16, 0, 0);
    };

    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    const unsigned key = (unsigned)(l15 >> 1);
    const unsigned a_base = (unsigned)(wave * 32 + l15) * ROWB;
    const unsigned b_base = (unsigned)(TBM + l15) * ROWB;

    if (p.nk > 0) issue(0, 0);
    for (int ks = 0; ks < p.nk; ++ks) {
        __builtin_amdgcn_s_waitcnt(0x0070);     // vmcnt(0) lgkmcnt(0): the LDS-DMA of the previous iteration
        __syncthreads();
        if (ks + 1 < p.nk) issue(ks + 1, (ks + 1) & 1);
        const char* st = smem + (ks & 1) * FSTAGE;
#pragma unroll
        for (int s32 = 0; s32 < 2; ++s32) {
            const unsigned sw = ((unsigned)(4 * s32 + g4) ^ key) << 4;
            bf16x8 af[2], bfr[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8*>(st + a_base + sw + i * (16 * ROWB));
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(st + b_base + sw + j * (16 * ROWB));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16s<H16>(bfr[j], af[i], acc[i][j]);
        }
    }

    // epilogue: weight fragment first (see store4): lane (l15, g4) holds channels n0 + j*16 + 4 g4 .. + 3 of pixel
    // m0 + wave*32 + i*16 + l15
    float bv[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[j][r] = p.bias ? p.bias[n0 + j * 16 + 4 * g4 + r] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wave * 32 + i * 16 + l15;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            store4<H16>(p.y, p.y16, (size_t)m * p.y_cs + (n0 + j * 16 + 4 * g4), acc[i][j], bv[j], p.act);
    }
}

// x fp32 [rows][C] -> 16-bit [rows][C8], channels zero-padded (the stems' input for conv_lp16f_kernel)
__global__ void lp16_pad_cvt_kernel(const float* __restrict__ x, int64_t rows, int C, int C8, int h16,
                                    void* __restrict__ out) {
    const int64_t total = rows * (C8 / 8);
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int c8n = C8 / 8;
    for (; i < total; i += stride) {
        const int64_t row = i / c8n;
        const int c0 = (int)(i - row * c8n) * 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = c0 + e < C ? x[row * C + c0 + e] : 0.f;
        if (h16) {
            f16x8 r;
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = (_Float16)v[e];
            reinterpret_cast<f16x8*>(out)[i] = r;
        } else {
            bf16x8 r;
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = (__bf16)v[e];
            reinterpret_cast<bf16x8*>(out)[i] = r;
        }
    }
}

// w fp32 [taps][Cin][Cout] -> 16-bit [Cout][Kpad], k = tap * C8 + c (zero padded)
__global__ void prep_weights_flat8_kernel(const float* __restrict__ w, int taps, int Cin, int Cout, int C8, int Kpad,
                                          int h16, void* __restrict__ out) {
    const int64_t total = (int64_t)Cout * Kpad;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int k = (int)(i % Kpad), n = (int)(i / Kpad);
        const int t = k / C8, c = k - t * C8;
        const float v = (t < taps && c < Cin) ? w[((int64_t)t * Cin + c) * Cout + n] : 0.f;
        if (h16) reinterpret_cast<_Float16*>(out)[i] = (_Float16)v;
        else reinterpret_cast<__bf16*>(out)[i] = (__bf16)v;
    }
}

// one kernel per tile width over the common body (a __global__ template on the tile width made this
// clang drop the host stubs without a diagnostic)
#define MMH_LPG_KERNEL(TBNV)                                                                                    \
    template <bool H16>                                                                                         \
    __global__ void __launch_bounds__(512, 2) conv_lp16g##TBNV##_kernel(const LpGConvKP p) {                    \
        conv_lp16g_body<H16, TBNV>(p);                                                                          \
    }                                                                                                           \
    int launch_lp16g_##TBNV(const LpGConvKP& p, bool h16, dim3 grid, hipStream_t st) {                          \
        constexpr int lds = (TBNV == 128 ? 3 : 2) * (TBM + TBNV) * ROWB;                                        \
        static int ready = -1;                                                                                  \
        if (ready != 0) {                                                                                       \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_lp16g##TBNV##_kernel<false>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);               \
            if (e == hipSuccess)                                                                                \
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_lp16g##TBNV##_kernel<true>),        \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds);                      \
            ready = e == hipSuccess ? 0 : mmh::fail("conv_lp16g_kernel: %s", hipGetErrorString(e));            \
        }                                                                                                       \
        if (ready != 0) return ready;                                                                           \
        if (h16) hipLaunchKernelGGL(conv_lp16g##TBNV##_kernel<true>, grid, dim3(512), lds, st, p);             \
        else hipLaunchKernelGGL(conv_lp16g##TBNV##_kernel<false>, grid, dim3(512), lds, st, p);                \
        return 0;                                                                                               \
    }
MMH_LPG_KERNEL(256)
MMH_LPG_KERNEL(128)
MMH_LPG_KERNEL(64)
#undef MMH_LPG_KERNEL

template <bool H16>
__global__ void __launch_bounds__(512, 2) conv_lp16_kernel(const LpConvKP p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wr = wave >> 2, wc = wave & 3;        // 2 x 4 waves: rows wr*128.., cols wc*64..

    // XCD-aware tile order: the workgroups of one XCD (blockIdx % 8) walk consecutive tiles with
    // the column tile fastest, so both column tiles of an A row panel - and neighbouring row panels,
    // which share their halo rows - are served by one L2
    const int per_xcd = (p.MT * p.NT + 7) / 8;
    const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= p.MT * p.NT) return;
    const int mt = tile / p.NT, nt = tile - mt * p.NT;
    const int m0 = mt * TBM, n0 = nt * TBN;
    const int M = p.B * p.H * p.W;

    // ---- DMA roles: wave w issues, per operand and k-step, 4 instructions of 8 rows x 128 B:
    // rows (w*4 + j)*8 + lane/8, physical chunk lane%8 <- logical chunk (lane%8) ^ ((row>>1)&7)
    int a_pix[4], a_hw[4];          // (b*H + oh)*W + ow, and oh << 16 | ow; a_pix < 0: row beyond M
    unsigned b_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (wave * 4 + j) * 8 + (lane >> 3);
        const unsigned q = (unsigned)((lane & 7) ^ ((r >> 1) & 7));
        const int m = m0 + r;
        const int b = m / (p.H * p.W);
        const int rem = m - b * (p.H * p.W);
        const int oh = rem / p.W, ow = rem - oh * p.W;
        a_pix[j] = m < M ? m : -1;
        a_hw[j] = (oh << 16) | ow;
        b_off[j] = (unsigned)(n0 + r) * (unsigned)p.C * 2u + q * 16u;      // + (tap*N*C + kc*64)*2 per k-step
    }
    const int KC = p.C / TBK;                 // channel chunks per tap
    const int nk = 9 * KC;
    unsigned a_off[4];                        // byte offset of this lane's 16 B inside x, or ~0u (zero page)
    auto set_tap = [&](int t) {
        const int kh = t / 3, kw = t - 3 * kh;
        const int dh = p.tap_sign * (kh - 1), dw = p.tap_sign * (kw - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int oh = a_hw[j] >> 16, ow = a_hw[j] & 0xffff;
            int ih = oh + dh, iw = ow + dw;
            bool ok = a_pix[j] >= 0;
            if (p.reflect) {
                ih = ih < 0 ? -ih : ih;
                iw = iw < 0 ? -iw : iw;
                ih = ih >= p.H ? 2 * (p.H - 1) - ih : ih;
                iw = iw >= p.W ? 2 * (p.W - 1) - iw : iw;
            } else {
                ok = ok && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            }
            const int r = (wave * 4 + j) * 8 + (lane >> 3);
            const unsigned q = (unsigned)((lane & 7) ^ ((r >> 1) & 7));
            const int src = a_pix[j] + (ih - oh) * p.W + (iw - ow);
            a_off[j] = ok ? (unsigned)src * (unsigned)p.cs * 2u + q * 16u : 0xffffffffu;
        }
    };
    // k order (mmh_set_option "lp16_tap_inner"): tap outer / 64-channel chunk inner by default - a lane
    // recomputes its 4 source pixels once per tap.  Chunk-outer order keeps the nine taps of a chunk in
    // L2 (FETCH_SIZE is 9.2x the input with tap-outer order: each tap streams 8 MiB per XCD through a
    // 4 MiB L2 and is served by the Infinity Cache) but pays the source-pixel arithmetic every k-step:
    // measured 7 % SLOWER (A/B in one process, tools/ab_lp16_shape.py) - the kernel is not fetch-bound.
    auto issue = [&](int ks, int stage) {
        int kc, t;
        if (p.tap_inner) { kc = ks / 9; t = ks - kc * 9; set_tap(t); }
        else { t = ks / KC; kc = ks - t * KC; if (kc == 0) set_tap(t); }
        char* sA = smem + stage * STAGE;
        char* sB = sA + TBM * ROWB;
        const unsigned kb = (unsigned)kc * (TBK * 2);
        const char* wbase = p.w + (size_t)t * p.N * p.C * 2 + kb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const char* g = a_off[j] != 0xffffffffu ? p.x + a_off[j] + kb : p.zeros + (lane & 7) * 16;
            __builtin_amdgcn_global_load_lds(g, (lds_vp)(sA + (wave * 4 + j) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds(wbase + b_off[j], (lds_vp)(sB + (wave * 4 + j) * 1024), 16, 0, 0);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment reads: row R, logical chunk q = 2*s16 + h -> byte R*128 + ((q ^ ((R>>1)&7)) * 16).  Every
    // row this lane reads (A: wr*128 + i*32 + l31, B: wc*64 + j*32 + l31) has the same key (l31>>1)&7,
    // so the swizzled chunk offset is one register per s16 and the tile offsets are immediates
    const unsigned key = (unsigned)((l31 >> 1) & 7);
    const unsigned a_base = (unsigned)(wr * 128 + l31) * ROWB;
    const unsigned b_base = (unsigned)(TBM + wc * 64 + l31) * ROWB;

    issue(0, 0);
    for (int ks = 0; ks < nk; ++ks) {
        // this wave's DMAs of k-step ks have landed (vmcnt(0)), then everyone's (barrier); the barrier
        // also means every wave has finished reading the other stage, which k-step ks+1 overwrites
        __builtin_amdgcn_s_waitcnt(0x0070);     // vmcnt(0) lgkmcnt(0): the LDS-DMA of the previous iteration
        __syncthreads();
        if (ks + 1 < nk) issue(ks + 1, (ks + 1) & 1);
        const char* st = smem + (ks & 1) * STAGE;
        // fragments of k16-step s+1 are read while the MFMAs of step s run; the scheduling barriers keep
        // the compiler from hoisting all four steps' reads (96 VGPRs) above the first MFMA
        bf16x8 af[2][4], bfr[2][2];
        auto load_frags = [&](int s16, int buf) {
            const unsigned sw = ((unsigned)(2 * s16 + h) ^ key) << 4;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[buf][i] = *reinterpret_cast<const bf16x8*>(st + a_base + sw + i * (32 * ROWB));
#pragma unroll
            for (int j = 0; j < 2; ++j)
                bfr[buf][j] = *reinterpret_cast<const bf16x8*>(st + b_base + sw + j * (32 * ROWB));
        };
        load_frags(0, 0);
#pragma unroll
        for (int s16 = 0; s16 < 4; ++s16) {
            if (s16 < 3) load_frags(s16 + 1, (s16 + 1) & 1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = mfma16<H16>(af[s16 & 1][i], bfr[s16 & 1][j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wr * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m >= M) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wc * 64 + j * 32 + l31;
                float v = acc[i][j][r];
                if (p.bias) v += p.bias[n];
                v = act_apply(v, p.act);
                if (p.y16) {
                    if (H16) reinterpret_cast<_Float16*>(p.y16)[(size_t)m * p.y_cs + n] = (_Float16)v;
                    else reinterpret_cast<__bf16*>(p.y16)[(size_t)m * p.y_cs + n] = (__bf16)v;
                } else {
                    p.y[(size_t)m * p.y_cs + n] = v;
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------
// wgrad: dw[tap][ci][co] = sum over pixels p of x16[src(p, tap)][ci] * dy16[p][co].
// Per (tap, 256 ci, 256 co) output tile and pixel range (split-K) one workgroup; the contraction
// index (pixel) is the SLOW index of both NHWC operands, so the tiles are staged [64 pixels][256
// channels] (512-B rows, one LDS-DMA instruction = 2 rows) and both MFMA operands are read
// TRANSPOSED with ds_read_b64_tr_b16.  A transposed 4x16 block touches 4 consecutive rows x 64 B:
// with a 512-B pitch they would all sit on the same banks, so the 16-byte chunk index is XOR-ed
// with (row & 3) << 2 (again on the DMA's source address and on the read address).  fp32 partial
// slabs per split, summed in a fixed order by lp16_slab_reduce_kernel.
// ---------------------------------------------------------------------------------------------
constexpr int WROWB = 512;                      // bytes per LDS row: 256 channels
constexpr int WSTAGE = 2 * 64 * WROWB;          // x tile + dy tile, 64 pixels each: 64 KiB
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct LpWgradKP {
    const char* x;          // 16-bit activations [B][H][W][Cin], pixel stride x_cs
    const char* dy;         // 16-bit output gradients [B][H][W][Cout], pixel stride dy_cs
    const char* zeros;
    float* slab;            // [S][9][Cin][Cout]
    int B, H, W, Cin, Cout, x_cs, dy_cs;
    int reflect, h16;
    int S, ksteps_per_split;    // pixel range of split s: [s*ksteps*64, (s+1)*ksteps*64)
    int CT, NT;                 // ci / co tiles
    int items;                  // S * CT * NT * 9
    int dbg;                    // timing-only ablation bits (results wrong): 1 no DMA, 2 no fragment reads, 4 no MFMA
};

// Operand fragment for lane: 8 consecutive rows (pixels) row0 + 8h .. of column col0 + (lane & 31) from a
// [pixel][256 channels] image with the chunk swizzle above.  tr_off() is the lane's byte offset for
// row0 = 0 (computed once per column block); the k16-step adds the immediate row0 * 512.
__device__ __forceinline__ unsigned tr_off(int col0, int lane) {
    const int h = lane >> 5, G1 = (lane >> 4) & 1, q = (lane & 15) >> 2, p2 = lane & 3;
    const int row = 8 * h + q;                              // row & 3 == q
    const int col = col0 + 16 * G1 + 4 * p2;                // element index, 8-byte aligned
    const unsigned chunk = (unsigned)(col >> 3) ^ ((unsigned)q << 2);
    return (unsigned)row * WROWB + (chunk << 4) + (unsigned)(col & 4) * 2u;
}
__device__ __forceinline__ bf16x8 tr_frag_at(const char* a) {
    typedef s16x4 __attribute__((address_space(3))) * lds_p;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a + 4 * WROWB));
    struct { s16x4 a, b; } both = {lo, hi};
    return __builtin_bit_cast(bf16x8, both);
}

template <bool H16>
__global__ void __launch_bounds__(512, 2) wgrad_lp16_kernel(const LpWgradKP p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wr = wave >> 2, wc = wave & 3;
    // XCD-contiguous work list, tap fastest: the nine taps of one (split, tile) read the same dy
    // tile and neighbouring x pixels, back to back on one L2
    const int per_xcd = (p.items + 7) / 8;
    int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (item >= p.items) return;
    const int tap = item % 9; item /= 9;
    const int nt = item % p.NT; item /= p.NT;
    const int ct = item % p.CT;
    const int split = item / p.CT;
    const int P = p.B * p.H * p.W;
    const int kh = tap / 3, kw = tap - 3 * kh;
    const int k0 = split * p.ksteps_per_split;
    const int k1 = min((P + 63) / 64, k0 + p.ksteps_per_split);

    // DMA roles: per k-step and operand 64 rows x 512 B = 32 KiB = 32 instructions of 2 rows; wave w
    // issues 4 per operand: rows (w*4 + j)*2 + lane/32, physical chunk lane%32 <- logical ^ ((row&3)<<2)
    unsigned x_coff[4], d_coff[4];
    int pix[4], poh[4], pow_[4];        // this lane's 4 pixel rows of the current k-step: linear index, (oh, ow)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int prow = (wave * 4 + j) * 2 + (lane >> 5);
        const unsigned c = (unsigned)(lane & 31) ^ ((unsigned)(prow & 3) << 2);
        x_coff[j] = (unsigned)(ct * 256) * 2u + c * 16u;
        d_coff[j] = (unsigned)(nt * 256) * 2u + c * 16u;
        pix[j] = k0 * 64 + prow;
        const int rem = pix[j] % (p.H * p.W);
        poh[j] = rem / p.W;
        pow_[j] = rem - poh[j] * p.W;
    }
    // k-steps are issued in order, each exactly once: the pixel coordinates advance by 64 per step
    // (no division in the loop)
    auto issue = [&](int stage) {
        char* sX = smem + stage * WSTAGE;
        char* sD = sX + 64 * WROWB;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = pix[j] < P;
            int ih = poh[j] + kh - 1, iw = pow_[j] + kw - 1;
            bool okx = ok;
            if (p.reflect) {
                ih = ih < 0 ? -ih : ih;
                iw = iw < 0 ? -iw : iw;
                ih = ih >= p.H ? 2 * (p.H - 1) - ih : ih;
                iw = iw >= p.W ? 2 * (p.W - 1) - iw : iw;
            } else {
                okx = okx && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            }
            const int src = pix[j] + (ih - poh[j]) * p.W + (iw - pow_[j]);
            const char* gx = okx ? p.x + (size_t)src * p.x_cs * 2 + x_coff[j] : p.zeros + (lane & 31) * 16;
            const char* gd = ok ? p.dy + (size_t)pix[j] * p.dy_cs * 2 + d_coff[j] : p.zeros + (lane & 31) * 16;
            __builtin_amdgcn_global_load_lds(gx, (lds_vp)(sX + (wave * 4 + j) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(gd, (lds_vp)(sD + (wave * 4 + j) * 1024), 16, 0, 0);
            pix[j] += 64;
            pow_[j] += 64;
            while (pow_[j] >= p.W) {
                pow_[j] -= p.W;
                if (++poh[j] == p.H) poh[j] = 0;
            }
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    unsigned a_tr[4], b_tr[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) a_tr[i] = tr_off(wr * 128 + i * 32, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j) b_tr[j] = tr_off(wc * 64 + j * 32, lane);

    // One barrier per k-step, in its MIDDLE (as conv_lp16p_kernel): the fragments of the first k16-step of
    // k-step ks+1 are read while the last k16-step of k-step ks multiplies, so no wave starts a k-step
    // waiting on LDS.  At the barrier stage ks is fully read (the DMA of ks+2 may overwrite it) and stage
    // ks+1, issued one k-step earlier, has landed.
    bf16x8 af[2][4], bfr[2][2];
    auto load_frags = [&](const char* sX, int s16, int buf) {
        if (p.dbg & 2) return;
        const char* sD = sX + 64 * WROWB;
#pragma unroll
        for (int i = 0; i < 4; ++i) af[buf][i] = tr_frag_at(sX + a_tr[i] + s16 * (16 * WROWB));
#pragma unroll
        for (int j = 0; j < 2; ++j) bfr[buf][j] = tr_frag_at(sD + b_tr[j] + s16 * (16 * WROWB));
    };
    auto mult = [&](int buf) {
        if (p.dbg & 4) return;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mfma16<H16>(af[buf][i], bfr[buf][j], acc[i][j]);
    };
    if (k0 < k1) {
        issue(0);
        if (k0 + 1 < k1) issue(1);
        __builtin_amdgcn_s_waitcnt(0x0070);     // vmcnt(0) lgkmcnt(0)
        __syncthreads();
        load_frags(smem, 0, 0);
    }
    for (int ks = k0; ks < k1; ++ks) {
        const char* sX = smem + ((ks - k0) & 1) * WSTAGE;
        const char* sN = smem + ((ks + 1 - k0) & 1) * WSTAGE;
        load_frags(sX, 1, 1);
        mult(0);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(sX, 2, 0);
        mult(1);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(sX, 3, 1);
        mult(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0070);     // stage ks is read (lgkmcnt), stage ks+1 has landed (vmcnt)
        __syncthreads();
        if (ks + 2 < k1 && !(p.dbg & 1)) issue((ks - k0) & 1);
        if (ks + 1 < k1) load_frags(sN, 0, 0);
        mult(1);
        __builtin_amdgcn_sched_barrier(0);
    }

    float* slab = p.slab + ((size_t)split * 9 + tap) * p.Cin * p.Cout;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ct * 256 + wr * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int co = nt * 256 + wc * 64 + j * 32 + l31;
                slab[(size_t)ci * p.Cout + co] = acc[i][j][r];
            }
        }
}

template <bool H16>
__global__ void __launch_bounds__(512, 2) wgrad_lp16r_kernel(const LpWgradKP p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wr = wave >> 2, wc = wave & 3;
    // XCD-contiguous work list, tap fastest: the nine taps of one (split, tile) read the same dy
    // tile and neighbouring x pixels, back to back on one L2
    const int per_xcd = (p.items + 7) / 8;
    int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (item >= p.items) return;
    const int tap = item % 9; item /= 9;
    const int nt = item % p.NT; item /= p.NT;
    const int ct = item % p.CT;
    const int split = item / p.CT;
    const int P = p.B * p.H * p.W;
    const int kh = tap / 3, kw = tap - 3 * kh;
    const int k0 = split * p.ksteps_per_split;
    const int k1 = min((P + 63) / 64, k0 + p.ksteps_per_split);

    // DMA in HALF k-steps of 32 pixels (16 KiB per operand) into a ring of five 32-KiB slots: the k-step that
    // is being multiplied holds two, three are in flight - 96 KiB instead of the 64 KiB of a two-stage
    // pipeline.  Why: with the reads and the MFMAs ablated the DMA stream alone takes 493 us of the
    // kernel's 769 (512->512; tools/ablate_lp16.py): 64 KiB in flight per CU over a ~1.5 us L2 / Infinity
    // Cache round trip is all the bandwidth a CU gets.  Per half and operand 32 rows x 512 B = 16
    // instructions of 2 rows; wave w issues 2 per operand: rows (w*2 + j)*2 + lane/32, physical chunk
    // lane%32 <- logical ^ ((row&3)<<2).  Halves are issued in order, each exactly once, also beyond the
    // split's range (zero page; never multiplied), so every wave has the same number of loads in flight
    // and the waits can name it: vmcnt(4) = "all but the newest half".  (32-pixel steps with four halves in
    // flight: 20 % slower - twice the barriers, no fragment prefetch across them.)
    constexpr int RH = 5;                           // ring slots
    constexpr int HBYTES = 64 * WROWB;              // 32 rows x 512 B x 2 operands
    const int pend = min(P, k1 * 64);
    unsigned x_coff[2], d_coff[2];
    int pix[2], poh[2], pow_[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int prow = (wave * 2 + j) * 2 + (lane >> 5);
        const unsigned c = (unsigned)(lane & 31) ^ ((unsigned)(prow & 3) << 2);
        x_coff[j] = (unsigned)(ct * 256) * 2u + c * 16u;
        d_coff[j] = (unsigned)(nt * 256) * 2u + c * 16u;
        pix[j] = k0 * 64 + prow;
        const int rem = pix[j] % (p.H * p.W);
        poh[j] = rem / p.W;
        pow_[j] = rem - poh[j] * p.W;
    }
    int slot_next = 0;
    auto issue_half = [&]() {
        char* sX = smem + slot_next * HBYTES;
        char* sD = sX + 32 * WROWB;
        slot_next = slot_next + 1 == RH ? 0 : slot_next + 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = pix[j] < pend;
            int ih = poh[j] + kh - 1, iw = pow_[j] + kw - 1;
            bool okx = ok;
            if (p.reflect) {
                ih = ih < 0 ? -ih : ih;
                iw = iw < 0 ? -iw : iw;
                ih = ih >= p.H ? 2 * (p.H - 1) - ih : ih;
                iw = iw >= p.W ? 2 * (p.W - 1) - iw : iw;
            } else {
                okx = okx && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            }
            const int src = pix[j] + (ih - poh[j]) * p.W + (iw - pow_[j]);
            const char* gx = okx ? p.x + (size_t)src * p.x_cs * 2 + x_coff[j] : p.zeros + (lane & 31) * 16;
            const char* gd = ok ? p.dy + (size_t)pix[j] * p.dy_cs * 2 + d_coff[j] : p.zeros + (lane & 31) * 16;
            mmh::lds_dma16(gx, __builtin_amdgcn_readfirstlane(mmh::lds_addr_of(sX + (wave * 2 + j) * 1024)));
            mmh::lds_dma16(gd, __builtin_amdgcn_readfirstlane(mmh::lds_addr_of(sD + (wave * 2 + j) * 1024)));
            pix[j] += 32;
            pow_[j] += 32;
            while (pow_[j] >= p.W) {
                pow_[j] -= p.W;
                if (++poh[j] == p.H) poh[j] = 0;
            }
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    unsigned a_tr[4], b_tr[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) a_tr[i] = tr_off(wr * 128 + i * 32, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j) b_tr[j] = tr_off(wc * 64 + j * 32, lane);

    bf16x8 af[2][4], bfr[2][2];
    auto load_frags = [&](const char* sH, int s1, int buf) {       // s1: k16-step inside the half (0 | 1)
        const char* sD = sH + 32 * WROWB;
#pragma unroll
        for (int i = 0; i < 4; ++i) af[buf][i] = tr_frag_at(sH + a_tr[i] + s1 * (16 * WROWB));
#pragma unroll
        for (int j = 0; j < 2; ++j) bfr[buf][j] = tr_frag_at(sD + b_tr[j] + s1 * (16 * WROWB));
    };
    auto mult = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mfma16<H16>(af[buf][i], bfr[buf][j], acc[i][j]);
    };
    if (k0 < k1) {
        issue_half(); issue_half(); issue_half();           // halves 0, 1, 2
    }
    int slot = 0;                                           // ring slot of the first half of k-step ks
    for (int ks = k0; ks < k1; ++ks) {
        // halves 2t and 2t+1 of this k-step have landed when at most the newest half (2t+2: 4 loads per
        // wave) is still in flight; the barrier also says every wave is done with k-step ks-1, whose
        // two slots the halves 2t+3 and 2t+4 now take
        __builtin_amdgcn_s_waitcnt(0x0070 | 4);             // vmcnt(4) lgkmcnt(0)
        __syncthreads();
        issue_half(); issue_half();
        const char* hA = smem + slot * HBYTES;
        const int slotB = slot + 1 == RH ? 0 : slot + 1;
        const char* hB = smem + slotB * HBYTES;
        slot = slotB + 1 == RH ? 0 : slotB + 1;
        load_frags(hA, 0, 0);
        load_frags(hA, 1, 1);
        mult(0);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(hB, 0, 0);
        mult(1);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(hB, 1, 1);
        mult(0);
        __builtin_amdgcn_sched_barrier(0);
        mult(1);
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_waitcnt(0x0070);     // drain the halves issued past the range before the LDS is released

    float* slab = p.slab + ((size_t)split * 9 + tap) * p.Cin * p.Cout;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ct * 256 + wr * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int co = nt * 256 + wc * 64 + j * 32 + l31;
                slab[(size_t)ci * p.Cout + co] = acc[i][j][r];
            }
        }
}



// ---------------------------------------------------------------------------------------------
// wgrad with flat (tap, channel) rows: the weight gradient of the stems (7x7, Cin 3..42), of the
// stride-2 3x3 convs and of ConvTranspose2d.  dw[(tap, ci)][co] = sum over output pixels of
// x[pixel*stride + tap - pad][ci] * dy[pixel][co]: GEMM rows are the flat index f = tap * C8 + ci (C8 =
// channels per tap, a multiple of 8 = one 16-byte chunk), tiled by 256; columns are co, tiled by
// NTW = 64 | 128 | 256; the contraction runs over output pixels in the ring of half k-steps of
// wgrad_lp16r_kernel.  Every DMA lane of the x tile carries its own tap (its 16-byte chunk is 8 channels
// of one tap), as in conv_lp16f_kernel.  Waves: WM x WN = 8x1 (NTW 64), 4x2 (128), 2x4 (256), wave tile
// (256/WM) x (NTW/WN) in 32x32x16 MFMAs from transposed reads.
struct LpWgradFKP {
    const char* x;          // 16-bit gathered tensor [B][H][W][x_cs], C8 channels used per tap
    const char* dy;         // 16-bit per-pixel tensor [B][Ho][Wo][dy_cs]
    const char* zeros;
    float* slab;            // [S][MT*256][Cout]
    int B, H, W, C8, x_cs;
    int Ho, Wo, Cout, dy_cs;
    int KH, KW, stride, pad, reflect;
    int Mflat;              // KH*KW*C8
    int S, ksteps_per_split, MT, NT, items;
};

template <bool H16, int WM, int WN, int NTW>
__device__ __forceinline__ void wgrad_lp16f_body(const LpWgradFKP& p) {
    constexpr int TM = 256 / WM / 32, TN = NTW / WN / 32;
    constexpr int RH = 5;
    constexpr int HBYTES = 64 * WROWB;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int per_xcd = (p.items + 7) / 8;
    int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (item >= p.items) return;
    const int mt = item % p.MT; item /= p.MT;       // row tiles fastest: they share the dy tile and the x pixels
    const int nt = item % p.NT;
    const int split = item / p.NT;
    const int P = p.B * p.Ho * p.Wo;
    const int k0 = split * p.ksteps_per_split;
    const int k1 = min((P + 63) / 64, k0 + p.ksteps_per_split);
    const int pend = min(P, k1 * 64);

    // per DMA instruction j of a half (rows (wave*2 + j)*2 + lane/32): this lane's logical chunk, its tap
    unsigned x_coff[2], d_coff[2];
    int a_kh[2], a_kw[2];
    bool a_ok[2], d_ok[2];
    int pix[2], pimg[2], poh[2], pow_[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int prow = (wave * 2 + j) * 2 + (lane >> 5);
        const unsigned c = (unsigned)(lane & 31) ^ ((unsigned)(prow & 3) << 2);
        const int f = mt * 256 + (int)c * 8;
        const int tap = f / p.C8;
        a_kh[j] = tap / p.KW;
        a_kw[j] = tap - a_kh[j] * p.KW;
        a_ok[j] = f < p.Mflat;
        x_coff[j] = (unsigned)(f - tap * p.C8) * 2u;
        d_ok[j] = (int)c * 8 < NTW;
        d_coff[j] = (unsigned)(nt * NTW + (int)c * 8) * 2u;
        pix[j] = k0 * 64 + prow;
        pimg[j] = pix[j] / (p.Ho * p.Wo);
        const int rem = pix[j] - pimg[j] * (p.Ho * p.Wo);
        poh[j] = rem / p.Wo;
        pow_[j] = rem - poh[j] * p.Wo;
    }
    int slot_next = 0;
    auto issue_half = [&]() {
        char* sX = smem + slot_next * HBYTES;
        char* sD = sX + 32 * WROWB;
        slot_next = slot_next + 1 == RH ? 0 : slot_next + 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = pix[j] < pend;
            int ih = poh[j] * p.stride + a_kh[j] - p.pad, iw = pow_[j] * p.stride + a_kw[j] - p.pad;
            bool okx = ok && a_ok[j];
            if (p.reflect) {
                ih = ih < 0 ? -ih : ih;
                iw = iw < 0 ? -iw : iw;
                ih = ih >= p.H ? 2 * (p.H - 1) - ih : ih;
                iw = iw >= p.W ? 2 * (p.W - 1) - iw : iw;
            } else {
                okx = okx && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            }
            const size_t src = ((size_t)pimg[j] * p.H + ih) * p.W + iw;
            const char* gx = okx ? p.x + src * p.x_cs * 2 + x_coff[j] : p.zeros + (lane & 31) * 16;
            const char* gd = (ok && d_ok[j]) ? p.dy + (size_t)pix[j] * p.dy_cs * 2 + d_coff[j] : p.zeros + (lane & 31) * 16;
            mmh::lds_dma16(gx, __builtin_amdgcn_readfirstlane(mmh::lds_addr_of(sX + (wave * 2 + j) * 1024)));
            mmh::lds_dma16(gd, __builtin_amdgcn_readfirstlane(mmh::lds_addr_of(sD + (wave * 2 + j) * 1024)));
            pix[j] += 32;
            pow_[j] += 32;
            while (pow_[j] >= p.Wo) {
                pow_[j] -= p.Wo;
                if (++poh[j] == p.Ho) { poh[j] = 0; ++pimg[j]; }
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    unsigned a_tr[TM], b_tr[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_tr[i] = tr_off(wm * (256 / WM) + i * 32, lane);
#pragma unroll
    for (int j = 0; j < TN; ++j) b_tr[j] = tr_off(wn * (NTW / WN) + j * 32, lane);

    bf16x8 af[2][TM], bfr[2][TN];
    auto load_frags = [&](const char* sH, int s1, int buf) {
        const char* sD = sH + 32 * WROWB;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[buf][i] = tr_frag_at(sH + a_tr[i] + s1 * (16 * WROWB));
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[buf][j] = tr_frag_at(sD + b_tr[j] + s1 * (16 * WROWB));
    };
    auto mult = [&](int buf) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = mfma16<H16>(af[buf][i], bfr[buf][j], acc[i][j]);
    };
    if (k0 < k1) { issue_half(); issue_half(); issue_half(); }
    int slot = 0;
    for (int ks = k0; ks < k1; ++ks) {
        __builtin_amdgcn_s_waitcnt(0x0070 | 4);             // vmcnt(4): all but the newest half; lgkmcnt(0)
        __syncthreads();
        issue_half(); issue_half();
        const char* hA = smem + slot * HBYTES;
        const int slotB = slot + 1 == RH ? 0 : slot + 1;
        const char* hB = smem + slotB * HBYTES;
        slot = slotB + 1 == RH ? 0 : slotB + 1;
        load_frags(hA, 0, 0);
        load_frags(hA, 1, 1);
        mult(0);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(hB, 0, 0);
        mult(1);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(hB, 1, 1);
        mult(0);
        __builtin_amdgcn_sched_barrier(0);
        mult(1);
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_waitcnt(0x0070);

    float* slab = p.slab + (size_t)split * ((size_t)p.MT * 256) * p.Cout;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = mt * 256 + wm * (256 / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int co = nt * NTW + wn * (NTW / WN) + j * 32 + l31;
                slab[(size_t)f * p.Cout + co] = acc[i][j][r];
            }
        }
}

#define MMH_WGF_KERNEL(NAME, WM, WN, NTW)                                                          \
    template <bool H16>                                                                            \
    __global__ void __launch_bounds__(512, 2) NAME(const LpWgradFKP p) {                           \
        wgrad_lp16f_body<H16, WM, WN, NTW>(p);                                                     \
    }
MMH_WGF_KERNEL(wgrad_lp16f64_kernel, 8, 1, 64)
MMH_WGF_KERNEL(wgrad_lp16f128_kernel, 4, 2, 128)
MMH_WGF_KERNEL(wgrad_lp16f256_kernel, 2, 4, 256)
#undef MMH_WGF_KERNEL

// dw[tap][ci][co] (+)= sum over splits of slab[s][tap*C8 + ci][co], ci < Cin; fixed order
__global__ void lp16f_slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int taps, int Cin,
                                         int C8, int Cout, int S, size_t slab_stride, int accumulate) {
    const int64_t n4 = (int64_t)taps * Cin * Cout / 4;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int co4 = Cout / 4;
    for (; i < n4; i += stride) {
        const int64_t row = i / co4;
        const int c4 = (int)(i - row * co4);
        const int tap = (int)(row / Cin), ci = (int)(row - (int64_t)tap * Cin);
        const size_t src = ((size_t)tap * C8 + ci) * Cout + (size_t)c4 * 4;
        float4 a = *reinterpret_cast<const float4*>(slab + src);
        for (int s = 1; s < S; ++s) {
            const float4 b = *reinterpret_cast<const float4*>(slab + (size_t)s * slab_stride + src);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        if (accumulate) {
            const float4 b = reinterpret_cast<const float4*>(dw)[i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        reinterpret_cast<float4*>(dw)[i] = a;
    }
}

// dw[i] (+)= sum over splits of slab[s][i], fixed order
__global__ void lp16_slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int64_t n4, int S,
                                        int accumulate) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        float4 a = reinterpret_cast<const float4*>(slab)[i];
        for (int s = 1; s < S; ++s) {
            const float4 b = reinterpret_cast<const float4*>(slab)[(int64_t)s * n4 + i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        if (accumulate) {
            const float4 b = reinterpret_cast<const float4*>(dw)[i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        reinterpret_cast<float4*>(dw)[i] = a;
    }
}

// fp32 -> 16-bit copy (the activation twin), 8 elements per lane
__global__ void cvt_lp16_kernel(const float* __restrict__ x, void* __restrict__ out, int64_t n8, int h16) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n8; i += stride) {
        const float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
        if (h16) {
            f16x8 r;
            r[0] = (_Float16)a.x; r[1] = (_Float16)a.y; r[2] = (_Float16)a.z; r[3] = (_Float16)a.w;
            r[4] = (_Float16)b.x; r[5] = (_Float16)b.y; r[6] = (_Float16)b.z; r[7] = (_Float16)b.w;
            reinterpret_cast<f16x8*>(out)[i] = r;
        } else {
            bf16x8 r;
            r[0] = (__bf16)a.x; r[1] = (__bf16)a.y; r[2] = (__bf16)a.z; r[3] = (__bf16)a.w;
            r[4] = (__bf16)b.x; r[5] = (__bf16)b.y; r[6] = (__bf16)b.z; r[7] = (__bf16)b.w;
            reinterpret_cast<bf16x8*>(out)[i] = r;
        }
    }
}

}  // namespace

// fprop / dgrad kernel (mmh_set_option "lp16_shape"): 19 = conv_lp16h2_kernel (default: conv_lp16h_kernel with the
// fragment address arithmetic out of the k-loop), 18 = conv_lp16h_kernel (MFMA 16x16x32,
// fragment reads pipelined into the MFMA stream, activation halo of a 16x16 pixel tile resident in LDS for
// all nine taps; 1040-1170 TFLOP/s on the PATBlock shapes), 17 = conv_lp16p_kernel (the same pipelining
// on 256-pixel row tiles, the activation tile re-fetched per tap; also what images smaller than 16x16
// take), 16 = without the pipelining (6-13 % slower), 32 = MFMA 32x32x16 (a further 6-9 % slower)
// wgrad kernel (mmh_set_option "lp16_wgrad_ring"): 2 = wgrad_lp16t_kernel (default: nine taps of a 64 x 128 tile resident,
// the input halo of a 4 x 16 pixel block staged once: wgrad_lp16t.hip), 1 = wgrad_lp16r_kernel (one tap of a 256 x 256
// tile per workgroup, ring of five LDS slots), 0 = wgrad_lp16_kernel (the same with two stages)
namespace mmh { int g_lp16_shape = 19; int g_lp16_tap_inner = 0; int g_lp16_dbg = 0; int g_lp16_wgrad_ring = 2; int g_lp16_persist = 1; }
using mmh::g_lp16_shape;

extern "C" {

int mmh_cvt_lp16(const void* x, int64_t n, int dtype, void* out, mmh_stream_t s) {
    MMH_REQUIRE(x && out && n > 0 && n % 8 == 0 && (dtype == MMH_BF16 || dtype == MMH_FP16),
                "mmh_cvt_lp16: bad arguments (n %% 8 == 0, dtype MMH_BF16 | MMH_FP16)");
    const int64_t n8 = n / 8;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(mmh::cdiv(n8, 256), 8192));
    hipLaunchKernelGGL(cvt_lp16_kernel, dim3(grid), dim3(256), 0, mmh::as_stream(s), static_cast<const float*>(x), out,
                       n8, dtype == MMH_FP16 ? 1 : 0);
    return mmh::check_launch("cvt_lp16_kernel");
}

int mmh_conv3x3_lp16_supported(const mmh_conv_desc* d) {
    return d && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->Cin % 64 == 0 && d->Cout % 64 == 0 &&
           d->Ho == d->H && d->Wo == d->W && (d->dtype == MMH_BF16 || d->dtype == MMH_FP16);
}

int mmh_conv3x3_lp16_fold_supported(const mmh_conv_desc* d) {
    return mmh_conv3x3_lp16_supported(d) && d->pad_mode == MMH_PAD_REFLECT && d->H % HT == 0 && d->W % HT == 0 &&
           d->H >= 2 * HT && d->W >= 2 * HT && d->Cin % TBN == 0 && g_lp16_shape == 19;
}

// mode 0: fprop  y[B,H,W,Cout] = conv(x16 [B,H,W,Cin], w16 = w_t [tap][Cout][Cin]) (+bias, act)
// mode 1: dgrad  dx[B,H,W,Cin] = zero-padded correlation of dy16 [B,H,W,Cout] with the flipped filter,
//                w16 = w_plain [tap][Cin][Cout]; for MMH_PAD_REFLECT this is the main term only
//                (the caller adds the border terms: mmh_conv2d_dgrad_border)
// mode 2: dgrad of a reflect-padded conv COMPLETE: mode 1 plus the pad ring's gradient folded onto rows 1 / H-2 and
//                columns 1 / W-2 inside the kernel (mmh_conv3x3_lp16_fold_supported; no border call follows)
static int conv3x3_lp16_impl(const mmh_conv_desc* d, int mode, const void* x16, const void* w16, const void* bias,
                             void* y, int y_is16, int act, const void* zeros, void* stats, mmh_stream_t s,
                             const void* addend = nullptr) {
    MMH_REQUIRE(mmh_conv3x3_lp16_supported(d) && x16 && w16 && y && zeros && (mode == 0 || mode == 1 || mode == 2),
                "mmh_conv3x3_lp16: 3x3 / stride 1 / pad 1, Cin, Cout %% 64 == 0, 16-bit dtype");
    MMH_REQUIRE(mode != 2 || mmh_conv3x3_lp16_fold_supported(d),
                "mmh_conv3x3_lp16: mode 2 (dgrad with the reflect fold) needs MMH_PAD_REFLECT, H and W multiples of 16 and >= 32, "
                "the halo kernel (lp16_shape 19)");
    LpConvKP p{};
    const int K = mode == 0 ? d->Cin : d->Cout, N = mode == 0 ? d->Cout : d->Cin;
    MMH_REQUIRE(N % TBN == 0, "mmh_conv3x3_lp16: output channels must be a multiple of 256 (got %d)", N);
    MMH_REQUIRE((mode == 0 ? d->y_cs : d->x_cs) % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0,
                "mmh_conv3x3_lp16: the output's pixel stride must be a multiple of 4 channels and y 16-byte aligned");
    p.x = static_cast<const char*>(x16);
    p.w = static_cast<const char*>(w16);
    p.zeros = static_cast<const char*>(zeros);
    if (y_is16) p.y16 = static_cast<char*>(y); else p.y = static_cast<float*>(y);
    p.bias = static_cast<const float*>(bias);
    p.B = d->B; p.H = d->H; p.W = d->W; p.C = K; p.cs = mode == 0 ? d->x_cs : d->y_cs;
    p.N = N; p.y_cs = mode == 0 ? d->y_cs : d->x_cs;
    p.tap_sign = mode == 0 ? 1 : -1;        // dgrad (modes 1, 2): dx[i] = sum_t w[t] dy[i + 1 - t]
    p.reflect = (mode == 0 && d->pad_mode == MMH_PAD_REFLECT) ? 1 : 0;
    p.act = act;
    p.h16 = d->dtype == MMH_FP16;
    p.tap_inner = mmh::g_lp16_tap_inner;
    p.dbg = mmh::g_lp16_dbg;
    if (y_is16 && p.y_cs % 8 != 0) p.dbg |= 128;       // 8-byte stores: 16-byte ones need (pix*y_cs + n)*2 16-byte aligned
    p.stats = static_cast<float*>(stats);
    p.addend = static_cast<const float*>(addend);
    const long long M = (long long)d->B * d->H * d->W;
    MMH_REQUIRE(M * (long long)std::max(p.cs, p.y_cs) < (1ll << 31) && d->H < 32768 && d->W < 65536,
                "mmh_conv3x3_lp16: tensor too large");
    p.MT = (int)((M + TBM - 1) / TBM);
    p.NT = N / TBN;
    const int per_xcd = (p.MT * p.NT + 7) / 8;
    static int ready = -1;
    if (ready != 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_lp16_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_lp16_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
        ready = e == hipSuccess ? 0 : mmh::fail("conv_lp16_kernel: %s", hipGetErrorString(e));
    }
    if (ready != 0) return ready;
    if (g_lp16_shape == 19 && d->H >= HT && d->W >= HT) {       // halo kernel, fragment addresses precomputed (default)
        constexpr int lds2 = 2 * HSTAGE_A2 + 2 * HSTAGE_B + HROUNDS2 * 64 * 4;    // + the fold variant's row-offset table
        static int ready19 = -1;
        if (ready19 != 0) {
            hipError_t e = hipSuccess;
            const void* fs[6] = {reinterpret_cast<const void*>(conv_lp16h2_kernel<false, 1, false>),
                                 reinterpret_cast<const void*>(conv_lp16h2_kernel<false, -1, false>),
                                 reinterpret_cast<const void*>(conv_lp16h2_kernel<true, 1, false>),
                                 reinterpret_cast<const void*>(conv_lp16h2_kernel<true, -1, false>),
                                 reinterpret_cast<const void*>(conv_lp16h2_kernel<false, -1, true>),
                                 reinterpret_cast<const void*>(conv_lp16h2_kernel<true, -1, true>)};
            for (const void* f : fs)
                if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
            ready19 = e == hipSuccess ? 0 : mmh::fail("conv_lp16h2_kernel: %s", hipGetErrorString(e));
        }
        if (ready19 != 0) return ready19;
        LpConvKP ph = p;
        ph.MT = d->B * ((d->H + HT - 1) / HT) * ((d->W + HT - 1) / HT);
        // one workgroup per tile, or - with more tiles than CUs - one PERSISTENT workgroup per CU
        // that walks its XCD's tiles (mmh_set_option("lp16_persist", 0): off)
        int wpx = (ph.MT * ph.NT + 7) / 8;
        if (mmh::g_lp16_persist && (mode != 2 || mmh::g_lp16_persist == 1)) {     // option value 2: not the reflect-fold variant
            static int cus = 0;
            if (!cus) {
                int dev = 0, n = 0;
                if (hipGetDevice(&dev) != hipSuccess ||
                    hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8)
                    n = 256;
                cus = n;
            }
            wpx = std::min(wpx, cus / 8);
        }
        const dim3 grid(8 * wpx);
        hipStream_t st = mmh::as_stream(s);
        if (mode == 2) {
            if (p.h16) hipLaunchKernelGGL((conv_lp16h2_kernel<true, -1, true>), grid, dim3(512), lds2, st, ph);
            else hipLaunchKernelGGL((conv_lp16h2_kernel<false, -1, true>), grid, dim3(512), lds2, st, ph);
        } else if (p.h16 && mode == 0) hipLaunchKernelGGL((conv_lp16h2_kernel<true, 1, false>), grid, dim3(512), lds2, st, ph);
        else if (p.h16) hipLaunchKernelGGL((conv_lp16h2_kernel<true, -1, false>), grid, dim3(512), lds2, st, ph);
        else if (mode == 0) hipLaunchKernelGGL((conv_lp16h2_kernel<false, 1, false>), grid, dim3(512), lds2, st, ph);
        else hipLaunchKernelGGL((conv_lp16h2_kernel<false, -1, false>), grid, dim3(512), lds2, st, ph);
        return mmh::check_launch("conv_lp16h2_kernel");
    }
    if (g_lp16_shape == 18 && d->H >= HT && d->W >= HT) {       // activation tile (halo) resident in LDS for all nine taps
        constexpr int lds = 2 * HSTAGE_A + 2 * HSTAGE_B;
        static int ready18 = -1;
        if (ready18 != 0) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_lp16h_kernel<false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_lp16h_kernel<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            ready18 = e == hipSuccess ? 0 : mmh::fail("conv_lp16h_kernel: %s", hipGetErrorString(e));
        }
        if (ready18 != 0) return ready18;
        LpConvKP ph = p;
        ph.MT = d->B * ((d->H + HT - 1) / HT) * ((d->W + HT - 1) / HT);
        const int pxh = (ph.MT * ph.NT + 7) / 8;
        if (p.h16) hipLaunchKernelGGL(conv_lp16h_kernel<true>, dim3(8 * pxh), dim3(512), lds, mmh::as_stream(s), ph);
        else hipLaunchKernelGGL(conv_lp16h_kernel<false>, dim3(8 * pxh), dim3(512), lds, mmh::as_stream(s), ph);
        return mmh::check_launch("conv_lp16h_kernel");
    }
    if (g_lp16_shape == 17 || g_lp16_shape == 18 || g_lp16_shape == 19) {       // 16x16x32 with the fragment reads pipelined into the MFMA stream
        static int ready17 = -1;
        if (ready17 != 0) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_lp16p_kernel<false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_lp16p_kernel<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
            ready17 = e == hipSuccess ? 0 : mmh::fail("conv_lp16p_kernel: %s", hipGetErrorString(e));
        }
        if (ready17 != 0) return ready17;
        if (p.h16)
            hipLaunchKernelGGL(conv_lp16p_kernel<true>, dim3(8 * per_xcd), dim3(512), 2 * STAGE, mmh::as_stream(s), p);
        else
            hipLaunchKernelGGL(conv_lp16p_kernel<false>, dim3(8 * per_xcd), dim3(512), 2 * STAGE, mmh::as_stream(s), p);
        return mmh::check_launch("conv_lp16p_kernel");
    }
    if (g_lp16_shape == 16) {
        static int ready16 = -1;
        if (ready16 != 0) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_lp16s_kernel<false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_lp16s_kernel<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
            ready16 = e == hipSuccess ? 0 : mmh::fail("conv_lp16s_kernel: %s", hipGetErrorString(e));
        }
        if (ready16 != 0) return ready16;
        if (p.h16)
            hipLaunchKernelGGL(conv_lp16s_kernel<true>, dim3(8 * per_xcd), dim3(512), 2 * STAGE, mmh::as_stream(s), p);
        else
            hipLaunchKernelGGL(conv_lp16s_kernel<false>, dim3(8 * per_xcd), dim3(512), 2 * STAGE, mmh::as_stream(s), p);
        return mmh::check_launch("conv_lp16s_kernel");
    }
    if (p.h16)
        hipLaunchKernelGGL(conv_lp16_kernel<true>, dim3(8 * per_xcd), dim3(512), 2 * STAGE, mmh::as_stream(s), p);
    else
        hipLaunchKernelGGL(conv_lp16_kernel<false>, dim3(8 * per_xcd), dim3(512), 2 * STAGE, mmh::as_stream(s), p);
    return mmh::check_launch("conv_lp16_kernel");
}

int mmh_conv3x3_lp16(const mmh_conv_desc* d, int mode, const void* x16, const void* w16, const void* bias,
                     void* y, int y_is16, int act, const void* zeros, mmh_stream_t s) {
    return conv3x3_lp16_impl(d, mode, x16, w16, bias, y, y_is16, act, zeros, nullptr, s);
}

// dgrad (mode 1 | 2) with an fp32 dx that also receives `addend` (fp32, dx's layout): dx = dgrad(dy) + addend.  The
// input of such a conv has a second consumer (the residual stream: PATBlock out = x1 + ..., models/Generator.py:115-130;
// ResnetBlock out = x + conv_block(x), models/Discriminator.py:50) whose gradient autograd would add in a pass of its own.
int mmh_conv3x3_lp16_dgrad_add_supported(const mmh_conv_desc* d) {
    return mmh_conv3x3_lp16_supported(d) && d->H >= HT && d->W >= HT && d->Cin % TBN == 0 && g_lp16_shape == 19;
}

int mmh_conv3x3_lp16_dgrad_add(const mmh_conv_desc* d, int mode, const void* dy16, const void* w16, const void* addend,
                               void* dx, const void* zeros, mmh_stream_t s) {
    MMH_REQUIRE((mode == 1 || mode == 2) && addend && mmh_conv3x3_lp16_dgrad_add_supported(d),
                "mmh_conv3x3_lp16_dgrad_add: mode 1 | 2 on the halo kernel (H, W >= 16, Cin %% 256 == 0)");
    MMH_REQUIRE((reinterpret_cast<uintptr_t>(addend) & 15) == 0, "mmh_conv3x3_lp16_dgrad_add: addend must be 16-byte aligned");
    return conv3x3_lp16_impl(d, mode, dy16, w16, nullptr, dx, 0, MMH_ACT_NONE, zeros, nullptr, s, addend);
}

// fprop with a 16-bit output whose per-(image, half tile, channel) partial statistics come out of the epilogue:
// chunks per image = 2 * (H / 16) * (W / 16) (0: not available - ragged tiles or another kernel selected)
int mmh_conv3x3_lp16_stats_chunks(const mmh_conv_desc* d) {
    if (!mmh_conv3x3_lp16_supported(d) || d->H % HT || d->W % HT || d->Cout % TBN || g_lp16_shape != 19) return 0;
    return 2 * (d->H / HT) * (d->W / HT);
}

int mmh_conv3x3_lp16_fprop_stats(const mmh_conv_desc* d, const void* x16, const void* w16, const void* bias, void* y16,
                                 void* stats, const void* zeros, mmh_stream_t s) {
    MMH_REQUIRE(stats && mmh_conv3x3_lp16_stats_chunks(d) > 0,
                "mmh_conv3x3_lp16_fprop_stats: needs H, W multiples of 16, Cout %% 256 == 0 and the halo kernel");
    return conv3x3_lp16_impl(d, 0, x16, w16, bias, y16, 1, MMH_ACT_NONE, zeros, stats, s);
}


// General 16-bit 3x3 convolution on conv_lp16g_kernel: pad 1, stride 1 | 2, 64 | 128 | 256-wide column
// tiles.  mode 0: fprop y = conv(x16, w16 = w_t [tap][Cout][Cin]) (+bias, act);
// mode 1: dgrad dx = conv^T(dy16, w16 = w_plain [tap][Cin][Cout]) on the zero-padded problem (for
// MMH_PAD_REFLECT the caller adds the border terms); ConvTranspose2d(k3,s2,p1,op1) is mode 1 of the
// stride-2 conv it is the adjoint of.  y / dx: fp32 or 16-bit (y_is16).
int mmh_conv_lp16_supported(const mmh_conv_desc* d, int mode) {
    if (!d || d->kh != 3 || d->kw != 3 || d->pad != 1 || (d->stride != 1 && d->stride != 2)) return 0;
    if (d->dtype != MMH_BF16 && d->dtype != MMH_FP16) return 0;
    const int K = mode == 0 ? d->Cin : d->Cout, N = mode == 0 ? d->Cout : d->Cin;
    if (K % 64 || N % 64) return 0;
    if (d->stride == 2 && (d->H % 2 || d->W % 2 || d->Ho != d->H / 2 || d->Wo != d->W / 2)) return 0;
    if (d->stride == 2 && d->pad_mode == MMH_PAD_REFLECT) return 0;
    if (d->stride == 1 && (d->Ho != d->H || d->Wo != d->W)) return 0;
    return 1;
}

int mmh_conv_lp16(const mmh_conv_desc* d, int mode, const void* x16, const void* w16, const void* bias,
                  void* y, int y_is16, int act, const void* zeros, mmh_stream_t s) {
    MMH_REQUIRE((mode == 0 || mode == 1) && mmh_conv_lp16_supported(d, mode) && x16 && w16 && y && zeros,
                "mmh_conv_lp16: 3x3 / pad 1 / stride 1|2 (even H, W; zero padding for stride 2), channels %% 64 == 0, "
                "16-bit dtype");
    if (mmh::conv_s2f_ok(d, mode) && d->y_cs % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0)
        return mmh::launch_conv_s2f(d, x16, w16, bias, y, y_is16, act, zeros, mmh::as_stream(s));
    if (mmh::conv_s2d_ok(d, mode) && d->x_cs % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0)
        return mmh::launch_conv_s2d(d, x16, w16, bias, y, y_is16, act, zeros, mmh::as_stream(s));
    if (mmh::conv_s1f_ok(d, mode) && d->x_cs % 4 == 0 && d->y_cs % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0)
        return mmh::launch_conv_s2f(d, x16, w16, bias, y, y_is16, act, zeros, mmh::as_stream(s), nullptr, mode);
    LpGConvKP p{};
    const int K = mode == 0 ? d->Cin : d->Cout, N = mode == 0 ? d->Cout : d->Cin;
    p.x = static_cast<const char*>(x16);
    p.w = static_cast<const char*>(w16);
    p.zeros = static_cast<const char*>(zeros);
    if (y_is16) p.y16 = static_cast<char*>(y); else p.y = static_cast<float*>(y);
    p.bias = static_cast<const float*>(bias);
    p.B = d->B; p.C = K; p.N = N;
    p.KW = 3; p.KH = 3; p.pad = 1; p.tstep = 1; p.nth = 3; p.ntw = 3; p.classes = 1; p.os = 1;
    p.act = act;
    if (mode == 0) {            // fprop: M-space = output pixels, source = x
        p.MH = d->Ho; p.MW = d->Wo; p.SH = d->H; p.SW = d->W; p.cs = d->x_cs; p.ss = d->stride;
        p.ah = p.aw = -1; p.sgn = 1; p.dsh = 0;
        p.OH = d->Ho; p.OW = d->Wo; p.y_cs = d->y_cs;
        p.reflect = (d->pad_mode == MMH_PAD_REFLECT) ? 1 : 0;
    } else if (d->stride == 1) {
        p.MH = d->H; p.MW = d->W; p.SH = d->Ho; p.SW = d->Wo; p.cs = d->y_cs; p.ss = 1;
        p.ah = p.aw = 1; p.sgn = -1; p.dsh = 0;
        p.OH = d->H; p.OW = d->W; p.y_cs = d->x_cs;
    } else {                    // stride-2 dgrad: four output-parity classes in grid.y
        p.MH = d->H / 2; p.MW = d->W / 2; p.SH = d->Ho; p.SW = d->Wo; p.cs = d->y_cs; p.ss = 1;
        p.sgn = -1; p.dsh = 1; p.tstep = 2; p.classes = 4; p.os = 2;
        p.OH = d->H; p.OW = d->W; p.y_cs = d->x_cs;
    }
    const long long M = (long long)p.B * p.MH * p.MW;
    MMH_REQUIRE(p.y_cs % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0,
                "mmh_conv_lp16: the output's pixel stride must be a multiple of 4 channels and y 16-byte aligned");
    p.st16 = p.y_cs % 8 == 0;       // else the 8-byte store4 epilogue (a stride of 4 mod 8 misaligns dwordx4 stores)
    MMH_REQUIRE((long long)p.B * p.SH * p.SW * p.cs < (1ll << 31) && (long long)p.B * p.OH * p.OW < (1ll << 31) &&
                    p.SH < 16384 && p.SW < 32768,
                "mmh_conv_lp16: tensor too large");
    p.MT = (int)((M + TBM - 1) / TBM);
    const int tbn = N % 256 == 0 ? 256 : (N % 128 == 0 ? 128 : 64);
    p.NT = N / tbn;
    const bool h16 = d->dtype == MMH_FP16;
    hipStream_t st = mmh::as_stream(s);
    const int per_xcd = (p.MT * p.NT + 7) / 8;
    const dim3 grid(8 * per_xcd, p.classes);
    int rc;
    if (tbn == 256) rc = launch_lp16g_256(p, h16, grid, st);
    else if (tbn == 128) rc = launch_lp16g_128(p, h16, grid, st);
    else rc = launch_lp16g_64(p, h16, grid, st);
    if (rc) return rc;
    return mmh::check_launch("conv_lp16g_kernel");
}


int mmh_conv_lp16_stats_chunks(const mmh_conv_desc* d) { return d ? mmh::conv_s2f_stats_chunks(d) : 0; }

int mmh_conv_lp16_fprop_stats(const mmh_conv_desc* d, const void* x16, const void* w16, const void* bias, void* y16,
                              void* stats, const void* zeros, mmh_stream_t s) {
    MMH_REQUIRE(d && mmh::conv_s2f_stats_chunks(d) > 0 && x16 && w16 && y16 && stats && zeros && d->y_cs % 4 == 0 &&
                    (reinterpret_cast<uintptr_t>(y16) & 15) == 0,
                "mmh_conv_lp16_fprop_stats: shapes with mmh_conv_lp16_stats_chunks(d) > 0 only (the stride-2 kernel of "
                "conv_s2_lp16.hip)");
    return mmh::launch_conv_s2f(d, x16, w16, bias, y16, 1, MMH_ACT_NONE, zeros, mmh::as_stream(s), static_cast<float*>(stats));
}

// ---- flat-K 16-bit fprop for the 7x7 stems (models/Generator.py:158-164, Discriminator.py:79-84) ----
int mmh_lp16_pad_cvt(const void* x, int64_t rows, int C, int C8, int dtype, void* out, mmh_stream_t s) {
    MMH_REQUIRE(x && out && rows > 0 && C > 0 && C8 >= C && C8 % 8 == 0 && (dtype == MMH_BF16 || dtype == MMH_FP16),
                "mmh_lp16_pad_cvt: bad arguments (C8 %% 8 == 0, C8 >= C, 16-bit dtype)");
    const int64_t total = rows * (C8 / 8);
    hipLaunchKernelGGL(lp16_pad_cvt_kernel, dim3((unsigned)std::min<int64_t>(mmh::cdiv(total, 256), 8192)), dim3(256), 0,
                       mmh::as_stream(s), static_cast<const float*>(x), rows, C, C8, dtype == MMH_FP16 ? 1 : 0, out);
    return mmh::check_launch("lp16_pad_cvt_kernel");
}

int mmh_prep_weights_lp16_flat8(const void* w, int taps, int Cin, int Cout, int C8, int dtype, void* out,
                                mmh_stream_t s) {
    MMH_REQUIRE(w && out && taps > 0 && Cin > 0 && Cout > 0 && C8 >= Cin && C8 % 8 == 0 &&
                    (dtype == MMH_BF16 || dtype == MMH_FP16),
                "mmh_prep_weights_lp16_flat8: bad arguments");
    const int Kpad = (taps * C8 + 63) / 64 * 64;
    const int64_t total = (int64_t)Cout * Kpad;
    hipLaunchKernelGGL(prep_weights_flat8_kernel, dim3((unsigned)std::min<int64_t>(mmh::cdiv(total, 256), 4096)), dim3(256),
                       0, mmh::as_stream(s), static_cast<const float*>(w), taps, Cin, Cout, C8, Kpad,
                       dtype == MMH_FP16 ? 1 : 0, out);
    return mmh::check_launch("prep_weights_flat8_kernel");
}

int mmh_conv_lp16_flat_supported(const mmh_conv_desc* d, int C8) {
    return d && d->kh == d->kw && d->kh * d->kw * (C8 / 8) >= 8 && d->stride == 1 && d->pad == d->kh / 2 && d->kh % 2 == 1 &&
           d->Cout % 64 == 0 && C8 % 8 == 0 && C8 >= d->Cin && C8 <= 64 && d->Ho == d->H && d->Wo == d->W &&
           (d->dtype == MMH_BF16 || d->dtype == MMH_FP16);
}

// y[B,H,W,Cout] = conv(x16p [B,H,W,C8], w_flat [Cout][Kpad]) (+bias, act); stride 1, 'same' padding
int mmh_conv_lp16_flat(const mmh_conv_desc* d, const void* x16p, int C8, const void* w_flat, const void* bias,
                       void* y, int y_is16, int act, const void* zeros, mmh_stream_t s) {
    MMH_REQUIRE(mmh_conv_lp16_flat_supported(d, C8) && x16p && w_flat && y && zeros,
                "mmh_conv_lp16_flat: odd square kernel, stride 1, same padding, Cout %% 64 == 0, C8 %% 8 == 0 <= 64");
    LpFlatKP p{};
    p.x = static_cast<const char*>(x16p);
    p.w = static_cast<const char*>(w_flat);
    p.zeros = static_cast<const char*>(zeros);
    if (y_is16) p.y16 = static_cast<char*>(y); else p.y = static_cast<float*>(y);
    p.bias = static_cast<const float*>(bias);
    p.B = d->B; p.H = d->H; p.W = d->W; p.C8 = C8; p.KH = d->kh; p.KW = d->kw; p.pad = d->pad;
    p.reflect = d->pad_mode == MMH_PAD_REFLECT ? 1 : 0;
    p.cpt = C8 / 8; p.q8 = 8 / p.cpt; p.r8 = 8 % p.cpt;
    p.Kpad = (d->kh * d->kw * C8 + 63) / 64 * 64;
    p.nk = p.Kpad / 64;
    p.N = d->Cout; p.y_cs = d->y_cs; p.act = act;
    const long long M = (long long)d->B * d->H * d->W;
    MMH_REQUIRE(M * (long long)std::max(C8, p.y_cs) < (1ll << 31) && d->H < 32768 && d->W < 65536,
                "mmh_conv_lp16_flat: tensor too large");
    MMH_REQUIRE(p.y_cs % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0,
                "mmh_conv_lp16_flat: the output's pixel stride must be a multiple of 4 channels and y 16-byte aligned");
    p.MT = (int)((M + TBM - 1) / TBM);
    p.NT = d->Cout / 64;
    constexpr int lds = 2 * (TBM + 64) * ROWB;
    static int ready = -1;
    if (ready != 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_lp16f_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_lp16f_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        ready = e == hipSuccess ? 0 : mmh::fail("conv_lp16f_kernel: %s", hipGetErrorString(e));
    }
    if (ready != 0) return ready;
    const int per_xcd = (p.MT * p.NT + 7) / 8;
    if (d->dtype == MMH_FP16)
        hipLaunchKernelGGL(conv_lp16f_kernel<true>, dim3(8 * per_xcd), dim3(512), lds, mmh::as_stream(s), p);
    else
        hipLaunchKernelGGL(conv_lp16f_kernel<false>, dim3(8 * per_xcd), dim3(512), lds, mmh::as_stream(s), p);
    return mmh::check_launch("conv_lp16f_kernel");
}


// ---- flat (tap, channel)-row 16-bit wgrad: stems, stride-2 convs, ConvTranspose2d ----
static void lp16f_geometry(const mmh_conv_desc* d, int C8, int& ntw, int& MT, int& NT, int& S, int& ksteps) {
    ntw = d->Cout % 256 == 0 ? 256 : (d->Cout % 128 == 0 ? 128 : 64);
    MT = (d->kh * d->kw * C8 + 255) / 256;
    NT = d->Cout / ntw;
    const long long P = (long long)d->B * d->Ho * d->Wo;
    ksteps = (int)((P + 63) / 64);
    S = std::max(1, 256 / (MT * NT));
    S = (int)std::min<long long>(S, std::max<long long>(1, ksteps / 8));
}

int mmh_wgrad_lp16_flat_supported(const mmh_conv_desc* d, int C8) {
    return d && (d->dtype == MMH_BF16 || d->dtype == MMH_FP16) && C8 % 8 == 0 && C8 >= d->Cin && d->Cout % 64 == 0 &&
           d->Cout % 4 == 0 && d->Cin % 4 == 0 && d->kh == d->kw && (d->stride == 1 || d->stride == 2) &&
           (d->pad_mode != MMH_PAD_REFLECT || d->stride == 1);
}

size_t mmh_wgrad_lp16_flat_ws_bytes(const mmh_conv_desc* d, int C8) {
    if (!mmh_wgrad_lp16_flat_supported(d, C8)) return 0;
    int ntw, MT, NT, S, ksteps;
    lp16f_geometry(d, C8, ntw, MT, NT, S, ksteps);
    return (size_t)S * MT * 256 * d->Cout * sizeof(float);
}

// dw [kh][kw][Cin][Cout] (fp32) (+)= wgrad of conv d from the 16-bit x16 [B][H][W][x_cs] (C8 channels per
// tap are read: C8 = Cin, or the padded width of mmh_lp16_pad_cvt for the stems, then x_cs = C8) and dy16.
int mmh_wgrad_lp16_flat(const mmh_conv_desc* d, const void* x16, int C8, int x_cs, const void* dy16, void* dw,
                        void* ws, size_t ws_bytes, int accumulate, const void* zeros, mmh_stream_t s) {
    MMH_REQUIRE(mmh_wgrad_lp16_flat_supported(d, C8) && x16 && dy16 && dw && ws && zeros && x_cs >= C8 && x_cs % 8 == 0,
                "mmh_wgrad_lp16_flat: square kernel, stride 1|2, Cout %% 64 == 0, C8 %% 8 == 0 >= Cin, 16-bit dtype");
    MMH_REQUIRE(ws_bytes >= mmh_wgrad_lp16_flat_ws_bytes(d, C8), "mmh_wgrad_lp16_flat: workspace too small");
    LpWgradFKP p{};
    p.x = static_cast<const char*>(x16); p.dy = static_cast<const char*>(dy16);
    p.zeros = static_cast<const char*>(zeros);
    p.slab = static_cast<float*>(ws);
    p.B = d->B; p.H = d->H; p.W = d->W; p.C8 = C8; p.x_cs = x_cs;
    p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout; p.dy_cs = d->y_cs;
    p.KH = d->kh; p.KW = d->kw; p.stride = d->stride; p.pad = d->pad;
    p.reflect = d->pad_mode == MMH_PAD_REFLECT ? 1 : 0;
    p.Mflat = d->kh * d->kw * C8;
    int ntw, ksteps;
    lp16f_geometry(d, C8, ntw, p.MT, p.NT, p.S, ksteps);
    p.ksteps_per_split = (ksteps + p.S - 1) / p.S;
    p.S = (ksteps + p.ksteps_per_split - 1) / p.ksteps_per_split;
    p.items = p.S * p.MT * p.NT;
    MMH_REQUIRE((long long)d->B * d->H * d->W * x_cs < (1ll << 31) && (long long)d->B * d->Ho * d->Wo * d->y_cs < (1ll << 31),
                "mmh_wgrad_lp16_flat: tensor too large");
    hipStream_t st = mmh::as_stream(s);
    const bool h16 = d->dtype == MMH_FP16;
    constexpr int lds = 5 * 64 * WROWB;
    const int per_xcd = (p.items + 7) / 8;
#define MMH_WGF_LAUNCH(NAME)                                                                                      \
    do {                                                                                                          \
        static int ready = -1;                                                                                    \
        if (ready != 0) {                                                                                         \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(NAME<false>),                       \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);                  \
            if (e == hipSuccess)                                                                                  \
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(NAME<true>),                               \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds);                         \
            ready = e == hipSuccess ? 0 : mmh::fail(#NAME ": %s", hipGetErrorString(e));                          \
        }                                                                                                         \
        if (ready != 0) return ready;                                                                             \
        if (h16) hipLaunchKernelGGL(NAME<true>, dim3(8 * per_xcd), dim3(512), lds, st, p);                        \
        else hipLaunchKernelGGL(NAME<false>, dim3(8 * per_xcd), dim3(512), lds, st, p);                           \
    } while (0)
    if (ntw == 256) MMH_WGF_LAUNCH(wgrad_lp16f256_kernel);
    else if (ntw == 128) MMH_WGF_LAUNCH(wgrad_lp16f128_kernel);
    else MMH_WGF_LAUNCH(wgrad_lp16f64_kernel);
#undef MMH_WGF_LAUNCH
    if (int rc = mmh::check_launch("wgrad_lp16f_kernel")) return rc;
    const int64_t n4 = (int64_t)d->kh * d->kw * d->Cin * d->Cout / 4;
    hipLaunchKernelGGL(lp16f_slab_reduce_kernel, dim3((unsigned)std::min<int64_t>(mmh::cdiv(n4, 256), 4096)), dim3(256), 0,
                       st, p.slab, static_cast<float*>(dw), d->kh * d->kw, d->Cin, C8, d->Cout, p.S,
                       (size_t)p.MT * 256 * d->Cout, accumulate);
    return mmh::check_launch("lp16f_slab_reduce_kernel");
}

static int lp16_wgrad_splits(const mmh_conv_desc* d) {
    const int tiles = 9 * (d->Cin / 256) * (d->Cout / 256);
    const long long ksteps = ((long long)d->B * d->H * d->W + 63) / 64;
    int S = std::max(1, 256 / tiles);       // one workgroup per CU, never a mostly empty second round
    S = (int)std::min<long long>(S, std::max<long long>(1, ksteps / 8));
    return S;
}

size_t mmh_wgrad3x3_lp16_ws_bytes(const mmh_conv_desc* d) {
    if (!d || d->Cin % 256 || d->Cout % 256) return 0;
    int S = lp16_wgrad_splits(d);
    if (mmh::wgrad_lp16t_supported(d)) S = std::max(S, mmh::wgrad_lp16t_splits(d));    // whichever kernel is selected
    return (size_t)S * 9 * d->Cin * d->Cout * sizeof(float);
}

// dw [3][3][Cin][Cout] (fp32) (+)= wgrad of the 3x3 / stride 1 / pad 1 conv from 16-bit x and dy.
int mmh_wgrad3x3_lp16(const mmh_conv_desc* d, const void* x16, const void* dy16, void* dw, void* ws,
                      size_t ws_bytes, int accumulate, const void* zeros, mmh_stream_t s) {
    MMH_REQUIRE(mmh_conv3x3_lp16_supported(d) && x16 && dy16 && dw && ws && zeros && d->Cin % 256 == 0 &&
                    d->Cout % 256 == 0,
                "mmh_wgrad3x3_lp16: 3x3 / stride 1 / pad 1, Cin and Cout %% 256 == 0, 16-bit dtype");
    MMH_REQUIRE(ws_bytes >= mmh_wgrad3x3_lp16_ws_bytes(d), "mmh_wgrad3x3_lp16: workspace too small");
    LpWgradKP p{};
    p.x = static_cast<const char*>(x16); p.dy = static_cast<const char*>(dy16);
    p.zeros = static_cast<const char*>(zeros);
    p.slab = static_cast<float*>(ws);
    p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.x_cs = d->x_cs; p.dy_cs = d->y_cs;
    p.reflect = d->pad_mode == MMH_PAD_REFLECT ? 1 : 0;
    p.h16 = d->dtype == MMH_FP16;
    p.dbg = mmh::g_lp16_dbg;
    const long long P = (long long)d->B * d->H * d->W;
    MMH_REQUIRE(P * (long long)std::max(p.x_cs, p.dy_cs) < (1ll << 31), "mmh_wgrad3x3_lp16: tensor too large");
    const int ksteps = (int)((P + 63) / 64);
    p.S = lp16_wgrad_splits(d);
    p.ksteps_per_split = (ksteps + p.S - 1) / p.S;
    p.S = (ksteps + p.ksteps_per_split - 1) / p.ksteps_per_split;      // every split owns >= 1 k-step
    p.CT = d->Cin / 256; p.NT = d->Cout / 256;
    p.items = p.S * p.CT * p.NT * 9;
    hipStream_t st = mmh::as_stream(s);
    if (mmh::g_lp16_wgrad_ring >= 2 && mmh::wgrad_lp16t_supported(d)) {
        if (int rc = mmh::launch_wgrad_lp16t(d, x16, dy16, p.slab, zeros, st)) return rc;
        const int64_t n4t = (int64_t)9 * d->Cin * d->Cout / 4;
        hipLaunchKernelGGL(lp16_slab_reduce_kernel, dim3((unsigned)std::min<int64_t>(mmh::cdiv(n4t, 256), 4096)), dim3(256),
                           0, st, p.slab, static_cast<float*>(dw), n4t, mmh::wgrad_lp16t_splits(d), accumulate);
        return mmh::check_launch("lp16_slab_reduce_kernel");
    }
    static int ready = -1;
    if (ready != 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_lp16_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WSTAGE);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_lp16_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WSTAGE);
        ready = e == hipSuccess ? 0 : mmh::fail("wgrad_lp16_kernel: %s", hipGetErrorString(e));
    }
    if (ready != 0) return ready;
    const int per_xcd = (p.items + 7) / 8;
    static int ring_ok = -1;        // 160 KiB of LDS for one workgroup: the whole CU
    if (ring_ok < 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_lp16r_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 64 * WROWB);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_lp16r_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 64 * WROWB);
        ring_ok = e == hipSuccess ? 1 : 0;
        if (!ring_ok) (void)hipGetLastError();
    }
    if (ring_ok && mmh::g_lp16_wgrad_ring) {
        if (p.h16) hipLaunchKernelGGL(wgrad_lp16r_kernel<true>, dim3(8 * per_xcd), dim3(512), 5 * 64 * WROWB, st, p);
        else hipLaunchKernelGGL(wgrad_lp16r_kernel<false>, dim3(8 * per_xcd), dim3(512), 5 * 64 * WROWB, st, p);
    } else {
        if (p.h16) hipLaunchKernelGGL(wgrad_lp16_kernel<true>, dim3(8 * per_xcd), dim3(512), 2 * WSTAGE, st, p);
        else hipLaunchKernelGGL(wgrad_lp16_kernel<false>, dim3(8 * per_xcd), dim3(512), 2 * WSTAGE, st, p);
    }
    if (int rc = mmh::check_launch("wgrad_lp16_kernel")) return rc;
    const int64_t n4 = (int64_t)9 * d->Cin * d->Cout / 4;
    hipLaunchKernelGGL(lp16_slab_reduce_kernel, dim3((unsigned)std::min<int64_t>(mmh::cdiv(n4, 256), 4096)), dim3(256),
                       0, st, p.slab, static_cast<float*>(dw), n4, p.S, accumulate);
    return mmh::check_launch("lp16_slab_reduce_kernel");
}

}  // extern "C"
