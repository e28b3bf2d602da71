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
#include "lp16_common.h"

namespace {
using namespace mmh::lp16;

typedef float f32x16 __attribute__((ext_vector_type(16)));
template <bool H16>
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {        // 32x32x16: the flat-row wgrad below
    if (H16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0,
                                                      0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// conv_lp16p_kernel: the row-tile form (256 consecutive output pixels x 256 channels, the activation tile re-staged per tap) on
// the 16x16x32 MFMA - wave tile 128 x 64 = 8 x 4 tiles of 16x16, 4 accumulator VGPRs each; A / B fragment of lane l: row
// l & 15, k = 8 (l >> 4) .. +7 of a 32-deep step; C/D: col = l & 15, row = 4 (l >> 4) + reg.  What images SMALLER than the
// halo kernel's 16 x 16 tile take (and mmh_set_option("lp16_shape", 17) everywhere: the independent second implementation the
// tests compare the halo kernel with).  The LDS fragment reads are software-pipelined INTO the MFMA stream - an ablation of the
// unpipelined form showed read, wait and multiply phases running back to back, both waves of a SIMD leaving the barrier in
// lockstep - a wave's 32-deep step is: 4 MFMAs on A fragment i, then the
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

// transposed reads of [pixel][256 channels] LDS images (ds_read_b64_tr_b16) for the flat-row wgrad below
constexpr int WROWB = 512;                      // bytes per LDS row: 256 channels
typedef short s16x4 __attribute__((ext_vector_type(4)));

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

// ---------------------------------------------------------------------------------------------
// wgrad with flat (tap, channel) rows: the weight gradient of the stems (7x7, Cin 3..42), of the
// stride-2 3x3 convs and of ConvTranspose2d.  dw[(tap, ci)][co] = sum over output pixels of
// x[pixel*stride + tap - pad][ci] * dy[pixel][co]: GEMM rows are the flat index f = tap * C8 + ci (C8 =
// channels per tap, a multiple of 8 = one 16-byte chunk), tiled by 256; columns are co, tiled by
// NTW = 64 | 128 | 256; the contraction runs over output pixels in a ring of five half k-steps (32 pixels, 32 KiB
// each: two being multiplied, three in flight).  Every DMA lane of the x tile carries its own tap (its 16-byte chunk is 8 channels
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

// fprop / dgrad kernel (mmh_set_option "lp16_shape"): 19 = conv_lp16h2_kernel (default; conv_lp16_halo.hip: the activation halo
// of a 16x16 pixel tile resident in LDS for all nine taps), 17 = conv_lp16p_kernel everywhere (256-pixel row tiles, the
// activation tile re-staged per tap: what images smaller than 16x16 take under 19 as well), 20 = the halo kernel with one
// wave per SIMD (A/B builds only: make AB=1).  Earlier generations (16, 18, 32) were removed in round 5.
// wgrad kernel (mmh_set_option "lp16_wgrad_ring"): 2 = wgrad_lp16t_kernel (default: nine taps of a 64 x 128 tile resident, the
// input halo of a 4 x 16 pixel block staged once: wgrad_lp16t.hip), 3 = the same with both waves of a SIMD issuing their DMA
// behind the barrier.  The one-tap ring kernels (0, 1) were removed in round 5.
namespace mmh { int g_lp16_shape = 19; int g_lp16_tap_inner = 0; int g_lp16_dbg = 0; int g_lp16_wgrad_ring = 2; int g_lp16_persist = 1; int g_lp16_wgrad_s2 = 1; }
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

int mmh_lp16_clock_stamps(void* host_pairs_u64, int max_workgroups) {
    return mmh::lp16::clock_stamps(static_cast<unsigned long long*>(host_pairs_u64), max_workgroups);
}

int mmh_conv3x3_lp16_supported(const mmh_conv_desc* d) {
    return d && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->Cin % 64 == 0 && d->Cout % 64 == 0 &&
           d->Ho == d->H && d->Wo == d->W && (d->dtype == MMH_BF16 || d->dtype == MMH_FP16);
}

int mmh_conv3x3_lp16_fold_supported(const mmh_conv_desc* d) {
    return mmh_conv3x3_lp16_supported(d) && d->pad_mode == MMH_PAD_REFLECT && d->H % HT == 0 && d->W % HT == 0 &&
           d->H >= 2 * HT && d->W >= 2 * HT && d->Cin % TBN == 0 && (g_lp16_shape == 19 || g_lp16_shape == 20);
}

// mode 0: fprop  y[B,H,W,Cout] = conv(x16 [B,H,W,Cin], w16 = w_t [tap][Cout][Cin]) (+bias, act)
// mode 1: dgrad  dx[B,H,W,Cin] = zero-padded correlation of dy16 [B,H,W,Cout] with the flipped filter,
//                w16 = w_plain [tap][Cin][Cout]; for MMH_PAD_REFLECT this is the main term only
//                (the caller adds the border terms: mmh_conv2d_dgrad_border)
// mode 2: dgrad of a reflect-padded conv COMPLETE: mode 1 plus the pad ring's gradient folded onto rows 1 / H-2 and
//                columns 1 / W-2 inside the kernel (mmh_conv3x3_lp16_fold_supported; no border call follows)
struct LpNbr {      // the norm-backward sums taken by a dgrad's epilogue (LpConvKP::nbr_*)
    const void *x, *bits, *mean, *invstd;
    float* part;
    float dsc;
    int groups;
};

static int conv3x3_lp16_impl(const mmh_conv_desc* d, int mode, const void* x16, const void* w16, const void* bias,
                             void* y, int y_is16, int act, const void* zeros, void* stats, mmh_stream_t s,
                             const void* addend = nullptr, const LpNbr* nbr = nullptr) {
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
    if (nbr) {
        p.nbr_x = static_cast<const char*>(nbr->x); p.nbr_bits = static_cast<const uint16_t*>(nbr->bits);
        p.nbr_mean = static_cast<const float*>(nbr->mean); p.nbr_invstd = static_cast<const float*>(nbr->invstd);
        p.nbr_part = nbr->part; p.nbr_dsc = nbr->dsc; p.nbr_groups = nbr->groups;
    }
    const long long M = (long long)d->B * d->H * d->W;
    MMH_REQUIRE(M * (long long)std::max(p.cs, p.y_cs) < (1ll << 31) && d->H < 32768 && d->W < 65536,
                "mmh_conv3x3_lp16: tensor too large");
    p.MT = (int)((M + TBM - 1) / TBM);
    p.NT = N / TBN;
    const int per_xcd = (p.MT * p.NT + 7) / 8;
    MMH_REQUIRE(g_lp16_shape == 19 || g_lp16_shape == 20 || g_lp16_shape == 17,
                "mmh_conv3x3_lp16: lp16_shape %d is not in this build (19 halo kernel, 17 row tiles, 20 A/B builds)", g_lp16_shape);
    if (g_lp16_shape != 17 && d->H >= HT && d->W >= HT)
        return launch_conv_lp16_halo(p, d, mode, g_lp16_shape == 20, mmh::as_stream(s));
    // row tiles: images smaller than the halo kernel's tile (and lp16_shape 17)
    MMH_REQUIRE(mode != 2, "mmh_conv3x3_lp16: mode 2 needs the halo kernel");
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
    MMH_REQUIRE(!p.stats && !p.addend && !p.nbr_part, "mmh_conv3x3_lp16: epilogue statistics / addend / sums need the halo kernel");
    if (p.h16)
        hipLaunchKernelGGL(conv_lp16p_kernel<true>, dim3(8 * per_xcd), dim3(512), 2 * STAGE, mmh::as_stream(s), p);
    else
        hipLaunchKernelGGL(conv_lp16p_kernel<false>, dim3(8 * per_xcd), dim3(512), 2 * STAGE, mmh::as_stream(s), p);
    return mmh::check_launch("conv_lp16p_kernel");
}

int mmh_conv3x3_lp16(const mmh_conv_desc* d, int mode, const void* x16, const void* w16, const void* bias,
                     void* y, int y_is16, int act, const void* zeros, mmh_stream_t s) {
    return conv3x3_lp16_impl(d, mode, x16, w16, bias, y, y_is16, act, zeros, nullptr, s);
}

// dgrad (mode 1 | 2) with an fp32 dx that also receives `addend` (fp32, dx's layout): dx = dgrad(dy) + addend.  The
// input of such a conv has a second consumer (the residual stream: PATBlock out = x1 + ..., models/Generator.py:115-130;
// ResnetBlock out = x + conv_block(x), models/Discriminator.py:50) whose gradient autograd would add in a pass of its own.
int mmh_conv3x3_lp16_dgrad_add_supported(const mmh_conv_desc* d) {
    return mmh_conv3x3_lp16_supported(d) && d->H >= HT && d->W >= HT && d->Cin % TBN == 0 && (g_lp16_shape == 19 || g_lp16_shape == 20);
}

int mmh_conv3x3_lp16_dgrad_add(const mmh_conv_desc* d, int mode, const void* dy16, const void* w16, const void* addend,
                               void* dx, const void* zeros, mmh_stream_t s) {
    MMH_REQUIRE((mode == 1 || mode == 2) && addend && mmh_conv3x3_lp16_dgrad_add_supported(d),
                "mmh_conv3x3_lp16_dgrad_add: mode 1 | 2 on the halo kernel (H, W >= 16, Cin %% 256 == 0)");
    MMH_REQUIRE((reinterpret_cast<uintptr_t>(addend) & 15) == 0, "mmh_conv3x3_lp16_dgrad_add: addend must be 16-byte aligned");
    return conv3x3_lp16_impl(d, mode, dy16, w16, nullptr, dx, 0, MMH_ACT_NONE, zeros, nullptr, s, addend);
}

// dgrad (mode 1 | 2) with a 16-bit dx that is the gradient of a norm's OUTPUT (conv -> norm -> ReLU -> Dropout -> pad -> conv,
// models/Generator.py:66-77: the second conv's input gradient enters the first norm's backward): the epilogue also takes
// that norm's backward sums s1 = sum dz, s2 = sum dz * xhat (dz = keep ? g * dsc : 0 of the values as stored) per half tile,
// and mmh_norm_bwd_sums_final adds the partials up - mmh_norm_bwd_reduce's pass over g, x and the keep bits (4.25 bytes
// per element) becomes a read of x and the bits inside this epilogue.  xn: the norm's input, 16-bit [B,H,W,Cin] contiguous;
// bits: its keep bits (16 per 8 elements) or NULL; mean / invstd [groups][Cin], groups = B (instance) | 1 (batch).
// Returns the chunks per image (0: not available - ragged tiles, another kernel selected, a strided dx).
int mmh_conv3x3_lp16_dgrad_nbr_chunks(const mmh_conv_desc* d, int mode) {
    if (!mmh_conv3x3_lp16_dgrad_add_supported(d) || d->H % HT || d->W % HT || d->x_cs != d->Cin || d->Cin % 8) return 0;
    if (mode == 2 ? !mmh_conv3x3_lp16_fold_supported(d) : mode != 1) return 0;
    return 2 * (d->H / HT) * (d->W / HT);
}

int mmh_conv3x3_lp16_dgrad_nbr(const mmh_conv_desc* d, int mode, const void* dy16, const void* w16, void* dx16, const void* xn,
                               const void* bits, const void* mean, const void* invstd, int groups, float drop_p, void* s1,
                               void* s2, void* ws, size_t ws_bytes, const void* zeros, mmh_stream_t s) {
    const int cpi = mmh_conv3x3_lp16_dgrad_nbr_chunks(d, mode);
    MMH_REQUIRE(cpi > 0, "mmh_conv3x3_lp16_dgrad_nbr: needs the halo kernel, H and W multiples of 16, a contiguous dx "
                         "(ask mmh_conv3x3_lp16_dgrad_nbr_chunks)");
    MMH_REQUIRE(xn && mean && invstd && s1 && s2 && ws && (groups == 1 || groups == d->B) && drop_p >= 0.f && drop_p < 1.f &&
                    ws_bytes >= (size_t)d->B * cpi * 2 * d->Cin * sizeof(float) && (reinterpret_cast<uintptr_t>(ws) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(xn) & 15) == 0,
                "mmh_conv3x3_lp16_dgrad_nbr: bad arguments (groups 1 | B, ws >= B * chunks * 2 * Cin floats, 16-byte aligned)");
    LpNbr nbr{xn, bits, mean, invstd, static_cast<float*>(ws), bits ? 1.f / (1.f - drop_p) : 1.f, groups};
    if (int rc = conv3x3_lp16_impl(d, mode, dy16, w16, nullptr, dx16, 1, MMH_ACT_NONE, zeros, nullptr, s, nullptr, &nbr)) return rc;
    return mmh_norm_bwd_sums_final(ws, groups, d->Cin, groups == 1 ? d->B * cpi : cpi, s1, s2, s);
}

// fprop with a 16-bit output whose per-(image, half tile, channel) partial statistics come out of the epilogue:
// chunks per image = 2 * (H / 16) * (W / 16) (0: not available - ragged tiles or another kernel selected)
int mmh_conv3x3_lp16_stats_chunks(const mmh_conv_desc* d) {
    if (!mmh_conv3x3_lp16_supported(d) || d->H % HT || d->W % HT || d->Cout % TBN || (g_lp16_shape != 19 && g_lp16_shape != 20)) return 0;
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

// the 3x3 / stride-2 / zero-pad convs with Cin % 64 == 0, Cout % 128 == 0 (the downsampling convs and ConvTranspose2d's
// adjoint view) go to the nine-tap halo kernel's stride-2 form (wgrad_lp16t.hip) instead of the flat im2col rows, which
// stage every x pixel 2.25 times and pad 576 flat rows to 768 (mmh_set_option("lp16_wgrad_s2", 0): the flat kernel)
static bool wgrad_s2_halo(const mmh_conv_desc* d, int C8) {
    return mmh::g_lp16_wgrad_s2 && d->stride == 2 && C8 == d->Cin && mmh::wgrad_lp16t_supported(d);
}

size_t mmh_wgrad_lp16_flat_ws_bytes(const mmh_conv_desc* d, int C8) {
    if (!mmh_wgrad_lp16_flat_supported(d, C8)) return 0;
    int ntw, MT, NT, S, ksteps;
    lp16f_geometry(d, C8, ntw, MT, NT, S, ksteps);
    const size_t flat = (size_t)S * MT * 256 * d->Cout * sizeof(float);
    return wgrad_s2_halo(d, C8) ? std::max(flat, (size_t)mmh::wgrad_lp16t_splits(d) * 9 * d->Cin * d->Cout * sizeof(float)) : flat;
}

// dw [kh][kw][Cin][Cout] (fp32) (+)= wgrad of conv d from the 16-bit x16 [B][H][W][x_cs] (C8 channels per
// tap are read: C8 = Cin, or the padded width of mmh_lp16_pad_cvt for the stems, then x_cs = C8) and dy16.
int mmh_wgrad_lp16_flat(const mmh_conv_desc* d, const void* x16, int C8, int x_cs, const void* dy16, void* dw,
                        void* ws, size_t ws_bytes, int accumulate, const void* zeros, mmh_stream_t s) {
    MMH_REQUIRE(mmh_wgrad_lp16_flat_supported(d, C8) && x16 && dy16 && dw && ws && zeros && x_cs >= C8 && x_cs % 8 == 0,
                "mmh_wgrad_lp16_flat: square kernel, stride 1|2, Cout %% 64 == 0, C8 %% 8 == 0 >= Cin, 16-bit dtype");
    MMH_REQUIRE(ws_bytes >= mmh_wgrad_lp16_flat_ws_bytes(d, C8), "mmh_wgrad_lp16_flat: workspace too small");
    if (wgrad_s2_halo(d, C8)) {
        MMH_REQUIRE((long long)d->B * d->H * d->W * x_cs < (1ll << 31) && (long long)d->B * d->Ho * d->Wo * d->y_cs < (1ll << 31),
                    "mmh_wgrad_lp16_flat: tensor too large");
        mmh_conv_desc e = *d;
        e.x_cs = x_cs;
        hipStream_t st2 = mmh::as_stream(s);
        if (int rc = mmh::launch_wgrad_lp16t(&e, x16, dy16, static_cast<float*>(ws), zeros, st2)) return rc;
        const int64_t n4t = (int64_t)9 * d->Cin * d->Cout / 4;
        hipLaunchKernelGGL(lp16_slab_reduce_kernel, dim3((unsigned)std::min<int64_t>(mmh::cdiv(n4t, 256), 4096)), dim3(256),
                           0, st2, static_cast<const float*>(ws), static_cast<float*>(dw), n4t, mmh::wgrad_lp16t_splits(d),
                           accumulate);
        return mmh::check_launch("lp16_slab_reduce_kernel");
    }
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

size_t mmh_wgrad3x3_lp16_ws_bytes(const mmh_conv_desc* d) {
    if (!d || d->Cin % 256 || d->Cout % 256 || !mmh::wgrad_lp16t_supported(d)) return 0;
    return (size_t)mmh::wgrad_lp16t_splits(d) * 9 * d->Cin * d->Cout * sizeof(float);
}

// dw [3][3][Cin][Cout] (fp32) (+)= wgrad of the 3x3 / stride 1 / pad 1 conv from 16-bit x and dy (wgrad_lp16t.hip).
int mmh_wgrad3x3_lp16(const mmh_conv_desc* d, const void* x16, const void* dy16, void* dw, void* ws,
                      size_t ws_bytes, int accumulate, const void* zeros, mmh_stream_t s) {
    MMH_REQUIRE(mmh_conv3x3_lp16_supported(d) && x16 && dy16 && dw && ws && zeros && d->Cin % 256 == 0 &&
                    d->Cout % 256 == 0 && mmh::wgrad_lp16t_supported(d),
                "mmh_wgrad3x3_lp16: 3x3 / stride 1 / pad 1, Cin and Cout %% 256 == 0, 16-bit dtype (reflect: H, W >= 2)");
    MMH_REQUIRE(ws_bytes >= mmh_wgrad3x3_lp16_ws_bytes(d), "mmh_wgrad3x3_lp16: workspace too small");
    const long long P = (long long)d->B * d->H * d->W;
    MMH_REQUIRE(P * (long long)std::max(d->x_cs, d->y_cs) < (1ll << 31), "mmh_wgrad3x3_lp16: tensor too large");
    hipStream_t st = mmh::as_stream(s);
    if (int rc = mmh::launch_wgrad_lp16t(d, x16, dy16, static_cast<float*>(ws), zeros, st)) return rc;
    const int64_t n4t = (int64_t)9 * d->Cin * d->Cout / 4;
    hipLaunchKernelGGL(lp16_slab_reduce_kernel, dim3((unsigned)std::min<int64_t>(mmh::cdiv(n4t, 256), 4096)), dim3(256),
                       0, st, static_cast<const float*>(ws), static_cast<float*>(dw), n4t, mmh::wgrad_lp16t_splits(d), accumulate);
    return mmh::check_launch("lp16_slab_reduce_kernel");
}

}  // extern "C"
