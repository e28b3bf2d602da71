// Implicit-GEMM convolution kernels for gfx950 (MI355X), fp32 on the f32 MFMA.
//
// One im2col-free gather serves all three passes of every Conv2d /
// ConvTranspose2d on the MM-HAND hot path (reference: the cuDNN kernels behind
// nn.Conv2d / nn.ReflectionPad2d / nn.ConvTranspose2d at
// models/Generator.py:40-113,158-259, models/Discriminator.py:14-99,
// losses/L1_plus_perceptualLoss.py:22-27):
//
//   fprop : y[pix][co]   = sum_{tap,ci} x[src(pix,tap)][ci]  * w[tap][ci][co]
//   dgrad : dx[pix][ci]  = sum_{tap,co} dy[src'(pix,tap)][co] * w[tap][ci][co]
//   wgrad : dw[tap,ci][co] = sum_pix    x[src(pix,tap)][ci]  * dy[pix][co]
//
// Activations are NHWC, so the contraction index (tap, channel) is contiguous
// in groups of 4 channels and every global access is a 16-byte load.
// ReflectionPad2d / zero padding / stride / transposed-conv parity classes are
// all folded into src(): no padded tensor and no im2col buffer ever exists.
//
// Tiling (wave64, v_mfma_f32_32x32x2_f32): 256 threads = 4 waves per
// workgroup, block tile 128 x BN x 32, each wave owns TM x TN tiles of 32x32
// with 16 accumulator registers each.  Global loads for k-step i+1 are issued
// into registers before the MFMAs of k-step i and written to LDS after them.
#include <algorithm>
#include <cstring>
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// The 16-bit MFMA kernels serve two element types with one body: bf16 (MMH_BF16) and IEEE fp16
// (MMH_FP16, the reference's apex O1 precision).  Tiles are carried as 16-bit lanes typed bf16x*;
// the kernel argument `h16` (wave-uniform) selects, by ONE branch at the top of each kernel, the
// instantiation of the body with the matching conversion and MFMA opcode (a per-MFMA runtime select
// costs 30-80 VGPRs: the compiler then keeps both accumulator paths alive).
__device__ __forceinline__ bf16x4 to_bf16x4(float4 v) {
    bf16x4 r;
    r[0] = (__bf16)v.x; r[1] = (__bf16)v.y; r[2] = (__bf16)v.z; r[3] = (__bf16)v.w;
    return r;
}
template <bool H16>
__device__ __forceinline__ bf16x4 to_lp4(float4 v) {
    if (H16) {
        f16x4 r;
        r[0] = (_Float16)v.x; r[1] = (_Float16)v.y; r[2] = (_Float16)v.z; r[3] = (_Float16)v.w;
        return __builtin_bit_cast(bf16x4, r);
    }
    return to_bf16x4(v);
}
template <bool H16>
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
    if (H16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0,
                                                      0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// 4-channel element access of the Winograd-domain tensors: LP = 0 fp32 (float4), 1 bf16, 2 fp16
// (8 bytes, RNE)
template <int LP>
__device__ __forceinline__ void wst4(void* base, long long idx4, float4 v) {
    if (LP) reinterpret_cast<bf16x4*>(base)[idx4] = to_lp4<LP == 2>(v);
    else reinterpret_cast<float4*>(base)[idx4] = v;
}
template <int LP>
__device__ __forceinline__ float4 wld4(const void* base, long long idx4) {
    if (LP == 2) {
        const f16x4 r = reinterpret_cast<const f16x4*>(base)[idx4];
        return make_float4((float)r[0], (float)r[1], (float)r[2], (float)r[3]);
    }
    if (LP == 1) {
        const bf16x4 r = reinterpret_cast<const bf16x4*>(base)[idx4];
        return make_float4((float)r[0], (float)r[1], (float)r[2], (float)r[3]);
    }
    return reinterpret_cast<const float4*>(base)[idx4];
}

constexpr int BM = 128;   // rows (pixels, or (tap,ci) for wgrad) per workgroup
constexpr int BK = 32;    // contraction depth per k-step
constexpr int LDA = 36;   // LDS row pitch (floats) of a [row][k] tile: conflict-free ds_read_b128
constexpr int LDW = 132;  // LDS row pitch of the wgrad [pixel][128] tile
constexpr int TABP = 8;   // pitch of the per-row source-offset tables of conv_igemm_body (taps per axis <= 8)
constexpr size_t tab_bytes(int rows) { return (size_t)rows * TABP * 2 * sizeof(int); }

// How a (pixel, tap) pair maps to a source pixel.
struct Gather {
    const float* src;
    unsigned src_bytes;         // extent of the source buffer (buffer-load range check)
    int srcH, srcW;
    unsigned src_cs;            // elements per source pixel
    int PH, PW;                 // pixel domain per image (rows of the GEMM)
    int TH, TW;                 // tap grid
    int C4;                     // channel groups (of 4) per tap
    int ap_h, at_h, a0_h;       // v_h = ph*ap_h + th*at_h + a0_h
    int ap_w, at_w, a0_w;
    int shift;                  // v >>= shift after the v >= 0 test (stride-2 dgrad)
    int reflect;                // mirror v into [0, srcH)
    int chunk_major;            // k order: 1 = (super-chunk, tap, chunk), 0 = flat (tap, group)
    int cw;                     // chunks (of 8 groups = 32 channels) per tap visit, chunk_major only
    int dc4, dtw, dth;          // flat order: advance of (group, tap column, tap row) per k-step (kstate_next)
};

struct ConvKP {
    Gather g;
    const float* w;
    unsigned w_bytes;
    float* out;
    const float* bias;
    int M, N;                   // GEMM rows (batch*PH*PW) and columns
    int nk;                     // k-steps
    int Kflat;                  // TH*TW*C4*4 (flat order bound)
    int wCin, wCout;            // weight tensor dims [taps][wCin][wCout]
    int wRows, wKper;           // bf16 path: prepared weights are [taps][wRows][wKper] (k contiguous)
    int flat16;                 // bf16 path, small Cin: flat k = (tap, channel), weights [N][wKper]
    int KW_true, kh0, kw0, tstep;  // true tap = (kh0+tstep*th)*KW_true + kw0+tstep*tw
    int OH, OW, o_p, o0_h, o0_w;   // output pixel = (ph*o_p+o0_h, pw*o_p+o0_w) in OHxOW
    unsigned out_cs;
    int out_linear;             // output row offset is simply m*out_cs
    int act;
    int dbg;                    // timing-only ablation bits (mmh_set_option "conv_dbg"): results wrong
    int xcd_remap;              // remap (blockIdx.y, blockIdx.x) so column tiles of a row tile share an XCD
    int accum;                  // epilogue adds into out instead of overwriting it
    int h16;                    // 16-bit kernels: 0 = bf16, 1 = fp16 operands
    int src16;                  // 16-bit kernels: the gathered tensor is already 16-bit in HBM (chunk-major k order only)
    float* stats;               // fp32 fprop: per (row tile, wave row) the count / mean / M2 of every output column
                                // [M/BM * WAVES_M][3][N] - the norm behind the conv merges them instead of reading y
};

struct KState { int th, tw, c4, j; };   // j: chunk index inside the current tap visit

__device__ __forceinline__ void kstate_init(KState& s, const Gather& g, int grp) {
    s.j = 0;
    if (g.chunk_major) { s.th = 0; s.tw = 0; s.c4 = grp; }
    else {
        int tap = grp / g.C4;
        s.c4 = grp - tap * g.C4;
        s.th = tap / g.TW;
        s.tw = tap - s.th * g.TW;
    }
}
__device__ __forceinline__ void kstate_next(KState& s, const Gather& g) {
    if (g.chunk_major) {
        // order (super-chunk of cw chunks, tap, chunk): a tap's row offsets serve cw k-steps
        s.c4 += 8;
        if (++s.j == g.cw) {
            s.j = 0;
            s.c4 -= 8 * g.cw;
            if (++s.tw == g.TW) { s.tw = 0; if (++s.th == g.TH) { s.th = 0; s.c4 += 8 * g.cw; } }
        }
    } else {
        // 8 groups further in the flat (tap, group) order: fixed deltas with two carries, no loop
        // (dc4 = 8 % C4, dtw = (8 / C4) % TW, dth = (8 / C4) / TW; set_korder)
        s.c4 += g.dc4;
        const int cy = s.c4 >= g.C4 ? 1 : 0;
        s.c4 -= cy ? g.C4 : 0;
        s.tw += g.dtw + cy;
        const int cy2 = s.tw >= g.TW ? 1 : 0;
        s.tw -= cy2 ? g.TW : 0;
        s.th += g.dth + cy2;
    }
}
__device__ __forceinline__ bool kstate_valid(const KState& s, const Gather& g) {
    return g.chunk_major ? (s.c4 < g.C4) : (s.th < g.TH);
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned OOB = 0xFFFFFFF0u;   // byte offset beyond any buffer: the load returns zeros

// 16-byte buffer load: an out-of-range offset yields zeros, so padding, ragged tiles and the
// k tail need no branches (raw buffer, hardware range check against num_records).
__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
    u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0);
    return __builtin_bit_cast(float4, r);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, bytes, 0x00020000);
}

// Byte offset of the source pixel (channel 0) for (pixel base bh,bw ; tap th,tw), or OOB.
__device__ __forceinline__ unsigned gather_base(const Gather& g, unsigned img_base, int bh, int bw,
                                                const KState& s, bool ok) {
    int vh = bh + s.th * g.at_h;
    int vw = bw + s.tw * g.at_w;
    if (g.reflect) {
        vh = vh < 0 ? -vh : vh;
        vw = vw < 0 ? -vw : vw;
        vh = vh >= g.srcH ? 2 * (g.srcH - 1) - vh : vh;
        vw = vw >= g.srcW ? 2 * (g.srcW - 1) - vw : vw;
    } else {
        ok = ok && (vh >= 0) && (vw >= 0);
        vh >>= g.shift;
        vw >>= g.shift;
        ok = ok && (vh < g.srcH) && (vw < g.srcW);
    }
    const unsigned off = (img_base + (unsigned)(vh * g.srcW + vw)) * g.src_cs * 4u;
    return ok ? off : OOB;
}

// Byte offset of the 4 channels of the source pixel for (pixel base bh,bw ; tap th,tw), or OOB.
__device__ __forceinline__ unsigned gather_off(const Gather& g, unsigned img_base, int bh, int bw,
                                               const KState& s, bool ok) {
    int vh = bh + s.th * g.at_h;
    int vw = bw + s.tw * g.at_w;
    if (g.reflect) {
        vh = vh < 0 ? -vh : vh;
        vw = vw < 0 ? -vw : vw;
        vh = vh >= g.srcH ? 2 * (g.srcH - 1) - vh : vh;
        vw = vw >= g.srcW ? 2 * (g.srcW - 1) - vw : vw;
    } else {
        ok = ok && (vh >= 0) && (vw >= 0);
        vh >>= g.shift;
        vw >>= g.shift;
        ok = ok && (vh < g.srcH) && (vw < g.srcW);
    }
    const unsigned off = ((img_base + (unsigned)(vh * g.srcW + vw)) * g.src_cs + (unsigned)s.c4 * 4u) * 4u;
    return ok ? off : OOB;
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == MMH_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == MMH_ACT_TANH) return tanhf(v);
    return v;
}

// ---------------------------------------------------------------------------
// fprop / dgrad kernel.  B_NMAJOR=false: weights tile is [k][n] in LDS (fprop,
// rows of w are contiguous in n).  B_NMAJOR=true: tile is [n][k] (dgrad: for a
// fixed ci the co run is contiguous in w).  DBUF: two LDS buffers and one
// barrier per k-step (2 workgroups/CU) instead of one buffer and two barriers
// (3 workgroups/CU; measured 5-6 % faster on the 512-channel shapes).
// ---------------------------------------------------------------------------
// LEVELS = 2 (fprop form only; mmh_set_option("conv_levels", 2)): two-level summation over the contraction, as in
// wino_gemm_kernel below - every k-step (32 of the (tap, channel) index) starts a fresh MFMA chain (C operand = 0) that is
// folded into the totals by vector adds, one tile row at a time.  A k-ordered fp32 chain over K = 9 * 512 carries a relative
// rounding error of ~u sqrt(K) / 2.4 = 1.1e-6 (oneDNN's blocked accumulation on the CPU: 2.4e-7), and at 256x256 the
// Generator's parameter gradients follow the FORWARD's error: the gradient-exact hybrid runs its forward 3x3
// convolutions on this form.
template <int BN, int WAVES_M, int WAVES_N, bool B_NMAJOR, bool DBUF, int BMT = 128, int LEVELS = 1>   // BMT: rows per workgroup
__device__ __forceinline__ void conv_igemm_body(const ConvKP& p, const int bx, const int by,
                                                const int gx, const int gy) {
    constexpr int WTM = BMT / WAVES_M, WTN = BN / WAVES_N;
    constexpr int RA = BMT / 32;    // A rows per thread
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int NB = BN / 32;  // float4 B loads per thread per k-step
    static_assert(WAVES_M * WAVES_N == 4, "4 waves");

    constexpr int NBUF = DBUF ? 2 : 1;
    constexpr bool IL = B_NMAJOR;   // dgrad: loads interleaved with the MFMA groups (+1-2 % measured)
    constexpr int ASZ = BMT * LDA;
    constexpr int BSZ = B_NMAJOR ? BN * LDA : BK * BN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As_base = smem;
    float* const Bs_base = smem + NBUF * ASZ;

    const Gather& g = p.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // XCD-aware tile mapping: workgroups are dealt round-robin over the 8 XCDs, so ids L, L+8,
    // L+16, ... share an L2.  XCD x walks its own contiguous band of row tiles, visiting all the
    // column tiles of a row tile back to back: the gathered A rows (and the halo rows shared with
    // the next row tile) are fetched into ONE L2 instead of several.  Speed/traffic only.
    int mt = by, nt = bx;
    if (p.xcd_remap) {
        const int band = gy / 8;                             // row tiles per XCD (whole bands)
        const int L = by * gx + bx;
        if (L < band * 8 * gx) {                             // the ragged rest keeps its ids
            const int x = L & 7, i = L >> 3;
            nt = i % gx;
            mt = x * band + i / gx;
        }
    }
    const int m0 = mt * BMT;
    const int n0 = nt * BN;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(g.src, g.src_bytes);
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(p.w, p.w_bytes);

    // --- per-thread A rows: RA rows, one k-group (tid&7) ---
    const int grp = tid & 7;
    unsigned a_img[RA];
    int a_bh[RA], a_bw[RA];
    bool a_ok[RA];
    const int PHW = g.PH * g.PW;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        int m = m0 + (tid >> 3) + 32 * i;
        a_ok[i] = m < p.M;
        int mm = a_ok[i] ? m : 0;
        int b = mm / PHW;
        int r = mm - b * PHW;
        int ph = r / g.PW;
        int pw = r - ph * g.PW;
        a_img[i] = (unsigned)b * (unsigned)(g.srcH * g.srcW);
        a_bh[i] = ph * g.ap_h + g.a0_h;
        a_bw[i] = pw * g.ap_w + g.a0_w;
    }
    // B tile, NMAJOR: this thread's weight rows (ci) are fixed: byte offset of (tap 0, ci, co 0)
    unsigned b_row[NB];
    if (B_NMAJOR) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int n = n0 + (tid >> 3) + 32 * i;
            b_row[i] = n < p.N ? (unsigned)n * (unsigned)p.wCout * 4u : OOB;
        }
    }
    // B tile, KMAJOR: column part of the byte offset
    unsigned b_col[NB];
    int b_krow[NB];
    if (!B_NMAJOR) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + 256 * i;
            b_krow[i] = idx / (BN / 4);
            const int n = n0 + 4 * (idx - b_krow[i] * (BN / 4));
            b_col[i] = n < p.N ? (unsigned)n * 4u : OOB;
        }
    }

    // Flat (tap, channel) k order (small Cin: the 7x7 stems, 3 -> 64 convs): every k-step is another tap, and
    // recomputing RA reflected / bounds-checked source offsets per thread and k-step costs 9 % of the kernel
    // (tools/ablate_narrow.py).  Instead the workgroup tabulates once, per tile row, the source row of
    // each vertical tap and the source column of each horizontal tap (-1 = padding): two LDS reads per
    // row and k-step replace the arithmetic.
    int* const tabH = reinterpret_cast<int*>(smem + NBUF * (ASZ + BSZ));     // [BMT][TABP] (img*srcH + vh)*srcW
    int* const tabW = tabH + BMT * TABP;                                    // [BMT][TABP] vw
    const bool use_tab = !g.chunk_major && g.TH <= TABP && g.TW <= TABP && !(p.dbg & 8);
    if (use_tab) {
        for (int r = tid; r < BMT; r += 256) {
            const int m = m0 + r;
            const bool okm = m < p.M;
            const int mm = okm ? m : 0;
            const int b = mm / PHW;
            const int rr = mm - b * PHW;
            const int ph = rr / g.PW;
            const int pw = rr - ph * g.PW;
            const int bh = ph * g.ap_h + g.a0_h, bw = pw * g.ap_w + g.a0_w;
#pragma unroll
            for (int t = 0; t < TABP; ++t) {
                int vh = bh + t * g.at_h, vw = bw + t * g.at_w;
                bool okh = okm && t < g.TH, okw = t < g.TW;
                if (g.reflect) {
                    vh = vh < 0 ? -vh : vh; vw = vw < 0 ? -vw : vw;
                    vh = vh >= g.srcH ? 2 * (g.srcH - 1) - vh : vh;
                    vw = vw >= g.srcW ? 2 * (g.srcW - 1) - vw : vw;
                } else {
                    okh = okh && vh >= 0; okw = okw && vw >= 0;
                    vh >>= g.shift; vw >>= g.shift;
                    okh = okh && vh < g.srcH; okw = okw && vw < g.srcW;
                }
                tabH[r * TABP + t] = okh ? (b * g.srcH + vh) * g.srcW : -1;
                tabW[r * TABP + t] = okw ? vw : -1;
            }
        }
        __syncthreads();
    }

    KState ks_t;   // this thread's k-group
    kstate_init(ks_t, g, grp);
    unsigned a_off[RA];   // row byte offsets of the current tap
#pragma unroll
    for (int i = 0; i < RA; ++i) a_off[i] = OOB;

    float4 ra[RA];
    float4 rb[NB];

    // The loader is cut into 4 parts so that it can either run as one burst before the MFMAs
    // (load_tiles) or be interleaved with the 4 MFMA groups of the k-step (p.interleave).
    auto load_part = [&](int ks, int part) {
        const bool kv = kstate_valid(ks_t, g);
        if (part == 0) {
            if (use_tab) {
                const bool kvt = kstate_valid(ks_t, g);
#pragma unroll
                for (int i = 0; i < RA; ++i) {
                    const int row = (tid >> 3) + 32 * i;
                    const int hv = tabH[row * TABP + (kvt ? ks_t.th : 0)], wv = tabW[row * TABP + ks_t.tw];
                    a_off[i] = (hv >= 0 && wv >= 0) ? (unsigned)(hv + wv) * g.src_cs * 4u : OOB;
                }
            } else if (ks_t.j == 0 && !((p.dbg & 4) && ks > 1)) {   // new tap: recompute the row offsets
#pragma unroll
                for (int i = 0; i < RA; ++i)
                    a_off[i] = gather_base(g, a_img[i], a_bh[i], a_bw[i], ks_t, a_ok[i]);
            }
        } else if (part == 1) {
#pragma unroll
            for (int i = 0; i < RA; ++i)
                ra[i] = bload4(rsA, (kv && a_off[i] != OOB) ? a_off[i] + (unsigned)ks_t.c4 * 16u : OOB);
        } else {
            constexpr int H0 = (NB + 1) / 2;
            const int i0 = part == 2 ? 0 : H0, i1 = part == 2 ? H0 : NB;
            if (!B_NMAJOR) {
                // weight rows [rowbase, rowbase+32) x columns [n0, n0+BN)
                int rowbase, rowlim;
                if (g.chunk_major) {   // all threads share (th,tw) and the 8-group chunk
                    rowbase = (ks_t.th * g.TW + ks_t.tw) * p.wCin + (ks_t.c4 - grp) * 4;
                    rowlim = rowbase + BK;
                } else {
                    rowbase = ks * BK;
                    rowlim = p.Kflat;
                }
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    if (i < i0 || i >= i1) continue;
                    const int row = rowbase + b_krow[i];
                    const unsigned off = (unsigned)row * (unsigned)p.wCout * 4u + b_col[i];
                    rb[i] = bload4(rsW, (row < rowlim && b_col[i] != OOB) ? off : OOB);
                }
            } else {
                // B[k=(tap,co)][n=ci] = w[(tap_true*wCin + ci)*wCout + co]; same k-group as A
                const int tap_true = (p.kh0 + p.tstep * ks_t.th) * p.KW_true + p.kw0 + p.tstep * ks_t.tw;
                const unsigned tap_off = ((unsigned)tap_true * (unsigned)p.wCin * (unsigned)p.wCout +
                                          (unsigned)ks_t.c4 * 4u) * 4u;
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    if (i < i0 || i >= i1) continue;
                    rb[i] = bload4(rsW, (kv && b_row[i] != OOB) ? tap_off + b_row[i] : OOB);
                }
            }
        }
    };
    auto load_tiles = [&](int ks) {
#pragma unroll
        for (int part = 0; part < 4; ++part) load_part(ks, part);
    };

    auto store_tiles = [&](int buf) {
        float* As = As_base + buf * ASZ;
        float* Bs = Bs_base + buf * BSZ;
#pragma unroll
        for (int i = 0; i < RA; ++i)
            *reinterpret_cast<float4*>(&As[((tid >> 3) + 32 * i) * LDA + grp * 4]) = ra[i];
        if (!B_NMAJOR) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                int idx = tid + 256 * i;
                *reinterpret_cast<float4*>(&Bs[idx * 4]) = rb[i];  // [krow][BN] is linear in idx
            }
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i)
                *reinterpret_cast<float4*>(&Bs[((tid >> 3) + 32 * i) * LDA + grp * 4]) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (p.nk > 0) {
        load_tiles(0);
        store_tiles(0);
    }
    __syncthreads();
    for (int ks = 0; ks < p.nk; ++ks) {
        const float* As = As_base + (DBUF ? (ks & 1) : 0) * ASZ;
        const float* Bs = Bs_base + (DBUF ? (ks & 1) : 0) * BSZ;
        const bool more = ks + 1 < p.nk;
        if (more) {
            if (!(p.dbg & 4)) kstate_next(ks_t, g);        // dbg 4: loads still issue, from stale addresses
            if (!(p.dbg & 1) && !IL) load_tiles(ks + 1);   // global -> registers, one burst
        }
        if constexpr (LEVELS == 2 && !B_NMAJOR) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                f32x16 part[TN];
#pragma unroll
                for (int kg = 0; kg < 4; ++kg) {
                    const float4 av = *reinterpret_cast<const float4*>(&As[(wm * WTM + i * 32 + l31) * LDA + kg * 8 + h * 4]);
                    float bsc[TN][4];
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) bsc[j][e] = Bs[(kg * 8 + h * 4 + e) * BN + wn * WTN + j * 32 + l31];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a = e == 0 ? av.x : e == 1 ? av.y : e == 2 ? av.z : av.w;
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            part[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bsc[j][e], (kg == 0 && e == 0) ? f32x16{} : part[j],
                                                                           0, 0, 0);
                    }
                }
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += part[j][r];
                __builtin_amdgcn_sched_barrier(0);      // row i + 1's chains must not start before this fold
            }
        } else {
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            if (IL && more) load_part(ks + 1, kg);         // ... or spread over the MFMA groups
            float4 av[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                av[i] = *reinterpret_cast<const float4*>(
                    &As[(wm * WTM + i * 32 + l31) * LDA + kg * 8 + h * 4]);
            float bsc[TN][4];
            if (B_NMAJOR) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float4 t = *reinterpret_cast<const float4*>(
                        &Bs[(wn * WTN + j * 32 + l31) * LDA + kg * 8 + h * 4]);
                    bsc[j][0] = t.x; bsc[j][1] = t.y; bsc[j][2] = t.z; bsc[j][3] = t.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        bsc[j][e] = Bs[(kg * 8 + h * 4 + e) * BN + wn * WTN + j * 32 + l31];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float a = e == 0 ? av[i].x : e == 1 ? av[i].y : e == 2 ? av[i].z : av[i].w;
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bsc[j][e], acc[i][j], 0, 0, 0);
                }
            }
        }
        }
        if (!(p.dbg & 2)) {
            if (!DBUF) __syncthreads();      // everyone is done reading the single buffer
            if (ks + 1 < p.nk) store_tiles(DBUF ? ((ks + 1) & 1) : 0);
            __syncthreads();
        }
    }

    // --- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ---
    // (the bias of the lane's TN columns is read ONCE: inside the store loop the compiler has to reload it per element -
    // it cannot prove that the stores to p.out leave p.bias alone - and waits for every reload)
    float bv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WTN + j * 32 + l31;
        bv[j] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m >= p.M) continue;
            unsigned off;
            if (p.out_linear) off = (unsigned)m * p.out_cs;
            else {
                int b = m / PHW;
                int rr = m - b * PHW;
                int ph = rr / g.PW;
                int pw = rr - ph * g.PW;
                off = ((unsigned)(b * p.OH + ph * p.o_p + p.o0_h) * (unsigned)p.OW +
                       (unsigned)(pw * p.o_p + p.o0_w)) * p.out_cs;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 32 + l31;
                if (n < p.N) {
                    float v = acc[i][j][r] + bv[j];
                    if (p.accum) v += p.out[off + n];
                    p.out[off + n] = apply_act(v, p.act);
                }
            }
        }
    }
    if (p.stats) {
        // statistics of this wave's WTM x WTN block of y per column (host: every row tile is full, no activation):
        // a lane holds TM*16 rows of column n, its partner lane ^ 32 the other half; Chan merge of the two
        constexpr float CNT = (float)(TM * 16);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WTN + j * 32 + l31;
            const float b = (p.bias && n < p.N) ? p.bias[n] : 0.f;
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[i][j][r] + b;
            const float mean_l = sum * (1.f / CNT);
            float m2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float dv = acc[i][j][r] + b - mean_l;
                    m2 += dv * dv;
                }
            const float mean_o = __shfl_xor(mean_l, 32, 64), m2_o = __shfl_xor(m2, 32, 64);
            const float dm = mean_o - mean_l;
            if (h == 0 && n < p.N) {
                float* o = p.stats + ((size_t)(mt * WAVES_M + wm) * 3) * p.N + n;
                o[0] = 2.f * CNT;
                o[p.N] = mean_l + 0.5f * dm;
                o[2 * p.N] = m2 + m2_o + dm * dm * (0.5f * CNT);
            }
        }
    }
}

template <int BN, int WAVES_M, int WAVES_N, bool B_NMAJOR, bool DBUF>
__global__ void __launch_bounds__(256, 2) conv_igemm_kernel(const ConvKP p) {
    conv_igemm_body<BN, WAVES_M, WAVES_N, B_NMAJOR, DBUF>(p, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y);
}

// the fprop form with two-level summation (LEVELS = 2 above): 128 x 128 block tile, wave tile 64 x 64
__global__ void __launch_bounds__(256, 2) conv_igemm_levels2_kernel(const ConvKP p) {
    conv_igemm_body<128, 2, 2, false, false, 128, 2>(p, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y);
}

// 256-row tiles for the narrow GEMMs (N <= 64: 7x7 stems, stride-2 dgrad): per MFMA half the weight
// tile traffic, barriers and address work of the 128-row tile
template <int BN, int WAVES_M, int WAVES_N, bool B_NMAJOR>
__global__ void __launch_bounds__(256, 2) conv_igemm_tall_kernel(const ConvKP p) {
    conv_igemm_body<BN, WAVES_M, WAVES_N, B_NMAJOR, false, 256>(p, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y);
}

// Several small problems in one launch (blockIdx.z picks the piece): the border terms of the
// reflect-pad dgrad.  Each piece has its own extent; surplus workgroups exit at once.
constexpr int MAXP = 9;
struct MultiKP {
    ConvKP p[MAXP];
    int start[MAXP + 1];    // first workgroup id of each piece (prefix sums); 1-D grid
    int n;
};
__device__ __forceinline__ int multi_piece(const MultiKP& mp, int L) {
    int k = 0;
    while (k + 1 < mp.n && L >= mp.start[k + 1]) ++k;
    return k;
}
template <int BN, int WAVES_M, int WAVES_N, bool B_NMAJOR, int BMT = 128>
__global__ void __launch_bounds__(256, 2) conv_igemm_multi_kernel(const MultiKP mp) {
    const int k = multi_piece(mp, blockIdx.x);
    const ConvKP& p = mp.p[k];
    const int gx = (p.N + BN - 1) / BN, gy = (p.M + BMT - 1) / BMT;
    const int local = blockIdx.x - mp.start[k];
    conv_igemm_body<BN, WAVES_M, WAVES_N, B_NMAJOR, false, BMT>(p, local % gx, local / gx, gx, gy);
}

// Batched plain GEMMs (blockIdx.z = batch): the 16 Winograd-domain products.  Each batch is
// the same problem on pointers advanced by fixed strides.
struct BatchKP {
    ConvKP p;
    long long src_bs, w_bs, out_bs;     // element strides between batches
};
template <int BN, int WAVES_M, int WAVES_N, bool B_NMAJOR>
__global__ void __launch_bounds__(256, 2) conv_igemm_batched_kernel(const BatchKP bp) {
    ConvKP p = bp.p;
    const long long z = blockIdx.z;
    p.g.src += z * bp.src_bs;
    p.w += z * bp.w_bs;
    p.out += z * bp.out_bs;
    conv_igemm_body<BN, WAVES_M, WAVES_N, B_NMAJOR, false>(p, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y);
}

// ---------------------------------------------------------------------------
// Winograd-domain GEMMs, dedicated kernel: P planes of C[M x N] = A[M x K] . B[K x N], all three
// row-major and dense (V [P][tiles][K], U [P][K][N] -> M [P][tiles][N]).  Same MFMA tiling and
// LDS layout as conv_igemm_body, but
//   * plain row/column addressing (no tap / reflect / stride arithmetic): one add per k-step;
//   * PERSISTENT workgroups: the contraction is short (K = 64..512 channels = 2..16 k-steps), so a
//     one-tile-per-workgroup launch spends a third of its life filling the pipeline and storing
//     the tile.  Here each workgroup walks a list of tiles with ONE software pipeline running
//     across tile boundaries: the loads of the next tile's first k-step are in flight while the
//     current tile's accumulators are stored;
//   * XCD-aware work list: XCD x owns a contiguous range of (plane, row tile, column tile) items
//     with the column tile fastest, and its resident workgroups take consecutive items, so the
//     column tiles of one A row panel and the whole B plane are served by one L2.
// Requires K % 32 == 0 and N % 32 == 0; the M tail is handled by the buffer range check (loads
// return 0, stores are dropped), whole column groups past N are skipped.
// ---------------------------------------------------------------------------
struct WinoGemmKP {
    const float* A;
    const float* B;
    float* C;
    int M, K, N, P;
    int MT, NT;         // row / column tiles per plane
    int W;              // work items = P * MT * NT
    int Wx;             // items per XCD
    int nb;             // workgroups per XCD (grid = 8 * nb)
};

// LEVELS = 2: two-level summation over the contraction.  The f32 MFMA chain is a k-ordered fmaf
// chain, so one accumulator over K = 512 carries a rounding error ~ eps*sqrt(sum_k k); the F(6x6,3x3)
// output transform (coefficients up to 32 per dimension) then amplifies it - that chain, not the
// transforms, is 85 % of the F(6x6,3x3) error against fp64 (tools/wino_error_model.py).  With
// LEVELS = 2 every k-step (32 channels: close to the optimal block sqrt(K)) starts a fresh chain in
// `part` (C operand = 0) that is folded into `acc` by 64 vector adds: ~3x less accumulated rounding.
// The chains are run one tile ROW at a time, so `part` is 32 VGPRs and 3 workgroups per CU remain.
constexpr int WINO_FOLD = 1;
template <int BN, int LEVELS>
__global__ void __launch_bounds__(256, 3) wino_gemm_kernel(const WinoGemmKP p) {
    constexpr int WTM = 64, WTN = BN / 2;
    constexpr int TM = 2, TN = WTN / 32;
    constexpr int NB = BN / 32;
    constexpr int ASZ = BM * LDA;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As = smem;
    float* const Bs = smem + ASZ;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int grp = tid & 7;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int wend = min(p.W, (xcd + 1) * p.Wx);
    int wc = xcd * p.Wx + slot;             // item being computed
    if (wc >= wend) return;
    const int KS = p.K / BK;
    const unsigned a_bytes = (unsigned)p.M * (unsigned)p.K * 4u;
    const unsigned b_bytes = (unsigned)p.K * (unsigned)p.N * 4u;
    const unsigned c_bytes = (unsigned)p.M * (unsigned)p.N * 4u;
    const unsigned a_step = BK * 4u, b_step = (unsigned)BK * (unsigned)p.N * 4u;

    // loader state (runs one k-step ahead of the MFMAs, possibly already in the next item)
    __amdgpu_buffer_rsrc_t rsA, rsB;
    unsigned a_off[4], b_off[NB];
    auto setup_load = [&](int w) {
        const int nt = w % p.NT;
        const int t = w / p.NT;
        const int mt = t % p.MT;
        const int xi = t / p.MT;
        rsA = make_rsrc(p.A + (size_t)xi * p.M * p.K, a_bytes);
        rsB = make_rsrc(p.B + (size_t)xi * p.K * p.N, b_bytes);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mt * BM + (tid >> 3) + 32 * i;
            a_off[i] = ((unsigned)m * (unsigned)p.K + grp * 4u) * 4u;   // m >= M: beyond num_records -> 0
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + 256 * i;
            const int krow = idx / (BN / 4);
            const int n = nt * BN + 4 * (idx - krow * (BN / 4));
            b_off[i] = n < p.N ? ((unsigned)krow * (unsigned)p.N + (unsigned)n) * 4u : OOB;
        }
    };
    float4 ra[4], rb[NB];
    auto issue_loads = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = bload4(rsA, a_off[i]);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = bload4(rsB, b_off[i]);
    };
    auto advance_k = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) a_off[i] += a_step;
#pragma unroll
        for (int i = 0; i < NB; ++i) b_off[i] = b_off[i] == OOB ? OOB : b_off[i] + b_step;
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<float4*>(&As[((tid >> 3) + 32 * i) * LDA + grp * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<float4*>(&Bs[(tid + 256 * i) * 4]) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    setup_load(wc);
    issue_loads();
    store_tiles();
    __syncthreads();
    for (;;) {
        for (int ks = 0; ks < KS; ++ks) {
            bool more = true;
            if (ks + 1 < KS) advance_k();
            else {
                more = wc + p.nb < wend;
                if (more) setup_load(wc + p.nb);
            }
            if (more) issue_loads();
            if constexpr (LEVELS == 2) {
                // two-level summation, one tile ROW at a time: the k-step's chain of row i lives in
                // part[TN] (C operand = 0 at its start) and is folded into acc[i] before row i+1 starts,
                // so the second accumulator set costs TN*16 = 32 VGPRs, not 64 (3 workgroups per CU stay)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    f32x16 part[TN];
#pragma unroll
                    for (int kg = 0; kg < 4; ++kg) {
                        const float4 av =
                            *reinterpret_cast<const float4*>(&As[(wm * WTM + i * 32 + l31) * LDA + kg * 8 + h * 4]);
                        float bsc[TN][4];
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                bsc[j][e] = Bs[(kg * 8 + h * 4 + e) * BN + wn * WTN + j * 32 + l31];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float a = e == 0 ? av.x : e == 1 ? av.y : e == 2 ? av.z : av.w;
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                part[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                    a, bsc[j][e], (kg == 0 && e == 0) ? f32x16{} : part[j], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] += part[j][r];
                    __builtin_amdgcn_sched_barrier(0);      // row i+1's chains must not start before this fold
                }
            } else {
#pragma unroll
                for (int kg = 0; kg < 4; ++kg) {
                    float4 av[TM];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        av[i] = *reinterpret_cast<const float4*>(&As[(wm * WTM + i * 32 + l31) * LDA + kg * 8 + h * 4]);
                    float bsc[TN][4];
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) bsc[j][e] = Bs[(kg * 8 + h * 4 + e) * BN + wn * WTN + j * 32 + l31];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            const float a = e == 0 ? av[i].x : e == 1 ? av[i].y : e == 2 ? av[i].z : av[i].w;
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bsc[j][e], acc[i][j], 0, 0, 0);
                        }
                    }
                }
            }
            if (ks == KS - 1) {
                // tile done: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Rows past M
                // land beyond num_records and are dropped by the buffer range check; whole 32-column
                // groups past N (N % 32 == 0) are skipped by a wave-uniform test.
                const int nt = wc % p.NT;
                const int t = wc / p.NT;
                const int mt = t % p.MT;
                const int xi = t / p.MT;
                const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(
                    p.C + (size_t)xi * p.M * p.N, 0, c_bytes, 0x00020000);
                const unsigned n4 = (unsigned)p.N * 4u;
                const int ncol0 = nt * BN + wn * WTN;
                unsigned vbase = (unsigned)(mt * BM + wm * WTM + 4 * h) * n4 + (unsigned)(ncol0 + l31) * 4u;
                asm volatile("" : "+v"(vbase));     // keep the store addressing inside this block (no hoisting)
                // the scalar offset of a buffer store is not range-checked: use it only for tiles fully inside M
                if (mt * BM + BM <= p.M) {
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        if (ncol0 + j * 32 < p.N) {
#pragma unroll
                            for (int i = 0; i < TM; ++i)
#pragma unroll
                                for (int r = 0; r < 16; ++r)
                                    __builtin_amdgcn_raw_buffer_store_b32(
                                        __float_as_uint(acc[i][j][r]), rsC, vbase + j * 128u,
                                        (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2)) * n4, 0);
                        }
                } else {
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        if (ncol0 + j * 32 < p.N) {
#pragma unroll
                            for (int i = 0; i < TM; ++i)
#pragma unroll
                                for (int r = 0; r < 16; ++r)
                                    __builtin_amdgcn_raw_buffer_store_b32(
                                        __float_as_uint(acc[i][j][r]), rsC,
                                        vbase + (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2)) * n4 + j * 128u, 0, 0);
                        }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            }
            __syncthreads();
            if (more) store_tiles();
            __syncthreads();
        }
        wc += p.nb;
        if (wc >= wend) break;
    }
}

// ---------------------------------------------------------------------------
// Winograd-domain wgrad GEMMs, dedicated kernel: slab[xi][split][Cin x Cout] = V[xi][t0:t1]^T . Yh[xi][t0:t1]
// (contraction over a range of tiles).  The TN counterpart of wino_gemm_kernel: same 128x128x32
// block tile, plain addressing (one add per k-step; the end of a split's tile range is enforced by
// the buffer descriptor's num_records, so the last k-step needs no mask), persistent workgroups
// with one software pipeline across work items, XCD-contiguous work list with the 128x128 output
// tiles of one (plane, split) - which read the same two panels - adjacent.  Requires
// Cin % 128 == 0 and Cout % 128 == 0; the generic conv_wgrad_kernel covers the rest.
// ---------------------------------------------------------------------------
struct WinoWgradKP {
    const float* V;
    const float* Y;
    float* slab;            // [P][S][Cin][Cout]
    int T;                  // tiles per plane
    int Cin, Cout, P, S;
    int t_per_split;        // multiple of BK
    int MT, NT;
    int W, Wx, nb;          // work items, items per XCD, workgroups per XCD
};

template <int OCC>
__global__ void __launch_bounds__(256, OCC) wino_wgrad_gemm_kernel(const WinoWgradKP p) {
    constexpr int BN = 128, WTM = 64, WTN = 64, TM = 2, TN = 2;
    constexpr int ASZ = BK * LDW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As = smem;             // [k = tile][128 ci], row pitch LDW
    float* const Bs = smem + ASZ;       // [k = tile][128 co]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int wend = min(p.W, (xcd + 1) * p.Wx);
    int wc = xcd * p.Wx + slot;
    if (wc >= wend) return;
    const unsigned a_step = (unsigned)BK * (unsigned)p.Cin * 4u, b_step = (unsigned)BK * (unsigned)p.Cout * 4u;
    const int krow = tid >> 5, kcol = (tid & 31) * 4;

    __amdgpu_buffer_rsrc_t rsA, rsB;
    unsigned a_off, b_off;              // row krow of the current k-step; rows +8, +16, +24 via the immediate
    auto item_ksteps = [&](int w) {
        const int split = (w / (p.MT * p.NT)) % p.S;
        const int t0 = split * p.t_per_split;
        const int t1 = min(p.T, t0 + p.t_per_split);
        return t1 > t0 ? (t1 - t0 + BK - 1) / BK : 0;
    };
    auto setup_load = [&](int w) {
        const int nt = w % p.NT;
        int t = w / p.NT;
        const int mt = t % p.MT; t /= p.MT;
        const int split = t % p.S;
        const int xi = t / p.S;
        const int t0 = split * p.t_per_split;
        const int t1 = min(p.T, t0 + p.t_per_split);
        // descriptors end at the split's last tile: rows past it read as zeros
        rsA = make_rsrc(p.V + (size_t)xi * p.T * p.Cin, (unsigned)t1 * (unsigned)p.Cin * 4u);
        rsB = make_rsrc(p.Y + (size_t)xi * p.T * p.Cout, (unsigned)t1 * (unsigned)p.Cout * 4u);
        a_off = ((unsigned)(t0 + krow) * (unsigned)p.Cin + (unsigned)(mt * BM + kcol)) * 4u;
        b_off = ((unsigned)(t0 + krow) * (unsigned)p.Cout + (unsigned)(nt * BN + kcol)) * 4u;
    };
    float4 ra[4], rb[4];
    auto issue_loads = [&]() {
        const unsigned a8 = 8u * (unsigned)p.Cin * 4u, b8 = 8u * (unsigned)p.Cout * 4u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = bload4(rsA, a_off + i * a8);
            rb[i] = bload4(rsB, b_off + i * b8);
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4*>(&As[(krow + 8 * i) * LDW + kcol]) = ra[i];
            *reinterpret_cast<float4*>(&Bs[(krow + 8 * i) * BN + kcol]) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    setup_load(wc);
    issue_loads();
    store_tiles();
    __syncthreads();
    for (;;) {
        const int KS = item_ksteps(wc);     // >= 1 by construction of the host-side split
        for (int ks = 0; ks < KS; ++ks) {
            bool more = true;
            if (ks + 1 < KS) { a_off += a_step; b_off += b_step; }
            else {
                more = wc + p.nb < wend;
                if (more) setup_load(wc + p.nb);
            }
            if (more) issue_loads();
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                float a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = As[(2 * kk + h) * LDW + wm * WTM + i * 32 + l31];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = Bs[(2 * kk + h) * BN + wn * WTN + j * 32 + l31];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            if (ks == KS - 1) {
                const int nt = wc % p.NT;
                int t = wc / p.NT;
                const int mt = t % p.MT; t /= p.MT;     // t = xi * S + split: the slab index
                const unsigned sl_bytes = (unsigned)p.Cin * (unsigned)p.Cout * 4u;
                const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(
                    p.slab + (size_t)t * p.Cin * p.Cout, 0, sl_bytes, 0x00020000);
                const unsigned n4 = (unsigned)p.Cout * 4u;
                unsigned vbase = (unsigned)(mt * BM + wm * WTM + 4 * h) * n4 + (unsigned)(nt * BN + wn * WTN + l31) * 4u;
                asm volatile("" : "+v"(vbase));
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][j][r]), rsC, vbase + j * 128u,
                                                                  (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2)) * n4, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            }
            __syncthreads();
            if (more) store_tiles();
            __syncthreads();
        }
        wc += p.nb;
        if (wc >= wend) break;
    }
}

// ---------------------------------------------------------------------------
// Winograd F(2x2, 3x3) for the stride-1 3x3 convolutions (fp32).  y = A^T[(G g G^T) . (B^T d B)]A:
// 16 multiplications per 2x2 output tile instead of 36, i.e. 2.25x fewer MFMA flops than the
// direct implicit GEMM.  Three kernels around the batched GEMM above:
//   wino_weights  : U[xi][K][N]   = G g G^T                  (once per weight update)
//   wino_input    : V[xi][tile][C] = B^T d B, d = 4x4 patch   (reflect / zero padding in the gather)
//   (16 GEMMs)    : M[xi][tile][N] = V[xi] . U[xi]
//   wino_output   : y tile = A^T M A (+bias, activation)
// ---------------------------------------------------------------------------
template <int BF>
__global__ void wino_weights_kernel(const float* __restrict__ w, void* __restrict__ Uv, int Cin,
                                    int Cout, int flip_transpose) {
    // w: [3][3][Cin][Cout].  fp32 U: [16][K][N] with (K,N) = (Cin,Cout), or (Cout,Cin) with the taps
    // flipped when flip_transpose (the dgrad filter).  bf16 U: [16][N][K] (contraction index
    // contiguous, what the bf16 MFMA operand reads want): [Cout][Cin], or [Cin][Cout] flipped.
    const int total = Cin * Cout;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    // thread -> (ci, co) with the output's fastest index fastest (coalesced plane stores): fp32 U is
    // [K][N] = [Cin][Cout] (or [Cout][Cin] flipped), bf16 U the transpose of that
    const bool co_fast = (flip_transpose != 0) == (BF != 0);
    const int ci = co_fast ? i / Cout : i % Cin, co = co_fast ? i - (i / Cout) * Cout : i / Cin;
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int ka = flip_transpose ? 2 - a : a, kb = flip_transpose ? 2 - b : b;
            g[a][b] = w[((size_t)(ka * 3 + kb) * Cin + ci) * Cout + co];
        }
    float t[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
    }
    const size_t plane = (size_t)Cin * Cout;
    const size_t o = (flip_transpose != 0) != (BF != 0) ? (size_t)co * Cin + ci : (size_t)ci * Cout + co;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float u4[4] = {t[a][0], 0.5f * (t[a][0] + t[a][1] + t[a][2]), 0.5f * (t[a][0] - t[a][1] + t[a][2]),
                             t[a][2]};
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (BF == 2) static_cast<_Float16*>(Uv)[(size_t)(a * 4 + b) * plane + o] = (_Float16)u4[b];
            else if (BF) static_cast<__bf16*>(Uv)[(size_t)(a * 4 + b) * plane + o] = (__bf16)u4[b];
            else static_cast<float*>(Uv)[(size_t)(a * 4 + b) * plane + o] = u4[b];
        }
    }
}

__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// V[xi][tile][C]: one thread per (tile, 4 channels).  Padding 1, reflect or zero.
template <int BF>
__global__ void wino_input_kernel(const float* __restrict__ x, void* __restrict__ V, int B, int H,
                                  int W, int C4, int reflect) {
    const int TH = H / 2, TW = W / 2;
    const long long tiles = (long long)B * TH * TW;
    const long long total = tiles * C4;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C4);
    const long long tile = i / C4;
    const int tx = (int)(tile % TW);
    const int ty = (int)((tile / TW) % TH);
    const int b = (int)(tile / ((long long)TW * TH));
    float4 d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int hh = 2 * ty - 1 + r;
        bool okh = true;
        if (reflect) { hh = hh < 0 ? -hh : hh; hh = hh >= H ? 2 * (H - 1) - hh : hh; }
        else okh = hh >= 0 && hh < H;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int ww = 2 * tx - 1 + q;
            bool ok = okh;
            if (reflect) { ww = ww < 0 ? -ww : ww; ww = ww >= W ? 2 * (W - 1) - ww : ww; }
            else ok = ok && ww >= 0 && ww < W;
            d[r][q] = ok ? reinterpret_cast<const float4*>(x)[(((long long)b * H + hh) * W + ww) * C4 + c]
                         : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float4 t[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        t[0][q] = f4sub(d[0][q], d[2][q]);
        t[1][q] = f4add(d[1][q], d[2][q]);
        t[2][q] = f4sub(d[2][q], d[1][q]);
        t[3][q] = f4sub(d[1][q], d[3][q]);
    }
    const long long plane = tiles * C4;
    const long long o = tile * C4 + c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        wst4<BF>(V, (long long)(r * 4 + 0) * plane + o, f4sub(t[r][0], t[r][2]));
        wst4<BF>(V, (long long)(r * 4 + 1) * plane + o, f4add(t[r][1], t[r][2]));
        wst4<BF>(V, (long long)(r * 4 + 2) * plane + o, f4sub(t[r][2], t[r][1]));
        wst4<BF>(V, (long long)(r * 4 + 3) * plane + o, f4sub(t[r][1], t[r][3]));
    }
}

// y[b, 2ty+i, 2tx+j, :] = (A^T M A)[i][j] (+ bias, act).  One thread per (tile, 4 channels).
template <int BF>
__global__ void wino_output_kernel(const void* __restrict__ M, float* __restrict__ y,
                                   const float* __restrict__ bias, int B, int H, int W, int C4,
                                   int act) {
    const int TH = H / 2, TW = W / 2;
    const long long tiles = (long long)B * TH * TW;
    const long long total = tiles * C4;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C4);
    const long long tile = i / C4;
    const int tx = (int)(tile % TW);
    const int ty = (int)((tile / TW) % TH);
    const int b = (int)(tile / ((long long)TW * TH));
    const long long plane = tiles * C4;
    float4 m[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) m[r][q] = wld4<BF>(M, (long long)(r * 4 + q) * plane + tile * C4 + c);
    float4 s[2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        s[0][q] = f4add(f4add(m[0][q], m[1][q]), m[2][q]);
        s[1][q] = f4sub(f4sub(m[1][q], m[2][q]), m[3][q]);
    }
    float4 bv = bias ? reinterpret_cast<const float4*>(bias)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        float4 y0 = f4add(f4add(f4add(s[r][0], s[r][1]), s[r][2]), bv);
        float4 y1 = f4add(f4sub(f4sub(s[r][1], s[r][2]), s[r][3]), bv);
        y0 = make_float4(apply_act(y0.x, act), apply_act(y0.y, act), apply_act(y0.z, act), apply_act(y0.w, act));
        y1 = make_float4(apply_act(y1.x, act), apply_act(y1.y, act), apply_act(y1.z, act), apply_act(y1.w, act));
        float4* o = reinterpret_cast<float4*>(y) + (((long long)b * H + 2 * ty + r) * W + 2 * tx) * C4 + c;
        o[0] = y0;
        o[C4] = y1;
    }
}

// ---------------------------------------------------------------------------
// bf16-MFMA variant of the fprop / dgrad kernel (mmh_conv_desc.dtype = MMH_BF16).
// Activations stay fp32 in HBM and are rounded to bf16 (RNE) while being staged into LDS;
// weights come pre-rounded to bf16 and laid out with the contraction index contiguous
// (mmh_prep_weights_bf16: [tap][Cout][Cin] for fprop, [tap][Cin][Cout] for dgrad), so both LDS
// tiles are [row][k] and both MFMA operands are one ds_read_b128 per 32x32x16 step.
// Accumulation and output are fp32.  k-step = 64 channels of one tap; needs channels % 64 == 0
// (the 3x3 stacks; the small-Cin stems keep the fp32 kernel).
// ---------------------------------------------------------------------------
constexpr int BK16 = 64;     // contraction depth per k-step (bf16 kernel)
constexpr int LDH = 72;      // LDS row pitch in bf16 elements (144 B: conflict-free ds_read_b128)


template <int BN, int WAVES_M, int WAVES_N, bool H16>
__device__ __forceinline__ void conv_igemm_bf16_body_t(const ConvKP& p, const int bx, const int by,
                                                       const int gx, const int gy) {
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int NBL = BN / 32;            // 16-byte weight loads per thread per k-step
    constexpr int ASZ = BM * LDH;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* const As = reinterpret_cast<__bf16*>(smem);
    __bf16* const Bs = As + ASZ;

    const Gather& g = p.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    int mt = by, nt = bx;
    if (p.xcd_remap) {
        const int band = gy / 8;
        const int L = by * gx + bx;
        if (L < band * 8 * gx) {
            const int x = L & 7, i = L >> 3;
            nt = i % gx;
            mt = x * band + i / gx;
        }
    }
    const int m0 = mt * BM, n0 = nt * BN;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(g.src, g.src_bytes);
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(p.w, p.w_bytes);

    // A: 128 rows x 16 groups of 4 channels; thread = (group tid&15, rows (tid>>4)+16i, i<8)
    const int grp = tid & 15;
    unsigned a_img[8];
    int a_bh[8], a_bw[8];
    bool a_ok[8];
    const int PHW = g.PH * g.PW;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int m = m0 + (tid >> 4) + 16 * i;
        a_ok[i] = m < p.M;
        int mm = a_ok[i] ? m : 0;
        int b = mm / PHW;
        int r = mm - b * PHW;
        int ph = r / g.PW;
        int pw = r - ph * g.PW;
        a_img[i] = (unsigned)b * (unsigned)(g.srcH * g.srcW);
        a_bh[i] = ph * g.ap_h + g.a0_h;
        a_bw[i] = pw * g.ap_w + g.a0_w;
    }
    // B: BN rows x 8 groups of 8 bf16 (16 B); thread = (group tid&7, rows (tid>>3)+32i)
    const int bgrp = tid & 7;
    unsigned b_row[NBL];
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
        const int n = n0 + (tid >> 3) + 32 * i;
        b_row[i] = n < p.N ? (unsigned)n * (unsigned)p.wKper * 2u : OOB;   // bytes
    }

    // k order: (super-chunk of cw chunks, tap, chunk); chunk = 64 channels
    int th = 0, tw = 0, cc = 0, j = 0;       // block-uniform
    unsigned a_off[8];
    float4 ra[8];
    uint4 rb[NBL];
    // flat order (small Cin): this thread's 4-channel group walks (tap, c4) 16 groups per k-step
    KState kf;
    int kstep = 0;
    if (p.flat16) {
        kf.j = 0;
        const int tap = grp / g.C4;
        kf.c4 = grp - tap * g.C4;
        kf.th = tap / g.TW;
        kf.tw = tap - kf.th * g.TW;
    }

    auto load_tiles = [&]() {
        if (p.flat16) {
            const bool kv = kf.th < g.TH;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                ra[i] = bload4(rsA, gather_off(g, a_img[i], a_bh[i], a_bw[i], kf, kv && a_ok[i]));
            const unsigned koff = (unsigned)(kstep * 64 + bgrp * 8) * 2u;
#pragma unroll
            for (int i = 0; i < NBL; ++i) {
                u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsW, b_row[i] != OOB ? koff + b_row[i] : OOB, 0, 0);
                rb[i] = __builtin_bit_cast(uint4, r);
            }
            return;
        }
        if (j == 0) {
            KState s; s.th = th; s.tw = tw; s.c4 = 0; s.j = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) a_off[i] = gather_base(g, a_img[i], a_bh[i], a_bw[i], s, a_ok[i]);
        }
        if (p.src16) {      // 16-bit source: half the byte offsets, 8 bytes per group of 4 channels
            const unsigned coff = (unsigned)(cc * 16 + grp) * 8u;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rsA, a_off[i] != OOB ? (a_off[i] >> 1) + coff : OOB, 0, 0);
                ra[i].x = __uint_as_float(r[0]);
                ra[i].y = __uint_as_float(r[1]);
            }
        } else {
            const unsigned coff = (unsigned)(cc * 16 + grp) * 16u;      // bytes: 4 fp32 channels per group
#pragma unroll
            for (int i = 0; i < 8; ++i) ra[i] = bload4(rsA, a_off[i] != OOB ? a_off[i] + coff : OOB);
        }
        const int tap_true = (p.kh0 + p.tstep * th) * p.KW_true + p.kw0 + p.tstep * tw;
        const unsigned tap_off = ((unsigned)tap_true * (unsigned)p.wRows * (unsigned)p.wKper +
                                  (unsigned)(cc * 64 + bgrp * 8)) * 2u;
#pragma unroll
        for (int i = 0; i < NBL; ++i) {
            u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsW, b_row[i] != OOB ? tap_off + b_row[i] : OOB, 0, 0);
            rb[i] = __builtin_bit_cast(uint4, r);
        }
    };
    auto advance = [&]() {
        if (p.flat16) {
            ++kstep;
            kf.c4 += 16;
            while (kf.c4 >= g.C4) { kf.c4 -= g.C4; if (++kf.tw == g.TW) { kf.tw = 0; ++kf.th; } }
            return;
        }
        ++cc;
        if (++j == g.cw) {
            j = 0;
            cc -= g.cw;
            if (++tw == g.TW) { tw = 0; if (++th == g.TH) { th = 0; cc += g.cw; } }
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (p.src16)
                *reinterpret_cast<uint2*>(&As[((tid >> 4) + 16 * i) * LDH + grp * 4]) =
                    make_uint2(__float_as_uint(ra[i].x), __float_as_uint(ra[i].y));
            else
                *reinterpret_cast<bf16x4*>(&As[((tid >> 4) + 16 * i) * LDH + grp * 4]) = to_lp4<H16>(ra[i]);
        }
#pragma unroll
        for (int i = 0; i < NBL; ++i)
            *reinterpret_cast<uint4*>(&Bs[((tid >> 3) + 32 * i) * LDH + bgrp * 8]) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;

    if (p.nk > 0) {
        load_tiles();
        store_tiles();
    }
    __syncthreads();
    for (int ks = 0; ks < p.nk; ++ks) {
        const bool more = ks + 1 < p.nk;
        if (more) {
            advance();
            if (!(p.dbg & 1)) load_tiles();
        }
#pragma unroll
        for (int s16 = 0; s16 < BK16 / 16; ++s16) {
            bf16x8 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(&As[(wm * WTM + i * 32 + l31) * LDH + s16 * 16 + h * 8]);
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
                bfr[jn] = *reinterpret_cast<const bf16x8*>(&Bs[(wn * WTN + jn * 32 + l31) * LDH + s16 * 16 + h * 8]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn)
                    acc[i][jn] = mfma16<H16>(af[i], bfr[jn], acc[i][jn]);
        }
        if (!(p.dbg & 2)) {
            __syncthreads();
            if (more) store_tiles();
            __syncthreads();
        }
    }

    float bv[TN];
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
        const int n = n0 + wn * WTN + jn * 32 + l31;
        bv[jn] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m >= p.M) continue;
            unsigned off;
            if (p.out_linear) off = (unsigned)m * p.out_cs;
            else {
                int b = m / PHW;
                int rr = m - b * PHW;
                int ph = rr / g.PW;
                int pw = rr - ph * g.PW;
                off = ((unsigned)(b * p.OH + ph * p.o_p + p.o0_h) * (unsigned)p.OW +
                       (unsigned)(pw * p.o_p + p.o0_w)) * p.out_cs;
            }
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
                const int n = n0 + wn * WTN + jn * 32 + l31;
                if (n < p.N) {
                    float v = acc[i][jn][r] + bv[jn];
                    if (p.accum) v += p.out[off + n];
                    p.out[off + n] = apply_act(v, p.act);
                }
            }
        }
    }
}

template <int BN, int WAVES_M, int WAVES_N>
__device__ __forceinline__ void conv_igemm_bf16_body(const ConvKP& p, const int bx, const int by,
                                                     const int gx, const int gy) {
    if (p.h16) conv_igemm_bf16_body_t<BN, WAVES_M, WAVES_N, true>(p, bx, by, gx, gy);
    else conv_igemm_bf16_body_t<BN, WAVES_M, WAVES_N, false>(p, bx, by, gx, gy);
}
template <int BN, int WAVES_M, int WAVES_N>
__global__ void __launch_bounds__(256, 2) conv_igemm_bf16_kernel(const ConvKP p) {
    conv_igemm_bf16_body<BN, WAVES_M, WAVES_N>(p, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y);
}
template <int BN, int WAVES_M, int WAVES_N>
__global__ void __launch_bounds__(256, 2) conv_igemm_bf16_multi_kernel(const MultiKP mp) {
    const int k = multi_piece(mp, blockIdx.x);
    const ConvKP& p = mp.p[k];
    const int gx = (p.N + BN - 1) / BN, gy = (p.M + BM - 1) / BM;
    const int local = blockIdx.x - mp.start[k];
    conv_igemm_bf16_body<BN, WAVES_M, WAVES_N>(p, local % gx, local / gx, gx, gy);
}

// dx += the border terms produced by the multi-piece dgrad launch.  One thread per (target
// pixel, 4 channels); every target sums its terms in a fixed order (deterministic).
// scratch: rows [B][2][W][C], cols [B][H][2][C], corners [B][4][C].
template <int LP>
__global__ void border_add_kernel(void* __restrict__ dx, const float* __restrict__ rows,
                                  const float* __restrict__ cols, const float* __restrict__ corners,
                                  int B, int H, int W, int C4) {
    const int nr = (H - 2 == 1) ? 1 : 2, nc = (W - 2 == 1) ? 1 : 2;
    const int per_img = nr * W + (H - nr) * nc;
    const int64_t total = (int64_t)B * per_img * C4;
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < total; t += stride) {
        const int c = (int)(t % C4);
        int64_t u = t / C4;
        const int e = (int)(u % per_img);
        const int b = (int)(u / per_img);
        int i, j;
        if (e < nr * W) { i = (e / W == 0) ? 1 : H - 2; j = e % W; }
        else {
            const int f = e - nr * W;
            int ii = f / nc;                       // index among rows that are not 1 / H-2
            i = ii >= 1 ? ii + 1 : ii;             // skip row 1
            if (nr == 2 && i >= H - 2) i += 1;     // skip row H-2
            j = (f % nc == 0) ? 1 : W - 2;
        }
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        auto add = [&](const float* src, int64_t idx) {
            const float4 v = reinterpret_cast<const float4*>(src)[idx];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        };
        if (i == 1) add(rows, (((int64_t)b * 2 + 0) * W + j) * C4 + c);
        if (i == H - 2) add(rows, (((int64_t)b * 2 + 1) * W + j) * C4 + c);
        if (j == 1) add(cols, (((int64_t)b * H + i) * 2 + 0) * C4 + c);
        if (j == W - 2) add(cols, (((int64_t)b * H + i) * 2 + 1) * C4 + c);
        if (i == 1 && j == 1) add(corners, ((int64_t)b * 4 + 0) * C4 + c);
        if (i == 1 && j == W - 2) add(corners, ((int64_t)b * 4 + 1) * C4 + c);
        if (i == H - 2 && j == 1) add(corners, ((int64_t)b * 4 + 2) * C4 + c);
        if (i == H - 2 && j == W - 2) add(corners, ((int64_t)b * 4 + 3) * C4 + c);
        const int64_t o = (((int64_t)b * H + i) * W + j) * C4 + c;
        float4 v = wld4<LP>(dx, o);
        v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w;
        wst4<LP>(dx, o, v);
    }
}

// ---------------------------------------------------------------------------
// wgrad kernel: slab[z][(tap,ci)][co] = sum over this split's pixels.
// Rows of the GEMM are the flat (tap, ci) index, the contraction runs over
// pixels; the x tile is gathered exactly as in fprop and read transposed.
// ---------------------------------------------------------------------------
struct WgradKP {
    Gather g;               // gathers x (fprop geometry); PH,PW = dy's domain
    const float* dy;
    unsigned dy_bytes;
    unsigned dy_cs;
    float* slab;            // [splits][Kflat][N]
    int Mrows;              // Kflat = TH*TW*C4*4
    int N;                  // Cout
    int P;                  // total pixels = batch*PH*PW
    int pix_per_split;      // multiple of 32
    int dbg;                // timing-only ablation bits (results wrong)
    int nsplit;             // splits per batch: blockIdx.z = batch * nsplit + split
    long long src_bs, dy_bs;   // element strides between batches (Winograd: 16 planes)
    int x16, dy16;             // 16-bit kernels: the gathered tensor / dy is already 16-bit in HBM
    int xcd_remap;          // grid size % 8 == 0: XCD x works on a contiguous range of (z, y, x) ids
    int h16;                    // 16-bit kernel: 0 = bf16, 1 = fp16 operands
};

// Workgroups are dealt round-robin over the 8 XCDs.  With the remap, XCD x takes the contiguous
// logical range [x*T/8, (x+1)*T/8) in order, so the column/row tiles of one (plane, split) -
// which read the same two panels - are resident on ONE XCD and share its L2.
__device__ __forceinline__ void wgrad_block_ids(const WgradKP& p, int& bx, int& by, int& bz) {
    bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
    if (p.xcd_remap) {
        const int gx = gridDim.x, gy = gridDim.y;
        const int total = gx * gy * (int)gridDim.z;
        const int L = bx + gx * (by + gy * bz);
        const int Lg = (L & 7) * (total >> 3) + (L >> 3);
        bx = Lg % gx;
        const int t = Lg / gx;
        by = t % gy;
        bz = t / gy;
    }
}

template <int BN, int WAVES_M, int WAVES_N, bool DBUF>
__global__ void __launch_bounds__(256, 2) conv_wgrad_kernel(const WgradKP p) {
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int NB = BN / 32;
    constexpr int NBUF = DBUF ? 2 : 1;
    constexpr int ASZ = BK * LDW;   // [pixel][128 rows]
    constexpr int BSZ = BK * BN;    // [pixel][BN]
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As_base = smem;
    float* const Bs_base = smem + NBUF * ASZ;

    const Gather& g = p.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    int bx, by, bz;
    wgrad_block_ids(p, bx, by, bz);
    const int m0 = by * BM;
    const int n0 = bx * BN;
    const int batch = bz / p.nsplit, split = bz - batch * p.nsplit;
    const int pbeg = split * p.pix_per_split;
    const int pend = min(p.P, pbeg + p.pix_per_split);
    const int PHW = g.PH * g.PW;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(g.src + batch * p.src_bs, g.src_bytes);
    const __amdgpu_buffer_rsrc_t rsD = make_rsrc(p.dy + batch * p.dy_bs, p.dy_bytes);

    // this thread's fixed (tap, channel group): flat group index
    KState kt;
    const int fg = (m0 >> 2) + (tid & 31);
    const bool g_ok = fg * 4 < p.Mrows;
    {
        int tap = fg / g.C4;
        kt.c4 = fg - tap * g.C4;
        kt.th = tap / g.TW;
        kt.tw = tap - kt.th * g.TW;
    }
    // this thread's 4 pixels of the current k-step, advanced by 32 pixels per step without
    // divisions: (image, row, col) counters.  The row part of the source offset (image base +
    // reflected/checked source row) only changes when the pixel row does, so it is cached.
    int px_b[4], px_h[4], px_w[4];
    unsigned hpart[4];     // (image*srcH + source row) * srcW, or OOB when the row is out of range
    auto row_part = [&](int b, int ph) -> unsigned {
        int vh = ph * g.ap_h + g.a0_h + kt.th * g.at_h;
        bool ok = true;
        if (g.reflect) {
            vh = vh < 0 ? -vh : vh;
            vh = vh >= g.srcH ? 2 * (g.srcH - 1) - vh : vh;
        } else {
            ok = vh >= 0 && vh < g.srcH;
        }
        return ok ? ((unsigned)b * (unsigned)g.srcH + (unsigned)vh) * (unsigned)g.srcW : OOB;
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pix = pbeg + (tid >> 5) + 8 * i;
        px_b[i] = pix / PHW;
        const int r = pix - px_b[i] * PHW;
        px_h[i] = r / g.PW;
        px_w[i] = r - px_h[i] * g.PW;
        hpart[i] = row_part(px_b[i], px_h[i]);
    }
    // dy tile columns
    unsigned d_col[NB];
    int d_prow[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int idx = tid + 256 * i;
        d_prow[i] = idx / (BN / 4);
        const int n = n0 + 4 * (idx - d_prow[i] * (BN / 4));
        d_col[i] = n < p.N ? (unsigned)n * 4u : OOB;
    }

    float4 ra[4];
    float4 rb[NB];
    auto load_tiles = [&](int pbase) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pix = pbase + (tid >> 5) + 8 * i;
            int vw = px_w[i] * g.ap_w + g.a0_w + kt.tw * g.at_w;
            bool ok = g_ok && pix < pend && hpart[i] != OOB;
            if (g.reflect) {
                vw = vw < 0 ? -vw : vw;
                vw = vw >= g.srcW ? 2 * (g.srcW - 1) - vw : vw;
            } else {
                ok = ok && vw >= 0 && vw < g.srcW;
            }
            const unsigned off = ((hpart[i] + (unsigned)vw) * g.src_cs + (unsigned)kt.c4 * 4u) * 4u;
            ra[i] = bload4(rsA, ok ? off : OOB);
            // advance 32 pixels
            px_w[i] += BK;
            if (px_w[i] >= g.PW) {
                do {
                    px_w[i] -= g.PW;
                    if (++px_h[i] == g.PH) { px_h[i] = 0; ++px_b[i]; }
                } while (px_w[i] >= g.PW);
                hpart[i] = row_part(px_b[i], px_h[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int pix = pbase + d_prow[i];
            const unsigned off = (unsigned)pix * p.dy_cs * 4u + d_col[i];
            rb[i] = bload4(rsD, (pix < pend && d_col[i] != OOB) ? off : OOB);
        }
    };
    auto store_tiles = [&](int buf) {
        float* As = As_base + buf * ASZ;
        float* Bs = Bs_base + buf * BSZ;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<float4*>(&As[((tid >> 5) + 8 * i) * LDW + (tid & 31) * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<float4*>(&Bs[(tid + 256 * i) * 4]) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (pbeg < pend) {
        load_tiles(pbeg);
        store_tiles(0);
        __syncthreads();
        int cur = 0;
        for (int pb = pbeg; pb < pend; pb += BK) {
            const float* As = As_base + cur * ASZ;
            const float* Bs = Bs_base + cur * BSZ;
            if (pb + BK < pend && !(p.dbg & 1)) load_tiles(pb + BK);
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                float a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = As[(2 * kk + h) * LDW + wm * WTM + i * 32 + l31];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = Bs[(2 * kk + h) * BN + wn * WTN + j * 32 + l31];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            if (!(p.dbg & 2)) {
                if (DBUF) cur ^= 1; else __syncthreads();
                if (pb + BK < pend) store_tiles(cur);
                __syncthreads();
            }
        }
    }

    float* slab = p.slab + (size_t)bz * (size_t)p.Mrows * p.N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m >= p.Mrows) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 32 + l31;
                if (n < p.N) slab[(size_t)m * p.N + n] = acc[i][j][r];
            }
        }
}

// ---------------------------------------------------------------------------
// bf16-MFMA wgrad.  Same GEMM as conv_wgrad_kernel (rows = flat (tap,ci), cols = co,
// contraction over pixels) but x and dy are rounded to bf16 while staged and both MFMA operands
// are read TRANSPOSED from LDS with ds_read_b64_tr_b16: the tiles sit as [pixel][channel] (the
// order they arrive in, channel contiguous) and the 32x32x16 operands need 8 consecutive pixels
// per lane.  Row pitch 320 B puts the 4 rows of a transposed 4x16 block on disjoint banks.
// ---------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int BKP = 64;      // pixels per k-step
constexpr int LDT = 160;     // LDS row pitch in bf16 elements (128 + 32 pad = 320 B)

__device__ __forceinline__ bf16x8 tr_frag(const __bf16* tile, int row0, int col0, int lane) {
    // operand fragment for lane: 8 consecutive rows (pixels) row0 + 8h + 0..7 of column
    // col0 + (lane & 31), from a row-major [pixel][channel] bf16 tile.
    const int h = lane >> 5, G1 = (lane >> 4) & 1, q = (lane & 15) >> 2, p2 = lane & 3;
    const __bf16* a = tile + (row0 + 8 * h + q) * LDT + col0 + 16 * G1 + 4 * p2;
    typedef s16x4 __attribute__((address_space(3))) * lds_p;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a + 4 * LDT));
    struct { s16x4 a, b; } both = {lo, hi};
    return __builtin_bit_cast(bf16x8, both);
}

template <int BN, int WAVES_M, int WAVES_N, bool H16, bool IO16>     // IO16: x and dy are 16-bit in HBM
__device__ __forceinline__ void conv_wgrad_bf16_body(const WgradKP& p) {
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int NBD = BN / 16;            // dy float4 loads per thread per k-step
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* const As = reinterpret_cast<__bf16*>(smem);     // [BKP][LDT]
    __bf16* const Bs = As + BKP * LDT;                      // [BKP][LDT]

    const Gather& g = p.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    int bx, by, bz;
    wgrad_block_ids(p, bx, by, bz);
    const int m0 = by * BM;
    const int n0 = bx * BN;
    const int batch = bz / p.nsplit, split = bz - batch * p.nsplit;
    const int pbeg = split * p.pix_per_split;
    const int pend = min(p.P, pbeg + p.pix_per_split);
    const int PHW = g.PH * g.PW;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(g.src + batch * p.src_bs, g.src_bytes);
    const __amdgpu_buffer_rsrc_t rsD = make_rsrc(p.dy + batch * p.dy_bs, p.dy_bytes);

    KState kt;
    const int fg = (m0 >> 2) + (tid & 31);
    const bool g_ok = fg * 4 < p.Mrows;
    {
        int tap = fg / g.C4;
        kt.c4 = fg - tap * g.C4;
        kt.th = tap / g.TW;
        kt.tw = tap - kt.th * g.TW;
        kt.j = 0;
    }
    // 8 pixels per thread per k-step: rows (tid>>5) + 8i
    int px_b[8], px_h[8], px_w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int pix = pbeg + (tid >> 5) + 8 * i;
        px_b[i] = pix / PHW;
        const int r = pix - px_b[i] * PHW;
        px_h[i] = r / g.PW;
        px_w[i] = r - px_h[i] * g.PW;
    }
    unsigned d_col[NBD];
    int d_prow[NBD];
#pragma unroll
    for (int i = 0; i < NBD; ++i) {
        const int idx = tid + 256 * i;
        d_prow[i] = idx / (BN / 4);
        const int n = n0 + 4 * (idx - d_prow[i] * (BN / 4));
        d_col[i] = n < p.N ? (unsigned)n * 4u : OOB;
    }

    float4 ra[8];
    float4 rb[NBD];
    auto load_tiles = [&](int pbase) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pix = pbase + (tid >> 5) + 8 * i;
            const unsigned goff = gather_off(g, (unsigned)px_b[i] * (unsigned)(g.srcH * g.srcW),
                                             px_h[i] * g.ap_h + g.a0_h, px_w[i] * g.ap_w + g.a0_w, kt,
                                             g_ok && pix < pend);
            if (IO16) {        // 16-bit source: half the byte offset, 8 bytes per group of 4 channels
                const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rsA, goff != OOB ? goff >> 1 : OOB, 0, 0);
                ra[i].x = __uint_as_float(r[0]);
                ra[i].y = __uint_as_float(r[1]);
            } else {
                ra[i] = bload4(rsA, goff);
            }
            px_w[i] += BKP;
            while (px_w[i] >= g.PW) {
                px_w[i] -= g.PW;
                if (++px_h[i] == g.PH) { px_h[i] = 0; ++px_b[i]; }
            }
        }
#pragma unroll
        for (int i = 0; i < NBD; ++i) {
            const int pix = pbase + d_prow[i];
            const unsigned off = (unsigned)pix * p.dy_cs * 4u + d_col[i];
            const bool ok = pix < pend && d_col[i] != OOB;
            if (IO16) {
                const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rsD, ok ? off >> 1 : OOB, 0, 0);
                rb[i].x = __uint_as_float(r[0]);
                rb[i].y = __uint_as_float(r[1]);
            } else {
                rb[i] = bload4(rsD, ok ? off : OOB);
            }
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __bf16* dst = &As[((tid >> 5) + 8 * i) * LDT + (tid & 31) * 4];
            if (IO16) *reinterpret_cast<uint2*>(dst) = make_uint2(__float_as_uint(ra[i].x), __float_as_uint(ra[i].y));
            else *reinterpret_cast<bf16x4*>(dst) = to_lp4<H16>(ra[i]);
        }
#pragma unroll
        for (int i = 0; i < NBD; ++i) {
            const int idx = tid + 256 * i;
            const int prow = idx / (BN / 4), c4 = idx - prow * (BN / 4);
            __bf16* dst = &Bs[prow * LDT + c4 * 4];
            if (IO16) *reinterpret_cast<uint2*>(dst) = make_uint2(__float_as_uint(rb[i].x), __float_as_uint(rb[i].y));
            else *reinterpret_cast<bf16x4*>(dst) = to_lp4<H16>(rb[i]);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (pbeg < pend) {
        load_tiles(pbeg);
        store_tiles();
        __syncthreads();
        for (int pb = pbeg; pb < pend; pb += BKP) {
            const bool more = pb + BKP < pend;
            if (more) load_tiles(pb + BKP);
#pragma unroll
            for (int s16 = 0; s16 < BKP / 16; ++s16) {
                bf16x8 af[TM], bfr[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = tr_frag(As, s16 * 16, wm * WTM + i * 32, lane);
#pragma unroll
                for (int j = 0; j < TN; ++j) bfr[j] = tr_frag(Bs, s16 * 16, wn * WTN + j * 32, lane);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = mfma16<H16>(af[i], bfr[j], acc[i][j]);
            }
            __syncthreads();
            if (more) store_tiles();
            __syncthreads();
        }
    }

    float* slab = p.slab + (size_t)bz * (size_t)p.Mrows * p.N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m >= p.Mrows) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 32 + l31;
                if (n < p.N) slab[(size_t)m * p.N + n] = acc[i][j][r];
            }
        }
}

template <int BN, int WAVES_M, int WAVES_N, bool IO16>
__global__ void __launch_bounds__(256, 2) conv_wgrad_bf16_kernel(const WgradKP p) {
    if (p.h16) conv_wgrad_bf16_body<BN, WAVES_M, WAVES_N, true, IO16>(p);
    else conv_wgrad_bf16_body<BN, WAVES_M, WAVES_N, false, IO16>(p);
}

// ---------------------------------------------------------------------------
// bf16 Winograd-domain GEMMs (F(2x2,3x3) under --opt_level O1/O2): V, U, M, Yhat are bf16 in HBM,
// products accumulate in fp32 on v_mfma_f32_32x32x16_bf16.
//
// wino_gemm_bf16_kernel: P planes of C[M x N] = A[M x K] . B[N x K]^T, both operands with the
// contraction index contiguous (V [P][tiles][K], U [P][N][K]), so both MFMA operands are one
// ds_read_b128 per lane from [row][LDH] LDS tiles.  Same persistent / XCD-aware structure as
// wino_gemm_kernel; k-step = 64.  The fp32 accumulators leave as bf16: each lane swaps with its
// neighbour (DPP quad_perm) so that even lanes store the (n, n+1) pair of one row and odd lanes
// the pair of the next row - 4-byte stores, 64 contiguous bytes per row and half-wave.
// Requires K % 64 == 0 and N % 32 == 0.
// ---------------------------------------------------------------------------
struct WinoGemmBfKP {
    const __bf16* A;
    const __bf16* B;
    __bf16* C;
    int M, K, N, P;
    int MT, NT, W, Wx, nb;
    int h16;
};

template <bool H16>
__device__ __forceinline__ unsigned pack_lp2(float lo, float hi) {
    if (H16) {
        f16x2 v;
        v[0] = (_Float16)lo; v[1] = (_Float16)hi;
        return __builtin_bit_cast(unsigned, v);
    }
    bf16x2 v;
    v[0] = (__bf16)lo; v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}

template <int BKS, bool H16>
__global__ void __launch_bounds__(256, 2) wino_gemm_bf16_kernel(const WinoGemmBfKP p) {
    constexpr int BN = 128, WTM = 64, WTN = 64, TM = 2, TN = 2;
    constexpr int LDB = BKS + 8;          // row pitch (bf16): 144 B / 272 B, conflict-free ds_read_b128
    constexpr int CPR = BKS / 8;          // 16-byte chunks per row
    constexpr int NLD = BM * CPR / 256;   // loads per thread and operand per k-step
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* const As = reinterpret_cast<__bf16*>(smem);     // [128][LDB]
    __bf16* const Bs = As + BM * LDB;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int chunk = tid % CPR, lrow = tid / CPR;
    constexpr int RPP = 256 / CPR;        // rows covered per pass
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int wend = min(p.W, (xcd + 1) * p.Wx);
    int wc = xcd * p.Wx + slot;
    if (wc >= wend) return;
    const int KS = p.K / BKS;
    const unsigned a_bytes = (unsigned)p.M * (unsigned)p.K * 2u;
    const unsigned b_bytes = (unsigned)p.N * (unsigned)p.K * 2u;
    const unsigned c_bytes = (unsigned)p.M * (unsigned)p.N * 2u;

    __amdgpu_buffer_rsrc_t rsA, rsB;
    unsigned a_off[NLD], b_off[NLD];
    auto setup_load = [&](int w) {
        const int nt = w % p.NT;
        const int t = w / p.NT;
        const int mt = t % p.MT;
        const int xi = t / p.MT;
        rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.A) + (size_t)xi * p.M * p.K, 0, a_bytes, 0x00020000);
        rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.B) + (size_t)xi * p.N * p.K, 0, b_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {   // rows past M / N start beyond num_records: the loads return zeros
            a_off[i] = ((unsigned)(mt * BM + lrow + RPP * i) * (unsigned)p.K + chunk * 8u) * 2u;
            b_off[i] = ((unsigned)(nt * BN + lrow + RPP * i) * (unsigned)p.K + chunk * 8u) * 2u;
        }
    };
    uint4 ra[NLD], rb[NLD];
    auto issue_loads = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            ra[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, a_off[i], 0, 0));
            rb[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsB, b_off[i], 0, 0));
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            *reinterpret_cast<uint4*>(&As[(lrow + RPP * i) * LDB + chunk * 8]) = ra[i];
            *reinterpret_cast<uint4*>(&Bs[(lrow + RPP * i) * LDB + chunk * 8]) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    setup_load(wc);
    issue_loads();
    store_tiles();
    __syncthreads();
    for (;;) {
        for (int ks = 0; ks < KS; ++ks) {
            bool more = true;
            if (ks + 1 < KS) {
#pragma unroll
                for (int i = 0; i < NLD; ++i) { a_off[i] += BKS * 2u; b_off[i] += BKS * 2u; }
            } else {
                more = wc + p.nb < wend;
                if (more) setup_load(wc + p.nb);
            }
            if (more) issue_loads();
#pragma unroll
            for (int s16 = 0; s16 < BKS / 16; ++s16) {
                bf16x8 af[TM], bfr[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[i] = *reinterpret_cast<const bf16x8*>(&As[(wm * WTM + i * 32 + l31) * LDB + s16 * 16 + h * 8]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bfr[j] = *reinterpret_cast<const bf16x8*>(&Bs[(wn * WTN + j * 32 + l31) * LDB + s16 * 16 + h * 8]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = mfma16<H16>(af[i], bfr[j], acc[i][j]);
            }
            if (ks == KS - 1) {
                const int nt = wc % p.NT;
                const int t = wc / p.NT;
                const int mt = t % p.MT;
                const int xi = t / p.MT;
                const __amdgpu_buffer_rsrc_t rsC =
                    __builtin_amdgcn_make_buffer_rsrc(p.C + (size_t)xi * p.M * p.N, 0, c_bytes, 0x00020000);
                const unsigned n2 = (unsigned)p.N * 2u;     // bytes per row
                const int odd = l31 & 1;
                const int ncol0 = nt * BN + wn * WTN;
                // even lanes store (n, n+1) of row m, odd lanes (n-1, n) of row m+1; rows past M fall
                // beyond num_records (the row term stays in the range-checked vector offset)
                unsigned vbase = (unsigned)(mt * BM + wm * WTM + 4 * h + odd) * n2 + (unsigned)(ncol0 + (l31 & ~1)) * 2u;
                asm volatile("" : "+v"(vbase));
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (ncol0 + j * 32 < p.N) {
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int r = 0; r < 16; r += 2) {
                                const float x0 = acc[i][j][r], x1 = acc[i][j][r + 1];
                                const float y0 = __builtin_bit_cast(
                                    float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x0), 0xB1, 0xF, 0xF, true));
                                const float y1 = __builtin_bit_cast(
                                    float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x1), 0xB1, 0xF, 0xF, true));
                                const unsigned v = odd ? pack_lp2<H16>(y1, x1) : pack_lp2<H16>(x0, y0);
                                const unsigned rd = (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2)) * n2;
                                __builtin_amdgcn_raw_buffer_store_b32(v, rsC, vbase + rd + j * 64u, 0, 0);
                            }
                    }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            }
            __syncthreads();
            if (more) store_tiles();
            __syncthreads();
        }
        wc += p.nb;
        if (wc >= wend) break;
    }
}

// wino_wgrad_gemm_bf16_kernel: slab[xi][split][Cin x Cout] (fp32) = V[xi][t0:t1]^T . Yh[xi][t0:t1], bf16
// operands with the contraction index (tile) as the SLOW dimension: tiles sit in LDS as
// [tile][channel] and both MFMA operands are read transposed (ds_read_b64_tr_b16, tr_frag).
// Persistent / XCD-aware like wino_wgrad_gemm_kernel; k-step = 64 tiles.  Cin, Cout % 128 == 0.
struct WinoWgradBfKP {
    const __bf16* V;
    const __bf16* Y;
    float* slab;
    int T, Cin, Cout, P, S, t_per_split, MT, NT, W, Wx, nb;
    int h16;
};

template <bool H16>
__global__ void __launch_bounds__(256, 2) wino_wgrad_gemm_bf16_kernel(const WinoWgradBfKP p) {
    constexpr int BN = 128, WTM = 64, WTN = 64, TM = 2, TN = 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* const As = reinterpret_cast<__bf16*>(smem);     // [BKP tiles][LDT]
    __bf16* const Bs = As + BKP * LDT;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int wend = min(p.W, (xcd + 1) * p.Wx);
    int wc = xcd * p.Wx + slot;
    if (wc >= wend) return;
    const int krow = tid >> 4, kchunk = tid & 15;       // 16 tile rows x 16 chunks of 8 channels per pass
    const unsigned a_step = (unsigned)BKP * (unsigned)p.Cin * 2u, b_step = (unsigned)BKP * (unsigned)p.Cout * 2u;

    __amdgpu_buffer_rsrc_t rsA, rsB;
    unsigned a_off, b_off;
    auto item_ksteps = [&](int w) {
        const int split = (w / (p.MT * p.NT)) % p.S;
        const int t0 = split * p.t_per_split;
        const int t1 = min(p.T, t0 + p.t_per_split);
        return t1 > t0 ? (t1 - t0 + BKP - 1) / BKP : 0;
    };
    auto setup_load = [&](int w) {
        const int nt = w % p.NT;
        int t = w / p.NT;
        const int mt = t % p.MT; t /= p.MT;
        const int split = t % p.S;
        const int xi = t / p.S;
        const int t0 = split * p.t_per_split;
        const int t1 = min(p.T, t0 + p.t_per_split);
        rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.V) + (size_t)xi * p.T * p.Cin, 0,
                                                (unsigned)t1 * (unsigned)p.Cin * 2u, 0x00020000);
        rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.Y) + (size_t)xi * p.T * p.Cout, 0,
                                                (unsigned)t1 * (unsigned)p.Cout * 2u, 0x00020000);
        a_off = ((unsigned)(t0 + krow) * (unsigned)p.Cin + (unsigned)(mt * BM + kchunk * 8)) * 2u;
        b_off = ((unsigned)(t0 + krow) * (unsigned)p.Cout + (unsigned)(nt * BN + kchunk * 8)) * 2u;
    };
    uint4 ra[4], rb[4];
    auto issue_loads = [&]() {
        const unsigned a16 = 16u * (unsigned)p.Cin * 2u, b16 = 16u * (unsigned)p.Cout * 2u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, a_off + i * a16, 0, 0));
            rb[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsB, b_off + i * b16, 0, 0));
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<uint4*>(&As[(krow + 16 * i) * LDT + kchunk * 8]) = ra[i];
            *reinterpret_cast<uint4*>(&Bs[(krow + 16 * i) * LDT + kchunk * 8]) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    setup_load(wc);
    issue_loads();
    store_tiles();
    __syncthreads();
    for (;;) {
        const int KS = item_ksteps(wc);
        for (int ks = 0; ks < KS; ++ks) {
            bool more = true;
            if (ks + 1 < KS) { a_off += a_step; b_off += b_step; }
            else {
                more = wc + p.nb < wend;
                if (more) setup_load(wc + p.nb);
            }
            if (more) issue_loads();
#pragma unroll
            for (int s16 = 0; s16 < BKP / 16; ++s16) {
                bf16x8 af[TM], bfr[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = tr_frag(As, s16 * 16, wm * WTM + i * 32, lane);
#pragma unroll
                for (int j = 0; j < TN; ++j) bfr[j] = tr_frag(Bs, s16 * 16, wn * WTN + j * 32, lane);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = mfma16<H16>(af[i], bfr[j], acc[i][j]);
            }
            if (ks == KS - 1) {
                const int nt = wc % p.NT;
                int t = wc / p.NT;
                const int mt = t % p.MT; t /= p.MT;
                const unsigned sl_bytes = (unsigned)p.Cin * (unsigned)p.Cout * 4u;
                const __amdgpu_buffer_rsrc_t rsC =
                    __builtin_amdgcn_make_buffer_rsrc(p.slab + (size_t)t * p.Cin * p.Cout, 0, sl_bytes, 0x00020000);
                const unsigned n4 = (unsigned)p.Cout * 4u;
                unsigned vbase = (unsigned)(mt * BM + wm * WTM + 4 * h) * n4 + (unsigned)(nt * BN + wn * WTN + l31) * 4u;
                asm volatile("" : "+v"(vbase));
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][j][r]), rsC, vbase + j * 128u,
                                                                  (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2)) * n4, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            }
            __syncthreads();
            if (more) store_tiles();
            __syncthreads();
        }
        wc += p.nb;
        if (wc >= wend) break;
    }
}

// Transpose of ReflectionPad2d: every real pixel sums the padded positions that mirror onto it.
__global__ void reflect_fold_kernel(const float* __restrict__ dxp, float* __restrict__ dx, int B,
                                    int H, int W, int C4, int p) {
    const int64_t total = (int64_t)B * H * W * C4;
    const int Hp = H + 2 * p, Wp = W + 2 * p;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        int c = (int)(i % C4);
        int64_t t = i / C4;
        int w = (int)(t % W); t /= W;
        int hh = (int)(t % H);
        int b = (int)(t / H);
        // padded coordinates that reflect to hh: hh itself, -hh (1<=hh<=p), 2(H-1)-hh (H-1-p<=hh<=H-2)
        int hs[3], ws[3], nh = 0, nw = 0;
        hs[nh++] = hh;
        if (hh >= 1 && hh <= p) hs[nh++] = -hh;
        if (hh >= H - 1 - p && hh <= H - 2) hs[nh++] = 2 * (H - 1) - hh;
        ws[nw++] = w;
        if (w >= 1 && w <= p) ws[nw++] = -w;
        if (w >= W - 1 - p && w <= W - 2) ws[nw++] = 2 * (W - 1) - w;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int a = 0; a < nh; ++a)
            for (int e = 0; e < nw; ++e) {
                const float4 v = reinterpret_cast<const float4*>(
                    dxp)[(((int64_t)b * Hp + hs[a] + p) * Wp + ws[e] + p) * C4 + c];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        reinterpret_cast<float4*>(dx)[i] = s;
    }
}

// --------------------------------------------------------------------------- host side
inline bool is16(int dtype) { return dtype == MMH_BF16 || dtype == MMH_FP16; }

int validate(const mmh_conv_desc* d) {
    MMH_REQUIRE(d != nullptr, "conv desc is NULL");
    MMH_REQUIRE(d->dtype == MMH_F32 || is16(d->dtype), "bad dtype=%d", d->dtype);
    MMH_REQUIRE(d->Cin % 4 == 0 && d->Cout % 4 == 0, "Cin/Cout must be multiples of 4 (%d,%d)",
                d->Cin, d->Cout);
    MMH_REQUIRE(d->x_cs % 4 == 0 && d->y_cs % 4 == 0 && d->x_cs >= d->Cin && d->y_cs >= d->Cout,
                "bad channel strides x_cs=%d y_cs=%d", d->x_cs, d->y_cs);
    MMH_REQUIRE(d->stride == 1 || d->stride == 2, "stride must be 1 or 2 (%d)", d->stride);
    MMH_REQUIRE(d->kh >= 1 && d->kw >= 1 && d->pad >= 0, "bad kernel/pad");
    MMH_REQUIRE(d->pad_mode == MMH_PAD_ZERO || (d->stride == 1 && d->pad < d->H && d->pad < d->W),
                "reflect padding needs stride 1 and pad < H,W");
    MMH_REQUIRE(d->Ho == (d->H + 2 * d->pad - d->kh) / d->stride + 1 &&
                    d->Wo == (d->W + 2 * d->pad - d->kw) / d->stride + 1,
                "Ho/Wo inconsistent with H,W,k,s,p");
    MMH_REQUIRE((int64_t)d->B * (d->H + 2 * d->pad) * (d->W + 2 * d->pad) * (int64_t)d->x_cs < (1ll << 30) &&
                    (int64_t)d->B * d->Ho * d->Wo * (int64_t)d->y_cs < (1ll << 30) &&
                    (int64_t)d->kh * d->kw * d->Cin * d->Cout < (1ll << 30),
                "tensor too large for 32-bit byte offsets (4 GiB per buffer)");
    return 0;
}

int g_conv_cw = 0;   // tuning knob: chunks per tap visit (0 = auto)

void set_korder(Gather& g, int& nk, int& Kflat) {
    Kflat = g.TH * g.TW * g.C4 * 4;
    if (g.C4 % 8 == 0) {
        g.chunk_major = 1;
        nk = (g.C4 / 8) * g.TH * g.TW;
        const int chunks = g.C4 / 8;
        g.cw = (g_conv_cw > 0 && chunks % g_conv_cw == 0) ? g_conv_cw
               : (chunks % 2 == 0 ? 2 : 1);   // 2: +2.7 % speed over 1 at equal HBM traffic; 4: +1.3 % more
                                              // speed but +23..36 % fabric reads (working set > 4 MiB L2)
    } else {
        g.chunk_major = 0; g.cw = 1; nk = (Kflat + BK - 1) / BK;
        g.dc4 = 8 % g.C4; g.dtw = (8 / g.C4) % g.TW; g.dth = (8 / g.C4) / g.TW;
    }
}

// Dynamic LDS above 64 KiB must be opted into once per kernel.
template <typename K>
int allow_lds(K kernel, size_t bytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return mmh::fail("hipFuncSetAttribute(LDS=%zu): %s", bytes, hipGetErrorString(e));
    return 0;
}

int g_conv_dbuf = 0;   // tuning knob (mmh_set_option "conv_dbuf")
int g_conv_dbg = 0;    // ablation bits, timing only
int g_conv_xcd = 1;    // XCD-aware tile mapping on/off

template <int BN, int WM, int WN, bool NMAJOR, bool DBUF>
int launch_conv_t(const ConvKP& p, hipStream_t st) {
    constexpr size_t lds = (DBUF ? 2 : 1) * (BM * LDA + (NMAJOR ? BN * LDA : BK * BN)) * sizeof(float) + tab_bytes(BM);
    static int ready = -1;
    if (ready != 0) ready = allow_lds(conv_igemm_kernel<BN, WM, WN, NMAJOR, DBUF>, lds);
    if (ready != 0) return ready;
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM);
    hipLaunchKernelGGL((conv_igemm_kernel<BN, WM, WN, NMAJOR, DBUF>), grid, dim3(256), lds, st, p);
    return mmh::check_launch("conv_igemm_kernel");
}

template <int BN, int WM, int WN, bool NMAJOR>
int launch_conv_tall_t(const ConvKP& p, hipStream_t st) {
    constexpr size_t lds = (256 * LDA + (NMAJOR ? BN * LDA : BK * BN)) * sizeof(float) + tab_bytes(256);
    static int ready = -1;
    if (ready != 0) ready = allow_lds(conv_igemm_tall_kernel<BN, WM, WN, NMAJOR>, lds);
    if (ready != 0) return ready;
    dim3 grid((p.N + BN - 1) / BN, (p.M + 255) / 256);
    hipLaunchKernelGGL((conv_igemm_tall_kernel<BN, WM, WN, NMAJOR>), grid, dim3(256), lds, st, p);
    return mmh::check_launch("conv_igemm_tall_kernel");
}

template <int BN, int WM, int WN>
int launch_conv_bf16_t(const ConvKP& p, hipStream_t st) {
    constexpr size_t lds = (size_t)(BM + BN) * LDH * 2;
    static int ready = -1;
    if (ready != 0) ready = allow_lds(conv_igemm_bf16_kernel<BN, WM, WN>, lds);
    if (ready != 0) return ready;
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM);
    hipLaunchKernelGGL((conv_igemm_bf16_kernel<BN, WM, WN>), grid, dim3(256), lds, st, p);
    return mmh::check_launch("conv_igemm_bf16_kernel");
}

// bf16 MFMA path: needs the contraction channels per tap to be a multiple of 64
bool bf16_ok(const ConvKP& p) { return p.g.C4 % 16 == 0 && p.N % 4 == 0; }

int launch_conv_bf16(ConvKP& p, hipStream_t st) {
    Gather& g = p.g;
    if (p.flat16) {
        g.chunk_major = 0;
        g.cw = 1;
        p.nk = (g.TH * g.TW * g.C4 + 15) / 16;      // 64 flat k per step
    } else {
        const int chunks = g.C4 / 16;               // 64-channel chunks per tap
        g.chunk_major = 1;
        g.cw = chunks % 4 == 0 ? 4 : (chunks % 2 == 0 ? 2 : 1);
        p.nk = chunks * g.TH * g.TW;
    }
    p.dbg = g_conv_dbg;
    {
        const int BNsel = p.N > 64 ? 128 : (p.N > 32 ? 64 : 32);
        const int gx = (p.N + BNsel - 1) / BNsel, gy = (p.M + BM - 1) / BM;
        p.xcd_remap = (g_conv_xcd && gx > 1 && gy >= 8) ? 1 : 0;
    }
    if (p.N > 64) return launch_conv_bf16_t<128, 2, 2>(p, st);
    if (p.N > 32) return launch_conv_bf16_t<64, 2, 2>(p, st);
    return launch_conv_bf16_t<32, 4, 1>(p, st);
}

int g_conv_tall = 1;    // 256x64 block tile for the stride-2 dgrad with N <= 64 (+6 %); 2: also plain fprop/dgrad
int g_conv_xcd1 = 1;    // XCD-banded row tiles also when there is a single column tile (halo rows meet in one L2)
int g_conv_bn256 = 1;   // 128x256 block tile (wave tile 64x128) when N % 256 == 0: +2.4 % on fprop
int g_conv_levels = 1;  // 2: fprop (N > 64, chunk-major k order) on conv_igemm_levels2_kernel - two-level summation

int launch_conv_levels2(const ConvKP& p, hipStream_t st) {
    constexpr size_t lds = (BM * LDA + BK * 128) * sizeof(float) + tab_bytes(BM);
    static int ready = -1;
    if (ready != 0) ready = allow_lds(conv_igemm_levels2_kernel, lds);
    if (ready != 0) return ready;
    dim3 grid((p.N + 127) / 128, (p.M + BM - 1) / BM);
    hipLaunchKernelGGL(conv_igemm_levels2_kernel, grid, dim3(256), lds, st, p);
    return mmh::check_launch("conv_igemm_levels2_kernel");
}

template <bool NMAJOR>
int launch_conv(const ConvKP& p, hipStream_t st) {
    const_cast<ConvKP&>(p).dbg = g_conv_dbg;
    if (g_conv_levels == 2 && !NMAJOR && p.N > 64 && p.g.chunk_major) {
        const int gx = (p.N + 127) / 128, gy = (p.M + BM - 1) / BM;
        const_cast<ConvKP&>(p).xcd_remap = (g_conv_xcd && gx > 1 && gy >= 8) ? 1 : 0;
        return launch_conv_levels2(p, st);
    }
    if (g_conv_bn256 && !NMAJOR && p.N % 256 == 0) {
        const int gx = p.N / 256, gy = (p.M + BM - 1) / BM;
        const_cast<ConvKP&>(p).xcd_remap = (g_conv_xcd && gx > 1 && gy >= 8) ? 1 : 0;
        return launch_conv_t<256, 2, 2, NMAJOR, false>(p, st);
    }
    if (g_conv_tall == 2 && !p.stats && p.N > 32 && p.N <= 64 && p.M >= 256 * 512) {   // measured: no gain on the stems (A/B only)
        const_cast<ConvKP&>(p).xcd_remap = (g_conv_xcd1 && (p.M + 255) / 256 >= 8) ? 1 : 0;
        return launch_conv_tall_t<64, 4, 1, NMAJOR>(p, st);
    }
    {
        const int BNsel = p.N > 64 ? 128 : (p.N > 32 ? 64 : 32);
        const int gx = (p.N + BNsel - 1) / BNsel, gy = (p.M + BM - 1) / BM;
        const_cast<ConvKP&>(p).xcd_remap = (g_conv_xcd && (gx > 1 || g_conv_xcd1) && gy >= 8) ? 1 : 0;
    }
    if (p.N > 64)
        return g_conv_dbuf ? launch_conv_t<128, 2, 2, NMAJOR, true>(p, st)
                           : launch_conv_t<128, 2, 2, NMAJOR, false>(p, st);
    if (p.N > 32) return launch_conv_t<64, 2, 2, NMAJOR, false>(p, st);
    return launch_conv_t<32, 4, 1, NMAJOR, false>(p, st);
}

// forward-orientation gather of x (used by fprop and wgrad)
Gather fwd_gather(const mmh_conv_desc* d, const void* x) {
    Gather g{};
    g.src = static_cast<const float*>(x);
    g.src_bytes = (unsigned)((size_t)d->B * d->H * d->W * d->x_cs * sizeof(float));
    g.srcH = d->H; g.srcW = d->W; g.src_cs = (unsigned)d->x_cs;
    g.PH = d->Ho; g.PW = d->Wo;
    g.TH = d->kh; g.TW = d->kw;
    g.C4 = d->Cin / 4;
    g.ap_h = d->stride; g.at_h = 1; g.a0_h = -d->pad;
    g.ap_w = d->stride; g.at_w = 1; g.a0_w = -d->pad;
    g.shift = 0;
    g.reflect = d->pad_mode == MMH_PAD_REFLECT;
    return g;
}

// partial rows the fp32 direct fprop writes when asked for output statistics: (M / 128) row tiles x wave rows of the
// kernel launch_conv picks for this N; 0 = not available (ragged row tiles, 16-bit, tall tiles)
int fprop_stats_chunks(const mmh_conv_desc* d) {
    if (mmh::stem_f32_ok(d)) return mmh::stem_f32_stats_chunks(d);
    const long long M = (long long)d->B * d->Ho * d->Wo;
    if (is16(d->dtype) || M % BM != 0 || d->Cout % 4 != 0) return 0;
    const int waves_m = d->Cout > 32 ? 2 : 4;
    return (int)(M / BM) * waves_m;
}

int do_fprop(const mmh_conv_desc* d, const void* x, const void* w, const void* bias, void* y,
             int act, hipStream_t st, float* stats = nullptr) {
    if (mmh::stem_f32_ok(d) && (!stats || mmh::stem_f32_stats_chunks(d) > 0))
        return mmh::launch_stem_f32(d, x, w, bias, y, act, stats, st);
    ConvKP p{};
    p.stats = stats;
    p.g = fwd_gather(d, x);
    set_korder(p.g, p.nk, p.Kflat);
    p.w = static_cast<const float*>(w);
    p.w_bytes = (unsigned)((size_t)d->kh * d->kw * d->Cin * d->Cout * sizeof(float));
    p.out = static_cast<float*>(y);
    p.bias = static_cast<const float*>(bias);
    p.M = d->B * d->Ho * d->Wo;
    p.N = d->Cout;
    p.wCin = d->Cin; p.wCout = d->Cout;
    p.KW_true = d->kw; p.kh0 = 0; p.kw0 = 0; p.tstep = 1;
    p.out_linear = 1; p.out_cs = (unsigned)d->y_cs;
    p.OH = d->Ho; p.OW = d->Wo; p.o_p = 1;
    p.act = act;
    if (is16(d->dtype)) {
        p.h16 = d->dtype == MMH_FP16;
        if (d->Cin % 64 == 0) {
            // w is the prepared bf16 tensor w_t [taps][Cout][Cin]
            p.wRows = d->Cout; p.wKper = d->Cin;
            p.w_bytes = (unsigned)((size_t)d->kh * d->kw * d->Cin * d->Cout * 2);
        } else {
            // small Cin: w is w_flat [Cout][Kpad], Kpad = ceil(taps*Cin/64)*64, flat k = (tap, ci)
            const int Kpad = (d->kh * d->kw * d->Cin + 63) / 64 * 64;
            p.flat16 = 1;
            p.wRows = d->Cout; p.wKper = Kpad;
            p.w_bytes = (unsigned)((size_t)d->Cout * Kpad * 2);
        }
        return launch_conv_bf16(p, st);
    }
    return launch_conv<false>(p, st);
}

// Gradient w.r.t. the conv input.  out: [B, OH, OW, Cin] with channel stride out_cs, where
// (OH,OW) is the padded domain for reflect mode and (H,W) otherwise.
int g_dgrad_s2_multi = 1;   // stride-2 dgrad / ConvTranspose fprop: 4 parity classes in one multi-piece launch
template <int BN, int WM, int WN, int BMT = 128>
int launch_multi_t(MultiKP& mp, bool bf16, hipStream_t st);

int do_dgrad(const mmh_conv_desc* d, const void* dy, const void* w, const void* bias, void* dx,
             int dx_cs, int act, hipStream_t st) {
    const int s = d->stride;
    const int off = d->pad_mode == MMH_PAD_REFLECT ? d->pad : 0;  // padded-domain origin shift
    const int OH = d->H + 2 * off, OW = d->W + 2 * off;
    MMH_REQUIRE(s == 1 || (OH % 2 == 0 && OW % 2 == 0), "stride-2 dgrad needs even H,W");
    if (mmh::dgrad_s2_halo_ok(d, dx_cs, act)) return mmh::launch_dgrad_s2_halo(d, dy, w, bias, dx, dx_cs, act, st);
    ConvKP classes[4];
    int ncls = 0;
    // one piece per output parity class (1 class for stride 1, 4 for stride 2)
    for (int ch = 0; ch < s; ++ch)
        for (int cw = 0; cw < s; ++cw) {
            // real coordinate hi = ph*s + ch - off; taps with (hi + pad - kh) % s == 0
            const int kh0 = ((ch - off + d->pad) % s + s) % s;
            const int kw0 = ((cw - off + d->pad) % s + s) % s;
            const int TH = kh0 < d->kh ? (d->kh - kh0 + s - 1) / s : 0;
            const int TW = kw0 < d->kw ? (d->kw - kw0 + s - 1) / s : 0;
            ConvKP p{};
            Gather& g = p.g;
            g.src = static_cast<const float*>(dy);
            g.src_bytes = (unsigned)((size_t)d->B * d->Ho * d->Wo * d->y_cs * sizeof(float));
            g.srcH = d->Ho; g.srcW = d->Wo; g.src_cs = (unsigned)d->y_cs;
            g.PH = OH / s; g.PW = OW / s;
            g.TH = TH; g.TW = TW;
            g.C4 = d->Cout / 4;
            // v = hi + pad - kh = ph*s + (ch - off + pad - kh0) - s*th
            g.ap_h = s; g.at_h = -s; g.a0_h = ch - off + d->pad - kh0;
            g.ap_w = s; g.at_w = -s; g.a0_w = cw - off + d->pad - kw0;
            g.shift = s == 2 ? 1 : 0;
            g.reflect = 0;
            set_korder(g, p.nk, p.Kflat);
            p.w = static_cast<const float*>(w);
            p.w_bytes = (unsigned)((size_t)d->kh * d->kw * d->Cin * d->Cout * sizeof(float));
            p.out = static_cast<float*>(dx);
            p.bias = static_cast<const float*>(bias);
            p.M = d->B * g.PH * g.PW;
            p.N = d->Cin;
            p.wCin = d->Cin; p.wCout = d->Cout;
            p.KW_true = d->kw; p.kh0 = kh0; p.kw0 = kw0; p.tstep = s;
            p.OH = OH; p.OW = OW; p.o_p = s; p.o0_h = ch; p.o0_w = cw;
            p.out_cs = (unsigned)dx_cs;
            p.out_linear = s == 1;
            p.act = act;
            if (TH == 0 || TW == 0) p.nk = 0;  // no tap reaches this class: writes zeros
            if (s == 2 && !is16(d->dtype) && g_dgrad_s2_multi) {   // collected, launched once below
                classes[ncls++] = p;
                continue;
            }
            int rc;
            if (is16(d->dtype)) {
                // w is the prepared 16-bit tensor [taps][Cin][Cout]
                p.h16 = d->dtype == MMH_FP16;
                p.wRows = d->Cin; p.wKper = d->Cout;
                p.w_bytes = (unsigned)((size_t)d->kh * d->kw * d->Cin * d->Cout * 2);
                MMH_REQUIRE(bf16_ok(p), "bf16 dgrad needs Cout %% 64 == 0 (Cout=%d)", d->Cout);
                const bool empty = (TH == 0 || TW == 0);
                rc = launch_conv_bf16(p, st);
                (void)empty;
            } else
                rc = launch_conv<true>(p, st);
            if (rc) return rc;
        }
    if (ncls) {
        // The four parity classes of a stride-2 dgrad have 1, 2, 2 and 4 taps: short contractions
        // whose separate launches each pay a pipeline fill and a tail.  One multi-piece launch,
        // heaviest class first, keeps the chip full across class boundaries.
        MultiKP mp{};
        const int order[4] = {3, 1, 2, 0};      // (ch,cw) = (1,1) has the most taps for k3 / pad 1
        for (int i = 0; i < ncls; ++i) mp.p[mp.n++] = classes[ncls == 4 ? order[i] : i];
        const int C = d->Cin;
        if (C > 64) return launch_multi_t<128, 2, 2>(mp, false, st);
        if (C > 32) {
            if (g_conv_tall && mp.p[0].M >= 256 * 128) return launch_multi_t<64, 4, 1, 256>(mp, false, st);
            return launch_multi_t<64, 2, 2>(mp, false, st);
        }
        return launch_multi_t<32, 4, 1>(mp, false, st);
    }
    return 0;
}

// ---------------------------------------------------------------------------
// Reflect-pad(1) dgrad, folded, without the padded domain.  With R = ReflectionPad2d(1),
// dx = R^T g where g is the full correlation of dy with the flipped taps on the padded domain.
// g restricted to the real HxW domain is an ordinary zero-padded dgrad (the main launch, tile
// aligned, full speed).  The pad ring folds onto rows/cols 1 and H-2/W-2 only, and each ring
// element is reached by ONE row (or column) of taps:
//   dx[1, j]   += sum_kw dy[0,   j+1-kw] w[0][kw]      dx[H-2, j] += sum_kw dy[H-1, j+1-kw] w[2][kw]
//   dx[i, 1]   += sum_kh dy[i+1-kh, 0]   w[kh][0]      dx[i, W-2] += sum_kh dy[i+1-kh, W-1] w[kh][2]
//   dx[1,1] += dy[0,0] w[0][0], ... (4 corners)
// i.e. 8 small GEMMs (~2 % of the main one) that accumulate into dx; they run as ONE multi-piece
// launch after the main kernel.  Versus the padded-domain + fold route this saves the 6.3 % ring
// rows, the misaligned 66-pixel rows and the fold pass.
// ---------------------------------------------------------------------------
template <int BN, int WM, int WN, int BMT>
int launch_multi_t(MultiKP& mp, bool bf16, hipStream_t st) {
    int total = 0;
    for (int i = 0; i < mp.n; ++i) {
        mp.start[i] = total;
        total += ((mp.p[i].N + BN - 1) / BN) * ((mp.p[i].M + BMT - 1) / BMT);
    }
    mp.start[mp.n] = total;
    if (BMT != BM) {
        constexpr size_t lds = (BMT * LDA + BN * LDA) * sizeof(float) + tab_bytes(BMT);
        static int ready = -1;
        if (ready != 0) ready = allow_lds(conv_igemm_multi_kernel<BN, WM, WN, true, BMT>, lds);
        if (ready != 0) return ready;
        hipLaunchKernelGGL((conv_igemm_multi_kernel<BN, WM, WN, true, BMT>), dim3(total), dim3(256), lds, st, mp);
    } else if (bf16) {
        constexpr size_t lds = (size_t)(BM + BN) * LDH * 2;
        static int ready = -1;
        if (ready != 0) ready = allow_lds(conv_igemm_bf16_multi_kernel<BN, WM, WN>, lds);
        if (ready != 0) return ready;
        hipLaunchKernelGGL((conv_igemm_bf16_multi_kernel<BN, WM, WN>), dim3(total), dim3(256), lds, st, mp);
    } else {
        constexpr size_t lds = (BM * LDA + BN * LDA) * sizeof(float) + tab_bytes(BM);
        static int ready = -1;
        if (ready != 0) ready = allow_lds(conv_igemm_multi_kernel<BN, WM, WN, true>, lds);
        if (ready != 0) return ready;
        hipLaunchKernelGGL((conv_igemm_multi_kernel<BN, WM, WN, true>), dim3(total), dim3(256), lds, st, mp);
    }
    return mmh::check_launch("conv_igemm_multi_kernel");
}

size_t reflect1_ws_bytes(const mmh_conv_desc* d) {
    return (size_t)d->B * (2 * d->W + 2 * d->H + 4) * d->Cin * sizeof(float);
}

int g_border_bn64 = 1;  // border-only dgrad launches (Winograd path): 64-wide tiles
// phase: bit 0 = the GEMMs (border pieces into ws, main term into dx), bit 1 = border_add (ws -> dx)
int do_dgrad_reflect1(const mmh_conv_desc* d, const void* dy, const void* w, void* dx, void* ws,
                      hipStream_t st, bool with_main = true, int phase = 3, bool dy16 = false, bool dx16 = false) {
    const int H = d->H, W = d->W, C = d->Cin;
    const bool bf16 = is16(d->dtype);
    float* rows = static_cast<float*>(ws);                       // [B][2][W][C]
    float* cols = rows + (size_t)d->B * 2 * W * C;               // [B][H][2][C]
    float* corners = cols + (size_t)d->B * H * 2 * C;            // [B][4][C]
    auto piece = [&](float* out, int OH, int OW, int PH, int PW, int o0h, int o0w, int TH, int TW,
                     int kh0, int kw0, int aph, int ath, int a0h, int apw, int atw, int a0w) {
        ConvKP p{};
        Gather& g = p.g;
        g.src = static_cast<const float*>(dy);
        g.src_bytes = (unsigned)((size_t)d->B * d->Ho * d->Wo * d->y_cs * (dy16 ? 2 : sizeof(float)));
        p.src16 = dy16 ? 1 : 0;
        g.srcH = d->Ho; g.srcW = d->Wo; g.src_cs = (unsigned)d->y_cs;
        g.PH = PH; g.PW = PW; g.TH = TH; g.TW = TW;
        g.C4 = d->Cout / 4;
        g.ap_h = aph; g.at_h = ath; g.a0_h = a0h;
        g.ap_w = apw; g.at_w = atw; g.a0_w = a0w;
        g.shift = 0; g.reflect = 0;
        if (bf16) {
            const int chunks = g.C4 / 16;
            g.chunk_major = 1;
            g.cw = chunks % 4 == 0 ? 4 : (chunks % 2 == 0 ? 2 : 1);
            p.nk = chunks * TH * TW;
            p.h16 = d->dtype == MMH_FP16;
            p.wRows = d->Cin; p.wKper = d->Cout;
            p.w_bytes = (unsigned)((size_t)d->kh * d->kw * d->Cin * d->Cout * 2);
        } else {
            set_korder(g, p.nk, p.Kflat);
            p.w_bytes = (unsigned)((size_t)d->kh * d->kw * d->Cin * d->Cout * sizeof(float));
        }
        p.w = static_cast<const float*>(w);
        p.out = out;
        p.M = d->B * PH * PW;
        p.N = C;
        p.wCin = d->Cin; p.wCout = d->Cout;
        p.KW_true = d->kw; p.kh0 = kh0; p.kw0 = kw0; p.tstep = 1;
        p.OH = OH; p.OW = OW; p.o_p = 1; p.o0_h = o0h; p.o0_w = o0w;
        p.out_cs = (unsigned)C;
        p.out_linear = (PH == OH && PW == OW) ? 1 : 0;
        p.act = MMH_ACT_NONE;
        return p;
    };
    MultiKP mp{};
    int n = 0;
    // Border pieces first (few workgroups, long serial k loops): they overlap with the main tiles.
    // ring rows -1 / H : taps kh = 0 / 2, source rows 0 / H-1  -> rows[b][0|1][j]
    mp.p[n++] = piece(rows, 2, W, 1, W, 0, 0, 1, 3, 0, 0, 0, 0, 0, 1, -1, 1);
    mp.p[n++] = piece(rows, 2, W, 1, W, 1, 0, 1, 3, 2, 0, 0, 0, H - 1, 1, -1, 1);
    // ring cols -1 / W : taps kw = 0 / 2, source cols 0 / W-1  -> cols[b][i][0|1]
    mp.p[n++] = piece(cols, H, 2, H, 1, 0, 0, 3, 1, 0, 0, 1, -1, 1, 0, 0, 0);
    mp.p[n++] = piece(cols, H, 2, H, 1, 0, 1, 3, 1, 0, 2, 1, -1, 1, 0, 0, W - 1);
    // ring corners -> corners[b][0..3]
    mp.p[n++] = piece(corners, 1, 4, 1, 1, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0);
    mp.p[n++] = piece(corners, 1, 4, 1, 1, 0, 1, 1, 1, 0, 2, 0, 0, 0, 0, 0, W - 1);
    mp.p[n++] = piece(corners, 1, 4, 1, 1, 0, 2, 1, 1, 2, 0, 0, 0, H - 1, 0, 0, 0);
    mp.p[n++] = piece(corners, 1, 4, 1, 1, 0, 3, 1, 1, 2, 2, 0, 0, H - 1, 0, 0, W - 1);
    // main: g on the real domain: source = (i + 1 - kh, j + 1 - kw), zero outside
    ConvKP main = piece(static_cast<float*>(dx), H, W, H, W, 0, 0, 3, 3, 0, 0, 1, -1, 1, 1, -1, 1);
    {
        const int BNsel = C > 64 ? 128 : (C > 32 ? 64 : 32);
        const int gx = (C + BNsel - 1) / BNsel, gy = (main.M + BM - 1) / BM;
        main.xcd_remap = (g_conv_xcd && gx > 1 && gy >= 8) ? 1 : 0;
    }
    if (bf16) MMH_REQUIRE(bf16_ok(main), "bf16 dgrad needs Cout %% 64 == 0 (Cout=%d)", d->Cout);
    if (with_main) mp.p[n++] = main;      // else: the main term was produced by the Winograd path
    mp.n = n;
    int rc = 0;
    if (!(phase & 1)) {
    } else
    if (!bf16 && g_conv_bn256 == 2 && C % 256 == 0) {   // measured: no gain for the [n][k] weight tile
        if (with_main) mp.p[n - 1].xcd_remap = (g_conv_xcd && C / 256 > 1 && (main.M + BM - 1) / BM >= 8) ? 1 : 0;
        rc = launch_multi_t<256, 2, 2>(mp, false, st);
    } else if (C > 64 && !(g_border_bn64 && !with_main && (!bf16 || C <= 256))) rc = launch_multi_t<128, 2, 2>(mp, bf16, st);   // 16-bit border-only: 64-wide tiles up to 256 channels (35 vs 43 us; 512: 74 vs 66)
    else if (C > 32) rc = launch_multi_t<64, 2, 2>(mp, bf16, st);   // border-only: narrower tiles, 2x the workgroups
    else rc = launch_multi_t<32, 4, 1>(mp, bf16, st);
    if (rc) return rc;
    if (!(phase & 2)) return 0;
    const int nr = (H - 2 == 1) ? 1 : 2, nc = (W - 2 == 1) ? 1 : 2;
    const int64_t total = (int64_t)d->B * (nr * W + (H - nr) * nc) * (C / 4);
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, 2048));
    if (dx16 && d->dtype == MMH_FP16)
        hipLaunchKernelGGL(border_add_kernel<2>, dim3(blocks), dim3(256), 0, st, dx, rows, cols, corners, d->B, H, W, C / 4);
    else if (dx16)
        hipLaunchKernelGGL(border_add_kernel<1>, dim3(blocks), dim3(256), 0, st, dx, rows, cols, corners, d->B, H, W, C / 4);
    else
        hipLaunchKernelGGL(border_add_kernel<0>, dim3(blocks), dim3(256), 0, st, dx, rows, cols, corners, d->B, H, W, C / 4);
    return mmh::check_launch("border_add_kernel");
}

// Yhat[xi][tile][C] = A dY A^T for the 2x2 output-gradient tile (Winograd wgrad).
template <int BF>
__global__ void wino_dy_kernel(const float* __restrict__ dy, void* __restrict__ Yh, int B, int H, int W,
                               int C4) {
    const int TH = H / 2, TW = W / 2;
    const long long tiles = (long long)B * TH * TW;
    const long long total = tiles * C4;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C4);
    const long long tile = i / C4;
    const int tx = (int)(tile % TW);
    const int ty = (int)((tile / TW) % TH);
    const int b = (int)(tile / ((long long)TW * TH));
    const float4* in = reinterpret_cast<const float4*>(dy) + (((long long)b * H + 2 * ty) * W + 2 * tx) * C4 + c;
    const float4 y00 = in[0], y01 = in[C4], y10 = in[(long long)W * C4], y11 = in[(long long)W * C4 + C4];
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    // rows of A dY: [y0, y0+y1, y0-y1, -y1]
    float4 t[4][2] = {{y00, y01}, {f4add(y00, y10), f4add(y01, y11)}, {f4sub(y00, y10), f4sub(y01, y11)},
                      {f4sub(z, y10), f4sub(z, y11)}};
    const long long plane = tiles * C4;
    const long long o = tile * C4 + c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        wst4<BF>(Yh, (long long)(r * 4 + 0) * plane + o, t[r][0]);
        wst4<BF>(Yh, (long long)(r * 4 + 1) * plane + o, f4add(t[r][0], t[r][1]));
        wst4<BF>(Yh, (long long)(r * 4 + 2) * plane + o, f4sub(t[r][0], t[r][1]));
        wst4<BF>(Yh, (long long)(r * 4 + 3) * plane + o, f4sub(z, t[r][1]));
    }
}

// dw[3][3][Cin][Cout] (+)= G^T dU G, dU: [16][Cin][Cout]
__global__ void wino_dw_kernel(const float* __restrict__ dU, float* __restrict__ dw, int64_t plane4,
                               int accumulate) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= plane4) return;
    const float4* in = reinterpret_cast<const float4*>(dU) + i;
    float4 u[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) u[a][b] = in[(int64_t)(a * 4 + b) * plane4];
    auto half = [](float4 v) { return make_float4(0.5f * v.x, 0.5f * v.y, 0.5f * v.z, 0.5f * v.w); };
    float4 t[3][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const float4 hs = half(f4add(u[1][b], u[2][b])), hd = half(f4sub(u[1][b], u[2][b]));
        t[0][b] = f4add(u[0][b], hs);
        t[1][b] = hd;
        t[2][b] = f4add(hs, u[3][b]);
    }
    float4* out = reinterpret_cast<float4*>(dw) + i;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float4 hs = half(f4add(t[a][1], t[a][2])), hd = half(f4sub(t[a][1], t[a][2]));
        float4 g0 = f4add(t[a][0], hs), g1 = hd, g2 = f4add(hs, t[a][3]);
        if (accumulate) {
            g0 = f4add(g0, out[(int64_t)(a * 3 + 0) * plane4]);
            g1 = f4add(g1, out[(int64_t)(a * 3 + 1) * plane4]);
            g2 = f4add(g2, out[(int64_t)(a * 3 + 2) * plane4]);
        }
        out[(int64_t)(a * 3 + 0) * plane4] = g0;
        out[(int64_t)(a * 3 + 1) * plane4] = g1;
        out[(int64_t)(a * 3 + 2) * plane4] = g2;
    }
}

// ---------------------------------------------------------------------------
// Winograd F(4x4, 3x3): 36 multiplications per 4x4 output tile (2.25 per output instead of 9):
// 4x fewer MFMA flops than the direct kernel and a transformed tensor only 2.25x the input
// (F(2x2,3x3): 4x).  Same pipeline: 36 batched GEMMs between the transforms.  Matrices are the
// standard ones for interpolation points {0, +-1, +-2, inf}; fp32 error with K = 512 is 2e-6
// relative (checked against fp64 direct convolution).  One thread per (tile, 2 channels).
// ---------------------------------------------------------------------------
struct F2 { float x, y; };
__device__ __forceinline__ F2 operator+(F2 a, F2 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ F2 operator-(F2 a, F2 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ F2 operator*(float k, F2 a) { return {k * a.x, k * a.y}; }

// out[0..5] = B^T in[0..5]
__device__ __forceinline__ void w4_bt(const F2* in, F2* out) {
    out[0] = 4.f * in[0] - 5.f * in[2] + in[4];
    out[1] = in[3] + in[4] - 4.f * (in[1] + in[2]);
    out[2] = 4.f * (in[1] - in[2]) - in[3] + in[4];
    out[3] = 2.f * (in[3] - in[1]) - in[2] + in[4];
    out[4] = 2.f * (in[1] - in[3]) - in[2] + in[4];
    out[5] = 4.f * in[1] - 5.f * in[3] + in[5];
}
// out[0..3] = A^T in[0..5]
__device__ __forceinline__ void w4_at(const F2* in, F2* out) {
    const F2 a = in[1] + in[2], b = in[1] - in[2], c = in[3] + in[4], d = in[3] - in[4];
    out[0] = in[0] + a + c;
    out[1] = b + 2.f * d;
    out[2] = a + 4.f * c;
    out[3] = b + 8.f * d + in[5];
}
// out[0..5] = A in[0..3]
__device__ __forceinline__ void w4_a(const F2* in, F2* out) {
    out[0] = in[0];
    out[1] = in[0] + in[1] + in[2] + in[3];
    out[2] = in[0] - in[1] + in[2] - in[3];
    out[3] = in[0] + 2.f * in[1] + 4.f * in[2] + 8.f * in[3];
    out[4] = in[0] - 2.f * in[1] + 4.f * in[2] - 8.f * in[3];
    out[5] = in[3];
}
// out[0..5] = G in[0..2]
__device__ __forceinline__ void w4_g(const float* in, float* out) {
    out[0] = 0.25f * in[0];
    out[1] = (-1.f / 6.f) * (in[0] + in[1] + in[2]);
    out[2] = (-1.f / 6.f) * (in[0] - in[1] + in[2]);
    out[3] = (1.f / 24.f) * in[0] + (1.f / 12.f) * in[1] + (1.f / 6.f) * in[2];
    out[4] = (1.f / 24.f) * in[0] - (1.f / 12.f) * in[1] + (1.f / 6.f) * in[2];
    out[5] = in[2];
}
// out[0..2] = G^T in[0..5]
__device__ __forceinline__ void w4_gt(const F2* in, F2* out) {
    out[0] = 0.25f * in[0] - (1.f / 6.f) * (in[1] + in[2]) + (1.f / 24.f) * (in[3] + in[4]);
    out[1] = (1.f / 6.f) * (in[2] - in[1]) + (1.f / 12.f) * (in[3] - in[4]);
    out[2] = (1.f / 6.f) * (in[3] + in[4] - in[1] - in[2]) + in[5];
}

__global__ void wino4_weights_kernel(const float* __restrict__ w, float* __restrict__ U, int Cin, int Cout,
                                     int flip_transpose) {
    const int total = Cin * Cout;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    // thread -> (ci, co) with the output's fastest index fastest (coalesced plane stores)
    const int ci = flip_transpose ? i % Cin : i / Cout, co = flip_transpose ? i / Cin : i - (i / Cout) * Cout;
    float g[3][3], t[6][3], col[3], o6[6];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int ka = flip_transpose ? 2 - a : a, kb = flip_transpose ? 2 - b : b;
            g[a][b] = w[((size_t)(ka * 3 + kb) * Cin + ci) * Cout + co];
        }
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        col[0] = g[0][b]; col[1] = g[1][b]; col[2] = g[2][b];
        w4_g(col, o6);
#pragma unroll
        for (int a = 0; a < 6; ++a) t[a][b] = o6[a];
    }
    const size_t plane = (size_t)Cin * Cout;
    const size_t o = flip_transpose ? (size_t)co * Cin + ci : (size_t)ci * Cout + co;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        w4_g(t[a], o6);
#pragma unroll
        for (int b = 0; b < 6; ++b) U[(size_t)(a * 6 + b) * plane + o] = o6[b];
    }
}

__global__ void wino4_input_kernel(const float* __restrict__ x, float* __restrict__ V, int B, int H, int W,
                                   int C2, int reflect, int xcd_remap) {
    const int TH = H / 4, TW = W / 4;
    const long long tiles = (long long)B * TH * TW;
    // workgroups are dealt round-robin over the 8 XCDs: give XCD x a contiguous range of tiles, so
    // that the 6x6 windows of neighbouring tiles (2 shared rows / columns each) meet in ONE L2
    // instead of being fetched from HBM once per XCD (measured: FETCH_SIZE 2.0x -> see profiles/)
    unsigned blk = blockIdx.x;
    if (xcd_remap) blk = (blk & 7u) * (gridDim.x >> 3) + (blk >> 3);
    long long i = (long long)blk * blockDim.x + threadIdx.x;
    if (i >= tiles * C2) return;
    const int c = (int)(i % C2);
    const long long tile = i / C2;
    const int tx = (int)(tile % TW), ty = (int)((tile / TW) % TH), b = (int)(tile / ((long long)TW * TH));
    const F2* xin = reinterpret_cast<const F2*>(x);
    F2 d[6][6], t[6][6], colv[6], o6[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        int hh = 4 * ty - 1 + r;
        bool okh = true;
        if (reflect) { hh = hh < 0 ? -hh : hh; hh = hh >= H ? 2 * (H - 1) - hh : hh; }
        else okh = hh >= 0 && hh < H;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            int ww = 4 * tx - 1 + q;
            bool ok = okh;
            if (reflect) { ww = ww < 0 ? -ww : ww; ww = ww >= W ? 2 * (W - 1) - ww : ww; }
            else ok = ok && ww >= 0 && ww < W;
            d[r][q] = ok ? xin[(((long long)b * H + hh) * W + ww) * C2 + c] : F2{0.f, 0.f};
        }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
#pragma unroll
        for (int r = 0; r < 6; ++r) colv[r] = d[r][q];
        w4_bt(colv, o6);
#pragma unroll
        for (int r = 0; r < 6; ++r) t[r][q] = o6[r];
    }
    const long long plane = tiles * C2;
    F2* out = reinterpret_cast<F2*>(V) + tile * C2 + c;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        w4_bt(t[r], o6);
#pragma unroll
        for (int q = 0; q < 6; ++q) out[(long long)(r * 6 + q) * plane] = o6[q];
    }
}

__global__ void wino4_output_kernel(const float* __restrict__ M, float* __restrict__ y,
                                    const float* __restrict__ bias, int B, int H, int W, int C2, int act) {
    const int TH = H / 4, TW = W / 4;
    const long long tiles = (long long)B * TH * TW;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tiles * C2) return;
    const int c = (int)(i % C2);
    const long long tile = i / C2;
    const int tx = (int)(tile % TW), ty = (int)((tile / TW) % TH), b = (int)(tile / ((long long)TW * TH));
    const long long plane = tiles * C2;
    const F2* in = reinterpret_cast<const F2*>(M) + tile * C2 + c;
    F2 m[6][6], s4[4][6], colv[6], o4[4];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int q = 0; q < 6; ++q) m[r][q] = in[(long long)(r * 6 + q) * plane];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
#pragma unroll
        for (int r = 0; r < 6; ++r) colv[r] = m[r][q];
        w4_at(colv, o4);
#pragma unroll
        for (int r = 0; r < 4; ++r) s4[r][q] = o4[r];
    }
    const F2 bv = bias ? reinterpret_cast<const F2*>(bias)[c] : F2{0.f, 0.f};
    F2* yo = reinterpret_cast<F2*>(y);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        w4_at(s4[r], o4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            F2 v = o4[q] + bv;
            v.x = apply_act(v.x, act); v.y = apply_act(v.y, act);
            yo[(((long long)b * H + 4 * ty + r) * W + 4 * tx + q) * C2 + c] = v;
        }
    }
}

// Yhat[36][tile][C] = A dY A^T for the 4x4 output-gradient tile
__global__ void wino4_dy_kernel(const float* __restrict__ dy, float* __restrict__ Yh, int B, int H, int W,
                                int C2) {
    const int TH = H / 4, TW = W / 4;
    const long long tiles = (long long)B * TH * TW;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tiles * C2) return;
    const int c = (int)(i % C2);
    const long long tile = i / C2;
    const int tx = (int)(tile % TW), ty = (int)((tile / TW) % TH), b = (int)(tile / ((long long)TW * TH));
    const F2* in = reinterpret_cast<const F2*>(dy);
    F2 yv[4][4], t[6][4], colv[4], o6[6];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) yv[r][q] = in[(((long long)b * H + 4 * ty + r) * W + 4 * tx + q) * C2 + c];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int r = 0; r < 4; ++r) colv[r] = yv[r][q];
        w4_a(colv, o6);
#pragma unroll
        for (int r = 0; r < 6; ++r) t[r][q] = o6[r];
    }
    const long long plane = tiles * C2;
    F2* out = reinterpret_cast<F2*>(Yh) + tile * C2 + c;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        w4_a(t[r], o6);
#pragma unroll
        for (int q = 0; q < 6; ++q) out[(long long)(r * 6 + q) * plane] = o6[q];
    }
}

// dw[3][3][Cin][Cout] (+)= G^T dU G, dU: [36][Cin][Cout]
__global__ void wino4_dw_kernel(const float* __restrict__ dU, float* __restrict__ dw, int64_t plane2,
                                int accumulate) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= plane2) return;
    const F2* in = reinterpret_cast<const F2*>(dU) + i;
    F2 u[6][6], t[3][6], colv[6], o3[3];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) u[a][b] = in[(int64_t)(a * 6 + b) * plane2];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
#pragma unroll
        for (int a = 0; a < 6; ++a) colv[a] = u[a][b];
        w4_gt(colv, o3);
#pragma unroll
        for (int a = 0; a < 3; ++a) t[a][b] = o3[a];
    }
    F2* out = reinterpret_cast<F2*>(dw) + i;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        w4_gt(t[a], o3);
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            F2 v = o3[b];
            if (accumulate) v = v + out[(int64_t)(a * 3 + b) * plane2];
            out[(int64_t)(a * 3 + b) * plane2] = v;
        }
    }
}

// ------------------------------------------------------------------ Winograd host side
template <int BN, int WM, int WN, bool NMAJOR>
int launch_batched_t(const BatchKP& bp, int nbatch, hipStream_t st) {
    constexpr size_t lds = (BM * LDA + (NMAJOR ? BN * LDA : BK * BN)) * sizeof(float) + tab_bytes(BM);
    static int ready = -1;
    if (ready != 0) ready = allow_lds(conv_igemm_batched_kernel<BN, WM, WN, NMAJOR>, lds);
    if (ready != 0) return ready;
    dim3 grid((bp.p.N + BN - 1) / BN, (bp.p.M + BM - 1) / BM, nbatch);
    hipLaunchKernelGGL((conv_igemm_batched_kernel<BN, WM, WN, NMAJOR>), grid, dim3(256), lds, st, bp);
    return mmh::check_launch("conv_igemm_batched_kernel");
}

int g_wino_bn256 = 0;   // 128-wide tiles: 3-5 % faster than 256 for these short-K GEMMs (more workgroups per CU)
int g_wino_xcd = 1;     // XCD-contiguous tile order in the input transform (halo rows meet in one L2)
int g_wino_gemm_v2 = 1; // dedicated persistent kernel (wino_gemm_kernel) when K % 32 == 0 and N >= 64
int g_wino_bf16_bk = 64;   // k-step of the bf16 NT GEMM: 64 | 128
int g_wino_bf16_occ = 3;   // resident workgroups per CU the bf16 Winograd GEMM grids are sized for
int g_wino_gemm_occ = 3;   // resident workgroups per CU the persistent grid is sized for
// 2 = two-level summation over the contraction for the F(6x6,3x3) GEMMs (64 planes), whose output
// transform amplifies accumulated rounding the most: measured 6.7e-6 -> 2.4e-6 relative L1 against
// fp64 on 512->512 (direct kernel 1.1e-6) for 5-9 % of the GEMM's time; 1 = one chain
// (mmh_set_option "wino_gemm_levels")
int g_wino_gemm_levels = 2;
int g_wino_gemm_bn = 0;       // column-tile width of the persistent GEMM: 0 = auto (128 when N > 64), 64, 128

template <int BN, int LEVELS>
static int launch_wino_gemm_t(const WinoGemmKP& p, hipStream_t st) {
    constexpr size_t lds = (size_t)(BM * LDA + BK * BN) * sizeof(float);
    static int ready = -1;
    if (ready != 0) ready = allow_lds(wino_gemm_kernel<BN, LEVELS>, lds);
    if (ready != 0) return ready;
    hipLaunchKernelGGL((wino_gemm_kernel<BN, LEVELS>), dim3(8 * p.nb), dim3(256), lds, st, p);
    return mmh::check_launch("wino_gemm_kernel");
}

static int wino_gemm_v2(const float* V, const float* U, float* Mo, long long tiles, int K, int N, hipStream_t st,
                        int nbatch, int levels) {
    WinoGemmKP p{};
    p.A = V; p.B = U; p.C = Mo;
    p.M = (int)tiles; p.K = K; p.N = N; p.P = nbatch;
    const int bn = g_wino_gemm_bn == 64 ? 64 : (N > 64 ? 128 : 64);
    p.MT = (p.M + BM - 1) / BM;
    p.NT = (N + bn - 1) / bn;
    p.W = nbatch * p.MT * p.NT;
    p.Wx = (p.W + 7) / 8;
    const bool two = (levels ? levels : g_wino_gemm_levels) == 2 && nbatch == 64 && K > WINO_FOLD * BK;   // nothing to fold below 2 blocks
    p.nb = std::min(p.Wx, 32 * g_wino_gemm_occ);       // 32 CUs per XCD
    if (two) return bn == 128 ? launch_wino_gemm_t<128, 2>(p, st) : launch_wino_gemm_t<64, 2>(p, st);
    return bn == 128 ? launch_wino_gemm_t<128, 1>(p, st) : launch_wino_gemm_t<64, 1>(p, st);
}

// 16 x ( [tiles x K] . [K x N] ): V [16][tiles][K], U [16][K][N] -> M [16][tiles][N]
int wino_gemm(const float* V, const float* U, float* Mo, long long tiles, int K, int N, hipStream_t st,
              int nbatch = 16, int levels = 0) {
    MMH_REQUIRE(tiles * (long long)std::max(K, N) < (1ll << 30), "winograd: tensor too large");
    if (g_wino_gemm_v2 && K % BK == 0 && N >= 64 && N % 32 == 0) return wino_gemm_v2(V, U, Mo, tiles, K, N, st, nbatch, levels);
    BatchKP bp{};
    ConvKP& p = bp.p;
    Gather& g = p.g;
    g.src = V;
    g.src_bytes = (unsigned)((size_t)tiles * K * sizeof(float));
    g.srcH = 1; g.srcW = (int)tiles; g.src_cs = (unsigned)K;
    g.PH = 1; g.PW = (int)tiles; g.TH = 1; g.TW = 1;
    g.C4 = K / 4;
    g.ap_h = 0; g.at_h = 0; g.a0_h = 0; g.ap_w = 1; g.at_w = 0; g.a0_w = 0;
    g.shift = 0; g.reflect = 0;
    set_korder(g, p.nk, p.Kflat);
    p.w = U;
    p.w_bytes = (unsigned)((size_t)K * N * sizeof(float));
    p.out = Mo;
    p.M = (int)tiles; p.N = N;
    p.wCin = K; p.wCout = N;
    p.KW_true = 1; p.kh0 = 0; p.kw0 = 0; p.tstep = 1;
    p.OH = 1; p.OW = (int)tiles; p.o_p = 1;
    p.out_cs = (unsigned)N; p.out_linear = 1;
    p.act = MMH_ACT_NONE;
    bp.src_bs = tiles * K; bp.w_bs = (long long)K * N; bp.out_bs = tiles * N;
    {
        const int bn = (N % 256 == 0 && g_wino_bn256) ? 256 : 128;
        const int gx = (N + bn - 1) / bn, gy = (p.M + BM - 1) / BM;
        p.xcd_remap = (g_conv_xcd && gx > 1 && gy >= 8) ? 1 : 0;
    }
    if (N % 256 == 0 && g_wino_bn256) return launch_batched_t<256, 2, 2, false>(bp, nbatch, st);
    if (N > 64) return launch_batched_t<128, 2, 2, false>(bp, nbatch, st);
    if (N > 32) return launch_batched_t<64, 2, 2, false>(bp, nbatch, st);
    return launch_batched_t<32, 4, 1, false>(bp, nbatch, st);
}

// Split-K factor: fill whole rounds of the 512 resident workgroups (256 CUs x 2 per CU, LDS
// bound) so the last round is not a mostly empty tail; keep >= 8 k-steps per split.
int g_wgrad_slots = 768;   // tuning knob (mmh_set_option "wgrad_slots")
int g_wgrad_bn256 = 1;     // 128x256 wgrad tile when Cout % 256 == 0
int g_wino_wgrad_v2 = 1;   // dedicated persistent kernel when Cin, Cout % 128 == 0
int g_wino_wgrad_occ = 3;  // 3 | 4 resident workgroups per CU (4 = 128-VGPR build)
int g_wgrad_xcd = 1;     // XCD-contiguous workgroup order for the batched Winograd wgrad GEMMs
int g_wino_wgrad_bn256 = 0;   // 128-wide tiles measured faster for the batched Winograd wgrad GEMMs
int g_wino_wgrad_slots = 2304;   // 3 waves of blocks: measured 20-29 % faster than 768 on the 512- and 256-channel shapes

int wgrad_splits(int Mrows, int N, int P) {
    const int bn = (g_wgrad_bn256 && N % 256 == 0) ? 256 : 128;
    const int tiles = ((Mrows + BM - 1) / BM) * ((N + bn - 1) / bn);
    const int slots = g_wgrad_slots;
    int max_splits = P / (8 * BK);
    if (max_splits > 256) max_splits = 256;
    if (max_splits < 1) max_splits = 1;
    int best = 1;
    double best_eff = 0.0;
    for (int s = 1; s <= max_splits; ++s) {
        const int blocks = tiles * s;
        const int rounds = (blocks + slots - 1) / slots;
        const double eff = (double)blocks / ((double)rounds * slots);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = s; }
        if (blocks >= 4 * slots) break;
    }
    return best;
}

int g_wgrad_dbuf = 0;   // tuning knob (mmh_set_option "wgrad_dbuf")

template <int BN, int WM, int WN, bool DBUF>
int launch_wgrad_t(const WgradKP& p, int splits, hipStream_t st) {
    constexpr size_t lds = (DBUF ? 2 : 1) * (BK * LDW + BK * BN) * sizeof(float);
    static int ready = -1;
    if (ready != 0) ready = allow_lds(conv_wgrad_kernel<BN, WM, WN, DBUF>, lds);
    if (ready != 0) return ready;
    dim3 grid((p.N + BN - 1) / BN, (p.Mrows + BM - 1) / BM, splits);
    hipLaunchKernelGGL((conv_wgrad_kernel<BN, WM, WN, DBUF>), grid, dim3(256), lds, st, p);
    return mmh::check_launch("conv_wgrad_kernel");
}

template <int BN, int WM, int WN>
int launch_wgrad_bf16_t(const WgradKP& p, int splits, hipStream_t st) {
    constexpr size_t lds = (size_t)2 * BKP * LDT * 2;
    dim3 grid((p.N + BN - 1) / BN, (p.Mrows + BM - 1) / BM, splits);
    if (p.x16) {
        static int ready16 = -1;
        if (ready16 != 0) ready16 = allow_lds(conv_wgrad_bf16_kernel<BN, WM, WN, true>, lds);
        if (ready16 != 0) return ready16;
        hipLaunchKernelGGL((conv_wgrad_bf16_kernel<BN, WM, WN, true>), grid, dim3(256), lds, st, p);
        return mmh::check_launch("conv_wgrad_bf16_kernel");
    }
    static int ready = -1;
    if (ready != 0) ready = allow_lds(conv_wgrad_bf16_kernel<BN, WM, WN, false>, lds);
    if (ready != 0) return ready;
    hipLaunchKernelGGL((conv_wgrad_bf16_kernel<BN, WM, WN, false>), grid, dim3(256), lds, st, p);
    return mmh::check_launch("conv_wgrad_bf16_kernel");
}

template <int BN, int WM, int WN>
int launch_wgrad_grid_t(const WgradKP& p, dim3 grid, hipStream_t st) {
    constexpr size_t lds = (BK * LDW + BK * BN) * sizeof(float);
    static int ready = -1;
    if (ready != 0) ready = allow_lds(conv_wgrad_kernel<BN, WM, WN, false>, lds);
    if (ready != 0) return ready;
    hipLaunchKernelGGL((conv_wgrad_kernel<BN, WM, WN, false>), grid, dim3(256), lds, st, p);
    return mmh::check_launch("conv_wgrad_kernel");
}

size_t wgrad_ws(const mmh_conv_desc* d) {
    const int Mrows = d->kh * d->kw * d->Cin;
    const int P = d->B * d->Ho * d->Wo;
    const size_t generic = (size_t)wgrad_splits(Mrows, d->Cout, P) * Mrows * d->Cout * sizeof(float);
    return mmh::wgrad_s2_strip_ok(d) ? std::max(generic, mmh::wgrad_s2_strip_ws_bytes(d)) : generic;
}

int do_wgrad(const mmh_conv_desc* d, const void* x, const void* dy, void* dw, void* ws,
             size_t ws_bytes, int accumulate, hipStream_t st, bool x16 = false, bool dy16 = false) {
    MMH_REQUIRE(!(x16 || dy16) || is16(d->dtype), "wgrad: 16-bit operands need a 16-bit dtype");
    MMH_REQUIRE(x16 == dy16, "wgrad: io16 must be 0 (both tensors fp32) or 3 (both 16-bit)");
    if (!x16 && mmh::wgrad_s2_strip_ok(d)) return mmh::launch_wgrad_s2_strip(d, x, dy, dw, ws, ws_bytes, accumulate, st);
    WgradKP p{};
    p.g = fwd_gather(d, x);
    p.g.chunk_major = 0;
    p.x16 = x16 ? 1 : 0;
    p.dy16 = dy16 ? 1 : 0;
    if (x16) p.g.src_bytes /= 2;
    p.dy = static_cast<const float*>(dy);
    p.dy_bytes = (unsigned)((size_t)d->B * d->Ho * d->Wo * d->y_cs * (dy16 ? 2 : sizeof(float)));
    p.dy_cs = (unsigned)d->y_cs;
    p.Mrows = d->kh * d->kw * d->Cin;
    p.N = d->Cout;
    p.P = d->B * d->Ho * d->Wo;
    const int splits = wgrad_splits(p.Mrows, p.N, p.P);
    MMH_REQUIRE(ws_bytes >= (size_t)splits * p.Mrows * p.N * sizeof(float),
                "wgrad workspace too small: %zu < %zu", ws_bytes,
                (size_t)splits * p.Mrows * p.N * sizeof(float));
    p.slab = static_cast<float*>(ws);
    const bool bf16 = is16(d->dtype);
    p.h16 = d->dtype == MMH_FP16;
    p.pix_per_split = (int)(mmh::cdiv(mmh::cdiv(p.P, splits), bf16 ? BKP : BK) * (bf16 ? BKP : BK));
    p.dbg = g_conv_dbg;
    p.nsplit = splits;
    int rc;
    if (bf16) {
        if (p.N > 64) rc = launch_wgrad_bf16_t<128, 2, 2>(p, splits, st);
        else if (p.N > 32) rc = launch_wgrad_bf16_t<64, 2, 2>(p, splits, st);
        else rc = launch_wgrad_bf16_t<32, 4, 1>(p, splits, st);
    } else if (g_wgrad_bn256 && p.N % 256 == 0) rc = launch_wgrad_t<256, 2, 2, false>(p, splits, st);
    else if (p.N > 64) rc = g_wgrad_dbuf ? launch_wgrad_t<128, 2, 2, true>(p, splits, st)
                                    : launch_wgrad_t<128, 2, 2, false>(p, splits, st);
    else if (p.N > 32) rc = launch_wgrad_t<64, 2, 2, false>(p, splits, st);
    else rc = launch_wgrad_t<32, 4, 1, false>(p, splits, st);
    if (rc) return rc;
    const int64_t n4 = (int64_t)p.Mrows * p.N / 4;
    return mmh::launch_slab_reduce(p.slab, static_cast<float*>(dw), n4, splits, accumulate, n4, st);
}

}  // namespace

extern "C" {

int mmh_set_option(const char* key, int value) {
    MMH_REQUIRE(key != nullptr, "mmh_set_option: NULL key");
    if (!strcmp(key, "conv_dbuf")) { g_conv_dbuf = value; return 0; }
    if (!strcmp(key, "conv_dbg")) { g_conv_dbg = value; return 0; }
    if (!strcmp(key, "conv_cw")) { g_conv_cw = value; return 0; }
    if (!strcmp(key, "conv_bn256")) { g_conv_bn256 = value; return 0; }
    if (!strcmp(key, "conv_levels")) { g_conv_levels = value == 2 ? 2 : 1; mmh::g_stem_f32_levels = g_conv_levels; return 0; }
    if (!strcmp(key, "wino_bn256")) { g_wino_bn256 = value; return 0; }
    if (!strcmp(key, "conv_xcd")) { g_conv_xcd = value; return 0; }
    if (!strcmp(key, "conv_xcd1")) { g_conv_xcd1 = value; return 0; }
    if (!strcmp(key, "conv_tall")) { g_conv_tall = value; return 0; }
    if (!strcmp(key, "wgrad_slots")) { g_wgrad_slots = value; return 0; }
    if (!strcmp(key, "wgrad_dbuf")) { g_wgrad_dbuf = value; return 0; }
    if (!strcmp(key, "wgrad_bn256")) { g_wgrad_bn256 = value; return 0; }
    if (!strcmp(key, "wgrad_xcd")) { g_wgrad_xcd = value; return 0; }
    if (!strcmp(key, "wino_wgrad_v2")) { g_wino_wgrad_v2 = value; return 0; }
    if (!strcmp(key, "wino_wgrad_occ")) { g_wino_wgrad_occ = value; return 0; }
    if (!strcmp(key, "wino_gemm_v2")) { g_wino_gemm_v2 = value; return 0; }
    if (!strcmp(key, "wino_xcd")) { g_wino_xcd = value; return 0; }
    if (!strcmp(key, "dgrad_s2_multi")) { g_dgrad_s2_multi = value; return 0; }
    if (!strcmp(key, "wino6_vec")) { mmh::g_wino6_vec = value; return 0; }
    if (!strcmp(key, "lp16_shape")) { mmh::g_lp16_shape = value; return 0; }
    if (!strcmp(key, "lp16_s2f")) { mmh::g_lp16_s2f = value; return 0; }
    if (!strcmp(key, "lp16_persist")) { mmh::g_lp16_persist = value; return 0; }
    if (!strcmp(key, "lp16_tap_inner")) { mmh::g_lp16_tap_inner = value; return 0; }
    if (!strcmp(key, "pw_v2")) { mmh::g_pw_v2 = value; return 0; }
    if (!strcmp(key, "col_chunks") && value > 0) { mmh::g_col_chunks = value; return 0; }
    if (!strcmp(key, "row_chunks") && value > 0) { mmh::g_row_chunks = value; return 0; }
    if (!strcmp(key, "dgrad_s2_halo")) { mmh::g_dgrad_s2_halo = value; return 0; }
    if (!strcmp(key, "wgrad_s2_strip")) { mmh::g_wgrad_s2_strip = value; return 0; }
    if (!strcmp(key, "stem_f32")) { mmh::g_stem_f32 = value; return 0; }
    if (!strcmp(key, "wino_wgrad_dma")) { mmh::g_wino_wgrad_dma = value; return 0; }
    if (!strcmp(key, "stem_f32_dbg")) { mmh::g_stem_f32_dbg = value; return 0; }
    if (!strcmp(key, "slab_reduce_par")) { mmh::g_slab_reduce_par = value; return 0; }
    if (!strcmp(key, "dgrad_s2_dbg")) { mmh::g_dgrad_s2_dbg = value; return 0; }
    if (!strcmp(key, "lp16_dbg")) { mmh::g_lp16_dbg = value; return 0; }
    if (!strcmp(key, "lp16_wgrad_ring")) { mmh::g_lp16_wgrad_ring = value; return 0; }
    if (!strcmp(key, "lp16_wgrad_s2")) { mmh::g_lp16_wgrad_s2 = value; return 0; }
    if (!strcmp(key, "border_bn64")) { g_border_bn64 = value; return 0; }
    if (!strcmp(key, "wino_gemm_occ")) { g_wino_gemm_occ = value; return 0; }
    if (!strcmp(key, "wino_gemm_levels")) { g_wino_gemm_levels = value; return 0; }
    if (!strcmp(key, "wino_gemm_bn")) { g_wino_gemm_bn = value; return 0; }
    if (!strcmp(key, "wino_bf16_occ")) { g_wino_bf16_occ = value; return 0; }
    if (!strcmp(key, "wino_bf16_bk")) { g_wino_bf16_bk = value; return 0; }
    if (!strcmp(key, "wino_wgrad_bn256")) { g_wino_wgrad_bn256 = value; return 0; }
    if (!strcmp(key, "wino_wgrad_slots")) { g_wino_wgrad_slots = value; return 0; }
    return mmh::fail("mmh_set_option: unknown key '%s'", key);
}

int mmh_conv2d_fprop(const mmh_conv_desc* d, const void* x, const void* w, const void* bias,
                     void* y, int act, mmh_stream_t s) {
    if (int rc = validate(d)) return rc;
    MMH_REQUIRE(x && w && y, "mmh_conv2d_fprop: NULL buffer");
    return do_fprop(d, x, w, bias, y, act, mmh::as_stream(s));
}

int mmh_conv2d_fprop_stats_chunks(const mmh_conv_desc* d) { return d ? fprop_stats_chunks(d) : 0; }

int mmh_conv2d_fprop_stats(const mmh_conv_desc* d, const void* x, const void* w, const void* bias, void* y,
                           void* stats, mmh_stream_t s) {
    if (int rc = validate(d)) return rc;
    MMH_REQUIRE(x && w && y && stats, "mmh_conv2d_fprop_stats: NULL buffer");
    MMH_REQUIRE(fprop_stats_chunks(d) > 0, "mmh_conv2d_fprop_stats: needs fp32 and B*Ho*Wo %% 128 == 0");
    return do_fprop(d, x, w, bias, y, MMH_ACT_NONE, mmh::as_stream(s), static_cast<float*>(stats));
}

int mmh_conv2d_dgrad(const mmh_conv_desc* d, const void* dy, const void* w, void* dx, int dx_cs,
                     mmh_stream_t s) {
    if (int rc = validate(d)) return rc;
    MMH_REQUIRE(dy && w && dx, "mmh_conv2d_dgrad: NULL buffer");
    MMH_REQUIRE(dx_cs % 4 == 0 && dx_cs >= d->Cin, "mmh_conv2d_dgrad: bad dx_cs=%d", dx_cs);
    return do_dgrad(d, dy, w, nullptr, dx, dx_cs, MMH_ACT_NONE, mmh::as_stream(s));
}

// ---- Winograd stages (tile = 2: F(2x2,3x3), 16 planes; tile = 4: F(4x4,3x3), 36 planes) ----
static int wino_planes(int tile) { return (tile + 2) * (tile + 2); }
static bool wino_tile_ok(int tile) { return tile == 2 || tile == 4 || tile == 6; }
// tiles per image side: F(6x6,3x3) tiles are ragged (any size), the others need divisibility
static bool wino_hw_ok(int H, int W, int tile) {
    return tile == 6 ? (H >= 4 && W >= 4) : (H >= tile + 2 && W >= tile + 2 && H % tile == 0 && W % tile == 0);
}
static long long wino_tiles(int B, int H, int W, int tile) {
    return (long long)B * ((H + tile - 1) / tile) * ((W + tile - 1) / tile);
}

static bool wino_dtype_ok(int dtype, int tile) { return dtype == MMH_F32 || (is16(dtype) && tile == 2); }

int mmh_wino_weights(const void* w, int Cin, int Cout, int flip_transpose, int tile, int dtype, void* U,
                     mmh_stream_t s) {
    MMH_REQUIRE(w && U && Cin > 0 && Cout > 0 && wino_tile_ok(tile) && wino_dtype_ok(dtype, tile),
                "mmh_wino_weights: bad arguments (tile 2 | 4 | 6; bf16 needs tile 2)");
    const dim3 grid((Cin * Cout + 255) / 256);
    if (tile == 6)
        return mmh::wino6_weights(static_cast<const float*>(w), static_cast<float*>(U), Cin, Cout, flip_transpose,
                                  mmh::as_stream(s));
    if (tile == 4)
        hipLaunchKernelGGL(wino4_weights_kernel, grid, dim3(256), 0, mmh::as_stream(s),
                           static_cast<const float*>(w), static_cast<float*>(U), Cin, Cout, flip_transpose);
    else if (dtype == MMH_FP16)
        hipLaunchKernelGGL(wino_weights_kernel<2>, grid, dim3(256), 0, mmh::as_stream(s),
                           static_cast<const float*>(w), U, Cin, Cout, flip_transpose);
    else if (dtype == MMH_BF16)
        hipLaunchKernelGGL(wino_weights_kernel<1>, grid, dim3(256), 0, mmh::as_stream(s),
                           static_cast<const float*>(w), U, Cin, Cout, flip_transpose);
    else
        hipLaunchKernelGGL(wino_weights_kernel<0>, grid, dim3(256), 0, mmh::as_stream(s),
                           static_cast<const float*>(w), U, Cin, Cout, flip_transpose);
    return mmh::check_launch("wino_weights_kernel");
}

// Reflect-fold dgrad (padded-domain tiles, in-tile fold): both ring partners must share a tile.
static bool wino_fold_ok(int H, int W) { return H >= 6 && W >= 6 && (H + 1) % 6 >= 2 && (W + 1) % 6 >= 2; }

int mmh_wino_weights_multi(const void* table, int n, int64_t total_blocks, mmh_stream_t s) {
    MMH_REQUIRE(table && n > 0 && total_blocks > 0 && total_blocks < (1ll << 31), "mmh_wino_weights_multi: bad arguments");
    return mmh::wino6_weights_multi(static_cast<const long long*>(table), n, total_blocks, mmh::as_stream(s));
}

int mmh_wino_input(const void* x, int B, int H, int W, int C, int reflect, int tile, int dtype, void* V,
                   mmh_stream_t s) {
    MMH_REQUIRE(x && V && B > 0 && wino_tile_ok(tile) && wino_hw_ok(H, W, tile) && C % 4 == 0 &&
                    wino_dtype_ok(dtype, tile) && reflect >= 0 && reflect <= 2,
                "mmh_wino_input: bad arguments");
    MMH_REQUIRE(reflect != 2 || (tile == 6 && dtype == MMH_F32 && wino_fold_ok(H, W)),
                "mmh_wino_input: pad mode 2 (padded-domain dgrad) needs tile 6, fp32, (H+1) %% 6 >= 2, (W+1) %% 6 >= 2");
    const long long tiles = wino_tiles(B, H, W, tile);
    if (tile == 6)
        return mmh::wino6_input(static_cast<const float*>(x), static_cast<float*>(V), B, H, W, C, reflect, g_wino_xcd,
                                mmh::as_stream(s));
    if (tile == 4) {
        const long long total = tiles * (C / 2);
        const unsigned nblk = (unsigned)((total + 255) / 256);
        hipLaunchKernelGGL(wino4_input_kernel, dim3(nblk), dim3(256), 0, mmh::as_stream(s),
                           static_cast<const float*>(x), static_cast<float*>(V), B, H, W, C / 2, reflect,
                           (g_wino_xcd && nblk % 8 == 0 && nblk >= 64) ? 1 : 0);
    } else {
        const long long total = tiles * (C / 4);
        const dim3 grid((unsigned)((total + 255) / 256));
        if (dtype == MMH_FP16)
            hipLaunchKernelGGL(wino_input_kernel<2>, grid, dim3(256), 0, mmh::as_stream(s),
                               static_cast<const float*>(x), V, B, H, W, C / 4, reflect);
        else if (dtype == MMH_BF16)
            hipLaunchKernelGGL(wino_input_kernel<1>, grid, dim3(256), 0, mmh::as_stream(s),
                               static_cast<const float*>(x), V, B, H, W, C / 4, reflect);
        else
            hipLaunchKernelGGL(wino_input_kernel<0>, grid, dim3(256), 0, mmh::as_stream(s),
                               static_cast<const float*>(x), V, B, H, W, C / 4, reflect);
    }
    return mmh::check_launch("wino_input_kernel");
}

int mmh_wino_input_dy(const void* dy, int B, int H, int W, int C, int tile, int dtype, void* V, void* Yh,
                      int fold, mmh_stream_t s) {
    MMH_REQUIRE(dy && V && Yh && B > 0 && tile == 6 && dtype == MMH_F32 && wino_hw_ok(H, W, tile) && C % 4 == 0,
                "mmh_wino_input_dy: needs tile 6, fp32");
    MMH_REQUIRE(!fold || (wino_fold_ok(H, W) && (H + 7) / 6 == (H + 5) / 6 && (W + 7) / 6 == (W + 5) / 6),
                "mmh_wino_input_dy: fold needs (H+1) %% 6 >= 2 and ceil((H+2)/6) == ceil(H/6) (W alike)");
    return mmh::wino6_input_dy(static_cast<const float*>(dy), static_cast<float*>(V), static_cast<float*>(Yh), B, H, W,
                               C, g_wino_xcd, fold ? 1 : 0, mmh::as_stream(s));
}

int mmh_wino_input_normact(const void* x, int B, int H, int W, int C, int reflect, void* V, const void* scale,
                           const void* shift, int groups, int relu, float drop_p, const void* drows,
                           mmh_stream_t s) {
    MMH_REQUIRE(x && V && scale && shift && B > 0 && wino_hw_ok(H, W, 6) && C % 8 == 0 && reflect >= 0 && reflect <= 1 &&
                    (groups == 1 || groups == B) && drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f) == (drows == nullptr) &&
                    (drop_p == 0.f || relu),
                "mmh_wino_input_normact: bad arguments (tile 6, fp32, C %% 8 == 0, groups 1 | B, dropout rows iff p > 0)");
    MMH_REQUIRE((long long)B * H * W * C < (1ll << 31), "mmh_wino_input_normact: 32-bit element offsets (numel < 2^31)");
    return mmh::wino6_input_normact(static_cast<const float*>(x), static_cast<float*>(V), B, H, W, C, reflect,
                                    g_wino_xcd, static_cast<const float*>(scale), static_cast<const float*>(shift),
                                    groups, relu, drop_p, static_cast<const uint32_t*>(drows), mmh::as_stream(s));
}

int mmh_wino_input_dy_normbwd(const void* g, const void* x, int B, int H, int W, int C, void* V, void* Yh, int fold,
                              const void* mean, const void* invstd, const void* gamma, const void* s1,
                              const void* s2, double count, const void* scale, const void* shift,
                              const void* drows, int groups, int relu, float drop_p, mmh_stream_t s) {
    MMH_REQUIRE(g && x && V && Yh && mean && invstd && s1 && s2 && scale && shift && B > 0 && wino_hw_ok(H, W, 6) &&
                    C % 8 == 0 && count > 0 && (groups == 1 || groups == B) && drop_p >= 0.f && drop_p < 1.f &&
                    (drop_p == 0.f) == (drows == nullptr),
                "mmh_wino_input_dy_normbwd: bad arguments (tile 6, fp32, C %% 8 == 0, groups 1 | B)");
    MMH_REQUIRE((long long)B * H * W * C < (1ll << 31), "mmh_wino_input_dy_normbwd: 32-bit element offsets (numel < 2^31)");
    MMH_REQUIRE(!fold || (wino_fold_ok(H, W) && (H + 7) / 6 == (H + 5) / 6 && (W + 7) / 6 == (W + 5) / 6),
                "mmh_wino_input_dy_normbwd: fold needs (H+1) %% 6 >= 2 and ceil((H+2)/6) == ceil(H/6) (W alike)");
    return mmh::wino6_input_dy_normbwd(
        static_cast<const float*>(g), static_cast<const float*>(x), static_cast<float*>(V), static_cast<float*>(Yh), B,
        H, W, C, g_wino_xcd, fold ? 1 : 0, static_cast<const float*>(mean), static_cast<const float*>(invstd),
        static_cast<const float*>(gamma), static_cast<const float*>(s1), static_cast<const float*>(s2), count,
        static_cast<const float*>(scale), static_cast<const float*>(shift), static_cast<const uint32_t*>(drows), groups,
        relu, drop_p, mmh::as_stream(s));
}

int mmh_wino_dy(const void* dy, int B, int H, int W, int C, int tile, int dtype, void* Yh, mmh_stream_t s) {
    MMH_REQUIRE(dy && Yh && B > 0 && wino_tile_ok(tile) && wino_hw_ok(H, W, tile) && C % 4 == 0 &&
                    wino_dtype_ok(dtype, tile),
                "mmh_wino_dy: bad arguments");
    const long long tiles = wino_tiles(B, H, W, tile);
    if (tile == 6)
        return mmh::wino6_dy(static_cast<const float*>(dy), static_cast<float*>(Yh), B, H, W, C, mmh::as_stream(s));
    if (tile == 4) {
        const long long total = tiles * (C / 2);
        hipLaunchKernelGGL(wino4_dy_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, mmh::as_stream(s),
                           static_cast<const float*>(dy), static_cast<float*>(Yh), B, H, W, C / 2);
    } else {
        const long long total = tiles * (C / 4);
        const dim3 grid((unsigned)((total + 255) / 256));
        if (dtype == MMH_FP16)
            hipLaunchKernelGGL(wino_dy_kernel<2>, grid, dim3(256), 0, mmh::as_stream(s),
                               static_cast<const float*>(dy), Yh, B, H, W, C / 4);
        else if (dtype == MMH_BF16)
            hipLaunchKernelGGL(wino_dy_kernel<1>, grid, dim3(256), 0, mmh::as_stream(s),
                               static_cast<const float*>(dy), Yh, B, H, W, C / 4);
        else
            hipLaunchKernelGGL(wino_dy_kernel<0>, grid, dim3(256), 0, mmh::as_stream(s),
                               static_cast<const float*>(dy), Yh, B, H, W, C / 4);
    }
    return mmh::check_launch("wino_dy_kernel");
}

static int wino_gemm_bf16(const void* V, const void* U, void* Mo, long long tiles, int K, int N, int nbatch, int h16,
                          hipStream_t st) {
    MMH_REQUIRE(K % BK16 == 0 && N % 32 == 0, "mmh_wino_gemm (bf16): needs K %% 64 == 0 and N %% 32 == 0");
    MMH_REQUIRE(tiles * (long long)std::max(K, N) < (1ll << 30), "winograd: tensor too large");
    WinoGemmBfKP p{};
    p.A = static_cast<const __bf16*>(V); p.B = static_cast<const __bf16*>(U); p.C = static_cast<__bf16*>(Mo);
    p.M = (int)tiles; p.K = K; p.N = N; p.P = nbatch;
    p.h16 = h16;
    p.MT = (p.M + BM - 1) / BM;
    p.NT = (N + 127) / 128;
    p.W = nbatch * p.MT * p.NT;
    p.Wx = (p.W + 7) / 8;
    if (g_wino_bf16_bk == 128 && K % 128 == 0) {
        p.nb = std::min(p.Wx, 32 * 2);
        constexpr size_t lds = (size_t)(2 * BM * (128 + 8)) * sizeof(__bf16);
        static int ready = -1;
        if (ready != 0) ready = allow_lds(wino_gemm_bf16_kernel<128, false>, lds) | allow_lds(wino_gemm_bf16_kernel<128, true>, lds);
        if (ready != 0) return ready;
        if (h16) hipLaunchKernelGGL((wino_gemm_bf16_kernel<128, true>), dim3(8 * p.nb), dim3(256), lds, st, p);
        else hipLaunchKernelGGL((wino_gemm_bf16_kernel<128, false>), dim3(8 * p.nb), dim3(256), lds, st, p);
    } else {
        p.nb = std::min(p.Wx, 32 * g_wino_bf16_occ);
        constexpr size_t lds = (size_t)(2 * BM * (64 + 8)) * sizeof(__bf16);
        static int ready = -1;
        if (ready != 0) ready = allow_lds(wino_gemm_bf16_kernel<64, false>, lds) | allow_lds(wino_gemm_bf16_kernel<64, true>, lds);
        if (ready != 0) return ready;
        if (h16) hipLaunchKernelGGL((wino_gemm_bf16_kernel<64, true>), dim3(8 * p.nb), dim3(256), lds, st, p);
        else hipLaunchKernelGGL((wino_gemm_bf16_kernel<64, false>), dim3(8 * p.nb), dim3(256), lds, st, p);
    }
    return mmh::check_launch("wino_gemm_bf16_kernel");
}

int mmh_wino_gemm(const void* V, const void* U, void* M, int64_t tiles, int K, int N, int nbatch, int dtype,
                  mmh_stream_t s) {
    MMH_REQUIRE(V && U && M && tiles > 0 && K % 32 == 0 && N % 4 == 0 && nbatch > 0 &&
                    (dtype == MMH_F32 || is16(dtype)),
                "mmh_wino_gemm: bad arguments");
    if (is16(dtype)) return wino_gemm_bf16(V, U, M, tiles, K, N, nbatch, dtype == MMH_FP16, mmh::as_stream(s));
    return wino_gemm(static_cast<const float*>(V), static_cast<const float*>(U), static_cast<float*>(M), tiles,
                     K, N, mmh::as_stream(s), nbatch);
}

int mmh_wino_gemm_levels(const void* V, const void* U, void* M, int64_t tiles, int K, int N, int nbatch, int levels,
                         mmh_stream_t s) {
    MMH_REQUIRE(V && U && M && tiles > 0 && K % 32 == 0 && N % 4 == 0 && nbatch > 0 && (levels == 1 || levels == 2),
                "mmh_wino_gemm_levels: bad arguments (fp32; levels 1 | 2)");
    return wino_gemm(static_cast<const float*>(V), static_cast<const float*>(U), static_cast<float*>(M), tiles, K, N,
                     mmh::as_stream(s), nbatch, levels);
}

int mmh_wino_output(const void* M, void* y, const void* bias, int B, int H, int W, int C, int act, int tile,
                    int dtype, void* stats, int fold, mmh_stream_t s) {
    MMH_REQUIRE(M && y && B > 0 && wino_tile_ok(tile) && wino_hw_ok(H, W, tile) && C % 4 == 0 &&
                    wino_dtype_ok(dtype, tile),
                "mmh_wino_output: bad arguments");
    const long long tiles = wino_tiles(B, H, W, tile);
    MMH_REQUIRE(!stats || (tile == 6 && dtype == MMH_F32), "mmh_wino_output: statistics need tile 6, fp32");
    MMH_REQUIRE(!fold || (tile == 6 && dtype == MMH_F32 && wino_fold_ok(H, W) && !bias && !stats && act == MMH_ACT_NONE),
                "mmh_wino_output: fold needs tile 6, fp32, no bias / activation / statistics, (H+1) %% 6 >= 2");
    if (tile == 6)
        return mmh::wino6_output(static_cast<const float*>(M), static_cast<float*>(y), static_cast<const float*>(bias),
                                 B, H, W, C, act, static_cast<float*>(stats), fold ? 1 : 0, mmh::as_stream(s));
    if (tile == 4) {
        const long long total = tiles * (C / 2);
        hipLaunchKernelGGL(wino4_output_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                           mmh::as_stream(s), static_cast<const float*>(M), static_cast<float*>(y),
                           static_cast<const float*>(bias), B, H, W, C / 2, act);
    } else {
        const long long total = tiles * (C / 4);
        const dim3 grid((unsigned)((total + 255) / 256));
        if (dtype == MMH_FP16)
            hipLaunchKernelGGL(wino_output_kernel<2>, grid, dim3(256), 0, mmh::as_stream(s), M,
                               static_cast<float*>(y), static_cast<const float*>(bias), B, H, W, C / 4, act);
        else if (dtype == MMH_BF16)
            hipLaunchKernelGGL(wino_output_kernel<1>, grid, dim3(256), 0, mmh::as_stream(s), M,
                               static_cast<float*>(y), static_cast<const float*>(bias), B, H, W, C / 4, act);
        else
            hipLaunchKernelGGL(wino_output_kernel<0>, grid, dim3(256), 0, mmh::as_stream(s), M,
                               static_cast<float*>(y), static_cast<const float*>(bias), B, H, W, C / 4, act);
    }
    return mmh::check_launch("wino_output_kernel");
}

// dU[xi][Cin][Cout] = sum over tiles V[xi][tile][Cin] * Yh[xi][tile][Cout]: nbatch split-K GEMMs in one
// launch, fixed-order slab reduction (deterministic).
static int wino_wgrad_splits(int Cin, int Cout, long long tiles, int nbatch) {
    const int bn = (g_wino_wgrad_bn256 && Cout % 256 == 0) ? 256 : 128;
    const int per = ((Cin + BM - 1) / BM) * ((Cout + bn - 1) / bn) * nbatch;
    const int s = std::max(1, g_wino_wgrad_slots / per);
    const long long maxs = std::max<long long>(1, tiles / (8 * BK));
    return (int)std::min<long long>(s, maxs);
}

size_t mmh_wino_wgrad_gemm_ws_bytes(int64_t tiles, int Cin, int Cout, int nbatch) {
    if (tiles <= 0 || Cin <= 0 || Cout <= 0 || nbatch <= 0) return 0;
    const size_t generic = (size_t)nbatch * wino_wgrad_splits(Cin, Cout, tiles, nbatch) * Cin * Cout * sizeof(float);
    return mmh::wino_wgrad_dma_ok(tiles, Cin, Cout, nbatch) ? std::max(generic, mmh::wino_wgrad_dma_ws_bytes(tiles, Cin, Cout, nbatch))
                                                            : generic;
}

int mmh_wino_wgrad_gemm(const void* V, const void* Yh, int64_t tiles, int Cin, int Cout, int nbatch, int dtype,
                        void* ws, size_t ws_bytes, void* dU, mmh_stream_t s) {
    MMH_REQUIRE(V && Yh && ws && dU && tiles > 0 && Cin % 4 == 0 && Cout % 4 == 0 &&
                    (dtype == MMH_F32 || is16(dtype)),
                "mmh_wino_wgrad_gemm: bad arguments");
    MMH_REQUIRE(ws_bytes >= mmh_wino_wgrad_gemm_ws_bytes(tiles, Cin, Cout, nbatch),
                "mmh_wino_wgrad_gemm: workspace too small");
    MMH_REQUIRE(tiles * (long long)std::max(Cin, Cout) < (1ll << 30), "mmh_wino_wgrad_gemm: tensor too large");
    hipStream_t st = mmh::as_stream(s);
    if (dtype == MMH_F32 && mmh::wino_wgrad_dma_ok(tiles, Cin, Cout, nbatch))
        return mmh::launch_wino_wgrad_dma(static_cast<const float*>(V), static_cast<const float*>(Yh), tiles, Cin, Cout, nbatch,
                                          static_cast<float*>(ws), static_cast<float*>(dU), st);
    const int splits = wino_wgrad_splits(Cin, Cout, tiles, nbatch);
    if (is16(dtype)) {
        MMH_REQUIRE(Cin % BM == 0 && Cout % 128 == 0, "mmh_wino_wgrad_gemm (bf16): needs Cin, Cout %% 128 == 0");
        WinoWgradBfKP q{};
        q.h16 = dtype == MMH_FP16;
        q.V = static_cast<const __bf16*>(V); q.Y = static_cast<const __bf16*>(Yh); q.slab = static_cast<float*>(ws);
        q.T = (int)tiles; q.Cin = Cin; q.Cout = Cout; q.P = nbatch;
        q.t_per_split = (int)(mmh::cdiv(mmh::cdiv(tiles, splits), BKP) * BKP);
        q.S = (int)mmh::cdiv(tiles, q.t_per_split);
        q.MT = Cin / BM; q.NT = Cout / 128;
        q.W = nbatch * q.S * q.MT * q.NT;
        q.Wx = (q.W + 7) / 8;
        q.nb = std::min(q.Wx, 32 * g_wino_bf16_occ);
        constexpr size_t lds = (size_t)(2 * BKP * LDT) * sizeof(__bf16);
        static int ready = -1;
        if (ready != 0) ready = allow_lds(wino_wgrad_gemm_bf16_kernel<false>, lds) | allow_lds(wino_wgrad_gemm_bf16_kernel<true>, lds);
        if (ready != 0) return ready;
        if (q.h16) hipLaunchKernelGGL(wino_wgrad_gemm_bf16_kernel<true>, dim3(8 * q.nb), dim3(256), lds, st, q);
        else hipLaunchKernelGGL(wino_wgrad_gemm_bf16_kernel<false>, dim3(8 * q.nb), dim3(256), lds, st, q);
        if (int rc = mmh::check_launch("wino_wgrad_gemm_bf16_kernel")) return rc;
        const int64_t n4 = (int64_t)Cin * Cout / 4;
        return mmh::launch_slab_reduce(q.slab, static_cast<float*>(dU), nbatch * n4, q.S, 0, n4, st);
    }
    if (g_wino_wgrad_v2 && Cin % BM == 0 && Cout % 128 == 0) {
        WinoWgradKP q{};
        q.V = static_cast<const float*>(V); q.Y = static_cast<const float*>(Yh); q.slab = static_cast<float*>(ws);
        q.T = (int)tiles; q.Cin = Cin; q.Cout = Cout; q.P = nbatch;
        q.t_per_split = (int)(mmh::cdiv(mmh::cdiv(tiles, splits), BK) * BK);
        q.S = (int)mmh::cdiv(tiles, q.t_per_split);       // every split owns at least one tile
        q.MT = Cin / BM; q.NT = Cout / 128;
        q.W = nbatch * q.S * q.MT * q.NT;
        q.Wx = (q.W + 7) / 8;
        const int occ = g_wino_wgrad_occ == 4 ? 4 : 3;
        q.nb = std::min(q.Wx, 32 * occ);
        constexpr size_t lds = (size_t)(BK * LDW + BK * 128) * sizeof(float);
        static int ready3 = -1, ready4 = -1;
        if (occ == 4) {
            if (ready4 != 0) ready4 = allow_lds(wino_wgrad_gemm_kernel<4>, lds);
            if (ready4 != 0) return ready4;
            hipLaunchKernelGGL(wino_wgrad_gemm_kernel<4>, dim3(8 * q.nb), dim3(256), lds, st, q);
        } else {
            if (ready3 != 0) ready3 = allow_lds(wino_wgrad_gemm_kernel<3>, lds);
            if (ready3 != 0) return ready3;
            hipLaunchKernelGGL(wino_wgrad_gemm_kernel<3>, dim3(8 * q.nb), dim3(256), lds, st, q);
        }
        if (int rc = mmh::check_launch("wino_wgrad_gemm_kernel")) return rc;
        const int64_t n4 = (int64_t)Cin * Cout / 4;
        return mmh::launch_slab_reduce(q.slab, static_cast<float*>(dU), nbatch * n4, q.S, 0, n4, st);
    }
    WgradKP p{};
    Gather& g = p.g;
    g.src = static_cast<const float*>(V);
    g.src_bytes = (unsigned)((size_t)tiles * Cin * sizeof(float));
    g.srcH = 1; g.srcW = (int)tiles; g.src_cs = (unsigned)Cin;
    g.PH = 1; g.PW = (int)tiles; g.TH = 1; g.TW = 1;
    g.C4 = Cin / 4;
    g.ap_h = 0; g.at_h = 0; g.a0_h = 0; g.ap_w = 1; g.at_w = 0; g.a0_w = 0;
    g.shift = 0; g.reflect = 0; g.chunk_major = 0; g.cw = 1;
    p.dy = static_cast<const float*>(Yh);
    p.dy_bytes = (unsigned)((size_t)tiles * Cout * sizeof(float));
    p.dy_cs = (unsigned)Cout;
    p.slab = static_cast<float*>(ws);
    p.Mrows = Cin; p.N = Cout; p.P = (int)tiles;
    p.pix_per_split = (int)(mmh::cdiv(mmh::cdiv(tiles, splits), BK) * BK);
    p.nsplit = splits;
    p.src_bs = tiles * Cin; p.dy_bs = tiles * Cout;
    const bool b256 = g_wino_wgrad_bn256 && p.N % 256 == 0;
    const int BNsel = b256 ? 256 : (p.N > 64 ? 128 : (p.N > 32 ? 64 : 32));
    dim3 grid((p.N + BNsel - 1) / BNsel, (p.Mrows + BM - 1) / BM, nbatch * splits);
    p.xcd_remap = (g_wgrad_xcd && (grid.x * grid.y * grid.z) % 8 == 0) ? 1 : 0;
    int rc;
    if (b256) rc = launch_wgrad_grid_t<256, 2, 2>(p, grid, st);
    else if (p.N > 64) rc = launch_wgrad_grid_t<128, 2, 2>(p, grid, st);
    else if (p.N > 32) rc = launch_wgrad_grid_t<64, 2, 2>(p, grid, st);
    else rc = launch_wgrad_grid_t<32, 4, 1>(p, grid, st);
    if (rc) return rc;
    const int64_t n4 = (int64_t)Cin * Cout / 4;
    return mmh::launch_slab_reduce(p.slab, static_cast<float*>(dU), nbatch * n4, splits, 0, n4, st);
}

int mmh_wino_dw(const void* dU, int Cin, int Cout, int tile, void* dw, int accumulate, mmh_stream_t s) {
    MMH_REQUIRE(dU && dw && Cin % 4 == 0 && Cout % 4 == 0 && wino_tile_ok(tile), "mmh_wino_dw: bad arguments");
    if (tile == 6)
        return mmh::wino6_dw(static_cast<const float*>(dU), static_cast<float*>(dw), Cin, Cout, accumulate,
                             mmh::as_stream(s));
    if (tile == 4) {
        const int64_t n2 = (int64_t)Cin * Cout / 2;
        hipLaunchKernelGGL(wino4_dw_kernel, dim3((unsigned)mmh::cdiv(n2, 256)), dim3(256), 0, mmh::as_stream(s),
                           static_cast<const float*>(dU), static_cast<float*>(dw), n2, accumulate);
    } else {
        const int64_t n4 = (int64_t)Cin * Cout / 4;
        hipLaunchKernelGGL(wino_dw_kernel, dim3((unsigned)mmh::cdiv(n4, 256)), dim3(256), 0, mmh::as_stream(s),
                           static_cast<const float*>(dU), static_cast<float*>(dw), n4, accumulate);
    }
    return mmh::check_launch("wino_dw_kernel");
}

// The eight reflect-border terms of a 3x3 / pad 1 dgrad, added into dx (whose main term came
// from the Winograd path).  ws: mmh_conv2d_dgrad_border_ws_bytes.
size_t mmh_conv2d_dgrad_border_ws_bytes(const mmh_conv_desc* d) { return d ? reflect1_ws_bytes(d) : 0; }

int mmh_conv2d_dgrad_border(const mmh_conv_desc* d, const void* dy, const void* w, void* dx, void* ws,
                            size_t ws_bytes, int phase, int io16, mmh_stream_t s) {
    if (int rc = validate(d)) return rc;
    MMH_REQUIRE(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->pad_mode == MMH_PAD_REFLECT &&
                    d->H >= 3 && d->W >= 3 && d->x_cs == d->Cin,
                "mmh_conv2d_dgrad_border: needs a dense 3x3 stride-1 reflect-pad-1 conv");
    MMH_REQUIRE(phase >= 1 && phase <= 3, "mmh_conv2d_dgrad_border: phase must be 1 (GEMMs), 2 (add) or 3 (both)");
    MMH_REQUIRE(dy && w && (dx || !(phase & 2)) && ws && ws_bytes >= reflect1_ws_bytes(d),
                "mmh_conv2d_dgrad_border: bad buffers");
    MMH_REQUIRE(!io16 || (is16(d->dtype) && d->y_cs == d->Cout), "mmh_conv2d_dgrad_border: io16 needs a 16-bit dtype and dense dy");
    return do_dgrad_reflect1(d, dy, w, dx, ws, mmh::as_stream(s), false, phase, (io16 & 1) != 0, (io16 & 2) != 0);
}

size_t mmh_conv2d_dgrad_folded_ws_bytes(const mmh_conv_desc* d) {
    if (!d || d->pad_mode != MMH_PAD_REFLECT || d->pad == 0) return 0;
    if (d->pad == 1 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->H >= 3 && d->W >= 3)
        return reflect1_ws_bytes(d);
    return (size_t)d->B * (d->H + 2 * d->pad) * (d->W + 2 * d->pad) * d->Cin * sizeof(float);
}

int mmh_conv2d_dgrad_folded(const mmh_conv_desc* d, const void* dy, const void* w, void* dx, void* ws,
                            size_t ws_bytes, mmh_stream_t s) {
    if (int rc = validate(d)) return rc;
    MMH_REQUIRE(dy && w && dx, "mmh_conv2d_dgrad_folded: NULL buffer");
    MMH_REQUIRE(d->x_cs == d->Cin, "mmh_conv2d_dgrad_folded: dx must be dense (x_cs == Cin)");
    hipStream_t st = mmh::as_stream(s);
    if (d->pad_mode != MMH_PAD_REFLECT || d->pad == 0) return do_dgrad(d, dy, w, nullptr, dx, d->Cin, MMH_ACT_NONE, st);
    const size_t need = mmh_conv2d_dgrad_folded_ws_bytes(d);
    MMH_REQUIRE(ws && ws_bytes >= need, "mmh_conv2d_dgrad_folded: workspace too small (%zu < %zu)", ws_bytes, need);
    if (d->pad == 1 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->H >= 3 && d->W >= 3)
        return do_dgrad_reflect1(d, dy, w, dx, ws, st);
    if (int rc = do_dgrad(d, dy, w, nullptr, ws, d->Cin, MMH_ACT_NONE, st)) return rc;
    return mmh_reflect_fold(ws, dx, d->B, d->H, d->W, d->Cin, d->pad, s);
}

size_t mmh_conv2d_wgrad_ws_bytes(const mmh_conv_desc* d) { return d ? wgrad_ws(d) : 0; }

int mmh_conv2d_wgrad(const mmh_conv_desc* d, const void* x, const void* dy, void* dw, void* ws,
                     size_t ws_bytes, int accumulate, int io16, mmh_stream_t s) {
    if (int rc = validate(d)) return rc;
    MMH_REQUIRE(x && dy && dw && ws, "mmh_conv2d_wgrad: NULL buffer");
    return do_wgrad(d, x, dy, dw, ws, ws_bytes, accumulate, mmh::as_stream(s), (io16 & 1) != 0, (io16 & 2) != 0);
}

// ConvTranspose2d == dgrad of the stride-2 conv `d`; its input plays dy, its output plays dx.
int mmh_convT2d_fprop(const mmh_conv_desc* d, const void* x, const void* w, const void* bias,
                      void* y, int y_cs, int act, mmh_stream_t s) {
    if (int rc = validate(d)) return rc;
    MMH_REQUIRE(d->pad_mode == MMH_PAD_ZERO, "mmh_convT2d: zero padding only");
    MMH_REQUIRE(x && w && y, "mmh_convT2d_fprop: NULL buffer");
    return do_dgrad(d, x, w, bias, y, y_cs, act, mmh::as_stream(s));
}

int mmh_convT2d_dgrad(const mmh_conv_desc* d, const void* dy, const void* w, void* dx,
                      mmh_stream_t s) {
    if (int rc = validate(d)) return rc;
    MMH_REQUIRE(d->pad_mode == MMH_PAD_ZERO, "mmh_convT2d: zero padding only");
    MMH_REQUIRE(dy && w && dx, "mmh_convT2d_dgrad: NULL buffer");
    return do_fprop(d, dy, w, nullptr, dx, MMH_ACT_NONE, mmh::as_stream(s));
}

int mmh_convT2d_wgrad(const mmh_conv_desc* d, const void* x, const void* dy, void* dw, void* ws,
                      size_t ws_bytes, int accumulate, int io16, mmh_stream_t s) {
    if (int rc = validate(d)) return rc;
    MMH_REQUIRE(x && dy && dw && ws, "mmh_convT2d_wgrad: NULL buffer");
    // dw[tap][Cout_T][Cin_T]: the transposed conv's output-gradient is gathered like the
    // stride-2 conv's input, its input is the per-pixel operand.
    return do_wgrad(d, dy, x, dw, ws, ws_bytes, accumulate, mmh::as_stream(s), (io16 & 2) != 0, (io16 & 1) != 0);
}

int mmh_reflect_fold(const void* dxp, void* dx, int B, int H, int W, int C, int p,
                     mmh_stream_t s) {
    MMH_REQUIRE(dxp && dx, "mmh_reflect_fold: NULL buffer");
    MMH_REQUIRE(C % 4 == 0 && p >= 0 && p < H && p < W,
                "mmh_reflect_fold: bad shape C=%d p=%d H=%d W=%d", C, p, H, W);
    const int64_t total = (int64_t)B * H * W * (C / 4);
    int blocks = (int)std::min<int64_t>(mmh::cdiv(total, 256), 4096);
    hipLaunchKernelGGL(reflect_fold_kernel, dim3(blocks), dim3(256), 0, mmh::as_stream(s),
                       static_cast<const float*>(dxp), static_cast<float*>(dx), B, H, W, C / 4, p);
    return mmh::check_launch("reflect_fold_kernel");
}

}  // extern "C"
