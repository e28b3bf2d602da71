// HBM-bound kernels of the MM-HAND step for gfx950: Batch/InstanceNorm statistics and
// apply (+ReLU+Dropout+residual), their backward, the PATBlock gate, the GAN/L1 loss
// reductions, fused Adam, NCHW<->NHWC packing with channel concat, bias gradients and the
// pose-map kernels.  Everything is NHWC fp32, 16 bytes per lane, grid-stride; reductions
// are two-stage with a fixed summation order (no float atomics) so results are
// reproducible run to run.
#include <algorithm>
#include <cmath>
#include "common.h"

namespace mmh { int g_pw_v2 = 1; int g_col_chunks = 2048; int g_row_chunks = 4096; }
// mmh_set_dropout_salt: a device uint64 that every dropout-drawing kernel adds to its by-value seed when it runs (NULL = none).
// A captured training step (hipGraph) replays the seeds its launches were captured with; the salt, advanced by a kernel
// inside the graph (mmh_u64_add), is what makes each replay draw fresh masks.
static const uint64_t* g_dropout_salt = nullptr;   // mmh_set_option("pw_v2"): 0 = first-generation pointwise kernels (A/B)

namespace {

constexpr int TPB = 256;

__device__ __forceinline__ float4 ld4(const float* p, int64_t i4) {
    return reinterpret_cast<const float4*>(p)[i4];
}
__device__ __forceinline__ void st4(float* p, int64_t i4, float4 v) {
    reinterpret_cast<float4*>(p)[i4] = v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
// store 4 values as fp32 (lp = 0), bf16 (1) or fp16 (2); i4 indexes groups of 4 elements
typedef __bf16 pw_bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 pw_f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4_lp(void* p, int64_t i4, float4 v, int lp) {
    if (lp == 1) {
        pw_bf16x4 r;
        r[0] = (__bf16)v.x; r[1] = (__bf16)v.y; r[2] = (__bf16)v.z; r[3] = (__bf16)v.w;
        reinterpret_cast<pw_bf16x4*>(p)[i4] = r;
    } else if (lp == 2) {
        pw_f16x4 r;
        r[0] = (_Float16)v.x; r[1] = (_Float16)v.y; r[2] = (_Float16)v.z; r[3] = (_Float16)v.w;
        reinterpret_cast<pw_f16x4*>(p)[i4] = r;
    } else {
        reinterpret_cast<float4*>(p)[i4] = v;
    }
}

// load 4 values stored as fp32 (lp = 0), bf16 (1) or fp16 (2); i4 indexes groups of 4 elements
__device__ __forceinline__ float4 ld4_lp(const void* p, int64_t i4, int lp) {
    if (lp == 1) {
        const pw_bf16x4 r = reinterpret_cast<const pw_bf16x4*>(p)[i4];
        return make_float4((float)r[0], (float)r[1], (float)r[2], (float)r[3]);
    } else if (lp == 2) {
        const pw_f16x4 r = reinterpret_cast<const pw_f16x4*>(p)[i4];
        return make_float4((float)r[0], (float)r[1], (float)r[2], (float)r[3]);
    }
    return reinterpret_cast<const float4*>(p)[i4];
}

inline int grid_for(int64_t work_items, int cap = 4096) {
    return (int)std::max<int64_t>(1, std::min<int64_t>(mmh::cdiv(work_items, TPB), cap));
}

// Column-reduction launch geometry shared by stats / bwd-reduce / colsum.
struct ColGeom {
    int lpp;      // lanes (float4 groups) per row = C/4
    int rpi;      // rows handled per block iteration = TPB / lpp
    int chunks;   // blocks per group
    int64_t rows_per_chunk;
};
inline ColGeom col_geom(int groups, int64_t rows, int C) {
    ColGeom g;
    g.lpp = C / 4;
    g.rpi = TPB / g.lpp;
    // few groups (a small batch of large images) -> many chunks per group, and the one-work-group-row final pass walks them all:
    // half as many chunks below 16 groups (512x512 B=4 bf16: -0.23 ms per step; neutral at B=32: tools/ab_step.py opt:col_chunks)
    int64_t want = std::max<int64_t>(1, (groups < 16 ? mmh::g_col_chunks / 2 : mmh::g_col_chunks) / std::max(groups, 1));
    int64_t maxc = std::max<int64_t>(1, rows / ((int64_t)g.rpi * 8));
    g.chunks = (int)std::min<int64_t>(std::min(want, maxc), 1024);
    g.rows_per_chunk = mmh::cdiv(rows, g.chunks);
    return g;
}

// ------------------------------------------------------------------ norm statistics
// partial layout: ws[((grp*chunks + chunk)*3 + {0:n,1:mean,2:M2})*C + c]
__global__ void norm_stats_partial(const void* __restrict__ x, int xlp, int64_t rows, int C, int cs,
                                   ColGeom cg, float* __restrict__ ws) {
    __shared__ float sh[3][TPB * 4];
    const int tid = threadIdx.x;
    const int q = tid % cg.lpp, rsub = tid / cg.lpp;
    const int grp = blockIdx.y, chunk = blockIdx.x;
    const int64_t r0 = (int64_t)chunk * cg.rows_per_chunk;
    const int64_t r1 = min(rows, r0 + cg.rows_per_chunk);
    float4 K = make_float4(0, 0, 0, 0), s = K, ss = K;
    float n = 0.f;
    if (rsub < cg.rpi) {
        const int64_t base4 = ((int64_t)grp * rows * cs) / 4 + q;
        const int cs4 = cs / 4;
        for (int64_t r = r0 + rsub; r < r1; r += cg.rpi) {
            float4 v = ld4_lp(x, base4 + r * cs4, xlp);
            if (n == 0.f) K = v;
            float dx = v.x - K.x, dy = v.y - K.y, dz = v.z - K.z, dw = v.w - K.w;
            s.x += dx; s.y += dy; s.z += dz; s.w += dw;
            ss.x += dx * dx; ss.y += dy * dy; ss.z += dz * dz; ss.w += dw * dw;
            n += 1.f;
        }
    }
    // per-thread (n, mean, M2) for 4 channels
    float inv = n > 0.f ? 1.f / n : 0.f;
    float mean[4] = {K.x + s.x * inv, K.y + s.y * inv, K.z + s.z * inv, K.w + s.w * inv};
    float m2[4] = {ss.x - s.x * s.x * inv, ss.y - s.y * s.y * inv, ss.z - s.z * s.z * inv,
                   ss.w - s.w * s.w * inv};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        sh[0][tid * 4 + e] = n;
        sh[1][tid * 4 + e] = mean[e];
        sh[2][tid * 4 + e] = m2[e] > 0.f ? m2[e] : 0.f;
    }
    __syncthreads();
    if (rsub == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float na = sh[0][tid * 4 + e], ma = sh[1][tid * 4 + e], qa = sh[2][tid * 4 + e];
            for (int j = 1; j < cg.rpi; ++j) {
                const int t = (j * cg.lpp + q) * 4 + e;
                float nb = sh[0][t], mb = sh[1][t], qb = sh[2][t];
                if (nb > 0.f) {
                    float nt = na + nb, d = mb - ma;
                    ma += d * (nb / nt);
                    qa += qb + d * d * (na * nb / nt);
                    na = nt;
                }
            }
            const int64_t o = ((int64_t)(grp * cg.chunks + chunk) * 3) * C + q * 4 + e;
            ws[o] = na; ws[o + C] = ma; ws[o + 2 * C] = qa;
        }
    }
}

// scale != NULL: also the finalize step of a norm without affine parameters (InstanceNorm) - scale = invstd,
// shift = -mean * invstd, exactly as norm_finalize_kernel computes them from the fp32 mean / M2 written here
// cnt != NULL (first level of a two-level merge): the merged count goes to cnt and the three outputs of group g are
// written at g * ostride + c - with cnt, mean, m2 one C apart and ostride = 3 C that is a coarser partial array.
__global__ void norm_stats_final(const float* __restrict__ ws, int groups, int C, int chunks,
                                 float* __restrict__ mean, float* __restrict__ m2, double count = 0.0,
                                 float eps = 0.f, float* __restrict__ scale = nullptr,
                                 float* __restrict__ shift = nullptr, float* __restrict__ invstd = nullptr,
                                 float* __restrict__ cnt = nullptr, int ostride = 0, int64_t istride = 0,
                                 const float* __restrict__ gamma = nullptr, const float* __restrict__ beta = nullptr,
                                 float* __restrict__ rmean = nullptr, float* __restrict__ rvar = nullptr,
                                 float momentum = 0.f) {
    // istride != 0: chunk k of group g starts at (g * chunks + k) * istride instead of ... * 3 C (the SyncBN message of several
    // norm sites gathered as one row per rank: a site's triple sits at a column offset of every row)
    const int64_t cstride = istride ? istride : (int64_t)3 * C;
    // 32 channels x 8 chunk-lanes per block: each lane merges its chunks (Chan), then the 8
    // lanes are merged in a fixed order.
    __shared__ double sh[3][8][32];
    const int cl = threadIdx.x & 31, kl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + cl;
    const bool ok = i < groups * C;
    const int grp = ok ? i / C : 0, c = ok ? i - grp * C : 0;
    double na = 0, ma = 0, qa = 0;
    auto merge = [&](double nb, double mb, double qb) {
        if (nb > 0) {
            double nt = na + nb, d = mb - ma;
            ma += d * (nb / nt);
            qa += qb + d * d * (na * nb / nt);
            na = nt;
        }
    };
    if (ok) {
        // 8 partial triples per iteration: their loads do not depend on the running merge, so 24 are
        // in flight instead of a load -> divide -> load chain (121 tiles per plane after a Winograd
        // conv: two rounds per lane); the merge ORDER is unchanged (k ascending per lane)
        int k = kl;
        for (; k + 56 < chunks; k += 64) {
            float nb[8], mb[8], qb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t o = (int64_t)(grp * chunks + k + 8 * u) * cstride + c;
                nb[u] = ws[o]; mb[u] = ws[o + C]; qb[u] = ws[o + 2 * C];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) merge(nb[u], mb[u], qb[u]);
        }
        for (; k + 24 < chunks; k += 32) {
            float nb[4], mb[4], qb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t o = (int64_t)(grp * chunks + k + 8 * u) * cstride + c;
                nb[u] = ws[o]; mb[u] = ws[o + C]; qb[u] = ws[o + 2 * C];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) merge(nb[u], mb[u], qb[u]);
        }
        for (; k < chunks; k += 8) {
            const int64_t o = (int64_t)(grp * chunks + k) * cstride + c;
            merge(ws[o], ws[o + C], ws[o + 2 * C]);
        }
    }
    sh[0][kl][cl] = na; sh[1][kl][cl] = ma; sh[2][kl][cl] = qa;
    __syncthreads();
    if (kl == 0 && ok) {
        for (int j = 1; j < 8; ++j) {
            double nb = sh[0][j][cl], mb = sh[1][j][cl], qb = sh[2][j][cl];
            if (nb > 0) {
                double nt = na + nb, d = mb - ma;
                ma += d * (nb / nt);
                qa += qb + d * d * (na * nb / nt);
                na = nt;
            }
        }
        const float meanf = (float)ma, m2f = (float)qa;
        const int64_t oi = ostride ? (int64_t)grp * ostride + c : i;
        mean[oi] = meanf;
        m2[oi] = m2f;
        if (cnt) cnt[oi] = (float)na;
        if (scale) {        // norm_finalize_kernel's arithmetic, operation for operation
            const float var = (float)((double)m2f / count);
            const float is = 1.f / sqrtf(var + eps);
            const float gm = gamma ? gamma[c] : 1.f;
            const float bt = beta ? beta[c] : 0.f;
            const float sc = gm * is;
            scale[i] = sc;
            shift[i] = bt - meanf * sc;
            invstd[i] = is;
            if (rmean && i < C) {
                const float unb = count > 1.0 ? (float)((double)m2f / (count - 1.0)) : var;
                rmean[c] = (1.f - momentum) * rmean[c] + momentum * meanf;
                rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
            }
        }
    }
}

__global__ void norm_finalize_kernel(const float* __restrict__ mean, const float* __restrict__ m2,
                                     double count, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, float eps, int groups, int C,
                                     float* __restrict__ scale, float* __restrict__ shift,
                                     float* __restrict__ invstd, float* __restrict__ rmean,
                                     float* __restrict__ rvar, float momentum) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= groups * C) return;
    int c = i % C;
    float var = (float)((double)m2[i] / count);
    float is = 1.f / sqrtf(var + eps);
    float gm = gamma ? gamma[c] : 1.f;
    float bt = beta ? beta[c] : 0.f;
    float sc = gm * is;
    scale[i] = sc;
    shift[i] = bt - mean[i] * sc;
    invstd[i] = is;
    if (rmean && i < C) {
        float unb = count > 1.0 ? (float)((double)m2[i] / (count - 1.0)) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean[i];
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
    }
}

// ------------------------------------------------------------------ dropout hash
__device__ __forceinline__ uint32_t mix_u32(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}

// all 64 bits of the same finalizer: four 16-bit uniforms per hash (second-generation dropout)
__device__ __forceinline__ uint64_t mix_u64(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Dropout decisions of one site as a bit array (bit e of byte i = element 8 i + e is kept), drawn once
// per iteration and read by every kernel that needs them: the norm-apply fused into a Winograd input
// transform and the norm backward kernels that decide again instead of reading stored keep bits.
// mask (uint8 per element, test hook) replaces the hash.  p = 0.5 (the reference's nn.Dropout(0.5)): the
// 64 bits of one hash are 64 decisions (8 bytes); other p: 16-bit uniforms against the threshold.
__device__ __forceinline__ unsigned drop_byte(int64_t i, uint32_t thr16, uint64_t seed, const uint8_t* __restrict__ mask) {
    unsigned b = 0;
    if (mask) {
        const uint2 m = reinterpret_cast<const uint2*>(mask)[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) b |= (((e < 4 ? m.x >> (8 * e) : m.y >> (8 * (e - 4))) & 0xffu) ? 1u : 0u) << e;
    } else if (thr16 == 0x8000u) {
        b = (unsigned)(mix_u64(seed, (uint64_t)(i >> 3)) >> (8 * (i & 7))) & 0xffu;
    } else {
        const uint64_t h0 = mix_u64(seed, (uint64_t)i * 2), h1 = mix_u64(seed, (uint64_t)i * 2 + 1);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            b |= ((unsigned)((e < 4 ? h0 >> (16 * e) : h1 >> (16 * (e - 4))) & 0xffffu) >= thr16 ? 1u : 0u) << e;
    }
    return b;
}

__global__ void dropout_bits_kernel(int64_t n8, uint32_t thr16, uint64_t seed, const uint8_t* __restrict__ mask,
                                    uint8_t* __restrict__ bits, const uint64_t* __restrict__ salt) {
    if (salt) seed += *salt;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    bits[i] = (uint8_t)drop_byte(i, thr16, seed, mask);
}

// Both layouts in one launch: one workgroup per image row (b, h) draws the row's bytes (global + LDS) and
// transposes them into the row words (see dropout_rows_kernel below).
__global__ void __launch_bounds__(TPB) dropout_both_kernel(int W, int C, int nW32, uint32_t thr16, uint64_t seed,
                                                           const uint8_t* __restrict__ mask, uint8_t* __restrict__ bits,
                                                           uint32_t* __restrict__ rows, const uint64_t* __restrict__ salt) {
    extern __shared__ uint8_t sm_bits[];
    if (salt) seed += *salt;
    const int c8 = C / 8;
    const int64_t row = blockIdx.x;
    const int nb = W * c8;
    for (int i = threadIdx.x; i < nb; i += TPB) {
        const unsigned b = drop_byte(row * nb + i, thr16, seed, mask);
        bits[row * nb + i] = (uint8_t)b;
        sm_bits[i] = (uint8_t)b;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < nW32 * C; t += TPB) {
        const int c = t % C, j = t / C;
        const int nw = min(32, W - 32 * j);
        const uint8_t* src = sm_bits + 32 * j * c8 + (c >> 3);
        uint32_t word = 0;
        for (int k = 0; k < nw; ++k) word |= ((uint32_t)(src[k * c8] >> (c & 7)) & 1u) << k;
        rows[(row * nW32 + j) * C + c] = word;
    }
}

// The same decisions as ROW WORDS for the kernels that walk a channel along an image row (the Winograd
// transforms with the norm arithmetic inside, wino6.hip): rows[((b*H + h)*nW32 + j)*C + c] bit k = element
// (b, h, 32 j + k, c).  One thread per (image row, word, channel); its 32 source bytes are shared with 7 neighbours.
__global__ void dropout_rows_kernel(const uint8_t* __restrict__ bits, int64_t nrows, int W, int C, int nW32,
                                    uint32_t* __restrict__ rows) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * nW32 * C) return;
    const int c = (int)(i % C);
    const int64_t t = i / C;
    const int j = (int)(t % nW32);
    const int64_t row = t / nW32;
    const int c8 = C / 8;
    const uint8_t* src = bits + (row * W + 32 * j) * c8 + (c >> 3);
    const int nw = min(32, W - 32 * j);
    uint32_t word = 0;
#pragma unroll 8
    for (int k = 0; k < nw; ++k) word |= ((uint32_t)(src[(int64_t)k * c8] >> (c & 7)) & 1u) << k;
    rows[i] = word;
}

__global__ void scale_shift_act_kernel(const void* __restrict__ x, int in_lp, const float* __restrict__ scale,
                                       const float* __restrict__ shift,
                                       const float* __restrict__ residual, void* __restrict__ out,
                                       int64_t n4, int64_t rows_per_group, int C4, int relu,
                                       float drop_p, uint64_t seed,
                                       const uint8_t* __restrict__ mask, uint8_t* __restrict__ keep_bits,
                                       int out_lp, const uint64_t* __restrict__ salt) {
    if (salt) seed += *salt;
    const uint32_t thr = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u;
    const float dsc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        int64_t row = i / C4;
        int c4 = (int)(i - row * C4);
        int64_t grp = row / rows_per_group;
        float4 v = ld4_lp(x, i, in_lp);
        float4 sc = ld4(scale, grp * C4 + c4), sf = ld4(shift, grp * C4 + c4);
        float r[4] = {v.x * sc.x + sf.x, v.y * sc.y + sf.y, v.z * sc.z + sf.z, v.w * sc.w + sf.w};
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = r[e] > 0.f ? r[e] : 0.f;
        }
        if (drop_p > 0.f) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bool keep = mask ? (mask[i * 4 + e] != 0) : (mix_u32(seed, (uint64_t)(i * 4 + e)) >= thr);
                r[e] = keep ? r[e] * dsc : 0.f;
            }
        }
        if (keep_bits)      // what the backward needs of `out`: which lanes survived ReLU / dropout
            keep_bits[i] = (uint8_t)((r[0] > 0.f) | ((r[1] > 0.f) << 1) | ((r[2] > 0.f) << 2) | ((r[3] > 0.f) << 3));
        if (residual) {
            float4 q = ld4(residual, i);
            r[0] += q.x; r[1] += q.y; r[2] += q.z; r[3] += q.w;
        }
        st4_lp(out, i, make_float4(r[0], r[1], r[2], r[3]), out_lp);
    }
}

// ------------------------------------------------------------------ second-generation row kernels
// The first-generation kernels above keep one 16-byte (fp32) or 8-byte (16-bit) load per lane in
// flight: ~16-32 KB per CU, against the ~50 KB per CU that 6 TB/s x the loaded HBM latency needs
// (Little's law) - they measured 2.8-4.8 TB/s.  These handle 8 channels per lane (16 B of a 16-bit
// tensor, 32 B of fp32) and issue the loads of UNR rows before using any of them; the thread's
// column group is fixed (no per-element division, scale / shift / mean / invstd live in registers).
// Geometry: block = 256 threads = (256 / C8) rows x C8 column groups, C8 = C / 8 a power of two <= 256;
// grid = (row chunks, groups).
constexpr int UNR = 4;
struct f8 { float v[8]; };
typedef __bf16 pw_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pw_f16x8 __attribute__((ext_vector_type(8)));

template <bool W16>
__device__ __forceinline__ f8 ld8(const void* p, int64_t i8, bool h16) {
    f8 r;
    if (W16) {
        if (h16) {
            const pw_f16x8 t = reinterpret_cast<const pw_f16x8*>(p)[i8];
#pragma unroll
            for (int e = 0; e < 8; ++e) r.v[e] = (float)t[e];
        } else {
            const uint4 t = reinterpret_cast<const uint4*>(p)[i8];
            const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                r.v[2 * e] = __uint_as_float(w[e] << 16);
                r.v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
            }
        }
    } else {
        const float4 a = reinterpret_cast<const float4*>(p)[2 * i8], b = reinterpret_cast<const float4*>(p)[2 * i8 + 1];
        r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
        r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    }
    return r;
}
template <bool W16>
__device__ __forceinline__ void st8(void* p, int64_t i8, const f8& r, bool h16) {
    if (W16) {
        if (h16) {
            pw_f16x8 t;
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = (_Float16)r.v[e];
            reinterpret_cast<pw_f16x8*>(p)[i8] = t;
        } else {
            pw_bf16x8 t;
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = (__bf16)r.v[e];
            reinterpret_cast<pw_bf16x8*>(p)[i8] = t;
        }
    } else {
        reinterpret_cast<float4*>(p)[2 * i8] = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
        reinterpret_cast<float4*>(p)[2 * i8 + 1] = make_float4(r.v[4], r.v[5], r.v[6], r.v[7]);
    }
}

// a lane's 8 channels as loaded (16 or 32 bytes): kept packed until used, so that UNR rows in flight
// cost 4 (16-bit) or 8 (fp32) registers each and the kernels keep 6-8 waves per SIMD
template <bool W16> struct Raw8;
template <> struct Raw8<true> { uint4 a; };
template <> struct Raw8<false> { float4 a, b; };
__device__ __forceinline__ void ldraw(Raw8<true>& r, const void* p, int64_t i8) {
    r.a = reinterpret_cast<const uint4*>(p)[i8];
}
__device__ __forceinline__ void ldraw(Raw8<false>& r, const void* p, int64_t i8) {
    r.a = reinterpret_cast<const float4*>(p)[2 * i8];
    r.b = reinterpret_cast<const float4*>(p)[2 * i8 + 1];
}
__device__ __forceinline__ f8 widen(const Raw8<true>& t, bool h16) {
    f8 r;
    const unsigned w[4] = {t.a.x, t.a.y, t.a.z, t.a.w};
    if (h16) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const h2 v = __builtin_bit_cast(h2, w[e]);
            r.v[2 * e] = (float)v[0];
            r.v[2 * e + 1] = (float)v[1];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            r.v[2 * e] = __uint_as_float(w[e] << 16);
            r.v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
        }
    }
    return r;
}
__device__ __forceinline__ f8 widen(const Raw8<false>& t, bool) {
    f8 r;
    r.v[0] = t.a.x; r.v[1] = t.a.y; r.v[2] = t.a.z; r.v[3] = t.a.w;
    r.v[4] = t.b.x; r.v[5] = t.b.y; r.v[6] = t.b.z; r.v[7] = t.b.w;
    return r;
}

__device__ __forceinline__ void opaque(Raw8<true>& r) {
    asm volatile("" : "+v"(r.a.x), "+v"(r.a.y), "+v"(r.a.z), "+v"(r.a.w));
}
__device__ __forceinline__ void opaque(Raw8<false>& r) {
    asm volatile("" : "+v"(r.a.x), "+v"(r.a.y), "+v"(r.a.z), "+v"(r.a.w), "+v"(r.b.x), "+v"(r.b.y), "+v"(r.b.z), "+v"(r.b.w));
}

struct RowGeom {
    int c8, rpi, chunks;
    int64_t rows_per_chunk;
};
inline bool row_geom_ok(int C) {
    const int c8 = C / 8;
    return C % 8 == 0 && c8 >= 1 && c8 <= TPB && (c8 & (c8 - 1)) == 0;
}
inline RowGeom row_geom(int groups, int64_t rows, int C) {
    RowGeom g;
    g.c8 = C / 8;
    g.rpi = TPB / g.c8;
    const int64_t step = (int64_t)g.rpi * UNR;                  // rows one block covers per iteration
    const int64_t want = std::max<int64_t>(1, mmh::g_row_chunks / std::max(groups, 1));
    const int64_t maxc = std::max<int64_t>(1, rows / (2 * step));
    g.chunks = (int)std::min(want, maxc);
    g.rows_per_chunk = mmh::cdiv(mmh::cdiv(rows, g.chunks), step) * step;
    g.chunks = (int)mmh::cdiv(rows, g.rows_per_chunk);
    return g;
}

template <bool XW, bool OW, bool EXTRA>      // EXTRA: residual and / or the test mask present
__global__ void __launch_bounds__(TPB) scale_shift_act_v2(
        const void* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
        const float* __restrict__ residual, void* __restrict__ out, int64_t rows, RowGeom rg, int relu,
        float drop_p, uint64_t seed, const uint8_t* __restrict__ mask, uint8_t* __restrict__ keep_bits,
        bool xh16, bool oh16, void* __restrict__ twin, bool th16, const uint64_t* __restrict__ salt) {
    if (salt) seed += *salt;
    const int q = threadIdx.x & (rg.c8 - 1), rsub = threadIdx.x / rg.c8;
    const int grp = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * rg.rows_per_chunk;
    const int64_t r1 = min(rows, r0 + rg.rows_per_chunk);
    const f8 sc = ld8<false>(scale, (int64_t)grp * rg.c8 + q, false);
    const f8 sf = ld8<false>(shift, (int64_t)grp * rg.c8 + q, false);
    // dropout: keep iff a 16-bit uniform >= thr16 (p = 0.5 exactly; other p to 1/65536); two hashes
    // of (seed, group-of-8 index) give the eight uniforms of a lane's channels
    const uint32_t thr16 = drop_p > 0.f ? (uint32_t)((double)drop_p * 65536.0) : 0u;
    const float dsc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const int64_t gbase = (int64_t)grp * rows;
    for (int64_t r = r0 + rsub; r < r1; r += (int64_t)rg.rpi * UNR) {
        Raw8<XW> xr[UNR];
        Raw8<false> rv[EXTRA ? UNR : 1];
        uint2 mk[EXTRA ? UNR : 1];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t rr = r + (int64_t)u * rg.rpi;
            if (rr < r1) {
                const int64_t i8 = (gbase + rr) * rg.c8 + q;
                ldraw(xr[u], x, i8);
                if (EXTRA && residual) ldraw(rv[u], residual, i8);
                if (EXTRA && mask) mk[u] = reinterpret_cast<const uint2*>(mask)[i8];
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t rr = r + (int64_t)u * rg.rpi;
            if (rr >= r1) continue;
            const int64_t i8 = (gbase + rr) * rg.c8 + q;
            const f8 xv = widen(xr[u], xh16);
            f8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = __builtin_fmaf(xv.v[e], sc.v[e], sf.v[e]);
                if (relu) t = t > 0.f ? t : 0.f;
                o.v[e] = t;
            }
            if (drop_p > 0.f) {
                uint64_t h0 = 0, h1 = 0;
                if (!(EXTRA && mask)) {
                    h0 = mix_u64(seed, (uint64_t)i8 * 2);
                    h1 = mix_u64(seed, (uint64_t)i8 * 2 + 1);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint2 m2 = mk[EXTRA ? u : 0];
                    const unsigned mb = e < 4 ? (m2.x >> (8 * e)) & 0xffu : (m2.y >> (8 * (e - 4))) & 0xffu;
                    const unsigned uf = (unsigned)((e < 4 ? h0 >> (16 * e) : h1 >> (16 * (e - 4))) & 0xffffu);
                    const bool keep = (EXTRA && mask) ? (mb != 0) : (uf >= thr16);
                    o.v[e] = keep ? o.v[e] * dsc : 0.f;
                }
            }
            if (keep_bits) {
                unsigned lo = 0, hi = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo |= (o.v[e] > 0.f ? 1u : 0u) << e;
                    hi |= (o.v[4 + e] > 0.f ? 1u : 0u) << e;
                }
                reinterpret_cast<uint16_t*>(keep_bits)[i8] = (uint16_t)(lo | (hi << 8));
            }
            if (EXTRA && residual) {
                const f8 q8 = widen(rv[EXTRA ? u : 0], false);
#pragma unroll
                for (int e = 0; e < 8; ++e) o.v[e] += q8.v[e];
            }
            st8<OW>(out, i8, o, oh16);
            if (twin) st8<true>(twin, i8, o, th16);     // the same values once more in 16 bits (the next conv's operand)
        }
    }
}

// norm statistics, second generation: same chunks and partial layout as norm_stats_partial
// (ws[((grp*chunks + chunk)*3 + {n,mean,M2})*C + c]); block = (256/C8 rows) x (C8 groups of 8 channels)
template <bool XW>
__global__ void __launch_bounds__(TPB) norm_stats_partial_v2(const void* __restrict__ x, bool xh16, int64_t rows,
                                                             int C, int c8, int rpi, int chunks,
                                                             int64_t rows_per_chunk, float* __restrict__ ws) {
    __shared__ float sh[3][TPB * 8];
    const int tid = threadIdx.x;
    const int q = tid & (c8 - 1), rsub = tid / c8;
    const int grp = blockIdx.y, chunk = blockIdx.x;
    const int64_t r0 = (int64_t)chunk * rows_per_chunk;
    const int64_t r1 = min(rows, r0 + rows_per_chunk);
    const int64_t gbase = (int64_t)grp * rows;
    float K[8], sm[8], ss[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { K[e] = 0.f; sm[e] = 0.f; ss[e] = 0.f; }
    float n = 0.f;
    for (int64_t r = r0 + rsub; r < r1; r += (int64_t)rpi * UNR) {
        Raw8<XW> xr[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t rr = r + (int64_t)u * rpi;
            if (rr < r1) ldraw(xr[u], x, (gbase + rr) * c8 + q);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (r + (int64_t)u * rpi >= r1) continue;
            const f8 xv = widen(xr[u], xh16);
            if (n == 0.f) {
#pragma unroll
                for (int e = 0; e < 8; ++e) K[e] = xv.v[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = xv.v[e] - K[e];
                sm[e] += d;
                ss[e] += d * d;
            }
            n += 1.f;
        }
    }
    const float inv = n > 0.f ? 1.f / n : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float m2 = ss[e] - sm[e] * sm[e] * inv;
        sh[0][tid * 8 + e] = n;
        sh[1][tid * 8 + e] = K[e] + sm[e] * inv;
        sh[2][tid * 8 + e] = m2 > 0.f ? m2 : 0.f;
    }
    __syncthreads();
    // the first 8*c8 threads each finish one channel: Chan merge over the rpi row lanes, fixed order
    for (int t = tid; t < 8 * c8; t += TPB) {
        const int qq = t >> 3, e = t & 7;
        float na = sh[0][qq * 8 + e], ma = sh[1][qq * 8 + e], qa = sh[2][qq * 8 + e];
        for (int j = 1; j < rpi; ++j) {
            const int o = (j * c8 + qq) * 8 + e;
            const float nb = sh[0][o], mb = sh[1][o], qb = sh[2][o];
            if (nb > 0.f) {
                const float nt = na + nb, d = mb - ma;
                ma += d * (nb / nt);
                qa += qb + d * d * (na * nb / nt);
                na = nt;
            }
        }
        const int64_t o = ((int64_t)(grp * chunks + chunk) * 3) * C + qq * 8 + e;
        ws[o] = na; ws[o + C] = ma; ws[o + 2 * C] = qa;
    }
}

// column reductions, second generation (MODE as in col_reduce_partial below; same partial layout)
// RC (MODE 1, fp32): no keep bits were stored (the norm's apply pass ran inside the consuming conv's
// input transform, wino6.hip): the lane decides again - kept by the dropout bit array `bits` (1 bit per
// element, NULL = no dropout) and fma(x, scale, shift) > 0, the very expression the forward evaluated
template <int MODE, bool AW, bool XW, bool RC = false>
__global__ void __launch_bounds__(TPB) col_reduce_partial_v2(
        const void* __restrict__ a, bool ah16, const uint8_t* __restrict__ bits, const void* __restrict__ x,
        bool xh16, const float* __restrict__ mean, const float* __restrict__ invstd, int64_t rows, int C, int c8,
        int rpi, int chunks, int64_t rows_per_chunk, int masked, float dsc, float* __restrict__ ws,
        const float* __restrict__ scale = nullptr, const float* __restrict__ shift = nullptr, int relu = 0) {
    constexpr int NOUT = MODE == 0 ? 1 : 2;
    __shared__ float sh[NOUT][TPB * 8];
    const int tid = threadIdx.x;
    const int q = tid & (c8 - 1), rsub = tid / c8;
    const int grp = blockIdx.y, chunk = blockIdx.x;
    const int64_t r0 = (int64_t)chunk * rows_per_chunk;
    const int64_t r1 = min(rows, r0 + rows_per_chunk);
    const int64_t gbase = (int64_t)grp * rows;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    f8 mu, is, sc, sf;
    if (MODE == 1) {
        mu = ld8<false>(mean, (int64_t)grp * c8 + q, false);
        is = ld8<false>(invstd, (int64_t)grp * c8 + q, false);
        if (RC) {
            sc = ld8<false>(scale, (int64_t)grp * c8 + q, false);
            sf = ld8<false>(shift, (int64_t)grp * c8 + q, false);
        }
    }
    for (int64_t r = r0 + rsub; r < r1; r += (int64_t)rpi * UNR) {
        Raw8<AW> gr[UNR];
        Raw8<XW> xr[MODE == 1 ? UNR : 1];
        unsigned kb[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t rr = r + (int64_t)u * rpi;
            if (rr < r1) {
                const int64_t i8 = (gbase + rr) * c8 + q;
                ldraw(gr[u], a, i8);
                if (MODE == 1) {
                    ldraw(xr[u], x, i8);
                    if (RC) kb[u] = bits ? bits[i8] : 0xffu;
                    else if (masked) kb[u] = reinterpret_cast<const uint16_t*>(bits)[i8];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (r + (int64_t)u * rpi >= r1) continue;
            const f8 gv = widen(gr[u], ah16);
            f8 xv;
            if (MODE == 1) xv = widen(xr[MODE == 1 ? u : 0], xh16);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float gg = gv.v[e];
                if (MODE == 1) {
                    if (RC) {
                        const bool keep = ((kb[u] >> e) & 1u) && (!relu || __builtin_fmaf(xv.v[e], sc.v[e], sf.v[e]) > 0.f);
                        gg = keep ? gg * dsc : 0.f;
                    } else if (masked) gg = (kb[u] >> (e < 4 ? e : e + 4)) & 1u ? gg * dsc : 0.f;
                    s2[e] += gg * ((xv.v[e] - mu.v[e]) * is.v[e]);
                }
                s1[e] += gg;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sh[0][tid * 8 + e] = s1[e];
        if (NOUT == 2) sh[NOUT - 1][tid * 8 + e] = s2[e];
    }
    __syncthreads();
    for (int t = tid; t < 8 * c8; t += TPB) {
        const int qq = t >> 3, e = t & 7;
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            float acc = sh[o][qq * 8 + e];
            for (int j = 1; j < rpi; ++j) acc += sh[o][(j * c8 + qq) * 8 + e];
            ws[((int64_t)(grp * chunks + chunk) * NOUT + o) * C + qq * 8 + e] = acc;
        }
    }
}

template <bool GW, bool XW, bool DW, bool RC = false>      // RC: as in col_reduce_partial_v2
__global__ void __launch_bounds__(TPB) norm_bwd_apply_v2(
        const void* __restrict__ g, bool gh16, const uint8_t* __restrict__ bits, const void* __restrict__ x,
        bool xh16, const float* __restrict__ mean, const float* __restrict__ invstd,
        const float* __restrict__ gamma, const float* __restrict__ s1, const float* __restrict__ s2,
        float inv_count, int64_t rows, RowGeom rg, int masked, float dsc, void* __restrict__ dx, bool dh16,
        const float* __restrict__ scale = nullptr, const float* __restrict__ shift = nullptr, int relu = 0) {
    const int q = threadIdx.x & (rg.c8 - 1), rsub = threadIdx.x / rg.c8;
    const int grp = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * rg.rows_per_chunk;
    const int64_t r1 = min(rows, r0 + rg.rows_per_chunk);
    const int64_t gi = (int64_t)grp * rg.c8 + q;
    const f8 mu = ld8<false>(mean, gi, false), is = ld8<false>(invstd, gi, false);
    const f8 a1 = ld8<false>(s1, gi, false), a2 = ld8<false>(s2, gi, false);
    f8 sc, sf;
    if (RC) { sc = ld8<false>(scale, gi, false); sf = ld8<false>(shift, gi, false); }
    float k0[8], k1[8], k2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float gm = gamma ? gamma[q * 8 + e] : 1.f;
        k0[e] = gm * is.v[e];
        k1[e] = a1.v[e] * inv_count;
        k2[e] = k0[e] * is.v[e] * (a2.v[e] * inv_count);
    }
    const int64_t gbase = (int64_t)grp * rows;
    for (int64_t r = r0 + rsub; r < r1; r += (int64_t)rg.rpi * UNR) {
        Raw8<GW> gr[UNR];
        Raw8<XW> xr[UNR];
        unsigned kb[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t rr = r + (int64_t)u * rg.rpi;
            if (rr < r1) {
                const int64_t i8 = (gbase + rr) * rg.c8 + q;
                ldraw(gr[u], g, i8);
                ldraw(xr[u], x, i8);
                if (RC) kb[u] = bits ? bits[i8] : 0xffu;
                else if (masked) kb[u] = reinterpret_cast<const uint16_t*>(bits)[i8];
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t rr = r + (int64_t)u * rg.rpi;
            if (rr >= r1) continue;
            const f8 gv = widen(gr[u], gh16), xv = widen(xr[u], xh16);
            f8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                // dx = k0*(dz - s1/count) - xhat*k0*s2/count, xhat = (x-mu)*invstd
                if (RC) {
                    const bool keep = ((kb[u] >> e) & 1u) && (!relu || __builtin_fmaf(xv.v[e], sc.v[e], sf.v[e]) > 0.f);
                    o.v[e] = mmh::norm_bwd_elem(gv.v[e], keep, dsc, xv.v[e], mu.v[e], k0[e], k1[e], k2[e]);
                } else {
                    const bool keep = !masked || ((kb[u] >> (e < 4 ? e : e + 4)) & 1u);
                    o.v[e] = mmh::norm_bwd_elem(gv.v[e], keep, masked ? dsc : 1.f, xv.v[e], mu.v[e], k0[e], k1[e], k2[e]);
                }
            }
            st8<DW>(dx, (gbase + rr) * rg.c8 + q, o, dh16);
        }
    }
}

// ------------------------------------------------------------------ norm backward in ONE pass (small planes)
// The two-pass backward (col_reduce_partial_v2<1> for s1 = sum dz, s2 = sum dz * xhat, then norm_bwd_apply_v2) reads g, x
// and the keep bits twice.  When a (group, channel) plane is small enough - InstanceNorm at 64x64: 4096 rows - a workgroup of
// 1024 threads holds the whole plane of 8 * LPR channels on chip - g and the bits in registers, x in LDS (brought by LDS-DMA:
// with both tensors in registers the kernel spilled at the 128-register limit of 1024 threads) - it loads everything ONCE, sums,
// exchanges the sums through LDS and applies: the reduce pass (its 4.25 B per element and its launch) is gone.
// Geometry: LPR = 2^lpr_log2 adjacent lanes share a row (lane q of them owns channels [8 (LPR blockIdx.x + q), +8): one wave
// instruction covers 64 / LPR rows x 16 LPR bytes, so with LPR = 2 every 128-byte line is pulled into 4 CUs' L1 instead of 8);
// a thread owns rows rs, rs + 1024 / LPR, ... - at most NR of them, rows * LPR <= 1024 * NR.  The sums are the two-pass
// kernels' terms (dz = keep ? g * dsc : 0; xhat = (x - mu) * invstd) in a different, fixed order (64-lane butterflies, then
// the 16 waves in order); the apply is the same expression regrouped around three coefficients.  Deterministic; equal to the
// two-pass result to rounding, not bit for bit.
// exactly the two-pass kernels' (dz = keep ? g * dsc : 0; mmh::norm_bwd_elem); only the ORDER of the plane sums differs
// (64-lane butterflies, then the 16 waves in order): deterministic, not bit-identical to the two-pass sums.
constexpr int FT = 1024;
// H16: the 16-bit tensors are IEEE fp16 (else bf16); MASKED: keep bits present - template parameters (run-time branches
// around the two widening paths and around each row's keep-bit load cost registers and serialised the loads).
// Register budget (1024 threads: 128): the resident plane is 36 (16-bit g) and everything per channel - sums, mean, invstd,
// three apply coefficients, the widened row - is held for FOUR channels at a time: both passes run twice over the resident
// rows, once per channel half (with all eight the kernel needed ~135 registers and spilled 60-110 dwords per lane).
template <bool H16>
__device__ __forceinline__ void widen2(unsigned w0, unsigned w1, float* o) {
    if (H16) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 a = __builtin_bit_cast(h2, w0), b = __builtin_bit_cast(h2, w1);
        o[0] = (float)a[0]; o[1] = (float)a[1]; o[2] = (float)b[0]; o[3] = (float)b[1];
    } else {
        o[0] = __uint_as_float(w0 << 16); o[1] = __uint_as_float(w0 & 0xffff0000u);
        o[2] = __uint_as_float(w1 << 16); o[3] = __uint_as_float(w1 & 0xffff0000u);
    }
}
// channels [4 h, 4 h + 4) of a lane's 8-channel unit
template <bool H16, int H> __device__ __forceinline__ void half4(const Raw8<true>& r, float* o) {
    if (H == 0) widen2<H16>(r.a.x, r.a.y, o); else widen2<H16>(r.a.z, r.a.w, o);
}
template <bool H16, int H> __device__ __forceinline__ void half4(const Raw8<false>& r, float* o) {
    const float4 v = H == 0 ? r.a : r.b;
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
template <bool H16> __device__ __forceinline__ uint2 pack4(const float* v) {
    uint2 r;
    if (H16) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 a, b;
        a[0] = (_Float16)v[0]; a[1] = (_Float16)v[1]; b[0] = (_Float16)v[2]; b[1] = (_Float16)v[3];
        r.x = __builtin_bit_cast(unsigned, a); r.y = __builtin_bit_cast(unsigned, b);
    } else {
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        b2 a, b;
        a[0] = (__bf16)v[0]; a[1] = (__bf16)v[1]; b[0] = (__bf16)v[2]; b[1] = (__bf16)v[3];
        r.x = __builtin_bit_cast(unsigned, a); r.y = __builtin_bit_cast(unsigned, b);
    }
    return r;
}

template <bool GW, bool DW, int NR, bool H16, bool MASKED>
struct PlaneBody {
    Raw8<GW> gr[NR];
    unsigned kb[(NR + 1) / 2];      // two rows' 16 keep bits per register
    const uint2* xs;                // this lane's first unit of the x plane in LDS, as 8-byte halves
    int rs, nslots, rows;

    // sums of channels [4 H, 4 H + 4) over this lane's rows
    template <int H> __device__ __forceinline__ void sums(const float* mu, const float* is, float dsc, float* s1, float* s2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const bool live = rs + u * nslots < rows;
            const uint2 xr = xs[2 * u * FT + H];
            float gv[4], xv[4];
            half4<H16, H>(gr[u], gv);
            widen2<H16>(xr.x, xr.y, xv);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float gg = gv[e];
                if (MASKED) gg = (kb[u / 2] >> ((u & 1) * 16 + 8 * H + e)) & 1u ? gg * dsc : 0.f;
                gg = live ? gg : 0.f;
                s2[e] += gg * ((xv[e] - mu[e]) * is[e]);
                s1[e] += gg;
            }
        }
    }
    // dx of channels [4 H, 4 H + 4): fma(k0, dz, fma(-k2, x, c))
    template <int H> __device__ __forceinline__ void apply(const float* k0, const float* k2, const float* cc, float dsm,
                                                            char* dp, unsigned unit0, unsigned ustride) {
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int rr = rs + u * nslots;
            const uint2 xr = xs[2 * u * FT + H];
            float gv[4], xv[4], o[4];
            half4<H16, H>(gr[u], gv);
            widen2<H16>(xr.x, xr.y, xv);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool keep = !MASKED || ((kb[u / 2] >> ((u & 1) * 16 + 8 * H + e)) & 1u);
                const float dz = keep ? gv[e] * dsm : 0.f;
                o[e] = __builtin_fmaf(k0[e], dz, __builtin_fmaf(-k2[e], xv[e], cc[e]));
            }
            if (rr < rows) {
                const unsigned unit = unit0 + (unsigned)u * ustride;
                if (DW) reinterpret_cast<uint2*>(dp)[2 * unit + H] = pack4<H16>(o);
                else reinterpret_cast<float4*>(dp)[2 * unit + H] = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    }
};

template <bool GW, bool DW, int NR, bool H16, bool MASKED>
__global__ void __launch_bounds__(FT) norm_bwd_plane_kernel(
        const void* __restrict__ g, const uint8_t* __restrict__ bits, const void* __restrict__ x,
        const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma, float inv_count,
        int rows, int c8, int lpr_log2, float dsc, void* __restrict__ dx,
        float* __restrict__ s1_out, float* __restrict__ s2_out) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char plane_x[];     // [NR][FT] 16-byte units: the x plane
    __shared__ float shw[FT / 64][8][16];
    __shared__ float sht[8][16];
    const int tid = threadIdx.x, lpr = 1 << lpr_log2;
    const int q = tid & (lpr - 1), rs = tid >> lpr_log2, nslots = FT >> lpr_log2;
    const int grp = blockIdx.y, cq = blockIdx.x * lpr + q;
    // the group's base as a pointer, everything below it as 32-bit indices of 8-channel units; rows past the plane's
    // end re-read its last row and contribute nothing
    const int64_t gb8 = (int64_t)grp * rows * c8;
    const char* gp = static_cast<const char*>(g) + gb8 * (GW ? 16 : 32);
    const char* xp = static_cast<const char*>(x) + gb8 * 16;
    const uint16_t* bp = reinterpret_cast<const uint16_t*>(bits) + (MASKED ? gb8 : 0);
    char* dp = static_cast<char*>(dx) + gb8 * (DW ? 16 : 32);
    // x: global -> LDS by DMA, one 16-byte unit per lane and row, unit (u, tid) at plane_x + 16 (u FT + tid): every lane
    // reads back exactly what its own DMA brought, so its own vmcnt wait is all the ordering there is (no barrier)
    const unsigned xl = mmh::lds_addr_of(plane_x) + (unsigned)__builtin_amdgcn_readfirstlane(tid & ~63) * 16u;
    const unsigned unit0 = (unsigned)(rs * c8 + cq), ustride = (unsigned)(nslots * c8);
    const bool tail = (NR - 1) * nslots + rs >= rows;       // some of this lane's rows lie past the plane
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        const unsigned unit = tail ? (unsigned)(min(rs + u * nslots, rows - 1) * c8 + cq) : unit0 + (unsigned)u * ustride;
        mmh::lds_dma16(xp + (size_t)unit * 16, xl + (unsigned)u * (FT * 16u));
    }
    PlaneBody<GW, DW, NR, H16, MASKED> pb;
    pb.rs = rs; pb.nslots = nslots; pb.rows = rows;
    unsigned b16[MASKED ? NR : 1];
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        const unsigned unit = tail ? (unsigned)(min(rs + u * nslots, rows - 1) * c8 + cq) : unit0 + (unsigned)u * ustride;
        ldraw(pb.gr[u], gp, unit);
        if (MASKED) b16[u] = bp[unit];
    }
    const int64_t gi = ((int64_t)grp * c8 + cq) * 2;       // in float4 units
    const float4* mean4 = reinterpret_cast<const float4*>(mean) + gi;     // re-read per channel half (L2 hits): 16 registers
    const float4* istd4 = reinterpret_cast<const float4*>(invstd) + gi;   // less than holding both halves through both passes
    float4 mu0 = mean4[0], is0 = istd4[0], mu1, is1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < (NR + 1) / 2; ++u)
        pb.kb[u] = MASKED ? (b16[MASKED ? 2 * u : 0] | (b16[MASKED ? 2 * u + 1 : 0] << 16)) : 0xffffffffu;
    pb.xs = reinterpret_cast<const uint2*>(plane_x) + 2 * tid;
    const int wave = tid >> 6, lane = tid & 63;
    {
        float s1[4], s2[4];
        const float mu[4] = {mu0.x, mu0.y, mu0.z, mu0.w}, is[4] = {is0.x, is0.y, is0.z, is0.w};
        pb.template sums<0>(mu, is, dsc, s1, s2);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            for (int off = lpr; off < 64; off <<= 1) { s1[e] += __shfl_xor(s1[e], off, 64); s2[e] += __shfl_xor(s2[e], off, 64); }
        if (lane < lpr) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { shw[wave][lane][e] = s1[e]; shw[wave][lane][8 + e] = s2[e]; }
        }
    }
    {
        float s1[4], s2[4];
        mu1 = mean4[1]; is1 = istd4[1];
        const float mu[4] = {mu1.x, mu1.y, mu1.z, mu1.w}, is[4] = {is1.x, is1.y, is1.z, is1.w};
        pb.template sums<1>(mu, is, dsc, s1, s2);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            for (int off = lpr; off < 64; off <<= 1) { s1[e] += __shfl_xor(s1[e], off, 64); s2[e] += __shfl_xor(s2[e], off, 64); }
        if (lane < lpr) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { shw[wave][lane][4 + e] = s1[e]; shw[wave][lane][12 + e] = s2[e]; }
        }
    }
    __syncthreads();
    if (tid < lpr * 16) {       // the 16 waves in order: fixed summation order
        const int qq = tid >> 4, k = tid & 15;
        float acc = shw[0][qq][k];
        for (int w = 1; w < FT / 64; ++w) acc += shw[w][qq][k];
        sht[qq][k] = acc;
        const int64_t o = ((int64_t)grp * c8 + blockIdx.x * lpr + qq) * 8 + (k & 7);
        (k < 8 ? s1_out : s2_out)[o] = acc;
    }
    __syncthreads();
    // dx = k0 (dz - s1/n) - (x - mu) k2  =  fma(k0, dz, fma(-k2, x, c)),  c = mu k2 - k0 s1/n
    const float dsm = MASKED ? dsc : 1.f;
    {
        mu0 = mean4[0]; is0 = istd4[0];
        asm volatile("" : "+v"(mu0.x), "+v"(is0.x));       // a fresh load, not the first pass's registers kept alive
        const float mu[4] = {mu0.x, mu0.y, mu0.z, mu0.w}, is[4] = {is0.x, is0.y, is0.z, is0.w};
        float k0[4], k2[4], cc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            k0[e] = (gamma ? gamma[cq * 8 + e] : 1.f) * is[e];
            k2[e] = k0[e] * is[e] * (sht[q][8 + e] * inv_count);
            cc[e] = __builtin_fmaf(mu[e], k2[e], -(k0[e] * (sht[q][e] * inv_count)));
        }
        pb.template apply<0>(k0, k2, cc, dsm, dp, unit0, ustride);
    }
    {
        mu1 = mean4[1]; is1 = istd4[1];
        asm volatile("" : "+v"(mu1.x), "+v"(is1.x));
        const float mu[4] = {mu1.x, mu1.y, mu1.z, mu1.w}, is[4] = {is1.x, is1.y, is1.z, is1.w};
        float k0[4], k2[4], cc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            k0[e] = (gamma ? gamma[cq * 8 + 4 + e] : 1.f) * is[e];
            k2[e] = k0[e] * is[e] * (sht[q][12 + e] * inv_count);
            cc[e] = __builtin_fmaf(mu[e], k2[e], -(k0[e] * (sht[q][4 + e] * inv_count)));
        }
        pb.template apply<1>(k0, k2, cc, dsm, dp, unit0, ustride);
    }
}

// rows per thread of the one-pass kernel by the type of g (fp32 g costs 8 registers per row: fewer rows fit)
inline int plane_nr(int g_dtype) { return g_dtype == MMH_F32 ? 4 : 8; }
inline int plane_lpr_log2(int64_t rows, int C, int g_dtype) {
    if (!row_geom_ok(C) || rows <= 0 || rows * (C / 8) >= (1ll << 30)) return -1;
    const int64_t cap = (int64_t)FT * plane_nr(g_dtype);
    if (rows > cap) return -1;
    int l = 0;
    while (l < 3 && (2 << l) <= C / 8 && rows * (2 << l) <= cap) ++l;
    return l;
}

// ------------------------------------------------------------------ column reductions
// MODE 0: colsum(x)            -> 1 output  (conv-bias gradient)
// MODE 1: norm backward sums   -> 2 outputs (s1 = sum dz, s2 = sum dz*xhat)
// partial layout: ws[((grp*chunks + chunk)*NOUT + o)*C + c]
template <int MODE>
__global__ void col_reduce_partial(const void* __restrict__ a, int alp, const float* __restrict__ outv,
                                   const void* __restrict__ x, int xlp, const float* __restrict__ mean,
                                   const float* __restrict__ invstd, int64_t rows, int C, int cs,
                                   int masked, float dsc, ColGeom cg, float* __restrict__ ws) {
    constexpr int NOUT = MODE == 0 ? 1 : 2;
    __shared__ float sh[NOUT][TPB * 4];
    const int tid = threadIdx.x;
    const int q = tid % cg.lpp, rsub = tid / cg.lpp;
    const int grp = blockIdx.y, chunk = blockIdx.x;
    const int64_t r0 = (int64_t)chunk * cg.rows_per_chunk;
    const int64_t r1 = min(rows, r0 + cg.rows_per_chunk);
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (rsub < cg.rpi) {
        const int64_t gbase = (int64_t)grp * rows;
        float4 mu = make_float4(0, 0, 0, 0), is = mu;
        if (MODE == 1) {
            mu = ld4(mean, (int64_t)grp * cg.lpp + q);
            is = ld4(invstd, (int64_t)grp * cg.lpp + q);
        }
        for (int64_t r = r0 + rsub; r < r1; r += cg.rpi) {
            const int64_t off = (gbase + r) * cs + q * 4;
            float4 g = ld4_lp(a, off >> 2, alp);
            float gv[4] = {g.x, g.y, g.z, g.w};
            if (MODE == 1) {
                if (masked == 2) {      // 4 keep bits per float4 (mmh_scale_shift_act keep_bits)
                    const unsigned kb = reinterpret_cast<const uint8_t*>(outv)[off >> 2];
#pragma unroll
                    for (int e = 0; e < 4; ++e) gv[e] = (kb >> e) & 1u ? gv[e] * dsc : 0.f;
                } else if (masked) {
                    float4 o = *reinterpret_cast<const float4*>(outv + off);
                    gv[0] = o.x > 0.f ? gv[0] * dsc : 0.f;
                    gv[1] = o.y > 0.f ? gv[1] * dsc : 0.f;
                    gv[2] = o.z > 0.f ? gv[2] * dsc : 0.f;
                    gv[3] = o.w > 0.f ? gv[3] * dsc : 0.f;
                }
                float4 xv = ld4_lp(x, off >> 2, xlp);
                float xh[4] = {(xv.x - mu.x) * is.x, (xv.y - mu.y) * is.y, (xv.z - mu.z) * is.z,
                               (xv.w - mu.w) * is.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) s2[e] += gv[e] * xh[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) s1[e] += gv[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        sh[0][tid * 4 + e] = s1[e];
        if (NOUT == 2) sh[NOUT - 1][tid * 4 + e] = s2[e];
    }
    __syncthreads();
    if (rsub == 0) {
#pragma unroll
        for (int o = 0; o < NOUT; ++o)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float acc = sh[o][tid * 4 + e];
                for (int j = 1; j < cg.rpi; ++j) acc += sh[o][(j * cg.lpp + q) * 4 + e];
                ws[((int64_t)(grp * cg.chunks + chunk) * NOUT + o) * C + q * 4 + e] = acc;
            }
    }
}

// out[o][grp][c] (+)= sum over chunks.  32 channels x KL chunk-lanes per block; each lane adds
// its chunks in double, the KL lanes are combined in a fixed order -> deterministic.  KL = 8; KL = 32 when one group has
// many chunks (BatchNorm: 1 group x 1024 chunks over 256 columns is EIGHT work-groups of 256 threads, each lane
// walking 128 chunks: 30 us per call, 4.4 ms per --norm batch step).
template <int KL>
__global__ void __launch_bounds__(32 * KL) col_reduce_final(const float* __restrict__ ws, int groups, int C, int chunks,
                                 int nout, float* __restrict__ o0, float* __restrict__ o1,
                                 int accumulate) {
    __shared__ double sh[2][KL][32];
    const int cl = threadIdx.x & 31, kl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + cl;
    const bool ok = i < groups * C;
    const int grp = ok ? i / C : 0, c = ok ? i - grp * C : 0;
    // both outputs in one sweep, 4 chunks per iteration: the loads are independent of the sums, so
    // up to 8 are in flight instead of one dependent load-add chain per output (20 us -> latency of
    // two round trips); the summation ORDER per lane is unchanged (k ascending), deterministic
    double a0 = 0, a1 = 0;
    if (ok) {
        const float* base = ws + ((int64_t)grp * chunks * nout) * C + c;
        const int64_t kstride = (int64_t)nout * C;
        int k = kl;
        for (; k + 3 * KL < chunks; k += 4 * KL) {
            float v0[4], v1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v0[u] = base[(int64_t)(k + KL * u) * kstride];
                v1[u] = nout > 1 ? base[(int64_t)(k + KL * u) * kstride + C] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { a0 += v0[u]; a1 += v1[u]; }
        }
        for (; k < chunks; k += KL) {
            a0 += base[(int64_t)k * kstride];
            if (nout > 1) a1 += base[(int64_t)k * kstride + C];
        }
    }
    sh[0][kl][cl] = a0;
    sh[1][kl][cl] = a1;
    __syncthreads();
    if (kl < nout && ok) {      // lane group 0 finishes output 0, group 1 output 1
        double t = sh[kl][0][cl];
#pragma unroll
        for (int j = 1; j < KL; ++j) t += sh[kl][j][cl];
        float* dst = kl == 0 ? o0 : o1;
        dst[i] = accumulate ? dst[i] + (float)t : (float)t;
    }
}

// 32 channels x 8 chunk-lanes per work-group; 32 chunk-lanes from 256 chunks per group up
#define MMH_COL_FINAL(ncols, nchunks, st, ...)                                                                     \
    do {                                                                                                         \
        if ((nchunks) >= 256)                                                                                    \
            hipLaunchKernelGGL(col_reduce_final<32>, dim3(((ncols) + 31) / 32), dim3(1024), 0, st, __VA_ARGS__); \
        else                                                                                                     \
            hipLaunchKernelGGL(col_reduce_final<8>, dim3(((ncols) + 31) / 32), dim3(256), 0, st, __VA_ARGS__);   \
    } while (0)

__global__ void norm_bwd_apply_kernel(const void* __restrict__ g, int glp, const float* __restrict__ outv,
                                      const void* __restrict__ x, int xlp, const float* __restrict__ mean,
                                      const float* __restrict__ invstd,
                                      const float* __restrict__ gamma, const float* __restrict__ s1,
                                      const float* __restrict__ s2, float inv_count, int64_t n4,
                                      int64_t rows_per_group, int C4, int masked, float dsc,
                                      void* __restrict__ dx, int dxlp) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        int64_t row = i / C4;
        int c4 = (int)(i - row * C4);
        int64_t gi = (row / rows_per_group) * C4 + c4;
        float4 gv4 = ld4_lp(g, i, glp), xv = ld4_lp(x, i, xlp), mu = ld4(mean, gi), is = ld4(invstd, gi);
        float4 a1 = ld4(s1, gi), a2 = ld4(s2, gi);
        float4 gm = gamma ? ld4(gamma, c4) : make_float4(1.f, 1.f, 1.f, 1.f);
        float gv[4] = {gv4.x, gv4.y, gv4.z, gv4.w};
        if (masked == 2) {
            const unsigned kb = reinterpret_cast<const uint8_t*>(outv)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) gv[e] = (kb >> e) & 1u ? gv[e] * dsc : 0.f;
        } else if (masked) {
            float4 o = ld4(outv, i);
            gv[0] = o.x > 0.f ? gv[0] * dsc : 0.f;
            gv[1] = o.y > 0.f ? gv[1] * dsc : 0.f;
            gv[2] = o.z > 0.f ? gv[2] * dsc : 0.f;
            gv[3] = o.w > 0.f ? gv[3] * dsc : 0.f;
        }
        float4 r;
        r.x = gm.x * is.x * (gv[0] - a1.x * inv_count - (xv.x - mu.x) * is.x * a2.x * inv_count);
        r.y = gm.y * is.y * (gv[1] - a1.y * inv_count - (xv.y - mu.y) * is.y * a2.y * inv_count);
        r.z = gm.z * is.z * (gv[2] - a1.z * inv_count - (xv.z - mu.z) * is.z * a2.z * inv_count);
        r.w = gm.w * is.w * (gv[3] - a1.w * inv_count - (xv.w - mu.w) * is.w * a2.w * inv_count);
        st4_lp(dx, i, r, dxlp);
    }
}

__global__ void act_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y,
                               float* __restrict__ dx, int64_t n4, int act) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        float4 gv = ld4(g, i), yv = ld4(y, i), r;
        if (act == MMH_ACT_RELU) {
            r.x = yv.x > 0.f ? gv.x : 0.f; r.y = yv.y > 0.f ? gv.y : 0.f;
            r.z = yv.z > 0.f ? gv.z : 0.f; r.w = yv.w > 0.f ? gv.w : 0.f;
        } else {
            r.x = gv.x * (1.f - yv.x * yv.x); r.y = gv.y * (1.f - yv.y * yv.y);
            r.z = gv.z * (1.f - yv.z * yv.z); r.w = gv.w * (1.f - yv.w * yv.w);
        }
        st4(dx, i, r);
    }
}

// dx16 = 16-bit(g * act'(y)): the activation backward of a conv epilogue fused with the conversion its 16-bit dgrad
// needs (VGG conv1_1 / conv1_2 + ReLU under the perceptual loss: frozen weights, so nobody else reads the fp32 product)
__global__ void act_bwd_lp16_kernel(const float* __restrict__ g, const float* __restrict__ y, void* __restrict__ out,
                                    int64_t n8, int act, int h16) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n8; i += stride) {
        const float4 g0 = ld4(g, 2 * i), g1 = ld4(g, 2 * i + 1), y0 = ld4(y, 2 * i), y1 = ld4(y, 2 * i + 1);
        const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float yv[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
        f8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            r.v[e] = act == MMH_ACT_RELU ? (yv[e] > 0.f ? gv[e] : 0.f) : gv[e] * (1.f - yv[e] * yv[e]);
        st8<true>(out, i, r, h16 != 0);
    }
}

// the same with g and / or y already in 16 bits (inside the VGG head the edge between conv1_1 and conv1_2 is 16-bit: the
// gradient arrives from conv1_2's 16-bit dgrad, the mask from conv1_1's 16-bit output)
template <bool GW, bool YW>
__global__ void act_bwd_lp16_io_kernel(const void* __restrict__ g, const void* __restrict__ y, void* __restrict__ out,
                                       int64_t n8, int act, int h16) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n8; i += stride) {
        Raw8<GW> gr; Raw8<YW> yr;
        ldraw(gr, g, i); ldraw(yr, y, i);
        const f8 gv = widen(gr, h16 != 0), yv = widen(yr, h16 != 0);
        f8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            r.v[e] = act == MMH_ACT_RELU ? (yv.v[e] > 0.f ? gv.v[e] : 0.f) : gv.v[e] * (1.f - yv.v[e] * yv.v[e]);
        st8<true>(out, i, r, h16 != 0);
    }
}

// ------------------------------------------------------------------ PATBlock gate
__global__ void gate_fwd_kernel(const float* __restrict__ x1, const float* __restrict__ s1,
                                const void* __restrict__ s2, const void* __restrict__ s3,
                                float* __restrict__ out, void* __restrict__ x2n,
                                void* __restrict__ x3n, int64_t n4, int C4, int cat_lp, int s_lp) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        float4 a = ld4(x1, i), b = ld4(s1, i), c = ld4_lp(s2, i, s_lp), d = ld4_lp(s3, i, s_lp), o;
        o.x = a.x + b.x * sigmoidf_(c.x) * sigmoidf_(d.x);
        o.y = a.y + b.y * sigmoidf_(c.y) * sigmoidf_(d.y);
        o.z = a.z + b.z * sigmoidf_(c.z) * sigmoidf_(d.z);
        o.w = a.w + b.w * sigmoidf_(c.w) * sigmoidf_(d.w);
        st4(out, i, o);
        if (x2n) {
            int64_t row = i / C4;
            int c4 = (int)(i - row * C4);
            st4_lp(x2n, row * 2 * C4 + c4, d, cat_lp);        // cat(s3, out)
            st4_lp(x2n, row * 2 * C4 + C4 + c4, o, cat_lp);
            st4_lp(x3n, row * 2 * C4 + c4, c, cat_lp);        // cat(s2, out)
            st4_lp(x3n, row * 2 * C4 + C4 + c4, o, cat_lp);
        }
    }
}

__global__ void gate_bwd_kernel(const float* __restrict__ g_out, const void* __restrict__ g_x2n,
                                const void* __restrict__ g_x3n, const float* __restrict__ s1,
                                const void* __restrict__ s2, const void* __restrict__ s3,
                                float* __restrict__ g_x1, float* __restrict__ g_s1,
                                void* __restrict__ g_s2, void* __restrict__ g_s3, int64_t n4,
                                int C4, int gcat_lp, int s_lp, int gs_lp) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const float4 z4 = make_float4(0, 0, 0, 0);
    for (; i < n4; i += stride) {
        int64_t row = i / C4;
        int c4 = (int)(i - row * C4);
        float4 G = g_out ? ld4(g_out, i) : z4;
        float4 e2 = z4, e3 = z4;  // direct grads on s2 (via x3n) and s3 (via x2n)
        if (g_x2n) {
            float4 t = ld4_lp(g_x2n, row * 2 * C4 + C4 + c4, gcat_lp);
            G.x += t.x; G.y += t.y; G.z += t.z; G.w += t.w;
            e3 = ld4_lp(g_x2n, row * 2 * C4 + c4, gcat_lp);
        }
        if (g_x3n) {
            float4 t = ld4_lp(g_x3n, row * 2 * C4 + C4 + c4, gcat_lp);
            G.x += t.x; G.y += t.y; G.z += t.z; G.w += t.w;
            e2 = ld4_lp(g_x3n, row * 2 * C4 + c4, gcat_lp);
        }
        float4 b = ld4(s1, i), c = ld4_lp(s2, i, s_lp), d = ld4_lp(s3, i, s_lp);
        float Gv[4] = {G.x, G.y, G.z, G.w}, bv[4] = {b.x, b.y, b.z, b.w};
        float cv[4] = {c.x, c.y, c.z, c.w}, dv[4] = {d.x, d.y, d.z, d.w};
        float e2v[4] = {e2.x, e2.y, e2.z, e2.w}, e3v[4] = {e3.x, e3.y, e3.z, e3.w};
        float r1[4], r2[4], r3[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a2 = sigmoidf_(cv[e]), a3 = sigmoidf_(dv[e]);
            r1[e] = Gv[e] * a2 * a3;
            r2[e] = Gv[e] * bv[e] * a3 * a2 * (1.f - a2) + e2v[e];
            r3[e] = Gv[e] * bv[e] * a2 * a3 * (1.f - a3) + e3v[e];
        }
        st4(g_x1, i, G);
        st4(g_s1, i, make_float4(r1[0], r1[1], r1[2], r1[3]));
        st4_lp(g_s2, i, make_float4(r2[0], r2[1], r2[2], r2[3]), gs_lp);
        st4_lp(g_s3, i, make_float4(r3[0], r3[1], r3[2], r3[3]), gs_lp);
    }
}

// ------------------------------------------------------------------ PATBlock gate with the block's last norm inside (16-bit mode)
// A PATBlock's stream-1 branch ends conv2 -> InstanceNorm -> (gate) (models/Generator.py:66-77,115-130): the norm's output s1
// has ONE reader, the gate.  Here the gate reads the conv output y2 (16 bits) and applies scale / shift itself
// (s1 = fma(y2, scale, shift): the expression mmh_scale_shift_act evaluates), so s1 is never written or read (the apply pass
// and 2 B per element of gate input are gone); backward, the gate's kernel - which recomputes s1 and produces the gradient
// gs1 of the norm's output anyway - also leaves the norm backward's plane sums (sum gs1, sum gs1 * xhat) in the partial layout
// and summation order of col_reduce_partial_v2<1>, so mmh_norm_bwd_reduce's pass over gs1 and y2 is gone as well.  Same
// arithmetic, same order: bit-identical to the unfused sequence (tests/test_pointwise_gpu.py).
// Geometry as the other second-generation row kernels: block = (256 / C8 rows) x C8 lanes of 8 channels, grid (chunks, groups).
template <bool SW>      // SW: s2 / s3 are 16-bit
__global__ void __launch_bounds__(TPB) gate_norm_fwd_kernel(
        const float* __restrict__ x1, const void* __restrict__ y2, bool yh16, const float* __restrict__ scale,
        const float* __restrict__ shift, const void* __restrict__ s2, const void* __restrict__ s3, bool sh16,
        float* __restrict__ out, void* __restrict__ x2n, void* __restrict__ x3n, int cat_lp, int64_t rows, RowGeom rg) {
    const int q = threadIdx.x & (rg.c8 - 1), rsub = threadIdx.x / rg.c8;
    const int grp = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * rg.rows_per_chunk;
    const int64_t r1 = min(rows, r0 + rg.rows_per_chunk);
    const f8 sc = ld8<false>(scale, (int64_t)grp * rg.c8 + q, false);
    const f8 sf = ld8<false>(shift, (int64_t)grp * rg.c8 + q, false);
    const int64_t gbase = (int64_t)grp * rows;
    constexpr int U = 2;
    for (int64_t r = r0 + rsub; r < r1; r += (int64_t)rg.rpi * U) {
        Raw8<false> xa[U];
        Raw8<true> ya[U];
        Raw8<SW> ca[U], da[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t rr = r + (int64_t)u * rg.rpi;
            if (rr < r1) {
                const int64_t i8 = (gbase + rr) * rg.c8 + q;
                ldraw(xa[u], x1, i8); ldraw(ya[u], y2, i8); ldraw(ca[u], s2, i8); ldraw(da[u], s3, i8);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t rr = r + (int64_t)u * rg.rpi;
            if (rr >= r1) continue;
            const int64_t row = gbase + rr;
            const f8 a = widen(xa[u], false), y = widen(ya[u], yh16), c = widen(ca[u], sh16), d = widen(da[u], sh16);
            f8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float b = __builtin_fmaf(y.v[e], sc.v[e], sf.v[e]);
                o.v[e] = a.v[e] + b * sigmoidf_(c.v[e]) * sigmoidf_(d.v[e]);
            }
            st8<false>(out, row * rg.c8 + q, o, false);
            if (x2n) {      // cat(s3, out) / cat(s2, out): rows of 2 C channels
                const int64_t j8 = row * 2 * rg.c8 + q;
                if (cat_lp) {
                    st8<true>(x2n, j8, d, cat_lp == 2); st8<true>(x2n, j8 + rg.c8, o, cat_lp == 2);
                    st8<true>(x3n, j8, c, cat_lp == 2); st8<true>(x3n, j8 + rg.c8, o, cat_lp == 2);
                } else {
                    st8<false>(x2n, j8, d, false); st8<false>(x2n, j8 + rg.c8, o, false);
                    st8<false>(x3n, j8, c, false); st8<false>(x3n, j8 + rg.c8, o, false);
                }
            }
        }
    }
}

template <bool GCW, bool SW>    // GCW: the concat gradients are 16-bit; SW: s2 / s3 (and their gradients) are 16-bit
__global__ void __launch_bounds__(TPB) gate_norm_bwd_kernel(
        const float* __restrict__ g_out, const void* __restrict__ g_x2n, const void* __restrict__ g_x3n, bool gch16,
        const void* __restrict__ y2, bool yh16, const float* __restrict__ scale, const float* __restrict__ shift,
        const float* __restrict__ mean, const float* __restrict__ invstd, const void* __restrict__ s2,
        const void* __restrict__ s3, bool sh16, float* __restrict__ g_x1, float* __restrict__ g_s1,
        void* __restrict__ g_s2, void* __restrict__ g_s3, int64_t rows, int C, int c8, int rpi, int chunks,
        int64_t rows_per_chunk, float* __restrict__ ws) {
    __shared__ float sh[2][TPB * 8];
    const int tid = threadIdx.x;
    const int q = tid & (c8 - 1), rsub = tid / c8;
    const int grp = blockIdx.y, chunk = blockIdx.x;
    const int64_t r0 = (int64_t)chunk * rows_per_chunk;
    const int64_t r1 = min(rows, r0 + rows_per_chunk);
    const int64_t gbase = (int64_t)grp * rows;
    const int64_t gi = (int64_t)grp * c8 + q;
    const f8 sc = ld8<false>(scale, gi, false), sf = ld8<false>(shift, gi, false);
    const f8 mu = ld8<false>(mean, gi, false), is = ld8<false>(invstd, gi, false);
    float s1[8], s2a[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2a[e] = 0.f; }
    for (int64_t r = r0 + rsub; r < r1; r += rpi) {      // one row per pass: ascending rows, the order of col_reduce_partial_v2
        const int64_t row = gbase + r;
        const int64_t i8 = row * c8 + q, j8 = row * 2 * c8 + q;
        Raw8<false> Gr; Raw8<GCW> t2, e3r, t3, e2r; Raw8<true> yr; Raw8<SW> cr, dr;
        if (g_out) ldraw(Gr, g_out, i8);
        if (g_x2n) { ldraw(e3r, g_x2n, j8); ldraw(t2, g_x2n, j8 + c8); }
        if (g_x3n) { ldraw(e2r, g_x3n, j8); ldraw(t3, g_x3n, j8 + c8); }
        ldraw(yr, y2, i8); ldraw(cr, s2, i8); ldraw(dr, s3, i8);
        f8 G, e2, e3;
#pragma unroll
        for (int e = 0; e < 8; ++e) { G.v[e] = 0.f; e2.v[e] = 0.f; e3.v[e] = 0.f; }
        if (g_out) G = widen(Gr, false);
        if (g_x2n) {
            const f8 t = widen(t2, gch16);
#pragma unroll
            for (int e = 0; e < 8; ++e) G.v[e] += t.v[e];
            e3 = widen(e3r, gch16);
        }
        if (g_x3n) {
            const f8 t = widen(t3, gch16);
#pragma unroll
            for (int e = 0; e < 8; ++e) G.v[e] += t.v[e];
            e2 = widen(e2r, gch16);
        }
        const f8 y = widen(yr, yh16), c = widen(cr, sh16), d = widen(dr, sh16);
        f8 o1, o2, o3;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float b = __builtin_fmaf(y.v[e], sc.v[e], sf.v[e]);
            const float a2 = sigmoidf_(c.v[e]), a3 = sigmoidf_(d.v[e]);
            o1.v[e] = G.v[e] * a2 * a3;
            o2.v[e] = G.v[e] * b * a3 * a2 * (1.f - a2) + e2.v[e];
            o3.v[e] = G.v[e] * b * a2 * a3 * (1.f - a3) + e3.v[e];
            const float gg = o1.v[e];
            s2a[e] += gg * ((y.v[e] - mu.v[e]) * is.v[e]);
            s1[e] += gg;
        }
        st8<false>(g_x1, i8, G, false);
        st8<false>(g_s1, i8, o1, false);
        st8<SW>(g_s2, i8, o2, sh16);
        st8<SW>(g_s3, i8, o3, sh16);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { sh[0][tid * 8 + e] = s1[e]; sh[1][tid * 8 + e] = s2a[e]; }
    __syncthreads();
    for (int t = tid; t < 8 * c8; t += TPB) {
        const int qq = t >> 3, e = t & 7;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            float acc = sh[o][qq * 8 + e];
            for (int j = 1; j < rpi; ++j) acc += sh[o][(j * c8 + qq) * 8 + e];
            ws[((int64_t)(grp * chunks + chunk) * 2 + o) * C + qq * 8 + e] = acc;
        }
    }
}

// ------------------------------------------------------------------ scalar losses
__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    return r;  // valid in thread 0
}

// MODE 0: BCE-with-logits vs constant target; MODE 1: |a-b|; MODE 2: (a-b)^2
template <int MODE>
__global__ void loss_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                    int64_t n, float target, float* __restrict__ partial) {
    __shared__ float sh[TPB / 64];
    float acc = 0.f;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n4 = n / 4;
    for (; i < n4; i += stride) {
        float4 v = ld4(a, i);
        float xs[4] = {v.x, v.y, v.z, v.w};
        if (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // softplus(-|x|) = log1p(exp(-|x|)): libm's log1pf made this pass COMPUTE-bound (~60 instructions per element:
                // 99 us for the 134 MB of one discriminator map, 1.35 TB/s).  t = exp(-|x|) is in (0, 1]: the hardware log of
                // 1 + t is exact to an ulp where t >= 2^-10, and below it two series terms are (error t^3 / 3 < 4e-10 relative)
                const float x = xs[e];
                const float t = __expf(-fabsf(x));
                const float sp = t < 9.765625e-4f ? t * (1.f - 0.5f * t) : __logf(1.f + t);
                acc += fmaxf(x, 0.f) - x * target + sp;
            }
        } else if (MODE == 1) {
            float4 w = ld4(b, i);
            acc += fabsf(v.x - w.x) + fabsf(v.y - w.y) + fabsf(v.z - w.z) + fabsf(v.w - w.w);
        } else {
            float4 w = ld4(b, i);
            const float d0 = v.x - w.x, d1 = v.y - w.y, d2 = v.z - w.z, d3 = v.w - w.w;
            acc += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        }
    }
    float r = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

__global__ void loss_final_kernel(const float* __restrict__ partial, int nparts, double scale,
                                  float* __restrict__ out) {
    // 256 lanes sum their strided partials in double (i ascending), then a fixed-order tree: deterministic,
    // and not one thread walking 4096 values (that serial loop was 50 us per loss)
    __shared__ double sh[TPB];
    double acc = 0;
    for (int i = threadIdx.x; i < nparts; i += TPB) acc += partial[i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = TPB / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(sh[0] * scale);
}

__global__ void bce_bwd_kernel(const float* __restrict__ x, int64_t n4, float target, float k,
                               const float* __restrict__ gs, float* __restrict__ dx) {
    const float kk = k * gs[0];
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        float4 v = ld4(x, i), r;
        r.x = kk * (sigmoidf_(v.x) - target); r.y = kk * (sigmoidf_(v.y) - target);
        r.z = kk * (sigmoidf_(v.z) - target); r.w = kk * (sigmoidf_(v.w) - target);
        st4(dx, i, r);
    }
}

__device__ __forceinline__ float sgn(float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }

__global__ void l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n4,
                              float k, const float* __restrict__ gs, float* __restrict__ da) {
    const float kk = k * gs[0];
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        float4 v = ld4(a, i), w = ld4(b, i), r;
        r.x = kk * sgn(v.x - w.x); r.y = kk * sgn(v.y - w.y);
        r.z = kk * sgn(v.z - w.z); r.w = kk * sgn(v.w - w.w);
        st4(da, i, r);
    }
}

// L1 between two 16-bit maps (the perceptual term on 16-bit VGG features: each feature is rounded once, the difference and
// the sum are fp32) and its gradient folded with the ReLU mask of the layer that produced `a`: out = [a > 0] * k * sign(a - b)
__global__ void l1_partial_lp16_kernel(const void* __restrict__ a, const void* __restrict__ b, int64_t n8, int h16,
                                       float* __restrict__ partial) {
    __shared__ float sh[TPB / 64];
    float acc = 0.f;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n8; i += stride) {
        Raw8<true> ar, br;
        ldraw(ar, a, i); ldraw(br, b, i);
        const f8 av = widen(ar, h16 != 0), bv = widen(br, h16 != 0);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += fabsf(av.v[e] - bv.v[e]);
    }
    float r = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

__global__ void l1_relu_bwd_lp16_kernel(const void* __restrict__ a, const void* __restrict__ b, int64_t n8, int h16,
                                        float k, const float* __restrict__ gs, void* __restrict__ out) {
    const float kk = k * gs[0];
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n8; i += stride) {
        Raw8<true> ar, br;
        ldraw(ar, a, i); ldraw(br, b, i);
        const f8 av = widen(ar, h16 != 0), bv = widen(br, h16 != 0);
        f8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r.v[e] = av.v[e] > 0.f ? kk * sgn(av.v[e] - bv.v[e]) : 0.f;
        st8<true>(out, i, r, h16 != 0);
    }
}

__global__ void mse_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n4,
                               float k, const float* __restrict__ gs, float* __restrict__ da) {
    const float kk = 2.f * k * gs[0];
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        float4 v = ld4(a, i), w = ld4(b, i), r;
        r.x = kk * (v.x - w.x); r.y = kk * (v.y - w.y);
        r.z = kk * (v.z - w.z); r.w = kk * (v.w - w.w);
        st4(da, i, r);
    }
}

// ------------------------------------------------------------------ Adam
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                            float* __restrict__ m, float* __restrict__ v, int64_t n, float b1,
                            float b2, float eps, float step_size, float inv_sqrt_bc2,
                            float gscale, const int* __restrict__ skip,
                            const float* __restrict__ loss_scale) {
    if (skip && *skip) return;      // a gradient of this iteration was not finite: leave p, m, v alone
    if (loss_scale) gscale /= *loss_scale;      // the backward ran on loss * scale (dynamic loss scaling)
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float gr = g[i] * gscale;
        float mm = b1 * m[i] + (1.f - b1) * gr;
        float vv = b2 * v[i] + (1.f - b2) * gr * gr;
        m[i] = mm;
        v[i] = vv;
        float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
        p[i] = p[i] - step_size * (mm / denom);
    }
}

// Adam with the step count and the learning rate RESIDENT ON THE DEVICE (mmh_adam_step_dev): one thread advances the count -
// unless this step is skipped (overflow), as apex does not count a skipped step - and leaves lr / (1 - b1^t) and
// 1 / sqrt(1 - b2^t), in the double arithmetic the host form uses, where the update kernel reads them.  Nothing about the
// launch depends on the iteration any more: it can sit in a captured graph.
__global__ void adam_tick_kernel(int* __restrict__ step, const float* __restrict__ lr, const int* __restrict__ skip,
                                 float b1, float b2, float* __restrict__ coef) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int t = *step;
    if (!(skip && *skip)) *step = ++t;
    if (t < 1) t = 1;
    const double bc1 = 1.0 - pow((double)b1, (double)t);
    const double bc2 = 1.0 - pow((double)b2, (double)t);
    coef[0] = (float)((double)*lr / bc1);
    coef[1] = (float)(1.0 / sqrt(bc2));
}

__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                float* __restrict__ m, float* __restrict__ v, int64_t n, float b1,
                                float b2, float eps, const float* __restrict__ coef,
                                float gscale, const int* __restrict__ skip,
                                const float* __restrict__ loss_scale) {
    if (skip && *skip) return;
    if (loss_scale) gscale /= *loss_scale;
    const float step_size = coef[0], inv_sqrt_bc2 = coef[1];
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float gr = g[i] * gscale;
        float mm = b1 * m[i] + (1.f - b1) * gr;
        float vv = b2 * v[i] + (1.f - b2) * gr * gr;
        m[i] = mm;
        v[i] = vv;
        float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
        p[i] = p[i] - step_size * (mm / denom);
    }
}

__global__ void u64_add_kernel(uint64_t* __restrict__ p, uint64_t inc) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *p += inc;
}

// ImagePool.query (util/image_pool.py:14-34) with the host's decisions handed over as device indices: per output image
// src[i] >= 0 -> the pool's slot src[i] (its content BEFORE this query), src[i] < 0 -> image -1 - src[i] of the batch;
// then dst[i] >= 0 -> the pool's slot dst[i] takes image i.  Two launches (every read of the pool before any write).
__global__ void pool_gather_kernel(const float* __restrict__ pool, const float* __restrict__ images, float* __restrict__ out,
                                   const int* __restrict__ src, int64_t n4) {
    const int img = blockIdx.y;
    const int sidx = src[img];
    const float* from = sidx >= 0 ? pool + (int64_t)sidx * n4 * 4 : images + (int64_t)(-1 - sidx) * n4 * 4;
    float* to = out + (int64_t)img * n4 * 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) st4(to, i, ld4(from, i));
}

__global__ void pool_scatter_kernel(float* __restrict__ pool, const float* __restrict__ images, const int* __restrict__ dst,
                                    int64_t n4) {
    const int img = blockIdx.y;
    const int d = dst[img];
    if (d < 0) return;
    const float* from = images + (int64_t)img * n4 * 4;
    float* to = pool + (int64_t)d * n4 * 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) st4(to, i, ld4(from, i));
}

// flag |= any(!isfinite(g)): the exponent field of inf/nan is all ones
__global__ void grad_nonfinite_kernel(const float* __restrict__ g, int64_t n4, int64_t n,
                                      int* __restrict__ flag, int* __restrict__ own) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (int64_t k = i; k < n4; k += stride) {
        const float4 v = ld4(g, k);
        bad |= (__float_as_uint(v.x) & 0x7f800000u) == 0x7f800000u;
        bad |= (__float_as_uint(v.y) & 0x7f800000u) == 0x7f800000u;
        bad |= (__float_as_uint(v.z) & 0x7f800000u) == 0x7f800000u;
        bad |= (__float_as_uint(v.w) & 0x7f800000u) == 0x7f800000u;
    }
    for (int64_t k = 4 * n4 + i; k < n; k += stride)
        bad |= (__float_as_uint(g[k]) & 0x7f800000u) == 0x7f800000u;
    if (bad) {              // every writer stores the same value
        *flag = 1;
        if (own) *own = 1;
    }
}

// apex.amp's dynamic loss scaler (LossScaler.update_scale): an overflow multiplies the scale by
// `backoff` and restarts the count of clean steps; `interval` clean steps in a row multiply it by
// `growth`.  state = {scale, clean steps} as two floats; one thread.
__global__ void loss_scale_update_kernel(float* __restrict__ state, const int* __restrict__ overflow,
                                         float growth, float backoff, float interval, float min_scale,
                                         float max_scale) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float scale = state[0], clean = state[1];
    if (*overflow) {
        scale = fmaxf(scale * backoff, min_scale);
        clean = 0.f;
    } else {
        clean += 1.f;
        if (clean >= interval) {
            scale = fminf(scale * growth, max_scale);
            clean = 0.f;
        }
    }
    state[0] = scale;
    state[1] = clean;
}

// ------------------------------------------------------------------ layout
struct PackArgs {
    mmh_plane_src s[4];
    int nsrc;
};
__global__ void pack_nhwc_kernel(PackArgs a, float* __restrict__ nhwc, int B, int H, int W, int Cd,
                                 int dir) {
    // one thread per pixel; the NHWC side moves as float4 (Cd % 4 == 0): 16-byte lanes instead of
    // Cd scalar stores 4*Cd bytes apart
    const int64_t total = (int64_t)B * H * W;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int w = (int)(i % W);
        const int64_t t = i / W;
        const int hh = (int)(t % H);
        const int b = (int)(t / H);
        float4* px = reinterpret_cast<float4*>(nhwc + i * Cd);
        // element address of packed channel c in its source plane set, or null (zero fill / skipped)
        auto chan = [&](int c) -> float* {
            int c0 = 0;
            for (int k = 0; k < a.nsrc; ++k) {
                const mmh_plane_src& s = a.s[k];
                if (c < c0 + s.C) {
                    float* sp = static_cast<float*>(s.ptr);
                    return sp ? sp + b * s.sb + hh * s.sh + w * s.sw + (int64_t)(c - c0) * s.sc : nullptr;
                }
                c0 += s.C;
            }
            return nullptr;
        };
        for (int g4 = 0; g4 < Cd / 4; ++g4) {
            if (dir == 0) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float* q = chan(4 * g4 + e);
                    v[e] = q ? *q : 0.f;
                }
                px[g4] = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                const float4 r = px[g4];
                const float v[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float* q = chan(4 * g4 + e);
                    if (q) *q = v[e];
                }
            }
        }
    }
}

// Gather, second form: a workgroup takes 256 consecutive pixels, reads every source channel with its lanes along the pixels
// (unit stride in an NCHW plane) into LDS and writes the NHWC side from there as whole 16-byte lanes in address order - the
// kernel above gives every thread one pixel, i.e. stores 4 Cd bytes apart (2 TB/s at 24 channels).  In the same pass it
// can write the 16-bit copy with the channels zero-padded to C8 that the 16-bit stems read (mmh_lp16_pad_cvt's output:
// the same conversions of the same values), so that pass and its read of the fp32 tensor are gone; nhwc may be NULL then.
constexpr int PACK_TP = 256;
__global__ void __launch_bounds__(PACK_TP) pack_nhwc_tile_kernel(PackArgs a, float* __restrict__ nhwc, void* __restrict__ out16,
                                                                 int64_t total, int H, int W, int Cd, int C8, int h16) {
    extern __shared__ float pk[];                      // [PACK_TP][pitch], pitch odd
    const int Cm = Cd > C8 ? Cd : C8, pitch = Cm | 1;
    const int tid = threadIdx.x;
    for (int64_t p0 = (int64_t)blockIdx.x * PACK_TP; p0 < total; p0 += (int64_t)gridDim.x * PACK_TP) {
        const int64_t i = p0 + tid;
        if (i < total) {
            const int w = (int)(i % W);
            const int64_t t = i / W;
            const int hh = (int)(t % H);
            const int64_t b = t / H;
            int c = 0;
            for (int k = 0; k < a.nsrc; ++k) {
                const mmh_plane_src& s = a.s[k];
                const float* sp = static_cast<const float*>(s.ptr);
                const float* q = sp ? sp + b * s.sb + hh * s.sh + w * s.sw : nullptr;
                for (int e = 0; e < s.C; ++e, ++c) pk[tid * pitch + c] = q ? q[(int64_t)e * s.sc] : 0.f;
            }
            for (; c < Cm; ++c) pk[tid * pitch + c] = 0.f;
        }
        __syncthreads();
        const int npx = (int)(total - p0 < PACK_TP ? total - p0 : PACK_TP);
        if (nhwc) {
            const int g4n = Cd / 4;
            float4* o = reinterpret_cast<float4*>(nhwc + p0 * Cd);
            for (int j = tid; j < npx * g4n; j += PACK_TP) {
                const int px = j / g4n, g = j - px * g4n;
                const float* r = pk + px * pitch + 4 * g;
                o[j] = make_float4(r[0], r[1], r[2], r[3]);
            }
        }
        if (out16) {
            const int g8n = C8 / 8;
            uint4* o = reinterpret_cast<uint4*>(static_cast<char*>(out16) + p0 * C8 * 2);
            for (int j = tid; j < npx * g8n; j += PACK_TP) {
                const int px = j / g8n, g = j - px * g8n;
                const float* r = pk + px * pitch + 8 * g;
                unsigned u[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = 8 * g + 2 * e < Cd ? r[2 * e] : 0.f, hi = 8 * g + 2 * e + 1 < Cd ? r[2 * e + 1] : 0.f;
                    u[e] = h16 ? ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)lo) |
                                  ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)hi) << 16))
                               : ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)lo) |
                                  ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)hi) << 16));
                }
                o[j] = make_uint4(u[0], u[1], u[2], u[3]);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ pose maps
__global__ void pose_heatmap_kernel(const double* __restrict__ uv, int n_maps, int H, int W,
                                    double sigma, float* __restrict__ out) {
    const int64_t total = (int64_t)n_maps * H * W;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        int x = (int)(i % W);
        int64_t t = i / W;
        int y = (int)(t % H);
        int k = (int)(t / H);
        double dx = (double)x - uv[2 * k], dy = (double)y - uv[2 * k + 1];
        double d2 = dx * dx + dy * dy;
        double v = exp(-d2 / 2.0 / sigma / sigma);
        if (v > 1.0) v = 1.0;
        if (v < 0.0099) v = 0.0;
        out[i] = (float)v;
    }
}

__global__ void map_to_cord_kernel(const float* __restrict__ maps, int H, int W, float thr,
                                   int* __restrict__ cords) {
    __shared__ float smax[TPB];
    __shared__ int sidx[TPB];
    const int k = blockIdx.x;
    const float* m = maps + (int64_t)k * H * W;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < H * W; i += blockDim.x) {
        float v = m[i];
        if (v > best) { best = v; bi = i; }
    }
    smax[threadIdx.x] = best;
    sidx[threadIdx.x] = bi;
    __syncthreads();
    for (int o = TPB / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            float v = smax[threadIdx.x + o];
            int j = sidx[threadIdx.x + o];
            if (v > smax[threadIdx.x] || (v == smax[threadIdx.x] && j < sidx[threadIdx.x])) {
                smax[threadIdx.x] = v;
                sidx[threadIdx.x] = j;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (smax[0] > thr && sidx[0] != 0x7fffffff) {
            cords[2 * k] = sidx[0] / W;      // y (row)
            cords[2 * k + 1] = sidx[0] % W;  // x (col)
        } else {
            cords[2 * k] = -1;
            cords[2 * k + 1] = -1;
        }
    }
}

// ------------------------------------------------------------------ input pipeline
__device__ __forceinline__ float norm_u8(uint8_t v) { return (float)(((double)v / 255.0 - 0.5) / 0.5); }
__device__ __forceinline__ float depth_u8(uint8_t g, uint8_t r) {
    double d = 256.0 * (double)g + (double)r;
    return (float)(((d / 700.0) - 0.5) / 0.5);
}
__device__ __forceinline__ float heat(double x, double y, double ux, double uy, double sigma) {
    double dx = x - ux, dy = y - uy;
    double v = exp(-(dx * dx + dy * dy) / 2.0 / sigma / sigma);
    if (v > 1.0) v = 1.0;
    if (v < 0.0099) v = 0.0;
    return (float)v;
}

__global__ void decode_inputs_kernel(const uint8_t* __restrict__ img1, const uint8_t* __restrict__ img2,
                                     const uint8_t* __restrict__ dep1, const uint8_t* __restrict__ dep2,
                                     const double* __restrict__ uv1, const double* __restrict__ uv2,
                                     int B, int H, int W, double sigma, float* __restrict__ xh1,
                                     float* __restrict__ xh2, float* __restrict__ xp,
                                     float* __restrict__ xd) {
    const int64_t total = (int64_t)B * H * W;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int x = (int)(i % W);
        const int64_t t = i / W;
        const int y = (int)(t % H);
        const int b = (int)(t / H);
        // images: BGR uint8 -> RGB normalised
        const uint8_t* p1 = img1 + i * 3;
        st4(xh1, i, make_float4(norm_u8(p1[2]), norm_u8(p1[1]), norm_u8(p1[0]), 0.f));
        const uint8_t* p2 = img2 + i * 3;
        st4(xh2, i, make_float4(norm_u8(p2[2]), norm_u8(p2[1]), norm_u8(p2[0]), 0.f));
        // depth: 256*G + R (BGR order: index 1 = G, 2 = R)
        const float d1 = depth_u8(dep1[i * 3 + 1], dep1[i * 3 + 2]);
        const float d2 = depth_u8(dep2[i * 3 + 1], dep2[i * 3 + 2]);
        st4(xd, i * 2, make_float4(d1, d1, d1, d2));
        st4(xd, i * 2 + 1, make_float4(d2, d2, 0.f, 0.f));
        // pose maps
        float* pp = xp + i * 44;
        const double* u1 = uv1 + (int64_t)b * 42;
        const double* u2 = uv2 + (int64_t)b * 42;
        for (int k = 0; k < 21; ++k) {
            pp[k] = heat((double)x, (double)y, u1[2 * k], u1[2 * k + 1], sigma);
            pp[21 + k] = heat((double)x, (double)y, u2[2 * k], u2[2 * k + 1], sigma);
        }
        pp[42] = 0.f;
        pp[43] = 0.f;
    }
}

// ------------------------------------------------------------------ bf16 weight copies
// T = __bf16 or _Float16
template <typename T>
__global__ void prep_weights_bf16_kernel(const float* __restrict__ w, int taps, int R, int C,
                                         T* __restrict__ plain, T* __restrict__ tr) {
    const int64_t total = (int64_t)taps * R * C;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int c = (int)(i % C);
        const int64_t t2 = i / C;
        const int r = (int)(t2 % R);
        const int t = (int)(t2 / R);
        const T v = (T)w[i];
        if (plain) plain[i] = v;
        if (tr) tr[((int64_t)t * C + c) * R + r] = v;
    }
}

// The same for MANY weights in one launch (mmh_prep_weights_lp16_multi): after an optimizer step every 16-bit copy of a
// network is stale, and 78 conversions of 5-15 us each cannot fill the chip (0.86 ms of the 16-bit step).  table: n rows of
// eight int64 - {w, plain, tr, taps, R, C, first block, fp16} - entry e owning the blocks [first(e), first(e + 1)): one
// block per (tap, 64 x 64 tile of the R x C matrix).  The tile goes through LDS so that both copies are written in rows
// (the one-weight kernel above scatters the transposed copy two bytes at a time: gathered into one launch that was
// SLOWER than the 78 small ones).
constexpr int PREP_TILE = 64;
__global__ void __launch_bounds__(TPB) prep_weights_lp16_multi_kernel(const long long* __restrict__ table, int n) {
    __shared__ int s_e;
    __shared__ unsigned short tile[PREP_TILE][PREP_TILE + 2];
    if (threadIdx.x == 0) {
        int lo = 0, hi = n - 1;                 // the last entry whose first block is <= blockIdx.x
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (table[mid * 8 + 6] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        s_e = lo;
    }
    __syncthreads();
    const long long* row = table + s_e * 8;
    const float* __restrict__ w = reinterpret_cast<const float*>(row[0]);
    unsigned short* __restrict__ plain = reinterpret_cast<unsigned short*>(row[1]);
    unsigned short* __restrict__ tr = reinterpret_cast<unsigned short*>(row[2]);
    const int R = (int)row[4], C = (int)row[5];
    const bool h16 = row[7] != 0;
    const int RT = (R + PREP_TILE - 1) / PREP_TILE, CT = (C + PREP_TILE - 1) / PREP_TILE;
    int bidx = (int)((long long)blockIdx.x - row[6]);
    const int ct = bidx % CT; bidx /= CT;
    const int rt = bidx % RT;
    const int t = bidx / RT;
    const int r0 = rt * PREP_TILE, c0 = ct * PREP_TILE;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;        // 16 x 16 threads, four values each way
    const float* wt = w + (int64_t)t * R * C;
    const bool vec = (C % 4 == 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 16 * k, c = c0 + 4 * tx;
        if (r >= R) continue;
        float f[4] = {0.f, 0.f, 0.f, 0.f};
        if (vec && c + 3 < C) {
            const float4 v = *reinterpret_cast<const float4*>(wt + (int64_t)r * C + c);
            f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
        } else {
            for (int e = 0; e < 4; ++e) if (c + e < C) f[e] = wt[(int64_t)r * C + c + e];
        }
        unsigned short h[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            h[e] = h16 ? __builtin_bit_cast(unsigned short, (_Float16)f[e]) : __builtin_bit_cast(unsigned short, (__bf16)f[e]);
            tile[ty + 16 * k][4 * tx + e] = h[e];
        }
        if (plain) {
            unsigned short* po = plain + ((int64_t)t * R + r) * C + c;
            if (vec && c + 3 < C) *reinterpret_cast<uint2*>(po) = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
            else for (int e = 0; e < 4; ++e) if (c + e < C) po[e] = h[e];
        }
    }
    __syncthreads();
    if (tr) {
        const bool rvec = (R % 4 == 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + ty + 16 * k, r = r0 + 4 * tx;
            if (c >= C) continue;
            unsigned short h[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = tile[4 * tx + e][ty + 16 * k];
            unsigned short* po = tr + ((int64_t)t * C + c) * R + r;
            if (rvec && r + 3 < R) *reinterpret_cast<uint2*>(po) = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
            else for (int e = 0; e < 4; ++e) if (r + e < R) po[e] = h[e];
        }
    }
}

// w [taps][R][C] fp32 -> flat [C][Kpad] bf16 with k = t*R + r (zero padded): fprop of small-Cin convs
template <typename T>
__global__ void prep_weights_bf16_flat_kernel(const float* __restrict__ w, int taps, int R, int C,
                                              int Kpad, T* __restrict__ out) {
    const int64_t total = (int64_t)C * Kpad;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int k = (int)(i % Kpad);
        const int n = (int)(i / Kpad);
        float v = 0.f;
        if (k < taps * R) v = w[(int64_t)k * C + n];      // [t][r][n] is flat in k = t*R + r
        out[i] = (T)v;
    }
}

inline bool is_dtype(int d) { return d == MMH_F32 || d == MMH_BF16 || d == MMH_FP16; }

int check_cols(const char* who, int C) {
    MMH_REQUIRE(C > 0 && C % 4 == 0 && C <= 1024, "%s: C must be a multiple of 4 in (0,1024], got %d",
                who, C);
    return 0;
}


// MaxPool2d(2, 2) on NHWC fp32 (the pooling layers of vgg19.features when --perceptual_layers reaches past index 3:
// losses/L1_plus_perceptualLoss.py:22-27).  One lane per (output pixel, 4 channels).  Backward: the gradient goes to the
// FIRST maximum of the window in scan order (0,0), (0,1), (1,0), (1,1) - what torch's max_pool2d does.
__global__ void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C4) {
    const int Ho = H / 2, Wo = W / 2;
    const int64_t total = (int64_t)B * Ho * Wo * C4;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int c = (int)(i % C4);
        int64_t t = i / C4;
        const int ow = (int)(t % Wo); t /= Wo;
        const int oh = (int)(t % Ho);
        const int b = (int)(t / Ho);
        const float4* r0 = reinterpret_cast<const float4*>(x) + (((int64_t)b * H + 2 * oh) * W + 2 * ow) * C4 + c;
        const float4* r1 = r0 + (int64_t)W * C4;
        const float4 a = r0[0], bb = r0[C4], cc = r1[0], d = r1[C4];
        float4 m;
        m.x = fmaxf(fmaxf(a.x, bb.x), fmaxf(cc.x, d.x)); m.y = fmaxf(fmaxf(a.y, bb.y), fmaxf(cc.y, d.y));
        m.z = fmaxf(fmaxf(a.z, bb.z), fmaxf(cc.z, d.z)); m.w = fmaxf(fmaxf(a.w, bb.w), fmaxf(cc.w, d.w));
        reinterpret_cast<float4*>(y)[i] = m;
    }
}

__global__ void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ dx,
                                    int B, int H, int W, int C4) {
    const int Ho = H / 2, Wo = W / 2;
    const int64_t total = (int64_t)B * Ho * Wo * C4;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int c = (int)(i % C4);
        int64_t t = i / C4;
        const int ow = (int)(t % Wo); t /= Wo;
        const int oh = (int)(t % Ho);
        const int b = (int)(t / Ho);
        const int64_t o0 = (((int64_t)b * H + 2 * oh) * W + 2 * ow) * C4 + c, o1 = o0 + (int64_t)W * C4;
        const float4* xv = reinterpret_cast<const float4*>(x);
        const float4 v[4] = {xv[o0], xv[o0 + C4], xv[o1], xv[o1 + C4]};
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 o[4];
        auto route = [&](float va, float vb, float vc, float vd, float gg, float& oa, float& ob, float& oc, float& od) {
            int k = 0; float m = va;
            if (vb > m) { m = vb; k = 1; }
            if (vc > m) { m = vc; k = 2; }
            if (vd > m) { m = vd; k = 3; }
            oa = k == 0 ? gg : 0.f; ob = k == 1 ? gg : 0.f; oc = k == 2 ? gg : 0.f; od = k == 3 ? gg : 0.f;
        };
        route(v[0].x, v[1].x, v[2].x, v[3].x, gv.x, o[0].x, o[1].x, o[2].x, o[3].x);
        route(v[0].y, v[1].y, v[2].y, v[3].y, gv.y, o[0].y, o[1].y, o[2].y, o[3].y);
        route(v[0].z, v[1].z, v[2].z, v[3].z, gv.z, o[0].z, o[1].z, o[2].z, o[3].z);
        route(v[0].w, v[1].w, v[2].w, v[3].w, gv.w, o[0].w, o[1].w, o[2].w, o[3].w);
        float4* dv = reinterpret_cast<float4*>(dx);
        dv[o0] = o[0]; dv[o0 + C4] = o[1]; dv[o1] = o[2]; dv[o1 + C4] = o[3];
    }
}

}  // namespace

extern "C" {

size_t mmh_norm_stats_ws_bytes(int groups, int64_t rows, int C) {
    if (groups <= 0 || rows <= 0 || C <= 0 || C % 4) return 0;
    ColGeom g = col_geom(groups, rows, C);
    return (size_t)groups * g.chunks * 3 * C * sizeof(float);
}

int mmh_norm_stats(const void* x, int groups, int64_t rows, int C, int cs, void* mean, void* m2,
                   void* ws, size_t ws_bytes, int x_dtype, mmh_stream_t s) {
    if (int rc = check_cols("mmh_norm_stats", C)) return rc;
    MMH_REQUIRE(is_dtype(x_dtype), "mmh_norm_stats: x_dtype must be MMH_F32 | MMH_BF16 | MMH_FP16");
    MMH_REQUIRE(x && mean && m2 && ws && groups > 0 && rows > 0 && cs >= C && cs % 4 == 0,
                "mmh_norm_stats: bad arguments");
    MMH_REQUIRE(ws_bytes >= mmh_norm_stats_ws_bytes(groups, rows, C), "mmh_norm_stats: workspace too small");
    ColGeom g = col_geom(groups, rows, C);
    hipStream_t st = mmh::as_stream(s);
    if (mmh::g_pw_v2 && row_geom_ok(C) && cs == C) {
        const int c8 = C / 8;
        if (x_dtype != MMH_F32)
            hipLaunchKernelGGL(norm_stats_partial_v2<true>, dim3(g.chunks, groups), dim3(TPB), 0, st, x,
                               x_dtype == MMH_FP16, rows, C, c8, TPB / c8, g.chunks, g.rows_per_chunk,
                               static_cast<float*>(ws));
        else
            hipLaunchKernelGGL(norm_stats_partial_v2<false>, dim3(g.chunks, groups), dim3(TPB), 0, st, x, false,
                               rows, C, c8, TPB / c8, g.chunks, g.rows_per_chunk, static_cast<float*>(ws));
    } else
    hipLaunchKernelGGL(norm_stats_partial, dim3(g.chunks, groups), dim3(TPB), 0, st,
                       x, x_dtype, rows, C, cs, g, static_cast<float*>(ws));
    hipLaunchKernelGGL(norm_stats_final, dim3((groups * C + 31) / 32), dim3(TPB), 0, st,
                       static_cast<const float*>(ws), groups, C, g.chunks, static_cast<float*>(mean),
                       static_cast<float*>(m2));
    return mmh::check_launch("norm_stats");
}

int mmh_norm_stats_merge(const void* partials, int groups, int chunks, int C, void* mean, void* m2,
                         mmh_stream_t s) {
    MMH_REQUIRE(partials && mean && m2 && groups > 0 && chunks > 0 && C > 0, "mmh_norm_stats_merge: bad arguments");
    hipLaunchKernelGGL(norm_stats_final, dim3((groups * C + 31) / 32), dim3(TPB), 0, mmh::as_stream(s),
                       static_cast<const float*>(partials), groups, C, chunks, static_cast<float*>(mean),
                       static_cast<float*>(m2));
    return mmh::check_launch("norm_stats_merge");
}

// Two-level merge for ONE group over many chunks (BatchNorm statistics from conv-epilogue partials: 1024 chunks per
// channel after a 16-bit 3x3 conv, 3872 after a Winograd one): `sub` blocks of chunks are merged in parallel into a coarser
// partial array in ws ([sub][3][C] floats), which the single-group merge then finishes.  chunks % sub == 0.
size_t mmh_norm_stats_merge2_ws_bytes(int sub, int C) { return sub > 0 && C > 0 ? (size_t)sub * 3 * C * sizeof(float) : 0; }

int mmh_norm_stats_merge2(const void* partials, int chunks, int C, int sub, void* ws, size_t ws_bytes, void* mean,
                          void* m2, mmh_stream_t s) {
    MMH_REQUIRE(partials && mean && m2 && ws && chunks > 0 && C > 0 && sub > 0 && chunks % sub == 0,
                "mmh_norm_stats_merge2: bad arguments (chunks %% sub == 0)");
    MMH_REQUIRE(ws_bytes >= mmh_norm_stats_merge2_ws_bytes(sub, C), "mmh_norm_stats_merge2: workspace too small");
    float* t = static_cast<float*>(ws);
    hipStream_t st = mmh::as_stream(s);
    hipLaunchKernelGGL(norm_stats_final, dim3((sub * C + 31) / 32), dim3(TPB), 0, st, static_cast<const float*>(partials),
                       sub, C, chunks / sub, t + C, t + 2 * C, 0.0, 0.f, nullptr, nullptr, nullptr, t, 3 * C);
    hipLaunchKernelGGL(norm_stats_final, dim3((C + 31) / 32), dim3(TPB), 0, st, static_cast<const float*>(t), 1, C, sub,
                       static_cast<float*>(mean), static_cast<float*>(m2));
    return mmh::check_launch("norm_stats_merge2");
}

int mmh_norm_stats_merge_finalize(const void* partials, int groups, int chunks, int C, double count, float eps,
                                  void* mean, void* m2, void* scale, void* shift, void* invstd, mmh_stream_t s) {
    MMH_REQUIRE(partials && mean && m2 && scale && shift && invstd && groups > 0 && chunks > 0 && C > 0 && count > 0,
                "mmh_norm_stats_merge_finalize: bad arguments");
    hipLaunchKernelGGL(norm_stats_final, dim3((groups * C + 31) / 32), dim3(TPB), 0, mmh::as_stream(s),
                       static_cast<const float*>(partials), groups, C, chunks, static_cast<float*>(mean),
                       static_cast<float*>(m2), count, eps, static_cast<float*>(scale), static_cast<float*>(shift),
                       static_cast<float*>(invstd));
    return mmh::check_launch("norm_stats_merge_finalize");
}

int mmh_syncbn_merge_finalize(const void* gathered, int ranks, int64_t rank_stride, int C, double count, float eps,
                              const void* gamma, const void* beta, void* mean, void* m2, void* scale, void* shift, void* invstd,
                              void* running_mean, void* running_var, float momentum, mmh_stream_t s) {
    MMH_REQUIRE(gathered && mean && m2 && scale && shift && invstd && ranks > 0 && C > 0 && count > 0 && rank_stride >= 3 * (int64_t)C,
                "mmh_syncbn_merge_finalize: bad arguments (rank_stride >= 3 C floats)");
    MMH_REQUIRE(!running_mean || running_var, "mmh_syncbn_merge_finalize: running_mean without running_var");
    hipLaunchKernelGGL(norm_stats_final, dim3((C + 31) / 32), dim3(TPB), 0, mmh::as_stream(s), static_cast<const float*>(gathered), 1, C,
                       ranks, static_cast<float*>(mean), static_cast<float*>(m2), count, eps, static_cast<float*>(scale),
                       static_cast<float*>(shift), static_cast<float*>(invstd), nullptr, 0, rank_stride,
                       static_cast<const float*>(gamma), static_cast<const float*>(beta), static_cast<float*>(running_mean),
                       static_cast<float*>(running_var), momentum);
    return mmh::check_launch("syncbn_merge_finalize");
}

int mmh_norm_finalize(const void* mean, const void* m2, double count, const void* gamma,
                      const void* beta, float eps, int groups, int C, void* scale, void* shift,
                      void* invstd, void* running_mean, void* running_var, float momentum,
                      mmh_stream_t s) {
    MMH_REQUIRE(mean && m2 && scale && shift && invstd && groups > 0 && C > 0 && count > 0,
                "mmh_norm_finalize: bad arguments");
    MMH_REQUIRE(!running_mean || (groups == 1 && running_var), "mmh_norm_finalize: running stats need groups==1");
    hipLaunchKernelGGL(norm_finalize_kernel, dim3((groups * C + TPB - 1) / TPB), dim3(TPB), 0,
                       mmh::as_stream(s), static_cast<const float*>(mean),
                       static_cast<const float*>(m2), count, static_cast<const float*>(gamma),
                       static_cast<const float*>(beta), eps, groups, C, static_cast<float*>(scale),
                       static_cast<float*>(shift), static_cast<float*>(invstd),
                       static_cast<float*>(running_mean), static_cast<float*>(running_var), momentum);
    return mmh::check_launch("norm_finalize");
}

static int scale_shift_act_impl(const void* x, const void* scale, const void* shift, const void* residual,
                                void* out, int groups, int64_t rows, int C, int relu, float drop_p,
                                uint64_t seed, const void* mask, void* keep_bits, int x_dtype, int out_dtype,
                                void* twin, int twin_dtype, mmh_stream_t s) {
    if (int rc = check_cols("mmh_scale_shift_act", C)) return rc;
    MMH_REQUIRE(!twin || (mmh::g_pw_v2 && row_geom_ok(C) && (twin_dtype == MMH_BF16 || twin_dtype == MMH_FP16)),
                "mmh_scale_shift_act_twin: the 16-bit twin needs C / 8 a power of two <= 256 and a 16-bit twin_dtype");
    MMH_REQUIRE(x && scale && shift && out && groups > 0 && rows > 0, "mmh_scale_shift_act: bad arguments");
    MMH_REQUIRE(is_dtype(x_dtype) && is_dtype(out_dtype),
                "mmh_scale_shift_act: x_dtype / out_dtype must be MMH_F32 | MMH_BF16 | MMH_FP16");
    MMH_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || relu),
                "mmh_scale_shift_act: dropout needs 0<=p<1 and a preceding ReLU");
    if (mmh::g_pw_v2 && row_geom_ok(C)) {
        const RowGeom rg = row_geom(groups, rows, C);
        const dim3 grid(rg.chunks, groups);
        const bool xw = x_dtype != MMH_F32, ow = out_dtype != MMH_F32;
        const bool extra = residual != nullptr || mask != nullptr;
#define MMH_SSA(XW, OW)                                                                                        \
    if (extra) MMH_SSA2(XW, OW, true); else MMH_SSA2(XW, OW, false)
#define MMH_SSA2(XW, OW, EX)                                                                                   \
    hipLaunchKernelGGL((scale_shift_act_v2<XW, OW, EX>), grid, dim3(TPB), 0, mmh::as_stream(s), x,            \
                       static_cast<const float*>(scale), static_cast<const float*>(shift),                    \
                       static_cast<const float*>(residual), out, rows, rg, relu, drop_p, seed,                \
                       static_cast<const uint8_t*>(mask), static_cast<uint8_t*>(keep_bits),                   \
                       x_dtype == MMH_FP16, out_dtype == MMH_FP16, twin, twin_dtype == MMH_FP16,                \
                       drop_p > 0.f ? g_dropout_salt : nullptr)
        if (xw && ow) { MMH_SSA(true, true); }
        else if (xw) { MMH_SSA(true, false); }
        else if (ow) { MMH_SSA(false, true); }
        else { MMH_SSA(false, false); }
#undef MMH_SSA
#undef MMH_SSA2
        return mmh::check_launch("scale_shift_act_v2");
    }
    const int64_t n4 = (int64_t)groups * rows * (C / 4);
    hipLaunchKernelGGL(scale_shift_act_kernel, dim3(grid_for(n4)), dim3(TPB), 0, mmh::as_stream(s),
                       x, x_dtype, static_cast<const float*>(scale),
                       static_cast<const float*>(shift), static_cast<const float*>(residual),
                       out, n4, rows, C / 4, relu, drop_p, seed,
                       static_cast<const uint8_t*>(mask), static_cast<uint8_t*>(keep_bits), out_dtype,
                       drop_p > 0.f ? g_dropout_salt : nullptr);
    return mmh::check_launch("scale_shift_act");
}

int mmh_scale_shift_act(const void* x, const void* scale, const void* shift, const void* residual,
                        void* out, int groups, int64_t rows, int C, int relu, float drop_p,
                        uint64_t seed, const void* mask, void* keep_bits, int x_dtype, int out_dtype,
                        mmh_stream_t s) {
    return scale_shift_act_impl(x, scale, shift, residual, out, groups, rows, C, relu, drop_p, seed, mask, keep_bits, x_dtype,
                                out_dtype, nullptr, MMH_BF16, s);
}

int mmh_scale_shift_act_twin(const void* x, const void* scale, const void* shift, const void* residual,
                             void* out, int groups, int64_t rows, int C, int relu, float drop_p,
                             uint64_t seed, const void* mask, void* keep_bits, int x_dtype, int out_dtype,
                             void* twin, int twin_dtype, mmh_stream_t s) {
    MMH_REQUIRE(twin, "mmh_scale_shift_act_twin: NULL twin (use mmh_scale_shift_act)");
    return scale_shift_act_impl(x, scale, shift, residual, out, groups, rows, C, relu, drop_p, seed, mask, keep_bits, x_dtype,
                                out_dtype, twin, twin_dtype, s);
}

size_t mmh_norm_bwd_ws_bytes(int groups, int64_t rows, int C) {
    if (groups <= 0 || rows <= 0 || C <= 0 || C % 4) return 0;
    ColGeom g = col_geom(groups, rows, C);
    return (size_t)groups * g.chunks * 2 * C * sizeof(float);
}

int mmh_norm_bwd_reduce(const void* g, const void* out, const void* x, const void* mean,
                        const void* invstd, int groups, int64_t rows, int C, int masked,
                        float drop_p, void* s1, void* s2, void* ws, size_t ws_bytes,
                        int g_dtype, int x_dtype, mmh_stream_t s) {
    if (int rc = check_cols("mmh_norm_bwd_reduce", C)) return rc;
    MMH_REQUIRE(is_dtype(g_dtype) && is_dtype(x_dtype), "mmh_norm_bwd_reduce: bad g_dtype / x_dtype");
    MMH_REQUIRE(g && x && mean && invstd && s1 && s2 && ws && (!masked || out),
                "mmh_norm_bwd_reduce: NULL buffer");
    MMH_REQUIRE(ws_bytes >= mmh_norm_bwd_ws_bytes(groups, rows, C), "mmh_norm_bwd_reduce: workspace too small");
    ColGeom cg = col_geom(groups, rows, C);
    hipStream_t st = mmh::as_stream(s);
    const float dsc = 1.f / (1.f - drop_p);
    if (mmh::g_pw_v2 && row_geom_ok(C) && masked != 1) {
        const int c8 = C / 8;
        const bool gw = g_dtype != MMH_F32, xw = x_dtype != MMH_F32;
#define MMH_CR(AW, XW)                                                                                        \
    hipLaunchKernelGGL((col_reduce_partial_v2<1, AW, XW>), dim3(cg.chunks, groups), dim3(TPB), 0, st, g,      \
                       g_dtype == MMH_FP16, static_cast<const uint8_t*>(out), x, x_dtype == MMH_FP16,         \
                       static_cast<const float*>(mean), static_cast<const float*>(invstd), rows, C, c8,       \
                       TPB / c8, cg.chunks, cg.rows_per_chunk, masked, dsc, static_cast<float*>(ws))
        if (gw && xw) MMH_CR(true, true);
        else if (gw) MMH_CR(true, false);
        else if (xw) MMH_CR(false, true);
        else MMH_CR(false, false);
#undef MMH_CR
    } else
    hipLaunchKernelGGL((col_reduce_partial<1>), dim3(cg.chunks, groups), dim3(TPB), 0, st,
                       g, g_dtype, static_cast<const float*>(out),
                       x, x_dtype, static_cast<const float*>(mean),
                       static_cast<const float*>(invstd), rows, C, C, masked, dsc, cg,
                       static_cast<float*>(ws));
    MMH_COL_FINAL(groups * C, cg.chunks, st,
                       static_cast<const float*>(ws), groups, C, cg.chunks, 2,
                       static_cast<float*>(s1), static_cast<float*>(s2), 0);
    return mmh::check_launch("norm_bwd_reduce");
}

// the second half of mmh_norm_bwd_reduce alone: partials [groups][chunks][2][C] (written by a convolution's epilogue,
// mmh_conv3x3_lp16_dgrad_nbr) -> s1, s2 [groups][C], summed in the reduce pass's fixed order
int mmh_norm_bwd_sums_final(const void* part, int groups, int C, int chunks, void* s1, void* s2, mmh_stream_t s) {
    MMH_REQUIRE(part && s1 && s2 && groups > 0 && C > 0 && chunks > 0, "mmh_norm_bwd_sums_final: bad arguments");
    hipStream_t st = mmh::as_stream(s);
    MMH_COL_FINAL(groups * C, chunks, st, static_cast<const float*>(part), groups, C, chunks, 2, static_cast<float*>(s1),
                  static_cast<float*>(s2), 0);
    return mmh::check_launch("norm_bwd_sums_final");
}

int mmh_norm_bwd_apply(const void* g, const void* out, const void* x, const void* mean,
                       const void* invstd, const void* gamma, const void* s1, const void* s2,
                       double count, int groups, int64_t rows, int C, int masked, float drop_p,
                       void* dx, int g_dtype, int x_dtype, int dx_dtype, mmh_stream_t s) {
    if (int rc = check_cols("mmh_norm_bwd_apply", C)) return rc;
    MMH_REQUIRE(is_dtype(g_dtype) && is_dtype(x_dtype) && is_dtype(dx_dtype),
                "mmh_norm_bwd_apply: bad g_dtype / x_dtype / dx_dtype");
    MMH_REQUIRE(g && x && mean && invstd && s1 && s2 && dx && (!masked || out) && count > 0,
                "mmh_norm_bwd_apply: bad arguments");
    if (mmh::g_pw_v2 && row_geom_ok(C) && masked != 1) {
        const RowGeom rg = row_geom(groups, rows, C);
        const int sel = (g_dtype != MMH_F32 ? 4 : 0) | (x_dtype != MMH_F32 ? 2 : 0) | (dx_dtype != MMH_F32 ? 1 : 0);
#define MMH_BA(GW, XW, DW)                                                                                     \
    hipLaunchKernelGGL((norm_bwd_apply_v2<GW, XW, DW>), dim3(rg.chunks, groups), dim3(TPB), 0,                \
                       mmh::as_stream(s), g, g_dtype == MMH_FP16, static_cast<const uint8_t*>(out), x,        \
                       x_dtype == MMH_FP16, static_cast<const float*>(mean), static_cast<const float*>(invstd), \
                       static_cast<const float*>(gamma), static_cast<const float*>(s1),                       \
                       static_cast<const float*>(s2), (float)(1.0 / count), rows, rg, masked,                 \
                       1.f / (1.f - drop_p), dx, dx_dtype == MMH_FP16)
        switch (sel) {
            case 0: MMH_BA(false, false, false); break;
            case 1: MMH_BA(false, false, true); break;
            case 2: MMH_BA(false, true, false); break;
            case 3: MMH_BA(false, true, true); break;
            case 4: MMH_BA(true, false, false); break;
            case 5: MMH_BA(true, false, true); break;
            case 6: MMH_BA(true, true, false); break;
            default: MMH_BA(true, true, true); break;
        }
#undef MMH_BA
        return mmh::check_launch("norm_bwd_apply_v2");
    }
    const int64_t n4 = (int64_t)groups * rows * (C / 4);
    hipLaunchKernelGGL(norm_bwd_apply_kernel, dim3(grid_for(n4)), dim3(TPB), 0, mmh::as_stream(s),
                       g, g_dtype, static_cast<const float*>(out),
                       x, x_dtype, static_cast<const float*>(mean),
                       static_cast<const float*>(invstd), static_cast<const float*>(gamma),
                       static_cast<const float*>(s1), static_cast<const float*>(s2),
                       (float)(1.0 / count), n4, rows, C / 4, masked, 1.f / (1.f - drop_p),
                       dx, dx_dtype);
    return mmh::check_launch("norm_bwd_apply");
}

int mmh_norm_bwd_fused_supported(int groups, int64_t rows, int C, int masked, int g_dtype, int x_dtype) {
    return (mmh::g_pw_v2 && groups > 0 && masked != 1 && is_dtype(g_dtype) && is_dtype(x_dtype) && x_dtype != MMH_F32 &&
            plane_lpr_log2(rows, C, g_dtype) >= 0) ? 1 : 0;
}

int mmh_norm_bwd_fused(const void* g, const void* out, const void* x, const void* mean, const void* invstd,
                       const void* gamma, double count, int groups, int64_t rows, int C, int masked, float drop_p,
                       void* s1, void* s2, void* dx, int g_dtype, int x_dtype, int dx_dtype, mmh_stream_t s) {
    if (int rc = check_cols("mmh_norm_bwd_fused", C)) return rc;
    MMH_REQUIRE(is_dtype(dx_dtype) && mmh_norm_bwd_fused_supported(groups, rows, C, masked, g_dtype, x_dtype),
                "mmh_norm_bwd_fused: unsupported shape / types (ask mmh_norm_bwd_fused_supported; use mmh_norm_bwd_reduce + _apply)");
    MMH_REQUIRE(g && x && mean && invstd && s1 && s2 && dx && (!masked || out) && count > 0, "mmh_norm_bwd_fused: bad arguments");
    MMH_REQUIRE((g_dtype == MMH_F32 || g_dtype == x_dtype) && (dx_dtype == MMH_F32 || dx_dtype == x_dtype),
                "mmh_norm_bwd_fused: the 16-bit tensors of one call share ONE format (bf16 or fp16)");
    const int l2 = plane_lpr_log2(rows, C, g_dtype);
    const dim3 grid((C / 8) >> l2, groups);
    const bool gw = g_dtype != MMH_F32, dw = dx_dtype != MMH_F32;
#define MMH_PL(GW, DW, NR, H16, MK)                                                                                 \
    {                                                                                                               \
        auto kfn = norm_bwd_plane_kernel<GW, DW, NR, H16, MK>;                                                      \
        static bool attr_done = false;                                                                              \
        if (!attr_done) {                                                                                           \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),                                  \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, NR * FT * 16);           \
            if (e != hipSuccess) return mmh::fail("hipFuncSetAttribute(norm_bwd_plane): %s", hipGetErrorString(e)); \
            attr_done = true;                                                                                       \
        }                                                                                                           \
        hipLaunchKernelGGL(kfn, grid, dim3(FT), NR * FT * 16, mmh::as_stream(s), g, static_cast<const uint8_t*>(out), \
                           x, static_cast<const float*>(mean), static_cast<const float*>(invstd),                   \
                           static_cast<const float*>(gamma), (float)(1.0 / count), (int)rows, C / 8, l2,            \
                           1.f / (1.f - drop_p), dx, static_cast<float*>(s1), static_cast<float*>(s2));             \
    }
#define MMH_PL2(GW, DW, NR)                                                                                         \
    {                                                                                                               \
        if (h16) { if (masked) MMH_PL(GW, DW, NR, true, true) else MMH_PL(GW, DW, NR, true, false) }                \
        else { if (masked) MMH_PL(GW, DW, NR, false, true) else MMH_PL(GW, DW, NR, false, false) }                  \
    }
    const bool h16 = x_dtype == MMH_FP16;
    if (gw && dw) MMH_PL2(true, true, 8)
    else if (gw) MMH_PL2(true, false, 8)
    else if (dw) MMH_PL2(false, true, 4)
    else MMH_PL2(false, false, 4)
#undef MMH_PL2
#undef MMH_PL
    return mmh::check_launch("norm_bwd_fused");
}

int mmh_dropout_bits(int64_t n, float drop_p, uint64_t seed, const void* mask, void* bits, mmh_stream_t s) {
    MMH_REQUIRE(n > 0 && n % 8 == 0 && bits && drop_p > 0.f && drop_p < 1.f, "mmh_dropout_bits: bad arguments");
    const int64_t n8 = n / 8;
    hipLaunchKernelGGL(dropout_bits_kernel, dim3((unsigned)mmh::cdiv(n8, TPB)), dim3(TPB), 0, mmh::as_stream(s), n8,
                       (uint32_t)((double)drop_p * 65536.0), seed, static_cast<const uint8_t*>(mask),
                       static_cast<uint8_t*>(bits), g_dropout_salt);
    return mmh::check_launch("dropout_bits");
}

int mmh_dropout_bits_both(int64_t image_rows, int W, int C, float drop_p, uint64_t seed, const void* mask, void* bits,
                          void* rows, mmh_stream_t s) {
    MMH_REQUIRE(bits && rows && image_rows > 0 && image_rows < (1ll << 31) && W > 0 && C > 0 && C % 8 == 0 &&
                    drop_p > 0.f && drop_p < 1.f,
                "mmh_dropout_bits_both: bad arguments");
    const int nW32 = (W + 31) / 32;
    const size_t lds = (size_t)W * (C / 8);
    const uint32_t thr16 = (uint32_t)((double)drop_p * 65536.0);
    if (lds > 48 * 1024) {      // a row that does not fit LDS: the two stand-alone kernels
        if (int rc = mmh_dropout_bits(image_rows * W * C, drop_p, seed, mask, bits, s)) return rc;
        return mmh_dropout_bits_rows(bits, image_rows, W, C, rows, s);
    }
    hipLaunchKernelGGL(dropout_both_kernel, dim3((unsigned)image_rows), dim3(TPB), lds, mmh::as_stream(s), W, C, nW32,
                       thr16, seed, static_cast<const uint8_t*>(mask), static_cast<uint8_t*>(bits),
                       static_cast<uint32_t*>(rows), g_dropout_salt);
    return mmh::check_launch("dropout_both");
}

int mmh_dropout_bits_rows(const void* bits, int64_t image_rows, int W, int C, void* rows, mmh_stream_t s) {
    MMH_REQUIRE(bits && rows && image_rows > 0 && W > 0 && C > 0 && C % 8 == 0, "mmh_dropout_bits_rows: bad arguments");
    const int nW32 = (W + 31) / 32;
    const int64_t n = image_rows * nW32 * C;
    hipLaunchKernelGGL(dropout_rows_kernel, dim3((unsigned)mmh::cdiv(n, TPB)), dim3(TPB), 0, mmh::as_stream(s),
                       static_cast<const uint8_t*>(bits), image_rows, W, C, nW32, static_cast<uint32_t*>(rows));
    return mmh::check_launch("dropout_rows");
}

int mmh_norm_bwd_reduce_rc(const void* g, const void* x, const void* mean, const void* invstd, const void* scale,
                           const void* shift, const void* dbits, int groups, int64_t rows, int C, int relu,
                           float drop_p, void* s1, void* s2, void* ws, size_t ws_bytes, mmh_stream_t s) {
    if (int rc = check_cols("mmh_norm_bwd_reduce_rc", C)) return rc;
    MMH_REQUIRE(row_geom_ok(C), "mmh_norm_bwd_reduce_rc: C/8 must be a power of two <= 256");
    MMH_REQUIRE(g && x && mean && invstd && scale && shift && s1 && s2 && ws && groups > 0 && rows > 0 &&
                    drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f) == (dbits == nullptr),
                "mmh_norm_bwd_reduce_rc: bad arguments");
    MMH_REQUIRE(ws_bytes >= mmh_norm_bwd_ws_bytes(groups, rows, C), "mmh_norm_bwd_reduce_rc: workspace too small");
    ColGeom cg = col_geom(groups, rows, C);
    hipStream_t st = mmh::as_stream(s);
    const int c8 = C / 8;
    hipLaunchKernelGGL((col_reduce_partial_v2<1, false, false, true>), dim3(cg.chunks, groups), dim3(TPB), 0, st, g,
                       false, static_cast<const uint8_t*>(dbits), x, false, static_cast<const float*>(mean),
                       static_cast<const float*>(invstd), rows, C, c8, TPB / c8, cg.chunks, cg.rows_per_chunk, 3,
                       1.f / (1.f - drop_p), static_cast<float*>(ws), static_cast<const float*>(scale),
                       static_cast<const float*>(shift), relu);
    MMH_COL_FINAL(groups * C, cg.chunks, st,
                       static_cast<const float*>(ws), groups, C, cg.chunks, 2,
                       static_cast<float*>(s1), static_cast<float*>(s2), 0);
    return mmh::check_launch("norm_bwd_reduce_rc");
}

int mmh_norm_bwd_apply_rc(const void* g, const void* x, const void* mean, const void* invstd, const void* gamma,
                          const void* s1, const void* s2, const void* scale, const void* shift, const void* dbits,
                          double count, int groups, int64_t rows, int C, int relu, float drop_p, void* dx,
                          mmh_stream_t s) {
    if (int rc = check_cols("mmh_norm_bwd_apply_rc", C)) return rc;
    MMH_REQUIRE(row_geom_ok(C), "mmh_norm_bwd_apply_rc: C/8 must be a power of two <= 256");
    MMH_REQUIRE(g && x && mean && invstd && s1 && s2 && scale && shift && dx && count > 0 && groups > 0 && rows > 0 &&
                    drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f) == (dbits == nullptr),
                "mmh_norm_bwd_apply_rc: bad arguments");
    const RowGeom rg = row_geom(groups, rows, C);
    hipLaunchKernelGGL((norm_bwd_apply_v2<false, false, false, true>), dim3(rg.chunks, groups), dim3(TPB), 0,
                       mmh::as_stream(s), g, false, static_cast<const uint8_t*>(dbits), x, false,
                       static_cast<const float*>(mean), static_cast<const float*>(invstd),
                       static_cast<const float*>(gamma), static_cast<const float*>(s1), static_cast<const float*>(s2),
                       (float)(1.0 / count), rows, rg, 3, 1.f / (1.f - drop_p), dx, false,
                       static_cast<const float*>(scale), static_cast<const float*>(shift), relu);
    return mmh::check_launch("norm_bwd_apply_rc");
}

size_t mmh_colsum_ws_bytes(int64_t rows, int C) {
    if (rows <= 0 || C <= 0 || C % 4) return 0;
    ColGeom g = col_geom(1, rows, C);
    return (size_t)g.chunks * C * sizeof(float);
}

int mmh_colsum(const void* x, int64_t rows, int C, int cs, void* out, void* ws, size_t ws_bytes,
               int accumulate, int x_dtype, mmh_stream_t s) {
    if (int rc = check_cols("mmh_colsum", C)) return rc;
    MMH_REQUIRE(is_dtype(x_dtype), "mmh_colsum: x_dtype must be MMH_F32 | MMH_BF16 | MMH_FP16");
    MMH_REQUIRE(x && out && ws && rows > 0 && cs >= C && cs % 4 == 0, "mmh_colsum: bad arguments");
    MMH_REQUIRE(ws_bytes >= mmh_colsum_ws_bytes(rows, C), "mmh_colsum: workspace too small");
    ColGeom cg = col_geom(1, rows, C);
    hipStream_t st = mmh::as_stream(s);
    if (mmh::g_pw_v2 && row_geom_ok(C) && cs == C) {
        const int c8 = C / 8;
        if (x_dtype != MMH_F32)
            hipLaunchKernelGGL((col_reduce_partial_v2<0, true, false>), dim3(cg.chunks, 1), dim3(TPB), 0, st, x,
                               x_dtype == MMH_FP16, nullptr, nullptr, false, nullptr, nullptr, rows, C, c8,
                               TPB / c8, cg.chunks, cg.rows_per_chunk, 0, 1.f, static_cast<float*>(ws));
        else
            hipLaunchKernelGGL((col_reduce_partial_v2<0, false, false>), dim3(cg.chunks, 1), dim3(TPB), 0, st, x,
                               false, nullptr, nullptr, false, nullptr, nullptr, rows, C, c8, TPB / c8,
                               cg.chunks, cg.rows_per_chunk, 0, 1.f, static_cast<float*>(ws));
    } else
    hipLaunchKernelGGL((col_reduce_partial<0>), dim3(cg.chunks, 1), dim3(TPB), 0, st,
                       x, x_dtype, nullptr, nullptr, 0, nullptr, nullptr, rows, C, cs,
                       0, 1.f, cg, static_cast<float*>(ws));
    MMH_COL_FINAL(C, cg.chunks, st,
                       static_cast<const float*>(ws), 1, C, cg.chunks, 1, static_cast<float*>(out),
                       nullptr, accumulate);
    return mmh::check_launch("colsum");
}

int mmh_act_bwd(const void* g, const void* y, void* dx, int64_t n, int act, mmh_stream_t s) {
    MMH_REQUIRE(g && y && dx && n > 0 && n % 4 == 0, "mmh_act_bwd: bad arguments");
    MMH_REQUIRE(act == MMH_ACT_RELU || act == MMH_ACT_TANH, "mmh_act_bwd: act must be relu or tanh");
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n / 4)), dim3(TPB), 0, mmh::as_stream(s),
                       static_cast<const float*>(g), static_cast<const float*>(y),
                       static_cast<float*>(dx), n / 4, act);
    return mmh::check_launch("act_bwd");
}

int mmh_act_bwd_lp16(const void* g, const void* y, int64_t n, int act, int dtype, void* out16, mmh_stream_t s) {
    MMH_REQUIRE(g && y && out16 && n > 0 && n % 8 == 0 && (dtype == MMH_BF16 || dtype == MMH_FP16),
                "mmh_act_bwd_lp16: bad arguments (n %% 8 == 0, dtype MMH_BF16 | MMH_FP16)");
    MMH_REQUIRE(act == MMH_ACT_RELU || act == MMH_ACT_TANH, "mmh_act_bwd_lp16: act must be relu or tanh");
    hipLaunchKernelGGL(act_bwd_lp16_kernel, dim3(grid_for(n / 8)), dim3(TPB), 0, mmh::as_stream(s),
                       static_cast<const float*>(g), static_cast<const float*>(y), out16, n / 8, act,
                       dtype == MMH_FP16 ? 1 : 0);
    return mmh::check_launch("act_bwd_lp16");
}

int mmh_act_bwd_lp16_io(const void* g, int g_is16, const void* y, int y_is16, int64_t n, int act, int dtype, void* out16,
                        mmh_stream_t s) {
    MMH_REQUIRE(g && y && out16 && n > 0 && n % 8 == 0 && (dtype == MMH_BF16 || dtype == MMH_FP16),
                "mmh_act_bwd_lp16_io: bad arguments (n %% 8 == 0, dtype MMH_BF16 | MMH_FP16)");
    MMH_REQUIRE(act == MMH_ACT_RELU || act == MMH_ACT_TANH, "mmh_act_bwd_lp16_io: act must be relu or tanh");
    const dim3 grid(grid_for(n / 8));
    const int h16 = dtype == MMH_FP16 ? 1 : 0;
    hipStream_t st = mmh::as_stream(s);
    if (g_is16 && y_is16) hipLaunchKernelGGL((act_bwd_lp16_io_kernel<true, true>), grid, dim3(TPB), 0, st, g, y, out16, n / 8, act, h16);
    else if (g_is16) hipLaunchKernelGGL((act_bwd_lp16_io_kernel<true, false>), grid, dim3(TPB), 0, st, g, y, out16, n / 8, act, h16);
    else if (y_is16) hipLaunchKernelGGL((act_bwd_lp16_io_kernel<false, true>), grid, dim3(TPB), 0, st, g, y, out16, n / 8, act, h16);
    else hipLaunchKernelGGL((act_bwd_lp16_io_kernel<false, false>), grid, dim3(TPB), 0, st, g, y, out16, n / 8, act, h16);
    return mmh::check_launch("act_bwd_lp16_io");
}

int mmh_patblock_gate_fwd(const void* x1, const void* s1, const void* s2, const void* s3, void* out,
                          void* x2n, void* x3n, int64_t rows, int C, int cat_dtype, int s23_dtype,
                          mmh_stream_t s) {
    if (int rc = check_cols("mmh_patblock_gate_fwd", C)) return rc;
    MMH_REQUIRE(is_dtype(cat_dtype) && is_dtype(s23_dtype),
                "mmh_patblock_gate_fwd: cat_dtype / s23_dtype must be MMH_F32 | MMH_BF16 | MMH_FP16");
    MMH_REQUIRE(x1 && s1 && s2 && s3 && out && rows > 0 && ((x2n == nullptr) == (x3n == nullptr)),
                "mmh_patblock_gate_fwd: bad arguments");
    const int64_t n4 = rows * (C / 4);
    hipLaunchKernelGGL(gate_fwd_kernel, dim3(grid_for(n4)), dim3(TPB), 0, mmh::as_stream(s),
                       static_cast<const float*>(x1), static_cast<const float*>(s1), s2, s3,
                       static_cast<float*>(out), x2n, x3n, n4, C / 4, cat_dtype, s23_dtype);
    return mmh::check_launch("gate_fwd");
}

int mmh_patblock_gate_bwd(const void* g_out, const void* g_x2n, const void* g_x3n, const void* s1,
                          const void* s2, const void* s3, void* g_x1, void* g_s1, void* g_s2,
                          void* g_s3, int64_t rows, int C, int gcat_dtype, int s23_dtype, int gs23_dtype,
                          mmh_stream_t s) {
    if (int rc = check_cols("mmh_patblock_gate_bwd", C)) return rc;
    MMH_REQUIRE(is_dtype(gcat_dtype) && is_dtype(s23_dtype) && is_dtype(gs23_dtype),
                "mmh_patblock_gate_bwd: bad gcat_dtype / s23_dtype / gs23_dtype");
    MMH_REQUIRE(s1 && s2 && s3 && g_x1 && g_s1 && g_s2 && g_s3 && rows > 0,
                "mmh_patblock_gate_bwd: bad arguments");
    const int64_t n4 = rows * (C / 4);
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(grid_for(n4)), dim3(TPB), 0, mmh::as_stream(s),
                       static_cast<const float*>(g_out), g_x2n, g_x3n, static_cast<const float*>(s1),
                       s2, s3, static_cast<float*>(g_x1), static_cast<float*>(g_s1), g_s2, g_s3, n4,
                       C / 4, gcat_dtype, s23_dtype, gs23_dtype);
    return mmh::check_launch("gate_bwd");
}

int mmh_patblock_gate_norm_supported(int groups, int64_t rows, int C) {
    return (mmh::g_pw_v2 && groups > 0 && rows > 0 && row_geom_ok(C)) ? 1 : 0;
}

int mmh_patblock_gate_norm_fwd(const void* x1, const void* y2, const void* scale, const void* shift, const void* s2,
                               const void* s3, void* out, void* x2n, void* x3n, int groups, int64_t rows, int C,
                               int y_dtype, int cat_dtype, int s23_dtype, mmh_stream_t s) {
    if (int rc = check_cols("mmh_patblock_gate_norm_fwd", C)) return rc;
    MMH_REQUIRE(mmh_patblock_gate_norm_supported(groups, rows, C) && is_dtype(cat_dtype) && is_dtype(s23_dtype) &&
                    (y_dtype == MMH_BF16 || y_dtype == MMH_FP16),
                "mmh_patblock_gate_norm_fwd: C / 8 a power of two <= 256, a 16-bit y2");
    MMH_REQUIRE(x1 && y2 && scale && shift && s2 && s3 && out && ((x2n == nullptr) == (x3n == nullptr)),
                "mmh_patblock_gate_norm_fwd: bad arguments");
    const RowGeom rg = row_geom(groups, rows, C);
    const dim3 grid(rg.chunks, groups);
#define MMH_GNF(SW)                                                                                                  \
    hipLaunchKernelGGL((gate_norm_fwd_kernel<SW>), grid, dim3(TPB), 0, mmh::as_stream(s), static_cast<const float*>(x1), \
                       y2, y_dtype == MMH_FP16, static_cast<const float*>(scale), static_cast<const float*>(shift), s2, \
                       s3, s23_dtype == MMH_FP16, static_cast<float*>(out), x2n, x3n, cat_dtype, rows, rg)
    if (s23_dtype != MMH_F32) MMH_GNF(true); else MMH_GNF(false);
#undef MMH_GNF
    return mmh::check_launch("gate_norm_fwd");
}

int mmh_patblock_gate_norm_bwd(const void* g_out, const void* g_x2n, const void* g_x3n, const void* y2, const void* scale,
                               const void* shift, const void* mean, const void* invstd, const void* s2, const void* s3,
                               void* g_x1, void* g_s1, void* g_s2, void* g_s3, void* sum1, void* sum2, void* ws,
                               size_t ws_bytes, int groups, int64_t rows, int C, int y_dtype, int gcat_dtype,
                               int s23_dtype, mmh_stream_t s) {
    if (int rc = check_cols("mmh_patblock_gate_norm_bwd", C)) return rc;
    MMH_REQUIRE(mmh_patblock_gate_norm_supported(groups, rows, C) && is_dtype(gcat_dtype) && is_dtype(s23_dtype) &&
                    (y_dtype == MMH_BF16 || y_dtype == MMH_FP16),
                "mmh_patblock_gate_norm_bwd: C / 8 a power of two <= 256, a 16-bit y2");
    MMH_REQUIRE(y2 && scale && shift && mean && invstd && s2 && s3 && g_x1 && g_s1 && g_s2 && g_s3 && sum1 && sum2 && ws,
                "mmh_patblock_gate_norm_bwd: bad arguments");
    MMH_REQUIRE(ws_bytes >= mmh_norm_bwd_ws_bytes(groups, rows, C), "mmh_patblock_gate_norm_bwd: workspace too small");
    const ColGeom cg = col_geom(groups, rows, C);       // the chunks of mmh_norm_bwd_reduce: the same partial sums
    const int c8 = C / 8;
    hipStream_t st = mmh::as_stream(s);
    const dim3 grid(cg.chunks, groups);
#define MMH_GNB(GCW, SW)                                                                                             \
    hipLaunchKernelGGL((gate_norm_bwd_kernel<GCW, SW>), grid, dim3(TPB), 0, st, static_cast<const float*>(g_out), g_x2n, \
                       g_x3n, gcat_dtype == MMH_FP16, y2, y_dtype == MMH_FP16, static_cast<const float*>(scale),     \
                       static_cast<const float*>(shift), static_cast<const float*>(mean),                           \
                       static_cast<const float*>(invstd), s2, s3, s23_dtype == MMH_FP16, static_cast<float*>(g_x1),  \
                       static_cast<float*>(g_s1), g_s2, g_s3, rows, C, c8, TPB / c8, cg.chunks, cg.rows_per_chunk,   \
                       static_cast<float*>(ws))
    const bool gcw = gcat_dtype != MMH_F32, sw = s23_dtype != MMH_F32;
    if (gcw && sw) MMH_GNB(true, true);
    else if (gcw) MMH_GNB(true, false);
    else if (sw) MMH_GNB(false, true);
    else MMH_GNB(false, false);
#undef MMH_GNB
    MMH_COL_FINAL(groups * C, cg.chunks, st, static_cast<const float*>(ws), groups, C, cg.chunks, 2,
                  static_cast<float*>(sum1), static_cast<float*>(sum2), 0);
    return mmh::check_launch("gate_norm_bwd");
}

size_t mmh_reduce_ws_bytes(int64_t n) { return (size_t)grid_for(n / 4, 4096) * sizeof(float); }

static int loss_fwd(int mode, const void* a, const void* b, int64_t n, float target, float weight,
                    double denom, void* out, void* ws, size_t ws_bytes, mmh_stream_t s) {
    MMH_REQUIRE(a && out && ws && n > 0 && n % 4 == 0 && denom > 0, "loss_fwd: bad arguments");
    const int blocks = grid_for(n / 4, 4096);
    MMH_REQUIRE(ws_bytes >= blocks * sizeof(float), "loss_fwd: workspace too small");
    hipStream_t st = mmh::as_stream(s);
    if (mode == 0)
        hipLaunchKernelGGL((loss_partial_kernel<0>), dim3(blocks), dim3(TPB), 0, st,
                           static_cast<const float*>(a), nullptr, n, target, static_cast<float*>(ws));
    else if (mode == 1)
        hipLaunchKernelGGL((loss_partial_kernel<1>), dim3(blocks), dim3(TPB), 0, st,
                           static_cast<const float*>(a), static_cast<const float*>(b), n, 0.f,
                           static_cast<float*>(ws));
    else
        hipLaunchKernelGGL((loss_partial_kernel<2>), dim3(blocks), dim3(TPB), 0, st,
                           static_cast<const float*>(a), static_cast<const float*>(b), n, 0.f,
                           static_cast<float*>(ws));
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(TPB), 0, st, static_cast<const float*>(ws),
                       blocks, (double)weight / denom, static_cast<float*>(out));
    return mmh::check_launch("loss_fwd");
}

int mmh_bce_logits_fwd(const void* x, int64_t n, float target, float weight, double denom, void* out,
                       void* ws, size_t ws_bytes, mmh_stream_t s) {
    return loss_fwd(0, x, nullptr, n, target, weight, denom, out, ws, ws_bytes, s);
}

int mmh_bce_logits_bwd(const void* x, int64_t n, float target, float weight, double denom,
                       const void* gscalar, void* dx, mmh_stream_t s) {
    MMH_REQUIRE(x && gscalar && dx && n > 0 && n % 4 == 0 && denom > 0, "mmh_bce_logits_bwd: bad arguments");
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(grid_for(n / 4)), dim3(TPB), 0, mmh::as_stream(s),
                       static_cast<const float*>(x), n / 4, target, (float)((double)weight / denom),
                       static_cast<const float*>(gscalar), static_cast<float*>(dx));
    return mmh::check_launch("bce_bwd");
}

int mmh_l1_fwd(const void* a, const void* b, int64_t n, float weight, double denom, void* out,
               void* ws, size_t ws_bytes, mmh_stream_t s) {
    MMH_REQUIRE(b != nullptr, "mmh_l1_fwd: NULL buffer");
    return loss_fwd(1, a, b, n, 0.f, weight, denom, out, ws, ws_bytes, s);
}

int mmh_l1_bwd(const void* a, const void* b, int64_t n, float weight, double denom,
               const void* gscalar, void* da, mmh_stream_t s) {
    MMH_REQUIRE(a && b && gscalar && da && n > 0 && n % 4 == 0 && denom > 0, "mmh_l1_bwd: bad arguments");
    hipLaunchKernelGGL(l1_bwd_kernel, dim3(grid_for(n / 4)), dim3(TPB), 0, mmh::as_stream(s),
                       static_cast<const float*>(a), static_cast<const float*>(b), n / 4,
                       (float)((double)weight / denom), static_cast<const float*>(gscalar),
                       static_cast<float*>(da));
    return mmh::check_launch("l1_bwd");
}

int mmh_l1_fwd_lp16(const void* a16, const void* b16, int64_t n, float weight, double denom, int dtype, void* out,
                    void* ws, size_t ws_bytes, mmh_stream_t s) {
    MMH_REQUIRE(a16 && b16 && out && ws && n > 0 && n % 8 == 0 && denom > 0 && (dtype == MMH_BF16 || dtype == MMH_FP16),
                "mmh_l1_fwd_lp16: bad arguments (n %% 8 == 0, dtype MMH_BF16 | MMH_FP16)");
    const int blocks = grid_for(n / 8, 4096);
    MMH_REQUIRE(ws_bytes >= blocks * sizeof(float), "mmh_l1_fwd_lp16: workspace too small (mmh_reduce_ws_bytes(n))");
    hipStream_t st = mmh::as_stream(s);
    hipLaunchKernelGGL(l1_partial_lp16_kernel, dim3(blocks), dim3(TPB), 0, st, a16, b16, n / 8, dtype == MMH_FP16 ? 1 : 0,
                       static_cast<float*>(ws));
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(TPB), 0, st, static_cast<const float*>(ws), blocks,
                       (double)weight / denom, static_cast<float*>(out));
    return mmh::check_launch("l1_fwd_lp16");
}

int mmh_l1_relu_bwd_lp16(const void* a16, const void* b16, int64_t n, float weight, double denom, const void* gscalar,
                         int dtype, void* out16, mmh_stream_t s) {
    MMH_REQUIRE(a16 && b16 && gscalar && out16 && n > 0 && n % 8 == 0 && denom > 0 && (dtype == MMH_BF16 || dtype == MMH_FP16),
                "mmh_l1_relu_bwd_lp16: bad arguments (n %% 8 == 0, dtype MMH_BF16 | MMH_FP16)");
    hipLaunchKernelGGL(l1_relu_bwd_lp16_kernel, dim3(grid_for(n / 8)), dim3(TPB), 0, mmh::as_stream(s), a16, b16, n / 8,
                       dtype == MMH_FP16 ? 1 : 0, (float)((double)weight / denom), static_cast<const float*>(gscalar), out16);
    return mmh::check_launch("l1_relu_bwd_lp16");
}

int mmh_maxpool2x2_fwd(const void* x, int B, int H, int W, int C, void* y, mmh_stream_t s) {
    MMH_REQUIRE(x && y && B > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 4 == 0,
                "mmh_maxpool2x2_fwd: NHWC fp32, even H and W, C %% 4 == 0");
    const int64_t total = (int64_t)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(grid_for(total)), dim3(TPB), 0, mmh::as_stream(s),
                       static_cast<const float*>(x), static_cast<float*>(y), B, H, W, C / 4);
    return mmh::check_launch("maxpool2_fwd");
}

int mmh_maxpool2x2_bwd(const void* x, const void* g, int B, int H, int W, int C, void* dx, mmh_stream_t s) {
    MMH_REQUIRE(x && g && dx && B > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 4 == 0,
                "mmh_maxpool2x2_bwd: NHWC fp32, even H and W, C %% 4 == 0");
    const int64_t total = (int64_t)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for(total)), dim3(TPB), 0, mmh::as_stream(s),
                       static_cast<const float*>(x), static_cast<const float*>(g), static_cast<float*>(dx), B, H, W, C / 4);
    return mmh::check_launch("maxpool2_bwd");
}

int mmh_mse_fwd(const void* a, const void* b, int64_t n, float weight, double denom, void* out,
                void* ws, size_t ws_bytes, mmh_stream_t s) {
    MMH_REQUIRE(b != nullptr, "mmh_mse_fwd: NULL buffer");
    return loss_fwd(2, a, b, n, 0.f, weight, denom, out, ws, ws_bytes, s);
}

int mmh_mse_bwd(const void* a, const void* b, int64_t n, float weight, double denom,
                const void* gscalar, void* da, mmh_stream_t s) {
    MMH_REQUIRE(a && b && gscalar && da && n > 0 && n % 4 == 0 && denom > 0, "mmh_mse_bwd: bad arguments");
    hipLaunchKernelGGL(mse_bwd_kernel, dim3(grid_for(n / 4)), dim3(TPB), 0, mmh::as_stream(s),
                       static_cast<const float*>(a), static_cast<const float*>(b), n / 4,
                       (float)((double)weight / denom), static_cast<const float*>(gscalar),
                       static_cast<float*>(da));
    return mmh::check_launch("mse_bwd");
}

int mmh_grad_nonfinite(const void* g, int64_t n, const void* flag_in, void* flag_out, void* own_out,
                       mmh_stream_t s) {
    MMH_REQUIRE(g && flag_out && n > 0, "mmh_grad_nonfinite: bad arguments");
    MMH_REQUIRE((reinterpret_cast<uintptr_t>(g) & 15) == 0, "mmh_grad_nonfinite: g must be 16-byte aligned");
    hipStream_t st = mmh::as_stream(s);
    hipError_t e = flag_in ? hipMemcpyAsync(flag_out, flag_in, sizeof(int), hipMemcpyDeviceToDevice, st)
                           : hipMemsetAsync(flag_out, 0, sizeof(int), st);
    if (e == hipSuccess && own_out) e = hipMemsetAsync(own_out, 0, sizeof(int), st);
    if (e != hipSuccess) return mmh::fail("mmh_grad_nonfinite: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(grad_nonfinite_kernel, dim3(grid_for(n / 4 + 1, 4096)), dim3(TPB), 0, st,
                       static_cast<const float*>(g), n / 4, n, static_cast<int*>(flag_out),
                       static_cast<int*>(own_out));
    return mmh::check_launch("grad_nonfinite");
}

int mmh_loss_scale_update(void* state, const void* overflow, float growth, float backoff, int interval,
                          float min_scale, float max_scale, mmh_stream_t s) {
    MMH_REQUIRE(state && overflow && growth >= 1.f && backoff > 0.f && backoff <= 1.f && interval > 0 &&
                    min_scale > 0.f && max_scale >= min_scale,
                "mmh_loss_scale_update: bad arguments");
    hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(64), 0, mmh::as_stream(s),
                       static_cast<float*>(state), static_cast<const int*>(overflow), growth, backoff,
                       (float)interval, min_scale, max_scale);
    return mmh::check_launch("loss_scale_update");
}

int mmh_adam_step(void* p, const void* g, void* m, void* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, int step, float grad_scale, const void* skip_flag,
                  const void* loss_scale, mmh_stream_t s) {
    MMH_REQUIRE(p && g && m && v && n > 0 && step >= 1, "mmh_adam_step: bad arguments");
    const double bc1 = 1.0 - std::pow((double)beta1, step);
    const double bc2 = 1.0 - std::pow((double)beta2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 8192)), dim3(TPB), 0, mmh::as_stream(s),
                       static_cast<float*>(p), static_cast<const float*>(g), static_cast<float*>(m),
                       static_cast<float*>(v), n, beta1, beta2, eps, (float)((double)lr / bc1),
                       (float)(1.0 / std::sqrt(bc2)), grad_scale, static_cast<const int*>(skip_flag),
                       static_cast<const float*>(loss_scale));
    return mmh::check_launch("adam");
}

int mmh_adam_step_dev(void* p, const void* g, void* m, void* v, int64_t n, const void* lr, float beta1, float beta2,
                      float eps, void* step, float grad_scale, const void* skip_flag, const void* loss_scale, void* coef,
                      mmh_stream_t s) {
    MMH_REQUIRE(p && g && m && v && n > 0 && lr && step && coef, "mmh_adam_step_dev: bad arguments");
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, mmh::as_stream(s), static_cast<int*>(step),
                       static_cast<const float*>(lr), static_cast<const int*>(skip_flag), beta1, beta2, static_cast<float*>(coef));
    hipLaunchKernelGGL(adam_dev_kernel, dim3(grid_for(n, 8192)), dim3(TPB), 0, mmh::as_stream(s),
                       static_cast<float*>(p), static_cast<const float*>(g), static_cast<float*>(m),
                       static_cast<float*>(v), n, beta1, beta2, eps, static_cast<const float*>(coef), grad_scale,
                       static_cast<const int*>(skip_flag), static_cast<const float*>(loss_scale));
    return mmh::check_launch("adam_dev");
}

int mmh_set_dropout_salt(const void* salt_u64) {
    g_dropout_salt = static_cast<const uint64_t*>(salt_u64);
    return 0;
}

int mmh_u64_add(void* value_u64, uint64_t inc, mmh_stream_t s) {
    MMH_REQUIRE(value_u64, "mmh_u64_add: NULL");
    hipLaunchKernelGGL(u64_add_kernel, dim3(1), dim3(64), 0, mmh::as_stream(s), static_cast<uint64_t*>(value_u64), inc);
    return mmh::check_launch("u64_add");
}

int mmh_pool_exchange(void* pool, const void* images, void* out, const void* src_idx, const void* dst_idx, int B,
                      int64_t elems_per_image, mmh_stream_t s) {
    MMH_REQUIRE(pool && images && out && src_idx && dst_idx && B > 0 && elems_per_image > 0 && elems_per_image % 4 == 0,
                "mmh_pool_exchange: bad arguments (elements per image must be a multiple of 4)");
    const int64_t n4 = elems_per_image / 4;
    const dim3 grid((unsigned)std::min<int64_t>(mmh::cdiv(n4, TPB), 256), (unsigned)B);
    hipLaunchKernelGGL(pool_gather_kernel, grid, dim3(TPB), 0, mmh::as_stream(s), static_cast<const float*>(pool),
                       static_cast<const float*>(images), static_cast<float*>(out), static_cast<const int*>(src_idx), n4);
    hipLaunchKernelGGL(pool_scatter_kernel, grid, dim3(TPB), 0, mmh::as_stream(s), static_cast<float*>(pool),
                       static_cast<const float*>(images), static_cast<const int*>(dst_idx), n4);
    return mmh::check_launch("pool_exchange");
}

int mmh_pack_nhwc(const mmh_plane_src* srcs, int nsrc, void* nhwc, int B, int H, int W, int Cd,
                  int dir, mmh_stream_t s) {
    MMH_REQUIRE(srcs && nhwc && nsrc >= 1 && nsrc <= 4 && B > 0 && H > 0 && W > 0 && Cd > 0 && Cd % 4 == 0 &&
                    (reinterpret_cast<uintptr_t>(nhwc) & 15) == 0,
                "mmh_pack_nhwc: bad arguments (Cd %% 4 == 0, 16-byte aligned NHWC buffer)");
    PackArgs a{};
    a.nsrc = nsrc;
    int tot = 0;
    for (int i = 0; i < nsrc; ++i) { a.s[i] = srcs[i]; tot += srcs[i].C; }
    MMH_REQUIRE(tot <= Cd, "mmh_pack_nhwc: %d source channels > Cd=%d", tot, Cd);
    hipLaunchKernelGGL(pack_nhwc_kernel, dim3(grid_for((int64_t)B * H * W)), dim3(TPB), 0,
                       mmh::as_stream(s), a, static_cast<float*>(nhwc), B, H, W, Cd, dir);
    return mmh::check_launch("pack_nhwc");
}

int mmh_pack_nhwc_lp16(const mmh_plane_src* srcs, int nsrc, void* nhwc, void* out16, int B, int H, int W, int Cd, int C8,
                       int dtype, mmh_stream_t s) {
    MMH_REQUIRE(srcs && (nhwc || out16) && nsrc >= 1 && nsrc <= 4 && B > 0 && H > 0 && W > 0 && Cd > 0 && Cd % 4 == 0 &&
                    Cd <= 56 && (reinterpret_cast<uintptr_t>(nhwc) & 15) == 0 && (reinterpret_cast<uintptr_t>(out16) & 15) == 0,
                "mmh_pack_nhwc_lp16: bad arguments (Cd %% 4 == 0, Cd <= 56, 16-byte aligned buffers)");
    MMH_REQUIRE(!out16 || ((dtype == MMH_BF16 || dtype == MMH_FP16) && C8 % 8 == 0 && C8 >= Cd && C8 <= 56),
                "mmh_pack_nhwc_lp16: the 16-bit copy needs a 16-bit dtype and Cd <= C8 <= 56, C8 %% 8 == 0");
    PackArgs a{};
    a.nsrc = nsrc;
    int tot = 0;
    for (int i = 0; i < nsrc; ++i) { a.s[i] = srcs[i]; tot += srcs[i].C; }
    MMH_REQUIRE(tot <= Cd, "mmh_pack_nhwc_lp16: %d source channels > Cd=%d", tot, Cd);
    const int64_t total = (int64_t)B * H * W;
    const int Cm = out16 && C8 > Cd ? C8 : Cd;
    const size_t lds = (size_t)PACK_TP * (Cm | 1) * sizeof(float);
    const int blocks = (int)std::min<int64_t>(mmh::cdiv(total, PACK_TP), 256 * 8);
    hipLaunchKernelGGL(pack_nhwc_tile_kernel, dim3(blocks), dim3(PACK_TP), lds, mmh::as_stream(s), a,
                       static_cast<float*>(nhwc), out16, total, H, W, Cd, out16 ? C8 : 0, dtype == MMH_FP16 ? 1 : 0);
    return mmh::check_launch("pack_nhwc_tile");
}

int mmh_pose_heatmaps(const void* uv, int n_maps, int H, int W, double sigma, void* out,
                      mmh_stream_t s) {
    MMH_REQUIRE(uv && out && n_maps > 0 && H > 0 && W > 0 && sigma > 0, "mmh_pose_heatmaps: bad arguments");
    hipLaunchKernelGGL(pose_heatmap_kernel, dim3(grid_for((int64_t)n_maps * H * W)), dim3(TPB), 0,
                       mmh::as_stream(s), static_cast<const double*>(uv), n_maps, H, W, sigma,
                       static_cast<float*>(out));
    return mmh::check_launch("pose_heatmaps");
}

int mmh_decode_inputs(const void* img1, const void* img2, const void* dep1, const void* dep2,
                      const void* uv1, const void* uv2, int B, int H, int W, double sigma,
                      void* x_h1, void* x_h2, void* x_p, void* x_d, mmh_stream_t s) {
    MMH_REQUIRE(img1 && img2 && dep1 && dep2 && uv1 && uv2 && x_h1 && x_h2 && x_p && x_d,
                "mmh_decode_inputs: NULL buffer");
    MMH_REQUIRE(B > 0 && H > 0 && W > 0 && sigma > 0, "mmh_decode_inputs: bad shape");
    hipLaunchKernelGGL(decode_inputs_kernel, dim3(grid_for((int64_t)B * H * W)), dim3(TPB), 0,
                       mmh::as_stream(s), static_cast<const uint8_t*>(img1),
                       static_cast<const uint8_t*>(img2), static_cast<const uint8_t*>(dep1),
                       static_cast<const uint8_t*>(dep2), static_cast<const double*>(uv1),
                       static_cast<const double*>(uv2), B, H, W, sigma, static_cast<float*>(x_h1),
                       static_cast<float*>(x_h2), static_cast<float*>(x_p), static_cast<float*>(x_d));
    return mmh::check_launch("decode_inputs");
}

int mmh_prep_weights_bf16(const void* w, int taps, int Cin, int Cout, void* w_plain, void* w_t,
                          mmh_stream_t s) {
    MMH_REQUIRE(w && (w_plain || w_t) && taps > 0 && Cin > 0 && Cout > 0, "mmh_prep_weights_bf16: bad arguments");
    hipLaunchKernelGGL(prep_weights_bf16_kernel<__bf16>, dim3(grid_for((int64_t)taps * Cin * Cout)), dim3(TPB), 0,
                       mmh::as_stream(s), static_cast<const float*>(w), taps, Cin, Cout,
                       static_cast<__bf16*>(w_plain), static_cast<__bf16*>(w_t));
    return mmh::check_launch("prep_weights_bf16");
}

int mmh_prep_weights_lp16_multi(const void* table, int n, int64_t total_blocks, mmh_stream_t s) {
    MMH_REQUIRE(table && n > 0 && total_blocks > 0 && total_blocks < (1ll << 31), "mmh_prep_weights_lp16_multi: bad arguments");
    hipLaunchKernelGGL(prep_weights_lp16_multi_kernel, dim3((unsigned)total_blocks), dim3(TPB), 0, mmh::as_stream(s),
                       static_cast<const long long*>(table), n);
    return mmh::check_launch("prep_weights_lp16_multi");
}

int mmh_prep_weights_fp16(const void* w, int taps, int Cin, int Cout, void* w_plain, void* w_t,
                          mmh_stream_t s) {
    MMH_REQUIRE(w && (w_plain || w_t) && taps > 0 && Cin > 0 && Cout > 0, "mmh_prep_weights_fp16: bad arguments");
    hipLaunchKernelGGL(prep_weights_bf16_kernel<_Float16>, dim3(grid_for((int64_t)taps * Cin * Cout)), dim3(TPB), 0,
                       mmh::as_stream(s), static_cast<const float*>(w), taps, Cin, Cout,
                       static_cast<_Float16*>(w_plain), static_cast<_Float16*>(w_t));
    return mmh::check_launch("prep_weights_fp16");
}

int mmh_prep_weights_fp16_flat(const void* w, int taps, int Cin, int Cout, void* w_flat, mmh_stream_t s) {
    MMH_REQUIRE(w && w_flat && taps > 0 && Cin > 0 && Cout > 0, "mmh_prep_weights_fp16_flat: bad arguments");
    const int Kpad = (taps * Cin + 63) / 64 * 64;
    hipLaunchKernelGGL(prep_weights_bf16_flat_kernel<_Float16>, dim3(grid_for((int64_t)Cout * Kpad)), dim3(TPB), 0,
                       mmh::as_stream(s), static_cast<const float*>(w), taps, Cin, Cout, Kpad,
                       static_cast<_Float16*>(w_flat));
    return mmh::check_launch("prep_weights_fp16_flat");
}

int mmh_prep_weights_bf16_flat(const void* w, int taps, int Cin, int Cout, void* w_flat,
                               mmh_stream_t s) {
    MMH_REQUIRE(w && w_flat && taps > 0 && Cin > 0 && Cout > 0, "mmh_prep_weights_bf16_flat: bad arguments");
    const int Kpad = (taps * Cin + 63) / 64 * 64;
    hipLaunchKernelGGL(prep_weights_bf16_flat_kernel<__bf16>, dim3(grid_for((int64_t)Cout * Kpad)), dim3(TPB), 0,
                       mmh::as_stream(s), static_cast<const float*>(w), taps, Cin, Cout, Kpad,
                       static_cast<__bf16*>(w_flat));
    return mmh::check_launch("prep_weights_bf16_flat");
}

int mmh_map_to_cord(const void* maps, int n_maps, int H, int W, float threshold, void* cords,
                    mmh_stream_t s) {
    MMH_REQUIRE(maps && cords && n_maps > 0 && H > 0 && W > 0, "mmh_map_to_cord: bad arguments");
    hipLaunchKernelGGL(map_to_cord_kernel, dim3(n_maps), dim3(TPB), 0, mmh::as_stream(s),
                       static_cast<const float*>(maps), H, W, threshold, static_cast<int*>(cords));
    return mmh::check_launch("map_to_cord");
}

}  // extern "C"
