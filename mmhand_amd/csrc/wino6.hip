// Winograd F(6x6, 3x3) transforms (fp32): 64 multiplications per 6x6 output tile instead of 324,
// i.e. 5.06x fewer MFMA flops than the direct 3x3 convolution and 21 % fewer than F(4x4,3x3), and
// a Winograd domain of 64/36 = 1.78x the activation instead of 2.25x.  Interpolation points
// 0, +-1, +-2, +-1/2, inf (the standard matrices).  fp32 rounding error measured against fp64:
// 5e-6 relative L1 at K = 512 (F(4x4,3x3): 2.4e-6, direct fp32: 2.4e-7) - far inside the 1e-3 bar.
//
// Tiles are RAGGED: TH = ceil(H/6), TW = ceil(W/6); the input transform reads zeros beyond the
// (reflect- or zero-) padded image, the output transform writes only pixels inside it, the
// output-gradient transform reads zeros outside it.  So every H, W >= 2 is eligible.
//
//   U  [64][K][N]      = G g G^T              wino6_weights   (flip_transpose: dgrad filter)
//   V  [64][tiles][C]  = B^T d B              wino6_input     (8x8 window, stride 6, pad 1)
//   y                  = A^T M A (+bias, act) wino6_output
//   Yh [64][tiles][C]  = A dY A^T             wino6_dy        (wgrad)
//   dw [3][3][Cin][Cout] = G^T dU G           wino6_dw
// One thread per (tile, 2 channels), like the F(4x4,3x3) kernels in conv_igemm.hip.
#include "common.h"

namespace {

struct F2 { float x, y; };
__device__ __forceinline__ F2 operator+(F2 a, F2 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ F2 operator-(F2 a, F2 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ F2 operator*(float k, F2 a) { return {k * a.x, k * a.y}; }

__device__ __forceinline__ float act6(float v, int act) {
    if (act == MMH_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == MMH_ACT_TANH) return tanhf(v);
    return v;
}

// out[0..7] = B^T in[0..7]
template <typename T>
__device__ __forceinline__ void w6_bt(const T* d, T* o) {
    const T e1 = d[2] - 4.25f * d[4] + d[6], f1 = d[1] - 4.25f * d[3] + d[5];
    const T e2 = 0.25f * d[2] - 1.25f * d[4] + d[6], f2 = 0.5f * d[1] - 2.5f * d[3] + 2.f * d[5];
    const T e3 = 4.f * d[2] - 5.f * d[4] + d[6], f3 = 2.f * d[1] - 2.5f * d[3] + 0.5f * d[5];
    o[0] = d[0] - d[6] + 5.25f * (d[4] - d[2]);
    o[1] = e1 + f1; o[2] = e1 - f1;
    o[3] = e2 + f2; o[4] = e2 - f2;
    o[5] = e3 + f3; o[6] = e3 - f3;
    o[7] = d[7] - d[1] + 5.25f * (d[3] - d[5]);
}
// out[0..5] = A^T in[0..7]
template <typename T>
__device__ __forceinline__ void w6_at(const T* m, T* y) {
    const T s1 = m[1] + m[2], t1 = m[1] - m[2], s2 = m[3] + m[4], t2 = m[3] - m[4], s3 = m[5] + m[6],
            t3 = m[5] - m[6];
    y[0] = m[0] + s1 + s2 + s3;
    y[1] = t1 + 2.f * t2 + 0.5f * t3;
    y[2] = s1 + 4.f * s2 + 0.25f * s3;
    y[3] = t1 + 8.f * t2 + 0.125f * t3;
    y[4] = s1 + 16.f * s2 + 0.0625f * s3;
    y[5] = t1 + 32.f * t2 + 0.03125f * t3 + m[7];
}
// out[0..7] = A in[0..5]
template <typename T>
__device__ __forceinline__ void w6_a(const T* y, T* o) {
    const T e = y[0] + y[2] + y[4], f = y[1] + y[3] + y[5];
    const T e2 = y[0] + 4.f * y[2] + 16.f * y[4], f2 = 2.f * y[1] + 8.f * y[3] + 32.f * y[5];
    const T e3 = y[0] + 0.25f * y[2] + 0.0625f * y[4], f3 = 0.5f * y[1] + 0.125f * y[3] + 0.03125f * y[5];
    o[0] = y[0];
    o[1] = e + f; o[2] = e - f;
    o[3] = e2 + f2; o[4] = e2 - f2;
    o[5] = e3 + f3; o[6] = e3 - f3;
    o[7] = y[5];
}
// out[0..7] = G in[0..2]
__device__ __forceinline__ void w6_g(const float* g, float* o) {
    const float a = (-2.f / 9.f) * (g[0] + g[2]), b = (-2.f / 9.f) * g[1];
    const float c = (1.f / 90.f) * g[0] + (2.f / 45.f) * g[2], d = (1.f / 45.f) * g[1];
    const float e = (32.f / 45.f) * g[0] + (8.f / 45.f) * g[2], f = (16.f / 45.f) * g[1];
    o[0] = g[0];
    o[1] = a + b; o[2] = a - b;
    o[3] = c + d; o[4] = c - d;
    o[5] = e + f; o[6] = e - f;
    o[7] = g[2];
}
// out[0..2] = G^T in[0..7]
template <typename T>
__device__ __forceinline__ void w6_gt(const T* u, T* w) {
    const T s1 = u[1] + u[2], t1 = u[1] - u[2], s2 = u[3] + u[4], t2 = u[3] - u[4], s3 = u[5] + u[6],
            t3 = u[5] - u[6];
    w[0] = u[0] + (-2.f / 9.f) * s1 + (1.f / 90.f) * s2 + (32.f / 45.f) * s3;
    w[1] = (-2.f / 9.f) * t1 + (1.f / 45.f) * t2 + (16.f / 45.f) * t3;
    w[2] = (-2.f / 9.f) * s1 + (2.f / 45.f) * s2 + (8.f / 45.f) * s3 + u[7];
}

__device__ __forceinline__ void wino6_weights_body(const float* __restrict__ w, float* __restrict__ U, int Cin, int Cout,
                                                   int flip_transpose, int i) {
    const int total = Cin * Cout;
    if (i >= total) return;
    // thread -> (ci, co) with the OUTPUT's fastest index fastest: 64 coalesced plane stores per
    // thread (the 9 filter reads are the strided side when the output is transposed)
    const int ci = flip_transpose ? i % Cin : i / Cout, co = flip_transpose ? i / Cin : i - (i / Cout) * Cout;
    float g[3][3], t[8][3], col[3], o8[8];
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
        w6_g(col, o8);
#pragma unroll
        for (int a = 0; a < 8; ++a) t[a][b] = o8[a];
    }
    const size_t plane = (size_t)Cin * Cout;
    const size_t o = flip_transpose ? (size_t)co * Cin + ci : (size_t)ci * Cout + co;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        w6_g(t[a], o8);
#pragma unroll
        for (int b = 0; b < 8; ++b) U[(size_t)(a * 8 + b) * plane + o] = o8[b];
    }
}

__global__ void wino6_weights_kernel(const float* __restrict__ w, float* __restrict__ U, int Cin, int Cout,
                                     int flip_transpose) {
    wino6_weights_body(w, U, Cin, Cout, flip_transpose, blockIdx.x * blockDim.x + threadIdx.x);
}

// Many filters in one launch (a network's 3x3 filters after an optimizer step: 74 launches of 8-25 us each, every one
// too small to fill the chip).  table[e] = {w, U, Cin, Cout, flip_transpose, first block}: entry e owns blocks
// [first block of e, first block of e + 1) of 256 threads.
__global__ void __launch_bounds__(256) wino6_weights_multi_kernel(const long long* __restrict__ table, int n) {
    int lo = 0, hi = n - 1;
    const long long b = blockIdx.x;
    while (lo < hi) {                   // last entry whose first block is <= b
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * 6 + 5] <= b) lo = mid; else hi = mid - 1;
    }
    const long long* e = table + lo * 6;
    wino6_weights_body(reinterpret_cast<const float*>(e[0]), reinterpret_cast<float*>(e[1]), (int)e[2], (int)e[3], (int)e[4],
                       (int)(b - e[5]) * 256 + threadIdx.x);
}

__device__ __forceinline__ float zero_of(float) { return 0.f; }
__device__ __forceinline__ F2 zero_of(F2) { return F2{0.f, 0.f}; }
__device__ __forceinline__ float act_of(float v, int act) { return act6(v, act); }
__device__ __forceinline__ F2 act_of(F2 v, int act) { return F2{act6(v.x, act), act6(v.y, act)}; }

// T = F2: one thread per (tile, 2 channels); T = float: per (tile, channel) - half the registers,
// twice the waves per SIMD (C2 is the channel count in units of T)
template <typename T>
__global__ void __launch_bounds__(256) wino6_input_kernel(const float* __restrict__ x, float* __restrict__ V, int B,
                                                          int H, int W, int C2, int reflect, int xcd_remap) {
    // reflect: 0 = zero padding 1, 1 = reflect padding 1 (tiles over the H x W outputs); 2 = zero
    // padding 2 with tiles over the (H+2) x (W+2) outputs of the FULL correlation: the dgrad of a
    // reflect-padded conv on its padded domain, folded back by wino6_output_kernel(fold)
    const int org = reflect == 2 ? 2 : 1;
    const int TH = (H + (reflect == 2 ? 7 : 5)) / 6, TW = (W + (reflect == 2 ? 7 : 5)) / 6;
    const long long tiles = (long long)B * TH * TW;
    // XCD-contiguous tile order: neighbouring 8x8 windows (2 shared rows / columns) meet in one L2
    unsigned blk = blockIdx.x;
    if (xcd_remap) blk = (blk & 7u) * (gridDim.x >> 3) + (blk >> 3);
    const long long i = (long long)blk * blockDim.x + threadIdx.x;
    if (i >= tiles * C2) return;
    const int c = (int)(i % C2);
    const long long tile = i / C2;
    const int tx = (int)(tile % TW), ty = (int)((tile / TW) % TH), b = (int)(tile / ((long long)TW * TH));
    const T* xin = reinterpret_cast<const T*>(x);
    T d[8][8], colv[8], o8[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        int hh = 6 * ty - org + r;
        bool okh;
        if (reflect == 1) {      // the padded image is rows -1 .. H; ragged tiles reach beyond it: zeros
            okh = hh <= H;
            hh = hh < 0 ? -hh : hh;
            hh = hh >= H ? 2 * (H - 1) - hh : hh;
            okh = okh && hh >= 0;
        } else {
            okh = hh >= 0 && hh < H;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            int ww = 6 * tx - org + q;
            bool ok;
            if (reflect == 1) {
                ok = okh && ww <= W;
                ww = ww < 0 ? -ww : ww;
                ww = ww >= W ? 2 * (W - 1) - ww : ww;
                ok = ok && ww >= 0;
            } else {
                ok = okh && ww >= 0 && ww < W;
            }
            d[r][q] = ok ? xin[(((long long)b * H + hh) * W + ww) * C2 + c] : zero_of(T{});
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {       // columns, in place
#pragma unroll
        for (int r = 0; r < 8; ++r) colv[r] = d[r][q];
        w6_bt(colv, o8);
#pragma unroll
        for (int r = 0; r < 8; ++r) d[r][q] = o8[r];
    }
    const long long plane = tiles * C2;
    T* out = reinterpret_cast<T*>(V) + tile * C2 + c;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        w6_bt(d[r], o8);
#pragma unroll
        for (int q = 0; q < 8; ++q) out[(long long)(r * 8 + q) * plane] = o8[q];
    }
}

// ---------------------------------------------------------------------------------------------
// Norm arithmetic inside the transforms.  The window is taken of relu(x*scale + shift) * keep / (1-p)
// instead of x - the InstanceNorm / BatchNorm apply, ReLU and dropout that stand between the conv that
// wrote x and this one (models/Generator.py:66-77) - so that activation is never written to HBM.
// scale / shift [groups][C] (groups = B or 1).  Dropout decisions come as ROW WORDS (mmh_dropout_bits_rows):
// drows[((b*H + h)*nW32 + j)*C + c] bit k = element (b, h, 32 j + k, c) is kept, so a thread (one channel,
// 8 columns of 8 rows) needs two words per row instead of a bit per element.
// Structure: every load of the window is issued before any arithmetic (two separate loop nests; DROP and
// relu are not branches) - with the arithmetic in the load loop the 64 loads serialise (2x the time).
struct NormPro {
    const float* scale;
    const float* shift;
    const uint32_t* drows;
    int nW32;           // ceil(W / 32)
    int per_image;      // groups == B
    float floor_;       // ReLU: 0, none: -inf
    float dsc;          // 1 / (1 - p)
};

// row / column geometry of one 8x8 window: source coordinates (reflect- or zero-padded), validity
struct WinGeom {
    int hh[8], ww[8];
    bool okh[8], okw[8];
};
__device__ __forceinline__ void win_geom(WinGeom& g, int ty, int tx, int org, int H, int W, int reflect) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        int hh = 6 * ty - org + r, ww = 6 * tx - org + r;
        bool okh, okw;
        if (reflect == 1) {
            okh = hh <= H; okw = ww <= W;
            hh = hh < 0 ? -hh : hh; ww = ww < 0 ? -ww : ww;
            hh = hh >= H ? 2 * (H - 1) - hh : hh; ww = ww >= W ? 2 * (W - 1) - ww : ww;
            okh = okh && hh >= 0; okw = okw && ww >= 0;
        } else {
            okh = hh >= 0 && hh < H; okw = ww >= 0 && ww < W;
        }
        g.hh[r] = okh ? hh : 0; g.ww[r] = okw ? ww : 0;
        g.okh[r] = okh; g.okw[r] = okw;
    }
}

template <bool DROP>
__global__ void __launch_bounds__(256) wino6_input_normact_kernel(const float* __restrict__ x, float* __restrict__ V,
                                                                  int B, int H, int W, int C, int reflect,
                                                                  int xcd_remap, const NormPro np) {
    const int TH = (H + 5) / 6, TW = (W + 5) / 6;
    const long long tiles = (long long)B * TH * TW;
    unsigned blk = blockIdx.x;
    if (xcd_remap) blk = (blk & 7u) * (gridDim.x >> 3) + (blk >> 3);
    const long long i = (long long)blk * blockDim.x + threadIdx.x;
    if (i >= tiles * C) return;
    const int c = (int)(i % C);
    const long long tile = i / C;
    const int tx = (int)(tile % TW), ty = (int)((tile / TW) % TH), b = (int)(tile / ((long long)TW * TH));
    WinGeom wg;
    win_geom(wg, ty, tx, 1, H, W, reflect);
    float d[8][8], colv[8], o8[8];
    uint32_t lo[DROP ? 8 : 1], hi[DROP ? 8 : 1];
    const int gi = (np.per_image ? b : 0) * C + c;
    const float sc = np.scale[gi], sf = np.shift[gi];
    int wq0 = 6 * tx - 1;
    wq0 = (wq0 < 0 ? 0 : wq0) >> 5;
    const int wq1 = wq0 + 1 < np.nW32 ? wq0 + 1 : wq0;
    // ---- loads
    // 32-bit element offsets (host: numel < 2^31): one address register per load instead of a pair
    unsigned co[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) co[q] = (unsigned)wg.ww[q] * (unsigned)C + (unsigned)c;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const unsigned row = (unsigned)(b * H + wg.hh[r]);
        const unsigned ro = row * (unsigned)W * (unsigned)C;
#pragma unroll
        for (int q = 0; q < 8; ++q) d[r][q] = x[ro + co[q]];     // clamped coordinates: always in range
        if (DROP) {
            lo[r] = np.drows[(row * (unsigned)np.nW32 + (unsigned)wq0) * (unsigned)C + (unsigned)c];
            hi[r] = np.drows[(row * (unsigned)np.nW32 + (unsigned)wq1) * (unsigned)C + (unsigned)c];
        }
    }
    // ---- arithmetic
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float t = __builtin_fmaf(d[r][q], sc, sf);
            t = fmaxf(t, np.floor_);
            bool keep = wg.okh[r] && wg.okw[q];         // padding and ragged-tile positions stay zero
            if (DROP) {
                const uint32_t wsel = (wg.ww[q] >> 5) == wq0 ? lo[r] : hi[r];
                keep = keep && ((wsel >> (wg.ww[q] & 31)) & 1u);
                t *= np.dsc;
            }
            d[r][q] = keep ? t : 0.f;
        }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int r = 0; r < 8; ++r) colv[r] = d[r][q];
        w6_bt(colv, o8);
#pragma unroll
        for (int r = 0; r < 8; ++r) d[r][q] = o8[r];
    }
    const long long plane = tiles * C;
    float* out = V + tile * C + c;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        w6_bt(d[r], o8);
#pragma unroll
        for (int q = 0; q < 8; ++q) out[(long long)(r * 8 + q) * plane] = o8[q];
    }
}

__device__ __forceinline__ void stat_add(float v, float& n, float& s) { n += 1.f; s += v; }
__device__ __forceinline__ void stat_add(F2, float&, float&) {}

// stats (T = float only): per (image, tile, channel) the count, mean and M2 of the tile's valid
// outputs, in the partial layout of norm_stats_partial (pointwise.hip) with chunk = tile index
// inside the image, so the InstanceNorm that follows the conv merges 121 partials per plane
// instead of re-reading the plane.
template <typename T>
__global__ void __launch_bounds__(256) wino6_output_kernel(const float* __restrict__ M, float* __restrict__ y,
                                                           const float* __restrict__ bias, int B, int H, int W,
                                                           int C2, int act, float* __restrict__ stats, int fold) {
    // fold: M holds the FULL correlation on the (H+2) x (W+2) padded domain (dgrad of a
    // ReflectionPad2d(1) conv); tile (ty, tx) covers padded rows 6ty .. 6ty+5.  The transpose of the
    // reflect padding adds ring row 0 onto padded row 2 and ring row H+1 onto padded row H-1 (columns
    // alike, corners by composition) - with this tile origin both partners always sit in the SAME
    // tile (host precondition (H+1) % 6 >= 2), so the fold happens in registers and y gets the
    // folded gradient on the real H x W domain: no border GEMMs, no second pass.
    const int TH = (H + (fold ? 7 : 5)) / 6, TW = (W + (fold ? 7 : 5)) / 6;
    const long long tiles = (long long)B * TH * TW;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tiles * C2) return;
    const int c = (int)(i % C2);
    const long long tile = i / C2;
    const int tx = (int)(tile % TW), ty = (int)((tile / TW) % TH), b = (int)(tile / ((long long)TW * TH));
    const long long plane = tiles * C2;
    const T* in = reinterpret_cast<const T*>(M) + tile * C2 + c;
    T s6[6][8], colv[8], o6[6];
#pragma unroll
    for (int q = 0; q < 8; ++q) {       // columns: 8 planes (r, q) -> 6 rows
#pragma unroll
        for (int r = 0; r < 8; ++r) colv[r] = in[(long long)(r * 8 + q) * plane];
        w6_at(colv, o6);
#pragma unroll
        for (int r = 0; r < 6; ++r) s6[r][q] = o6[r];
    }
    const T bv = bias ? reinterpret_cast<const T*>(bias)[c] : zero_of(T{});
    T* yo = reinterpret_cast<T*>(y);
    T vals[6][6];
    if (fold) {
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            w6_at(s6[r], o6);
#pragma unroll
            for (int q = 0; q < 6; ++q) vals[r][q] = o6[q];
        }
        const int rb = H + 1 - 6 * ty, cb = W + 1 - 6 * tx;     // local index of the bottom / right ring
#pragma unroll
        for (int q = 0; q < 6; ++q) {       // rows first (ring columns included), then columns: corners compose
            if (ty == 0) vals[2][q] = vals[2][q] + vals[0][q];
#pragma unroll
            for (int r = 2; r < 6; ++r)
                if (r == rb) vals[r - 2][q] = vals[r - 2][q] + vals[r][q];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            if (tx == 0) vals[r][2] = vals[r][2] + vals[r][0];
#pragma unroll
            for (int q = 2; q < 6; ++q)
                if (q == cb) vals[r][q - 2] = vals[r][q - 2] + vals[r][q];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int ph = 6 * ty + r;
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int pw = 6 * tx + q;
                if (ph >= 1 && ph <= H && pw >= 1 && pw <= W)
                    yo[(((long long)b * H + ph - 1) * W + pw - 1) * C2 + c] = vals[r][q];
            }
        }
        return;
    }
    float n = 0.f, sum = 0.f;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int hh = 6 * ty + r;
        w6_at(s6[r], o6);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int ww = 6 * tx + q;
            vals[r][q] = act_of(o6[q] + bv, act);
            if (hh < H && ww < W) {
                yo[(((long long)b * H + hh) * W + ww) * C2 + c] = vals[r][q];
                stat_add(vals[r][q], n, sum);
            }
        }
    }
    if (stats && sizeof(T) == 4) {
        const float mean = sum / n;         // n >= 1: a tile always holds at least one pixel of the image
        float m2 = 0.f;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int q = 0; q < 6; ++q)
                if (6 * ty + r < H && 6 * tx + q < W) {
                    const float d = *reinterpret_cast<const float*>(&vals[r][q]) - mean;
                    m2 += d * d;
                }
        float* o = stats + ((long long)(b * (TH * TW) + ty * TW + tx) * 3) * C2 + c;
        o[0] = n; o[C2] = mean; o[2 * C2] = m2;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) wino6_dy_kernel(const float* __restrict__ dy, float* __restrict__ Yh, int B,
                                                       int H, int W, int C2) {
    const int TH = (H + 5) / 6, TW = (W + 5) / 6;
    const long long tiles = (long long)B * TH * TW;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tiles * C2) return;
    const int c = (int)(i % C2);
    const long long tile = i / C2;
    const int tx = (int)(tile % TW), ty = (int)((tile / TW) % TH), b = (int)(tile / ((long long)TW * TH));
    const T* in = reinterpret_cast<const T*>(dy);
    T t[8][6], colv[6], o8[8];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int ww = 6 * tx + q;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int hh = 6 * ty + r;
            colv[r] = (hh < H && ww < W) ? in[(((long long)b * H + hh) * W + ww) * C2 + c] : zero_of(T{});
        }
        w6_a(colv, o8);
#pragma unroll
        for (int r = 0; r < 8; ++r) t[r][q] = o8[r];
    }
    const long long plane = tiles * C2;
    T* out = reinterpret_cast<T*>(Yh) + tile * C2 + c;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        w6_a(t[r], o8);
#pragma unroll
        for (int q = 0; q < 8; ++q) out[(long long)(r * 8 + q) * plane] = o8[q];
    }
}

// Both backward transforms of a loaded window d: Yh = A d' A^T of the inner 6x6 tile, V = B^T d B
template <typename T>
__device__ __forceinline__ void dy_transforms(T (&d)[8][8], float* __restrict__ V, float* __restrict__ Yh,
                                              long long plane, long long at, int fold) {
    T colv[8], o8[8];
    {   // Yh from the inner 6x6 tile (rows / columns org..org+5 of the window)
        T t[8][6];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
#pragma unroll
            for (int r = 0; r < 6; ++r) colv[r] = fold ? d[r + 2][q + 2] : d[r + 1][q + 1];
            w6_a(colv, o8);
#pragma unroll
            for (int r = 0; r < 8; ++r) t[r][q] = o8[r];
        }
        T* out = reinterpret_cast<T*>(Yh) + at;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            w6_a(t[r], o8);
#pragma unroll
            for (int q = 0; q < 8; ++q) out[(long long)(r * 8 + q) * plane] = o8[q];
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int r = 0; r < 8; ++r) colv[r] = d[r][q];
        w6_bt(colv, o8);
#pragma unroll
        for (int r = 0; r < 8; ++r) d[r][q] = o8[r];
    }
    T* out = reinterpret_cast<T*>(V) + at;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        w6_bt(d[r], o8);
#pragma unroll
        for (int q = 0; q < 8; ++q) out[(long long)(r * 8 + q) * plane] = o8[q];
    }
}

// Backward of one conv needs BOTH transforms of dy: V' = B^T d B of the zero-padded 8x8 window (the
// dgrad GEMM operand) and Yh = A dY A^T of the 6x6 tile inside it (the wgrad GEMM operand).  One
// kernel loads the window once (one read of dy instead of two).
template <typename T>
__global__ void __launch_bounds__(256) wino6_input_dy_kernel(const float* __restrict__ dy, float* __restrict__ V,
                                                             float* __restrict__ Yh, int B, int H, int W, int C2,
                                                             int xcd_remap, int fold) {
    // fold: the dgrad operand is taken on the padded domain (window origin -2, see
    // wino6_output_kernel); the wgrad tile dy[6ty .. 6ty+5] then sits at rows / columns 2..7 of the
    // window.  Host precondition: ceil((H+2)/6) == ceil(H/6), so both operands share one tile grid.
    const int org = fold ? 2 : 1;
    const int TH = (H + 5) / 6, TW = (W + 5) / 6;
    const long long tiles = (long long)B * TH * TW;
    unsigned blk = blockIdx.x;
    if (xcd_remap) blk = (blk & 7u) * (gridDim.x >> 3) + (blk >> 3);
    const long long i = (long long)blk * blockDim.x + threadIdx.x;
    if (i >= tiles * C2) return;
    const int c = (int)(i % C2);
    const long long tile = i / C2;
    const int tx = (int)(tile % TW), ty = (int)((tile / TW) % TH), b = (int)(tile / ((long long)TW * TH));
    const T* in = reinterpret_cast<const T*>(dy);
    T d[8][8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int hh = 6 * ty - org + r;
        const bool okh = hh >= 0 && hh < H;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int ww = 6 * tx - org + q;
            d[r][q] = (okh && ww >= 0 && ww < W) ? in[(((long long)b * H + hh) * W + ww) * C2 + c] : zero_of(T{});
        }
    }
    dy_transforms(d, V, Yh, tiles * C2, tile * C2 + c, fold);
}

// Norm backward inside the transform: dy is not read but computed per element from the gradient g of the
// norm's OUTPUT, the norm's input x and the per-(group, channel) sums of mmh_norm_bwd_reduce_rc - the
// arithmetic of norm_bwd_apply_v2<.., RC> (pointwise.hip; mmh::norm_bwd_elem) - so the gradient of the conv
// output between that conv and its norm is never written to HBM.  Loads first, arithmetic after, as in
// wino6_input_normact_kernel; x is staged four rows at a time (96 instead of 128 live window registers).
struct NormBwdPro {
    const float* x;         // the norm's input = this conv's forward output
    const float* mean;
    const float* invstd;
    const float* gamma;     // NULL: 1
    const float* s1;
    const float* s2;
    float inv_count;
    int relu;
    NormPro np;             // scale / shift / dropout row words of the forward apply (the keep decision)
};

template <bool DROP>
__global__ void __launch_bounds__(256, 4) wino6_input_dy_normbwd_kernel(const float* __restrict__ g, float* __restrict__ V,
                                                                     float* __restrict__ Yh, int B, int H, int W,
                                                                     int C, int xcd_remap, int fold,
                                                                     const NormBwdPro nb) {
    const int org = fold ? 2 : 1;
    const int TH = (H + 5) / 6, TW = (W + 5) / 6;
    const long long tiles = (long long)B * TH * TW;
    unsigned blk = blockIdx.x;
    if (xcd_remap) blk = (blk & 7u) * (gridDim.x >> 3) + (blk >> 3);
    const long long i = (long long)blk * blockDim.x + threadIdx.x;
    if (i >= tiles * C) return;
    const int c = (int)(i % C);
    const long long tile = i / C;
    const int tx = (int)(tile % TW), ty = (int)((tile / TW) % TH), b = (int)(tile / ((long long)TW * TH));
    WinGeom wg;
    win_geom(wg, ty, tx, org, H, W, 0);
    const int gi = (nb.np.per_image ? b : 0) * C + c;
    const float sc = nb.np.scale[gi], sf = nb.np.shift[gi], mu = nb.mean[gi], is = nb.invstd[gi];
    const float k0 = (nb.gamma ? nb.gamma[c] : 1.f) * is;
    const float k1 = nb.s1[gi] * nb.inv_count;
    const float k2 = k0 * is * (nb.s2[gi] * nb.inv_count);
    int wq0 = 6 * tx - org;
    wq0 = (wq0 < 0 ? 0 : wq0) >> 5;
    const int wq1 = wq0 + 1 < nb.np.nW32 ? wq0 + 1 : wq0;
    float d[8][8];
    uint32_t lo[2], hi[2];
    unsigned co[8], ro[8];     // 32-bit element offsets (host: numel < 2^31)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        co[q] = (unsigned)wg.ww[q] * (unsigned)C + (unsigned)c;
        ro[q] = (unsigned)(b * H + wg.hh[q]) * (unsigned)W * (unsigned)C;
    }
    // two window rows per stage: their g, x and dropout words are loaded together, then turned into dy; the
    // scheduling barrier keeps the next stage's loads behind this stage's arithmetic (all 128 window loads
    // in flight at once need > 168 registers and spill)
#pragma unroll
    for (int stage = 0; stage < 4; ++stage) {
        float xw[2][8];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int rr = 2 * stage + r;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                d[rr][q] = g[ro[rr] + co[q]];
                xw[r][q] = nb.x[ro[rr] + co[q]];
            }
            if (DROP) {
                const unsigned row = (unsigned)(b * H + wg.hh[rr]);
                lo[r] = nb.np.drows[(row * (unsigned)nb.np.nW32 + (unsigned)wq0) * (unsigned)C + (unsigned)c];
                hi[r] = nb.np.drows[(row * (unsigned)nb.np.nW32 + (unsigned)wq1) * (unsigned)C + (unsigned)c];
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int rr = 2 * stage + r;
                bool keep = !nb.relu || __builtin_fmaf(xw[r][q], sc, sf) > 0.f;
                if (DROP) {
                    const uint32_t wsel = (wg.ww[q] >> 5) == wq0 ? lo[r] : hi[r];
                    keep = keep && ((wsel >> (wg.ww[q] & 31)) & 1u);
                }
                const float o = mmh::norm_bwd_elem(d[rr][q], keep, nb.np.dsc, xw[r][q], mu, k0, k1, k2);
                d[rr][q] = (wg.okh[rr] && wg.okw[q]) ? o : 0.f;
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    dy_transforms(d, V, Yh, tiles * C, tile * C + c, fold);
}

// dw[3][3][Cin][Cout] (+)= G^T dU G, dU: [64][Cin][Cout]
__global__ void __launch_bounds__(256) wino6_dw_kernel(const float* __restrict__ dU, float* __restrict__ dw, int64_t plane2, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= plane2) return;
    const F2* in = reinterpret_cast<const F2*>(dU) + i;
    F2 t[3][8], colv[8], o3[3];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int r = 0; r < 8; ++r) colv[r] = in[(int64_t)(r * 8 + q) * plane2];
        w6_gt(colv, o3);
#pragma unroll
        for (int r = 0; r < 3; ++r) t[r][q] = o3[r];
    }
    F2* out = reinterpret_cast<F2*>(dw) + i;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        w6_gt(t[a], o3);
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            F2 v = o3[b];
            if (accumulate) v = v + out[(int64_t)(a * 3 + b) * plane2];
            out[(int64_t)(a * 3 + b) * plane2] = v;
        }
    }
}

}  // namespace

namespace mmh {

int wino6_weights(const float* w, float* U, int Cin, int Cout, int flip_transpose, hipStream_t st) {
    hipLaunchKernelGGL(wino6_weights_kernel, dim3((Cin * Cout + 255) / 256), dim3(256), 0, st, w, U, Cin, Cout,
                       flip_transpose);
    return check_launch("wino6_weights_kernel");
}

int wino6_weights_multi(const long long* table, int n, long long total_blocks, hipStream_t st) {
    hipLaunchKernelGGL(wino6_weights_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, st, table, n);
    return check_launch("wino6_weights_multi_kernel");
}

int g_wino6_vec = 0x0;   // bit 0 / 1 / 2: input / output / dy transform works on 2 channels per thread

int wino6_input(const float* x, float* V, int B, int H, int W, int C, int reflect, int xcd, hipStream_t st) {
    const int ext = reflect == 2 ? 7 : 5;
    const long long tiles = (long long)B * ((H + ext) / 6) * ((W + ext) / 6);
    const int cpt = (g_wino6_vec & 1) ? 2 : 1;
    const unsigned nblk = (unsigned)((tiles * (C / cpt) + 255) / 256);
    const int remap = (xcd && nblk % 8 == 0 && nblk >= 64) ? 1 : 0;
    if (cpt == 2)
        hipLaunchKernelGGL(wino6_input_kernel<F2>, dim3(nblk), dim3(256), 0, st, x, V, B, H, W, C / 2, reflect, remap);
    else
        hipLaunchKernelGGL(wino6_input_kernel<float>, dim3(nblk), dim3(256), 0, st, x, V, B, H, W, C, reflect, remap);
    return check_launch("wino6_input_kernel");
}

int wino6_output(const float* M, float* y, const float* bias, int B, int H, int W, int C, int act, float* stats,
                 int fold, hipStream_t st) {
    const int ext = fold ? 7 : 5;
    const long long tiles = (long long)B * ((H + ext) / 6) * ((W + ext) / 6);
    if ((g_wino6_vec & 2) && !stats)
        hipLaunchKernelGGL(wino6_output_kernel<F2>, dim3((unsigned)((tiles * (C / 2) + 255) / 256)), dim3(256), 0, st, M,
                           y, bias, B, H, W, C / 2, act, nullptr, fold);
    else
        hipLaunchKernelGGL(wino6_output_kernel<float>, dim3((unsigned)((tiles * C + 255) / 256)), dim3(256), 0, st, M, y,
                           bias, B, H, W, C, act, stats, fold);
    return check_launch("wino6_output_kernel");
}

int wino6_dy(const float* dy, float* Yh, int B, int H, int W, int C, hipStream_t st) {
    const long long tiles = (long long)B * ((H + 5) / 6) * ((W + 5) / 6);
    if (g_wino6_vec & 4)
        hipLaunchKernelGGL(wino6_dy_kernel<F2>, dim3((unsigned)((tiles * (C / 2) + 255) / 256)), dim3(256), 0, st, dy, Yh,
                           B, H, W, C / 2);
    else
        hipLaunchKernelGGL(wino6_dy_kernel<float>, dim3((unsigned)((tiles * C + 255) / 256)), dim3(256), 0, st, dy, Yh, B,
                           H, W, C);
    return check_launch("wino6_dy_kernel");
}

int wino6_input_normact(const float* x, float* V, int B, int H, int W, int C, int reflect, int xcd,
                        const float* scale, const float* shift, int groups, int relu, float drop_p,
                        const uint32_t* drows, hipStream_t st) {
    const long long tiles = (long long)B * ((H + 5) / 6) * ((W + 5) / 6);
    const unsigned nblk = (unsigned)((tiles * C + 255) / 256);
    const int remap = (xcd && nblk % 8 == 0 && nblk >= 64) ? 1 : 0;
    NormPro np{scale, shift, drows, (W + 31) / 32, groups == 1 ? 0 : 1, relu ? 0.f : -INFINITY, 1.f / (1.f - drop_p)};
    if (drows)
        hipLaunchKernelGGL(wino6_input_normact_kernel<true>, dim3(nblk), dim3(256), 0, st, x, V, B, H, W, C, reflect,
                           remap, np);
    else
        hipLaunchKernelGGL(wino6_input_normact_kernel<false>, dim3(nblk), dim3(256), 0, st, x, V, B, H, W, C, reflect,
                           remap, np);
    return check_launch("wino6_input_normact_kernel");
}

int wino6_input_dy_normbwd(const float* g, const float* x, float* V, float* Yh, int B, int H, int W, int C, int xcd,
                           int fold, const float* mean, const float* invstd, const float* gamma, const float* s1,
                           const float* s2, double count, const float* scale, const float* shift,
                           const uint32_t* drows, int groups, int relu, float drop_p, hipStream_t st) {
    const long long tiles = (long long)B * ((H + 5) / 6) * ((W + 5) / 6);
    const unsigned nblk = (unsigned)((tiles * C + 255) / 256);
    const int remap = (xcd && nblk % 8 == 0 && nblk >= 64) ? 1 : 0;
    NormBwdPro nb{x, mean, invstd, gamma, s1, s2, (float)(1.0 / count), relu,
                  NormPro{scale, shift, drows, (W + 31) / 32, groups == 1 ? 0 : 1, 0.f, 1.f / (1.f - drop_p)}};
    if (drows)
        hipLaunchKernelGGL(wino6_input_dy_normbwd_kernel<true>, dim3(nblk), dim3(256), 0, st, g, V, Yh, B, H, W, C, remap,
                           fold, nb);
    else
        hipLaunchKernelGGL(wino6_input_dy_normbwd_kernel<false>, dim3(nblk), dim3(256), 0, st, g, V, Yh, B, H, W, C, remap,
                           fold, nb);
    return check_launch("wino6_input_dy_normbwd_kernel");
}

int wino6_input_dy(const float* dy, float* V, float* Yh, int B, int H, int W, int C, int xcd, int fold,
                   hipStream_t st) {
    const long long tiles = (long long)B * ((H + 5) / 6) * ((W + 5) / 6);
    const unsigned nblk = (unsigned)((tiles * C + 255) / 256);
    hipLaunchKernelGGL(wino6_input_dy_kernel<float>, dim3(nblk), dim3(256), 0, st, dy, V, Yh, B, H, W, C,
                       (xcd && nblk % 8 == 0 && nblk >= 64) ? 1 : 0, fold);
    return check_launch("wino6_input_dy_kernel");
}

int wino6_dw(const float* dU, float* dw, int Cin, int Cout, int accumulate, hipStream_t st) {
    const int64_t n2 = (int64_t)Cin * Cout / 2;
    hipLaunchKernelGGL(wino6_dw_kernel, dim3((unsigned)cdiv(n2, 256)), dim3(256), 0, st, dU, dw, n2, accumulate);
    return check_launch("wino6_dw_kernel");
}

}  // namespace mmh
