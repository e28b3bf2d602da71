// fp32 wgrad of the 7x7 / stride 1 / reflect-pad 3 stems (Cin = 4 .. 44 -> 64; models/Generator.py:158-168,
// models/Discriminator.py:60-64) from an LDS-resident input band.
//
// The generic conv_wgrad_kernel gathers its (tap, channel) x pixel operand from global memory per tap: every
// input element travels L2 -> L1 -> LDS 49 times (10 GB per launch at 24 channels) and the kernel sits at
// 105-110 TF with 20 % of its time in that load path (tools/ablate_narrow.py).  Here
//   * the input is reflect-padded ONCE into [B][H+6][W+6][Cin] (a copy of 1.05x its size), so that the 7 taps
//     of one filter row at a pixel are 7*Cin CONTIGUOUS floats: row (kw, ci) of the GEMM is element
//     c*Cin + (kw*Cin + ci) of the padded image row;
//   * a workgroup owns one filter row kh and walks tiles of 2 x 64 pixels: the tile's two input rows
//     (70 padded columns) and its 128 x 64 block of dy are staged in LDS once and both MFMA operands are
//     plain ds_read_b32 of 32 consecutive floats (A: rows of the band at a pixel offset; B: dy columns);
//   * output rows 7*Cin padded to RB*32, 64 columns: RB*2 MFMA 32x32 tiles shared by 4 waves (column half x
//     interleaved row blocks), contraction over the pixels, split-K over tile ranges into fp32 slabs summed
//     in a fixed order (deterministic).
// Each input element is read 7 times (once per filter row), not 49.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));     // (HIP's float4 class kept the prefetch registers in scratch)
constexpr int TR = 2;           // image rows per tile
constexpr int TC = 64;          // image columns per tile
constexpr int XC = TC + 6;      // padded columns per staged row

__global__ void reflect_pad3_kernel(const float4* __restrict__ x, float4* __restrict__ xp, int B, int H, int W, int C4,
                                    int xcs4) {
    const int Hp = H + 6, Wp = W + 6;
    const long long n = (long long)B * Hp * Wp * C4;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C4);
    long long t = i / C4;
    const int wp = (int)(t % Wp);
    t /= Wp;
    const int hp = (int)(t % Hp), b = (int)(t / Hp);
    int h = hp - 3, w = wp - 3;
    h = h < 0 ? -h : h; w = w < 0 ? -w : w;
    h = h >= H ? 2 * (H - 1) - h : h; w = w >= W ? 2 * (W - 1) - w : w;
    xp[i] = x[(((long long)b * H + h) * W + w) * xcs4 + c];
}

struct StemWgKP {
    const float* xp;        // [B][H+6][W+6][Cin]
    const float* dy;        // [B][H][W][64]
    float* slab;            // [nsplit][49 * Cin][64]
    int B, H, W, Cin;
    int run;                // 7 * Cin: rows per filter row
    int khg;                // filter rows per workgroup: the dy tile staged once serves khg * run rows
    int tiles;              // B * (H / TR) * (W / TC)
    int tiles_per_split;
};

constexpr int RBMAX = 14;   // row blocks of 32 per workgroup (khg * run <= 448)

// NTW: MFMA tiles per wave (row blocks rb0, rb0 + 2, ...: all computed, rows past the last one are never stored);
// NXMAX: float4 loads of the input band per thread and tile; MULTI: several filter rows per workgroup (a row
// block may straddle two of them: per-lane offsets instead of immediates)
template <int NTW, int NXMAX, bool MULTI>
__global__ void __launch_bounds__(256) stem_wgrad_kernel(const StemWgKP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int xrow = XC * p.Cin;                        // floats per staged input row
    const int kh0 = blockIdx.y * p.khg;
    const int nkh = min(p.khg, 7 - kh0);                // filter rows of this workgroup
    const int rows = nkh * p.run;                       // valid GEMM rows: flat (kh, kw, ci) from kh0 * run
    const int brows = TR + nkh - 1;                     // input rows of the band
    float* const xs = smem;                             // [brows][xrow] + 32 floats of slack
    float* const dys = smem + (((TR + p.khg - 1) * xrow + 32 + 3) & ~3);    // [TR*TC][64]  (rows of a block past
                                                                            // the run read on into dys: finite values)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int cb = wave & 1, rb0 = wave >> 1;           // this wave: column half cb, row blocks rb0, rb0+2, ...
    const int split = blockIdx.x;
    const int t0 = split * p.tiles_per_split, t1 = min(p.tiles, t0 + p.tiles_per_split);
    const int Hp = p.H + 6, Wp = p.W + 6;
    const int tw = p.W / TC, th = p.H / TR;
    const int xrow4 = xrow / 4;
    const int nx4 = brows * xrow4;                      // float4 elements of the band

    // this lane's row of each of its row blocks: (kh, j) -> offset inside the band; rows past the last one read
    // the slack (finite garbage into accumulators that are never stored)
    int aoff[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        int m = (rb0 + 2 * i) * 32 + l31;
        m = m < rows ? m : rows - 1;
        const int khl = m / p.run;
        aoff[i] = khl * xrow + (m - khl * p.run);
    }

    f4 rx[NXMAX], rd[8];
    const long long rowstep4 = (long long)Wp * p.Cin / 4, drow4 = (long long)p.W * 16;
    int xoff[NXMAX];
    bool xok[NXMAX];
#pragma unroll
    for (int i = 0; i < NXMAX; ++i) {
        const int e = tid + 256 * i;
        xok[i] = e < nx4;
        const int r = xok[i] ? e / xrow4 : 0, o = xok[i] ? e - r * xrow4 : 0;
        xoff[i] = (int)(r * rowstep4 + o);
    }
    int doff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = tid + 256 * i;                    // [TR][TC*16] float4
        const int r = e / (TC * 16), o = e - r * (TC * 16);
        doff[i] = (int)(r * drow4 + o);
    }
#define STEM_PREFETCH(T_)                                                                                        \
    {                                                                                                            \
        const int t_ = (T_);                                                                                     \
        const int wx = t_ % tw, hy = (t_ / tw) % th, b = t_ / (tw * th);                                         \
        const f4* xsrc = reinterpret_cast<const f4*>(                                                            \
            p.xp + (((long long)b * Hp + hy * TR + kh0) * Wp + wx * TC) * p.Cin);                                \
        const f4* dsrc = reinterpret_cast<const f4*>(p.dy + (((long long)b * p.H + hy * TR) * p.W + wx * TC) * 64); \
        _Pragma("unroll") for (int i = 0; i < NXMAX; ++i) rx[i] = xsrc[xoff[i]];                                 \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) rd[i] = dsrc[doff[i]];                                     \
    }
#define STEM_STAGE()                                                                                             \
    {                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < NXMAX; ++i)                                                        \
            if (xok[i]) reinterpret_cast<f4*>(xs)[tid + 256 * i] = rx[i];                                        \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) reinterpret_cast<f4*>(dys)[tid + 256 * i] = rd[i];         \
    }

    f32x16 acc[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    if (t0 < t1) {
        STEM_PREFETCH(t0)
        STEM_STAGE()
        __syncthreads();
        for (int t = t0; t < t1; ++t) {
            const bool more = t + 1 < t1;
            if (more) STEM_PREFETCH(t + 1)
#pragma unroll 2
            for (int kk = 0; kk < TR * TC / 2; ++kk) {
                const int pa = 2 * kk + h;              // this lane half's pixel
                const int r = pa >> 6, c = pa & 63;
                const float* xa = xs + r * xrow + c * p.Cin + (MULTI ? 0 : rb0 * 32 + l31);
                const float bv = dys[pa * 64 + cb * 32 + l31];
#pragma unroll
                for (int i = 0; i < NTW; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(MULTI ? xa[aoff[i]] : xa[64 * i], bv, acc[i], 0, 0, 0);
            }
            __syncthreads();
            if (more) STEM_STAGE()
            __syncthreads();
        }
    }
#undef STEM_PREFETCH
#undef STEM_STAGE
    // C/D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); GEMM row m of this workgroup is
    // row kh0 * run + m of dw [49 * Cin][64]
    float* out = p.slab + ((long long)split * 7 * p.run + (long long)kh0 * p.run) * 64 + cb * 32 + l31;
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (rb0 + 2 * i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < rows) out[(long long)m * 64] = acc[i][r];
        }
}

// filter rows per workgroup: as many as fit RBMAX row blocks (44 ch: 1, 24 ch: 2, 8 ch and fewer: 7)
int stem_khg(int Cin) {
    int k = (RBMAX * 32) / (7 * Cin);
    return k < 1 ? 1 : (k > 7 ? 7 : k);
}

int stem_splits(int tiles, int groups) {
    int s = 512 / groups;       // two workgroups per CU
    if (s < 1) s = 1;
    return s < tiles ? s : tiles;
}

bool stem_ok(const mmh_conv_desc* d) {
    return d->kh == 7 && d->kw == 7 && d->stride == 1 && d->pad == 3 && d->pad_mode == MMH_PAD_REFLECT &&
           d->dtype == MMH_F32 && d->Cout == 64 && d->y_cs == 64 &&
           (d->Cin == 8 || d->Cin == 44) &&     // where it beats the generic kernel (B=32 @256x256: 44 ch 118 vs 110 TF,
                                                // 8 ch 89 vs 81; 24 ch: 168 rows per filter row pad badly to 32-row
                                                // MFMA blocks - 89-99 vs 107; 4 ch: 73 vs 73)
           d->x_cs >= d->Cin && d->x_cs % 4 == 0 && d->H % TR == 0 && d->W % TC == 0 && d->H >= 4 && d->W >= 4 &&
           d->Ho == d->H && d->Wo == d->W;
}

}  // namespace

extern "C" {

int mmh_conv7_stem_wgrad_supported(const mmh_conv_desc* d) { return d && stem_ok(d) ? 1 : 0; }

size_t mmh_conv7_stem_wgrad_ws_bytes(const mmh_conv_desc* d) {
    if (!d || !stem_ok(d)) return 0;
    const size_t xp = (size_t)d->B * (d->H + 6) * (d->W + 6) * d->Cin;
    const int tiles = d->B * (d->H / TR) * (d->W / TC);
    const int groups = (7 + stem_khg(d->Cin) - 1) / stem_khg(d->Cin);
    return (xp + (size_t)stem_splits(tiles, groups) * 49 * d->Cin * 64) * sizeof(float) + 256;
}

int mmh_conv7_stem_wgrad(const mmh_conv_desc* d, const void* x, const void* dy, void* dw, void* ws, size_t ws_bytes,
                         int accumulate, mmh_stream_t s) {
    MMH_REQUIRE(d && stem_ok(d), "mmh_conv7_stem_wgrad: needs a 7x7 / stride 1 / reflect pad 3 fp32 conv, Cin 8 | 44, "
                                 "Cout == 64 dense, H %% 2 == 0, W %% 64 == 0");
    MMH_REQUIRE(x && dy && dw && ws && ws_bytes >= mmh_conv7_stem_wgrad_ws_bytes(d), "mmh_conv7_stem_wgrad: bad buffers");
    hipStream_t st = mmh::as_stream(s);
    const int Cin = d->Cin;
    const long long np4 = (long long)d->B * (d->H + 6) * (d->W + 6) * (Cin / 4);
    float* xp = static_cast<float*>(ws);
    hipLaunchKernelGGL(reflect_pad3_kernel, dim3((unsigned)mmh::cdiv(np4, 256)), dim3(256), 0, st,
                       static_cast<const float4*>(x), reinterpret_cast<float4*>(xp), d->B, d->H, d->W, Cin / 4,
                       d->x_cs / 4);
    if (int rc = mmh::check_launch("reflect_pad3_kernel")) return rc;
    StemWgKP p{};
    p.xp = xp;
    p.dy = static_cast<const float*>(dy);
    p.slab = xp + (((size_t)np4 * 4 + 63) & ~(size_t)63);
    p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = Cin;
    p.run = 7 * Cin;
    p.khg = stem_khg(Cin);
    p.tiles = d->B * (d->H / TR) * (d->W / TC);
    const int groups = (7 + p.khg - 1) / p.khg;
    const int splits = stem_splits(p.tiles, groups);
    p.tiles_per_split = (p.tiles + splits - 1) / splits;
    const size_t lds = ((size_t)(((TR + p.khg - 1) * XC * Cin + 32 + 3) & ~3) + (size_t)TR * TC * 64) * sizeof(float);
#define MMH_STEM(NT, NX, MU)                                                                                    \
    {                                                                                                            \
        static int ready = -1;                                                                                   \
        if (ready != 0) {                                                                                        \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(stem_wgrad_kernel<NT, NX, MU>),     \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);           \
            ready = e == hipSuccess ? 0 : mmh::fail("hipFuncSetAttribute: %s", hipGetErrorString(e));            \
        }                                                                                                        \
        if (ready != 0) return ready;                                                                            \
        hipLaunchKernelGGL((stem_wgrad_kernel<NT, NX, MU>), dim3(splits, groups), dim3(256), lds, st, p);        \
    }
    if (Cin == 44) MMH_STEM(5, 7, false)            // 308 rows: 10 blocks, one filter row per workgroup
    else MMH_STEM(7, 5, true)                       // Cin 8: all 7 filter rows, 392 rows, 13 blocks
#undef MMH_STEM
    if (int rc = mmh::check_launch("stem_wgrad_kernel")) return rc;
    const int n4 = 49 * Cin * 64 / 4;
    return mmh::launch_slab_reduce(p.slab, static_cast<float*>(dw), n4, splits, accumulate, n4, st);
}

}  // extern "C"
