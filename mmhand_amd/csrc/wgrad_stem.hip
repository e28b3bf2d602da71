// 16-bit wgrad of the 7x7 / stride-1 / pad-3 stems (Cin = 3 .. 48 input channels padded to C8, 64 output channels):
//
//   dw[kh][kw][c][n] = sum over output pixels p of x[p + (kh - 3, kw - 3)][c] * dy[p][n]
//   (models/Generator.py:158-164, models/Discriminator.py:60-64: the weight gradient of every stem, seven per step)
//
// The first-generation implicit-GEMM wgrad ran these at 360 TFLOP/s (0.4 / 0.9 / 1.5 ms at 8 / 24 / 44 channels): with
// so few channels per tap its 64-deep k-steps are mostly address arithmetic.  Here the contraction runs over PIXELS of
// an image row and the filter COLUMN taps are flattened into the operand: in NHWC a row of the (padded) input is one
// contiguous array, and the window of output pixel ow under filter row kh - the 7 x C8 values x[oh + kh][ow .. ow + 6][*]
// - starts at element ow * C8 of that array.  So for a fixed filter row
//
//   dw[kh][j = kw * C8 + c][n] = sum_ow  R_kh[ow * C8 + j] * dy[ow][n],       R_kh = input row oh + kh - 3, flat
//
// is a GEMM whose A operand is the row array read at pixel pitch C8 (the MFMA's k index is the pixel: both operands are
// k-major in memory, so both are read TRANSPOSED from LDS with ds_read_b64_tr_b16, as in wgrad_lp16t.hip) - no im2col,
// every input byte staged once per pixel block and used by all 49 taps.
//
// Work-group = 512 threads; it walks blocks of 4 x 16 output pixels (ring of four LDS stages by LDS-DMA: the block's
// 10 x 22-pixel input halo as ten flat rows, reflect / zero padding folded into the DMA's source addresses, and its
// 64 x 64 dy tile), split-K over block ranges.  MFMA 32x32x16: a tile is 32 flattened columns j x 32 output channels of
// one filter row; the (filter row, j tile) pairs are dealt round-robin to the eight waves, each wave holding both
// channel tiles of its pairs (an A fragment serves two MFMAs).  Accumulators: up to 6 pairs x 2 x 16 VGPRs per wave;
// C8 = 48 (7 x 11 pairs) takes two work-group kinds (filter rows 0-3 / 4-6).  fp32 slabs [split][7][JP][64] are summed
// in a fixed order into dw by stem_slab_reduce_kernel (deterministic).
#include <algorithm>
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int BR = 4, BC = 16;              // pixel block: 4 rows x 16 columns
constexpr int HR = BR + 6, HC = BC + 6;     // halo 10 x 22 pixels
constexpr int XROUNDS = 3;                  // DMA instructions per wave and stage for x (3 x 512 x 16 B >= 10 x 22 x 96 B)
constexpr int DSTAGE = BR * BC * 128;       // dy tile [64 pixels][64 channels]: 8 KiB
constexpr int RING = 4;
constexpr int PW = 6;                       // (filter row, j tile) pairs per wave at most

struct StemWgKP {
    const char* x;          // 16-bit [B][H][W][C8]
    const char* dy;         // 16-bit [B][H][W][dy_cs], channels 0..63
    const char* zeros;
    float* slab;            // [S][7][JP][64]
    int B, H, W, C8, dy_cs, reflect;
    int TR, TC, nblk, bps, S, KG;
    int Jt, JP;             // j tiles of 32, JP = 32 * Jt
    int rp;                 // LDS bytes per halo row = 22 * C8 * 2
    int dyH, dyW, dyo;      // dy's own extent; dyo > 0: output pixel (oh, ow) reads dy at reflect(oh - dyo, ow - dyo) (head wgrad)
    int xstage, tstage;     // x bytes per stage (rounded up to the DMA rounds' footprint), stage bytes
};

template <bool H16>
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    if (H16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0,
                                                      0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// 8 consecutive k rows of one column per lane: two transposed reads 4 rows apart (rows `rowbytes` apart)
__device__ __forceinline__ bf16x8 tr_frag(unsigned addr, unsigned rowbytes) {
    typedef s16x4 __attribute__((address_space(3))) * lds_p;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(size_t)(addr));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(size_t)(addr + 4 * rowbytes));
    struct { s16x4 a, b; } both = {lo, hi};
    return __builtin_bit_cast(bf16x8, both);
}

template <bool H16>
__global__ void __launch_bounds__(512, 2) wgrad_stem_kernel(const StemWgKP p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int per_xcd = (p.S * p.KG + 7) / 8;
    const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (item >= p.S * p.KG) return;
    const int kg = item % p.KG, split = item / p.KG;
    const int kh_lo = kg == 0 ? 0 : 4, nkh = p.KG == 1 ? 7 : (kg == 0 ? 4 : 3);
    const int npairs = nkh * p.Jt;
    const int blk0 = split * p.bps, blk1 = min(p.nblk, blk0 + p.bps);
    const int nsteps = blk1 - blk0;

    // ---- DMA roles.  x: unit u = round * 512 + tid of the stage's flat halo image (10 rows x upr units of 16 bytes):
    // row u / upr, pixel (u % upr) / c8u, chunk (u % upr) % c8u.  dy: unit tid: pixel tid / 8, chunk tid % 8.
    const int c8u = p.C8 / 8, upr = HC * c8u, units = HR * upr;
    int x_hy[XROUNDS], x_hx[XROUNDS];
    unsigned x_ck[XROUNDS];
#pragma unroll
    for (int r = 0; r < XROUNDS; ++r) {
        const int u = r * 512 + tid;
        const int row = u / upr, ur = u - row * upr;
        x_hy[r] = u < units ? row : -1000;
        x_hx[r] = ur / c8u;
        x_ck[r] = (unsigned)(ur - x_hx[r] * c8u) * 16u;
    }
    const int d_py = tid >> 7, d_px = (tid >> 3) & 15;
    const unsigned d_ck = (unsigned)(tid & 7) * 16u;
    int nb = blk0;
    int nb_img = nb / (p.TR * p.TC);
    int nb_tr = (nb - nb_img * p.TR * p.TC) / p.TC;
    int nb_tc = nb - (nb_img * p.TR + nb_tr) * p.TC;
    int slot_next = 0;
    const unsigned lds0 = mmh::lds_addr_of(smem);
    const unsigned wdst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);
    auto issue = [&]() {
        const unsigned sbase = (unsigned)slot_next * (unsigned)p.tstage;
        slot_next = (slot_next + 1) & (RING - 1);
        const bool live = nb < blk1;
        const int r0 = nb_tr * BR, c0 = nb_tc * BC;
#pragma unroll
        for (int r = 0; r < XROUNDS; ++r) {
            int ih = r0 + x_hy[r] - 3, iw = c0 + x_hx[r] - 3;
            if (p.reflect) {
                ih = ih < 0 ? -ih : ih;
                iw = iw < 0 ? -iw : iw;
                ih = ih >= p.H ? 2 * (p.H - 1) - ih : ih;
                iw = iw >= p.W ? 2 * (p.W - 1) - iw : iw;
            }
            const bool ok = live && x_hy[r] >= 0 && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            const char* g = ok ? p.x + (size_t)((nb_img * p.H + ih) * p.W + iw) * (size_t)(p.C8 * 2) + x_ck[r]
                               : p.zeros + (lane & 7) * 16;
            mmh::lds_dma16(g, wdst + sbase + (unsigned)r * 8192u);
        }
        {
            const int oh = r0 + d_py, ow = c0 + d_px;
            const bool ok = live && oh < p.H && ow < p.W;
            int sh = oh - p.dyo, sw = ow - p.dyo;
            if (p.dyo) {
                sh = sh < 0 ? -sh : sh;
                sw = sw < 0 ? -sw : sw;
                sh = sh >= p.dyH ? 2 * (p.dyH - 1) - sh : sh;
                sw = sw >= p.dyW ? 2 * (p.dyW - 1) - sw : sw;
            }
            const char* g = ok ? p.dy + (size_t)((nb_img * p.dyH + sh) * p.dyW + sw) * (size_t)(p.dy_cs * 2) + d_ck
                               : p.zeros + (lane & 7) * 16;
            mmh::lds_dma16(g, wdst + sbase + (unsigned)p.xstage);
        }
        ++nb;
        if (++nb_tc == p.TC) {
            nb_tc = 0;
            if (++nb_tr == p.TR) { nb_tr = 0; ++nb_img; }
        }
    };

    f32x16 acc[PW][2];
#pragma unroll
    for (int i = 0; i < PW; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;

    // ---- fragment addresses.  Lane: k row tk = 8 h + q of the k16-step (second read: + 4 rows), columns 16 G1 + 4 p2 ..
    // + 3 of the tile's 32 (flattened j for A, output channel for B)
    const int G1 = (lane >> 4) & 1, q = (lane & 15) >> 2, p2 = lane & 3;
    const int tk = 8 * h + q;
    const unsigned xrow = (unsigned)p.C8 * 2u;                                  // bytes between consecutive pixels
    const unsigned a_lane = lds0 + (unsigned)tk * xrow + (unsigned)(16 * G1 + 4 * p2) * 2u;
    const unsigned b_lane = lds0 + (unsigned)p.xstage + (unsigned)tk * 128u + (unsigned)(16 * G1 + 4 * p2) * 2u;
    // this wave's pairs: pair index wave + 8 i -> (filter row, j tile): byte offset of its A tile inside a stage
    unsigned a_off[PW];
    int p_kh[PW], p_jt[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int pi = wave + 8 * i;
        const int khl = pi / p.Jt, jt = pi - khl * p.Jt;
        p_kh[i] = kh_lo + khl; p_jt[i] = jt;
        a_off[i] = (unsigned)(p_kh[i] * p.rp) + (unsigned)jt * 64u;
    }

    if (nsteps > 0) {
        issue(); issue(); issue();
        __builtin_amdgcn_s_waitcnt(0x0070 | 8);         // vmcnt(8): stage 0 has landed (stages 1, 2 may be in flight)
        __syncthreads();
        int slot = 0;
        for (int s = 0; s < nsteps; ++s) {
            const unsigned sb = (unsigned)slot * (unsigned)p.tstage;
#pragma unroll
            for (int kk = 0; kk < BR; ++kk) {
                if (kk == BR - 1) {
                    // stage s+1 (issued two blocks ago) must have landed before the next block reads it; every wave is
                    // past block s-1, whose slot stage s+3 takes
                    __builtin_amdgcn_s_waitcnt(0x0070 | 4);     // vmcnt(4) lgkmcnt(0)
                    __syncthreads();
                    issue();
                }
                const bf16x8 b0 = tr_frag(b_lane + sb + (unsigned)(kk * 16) * 128u, 128u);
                const bf16x8 b1 = tr_frag(b_lane + sb + (unsigned)(kk * 16) * 128u + 64u, 128u);
#pragma unroll
                for (int i = 0; i < PW; ++i) {
                    if (wave + 8 * i < npairs) {
                        const bf16x8 af = tr_frag(a_lane + sb + a_off[i] + (unsigned)(kk * p.rp), xrow);
                        acc[i][0] = mfma32<H16>(af, b0, acc[i][0]);
                        acc[i][1] = mfma32<H16>(af, b1, acc[i][1]);
                    }
                }
            }
            slot = (slot + 1) & (RING - 1);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0070);     // drain the stages issued past the range before the LDS is released

    float* slab = p.slab + (size_t)split * 7 * p.JP * 64;
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        if (wave + 8 * i >= npairs) continue;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = p_jt[i] * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                slab[((size_t)p_kh[i] * p.JP + j) * 64 + c * 32 + l31] = acc[i][c][r];
            }
    }
}

// dw[kh][kw][c][n] (+)= sum over splits of slab[s][kh][kw * C8 + c][n], c < Cin: fixed order (slab group g of 8 sums slabs
// g, g + 8, ...; then the eight partial sums in order)
// flipT (head wgrad): the result is written as dw[6 - kh][6 - kw][n][c] with 4 columns c instead of dw[kh][kw][c][n]
__global__ void __launch_bounds__(512) stem_slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int S,
                                                               int JP, int C8, int Cin, int accumulate, int flipT) {
    __shared__ float part[8][64];
    const int n = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int t = blockIdx.x;                                   // over 49 * Cin
    const int c = t % Cin, tap = t / Cin;
    const int kh = tap / 7, kw = tap - kh * 7;
    const size_t o = ((size_t)kh * JP + kw * C8 + c) * 64 + n, ss = (size_t)7 * JP * 64;
    float a0 = 0.f, a1 = 0.f;
    int s = g;
    for (; s + 8 < S; s += 16) {
        const float u = slab[(size_t)s * ss + o], v = slab[(size_t)(s + 8) * ss + o];
        a0 += u;
        a1 += v;
    }
    if (s < S) a0 += slab[(size_t)s * ss + o];
    part[g][n] = a0 + a1;
    __syncthreads();
    if (g == 0) {
        float a = part[0][n];
#pragma unroll
        for (int q = 1; q < 8; ++q) a += part[q][n];
        const int i = flipT ? (((6 - kh) * 7 + (6 - kw)) * 64 + n) * 4 + c : t * 64 + n;
        dw[i] = accumulate ? dw[i] + a : a;
    }
}

struct Plan { int Jt, JP, KG, S, bps, nblk, TR, TC, rp, xstage, tstage; };

bool plan(const mmh_conv_desc* d, int C8, Plan& q) {
    if (!d || d->kh != 7 || d->kw != 7 || d->stride != 1 || d->pad != 3 || d->Ho != d->H || d->Wo != d->W) return false;
    if (d->dtype != MMH_BF16 && d->dtype != MMH_FP16) return false;
    if (d->Cout != 64 || d->y_cs < 64 || d->y_cs % 8 || C8 % 8 || C8 < 8 || C8 > 48 || d->Cin > C8 || d->Cin < 1) return false;
    if (d->pad_mode == MMH_PAD_REFLECT && (d->H < 4 || d->W < 4)) return false;
    q.Jt = (7 * C8 + 31) / 32; q.JP = 32 * q.Jt;
    q.KG = 7 * q.Jt > 8 * PW ? 2 : 1;
    if ((q.KG == 2 ? 4 : 7) * q.Jt > 8 * PW) return false;
    q.TR = (d->H + BR - 1) / BR; q.TC = (d->W + BC - 1) / BC;
    q.nblk = d->B * q.TR * q.TC;
    int S = std::max(1, 256 / q.KG);
    S = (int)std::min<long long>(S, std::max<long long>(1, q.nblk / 4));
    q.bps = (q.nblk + S - 1) / S;
    q.S = (q.nblk + q.bps - 1) / q.bps;
    q.rp = HC * C8 * 2;
    // the transposed reads of the last j tile run up to 32 * Jt - 7 * C8 elements + one pixel past the last row's end
    q.xstage = XROUNDS * 8192;
    q.tstage = q.xstage + DSTAGE;
    return HR * q.rp + 1024 <= q.xstage;
}

}  // namespace

int mmh_wgrad_stem_lp16_supported(const mmh_conv_desc* d, int C8) {
    Plan q;
    return plan(d, C8, q) ? 1 : 0;
}

size_t mmh_wgrad_stem_lp16_ws_bytes(const mmh_conv_desc* d, int C8) {
    Plan q;
    if (!plan(d, C8, q)) return 0;
    return (size_t)q.S * 7 * q.JP * 64 * sizeof(float);
}

namespace {

// dy fp32 [B][H][W][cs] (4 channels) -> E 16-bit [B][H + 6][W + 6][8]: dy at offset (3, 3), zeros around and in channels 4..7
__global__ void head_wgrad_embed_kernel(const float* __restrict__ dy, int B, int H, int W, int cs, int h16,
                                        uint4* __restrict__ E) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int Hp = H + 6, Wp = W + 6;
    if (i >= (int64_t)B * Hp * Wp) return;
    const int x = (int)(i % Wp) - 3;
    const int64_t t = i / Wp;
    const int y = (int)(t % Hp) - 3, b = (int)(t / Hp);
    uint4 o = make_uint4(0u, 0u, 0u, 0u);
    if (y >= 0 && y < H && x >= 0 && x < W) {
        const float4 v = *reinterpret_cast<const float4*>(dy + ((size_t)(b * H + y) * W + x) * cs);
        auto cv = [&](float f) -> unsigned {
            return h16 ? (unsigned)__builtin_bit_cast(unsigned short, (_Float16)f)
                       : (unsigned)__builtin_bit_cast(unsigned short, (__bf16)f);
        };
        o.x = cv(v.x) | (cv(v.y) << 16);
        o.y = cv(v.z) | (cv(v.w) << 16);
    }
    E[i] = o;
}

// d: the stem-shaped problem (x = 16-bit [B][H][W][C8], 64 dy channels).  dyo / dyH / dyW: see StemWgKP
int launch_stem_wgrad(const mmh_conv_desc* d, const Plan& q, const void* x16p, int C8, const void* dy16, int dyH, int dyW,
                      int dyo, void* dw, void* slab, int accumulate, int flipT, const void* zeros, hipStream_t st) {
    StemWgKP p{};
    p.x = static_cast<const char*>(x16p); p.dy = static_cast<const char*>(dy16); p.zeros = static_cast<const char*>(zeros);
    p.slab = static_cast<float*>(slab);
    p.B = d->B; p.H = d->H; p.W = d->W; p.C8 = C8; p.dy_cs = d->y_cs; p.reflect = d->pad_mode == MMH_PAD_REFLECT ? 1 : 0;
    p.TR = q.TR; p.TC = q.TC; p.nblk = q.nblk; p.bps = q.bps; p.S = q.S; p.KG = q.KG; p.Jt = q.Jt; p.JP = q.JP;
    p.rp = q.rp; p.xstage = q.xstage; p.tstage = q.tstage;
    p.dyH = dyH; p.dyW = dyW; p.dyo = dyo;
    const int lds = RING * q.tstage;
    static int ready = -1;
    if (ready != 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_stem_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, RING * (XROUNDS * 8192 + DSTAGE));
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_stem_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, RING * (XROUNDS * 8192 + DSTAGE));
        ready = e == hipSuccess ? 0 : mmh::fail("wgrad_stem_kernel: %s", hipGetErrorString(e));
    }
    if (ready != 0) return ready;
    const int per_xcd = (q.S * q.KG + 7) / 8;
    if (d->dtype == MMH_FP16) hipLaunchKernelGGL(wgrad_stem_kernel<true>, dim3(8 * per_xcd), dim3(512), lds, st, p);
    else hipLaunchKernelGGL(wgrad_stem_kernel<false>, dim3(8 * per_xcd), dim3(512), lds, st, p);
    hipLaunchKernelGGL(stem_slab_reduce_kernel, dim3(49 * d->Cin), dim3(512), 0, st, static_cast<const float*>(slab),
                       static_cast<float*>(dw), q.S, q.JP, C8, d->Cin, accumulate, flipT);
    return mmh::check_launch("wgrad_stem_kernel");
}

// the head conv d (7x7 / reflect pad 3 / 64 -> 4 incl. padding) as the stem-shaped problem e on the padded domain
bool head_wgrad_desc(const mmh_conv_desc* d, mmh_conv_desc& e, Plan& q) {
    if (!d || d->kh != 7 || d->kw != 7 || d->stride != 1 || d->pad != 3 || d->pad_mode != MMH_PAD_REFLECT) return false;
    if (d->Cin != 64 || d->Cout != 4 || d->Ho != d->H || d->Wo != d->W || d->H < 8 || d->W < 8) return false;
    if (d->dtype != MMH_BF16 && d->dtype != MMH_FP16) return false;
    if (d->y_cs < 4 || d->y_cs % 4 || d->x_cs < 64 || d->x_cs % 8) return false;
    e = *d;
    e.H = e.Ho = d->H + 6; e.W = e.Wo = d->W + 6; e.Cin = 4; e.Cout = 64; e.pad_mode = MMH_PAD_ZERO; e.x_cs = 8;
    e.y_cs = d->x_cs;
    return plan(&e, 8, q) && (long long)e.B * e.H * e.W * std::max(8, e.y_cs) < (1ll << 31);
}

}  // namespace

int mmh_wgrad_stem_lp16(const mmh_conv_desc* d, const void* x16p, int C8, const void* dy16, void* dw, void* ws,
                        size_t ws_bytes, int accumulate, const void* zeros, mmh_stream_t s) {
    Plan q;
    MMH_REQUIRE(plan(d, C8, q) && x16p && dy16 && dw && ws && zeros,
                "mmh_wgrad_stem_lp16: 7x7 / stride 1 / pad 3, Cout == 64, C8 %% 8 == 0 in 8..48, Cin <= C8, 16-bit dtype");
    MMH_REQUIRE(ws_bytes >= mmh_wgrad_stem_lp16_ws_bytes(d, C8), "mmh_wgrad_stem_lp16: workspace too small");
    MMH_REQUIRE((long long)d->B * d->H * d->W * std::max(C8, d->y_cs) < (1ll << 31), "mmh_wgrad_stem_lp16: tensor too large");
    return launch_stem_wgrad(d, q, x16p, C8, dy16, d->H, d->W, 0, dw, ws, accumulate, 0, zeros, mmh::as_stream(s));
}

// ---- the Generator head's weight gradient (ReflectionPad2d(3) + Conv2d(64, 3, 7), models/Generator.py:254-259) ----
//   dw[kh][kw][ci][co] = sum_p xpad[p + (kh, kw)][ci] * dy[p][co]
// With q = p + (kh, kw) running over the padded domain D = (H + 6) x (W + 6) and E = dy embedded in D at offset (3, 3) (zeros
// around, 4 -> 8 channels) this is  sum_q xpad[q][ci] * E[q + (3 - kh, 3 - kw)][co]: the STEM's weight gradient on D with E
// as the 8-channel input, the 64 channels of xpad as the output gradient and the taps mirrored - so the stem kernel runs it
// (the reflect padding of x is folded into its dy loader's source addresses; nothing is materialised but E), and the slab
// reduction writes the mirrored, transposed result.  Both operands in 16 bits, fp32 accumulation (it was an fp32 vector-ALU
// kernel reading the fp32 x: 0.7 ms at B = 32, 256 x 256).
int mmh_conv7_head_wgrad_lp16_supported(const mmh_conv_desc* d) {
    mmh_conv_desc e;
    Plan q;
    return head_wgrad_desc(d, e, q) ? 1 : 0;
}

size_t mmh_conv7_head_wgrad_lp16_ws_bytes(const mmh_conv_desc* d) {
    mmh_conv_desc e;
    Plan q;
    if (!head_wgrad_desc(d, e, q)) return 0;
    const size_t embed = ((size_t)e.B * e.H * e.W * 16 + 255) & ~(size_t)255;
    return embed + (size_t)q.S * 7 * q.JP * 64 * sizeof(float);
}

int mmh_conv7_head_wgrad_lp16(const mmh_conv_desc* d, const void* x16, const void* dy, void* dw, void* ws, size_t ws_bytes,
                              int accumulate, const void* zeros, mmh_stream_t s) {
    mmh_conv_desc e;
    Plan q;
    MMH_REQUIRE(head_wgrad_desc(d, e, q),
                "mmh_conv7_head_wgrad_lp16: the head conv only (7x7 / reflect pad 3 / 64 -> 4, 16-bit dtype)");
    MMH_REQUIRE(x16 && dy && dw && ws && zeros && ws_bytes >= mmh_conv7_head_wgrad_lp16_ws_bytes(d) &&
                    (reinterpret_cast<uintptr_t>(ws) & 15) == 0,
                "mmh_conv7_head_wgrad_lp16: bad buffers (ws 16-byte aligned, mmh_conv7_head_wgrad_lp16_ws_bytes)");
    hipStream_t st = mmh::as_stream(s);
    const size_t px = (size_t)e.B * e.H * e.W;
    char* E = static_cast<char*>(ws);
    char* slab = E + ((px * 16 + 255) & ~(size_t)255);
    hipLaunchKernelGGL(head_wgrad_embed_kernel, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, st,
                       static_cast<const float*>(dy), d->B, d->H, d->W, d->y_cs, d->dtype == MMH_FP16 ? 1 : 0,
                       reinterpret_cast<uint4*>(E));
    if (int rc = mmh::check_launch("head_wgrad_embed_kernel")) return rc;
    return launch_stem_wgrad(&e, q, E, 8, x16, d->H, d->W, 3, dw, slab, accumulate, 1, zeros, st);
}
