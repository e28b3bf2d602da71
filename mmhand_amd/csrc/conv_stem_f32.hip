// fp32 fprop of the 7x7 / stride-1 / pad-3 stems (4 .. 48 input channels, 64 output channels; models/Generator.py:158-164,
// models/Discriminator.py:60-64: ReflectionPad2d(3) + Conv2d(c, 64, 7)) with the input halo resident in LDS and the filter's
// COLUMN taps flattened into the contraction, as conv_stem16.hip does in 16 bits.
//
// The generic route (conv_igemm_kernel) gathers an im2col tile per k-step through registers into LDS: 118 - 121 TFLOP/s
// (0.75 - 0.77 of the fp32 MFMA peak) at 24 / 44 channels, 100 (0.63) at 8; ablating its global loads gives 137.  In NHWC a
// row of the input is one contiguous array and the 7 x Cin window of output pixel ow under filter row kh starts at element
// ow * Cin of it, so per filter row
//
//   y[ow][n] += sum_j R_kh[ow * Cin + j] * w[kh][j][n],     j = kw * Cin + c     (w[kh] IS [7 Cin][64] in memory)
//
// is a GEMM whose pixel operand is read straight from the halo row - ds_read_b128 of four consecutive j at pixel pitch
// Cin * 4 bytes - and whose filter operand is a plain copy of w[kh]: ds_read_b32 of 32 consecutive output channels.  No
// im2col tile, no transposed or swizzled staging, no prepared weights.
//
// Work-group = 512 threads = 8 waves, output tile 8 rows x 16 pixels x 64 channels: wave = (2 rows x 16 pixels) x 32 channels
// = one 32x32 accumulator tile.  The 14 x 22 pixel halo (reflect / zero padding folded into the DMA's source addresses) is
// double-buffered where LDS allows (the next tile's halo is requested when the current tile starts); the filter streams in
// phases of up to 160 j (at most 40 KiB, two stages).  Per 8-deep k-step: 1 ds_read_b128 + 4 ds_read_b32 with immediate offsets and
// 4 v_mfma_f32_32x32x2_f32.  Epilogue: bias, activation, and - for the norm layer behind the stem - the (count, mean, M2)
// partials of each wave's 32 pixels per channel (mmh_conv2d_fprop_stats).  Persistent, XCD-contiguous tile lists.
#include <algorithm>
#include "common.h"

namespace mmh { int g_stem_f32 = 1; int g_stem_f32_dbg = 0; int g_stem_f32_levels = 1; }   // levels 2: with mmh_set_option("conv_levels", 2)

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const f32x4 __attribute__((address_space(3))) * lds_f4_p;
typedef const float __attribute__((address_space(3))) * lds_f_p;
__device__ __forceinline__ f32x4 lds_f4(unsigned addr, int imm) { return *reinterpret_cast<lds_f4_p>((size_t)(addr + (unsigned)imm)); }
__device__ __forceinline__ float lds_f(unsigned addr, int imm) { return *reinterpret_cast<lds_f_p>((size_t)(addr + (unsigned)imm)); }

constexpr int NT = 512;
constexpr int TC = 16, HC = TC + 6;             // output tile: 8 MT rows x 16 pixels; halo (8 MT + 6) x 22 pixels
constexpr int PJ_MAX = 160;                     // filter rows (j) per phase at most: a 40 KiB stage

__device__ char g_zero_line[128];

struct StemF32KP {
    const float* x;         // [B][H][W][x_cs]
    const float* w;         // [7][7 Cin][64]
    const float* bias;
    float* y;               // [B][H][W][y_cs]
    float* stats;           // [tiles * 4][3][64] or null
    int B, H, W, Cin, x_cs, y_cs, reflect, act;
    int J;                  // 7 Cin
    int parts;              // phases per filter row
    int pj;                 // filter rows (j) per phase, a multiple of 8
    int wst_b, wrounds;     // LDS bytes / DMA rounds of one filter stage
    int rp;                 // bytes per halo row = 22 Cin 4
    int halo_b;             // LDS bytes of one halo buffer (whole DMA rounds)
    int hrounds;            // DMA rounds per halo
    int nhalo;              // 1 or 2 halo buffers
    int TX, TY, tiles, per_xcd, slots, dbg;
};

__device__ __forceinline__ float act_of(float v, int act) {
    if (act == MMH_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == MMH_ACT_TANH) return tanhf(v);
    return v;
}

// MT: 32-pixel accumulator tiles per wave: output tile 8 MT rows x 16 pixels (wave: rows 2 pg, 2 pg + 1 of each half of 8)
// LEVELS = 2 (mmh_set_option("conv_levels", 2), the accuracy modes of ops.set_winograd_mode): two-level summation - every
// filter phase (<= 160 of the 7 x 7 Cin contraction values) runs its own MFMA chain in `part`, folded into the totals by
// vector adds - instead of one k-ordered chain up to 2156 deep (conv_igemm.hip: conv_igemm_levels2_kernel)
template <int MT, int LEVELS = 1>
__global__ void __launch_bounds__(NT, 1) conv_stem_f32_kernel(const StemF32KP p) {
    constexpr int TR = 8 * MT, HR = TR + 6;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, kg = lane >> 5;
    const int pg = wave & 3, nh = wave >> 2;
    const unsigned lds0 = mmh::lds_addr_of(smem);
    const unsigned wdst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);
    const unsigned w_lds = (unsigned)(p.nhalo * p.halo_b);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int t_end = min(p.tiles, (xcd + 1) * p.per_xcd);
    const int tpi = p.TX * p.TY;
    const int c4 = p.Cin >> 2, cpr = HC * c4, units = HR * cpr;     // 16-byte chunks per pixel / halo row / halo
    const void* const zero = g_zero_line + (lane & 7) * 16;

    // phase q = (filter row, part): rows j0 .. j0 + 159 of w[kh] -> stage st (the stages alternate ACROSS tiles: a tile may
    // have an odd number of phases); rows past 7 Cin are zero
    const int nq = 7 * p.parts;
    auto issue_w = [&](int q, int st) {
        const int kh = q / p.parts, part = q - kh * p.parts;
        const int j0 = part * p.pj;
        const float* src = p.w + ((size_t)kh * p.J + j0) * 64;
        const int rows = min(p.pj, p.J - j0);
        const unsigned dst = wdst + w_lds + (unsigned)(st * p.wst_b);
        for (int rr = 0; rr < p.wrounds; ++rr) {
            const int u = rr * NT + tid;
            mmh::lds_dma16((u >> 4) < rows ? (const void*)(src + u * 4) : zero, dst + (unsigned)(rr * NT * 16));
        }
    };
    auto issue_halo = [&](int tile, int buf) {
        const bool live = tile < t_end;
        const int b = tile / tpi;
        const int trem = tile - b * tpi;
        const int ty = trem / p.TX, tx = trem - ty * p.TX;
        for (int rr = 0; rr < p.hrounds; ++rr) {
            const int u = rr * NT + tid;
            const int hrow = u / cpr, c = u - hrow * cpr;
            const int px = c / c4, ck = c - px * c4;
            int ih = ty * TR - 3 + hrow, iw = tx * TC - 3 + px;
            if (p.reflect) {
                ih = ih < 0 ? -ih : ih;
                iw = iw < 0 ? -iw : iw;
                ih = ih >= p.H ? 2 * (p.H - 1) - ih : ih;
                iw = iw >= p.W ? 2 * (p.W - 1) - iw : iw;
            }
            const bool ok = live && u < units && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            const void* g = ok ? (const void*)(p.x + ((size_t)(b * p.H + ih) * p.W + iw) * (size_t)p.x_cs + 4 * ck) : zero;
            mmh::lds_dma16(g, wdst + (unsigned)(buf * p.halo_b) + (unsigned)(rr * NT * 16));
        }
    };

    // lane constants: pixel fragment (MFMA row m = tile row 2 pg + m / 16, pixel m % 16; j slice 4 kg), filter fragment
    // (row 4 kg + t of the k-step, channel 32 nh + m)
    const unsigned a_lane = lds0 + (unsigned)((2 * pg + (m >> 4)) * p.rp + ((m & 15) * p.Cin + 4 * kg) * 4);
    const unsigned b_lane = lds0 + w_lds + (unsigned)((4 * kg * 64 + 32 * nh + m) * 4);
    const int n = 32 * nh + m;
    const float bv = p.bias ? p.bias[n] : 0.f;

    int tile = xcd * p.per_xcd + slot;
    if (tile >= t_end) return;
    issue_halo(tile, 0);
    issue_w(0, 0);
    int buf = 0, st = 0;
    for (; tile < t_end; tile += p.slots) {
        f32x16 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
        for (int q = 0; q < nq; ++q) {
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0x0070);                 // this thread's DMA (halo, phase q) has landed
            __builtin_amdgcn_s_barrier();                       // ... everybody's; stage (q + 1) & 1 is no longer read
            asm volatile("" ::: "memory");
            if (!(p.dbg & 1)) issue_w(q + 1 < nq ? q + 1 : 0, st ^ 1);
            if (q == 0 && p.nhalo == 2 && !(p.dbg & 2)) issue_halo(tile + p.slots, buf ^ 1);
            const int kh = q / p.parts, part = q - kh * p.parts;
            const int kcs = (min(p.pj, p.J - part * p.pj) + 7) >> 3;
            f32x16 chain[MT];       // LEVELS == 2: this phase's own MFMA chains
            if (LEVELS == 2) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) chain[mt][i] = 0.f;
            }
            auto mf = [&](int mt, float a, float b) {
                if (LEVELS == 2) chain[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, chain[mt], 0, 0, 0);
                else acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[mt], 0, 0, 0);
            };
            unsigned ab = a_lane + (unsigned)(buf * p.halo_b + kh * p.rp + part * p.pj * 4);
            unsigned bb = b_lane + (unsigned)(st * p.wst_b);
            st ^= 1;
            // fragments of k-step kc + 1 are requested before k-step kc multiplies
            f32x4 a0[MT], a1[MT];
            float b0[4], b1[4];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a0[mt] = lds_f4(ab + (unsigned)(mt * 8 * p.rp), 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) b0[t] = lds_f(bb, 256 * t);
            int kc = 0;
            if (p.dbg & 4) {
                for (; kc < kcs; ++kc) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            mf(mt, a0[mt][t], b0[t]);
                }
            }
            for (; kc + 2 <= kcs; kc += 2) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a1[mt] = lds_f4(ab + (unsigned)(mt * 8 * p.rp), 32);
#pragma unroll
                for (int t = 0; t < 4; ++t) b1[t] = lds_f(bb, 2048 + 256 * t);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        mf(mt, a0[mt][t], b0[t]);
                ab += 64;
                bb += 4096;
                // past the last k-step: inside the stage / halo buffer, never multiplied
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a0[mt] = lds_f4(ab + (unsigned)(mt * 8 * p.rp), 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) b0[t] = lds_f(bb, 256 * t);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        mf(mt, a1[mt][t], b1[t]);
            }
            if (kc < kcs) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        mf(mt, a0[mt][t], b0[t]);
            }
            if (LEVELS == 2) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[mt][i] += chain[mt][i];
            }
        }
        const int b = tile / tpi;
        const int trem = tile - b * tpi;
        const int ty = trem / p.TX, tx = trem - ty * p.TX;
        if (p.nhalo == 1) {
            __builtin_amdgcn_s_barrier();                       // every wave is done with the only halo buffer
            if (!(p.dbg & 2)) issue_halo(tile + p.slots, 0);
        } else {
            buf ^= 1;
        }
        if (p.dbg & 8) continue;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            // acc[mt][i]: channel n, MFMA row (i & 3) + 8 (i >> 2) + 4 kg of tile rows 8 mt + 2 pg, + 1
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int mm = (i & 3) + 8 * (i >> 2) + 4 * kg;
                const int oh = ty * TR + 8 * mt + 2 * pg + (mm >> 4), ow = tx * TC + (mm & 15);
                if (oh < p.H && ow < p.W)
                    p.y[((size_t)(b * p.H + oh) * p.W + ow) * (size_t)p.y_cs + n] = act_of(acc[mt][i] + bv, p.act);
            }
        }
        if (p.stats) {
            // (count, mean, M2) of this wave's 32 MT pixels per channel: a lane holds 16 of each accumulator tile, lane ^ 32
            // the other 16; Chan merges of equal counts
            float mean_w = 0.f, m2_w = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float sum = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) sum += acc[mt][i] + bv;
                const float mean_l = sum * (1.f / 16.f);
                float m2 = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float dv = acc[mt][i] + bv - mean_l;
                    m2 += dv * dv;
                }
                const float mean_o = __shfl_xor(mean_l, 32, 64), m2_o = __shfl_xor(m2, 32, 64);
                const float dm = mean_o - mean_l;
                const float mean_t = mean_l + 0.5f * dm, m2_t = m2 + m2_o + dm * dm * 8.f;      // 32 pixels
                if (mt == 0) {
                    mean_w = mean_t;
                    m2_w = m2_t;
                } else {
                    const float d2 = mean_t - mean_w;
                    mean_w += 0.5f * d2;
                    m2_w += m2_t + d2 * d2 * 16.f;                                              // 32 * 32 / 64
                }
            }
            if (kg == 0) {
                float* o = p.stats + ((size_t)(tile * 4 + pg) * 3) * 64 + n;
                o[0] = 32.f * MT;
                o[64] = mean_w;
                o[128] = m2_w;
            }
        }
    }
}

struct Plan { int mt, parts, pj, wst_b, wrounds, rp, halo_b, hrounds, nhalo, lds; };
// two 32-pixel tiles per wave (16 x 16 output pixels: half the filter traffic and filter reads per MFMA) and two halo buffers
// where 160 KiB of LDS allow; the filter phase shrinks (down to 40 j) before either is given up
bool plan(const mmh_conv_desc* d, Plan& q) {
    if (!d || d->dtype != MMH_F32 || d->kh != 7 || d->kw != 7 || d->stride != 1 || d->pad != 3 || d->Ho != d->H ||
        d->Wo != d->W || d->Cout != 64 || d->Cin % 4 || d->Cin < 4 || d->Cin > 48 || d->x_cs % 4 || d->x_cs < d->Cin ||
        d->y_cs < 64)
        return false;
    if (d->pad_mode == MMH_PAD_REFLECT && (d->H < 4 || d->W < 4)) return false;
    if ((size_t)d->B * d->H * d->W * std::max(d->x_cs, d->y_cs) >= (1ull << 31)) return false;
    const int J8 = (7 * d->Cin + 7) & ~7;
    q.rp = HC * d->Cin * 4;
    for (int mt = 2; mt >= 1; --mt)
        for (int nhalo = 2; nhalo >= 1; --nhalo) {
            // the last pixels' k-steps read up to 32 bytes x (k-steps of the last part) past their row: keep the tail inside
            const int bytes = (8 * mt + 6) * q.rp + 8 * 32;
            q.hrounds = (bytes + NT * 16 - 1) / (NT * 16);
            q.halo_b = q.hrounds * NT * 16;
            for (q.parts = (J8 + PJ_MAX - 1) / PJ_MAX; q.parts <= 8; ++q.parts) {
                q.pj = ((J8 + q.parts - 1) / q.parts + 7) & ~7;
                if (q.pj < 40 && q.parts > 1) break;
                q.wrounds = (q.pj * 256 + NT * 16 - 1) / (NT * 16);
                q.wst_b = q.wrounds * NT * 16;
                // + 4 KiB: the fragment prefetch of the k-step behind a phase's last one reads (never multiplies) past the stage
                q.lds = nhalo * q.halo_b + 2 * q.wst_b + 4096;
                if (q.lds <= 160 * 1024) {
                    q.mt = mt;
                    q.nhalo = nhalo;
                    return true;
                }
            }
        }
    return false;
}

}  // namespace

namespace mmh {

bool stem_f32_ok(const mmh_conv_desc* d) {
    Plan q;
    return g_stem_f32 && plan(d, q);
}

// partial rows written for mmh_conv2d_fprop_stats: one per wave row group of a full tile (32 or 64 pixels); 0 = ragged tiles
int stem_f32_stats_chunks(const mmh_conv_desc* d) {
    Plan q;
    if (!plan(d, q)) return 0;
    const int tr = 8 * q.mt;
    return (d->H % tr == 0 && d->W % TC == 0) ? d->B * (d->H / tr) * (d->W / TC) * 4 : 0;
}

int launch_stem_f32(const mmh_conv_desc* d, const void* x, const void* w, const void* bias, void* y, int act, float* stats,
                    hipStream_t st) {
    Plan q;
    MMH_REQUIRE(plan(d, q), "conv_stem_f32: unsupported shape");
    MMH_REQUIRE(!stats || stem_f32_stats_chunks(d) > 0, "conv_stem_f32: statistics need whole 8 x 16 tiles");
    StemF32KP p{};
    p.x = static_cast<const float*>(x);
    p.w = static_cast<const float*>(w);
    p.bias = static_cast<const float*>(bias);
    p.y = static_cast<float*>(y);
    p.stats = stats;
    p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.x_cs = d->x_cs; p.y_cs = d->y_cs;
    p.reflect = d->pad_mode == MMH_PAD_REFLECT; p.act = act;
    p.J = 7 * d->Cin; p.parts = q.parts; p.pj = q.pj; p.wst_b = q.wst_b; p.wrounds = q.wrounds; p.rp = q.rp; p.halo_b = q.halo_b; p.hrounds = q.hrounds; p.nhalo = q.nhalo;
    p.TY = (d->H + 8 * q.mt - 1) / (8 * q.mt);
    p.TX = (d->W + TC - 1) / TC;
    p.tiles = p.B * p.TX * p.TY;
    p.per_xcd = (p.tiles + 7) / 8;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return fail("conv_stem_f32: cannot query the device");
        cus = prop.multiProcessorCount;
    }
    p.slots = std::max(1, std::min(cus / 8, p.per_xcd));
    p.dbg = g_stem_f32_dbg;
    static int ready = -1;
    if (ready != 0) {
        for (const void* k : {reinterpret_cast<const void*>(conv_stem_f32_kernel<1>), reinterpret_cast<const void*>(conv_stem_f32_kernel<2>),
                              reinterpret_cast<const void*>(conv_stem_f32_kernel<1, 2>), reinterpret_cast<const void*>(conv_stem_f32_kernel<2, 2>)}) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return fail("conv_stem_f32: %s", hipGetErrorString(e));
        }
        ready = 0;
    }
    if (g_stem_f32_levels == 2) {
        if (q.mt == 2) hipLaunchKernelGGL((conv_stem_f32_kernel<2, 2>), dim3(8 * p.slots), dim3(NT), q.lds, st, p);
        else hipLaunchKernelGGL((conv_stem_f32_kernel<1, 2>), dim3(8 * p.slots), dim3(NT), q.lds, st, p);
    } else if (q.mt == 2) hipLaunchKernelGGL(conv_stem_f32_kernel<2>, dim3(8 * p.slots), dim3(NT), q.lds, st, p);
    else hipLaunchKernelGGL(conv_stem_f32_kernel<1>, dim3(8 * p.slots), dim3(NT), q.lds, st, p);
    return check_launch("conv_stem_f32_kernel");
}

}  // namespace mmh
