// 16-bit fprop of the 7x7 / stride-1 / pad-3 stems (3 .. 48 input channels padded to C8, 64 output channels) with the
// input halo resident in LDS and the filter's COLUMN taps flattened into the contraction
// (models/Generator.py:158-164, models/Discriminator.py:60-64; the generation path runs three of them per batch).
//
// conv_lp16f_kernel (conv_lp16.hip) stages, per 64-deep k-step, a [256 pixels][64] im2col tile gathered tap by tap:
// 40 KB of LDS-DMA per 2.1 MFLOP - it lives on DMA latency (585 TFLOP/s at 24 channels).  In NHWC a row of the input is
// one contiguous array and the window of output pixel ow under filter row kh - 7 x C8 values - starts at element
// ow * C8 of it (wgrad_stem.hip uses the same fact), so per filter row
//
//   y[ow][n] += sum_j  R_kh[ow * C8 + j] * Wf[kh][n][j],      j = kw * C8 + c  (zero-padded to JP = 32 * Jt)
//
// is a GEMM whose pixel operand is read straight from the flat halo row: the fragment of lane (pixel, k slice) is 16
// contiguous bytes at pixel pitch C8 * 2 (overlapping windows; conflict-free for C8 = 8, 24, two-way for 48).  No im2col.
//
// Work-group = 256 threads = 4 waves, output tile 16 rows x 16 pixels x 64 channels (wave: 4 rows).  Its 22 x 22 pixel
// halo (22 flat rows, <= 46 KiB; reflect / zero padding folded into the DMA's source addresses) is staged once; the
// filter is streamed one filter ROW at a time - [64 n][JP + 8] 16-bit, 8 - 45 KiB, two stages (the +8 element pitch makes
// the sixteen weight rows of a fragment land on distinct banks; the padded layout is prepared in global memory so the
// linear LDS-DMA reproduces it).  MFMA 16x16x32 with the weight fragment first: D[row = channel][column = pixel], so a
// lane stores four consecutive channels of a pixel (store4's layout).  72 KiB of LDS at 24 channels: two work-groups
// per CU.
#include <algorithm>
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const bf16x8 __attribute__((address_space(3))) * lds_frag_p;
__device__ __forceinline__ bf16x8 lds_frag(unsigned addr) { return *reinterpret_cast<lds_frag_p>((size_t)addr); }

constexpr int TS = 16;                      // output tile 16 x 16

struct Stem16KP {
    const char* x;          // 16-bit [B][H][W][C8]
    const char* w;          // 16-bit [7][64][wpitch / 2]
    const char* zeros;
    const float* bias;
    float* y;               // fp32 [B][H][W][y_cs] ...
    char* y16;              // ... or 16-bit
    int B, H, W, C8, y_cs, reflect, act;
    int ks, hs;             // filter size 7 (the stems) | 3 (VGG19's conv1_1, C8 = 8), halo size = 16 + ks - 1
    int Jt;                 // k-steps of 32 per filter row
    int rp;                 // bytes per halo row = 22 * C8 * 2
    int wpitch;             // bytes per weight row = (32 * Jt + 8) * 2
    int halo_b, wst_b;      // LDS bytes of the halo region / of one weight stage (whole DMA rounds)
    int TX, TY, tiles;
    float* stats;           // per (image, tile, wave, channel) count / mean / M2 of the stored outputs, [B][chunks][3][64], or null
};

template <bool H16>
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    if (H16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0,
                                                      0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// RW = output rows per wave: 4 (256 threads; two work-groups per CU where LDS allows) or 2 (512 threads: the 40-48 channel
// stems, whose 144 KiB of LDS admit one work-group per CU - eight waves instead of four to hide the fragment reads)
template <bool H16, int RW>
__global__ void __launch_bounds__(64 * (TS / RW), 2) conv_stem16_kernel(const Stem16KP p) {
    constexpr int NT = 64 * (TS / RW);          // threads
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int per_xcd = (p.tiles + 7) / 8;
    const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= p.tiles) return;
    const int b = tile / (p.TX * p.TY);
    const int trem = tile - b * (p.TX * p.TY);
    const int ty = trem / p.TX, tx = trem - ty * p.TX;
    const int oh0 = ty * TS, ow0 = tx * TS;
    const unsigned lds0 = mmh::lds_addr_of(smem);
    const unsigned wdst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);

    // filter row kh -> weight stage kh & 1 (linear copy of the padded [64][wpitch] image)
    const int w_units = 64 * p.wpitch / 16;
    auto issue_w = [&](int kh) {
        const char* src = p.w + (size_t)kh * 64 * p.wpitch;
        const unsigned dst = wdst + (unsigned)p.halo_b + (unsigned)((kh & 1) * p.wst_b);
        for (int r = 0; r * NT < w_units; ++r) {
            const int u = r * NT + tid;
            mmh::lds_dma16(u < w_units ? src + (size_t)u * 16 : p.zeros + (lane & 7) * 16, dst + (unsigned)(r * NT * 16));
        }
    };
    // halo: unit u of the flat image (22 rows x upr units): row u / upr, pixel (u % upr) / c8u, chunk (u % upr) % c8u;
    // units past the image (the rounds' tail) are zero-filled: the last pixels' padded k columns read into them
    {
        const int c8u = p.C8 / 8, upr = p.hs * c8u, units = p.hs * upr, hpad = p.ks >> 1;
        for (int r = 0; r * NT * 16 < p.halo_b; ++r) {
            const int u = r * NT + tid;
            const int row = u / upr, ur = u - row * upr;
            const int hx = ur / c8u, ck = ur - hx * c8u;
            int ih = oh0 + row - hpad, iw = ow0 + hx - hpad;
            if (p.reflect) {
                ih = ih < 0 ? -ih : ih;
                iw = iw < 0 ? -iw : iw;
                ih = ih >= p.H ? 2 * (p.H - 1) - ih : ih;
                iw = iw >= p.W ? 2 * (p.W - 1) - iw : iw;
            }
            const bool ok = u < units && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            const char* g = ok ? p.x + (size_t)((b * p.H + ih) * p.W + iw) * (size_t)(p.C8 * 2) + (unsigned)ck * 16u
                               : p.zeros + (lane & 7) * 16;
            mmh::lds_dma16(g, wdst + (unsigned)(r * NT * 16));
        }
    }
    issue_w(0);

    f32x4 acc[RW][4];
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // lane constants: pixel fragment (row 4 wave, pixel l15, k slice g4), weight fragment (row l15 of an n tile, k slice g4)
    const unsigned x_lane = lds0 + (unsigned)(RW * wave) * (unsigned)p.rp + (unsigned)(l15 * p.C8 + 8 * g4) * 2u;
    const unsigned w_lane = lds0 + (unsigned)p.halo_b + (unsigned)l15 * (unsigned)p.wpitch + (unsigned)(8 * g4) * 2u;
    const unsigned w_nt = 16u * (unsigned)p.wpitch;        // bytes between n tiles

    for (int kh = 0; kh < p.ks; ++kh) {
        __builtin_amdgcn_s_waitcnt(0x0070);                 // this thread's DMA (halo, filter row kh) has landed
        __syncthreads();                                    // ... everybody's; filter row kh - 1 is no longer read
        if (kh + 1 < p.ks) issue_w(kh + 1);
        const unsigned wb = w_lane + (unsigned)((kh & 1) * p.wst_b);
        const unsigned xb = x_lane + (unsigned)kh * (unsigned)p.rp;
        for (int ks = 0; ks < p.Jt; ++ks) {
            bf16x8 wf[4], xf[RW];
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = lds_frag(wb + (unsigned)j * w_nt + (unsigned)ks * 64u);
#pragma unroll
            for (int i = 0; i < RW; ++i) xf[i] = lds_frag(xb + (unsigned)i * (unsigned)p.rp + (unsigned)ks * 64u);
#pragma unroll
            for (int i = 0; i < RW; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<H16>(wf[j], xf[i], acc[i][j]);
        }
    }

    float bv[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[j][r] = p.bias ? p.bias[j * 16 + 4 * g4 + r] : 0.f;
    const int ow = ow0 + l15;
    // (16-byte stores after the lane-pair trade of common.h's pair_swap8: 24 -> 64 396 us against 348 with 8-byte stores - off)
    if (p.y16 && false) {
        const int cb0 = ((g4 & 1) ? 16 : 0) + 4 * (g4 & 2);
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const int oh = oh0 + RW * wave + i;
            const size_t m = ((size_t)b * p.H + oh) * p.W + ow;
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                float v[8];
                mmh::pair_swap8(acc[i][2 * jp], acc[i][2 * jp + 1], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float t = v[e] + (p.bias ? p.bias[jp * 32 + cb0 + e] : 0.f);
                    v[e] = p.act == MMH_ACT_RELU ? (t > 0.f ? t : 0.f) : (p.act == MMH_ACT_TANH ? tanhf(t) : t);
                }
                if (oh < p.H && ow < p.W) mmh::store8_lp16<H16>(p.y16 + (m * p.y_cs + (jp * 32 + cb0)) * 2, v);
            }
        }
    } else
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int oh = oh0 + RW * wave + i;
        if (oh >= p.H || ow >= p.W) continue;
        const size_t m = ((size_t)b * p.H + oh) * p.W + ow;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 v = acc[i][j];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = v[r] + bv[j][r];
                v[r] = p.act == MMH_ACT_RELU ? (t > 0.f ? t : 0.f) : (p.act == MMH_ACT_TANH ? tanhf(t) : t);
            }
            const size_t elem = m * p.y_cs + (j * 16 + 4 * g4);
            if (p.y16) {
                if (H16) {
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                    h4 o = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                    *reinterpret_cast<h4*>(p.y16 + elem * 2) = o;
                } else {
                    typedef __bf16 b4 __attribute__((ext_vector_type(4)));
                    b4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    *reinterpret_cast<b4*>(p.y16 + elem * 2) = o;
                }
            } else {
                *reinterpret_cast<f32x4*>(p.y + elem) = v;
            }
        }
    }
    if (p.stats) {      // full tiles, 16-bit output, no activation (host side): the norm behind the stem merges these partials
        constexpr int WPT = TS / RW;        // waves (= partials) per tile
        float* sp = p.stats + ((size_t)(b * (p.TX * p.TY * WPT) + (ty * p.TX + tx) * WPT + wave) * 3) * 64 + 4 * g4;
        mmh::wave_tile_stats<RW, 4>([&](int i, int j, int r) {
            const float t = acc[i][j][r] + bv[j][r];
            return H16 ? (float)(_Float16)t : (float)(__bf16)t;
        }, l15, sp, 64);
    }
}

// w fp32 [ks][ks][Cin][64] -> 16-bit [ks][64][pitch], k = kw * C8 + c, zero padded (pitch = 32 * Jt + 8 elements)
__global__ void prep_stem16_w_kernel(const float* __restrict__ w, int Cin, int C8, int pitch, int h16,
                                     unsigned short* __restrict__ out, int ks) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ks * 64 * pitch) return;
    const int j = i % pitch, n = (i / pitch) & 63, kh = i / (pitch * 64);
    const int kw = j / C8, c = j - kw * C8;
    const float v = (kw < ks && c < Cin) ? w[((size_t)(kh * ks + kw) * Cin + c) * 64 + n] : 0.f;
    out[i] = h16 ? __builtin_bit_cast(unsigned short, (_Float16)v) : __builtin_bit_cast(unsigned short, (__bf16)v);
}

// ---- the Generator head's input gradient as a stem-shaped convolution (mmh_conv7_head_dgrad_lp16) ----
// The head (ReflectionPad2d(3) + Conv2d(64, 3, 7), models/Generator.py:254-259) sends a 4-column gradient back to 64
// channels: dxpad[u][v][ci] = sum w[kh][kw][ci][co] dy[u - kh][v - kw][co] over the padded domain, then the pad ring folded
// back.  That is a 'same' zero-padded 7x7 conv from 4 (-> C8 = 8) to 64 channels of dy embedded in the padded domain, with
// the filter mirrored and transposed: the shape the stem kernel above is built for (as an fp32 implicit GEMM with 196-deep
// contraction plus an fp32 fold pass it cost 0.9 ms of the 16-bit step).

// head weight fp32 [7][7][64 ci][4 co] -> stem16 layout [7 kh'][64 n = ci][pitch], k = kw' * 8 + co, of the mirrored filter
__global__ void head_dgrad_w_kernel(const float* __restrict__ w, int pitch, int h16, unsigned short* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 7 * 64 * pitch) return;
    const int j = i % pitch, n = (i / pitch) & 63, kh = i / (pitch * 64);
    const int kw = j / 8, c = j - kw * 8;
    const float v = (kw < 7 && c < 4) ? w[((size_t)((6 - kh) * 7 + (6 - kw)) * 64 + n) * 4 + c] : 0.f;
    out[i] = h16 ? __builtin_bit_cast(unsigned short, (_Float16)v) : __builtin_bit_cast(unsigned short, (__bf16)v);
}

// dy fp32 [B][H][W][cs] (4 channels) -> E 16-bit [B][H + 6][W + 6][8]: dy at offset (3, 3), zeros around and in channels 4..7
__global__ void head_dgrad_embed_kernel(const float* __restrict__ dy, int B, int H, int W, int cs, int h16,
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
            return h16 ? (unsigned)__builtin_bit_cast(unsigned short, (_Float16)f) : (unsigned)__builtin_bit_cast(unsigned short, (__bf16)f);
        };
        o.x = cv(v.x) | (cv(v.y) << 16);
        o.y = cv(v.z) | (cv(v.w) << 16);
    }
    E[i] = o;
}

// transpose of ReflectionPad2d(3) on a 16-bit padded-domain gradient [B][H + 6][W + 6][64]: 8 channels per thread, fp32 sums,
// dx fp32 or 16-bit [B][H][W][cs]
template <bool H16>
__global__ void head_dgrad_fold_kernel(const uint4* __restrict__ dxp, int B, int H, int W, int cs, float* __restrict__ dx,
                                       char* __restrict__ dx16) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // (pixel, 8-channel group)
    if (i >= (int64_t)B * H * W * 8) return;
    const int c8 = (int)(i & 7);
    const int64_t px = i >> 3;
    const int w = (int)(px % W);
    const int64_t t = px / W;
    const int h = (int)(t % H), b = (int)(t / H);
    int ph[3], pw[3], nh = 0, nw = 0;
    ph[nh++] = h + 3;
    if (h >= 1 && h <= 3) ph[nh++] = 3 - h;
    if (h >= H - 4 && h <= H - 2) ph[nh++] = 2 * (H - 1) - h + 3;
    pw[nw++] = w + 3;
    if (w >= 1 && w <= 3) pw[nw++] = 3 - w;
    if (w >= W - 4 && w <= W - 2) pw[nw++] = 2 * (W - 1) - w + 3;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < nh; ++a)
        for (int c = 0; c < nw; ++c) {
            const uint4 v = dxp[((size_t)(b * (H + 6) + ph[a]) * (W + 6) + pw[c]) * 8 + c8];
            const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned short lo = (unsigned short)(u[e] & 0xffffu), hi = (unsigned short)(u[e] >> 16);
                s[2 * e] += H16 ? (float)__builtin_bit_cast(_Float16, lo) : (float)__builtin_bit_cast(__bf16, lo);
                s[2 * e + 1] += H16 ? (float)__builtin_bit_cast(_Float16, hi) : (float)__builtin_bit_cast(__bf16, hi);
            }
        }
    const size_t o = (size_t)px * cs + c8 * 8;
    if (dx16) {
        unsigned q[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned lo = H16 ? (unsigned)__builtin_bit_cast(unsigned short, (_Float16)s[2 * e])
                                    : (unsigned)__builtin_bit_cast(unsigned short, (__bf16)s[2 * e]);
            const unsigned hi = H16 ? (unsigned)__builtin_bit_cast(unsigned short, (_Float16)s[2 * e + 1])
                                    : (unsigned)__builtin_bit_cast(unsigned short, (__bf16)s[2 * e + 1]);
            q[e] = lo | (hi << 16);
        }
        *reinterpret_cast<uint4*>(dx16 + o * 2) = make_uint4(q[0], q[1], q[2], q[3]);
    } else {
        *reinterpret_cast<float4*>(dx + o) = make_float4(s[0], s[1], s[2], s[3]);
        *reinterpret_cast<float4*>(dx + o + 4) = make_float4(s[4], s[5], s[6], s[7]);
    }
}

struct Plan { int Jt, pitch, rp, halo_b, wst_b, lds, rw, ks, hs; };

// 7x7 / pad 3 (the stems), or 3x3 / pad 1 with C8 == 8 (VGG19's conv1_1, losses/L1_plus_perceptualLoss.py:22-27: 3 -> 64 at
// full resolution; the flat-K kernel padded its 72-deep contraction to 128 and gathered an im2col tile per k-step)
bool plan(const mmh_conv_desc* d, int C8, Plan& q) {
    if (!d || d->kh != d->kw || d->stride != 1 || d->Ho != d->H || d->Wo != d->W) return false;
    if (!((d->kh == 7 && d->pad == 3) || (d->kh == 3 && d->pad == 1 && C8 == 8))) return false;
    q.ks = d->kh; q.hs = TS + d->kh - 1;
    if (d->dtype != MMH_BF16 && d->dtype != MMH_FP16) return false;
    if (d->Cout != 64 || d->y_cs < 64 || d->y_cs % 4 || C8 % 8 || C8 < 8 || C8 > 48 || d->Cin > C8 || d->Cin < 1) return false;
    if (d->pad_mode == MMH_PAD_REFLECT && (d->H < 4 || d->W < 4)) return false;
    q.Jt = (q.ks * C8 + 31) / 32;
    q.pitch = 32 * q.Jt + 8;
    q.rp = q.hs * C8 * 2;
    // regions in whole DMA rounds (threads x 16 bytes): 4 KiB for the four-wave kernel; where two of its work-groups do
    // not fit a CU anyway, the eight-wave kernel (8 KiB rounds)
    for (q.rw = 4; q.rw >= 2; q.rw -= 2) {
        const int rb = q.rw == 4 ? 4096 : 8192;
        q.halo_b = (q.hs * q.rp + 256 + rb - 1) / rb * rb;     // + the last pixels' padded k columns
        q.wst_b = (64 * q.pitch * 2 + rb - 1) / rb * rb;
        q.lds = q.halo_b + 2 * q.wst_b;
        if (q.rw == 4 && q.lds <= 80 * 1024) return true;
    }
    q.rw = 2;
    return q.lds <= 160 * 1024;
}

}  // namespace

int mmh_conv_stem16_supported(const mmh_conv_desc* d, int C8) {
    Plan q;
    return plan(d, C8, q) ? 1 : 0;
}

size_t mmh_conv_stem16_weights_bytes_k(int C8, int ks) {
    if (C8 % 8 || C8 < 8 || C8 > 48 || !(ks == 7 || (ks == 3 && C8 == 8))) return 0;
    return (size_t)ks * 64 * (32 * ((ks * C8 + 31) / 32) + 8) * 2;
}

size_t mmh_conv_stem16_weights_bytes(int C8) { return mmh_conv_stem16_weights_bytes_k(C8, 7); }

int mmh_prep_weights_stem16_k(const void* w, int Cin, int C8, int ks, int dtype, void* out, mmh_stream_t s) {
    MMH_REQUIRE(w && out && mmh_conv_stem16_weights_bytes_k(C8, ks) > 0 && Cin >= 1 && Cin <= C8 &&
                    (dtype == MMH_BF16 || dtype == MMH_FP16),
                "mmh_prep_weights_stem16: bad arguments (C8 %% 8 == 0 in 8..48, Cin <= C8, 7x7 | 3x3 at C8 == 8, 16-bit dtype)");
    const int pitch = 32 * ((ks * C8 + 31) / 32) + 8;
    const int n = ks * 64 * pitch;
    hipLaunchKernelGGL(prep_stem16_w_kernel, dim3((n + 255) / 256), dim3(256), 0, mmh::as_stream(s),
                       static_cast<const float*>(w), Cin, C8, pitch, dtype == MMH_FP16 ? 1 : 0,
                       static_cast<unsigned short*>(out), ks);
    return mmh::check_launch("prep_stem16_w_kernel");
}

int mmh_prep_weights_stem16(const void* w, int Cin, int C8, int dtype, void* out, mmh_stream_t s) {
    return mmh_prep_weights_stem16_k(w, Cin, C8, 7, dtype, out, s);
}

static int conv_stem16_impl(const mmh_conv_desc* d, const void* x16p, int C8, const void* w_stem16, const void* bias, void* y,
                           int y_is16, int act, const void* zeros, void* stats, mmh_stream_t s);

int mmh_conv_stem16(const mmh_conv_desc* d, const void* x16p, int C8, const void* w_stem16, const void* bias, void* y,
                    int y_is16, int act, const void* zeros, mmh_stream_t s) {
    return conv_stem16_impl(d, x16p, C8, w_stem16, bias, y, y_is16, act, zeros, nullptr, s);
}

int mmh_conv_stem16_stats_chunks(const mmh_conv_desc* d, int C8) {
    Plan q;
    if (!d || !plan(d, C8, q) || d->H % TS || d->W % TS) return 0;
    return (d->H / TS) * (d->W / TS) * (TS / q.rw);
}

int mmh_conv_stem16_stats(const mmh_conv_desc* d, const void* x16p, int C8, const void* w_stem16, const void* bias, void* y16,
                          void* stats, const void* zeros, mmh_stream_t s) {
    MMH_REQUIRE(mmh_conv_stem16_stats_chunks(d, C8) > 0 && stats, "mmh_conv_stem16_stats: H, W multiples of 16 and a stats buffer");
    return conv_stem16_impl(d, x16p, C8, w_stem16, bias, y16, 1, MMH_ACT_NONE, zeros, stats, s);
}

static int conv_stem16_impl(const mmh_conv_desc* d, const void* x16p, int C8, const void* w_stem16, const void* bias, void* y,
                           int y_is16, int act, const void* zeros, void* stats, mmh_stream_t s) {
    Plan q;
    MMH_REQUIRE(plan(d, C8, q) && x16p && w_stem16 && y && zeros,
                "mmh_conv_stem16: 7x7 / pad 3 (or 3x3 / pad 1 at C8 == 8), stride 1, Cout == 64, C8 %% 8 == 0 in 8..48, Cin <= C8, "
                "16-bit dtype");
    MMH_REQUIRE((reinterpret_cast<uintptr_t>(y) & 15) == 0 && (long long)d->B * d->H * d->W * std::max(C8, d->y_cs) < (1ll << 31),
                "mmh_conv_stem16: y must be 16-byte aligned; tensor too large");
    Stem16KP p{};
    p.x = static_cast<const char*>(x16p); p.w = static_cast<const char*>(w_stem16); p.zeros = static_cast<const char*>(zeros);
    p.bias = static_cast<const float*>(bias);
    if (y_is16) p.y16 = static_cast<char*>(y); else p.y = static_cast<float*>(y);
    p.B = d->B; p.H = d->H; p.W = d->W; p.C8 = C8; p.y_cs = d->y_cs; p.reflect = d->pad_mode == MMH_PAD_REFLECT ? 1 : 0;
    p.stats = static_cast<float*>(stats);
    p.act = act; p.Jt = q.Jt; p.rp = q.rp; p.wpitch = q.pitch * 2; p.halo_b = q.halo_b; p.wst_b = q.wst_b;
    p.ks = q.ks; p.hs = q.hs;
    p.TX = (d->W + TS - 1) / TS; p.TY = (d->H + TS - 1) / TS; p.tiles = d->B * p.TX * p.TY;
    static int ready = -1;
    if (ready != 0) {
        hipError_t e = hipSuccess;
        const void* fs[4] = {reinterpret_cast<const void*>(conv_stem16_kernel<false, 4>),
                             reinterpret_cast<const void*>(conv_stem16_kernel<true, 4>),
                             reinterpret_cast<const void*>(conv_stem16_kernel<false, 2>),
                             reinterpret_cast<const void*>(conv_stem16_kernel<true, 2>)};
        for (const void* f : fs)
            if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        ready = e == hipSuccess ? 0 : mmh::fail("conv_stem16_kernel: %s", hipGetErrorString(e));
    }
    if (ready != 0) return ready;
    const dim3 grid(8 * ((p.tiles + 7) / 8));
    hipStream_t st = mmh::as_stream(s);
    const bool h16 = d->dtype == MMH_FP16;
    if (q.rw == 2) {                // one work-group per CU anyway: eight waves
        if (h16) hipLaunchKernelGGL((conv_stem16_kernel<true, 2>), grid, dim3(512), q.lds, st, p);
        else hipLaunchKernelGGL((conv_stem16_kernel<false, 2>), grid, dim3(512), q.lds, st, p);
    } else {
        if (h16) hipLaunchKernelGGL((conv_stem16_kernel<true, 4>), grid, dim3(256), q.lds, st, p);
        else hipLaunchKernelGGL((conv_stem16_kernel<false, 4>), grid, dim3(256), q.lds, st, p);
    }
    return mmh::check_launch("conv_stem16_kernel");
}

// ---- mmh_conv7_head_dgrad_lp16: see the kernels above.  d = the head conv (Cin = 64, Cout = 4 incl. padding, 7x7, reflect) ----
static bool head_dgrad_desc(const mmh_conv_desc* d, mmh_conv_desc& e) {
    if (!d || d->kh != 7 || d->kw != 7 || d->stride != 1 || d->pad != 3 || d->pad_mode != MMH_PAD_REFLECT) return false;
    if (d->Cin != 64 || d->Cout != 4 || d->Ho != d->H || d->Wo != d->W || d->H < 8 || d->W < 8) return false;
    if (d->dtype != MMH_BF16 && d->dtype != MMH_FP16) return false;
    if (d->y_cs < 4 || d->y_cs % 4 || d->x_cs < 64 || d->x_cs % 8) return false;
    e = *d;
    e.H = e.Ho = d->H + 6; e.W = e.Wo = d->W + 6; e.Cin = 4; e.Cout = 64; e.pad_mode = MMH_PAD_ZERO; e.x_cs = 8; e.y_cs = 64;
    return mmh_conv_stem16_supported(&e, 8) != 0;
}

int mmh_conv7_head_dgrad_lp16_supported(const mmh_conv_desc* d) {
    mmh_conv_desc e;
    return head_dgrad_desc(d, e) ? 1 : 0;
}

size_t mmh_conv7_head_dgrad_lp16_ws_bytes(const mmh_conv_desc* d) {
    mmh_conv_desc e;
    if (!head_dgrad_desc(d, e)) return 0;
    const size_t px = (size_t)e.B * e.H * e.W;
    return ((mmh_conv_stem16_weights_bytes(8) + 255) & ~(size_t)255) + px * 8 * 2 + px * 64 * 2;
}

int mmh_conv7_head_dgrad_lp16(const mmh_conv_desc* d, const void* dy, const void* w, void* dx, int dx_is16, void* ws,
                              size_t ws_bytes, const void* zeros, mmh_stream_t s) {
    mmh_conv_desc e;
    MMH_REQUIRE(head_dgrad_desc(d, e), "mmh_conv7_head_dgrad_lp16: the head conv only (7x7 / reflect pad 3 / 64 -> 4, 16-bit dtype)");
    MMH_REQUIRE(dy && w && dx && ws && zeros && ws_bytes >= mmh_conv7_head_dgrad_lp16_ws_bytes(d) &&
                    (reinterpret_cast<uintptr_t>(ws) & 255) == 0 && (reinterpret_cast<uintptr_t>(dx) & 15) == 0,
                "mmh_conv7_head_dgrad_lp16: bad buffers (ws 256-byte aligned, mmh_conv7_head_dgrad_lp16_ws_bytes)");
    hipStream_t st = mmh::as_stream(s);
    const int h16 = d->dtype == MMH_FP16 ? 1 : 0;
    char* wst = static_cast<char*>(ws);
    char* E = wst + ((mmh_conv_stem16_weights_bytes(8) + 255) & ~(size_t)255);
    const size_t px = (size_t)e.B * e.H * e.W;
    char* dxp = E + px * 16;
    const int pitch = 32 * ((7 * 8 + 31) / 32) + 8;
    hipLaunchKernelGGL(head_dgrad_w_kernel, dim3((7 * 64 * pitch + 255) / 256), dim3(256), 0, st, static_cast<const float*>(w),
                       pitch, h16, reinterpret_cast<unsigned short*>(wst));
    hipLaunchKernelGGL(head_dgrad_embed_kernel, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, st,
                       static_cast<const float*>(dy), d->B, d->H, d->W, d->y_cs, h16, reinterpret_cast<uint4*>(E));
    if (int rc = mmh::check_launch("head_dgrad_embed_kernel")) return rc;
    if (int rc = conv_stem16_impl(&e, E, 8, wst, nullptr, dxp, 1, MMH_ACT_NONE, zeros, nullptr, s)) return rc;
    const size_t n = (size_t)d->B * d->H * d->W * 8;
    if (h16) hipLaunchKernelGGL(head_dgrad_fold_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                                reinterpret_cast<const uint4*>(dxp), d->B, d->H, d->W, d->x_cs,
                                dx_is16 ? nullptr : static_cast<float*>(dx), dx_is16 ? static_cast<char*>(dx) : nullptr);
    else hipLaunchKernelGGL(head_dgrad_fold_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                            reinterpret_cast<const uint4*>(dxp), d->B, d->H, d->W, d->x_cs,
                            dx_is16 ? nullptr : static_cast<float*>(dx), dx_is16 ? static_cast<char*>(dx) : nullptr);
    return mmh::check_launch("head_dgrad_fold_kernel");
}
