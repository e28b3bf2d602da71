// 7x7 / stride 1 convolutions with FOUR output channels from a 64-channel 16-bit input, on the MFMA.
//
//   mode 0  the Generator head  ReflectionPad2d(3) + Conv2d(64, 3 -> 4, 7) + Tanh   (models/Generator.py:254-259)
//   mode 1  the gradient of a Discriminator stem  ReflectionPad2d(3) + Conv2d(Cin, 64, 7)  towards the first four
//           input channels - the generated image inside cat(img, pose) / cat(img, img)
//           (models/Discriminator.py:60-64 seen from models/MMHandModel.py:238-243); with KS = 3 the same for VGG19's
//           conv1_1, Conv2d(3, 64, 3, padding=1): the perceptual loss's gradient towards the generated image
//           (losses/L1_plus_perceptualLoss.py:22-27,60-67)
//
// conv_thin.hip computes these on the vector ALU (a 32-wide MFMA tile wastes 7/8 of the matrix core on 4 columns):
// ~1.0 ms each at 256x256, B=32 - 50 TFLOP/s on a machine whose 16-bit MFMA sustains 2000.  Here the 16-column MFMA
// (v_mfma_f32_16x16x32) runs with 4 of its 16 weight rows meaningful: a quarter of the matrix core is still 10x the
// vector ALU.  The weight fragment is the MFMA's first operand, so D[row = channel][column = pixel]: lanes 0-15 hold
// the four channels of one pixel each (one 16-byte store); weight-fragment lanes 4-15 simply re-read rows 0-3 - their
// rows of D are copies nobody stores, which costs nothing and needs no zero padding in LDS.
//
// Work-group = 256 threads = 4 waves, output tile 8 rows x 16 pixels; its (8+6) x (16+6) x 64-channel halo (38.5 KiB,
// one 128-byte LDS row per pixel, 16-byte chunks XOR-ed with `halo column & 6`: conflict-free for all seven tap columns
// under ds_read_b128's lane groups, checked by enumeration) and the whole [49][4][64] filter (24.5 KiB) come in by
// LDS-DMA once; then 49 taps x 2 k-halves x 2 rows = 196 MFMAs per wave whose fragment addresses are 14 + 2 lane
// constants plus immediates.  68 KiB of LDS: two work-groups per CU, one loading while the other multiplies.
// mode 1 runs the same kernel on dy with the flipped filter over the PADDED domain (zero padding 6) and folds the
// pad ring back (transpose of ReflectionPad2d(3)) with fold7_kernel.
#include <algorithm>
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const bf16x8 __attribute__((address_space(3))) * lds_frag_p;
__device__ __forceinline__ bf16x8 lds_frag(unsigned addr) { return *reinterpret_cast<lds_frag_p>(addr); }

constexpr int TW = 16, TH = 8;                  // output tile
template <int KS> struct Geo {                  // KS x KS taps (7: head / stems; 3: VGG conv1_1's image gradient)
    static constexpr int HPW = TW + KS - 1, HPH = TH + KS - 1;     // halo 22 x 14 | 18 x 10
    static constexpr int HROWS = HPW * HPH;                        // LDS rows of 128 bytes
    static constexpr int HRD = (HROWS + 31) / 32;                  // DMA rounds of 32 rows (256 threads x 16 bytes)
    static constexpr int WROWS = KS * KS * 4;                      // filter rows [tap][channel]
    static constexpr int WRD = (WROWS + 31) / 32;
    static constexpr int HALO_B = HRD * 32 * 128;                  // the rounds' footprint: 40960 | 24576
    static constexpr int W_B = WRD * 32 * 128;                     // 28672 | 8192
    static constexpr int LDS_B = HALO_B + W_B;                     // 69632 | 32768
};

struct C7KP {
    const char* x;          // 16-bit [B][SH][SW][cs], channels 0..63
    const char* w;          // 16-bit [49][4][64]
    const char* zeros;      // >= 128 zero bytes
    const float* bias;      // [4] or nullptr
    float* y;               // fp32 [B][OH][OW][ycs], channels 0..3 written
    int B, SH, SW, cs, OH, OW, ycs;
    int pad, reflect, act;
    int TX, TY, tiles;
};

template <bool H16>
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    if (H16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0,
                                                      0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <bool H16, int KS>
__global__ void __launch_bounds__(256, 2) conv7_n4_kernel(const C7KP p) {
    typedef Geo<KS> G_;
    constexpr int HPW = G_::HPW, HROWS = G_::HROWS, HRD = G_::HRD, WROWS = G_::WROWS, WRD = G_::WRD, HALO_B = G_::HALO_B;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    // one XCD walks a contiguous range of tiles (neighbouring tiles share halo rows in its L2)
    const int per_xcd = (p.tiles + 7) / 8;
    const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= p.tiles) return;
    const int b = tile / (p.TX * p.TY);
    const int trem = tile - b * (p.TX * p.TY);
    const int ty = trem / p.TX, tx = trem - ty * p.TX;
    const int oh0 = ty * TH, ow0 = tx * TW;

    // filter: rows [tap][channel], chunk key 2 * channel (the four rows a fragment touches land on distinct banks)
#pragma unroll
    for (int rd = 0; rd < WRD; ++rd) {
        const int r = rd * 32 + wave * 8 + (lane >> 3);
        if (r < WROWS) {
            const unsigned q = (unsigned)((lane & 7) ^ (2 * (r & 3)));
            __builtin_amdgcn_global_load_lds(p.w + (size_t)r * 128 + q * 16, (lds_vp)(smem + HALO_B + (rd * 32 + wave * 8) * 128),
                                             16, 0, 0);
        }
    }
    // halo: LDS row = hy * 22 + hx, chunk key hx & 6; pixels outside the source (or its mirror image) read zeros
#pragma unroll
    for (int rd = 0; rd < HRD; ++rd) {
        const int r = rd * 32 + wave * 8 + (lane >> 3);
        if (r < HROWS) {
            const int hy = r / HPW, hx = r - hy * HPW;
            int ih = oh0 + hy - p.pad, iw = ow0 + hx - p.pad;
            if (p.reflect) {
                ih = ih < 0 ? -ih : ih;
                iw = iw < 0 ? -iw : iw;
                ih = ih >= p.SH ? 2 * (p.SH - 1) - ih : ih;
                iw = iw >= p.SW ? 2 * (p.SW - 1) - iw : iw;
            }
            const bool ok = ih >= 0 && ih < p.SH && iw >= 0 && iw < p.SW;
            const unsigned q = (unsigned)((lane & 7) ^ (hx & 6));
            const char* g = ok ? p.x + ((size_t)(b * p.SH + ih) * p.SW + iw) * p.cs * 2 + q * 16 : p.zeros + (lane & 7) * 16;
            __builtin_amdgcn_global_load_lds(g, (lds_vp)(smem + (rd * 32 + wave * 8) * 128), 16, 0, 0);
        }
    }

    const unsigned lds0 = mmh::lds_addr_of(smem);
    // lane constants: the pixel fragment of tap column kw, k-half h, at this wave's first output row; the filter fragment
    unsigned aB[KS][2], wB[2];
#pragma unroll
    for (int kw = 0; kw < KS; ++kw)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned hx = (unsigned)(kw + l15);
            aB[kw][h] = lds0 + (unsigned)(2 * wave * HPW) * 128u + hx * 128u + ((((unsigned)(4 * h + g4)) ^ (hx & 6u)) << 4);
        }
#pragma unroll
    for (int h = 0; h < 2; ++h)
        wB[h] = lds0 + HALO_B + (unsigned)(l15 & 3) * 128u + ((((unsigned)(4 * h + g4)) ^ (unsigned)(2 * (l15 & 3))) << 4);

    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    __builtin_amdgcn_s_waitcnt(0x0070);     // vmcnt(0): this wave's DMA has landed
    __syncthreads();
#pragma unroll
    for (int kh = 0; kh < KS; ++kh)
#pragma unroll
        for (int kw = 0; kw < KS; ++kw)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bf16x8 wf = lds_frag(wB[h] + (unsigned)((kh * KS + kw) * 512));
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bf16x8 af = lds_frag(aB[kw][h] + (unsigned)((i + kh) * HPW * 128));
                    acc[i] = mfma16<H16>(wf, af, acc[i]);
                }
            }

    if (g4 == 0) {
        float bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = p.bias ? p.bias[r] : 0.f;
        const int ow = ow0 + l15;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int oh = oh0 + 2 * wave + i;
            if (oh < p.OH && ow < p.OW) {
                f32x4 v = acc[i];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float t = v[r] + bv[r];
                    v[r] = p.act == MMH_ACT_RELU ? (t > 0.f ? t : 0.f) : (p.act == MMH_ACT_TANH ? tanhf(t) : t);
                }
                *reinterpret_cast<f32x4*>(p.y + ((size_t)(b * p.OH + oh) * p.OW + ow) * p.ycs) = v;
            }
        }
    }
}

// the 16-bit filter [49][4][64]: mode 0 from the head's w [7][7][64][Cout] (channels n < min(4, Cout)), mode 1 the
// flipped filter of the stem's w [7][7][Cin][64] restricted to input channels n < 4
__global__ void prep_w7n4_kernel(const float* __restrict__ w, int taps, int Cin, int Cout, int mode, int h16,
                                 unsigned short* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= taps * 4 * 64) return;
    const int c = i & 63, n = (i >> 6) & 3, t = i >> 8;
    float v;
    if (mode == 0) v = n < Cout ? w[((size_t)t * Cin + c) * Cout + n] : 0.f;
    else v = n < Cin ? w[((size_t)(taps - 1 - t) * Cin + n) * Cout + c] : 0.f;
    if (h16) out[i] = __builtin_bit_cast(unsigned short, (_Float16)v);
    else out[i] = __builtin_bit_cast(unsigned short, (__bf16)v);
}

// transpose of ReflectionPad2d(3) for four channels: dx[b,h,w,0..3] (pixel stride cs) = sum of dxp [B][H+6][W+6][4] over
// the padded positions that mirror onto (h, w)
__global__ void fold7_kernel(const float* __restrict__ dxp, float* __restrict__ dx, int B, int H, int W, int cs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * H * W) return;
    const int w = (int)(i % W);
    const int64_t t = i / W;
    const int h = (int)(t % H), b = (int)(t / H);
    int ph[3], pw[3], nh = 0, nw = 0;
    ph[nh++] = h + 3;
    if (h >= 1 && h <= 3) ph[nh++] = 3 - h;
    if (h >= H - 4 && h <= H - 2) ph[nh++] = 2 * (H - 1) - h + 3;
    pw[nw++] = w + 3;
    if (w >= 1 && w <= 3) pw[nw++] = 3 - w;
    if (w >= W - 4 && w <= W - 2) pw[nw++] = 2 * (W - 1) - w + 3;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < nh; ++a)
        for (int c = 0; c < nw; ++c)
            s += *reinterpret_cast<const f32x4*>(dxp + ((size_t)(b * (H + 6) + ph[a]) * (W + 6) + pw[c]) * 4);
    *reinterpret_cast<f32x4*>(dx + (size_t)i * cs) = s;
}


bool supported(const mmh_conv_desc* d, int mode) {
    if (!d || d->kh != d->kw || (d->kh != 7 && d->kh != 3) || d->stride != 1 || d->pad != d->kh / 2 || d->Ho != d->H ||
        d->Wo != d->W)
        return false;
    if (d->kh == 3 && (mode != 1 || d->pad_mode != MMH_PAD_ZERO)) return false;     // 3x3: VGG conv1_1's image gradient only
    if (d->dtype != MMH_BF16 && d->dtype != MMH_FP16) return false;
    if (d->H < 8 || d->W < 8) return false;             // single mirror per side; the fold's index lists
    if (mode == 0) return d->Cin == 64 && d->Cout >= 1 && d->Cout <= 4 && d->x_cs >= 64 && d->x_cs % 8 == 0 && d->y_cs % 4 == 0;
    if (mode == 1) return d->Cout == 64 && d->Cin >= 1 && d->y_cs >= 64 && d->y_cs % 8 == 0 && d->x_cs % 4 == 0 && d->x_cs >= 4;
    return false;
}

}  // namespace

int mmh_conv7_n4_lp16_supported(const mmh_conv_desc* d, int mode) { return supported(d, mode) ? 1 : 0; }

size_t mmh_conv7_n4_lp16_ws_bytes(const mmh_conv_desc* d, int mode) {
    if (!supported(d, mode)) return 0;
    size_t n = 25600;                                   // the 16-bit filter, padded to 512 bytes
    if (mode == 1 && d->pad_mode == MMH_PAD_REFLECT) n += (size_t)d->B * (d->H + 6) * (d->W + 6) * 4 * sizeof(float);
    return n;
}

int mmh_conv7_n4_lp16(const mmh_conv_desc* d, int mode, const void* x16, const void* w, const void* bias, void* y,
                      int act, void* ws, size_t ws_bytes, const void* zeros, mmh_stream_t s) {
    MMH_REQUIRE(supported(d, mode) && x16 && w && y && ws && zeros,
                "mmh_conv7_n4_lp16: 7x7 / stride 1 / pad 3 (mode 1 also 3x3 / pad 1, zero padding), 16-bit dtype; mode 0: "
                "Cin == 64, Cout <= 4; mode 1: Cout == 64");
    MMH_REQUIRE(ws_bytes >= mmh_conv7_n4_lp16_ws_bytes(d, mode), "mmh_conv7_n4_lp16: workspace too small");
    MMH_REQUIRE((reinterpret_cast<uintptr_t>(y) & 15) == 0 && (reinterpret_cast<uintptr_t>(ws) & 15) == 0,
                "mmh_conv7_n4_lp16: y and ws must be 16-byte aligned");
    hipStream_t st = mmh::as_stream(s);
    const bool h16 = d->dtype == MMH_FP16;
    const bool refl = d->pad_mode == MMH_PAD_REFLECT;
    char* w16 = static_cast<char*>(ws);
    float* dxp = reinterpret_cast<float*>(w16 + 25600);
    const int taps = d->kh * d->kw;
    hipLaunchKernelGGL(prep_w7n4_kernel, dim3((taps * 4 * 64 + 255) / 256), dim3(256), 0, st, static_cast<const float*>(w),
                       taps, d->Cin, d->Cout, mode, h16 ? 1 : 0, reinterpret_cast<unsigned short*>(w16));
    C7KP p{};
    p.x = static_cast<const char*>(x16);
    p.w = w16;
    p.zeros = static_cast<const char*>(zeros);
    p.B = d->B; p.SH = d->H; p.SW = d->W;
    if (mode == 0) {
        p.bias = static_cast<const float*>(bias);
        p.y = static_cast<float*>(y);
        p.cs = d->x_cs; p.OH = d->H; p.OW = d->W; p.ycs = d->y_cs; p.pad = d->pad; p.reflect = refl ? 1 : 0; p.act = act;
    } else if (refl) {          // padded domain: dxp[pp] = sum_k wflip[k] dy[pp + k - 6], zero outside
        p.y = dxp;
        p.cs = d->y_cs; p.OH = d->H + 6; p.OW = d->W + 6; p.ycs = 4; p.pad = 6; p.reflect = 0; p.act = MMH_ACT_NONE;
    } else {
        p.y = static_cast<float*>(y);
        p.cs = d->y_cs; p.OH = d->H; p.OW = d->W; p.ycs = d->x_cs; p.pad = d->pad; p.reflect = 0; p.act = MMH_ACT_NONE;
    }
    MMH_REQUIRE((long long)p.B * p.SH * p.SW * p.cs < (1ll << 31) && (long long)p.B * p.OH * p.OW * p.ycs < (1ll << 31),
                "mmh_conv7_n4_lp16: tensor too large");
    p.TX = (p.OW + TW - 1) / TW; p.TY = (p.OH + TH - 1) / TH; p.tiles = p.B * p.TX * p.TY;
    static int ready = -1;
    if (ready != 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv7_n4_kernel<false, 7>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Geo<7>::LDS_B);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv7_n4_kernel<true, 7>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, Geo<7>::LDS_B);
        ready = e == hipSuccess ? 0 : mmh::fail("conv7_n4_kernel: %s", hipGetErrorString(e));
    }
    if (ready != 0) return ready;
    const dim3 grid(8 * ((p.tiles + 7) / 8));
    if (d->kh == 3) {
        if (h16) hipLaunchKernelGGL((conv7_n4_kernel<true, 3>), grid, dim3(256), Geo<3>::LDS_B, st, p);
        else hipLaunchKernelGGL((conv7_n4_kernel<false, 3>), grid, dim3(256), Geo<3>::LDS_B, st, p);
    } else {
        if (h16) hipLaunchKernelGGL((conv7_n4_kernel<true, 7>), grid, dim3(256), Geo<7>::LDS_B, st, p);
        else hipLaunchKernelGGL((conv7_n4_kernel<false, 7>), grid, dim3(256), Geo<7>::LDS_B, st, p);
    }
    if (mode == 1 && refl) {
        const int64_t n = (int64_t)d->B * d->H * d->W;
        hipLaunchKernelGGL(fold7_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dxp, static_cast<float*>(y),
                           d->B, d->H, d->W, d->x_cs);
    }
    return mmh::check_launch("conv7_n4_kernel");
}
