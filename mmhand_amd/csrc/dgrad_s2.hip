// fp32 dgrad of the 3x3 / stride-2 / zero-pad-1 downsampling conv 64 -> 128 (models/Generator.py:192-199: the three
// streams' first nn.Conv2d(ngf, 2 ngf, 3, 2, 1); the same arithmetic is the fprop of the decoder's
// nn.ConvTranspose2d(2 ngf, ngf, 3, 2, 1, output_padding=1), models/Generator.py:212-219) with the dy halo resident in LDS.
//
// The generic route (conv_igemm_multi_kernel) runs the four output parity classes as four separate implicit GEMMs with
// 1, 2, 2 and 4 taps: contractions of 128 ... 512 over an N of 64, every class re-gathering dy and writing every other
// pixel of dx - 76 TFLOP/s = 0.49 of the fp32 MFMA peak at 64 -> 128 @256x256.  Here one work-group owns an 8 x 16 block
// of dy positions: dx[2 ph + a][2 pw + b] (a, b in {0, 1}) needs dy[ph .. ph + 1][pw .. pw + 1] only,
//
//   class (0,0): tap (1,1) dy[ph][pw]                   class (0,1): (1,0) dy[ph][pw+1], (1,2) dy[ph][pw]
//   class (1,0): (0,1) dy[ph+1][pw], (2,1) dy[ph][pw]   class (1,1): (0,0) dy[ph+1][pw+1], (0,2) dy[ph+1][pw],
//                                                                     (2,0) dy[ph][pw+1],   (2,2) dy[ph][pw]
//
// so the 9 x 17 halo of dy (78 KiB, LDS-DMA, zero-filled past the image) is staged ONCE for all nine taps and all four
// classes, the filter is streamed one tap at a time ([64 ci][128 co] = 32 KiB, two stages) and the 16 x 32 x 64 block of
// dx leaves in whole rows.  512 threads = 8 waves: wave = (2 dy rows x 16 positions) x 32 input channels x 4 classes =
// four 32x32 accumulator tiles; per tap and 8-deep k-chunk one ds_read_b128 of dy and one of the filter feed four
// v_mfma_f32_32x32x2_f32.  LDS images are XOR-swizzled per 16-byte chunk (key = halo column & 15 for dy, ci & 15 for
// the filter: distinct over each of ds_read_b128's 16-lane groups), applied on the global side of the DMA.  The kernel is
// persistent (one work-group per CU, XCD-contiguous tile lists): the next tile's halo and first tap are requested before
// the current tile's stores.
#include <algorithm>
#include "common.h"

namespace mmh { int g_dgrad_s2_halo = 1; }

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const f32x4 __attribute__((address_space(3))) * lds_f4_p;
__device__ __forceinline__ f32x4 lds_f4(unsigned addr) { return *reinterpret_cast<lds_f4_p>((size_t)addr); }

constexpr int CI = 64, CO = 128;
constexpr int TH = 8, TW = 16;                  // dy positions per tile
constexpr int HWD = TW + 1;                     // halo 9 x 17
constexpr int HPIX_REAL = (TH + 1) * HWD;       // 153
constexpr int PIXB = CO * 4;                    // 512 bytes per dy pixel / per filter row
constexpr int NT = 512;
constexpr int PPR = NT * 16 / PIXB;             // pixels (rows) per DMA round = 16
constexpr int HROUNDS = (HPIX_REAL + PPR - 1) / PPR;     // 10
constexpr int HALO_B = HROUNDS * PPR * PIXB;    // 81920
constexpr int WST_B = CI * PIXB;                // 32768
constexpr int WROUNDS = CI / PPR;               // 4
constexpr int LDS_B = HALO_B + 2 * WST_B;       // 147456

__device__ char g_zero_line[128];               // DMA source of the zero padding

struct DgradS2KP {
    const float* dy;        // [B][Ho][Wo][128]
    const float* w;         // [3][3][64][128]
    const float* bias;      // [64] or null (ConvTranspose fprop)
    float* dx;              // [B][2 Ho][2 Wo][dx_cs]
    int B, Ho, Wo, dx_cs, act;
    int TX, TY, tiles, per_xcd, slots;
};

__device__ __forceinline__ float act_of(float v, int act) {
    if (act == MMH_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == MMH_ACT_TANH) return tanhf(v);
    return v;
}

__global__ void __launch_bounds__(NT, 1) dgrad_s2_kernel(const DgradS2KP p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kg = lane >> 5;
    const int pg = wave & 3, ch = wave >> 2;
    const unsigned lds0 = mmh::lds_addr_of(smem);
    const unsigned wdst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int t_end = min(p.tiles, (xcd + 1) * p.per_xcd);
    const int drow = tid >> 5, dchunk = tid & 31;           // DMA role: row within a round, 16-byte chunk of the row

    auto issue_w = [&](int tap) {
        const float* src = p.w + (size_t)tap * (CI * CO);
        const unsigned dst = wdst + (unsigned)HALO_B + (unsigned)((tap & 1) * WST_B);
#pragma unroll
        for (int rr = 0; rr < WROUNDS; ++rr) {
            const int row = rr * PPR + drow;
            mmh::lds_dma16(src + row * CO + ((dchunk ^ (row & 15)) << 2), dst + (unsigned)(rr * NT * 16));
        }
    };
    auto issue_halo = [&](int tile) {
        const int b = tile / (p.TX * p.TY);
        const int trem = tile - b * (p.TX * p.TY);
        const int ty = trem / p.TX, tx = trem - ty * p.TX;
        const int ph0 = ty * TH, pw0 = tx * TW;
#pragma unroll
        for (int rr = 0; rr < HROUNDS; ++rr) {
            const int hp = rr * PPR + drow;
            const int hy = hp / HWD, hx = hp - hy * HWD;
            const int ph = ph0 + hy, pw = pw0 + hx;
            const bool ok = hp < HPIX_REAL && ph < p.Ho && pw < p.Wo;
            const void* g = ok ? (const void*)(p.dy + ((size_t)(b * p.Ho + ph) * p.Wo + pw) * CO + ((dchunk ^ (hx & 15)) << 2))
                               : (const void*)(g_zero_line + (lane & 7) * 16);
            mmh::lds_dma16(g, wdst + (unsigned)(rr * NT * 16));
        }
    };

    // lane constants of the fragment reads.  dy fragment of tap shift (dh, dw): MFMA row r = position (2 pg + r / 16 + dh,
    // r % 16 + dw), chunk (2 kc + kg) ^ key; the address is base | (kx << 4), XOR-ed with kc << 5 per k-chunk
    unsigned a_lane[2][2];
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int dw = 0; dw < 2; ++dw) {
            const int hx = (r & 15) + dw;
            a_lane[dh][dw] = lds0 + (unsigned)(((2 * pg + (r >> 4) + dh) * HWD + hx) * PIXB) + (unsigned)((kg ^ (hx & 15)) << 4);
        }
    const unsigned b_lane = lds0 + (unsigned)HALO_B + (unsigned)((ch * 32 + r) * PIXB) + (unsigned)((kg ^ (r & 15)) << 4);

    int tile = xcd * p.per_xcd + slot;
    if (tile >= t_end) return;
    issue_halo(tile);
    issue_w(0);
    for (; tile < t_end; tile += p.slots) {
        f32x16 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;

#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap - 3 * (tap / 3);
            const int dh = kh == 0 ? 1 : 0, dw = kw == 0 ? 1 : 0;
            const int cls = (kh != 1 ? 2 : 0) + (kw != 1 ? 1 : 0);
            __builtin_amdgcn_s_waitcnt(0x0070);             // this thread's DMA (halo, tap) has landed; stores too
            __syncthreads();                                // ... everybody's; stage (tap + 1) & 1 is no longer read
            if (tap + 1 < 9) issue_w(tap + 1);
            unsigned ab = a_lane[dh][dw];
            unsigned bb = b_lane + (unsigned)((tap & 1) * WST_B);
            asm volatile("" : "+v"(ab), "+v"(bb));          // keeps the 15 XOR-ed addresses per operand out of the tile loop's preheader
            f32x4 af = lds_f4(ab), bf = lds_f4(bb);
#pragma unroll
            for (int kc = 0; kc < 16; ++kc) {
                f32x4 an, bn;
                if (kc + 1 < 16) {
                    an = lds_f4(ab ^ (unsigned)((kc + 1) << 5));
                    bn = lds_f4(bb ^ (unsigned)((kc + 1) << 5));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[cls] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], bf[j], acc[cls], 0, 0, 0);
                if (kc + 1 < 16) {
                    af = an;
                    bf = bn;
                }
            }
        }
        // every wave is done with the halo and with stage 0 (tap 8): request the next tile under this tile's stores
        const int b = tile / (p.TX * p.TY);
        const int trem = tile - b * (p.TX * p.TY);
        const int ty = trem / p.TX, tx = trem - ty * p.TX;
        const int next = tile + p.slots;
        __syncthreads();
        if (next < t_end) {
            issue_halo(next);
            issue_w(0);
        }
        const int n = ch * 32 + r;
        const float bv = p.bias ? p.bias[n] : 0.f;
        const int H = 2 * p.Ho, W = 2 * p.Wo;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = (i & 3) + 8 * (i >> 2) + 4 * kg;
            const int ph = ty * TH + 2 * pg + (m >> 4), pw = tx * TW + (m & 15);
            if (ph < p.Ho && pw < p.Wo) {
                float* o = p.dx + ((size_t)(b * H + 2 * ph) * W + 2 * pw) * (size_t)p.dx_cs + n;
                const size_t rs = (size_t)W * p.dx_cs;
                o[0] = act_of(acc[0][i] + bv, p.act);
                o[p.dx_cs] = act_of(acc[1][i] + bv, p.act);
                o[rs] = act_of(acc[2][i] + bv, p.act);
                o[rs + p.dx_cs] = act_of(acc[3][i] + bv, p.act);
            }
        }
    }
}

}  // namespace

namespace mmh {

bool dgrad_s2_halo_ok(const mmh_conv_desc* d, int dx_cs) {
    return g_dgrad_s2_halo && d->dtype == MMH_F32 && d->kh == 3 && d->kw == 3 && d->stride == 2 && d->pad == 1 &&
           d->pad_mode == MMH_PAD_ZERO && d->Cin == CI && d->Cout == CO && d->y_cs == CO && d->H == 2 * d->Ho &&
           d->W == 2 * d->Wo && dx_cs >= CI && (size_t)d->B * d->H * d->W * dx_cs < (1ull << 31);
}

int launch_dgrad_s2_halo(const mmh_conv_desc* d, const void* dy, const void* w, const void* bias, void* dx, int dx_cs,
                         int act, hipStream_t st) {
    DgradS2KP p{};
    p.dy = static_cast<const float*>(dy);
    p.w = static_cast<const float*>(w);
    p.bias = static_cast<const float*>(bias);
    p.dx = static_cast<float*>(dx);
    p.B = d->B; p.Ho = d->Ho; p.Wo = d->Wo; p.dx_cs = dx_cs; p.act = act;
    p.TY = (d->Ho + TH - 1) / TH;
    p.TX = (d->Wo + TW - 1) / TW;
    p.tiles = p.B * p.TX * p.TY;
    p.per_xcd = (p.tiles + 7) / 8;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return fail("dgrad_s2: cannot query the device");
        cus = prop.multiProcessorCount;
    }
    p.slots = std::max(1, std::min(cus / 8, p.per_xcd));
    static int ready = -1;
    if (ready != 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dgrad_s2_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
        if (e != hipSuccess) return fail("dgrad_s2: %s", hipGetErrorString(e));
        ready = 0;
    }
    hipLaunchKernelGGL(dgrad_s2_kernel, dim3(8 * p.slots), dim3(NT), LDS_B, st, p);
    return check_launch("dgrad_s2_kernel");
}

}  // namespace mmh

extern "C" int mmh_dgrad_s2_halo_supported(const mmh_conv_desc* d, int dx_cs) {
    return d && mmh::dgrad_s2_halo_ok(d, dx_cs) ? 1 : 0;
}
