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
#include <type_traits>
#include "common.h"

namespace mmh { int g_dgrad_s2_halo = 1; int g_dgrad_s2_dbg = 0; }

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const f32x4 __attribute__((address_space(3))) * lds_f4_p;
__device__ __forceinline__ f32x4 lds_f4(unsigned addr) { return *reinterpret_cast<lds_f4_p>((size_t)addr); }

constexpr int CI = 64, CO = 128;
constexpr int KHALF = CO / 2;                   // the contraction runs in two halves of 64 output channels
constexpr int TH = 8, TW = 16;                  // dy positions per tile
constexpr int HWD = TW + 1;                     // halo 9 x 17
constexpr int HPIX_REAL = (TH + 1) * HWD;       // 153
constexpr int PIXB = KHALF * 4;                 // 256 bytes per dy pixel half / per filter row half
constexpr int NT = 512;
constexpr int PPR = NT * 16 / PIXB;             // pixels (rows) per DMA round = 32
constexpr int HROUNDS = (HPIX_REAL + PPR - 1) / PPR;     // 5
constexpr int HALO_B = HROUNDS * PPR * PIXB;    // 40960 per half
constexpr int WST_B = CI * PIXB;                // 16384: one tap, one half
constexpr int WROUNDS = CI / PPR;               // 2
constexpr int NWST = 2;
constexpr int LDS_B = 2 * HALO_B + NWST * WST_B;    // 114688

__device__ char g_zero_line[128];               // DMA source of the zero padding
// s_waitcnt vmcnt(n) lgkmcnt(0) (gfx9 encoding: vmcnt in bits 3:0 and 15:14, expcnt 6:4 left at 7)
constexpr int wait_vm(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0070; }
template <int V> using IC = std::integral_constant<int, V>;
// taps in the order 4 | 3 5 | 1 7 | 0 2 6 8: the output classes complete after 1, 3, 5 and 9 taps
constexpr int tap_of(int i) { return i == 0 ? 4 : i == 1 ? 3 : i == 2 ? 5 : i == 3 ? 1 : i == 4 ? 7 : i == 5 ? 0 : i == 6 ? 2 : i == 7 ? 6 : 8; }

struct DgradS2KP {
    const float* dy;        // [B][Ho][Wo][128]
    const float* w;         // [3][3][64][128]
    const float* bias;      // [64] or null (ConvTranspose fprop)
    float* dx;              // [B][2 Ho][2 Wo][dx_cs]
    int B, Ho, Wo, dx_cs, act;
    int TX, TY, tiles, per_xcd, slots, dbg;
};

__device__ __forceinline__ float act_of(float v, int act) {
    if (act == MMH_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == MMH_ACT_TANH) return tanhf(v);
    return v;
}

// lane ^ 1 / lane ^ 2 exchange inside a quad (DPP quad_perm [1,0,3,2] / [2,3,0,1])
template <int CTRL>
__device__ __forceinline__ float quad_xchg(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// 4 x 4 transpose over the four lanes of a quad: in: lane j holds column j of rows 0..3 (v[0..3]); out: lane j holds
// columns 0..3 of row j
__device__ __forceinline__ f32x4 quad_transpose(float v0, float v1, float v2, float v3, bool odd, bool hi) {
    const float r01 = quad_xchg<0xB1>(odd ? v0 : v1), r23 = quad_xchg<0xB1>(odd ? v2 : v3);
    const float p0x = odd ? r01 : v0, p0y = odd ? v1 : r01;        // rows 0 / 1, two columns
    const float p1x = odd ? r23 : v2, p1y = odd ? v3 : r23;        // rows 2 / 3
    const float sx = quad_xchg<0x4E>(hi ? p0x : p1x), sy = quad_xchg<0x4E>(hi ? p0y : p1y);
    f32x4 o;
    o[0] = hi ? sx : p0x;
    o[1] = hi ? sy : p0y;
    o[2] = hi ? p1x : sx;
    o[3] = hi ? p1y : sy;
    return o;
}

// EPI: bias and / or activation in the epilogue (the ConvTranspose2d forward); false: plain dgrad
template <bool EPI>
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
    const int drow = tid >> 4, dchunk = tid & 15;           // DMA role: row within a round, 16-byte chunk of the row
    const int tpi = p.TX * p.TY;

    // filter of phase q = (half, tap) -> stage q & 1: rows ci, 64 output channels of the half
    auto issue_w = [&](int q) {
        const int half = q >= 9 ? 1 : 0, tap = tap_of(q - 9 * half);
        const float* src = p.w + (size_t)tap * (CI * CO) + half * KHALF;
        const unsigned dst = wdst + (unsigned)(2 * HALO_B) + (unsigned)((q & (NWST - 1)) * WST_B);
#pragma unroll
        for (int rr = 0; rr < WROUNDS; ++rr) {
            const int row = rr * PPR + drow;
            mmh::lds_dma16(src + row * CO + ((dchunk ^ (row & 15)) << 2), dst + (unsigned)(rr * NT * 16));
        }
    };
    // one half (64 output channels) of the 9 x 17 dy halo of a tile; a tile past the list reads the zero line
    auto issue_halo = [&](int tile, int half) {
        const bool live = tile < t_end;
        const int b = tile / tpi;
        const int trem = tile - b * tpi;
        const int ty = trem / p.TX, tx = trem - ty * p.TX;
        const int ph0 = ty * TH, pw0 = tx * TW;
#pragma unroll
        for (int rr = 0; rr < HROUNDS; ++rr) {
            const int hp = rr * PPR + drow;
            const int hy = hp / HWD, hx = hp - hy * HWD;
            const int ph = ph0 + hy, pw = pw0 + hx;
            const bool ok = live && hp < HPIX_REAL && ph < p.Ho && pw < p.Wo;
            const void* g = ok ? (const void*)(p.dy + ((size_t)(b * p.Ho + ph) * p.Wo + pw) * CO + half * KHALF +
                                               ((dchunk ^ (hx & 15)) << 2))
                               : (const void*)(g_zero_line + (lane & 7) * 16);
            mmh::lds_dma16(g, wdst + (unsigned)(half * HALO_B) + (unsigned)(rr * NT * 16));
        }
    };

    // lane constants of the fragment reads.  dy fragment of tap shift (dh, dw): MFMA row r = position (2 pg + r / 16 + dh,
    // r % 16 + dw), chunk (2 kc + kg) ^ key; the address is base | ((kg ^ key) << 4), XOR-ed with kc << 5 per k-chunk
    unsigned a_lane[2][2];
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int dw = 0; dw < 2; ++dw) {
            const int hx = (r & 15) + dw;
            a_lane[dh][dw] = lds0 + (unsigned)(((2 * pg + (r >> 4) + dh) * HWD + hx) * PIXB) + (unsigned)((kg ^ (hx & 15)) << 4);
        }
    const unsigned b_lane = lds0 + (unsigned)(2 * HALO_B) + (unsigned)((ch * 32 + r) * PIXB) + (unsigned)((kg ^ (r & 15)) << 4);
    const bool odd = lane & 1, hi = lane & 2;
    const int n4 = ch * 32 + (r & ~3);          // after the epilogue's transpose: this lane's four channels
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (EPI && p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + n4);
    // store offsets (floats) inside a tile's block of dx: lane part = column 2 (4 kg + j), channels n4; the wave's rows 4 pg
    const unsigned st_lane = (unsigned)(2 * (4 * kg + (lane & 3)) * p.dx_cs + n4);
    const int pw_lane = 4 * kg + (lane & 3);

    int tile = xcd * p.per_xcd + slot;
    if (tile >= t_end) return;
    issue_halo(tile, 0);
    issue_w(0);
    const int H = 2 * p.Ho, W = 2 * p.Wo;
    const size_t rs = (size_t)W * p.dx_cs;
    bool counted = false;       // the previous tile left exactly 4 stores behind the DMA of phase 0
    for (; tile < t_end; tile += p.slots) {
        const int b = tile / tpi;
        const int trem = tile - b * tpi;
        const int ty = trem / p.TX, tx = trem - ty * p.TX;
        const bool full = ty * TH + TH <= p.Ho && tx * TW + TW <= p.Wo;     // whole tile: every lane issues every store
        f32x16 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;

        // Output class c = (row parity, column parity) is complete.  acc[c][i]: column n = ch 32 + r, row m = (i & 3) +
        // 8 (i >> 2) + 4 kg.  A 4 x 4 transpose over each quad of lanes gives lane (4 qd + j) the four channels
        // 4 qd .. 4 qd + 3 of row 8 g + 4 kg + j: four 16-byte stores per lane and class.
        float* const tile_o = p.dx + ((size_t)(b * H + 2 * (ty * TH + 2 * pg)) * W + 2 * tx * TW) * (size_t)p.dx_cs;   // uniform
        const int pw_room = p.Wo - tx * TW, ph_room = p.Ho - (ty * TH + 2 * pg);
        auto store_class = [&](auto cc) {
            constexpr int c = decltype(cc)::value;
            if (p.dbg & 1) return;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // row m = 8 g + 4 kg + j of the wave's 32 positions: dy row g / 2, column 8 (g & 1) + 4 kg + j
                float* const o = tile_o + (size_t)(2 * (g >> 1) + (c >> 1)) * rs + (size_t)((16 * (g & 1) + (c & 1)) * p.dx_cs);
                f32x4 v = quad_transpose(acc[c][4 * g], acc[c][4 * g + 1], acc[c][4 * g + 2], acc[c][4 * g + 3], odd, hi);
                if (EPI) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_of(v[e] + bv[e], p.act);
                }
                if (p.dbg & 16) asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(st_lane));
                else if (full || ((g >> 1) < ph_room && 8 * (g & 1) + pw_lane < pw_room)) *reinterpret_cast<f32x4*>(o + st_lane) = v;
            }
        };

        // Phase q = (half of the contraction, tap), taps in the order 4 | 3 5 | 1 7 | 0 2 6 8 so that the four classes
        // complete after phases 9, 11, 13 and 17 and each leaves while the next phases multiply (all 256 CUs run in step: 128 KiB
        // per CU written at once is a 32 MB burst that the next phase would have to wait out).  The vector-memory queue
        // retires in order - loads, LDS-DMA and stores alike - so the wait in front of phase q is vmcnt(what was issued behind
        // W(q)): the 5 halo DMA of phases 0 and 9, the 4 stores of a class.  The second halo half of this tile is requested
        // in phase 0, the first half of the NEXT tile in phase 9: both land nine phases before they are read.
        auto phase = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
            constexpr int half = q >= 9 ? 1 : 0, tap = tap_of(q - 9 * half);
            constexpr int kh = tap / 3, kw = tap - 3 * (tap / 3);
            constexpr int dh = kh == 0 ? 1 : 0, dw = kw == 0 ? 1 : 0;
            constexpr int cls = (kh != 1 ? 2 : 0) + (kw != 1 ? 1 : 0);
            asm volatile("" ::: "memory");
            if (q == 1 || q == 10) {
                __builtin_amdgcn_s_waitcnt(wait_vm(HROUNDS));
            } else if (q == 0) {
                if (counted) __builtin_amdgcn_s_waitcnt(wait_vm(4));
                else __builtin_amdgcn_s_waitcnt(wait_vm(0));
            } else if (q == 11 || q == 13 || q == 15) {
                if (full) __builtin_amdgcn_s_waitcnt(wait_vm(4));
                else __builtin_amdgcn_s_waitcnt(wait_vm(0));
            } else {
                __builtin_amdgcn_s_waitcnt(wait_vm(0));
            }
            __builtin_amdgcn_s_barrier();       // everybody's W(q) landed; stage (q + 1) & 1 is no longer read
            asm volatile("" ::: "memory");
            if (!(p.dbg & 4)) issue_w(q + 1 < 18 ? q + 1 : 0);
            if (q == 0 && !(p.dbg & 2)) issue_halo(tile, 1);
            if (q == 9 && !(p.dbg & 2)) issue_halo(tile + p.slots, 0);
            // the two waves of a SIMD (ch = 0 / 1) transpose and store a finished class at opposite ends of the phase: one
            // wave's vector-ALU work runs under the other's multiplies
            if (ch == 0) {
                if (q == 10) store_class(IC<0>{});
                if (q == 12) store_class(IC<1>{});
                if (q == 14) store_class(IC<2>{});
            }
            unsigned ab = a_lane[dh][dw] + (unsigned)(half * HALO_B);
            unsigned bb = b_lane + (unsigned)((q & (NWST - 1)) * WST_B);
            asm volatile("" : "+v"(ab), "+v"(bb));          // keeps the XOR-ed addresses out of the tile loop's preheader
            f32x4 af = lds_f4(ab), bf = lds_f4(bb);
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                f32x4 an, bn;
                if (kc + 1 < 8) {
                    an = lds_f4(ab ^ (unsigned)((kc + 1) << 5));
                    bn = lds_f4(bb ^ (unsigned)((kc + 1) << 5));
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    acc[cls] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[jj], bf[jj], acc[cls], 0, 0, 0);
                if (kc + 1 < 8) {
                    af = an;
                    bf = bn;
                }
            }
            if (ch != 0) {
                if (q == 10) store_class(IC<0>{});
                if (q == 12) store_class(IC<1>{});
                if (q == 14) store_class(IC<2>{});
            }
        };
        phase(IC<0>{}); phase(IC<1>{}); phase(IC<2>{}); phase(IC<3>{}); phase(IC<4>{}); phase(IC<5>{});
        phase(IC<6>{}); phase(IC<7>{}); phase(IC<8>{}); phase(IC<9>{}); phase(IC<10>{}); phase(IC<11>{});
        phase(IC<12>{}); phase(IC<13>{}); phase(IC<14>{}); phase(IC<15>{}); phase(IC<16>{}); phase(IC<17>{});
        store_class(IC<3>{});
        counted = full && !(p.dbg & 1);
    }
}

}  // namespace

namespace mmh {

bool dgrad_s2_halo_ok(const mmh_conv_desc* d, int dx_cs) {
    return g_dgrad_s2_halo && d->dtype == MMH_F32 && d->kh == 3 && d->kw == 3 && d->stride == 2 && d->pad == 1 &&
           d->pad_mode == MMH_PAD_ZERO && d->Cin == CI && d->Cout == CO && d->y_cs == CO && d->H == 2 * d->Ho &&
           d->W == 2 * d->Wo && dx_cs >= CI && dx_cs % 4 == 0 && (size_t)d->B * d->H * d->W * dx_cs < (1ull << 31);
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
    p.dbg = g_dgrad_s2_dbg;
    static int ready = -1;
    if (ready != 0) {
        for (const void* k : {reinterpret_cast<const void*>(dgrad_s2_kernel<false>), reinterpret_cast<const void*>(dgrad_s2_kernel<true>)}) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
            if (e != hipSuccess) return fail("dgrad_s2: %s", hipGetErrorString(e));
        }
        ready = 0;
    }
    if (bias || act != MMH_ACT_NONE) hipLaunchKernelGGL(dgrad_s2_kernel<true>, dim3(8 * p.slots), dim3(NT), LDS_B, st, p);
    else hipLaunchKernelGGL(dgrad_s2_kernel<false>, dim3(8 * p.slots), dim3(NT), LDS_B, st, p);
    return check_launch("dgrad_s2_kernel");
}

}  // namespace mmh

extern "C" int mmh_dgrad_s2_halo_supported(const mmh_conv_desc* d, int dx_cs) {
    return d && mmh::dgrad_s2_halo_ok(d, dx_cs) ? 1 : 0;
}
