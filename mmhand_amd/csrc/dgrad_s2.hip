// fp32 dgrad of the 3x3 / stride-2 / zero-pad-1 downsampling convs 64 -> 128 and 128 -> 256 (models/Generator.py:192-199:
// nn.Conv2d(ngf m, 2 ngf m, 3, 2, 1) of the three streams; the same arithmetic is the fprop of the decoder's
// nn.ConvTranspose2d(2 ngf m, ngf m, 3, 2, 1, output_padding=1), models/Generator.py:212-219) with the dy halo resident in LDS.
//
// The generic route (conv_igemm_multi_kernel) runs the four output parity classes as four separate implicit GEMMs with
// 1, 2, 2 and 4 taps: short contractions over a narrow N, every class re-gathering dy and writing every other pixel of
// dx - 76 TFLOP/s = 0.49 of the fp32 MFMA peak at 64 -> 128 @256x256.  Here one work-group owns an 8 x 16 block of dy
// positions: dx[2 ph + a][2 pw + b] (a, b in {0, 1}) needs dy[ph .. ph + 1][pw .. pw + 1] only,
//
//   class (0,0): tap (1,1) dy[ph][pw]                   class (0,1): (1,0) dy[ph][pw+1], (1,2) dy[ph][pw]
//   class (1,0): (0,1) dy[ph+1][pw], (2,1) dy[ph][pw]   class (1,1): (0,0) dy[ph+1][pw+1], (0,2) dy[ph+1][pw],
//                                                                     (2,0) dy[ph][pw+1],   (2,2) dy[ph][pw]
//
// so the 9 x 17 halo of dy serves all nine taps and all four classes.  The contraction (Cout) runs in chunks of 64
// channels: a halo chunk is 40 KiB (LDS-DMA, zero-filled past the image), TWO of them are resident, and while chunk k
// multiplies, chunk k + 1 - of this tile or of the next one - is on its way, nine phases ahead of its first read.  The
// filter is streamed one (chunk, tap) at a time ([Cin][64] = 16 / 32 KiB, two stages).  512 threads = 8 waves:
// wave = (2 dy rows x 16 positions) x half of Cin x 4 classes = 4 (8 at Cin 128) 32x32 accumulator tiles; per 8-deep
// k-step one ds_read_b128 of dy and one (two) of the filter feed four (eight) v_mfma_f32_32x32x2_f32.  LDS images are
// XOR-swizzled per 16-byte chunk (key = halo column & 15 for dy, ci & 15 for the filter: distinct over each of
// ds_read_b128's 16-lane groups), applied on the global side of the DMA.  The kernel is persistent (one work-group per
// CU, XCD-contiguous tile lists).  Taps run in the order 4 | 3 5 | 1 7 | 0 2 6 8, so in the last chunk the classes
// complete one after the other and each leaves - a 4 x 4 transpose over lane quads, 16-byte stores - under the
// following phases' multiplies; the two waves of a SIMD do that at opposite ends of a phase.  Measured (B=32):
// 64 -> 128 @256x256 620 us = 125 TFLOP/s = 0.79 (generic 1003 us), bit-identical to the generic kernels (same order of
// summation per class).  Ablations (tools/ablate_dgrad_s2.py): no DMA, no stores 545 us = 142 TFLOP/s.
#include <algorithm>
#include <type_traits>
#include "common.h"

namespace mmh { int g_dgrad_s2_halo = 1; int g_dgrad_s2_dbg = 0; }

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const f32x4 __attribute__((address_space(3))) * lds_f4_p;
__device__ __forceinline__ f32x4 lds_f4(unsigned addr) { return *reinterpret_cast<lds_f4_p>((size_t)addr); }

constexpr int KC = 64;                          // channels of one chunk of the contraction
constexpr int TH = 8, TW = 16;                  // dy positions per tile
constexpr int HWD = TW + 1;                     // halo 9 x 17
constexpr int HPIX_REAL = (TH + 1) * HWD;       // 153
constexpr int PIXB = KC * 4;                    // 256 bytes per dy pixel chunk / per filter row chunk
constexpr int NT = 512;
constexpr int PPR = NT * 16 / PIXB;             // pixels (rows) per DMA round = 32
constexpr int HROUNDS = (HPIX_REAL + PPR - 1) / PPR;     // 5
constexpr int HALO_B = HROUNDS * PPR * PIXB;    // 40960 per chunk
constexpr int lds_bytes(int ntn) { return 2 * HALO_B + 2 * (64 * ntn) * PIXB; }      // 114688 / 147456

__device__ char g_zero_line[128];               // DMA source of the zero padding
// LDS-DMA with a scalar base and a 32-bit lane offset (mmh::lds_dma16 takes a 64-bit pointer per lane)
__device__ __forceinline__ void dma16_s(const void* sbase, unsigned voff, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_base) : "memory", "m0");
}
// s_waitcnt vmcnt(n) lgkmcnt(0) (gfx9 encoding: vmcnt in bits 3:0 and 15:14, expcnt 6:4 left at 7)
constexpr int wait_vm(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0070; }
template <int V> using IC = std::integral_constant<int, V>;
// taps in the order 4 | 3 5 | 1 7 | 0 2 6 8: the output classes complete after 1, 3, 5 and 9 taps
constexpr int tap_of(int i) { return i == 0 ? 4 : i == 1 ? 3 : i == 2 ? 5 : i == 3 ? 1 : i == 4 ? 7 : i == 5 ? 0 : i == 6 ? 2 : i == 7 ? 6 : 8; }

struct DgradS2KP {
    const float* dy;        // [B][Ho][Wo][Cout]
    const float* w;         // [3][3][Cin][Cout]
    const float* bias;      // [Cin] or null (ConvTranspose fprop)
    float* dx;              // [B][2 Ho][2 Wo][dx_cs]
    int B, Ho, Wo, dx_cs, act;
    int TX, TY, tiles, per_xcd, slots, dbg;
};


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

// EPI: bias and / or ReLU in the epilogue (the ConvTranspose2d forward); false: plain dgrad.
// NTN: 32-column accumulator tiles per wave and class (Cin = 64 NTN).  KQ: chunks of the contraction (Cout = 64 KQ, even).
template <bool EPI, int NTN, int KQ>
__global__ void __launch_bounds__(NT, 1) dgrad_s2_kernel(const DgradS2KP p) {
    constexpr int CI = 64 * NTN, CO = KC * KQ;
    constexpr int WST_B = CI * PIXB;            // one (chunk, tap) of the filter
    constexpr int WROUNDS = CI / PPR;
    constexpr int NQ = 9 * KQ;                  // phases per tile
    constexpr int NSC = 4 * NTN;                // stores per lane and class
    static_assert(KQ % 2 == 0, "the halo buffers alternate per chunk and a tile must start on buffer 0");
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

    // filter of phase q = (chunk, tap) -> stage q & 1: rows ci, the chunk's 64 output channels.  Row rr 32 + drow of a round:
    // the swizzle key (row & 15) does not depend on the round, so ONE lane offset serves every (phase, round); the rest of
    // the address is scalar (wq is opaque per tile: 36 phases x 4 rounds of hoisted addresses would not fit in registers)
    const unsigned w_lane_off = (unsigned)(drow * CO + ((dchunk ^ (drow & 15)) << 2)) * 4u;
    const float* wq = p.w;
    auto issue_w = [&](int q) {
        const int kq = q / 9, tap = tap_of(q - 9 * kq);
        const float* src = wq + (size_t)tap * (CI * CO) + kq * KC;
        const unsigned dst = wdst + (unsigned)(2 * HALO_B) + (unsigned)((q & 1) * WST_B);
#pragma unroll
        for (int rr = 0; rr < WROUNDS; ++rr) dma16_s(src + rr * PPR * CO, w_lane_off, dst + (unsigned)(rr * NT * 16));
    };
    // chunk kq of the 9 x 17 dy halo of a tile -> halo buffer kq & 1; a tile past the list reads the zero line.  Halo pixel
    // hp = rr 32 + drow of round rr: its (row, column) and its offset in dy are lane constants
    unsigned h_off[HROUNDS], h_yx[HROUNDS];
#pragma unroll
    for (int rr = 0; rr < HROUNDS; ++rr) {
        const int hp = rr * PPR + drow;
        const int hy = hp / HWD, hx = hp - hy * HWD;
        h_off[rr] = (unsigned)((hy * p.Wo + hx) * CO + ((dchunk ^ (hx & 15)) << 2));
        h_yx[rr] = hp < HPIX_REAL ? (unsigned)(hy | (hx << 8)) : 0xffffu;
    }
    auto issue_halo = [&](int tile, int kq) {
        const bool live = tile < t_end;
        const int b = tile / tpi;
        const int trem = tile - b * tpi;
        const int ty = trem / p.TX, tx = trem - ty * p.TX;
        const int ph_room = live ? p.Ho - ty * TH : 0, pw_room = p.Wo - tx * TW;
        const float* src = p.dy + ((size_t)(b * p.Ho + ty * TH) * p.Wo + tx * TW) * CO + kq * KC;      // uniform
#pragma unroll
        for (int rr = 0; rr < HROUNDS; ++rr) {
            const bool ok = (int)(h_yx[rr] & 0xff) < ph_room && (int)(h_yx[rr] >> 8) < pw_room;
            const void* g = ok ? (const void*)(src + h_off[rr]) : (const void*)(g_zero_line + (lane & 7) * 16);
            mmh::lds_dma16(g, wdst + (unsigned)((kq & 1) * HALO_B) + (unsigned)(rr * NT * 16));
        }
    };

    // lane constants of the fragment reads.  dy fragment of tap shift (dh, dw): MFMA row r = position (2 pg + r / 16 + dh,
    // r % 16 + dw), chunk (2 kc + kg) ^ key; the address is base | ((kg ^ key) << 4), XOR-ed with kc << 5 per k-step
    unsigned a_lane[2][2];
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int dw = 0; dw < 2; ++dw) {
            const int hx = (r & 15) + dw;
            a_lane[dh][dw] = lds0 + (unsigned)(((2 * pg + (r >> 4) + dh) * HWD + hx) * PIXB) + (unsigned)((kg ^ (hx & 15)) << 4);
        }
    // filter fragment of column tile nt: row (ch NTN + nt) 32 + r
    const unsigned b_lane = lds0 + (unsigned)(2 * HALO_B) + (unsigned)((ch * NTN * 32 + r) * PIXB) + (unsigned)((kg ^ (r & 15)) << 4);
    const bool odd = lane & 1, hi = lane & 2;
    const int n4 = ch * NTN * 32 + (r & ~3);        // after the epilogue's transpose: this lane's four channels (tile 0)
    f32x4 bv[NTN];
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt) {
        bv[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (EPI && p.bias) bv[nt] = *reinterpret_cast<const f32x4*>(p.bias + n4 + 32 * nt);
    }
    // store offsets (floats) inside a tile's block of dx: lane part = column 2 (4 kg + j), channels n4; the wave's rows 4 pg
    const unsigned st_lane = (unsigned)(2 * (4 * kg + (lane & 3)) * p.dx_cs + n4);
    const int pw_lane = 4 * kg + (lane & 3);
    const float relu_floor = p.act == MMH_ACT_RELU ? 0.f : -__builtin_inff();

    int tile = xcd * p.per_xcd + slot;
    if (tile >= t_end) return;
    issue_halo(tile, 0);
    issue_w(0);
    const int H = 2 * p.Ho, W = 2 * p.Wo;
    const size_t rs = (size_t)W * p.dx_cs;
    bool counted = false;       // the previous tile left exactly NSC stores behind the DMA of phase 0
    for (; tile < t_end; tile += p.slots) {
        const int b = tile / tpi;
        const int trem = tile - b * tpi;
        const int ty = trem / p.TX, tx = trem - ty * p.TX;
        const bool full = ty * TH + TH <= p.Ho && tx * TW + TW <= p.Wo;     // whole tile: every lane issues every store
        wq = p.w;
        asm volatile("" : "+s"(wq));
        f32x16 acc[4][NTN];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[c][nt][i] = 0.f;

        // Output class c = (row parity, column parity) is complete.  acc[c][nt][i]: column n = (ch NTN + nt) 32 + r, row
        // m = (i & 3) + 8 (i >> 2) + 4 kg.  A 4 x 4 transpose over each quad of lanes gives lane (4 qd + j) the four channels
        // 4 qd .. 4 qd + 3 of row 8 g + 4 kg + j: NSC 16-byte stores per lane and class.
        float* const tile_o = p.dx + ((size_t)(b * H + 2 * (ty * TH + 2 * pg)) * W + 2 * tx * TW) * (size_t)p.dx_cs;   // uniform
        const int pw_room = p.Wo - tx * TW, ph_room = p.Ho - (ty * TH + 2 * pg);
        auto store_class = [&](auto cc) {
            constexpr int c = decltype(cc)::value;
            if (p.dbg & 1) return;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // row m = 8 g + 4 kg + j of the wave's 32 positions: dy row g / 2, column 8 (g & 1) + 4 kg + j
                float* const o = tile_o + (size_t)(2 * (g >> 1) + (c >> 1)) * rs + (size_t)((16 * (g & 1) + (c & 1)) * p.dx_cs);
                const bool ok = full || ((g >> 1) < ph_room && 8 * (g & 1) + pw_lane < pw_room);
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt) {
                    f32x4 v = quad_transpose(acc[c][nt][4 * g], acc[c][nt][4 * g + 1], acc[c][nt][4 * g + 2], acc[c][nt][4 * g + 3],
                                             odd, hi);
                    if (EPI) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] + bv[nt][e], relu_floor);
                    }
                    if (p.dbg & 16) asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(st_lane));
                    else if (ok) *reinterpret_cast<f32x4*>(o + st_lane + 32 * nt) = v;
                }
            }
        };

        // Phase q = (chunk of the contraction, tap).  The vector-memory queue retires in order - loads, LDS-DMA and stores
        // alike - so the wait in front of phase q is vmcnt(what was issued behind W(q)): the 5 halo DMA of a chunk's first
        // phase, the NSC stores of a class (a ragged tile's stores are predicated: it waits for everything).  At the first
        // phase of chunk kq the halo of chunk kq + 1 is requested - for the last chunk, chunk 0 of the NEXT tile.
        auto phase = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
            constexpr int kq = q / 9, ti = q - 9 * kq, tap = tap_of(ti);
            constexpr int kh = tap / 3, kw = tap - 3 * (tap / 3);
            constexpr int dh = kh == 0 ? 1 : 0, dw = kw == 0 ? 1 : 0;
            constexpr int cls = (kh != 1 ? 2 : 0) + (kw != 1 ? 1 : 0);
            constexpr bool last = kq == KQ - 1;
            // what the previous phase issued behind W(q)
            constexpr bool after_halo = q > 0 && (q - 1) % 9 == 0;
            constexpr bool after_store = (q == 0) || (kq == KQ - 1 && (ti == 2 || ti == 4 || ti == 6));
            asm volatile("" ::: "memory");
            if (after_halo) {
                __builtin_amdgcn_s_waitcnt(wait_vm(HROUNDS));
            } else if (after_store) {
                if (q == 0 ? counted : full) __builtin_amdgcn_s_waitcnt(wait_vm(NSC));
                else __builtin_amdgcn_s_waitcnt(wait_vm(0));
            } else {
                __builtin_amdgcn_s_waitcnt(wait_vm(0));
            }
            __builtin_amdgcn_s_barrier();       // everybody's W(q) landed; stage (q + 1) & 1 is no longer read
            asm volatile("" ::: "memory");
            if (!(p.dbg & 4)) issue_w(q + 1 < NQ ? q + 1 : 0);
            if (ti == 0 && !(p.dbg & 2)) {
                if (last) issue_halo(tile + p.slots, 0);
                else issue_halo(tile, kq + 1);
            }
            // the two waves of a SIMD (ch = 0 / 1) transpose and store a finished class at opposite ends of the phase: one
            // wave's vector-ALU work runs under the other's multiplies
            if (last && ch == 0) {
                if (ti == 1) store_class(IC<0>{});
                if (ti == 3) store_class(IC<1>{});
                if (ti == 5) store_class(IC<2>{});
            }
            unsigned ab = a_lane[dh][dw] + (unsigned)((kq & 1) * HALO_B);
            unsigned bb = b_lane + (unsigned)((q & 1) * WST_B);
            asm volatile("" : "+v"(ab), "+v"(bb));          // keeps the XOR-ed addresses out of the tile loop's preheader
            f32x4 af = lds_f4(ab), bf[NTN];
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) bf[nt] = lds_f4(bb + (unsigned)(nt * 32 * PIXB));
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                f32x4 an, bn[NTN];
                if (kc + 1 < 8) {
                    an = lds_f4(ab ^ (unsigned)((kc + 1) << 5));
#pragma unroll
                    for (int nt = 0; nt < NTN; ++nt) bn[nt] = lds_f4((bb ^ (unsigned)((kc + 1) << 5)) + (unsigned)(nt * 32 * PIXB));
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int nt = 0; nt < NTN; ++nt)
                        acc[cls][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[jj], bf[nt][jj], acc[cls][nt], 0, 0, 0);
                if (kc + 1 < 8) {
                    af = an;
#pragma unroll
                    for (int nt = 0; nt < NTN; ++nt) bf[nt] = bn[nt];
                }
            }
            if (last && ch != 0) {
                if (ti == 1) store_class(IC<0>{});
                if (ti == 3) store_class(IC<1>{});
                if (ti == 5) store_class(IC<2>{});
            }
        };
        auto chunk = [&](auto kc_) {
            constexpr int q0 = 9 * decltype(kc_)::value;
            phase(IC<q0>{}); phase(IC<q0 + 1>{}); phase(IC<q0 + 2>{}); phase(IC<q0 + 3>{}); phase(IC<q0 + 4>{});
            phase(IC<q0 + 5>{}); phase(IC<q0 + 6>{}); phase(IC<q0 + 7>{}); phase(IC<q0 + 8>{});
        };
        chunk(IC<0>{});
        chunk(IC<1>{});
        if constexpr (KQ == 4) {
            chunk(IC<2>{});
            chunk(IC<3>{});
        }
        store_class(IC<3>{});
        counted = full && !(p.dbg & 1);
    }
}

}  // namespace

namespace mmh {

bool dgrad_s2_halo_ok(const mmh_conv_desc* d, int dx_cs, int act) {
    return g_dgrad_s2_halo && act != MMH_ACT_TANH && d->dtype == MMH_F32 && d->kh == 3 && d->kw == 3 && d->stride == 2 && d->pad == 1 &&
           d->pad_mode == MMH_PAD_ZERO && ((d->Cin == 64 && d->Cout == 128) || (d->Cin == 128 && d->Cout == 256)) &&
           d->y_cs == d->Cout && d->H == 2 * d->Ho && d->W == 2 * d->Wo && dx_cs >= d->Cin && dx_cs % 4 == 0 &&
           (size_t)d->B * d->H * d->W * dx_cs < (1ull << 31);
}

template <bool EPI, int NTN, int KQ>
static int launch_t(const DgradS2KP& p, hipStream_t st) {
    static int ready = -1;
    if (ready != 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dgrad_s2_kernel<EPI, NTN, KQ>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(NTN));
        if (e != hipSuccess) return fail("dgrad_s2: %s", hipGetErrorString(e));
        ready = 0;
    }
    hipLaunchKernelGGL((dgrad_s2_kernel<EPI, NTN, KQ>), dim3(8 * p.slots), dim3(NT), lds_bytes(NTN), st, p);
    return check_launch("dgrad_s2_kernel");
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
    const bool epi = bias || act != MMH_ACT_NONE;
    if (d->Cin == 64) return epi ? launch_t<true, 1, 2>(p, st) : launch_t<false, 1, 2>(p, st);
    return epi ? launch_t<true, 2, 4>(p, st) : launch_t<false, 2, 4>(p, st);
}

}  // namespace mmh

extern "C" int mmh_dgrad_s2_halo_supported(const mmh_conv_desc* d, int dx_cs) {
    return d && mmh::dgrad_s2_halo_ok(d, dx_cs, MMH_ACT_NONE) ? 1 : 0;
}
