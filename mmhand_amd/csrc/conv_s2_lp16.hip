// 16-bit fprop of the 3x3 / stride-2 / zero-pad-1 down-sampling convolutions (nn.Conv2d(c, 2c, 3, 2, 1): models/Generator.py
// :166-180, models/Discriminator.py:86-92; 64 -> 128 at 256^2 -> 128^2 and 128 -> 256 at 128^2 -> 64^2 in the step) and,
// the same GEMM, the input gradient of the decoder's ConvTranspose2d(2c, c, 3, 2, 1, 1) - gfx950.
//
// Why a kernel of its own: on the general kernel (conv_lp16g_kernel, conv_lp16.hip) these launches sit at
// 0.16 / 0.23 of the bf16 MFMA peak.  Their contraction is short (K = 9 * 64 or 9 * 128) and the general kernel re-stages,
// per 256-pixel tile and tap, one shifted copy of the input AND the tap's weights through LDS-DMA: 432 KB per 37.7 MFLOP
// tile (87 FLOP per staged byte), and a CU pulls 25-45 GB/s through that path.  This kernel stages per output tile only
// what is new:
//   * WEIGHTS LIVE IN REGISTERS for the whole launch.  The MFMA's first operand is the weight fragment (16 channels x 32
//     K-values: 4 VGPRs per lane); a wave owns NJ 16-channel column tiles and keeps their fragments of all nine taps and the
//     whole contraction - 9 taps x (C / 32) k-steps x NJ x 4 = 144 VGPRs for (C = 64, NJ = 2) and for (C = 128, NJ = 1) -
//     loaded once per workgroup, which is persistent (one per CU) and walks a list of output tiles;
//   * the INPUT HALO of an 8 x 16 output tile (17 x 33 pixels) is brought into LDS once per 64-channel chunk (72 KB) by
//     LDS-DMA and read by all nine taps; two chunk buffers: the halo of the next chunk (of this tile or of the next)
//     travels while the current one is multiplied.  Per tile 72 KB (C = 64) for 18.9 MFLOP: 262 FLOP per staged byte;
//   * stride 2 in the image becomes stride 1 in LDS: a halo row is stored DE-INTERLEAVED - its even columns in slots 0..16,
//     its odd columns in slots 17..32 (the DMA's per-lane source address does that for free) - so the 16 output pixels of
//     a tile row read, for kw = 0 / 1 / 2, sixteen CONSECUTIVE 128-byte slots starting at 0 / 17 / 1, exactly the fragment
//     shape of the stride-1 halo kernel, with its swizzle (16-byte chunk index ^ (slot & 6): conflict-free for every start
//     under ds_read_b128's real lane groups).
// Waves: 8 = MW (along the tile's 8 rows) x 8 / MW column groups of NJ x 16 channels: 128 output channels per workgroup
// (N = 256: two workgroups per tile list, neighbours on one XCD, so the second reads the halos from L2).
// HBM floor of 64 -> 128 at B = 32: 268 MB in + 134 MB out = 67 us at 6 TB/s = 0.46 of the bf16 peak - these convs are
// close to memory-bound at the rates the kernel is built for.
#include <algorithm>
#include "common.h"

namespace mmh { int g_lp16_s2f = 1; }       // mmh_set_option("lp16_s2f", 0): the general kernel (A/B); 2: also for 128 input channels

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 16;                  // output tile
constexpr int ODD0 = 17;                        // stride 2: first slot of the odd columns (halo 17 x 33 input pixels in 34-slot
                                                // rows: 578 LDS rows of 128 B = 73984 B per chunk buffer, 10 DMA instructions per
                                                // wave and chunk, two buffers = 147968 B; HaloGeom below)

struct S2KP {
    const char* x;          // 16-bit [B][H][W][cs]
    const char* w;          // 16-bit [tap][N][C]
    const char* zeros;
    float* y;               // fp32 [B][Ho][Wo][y_cs] ...
    char* y16;              // ... or 16-bit
    const float* bias;
    int B, H, W, cs, C;
    int Ho, Wo, N, y_cs, act;
    int TX, TY, tiles;      // tiles = B * TY * TX
    int nsplit;             // N / 128
    int lists;              // tile lists = workgroups / nsplit
    float* stats;           // per (image, tile, row half, channel) count / mean / M2 of the stored outputs, [B][chunks][3][N], or null
    int rev;                // taps mirrored (w = the plain copy [tap][Cin][Cout]: the input gradient of a stride-1 conv)
    int dbg;                // timing-only ablations (mmh_set_option "lp16_dbg"; results wrong): 1 no halo DMA after the first, 2 no MFMAs
};

template <bool H16>
__device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) {
    if (H16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <bool H16>
__device__ __forceinline__ void store4(float* y, char* y16, size_t elem, f32x4 v, const float* bv, int act) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float t = v[r] + bv[r];
        v[r] = act == MMH_ACT_RELU ? (t > 0.f ? t : 0.f) : (act == MMH_ACT_TANH ? tanhf(t) : t);
    }
    if (y16) {
        if (H16) {
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            const h4 o = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            *reinterpret_cast<h4*>(y16 + elem * 2) = o;
        } else {
            typedef __bf16 b4 __attribute__((ext_vector_type(4)));
            const b4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            *reinterpret_cast<b4*>(y16 + elem * 2) = o;
        }
    } else {
        *reinterpret_cast<f32x4*>(y + elem) = v;
    }
}

typedef const bf16x8 __attribute__((address_space(3))) * lds_frag_p;
__device__ __forceinline__ bf16x8 lds_frag(unsigned addr) { return *reinterpret_cast<lds_frag_p>(addr); }

// The halo of an 8 x 16 output tile under image stride S: S = 2 as above (de-interleaved rows); S = 1 (the stride-1 form of the
// same kernel: VGG19's conv1_2, 64 -> 64 at full resolution, losses/L1_plus_perceptualLoss.py:22-27 - on the general
// kernel 359-400 us = 0.16 of the peak, three launches per step, because that kernel re-stages the input once per tap: 360 KB
// per 256-pixel tile where this one brings 23 KB) a plain 10 x 18 halo, the taps starting at slots 0 / 1 / 2.
template <int S> struct HaloGeom {
    static constexpr int HH = S * TH + (3 - S), HWD = S * TW + (3 - S);
    static constexpr int PITCH = S == 2 ? 34 : 18;      // LDS slots per halo row (even: slot parity = row parity)
    static constexpr int ROWS = HH * PITCH, BUF_B = ROWS * 128, ROUNDS = (ROWS + 63) / 64, LDS_B = 2 * BUF_B;
};

// KC = C / 64 chunks; MW waves along the rows (MI = 8 / MW rows each), NJ column tiles per wave; S the image stride; NOUT the
// output channels of one workgroup (128; 64 for the 64 -> 64 convs)
template <bool H16, int KC, int MW, int NJ, bool STATS, int S = 2, int NOUT = 128>
__global__ void __launch_bounds__(512) conv_s2f_kernel(const S2KP p) {
    constexpr int MI = TH / MW;             // output rows (16-pixel MFMA column tiles) per wave
    constexpr int NG = 8 / MW;              // column groups
    static_assert(NG * NJ * 16 == NOUT, "a workgroup covers NOUT output channels");
    typedef HaloGeom<S> G;
    constexpr int PITCH = G::PITCH, ROWS = G::ROWS, BUF_B = G::BUF_B, ROUNDS = G::ROUNDS, HWD = G::HWD;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wm = wave / NG, wn = wave - wm * NG;
    // persistent tile lists: XCD x owns a contiguous range of tiles, cut into `lists / 8` lists; the workgroups of one
    // list (one per 128-column half) are neighbours on that XCD
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nh = slot % p.nsplit, lst = slot / p.nsplit;
    const int per_xcd = (p.tiles + 7) / 8, lists_x = p.lists / 8;
    const int per_list = (per_xcd + lists_x - 1) / lists_x;
    const int t_begin = xcd * per_xcd + lst * per_list;
    const int t_end = min(min(t_begin + per_list, (xcd + 1) * per_xcd), p.tiles);
    if (t_begin >= t_end) return;
    const int n0 = nh * NOUT + wn * (NJ * 16);

    // the wave's weights: fragment (tap, k32 step s, column tile j) = rows n0 + 16 j + l15, K-values 32 s + 8 g4 .. + 7
    bf16x8 wf[9][2 * KC][NJ];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int s = 0; s < 2 * KC; ++s)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                wf[t][s][j] = *reinterpret_cast<const bf16x8*>(       // p.rev: the taps mirrored (the dgrad of a stride-1 conv)
                    p.w + ((size_t)((p.rev ? 8 - t : t) * p.N + n0 + 16 * j + l15) * p.C + 32 * s + 8 * g4) * 2);
    float bv[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[j][r] = p.bias ? p.bias[n0 + 16 * j + 4 * g4 + r] : 0.f;

    // halo DMA roles: round rd moves LDS rows rd * 64 + wave * 8 + lane / 8 (row = hy * PITCH + slot), 16-byte chunk lane & 7
    // of the row = global chunk (lane & 7) ^ (slot & 6).  The source offsets are computed where the chunk is issued (once
    // per 72 KB of DMA) and not kept: ten registers the resident weights need more
    const unsigned lds0 = mmh::lds_addr_of(smem);
    auto issue_chunk = [&](int tile, int kc, int buf) {
        const int b = tile / (p.TX * p.TY);
        const int rem = tile - b * (p.TX * p.TY);
        const int ty = rem / p.TX, tx = rem - ty * p.TX;
        const int ih0 = S * ty * TH - 1, iw0 = S * tx * TW - 1;
        const char* xb = p.x + (size_t)kc * 128;
        const unsigned dst = lds0 + (unsigned)buf * BUF_B + (unsigned)wave * 1024u;
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const int r = rd * 64 + wave * 8 + (lane >> 3);
            const int hy = r / PITCH, sl = r - hy * PITCH;
            const int hx = S == 1 ? sl : (sl < ODD0 ? 2 * sl : 2 * (sl - ODD0) + 1);
            const int ih = ih0 + hy, iw = iw0 + hx;
            const bool row = r < ROWS && sl < HWD;
            const bool ok = row && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            const unsigned q8 = (unsigned)((lane & 7) ^ (sl & 6));
            if (row) {
                const char* g = ok ? xb + ((size_t)((unsigned)((b * p.H + ih) * p.W + iw) * (unsigned)p.cs * 2u + q8 * 16u))
                                   : p.zeros + (lane & 7) * 16;
                mmh::lds_dma16(g, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + (unsigned)rd * 8192u)));
            }
        }
    };
    // fragment addresses: pixel column l15 of output row wm * MI + i under tap (kh, kw) = halo row 2 (wm MI + i) + kh, slot
    // start(kw) + l15, start = 0 / 17 / 1; chunk (4 hf + g4) ^ (slot & 6).  Lane-constant part per kw; the row and the
    // buffer are immediates / scalars
    unsigned aL[3][2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const unsigned sl = (unsigned)((S == 1 ? kw : (kw == 0 ? 0 : (kw == 1 ? ODD0 : 1))) + l15);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
            aL[kw][hf] = lds0 + ((unsigned)(S * wm * MI) * PITCH + sl) * 128u + ((((unsigned)(4 * hf + g4)) ^ (sl & 6u)) << 4);
    }

    f32x4 acc[MI][NJ];
    int q = 0;                              // chunk items done: item q lives in buffer q & 1
    issue_chunk(t_begin, 0, 0);
    bool after_store = false;
    for (int tile = t_begin; tile < t_end; ++tile) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            // item q has landed: the queue retires in order, so everything but the MI * NJ stores of the epilogue just
            // behind the DMA is enough after a tile boundary
            if (after_store && STATS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MI * NJ + 3) : "memory");
            else if (after_store) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MI * NJ) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            after_store = false;
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // the next item into the other buffer (everybody is past its last reader: the barrier above)
            if (!(p.dbg & 1)) {
                if (kc + 1 < KC) issue_chunk(tile, kc + 1, (q + 1) & 1);
                else if (tile + 1 < t_end) issue_chunk(tile + 1, 0, (q + 1) & 1);
            }
            const unsigned boff = (unsigned)(q & 1) * BUF_B;
            // blocks of RB rows under one (tap, k32 step): RB pixel fragments (ds_read_b128) feed RB * NJ MFMAs; the fragments
            // of block n + 1 are requested before the MFMAs of block n (one block of lookahead, pinned: left alone the
            // scheduler hoists the reads of several taps and spills the resident weights)
            constexpr int RB = MI < 4 ? MI : 4, RH = MI / RB, NBLK = 9 * 2 * RH;
            auto frag = [&](int blk, int i) -> bf16x8 {
                const int tap = blk / (2 * RH), r2 = blk - tap * (2 * RH), hf = r2 / RH, rh = r2 - hf * RH;
                const int kh = tap / 3, kw = tap - kh * 3;
                return lds_frag(aL[kw][hf] + boff + (unsigned)((S * (rh * RB + i) + kh) * PITCH) * 128u);
            };
            bf16x8 af[2][RB];
            if (!(p.dbg & 2)) {
#pragma unroll
            for (int i = 0; i < RB; ++i) af[0][i] = frag(0, i);
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk) {
                const int tap = blk / (2 * RH), r2 = blk - tap * (2 * RH), hf = r2 / RH, rh = r2 - hf * RH;
                if (blk + 1 < NBLK) {
#pragma unroll
                    for (int i = 0; i < RB; ++i) af[(blk + 1) & 1][i] = frag(blk + 1, i);
                }
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[rh * RB + i][j] = mfma<H16>(wf[tap][2 * kc + hf][j], af[blk & 1][i], acc[rh * RB + i][j]);
                if (blk + 1 < NBLK) __builtin_amdgcn_sched_group_barrier(0x100, RB, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, RB * NJ, 0);
            }
            }
            ++q;
        }
        // epilogue: lane (l15, g4) holds channels 4 g4 .. + 3 of pixel l15 of each of its rows, in each of its NJ column tiles
        // (tried: the lanes g4 / g4 ^ 1 trading one accumulator each so that a lane stores 16 bytes - half the store
        // instructions, what gives conv_lp16h2_kernel 11 % at 256 -> 256 (common.h: pair_swap8) - ran 152 us (__shfl_xor) and
        // 163 us (v_permlane16_swap) against 116: tools/bench_s2f.py; and in the stride-1 form, which has registers to spare,
        // 246 us against 210: tools/bench_s1f.py)
        const int b = tile / (p.TX * p.TY);
        const int rem = tile - b * (p.TX * p.TY);
        const int ty = rem / p.TX, tx = rem - ty * p.TX;
        {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const size_t pix = ((size_t)b * p.Ho + (ty * TH + wm * MI + i)) * p.Wo + (tx * TW + l15);
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    store4<H16>(p.y, p.y16, pix * p.y_cs + (n0 + 16 * j + 4 * g4), acc[i][j], bv[j], p.act);
            }
        }
        if (STATS) {        // the InstanceNorm behind this conv merges these partials instead of reading y (common.h)
            const int chunks = p.TX * p.TY * MW;
            float* sp = p.stats + ((size_t)(b * chunks + (ty * p.TX + tx) * MW + wm) * 3) * p.N + n0 + 4 * g4;
            mmh::wave_tile_stats<MI, NJ>([&](int i, int j, int r) {
                const float t = acc[i][j][r] + bv[j][r];
                return H16 ? (float)(_Float16)t : (float)(__bf16)t;
            }, l15, sp, p.N);
        }
        after_store = true;
    }
}

// ---------------------------------------------------------------------------------------------
// The input gradient of the same convolution (and the forward of the decoder's ConvTranspose2d(2c, c, 3, 2, 1, 1)): dx[2p + a]
// [2q + b] = sum over the taps of parity class (a, b) of dy[p + dh][q + dw] . w[kh][kw], dh = (kh == 0), dw = (kw == 0) - one
// tap for class (0,0), two for (0,1) and (1,0), four for (1,1).  The general kernel runs the four classes as four GEMMs that
// each re-stage dy and their taps' weights; here a tile of 8 x 16 dy positions brings its 9 x 17 halo of dy (128 channels:
// 41 KB) in ONCE for all nine taps and all four classes, and the weights are register-resident as in the fprop kernel.
//
// Round 5: the waves split the CLASSES, not the rows.  Round 4's form gave a wave one 16-channel column tile of dx, four of the
// tile's eight rows, all nine taps (144 weight registers) and all four classes (64 accumulators): with the double-buffered
// fragments that is 260 live registers against the 256 two waves per SIMD leave - 37 dwords in scratch, and every reload is a
// scratch_load followed by s_waitcnt vmcnt(0): in front of each of the six halo-DMA instructions (the DMA addresses had gone
// to scratch) and inside the MFMA block (weight fragments had), i.e. every tile waited for its own next-tile DMA and for the
// previous tile's stores.  That is the "stores and DMA each cost what they would alone and nothing overlaps" of round 4's
// ablations - found with VERDICT r4 #2's counters (no memory-side stall: TCC_EA0_WRREQ_STALL 0.25 M against a plain fill's
// 3.2 M on the same bytes, profiles/r05_pmc_s2d.txt) and then in the ISA (tools/isa.sh: private segment 148 bytes).
// Now a wave owns one column tile, ALL eight rows and the taps of TWO classes: type 0 (waves 0-3) the classes (1,1) + (0,0) -
// five taps, 80 weight registers -, type 1 (waves 4-7) the classes (0,1) + (1,0) - four taps, 64 registers; wave w and w + 4
// share a SIMD, so every SIMD multiplies 5 + 4 taps as before.  Per (k32 step, row) the shifted fragments (dh, dw) of row i
// are the (0, dw) fragments of row i + 1: a ring of four row slots (two fragments each) is refilled one row ahead of use -
// 18 reads per k-step instead of 32.  ~200 registers, nothing in scratch: the DMA of the next tile's halo and the stores of the
// last one now really are in flight under the multiplies.  Only for 128 -> 64 channels (the dgrad of the 64 -> 128
// down-sampling conv, the ConvTranspose2d 128 -> 64).
constexpr int DH = TH + 1, DW = TW + 1;            // halo 9 x 17 positions of dy
constexpr int DPITCH = 18;
constexpr int DROWS = DH * DPITCH;                 // 162 LDS rows per 64-channel chunk image
constexpr int DCHUNK_B = DROWS * 128;              // 20736 B
constexpr int DROUNDS = (DROWS + 63) / 64;         // 3 DMA instructions per wave and chunk
constexpr int DBUF_B = 2 * DCHUNK_B;               // both chunks of a tile
constexpr int DLDS_B = 2 * DBUF_B;                 // two tiles: 82944 B

struct S2DKP {
    const char* g;          // 16-bit dy [B][Ho][Wo][cs], 128 channels
    const char* w;          // 16-bit [tap][N = 64][K = 128]
    const char* zeros;
    float* y;               // dx fp32 [B][2 Ho][2 Wo][y_cs] ...
    char* y16;              // ... or 16-bit
    const float* bias;
    int B, Ho, Wo, cs, y_cs, act;
    int TX, TY, tiles, lists;
};

// TYPE 0: taps (1,1) -> class (0,0) [slot 0]; (2,2), (2,0), (0,2), (0,0) -> class (1,1) [slot 1]
// TYPE 1: taps (1,2), (1,0) -> class (0,1) [slot 0]; (2,1), (0,1) -> class (1,0) [slot 1]
// per tap: weight index 3 kh + kw, accumulator slot, shift (dh, dw) = (kh == 0, kw == 0); the unshifted taps first (their
// fragments are the older ones of the ring)
template <int TYPE> struct S2DTaps;
template <> struct S2DTaps<0> {
    static constexpr int N = 5;
    static constexpr int tap[5] = {4, 8, 6, 2, 0}, slot[5] = {0, 1, 1, 1, 1}, dh[5] = {0, 0, 0, 1, 1}, dw[5] = {0, 0, 1, 0, 1};
    static constexpr int cls[2] = {0, 3};          // class 2 a + b of the two accumulator slots
};
template <> struct S2DTaps<1> {
    static constexpr int N = 4;
    static constexpr int tap[4] = {5, 3, 7, 1}, slot[4] = {0, 0, 1, 1}, dh[4] = {0, 0, 0, 1}, dw[4] = {0, 1, 0, 0};
    static constexpr int cls[2] = {1, 2};
};

template <bool H16, int TYPE>
__device__ __forceinline__ void conv_s2d_body(const S2DKP& p, char* smem, int lane, int wave, int t_begin, int t_end) {
    typedef S2DTaps<TYPE> T;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int nj = wave & 3;                        // column tile of dx
    const int n0 = nj * 16;

    bf16x8 wf[T::N][4];     // (tap, k32 step): rows n0 + l15 of [tap][64][128], K-values 32 s + 8 g4 .. + 7
#pragma unroll
    for (int t = 0; t < T::N; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            wf[t][k] = *reinterpret_cast<const bf16x8*>(p.w + ((size_t)(T::tap[t] * 64 + n0 + l15) * 128 + 32 * k + 8 * g4) * 2);
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = p.bias ? p.bias[n0 + 4 * g4 + r] : 0.f;

    const unsigned lds0 = mmh::lds_addr_of(smem);
    auto issue_tile = [&](int tile, int buf) {      // both 64-channel chunks of the tile's dy halo
        const int b = tile / (p.TX * p.TY);
        const int rem = tile - b * (p.TX * p.TY);
        const int ty = rem / p.TX, tx = rem - ty * p.TX;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            const unsigned dst = lds0 + (unsigned)buf * DBUF_B + (unsigned)kc * DCHUNK_B + (unsigned)wave * 1024u;
#pragma unroll
            for (int rd = 0; rd < DROUNDS; ++rd) {
                const int r = rd * 64 + wave * 8 + (lane >> 3);
                const int hy = r / DPITCH, sl = r - hy * DPITCH;
                const int ph = ty * TH + hy, qw = tx * TW + sl;
                const bool row = r < DROWS && sl < DW;
                const bool ok = row && ph < p.Ho && qw < p.Wo;
                const unsigned q8 = (unsigned)((lane & 7) ^ (sl & 6));
                if (row) {
                    const char* gsrc = ok ? p.g + ((size_t)((unsigned)((b * p.Ho + ph) * p.Wo + qw) * (unsigned)p.cs * 2u) + kc * 128 + q8 * 16u)
                                          : p.zeros + (lane & 7) * 16;
                    mmh::lds_dma16(gsrc, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + (unsigned)rd * 8192u)));
                }
            }
        }
    };
    // pixel fragment of halo row r (0..8), column shift dw, k32 step (kc, hf): LDS row r * DPITCH + dw + l15
    unsigned aL[2][2];
#pragma unroll
    for (int dw = 0; dw < 2; ++dw) {
        const unsigned sl = (unsigned)(dw + l15);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
            aL[dw][hf] = lds0 + sl * 128u + ((((unsigned)(4 * hf + g4)) ^ (sl & 6u)) << 4);
    }

    f32x4 acc[2][TH];       // [accumulator slot][row]
    int nt = 0;
    issue_tile(t_begin, 0);
    bool after_store = false;
    for (int tile = t_begin; tile < t_end; ++tile, ++nt) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < TH; ++i) acc[c][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (after_store) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");     // all but the 16 stores behind the halo DMA
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (tile + 1 < t_end) issue_tile(tile + 1, (nt + 1) & 1);
        const unsigned boff = (unsigned)(nt & 1) * DBUF_B;
        // Row slots L = 9 k + r (k32 step k, halo row r): 36 per tile, two fragments each (dw = 0, 1), in a ring of four.
        // Block (k, i) multiplies the slots 9 k + i (dh = 0) and 9 k + i + 1 (dh = 1) and first requests every slot up to
        // three ahead of its own (the ring slot it overwrites belongs to a block that is done).
        bf16x8 R[4][2];
        auto load_slot = [&](int L) {
            const int k = L / 9, r = L - 9 * k, kc = k >> 1, hf = k & 1;
            const unsigned a = boff + (unsigned)kc * DCHUNK_B + (unsigned)(r * DPITCH) * 128u;
            R[L & 3][0] = lds_frag(aL[0][hf] + a);
            if (TYPE == 0 || r != TH) R[L & 3][1] = lds_frag(aL[1][hf] + a);     // type 1 never shifts row 8 sideways
        };
        load_slot(0); load_slot(1); load_slot(2);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int i = 0; i < TH; ++i) {
                const int base = 9 * k + i;
                const int prev_hi = (k == 0 && i == 0) ? 2 : ((i == 0 ? 9 * (k - 1) + TH - 1 : base - 1) + 3 > 35 ? 35 : (i == 0 ? 9 * (k - 1) + TH - 1 : base - 1) + 3);
                const int hi = base + 3 > 35 ? 35 : base + 3;
                int nload = 0;
#pragma unroll
                for (int L = 0; L < 36; ++L)
                    if (L > prev_hi && L <= hi) { load_slot(L); nload += (TYPE == 0 || (L % 9) != TH) ? 2 : 1; }
#pragma unroll
                for (int t = 0; t < T::N; ++t)
                    acc[T::slot[t]][i] = mfma<H16>(wf[t][k], R[(base + T::dh[t]) & 3][T::dw[t]], acc[T::slot[t]][i]);
                if (nload == 4) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                else if (nload == 3) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                else if (nload == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                else if (nload == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, T::N, 0);
            }
        }
        // epilogue: slot c = class (a, b), row i -> dx pixel (2 (ty 8 + i) + a, 2 (tx 16 + l15) + b), channels n0 + 4 g4 .. + 3
        const int b = tile / (p.TX * p.TY);
        const int rem = tile - b * (p.TX * p.TY);
        const int ty = rem / p.TX, tx = rem - ty * p.TX;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < TH; ++i) {
                const int oh = 2 * (ty * TH + i) + (T::cls[c] >> 1), ow = 2 * (tx * TW + l15) + (T::cls[c] & 1);
                const size_t pix = ((size_t)b * (2 * p.Ho) + oh) * (2 * p.Wo) + ow;
                store4<H16>(p.y, p.y16, pix * p.y_cs + (n0 + 4 * g4), acc[c][i], bv, p.act);
            }
        after_store = true;
    }
}

template <bool H16>
__global__ void __launch_bounds__(512) conv_s2d_kernel(const S2DKP p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, lst = blockIdx.x >> 3;
    const int per_xcd = (p.tiles + 7) / 8, lists_x = p.lists / 8;
    const int per_list = (per_xcd + lists_x - 1) / lists_x;
    const int t_begin = xcd * per_xcd + lst * per_list;
    const int t_end = min(min(t_begin + per_list, (xcd + 1) * per_xcd), p.tiles);
    if (t_begin >= t_end) return;
    if (wave < 4) conv_s2d_body<H16, 0>(p, smem, lane, wave, t_begin, t_end);
    else conv_s2d_body<H16, 1>(p, smem, lane, wave, t_begin, t_end);
}

int g_cus = 0;

}  // namespace

namespace mmh {

// the stride-1 form: 64 -> 64, zero padding (VGG19 conv1_2 forward, mode 0, and its input gradient, mode 1)
bool conv_s1f_ok(const mmh_conv_desc* d, int mode) {
    if (!g_lp16_s2f || !d || (mode != 0 && mode != 1)) return false;
    if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->pad_mode != MMH_PAD_ZERO) return false;
    if (d->dtype != MMH_BF16 && d->dtype != MMH_FP16) return false;
    if (d->Cin != 64 || d->Cout != 64 || d->Ho != d->H || d->Wo != d->W || d->H % TH || d->W % TW) return false;
    if ((long long)d->B * d->H * d->W * std::max(d->x_cs, d->y_cs) >= (1ll << 31)) return false;
    return true;
}

bool conv_s2f_ok(const mmh_conv_desc* d, int mode) {
    if (!g_lp16_s2f || mode != 0 || !d) return false;
    if (d->kh != 3 || d->kw != 3 || d->stride != 2 || d->pad != 1 || d->pad_mode != MMH_PAD_ZERO) return false;
    if (d->dtype != MMH_BF16 && d->dtype != MMH_FP16) return false;
    // 128 input channels (128 -> 256): built and tested (mmh_set_option("lp16_s2f", 2)), but with one column tile per wave
    // every wave reads the whole halo from LDS and the input is staged twice (two 128-column halves): 139 us against the
    // general kernel's 97 at B = 32 (tools/bench_s2f.py) - not taken by default
    if (d->Cin != 64 && !(d->Cin == 128 && g_lp16_s2f == 2)) return false;
    if (d->Cout % 128 || d->H % 2 || d->W % 2 || d->Ho != d->H / 2 || d->Wo != d->W / 2) return false;
    if (d->Ho % TH || d->Wo % TW) return false;
    if ((long long)d->B * d->H * d->W * d->x_cs >= (1ll << 31) || (long long)d->B * d->Ho * d->Wo >= (1ll << 31)) return false;
    return true;
}

int conv_s2f_stats_chunks(const mmh_conv_desc* d) {        // partials per image, 0 = no statistics from this kernel
    if (!conv_s2f_ok(d, 0)) return 0;
    return (d->Ho / TH) * (d->Wo / TW) * (d->Cin == 64 ? 2 : 1);
}

bool conv_s2d_ok(const mmh_conv_desc* d, int mode) {
    if (!g_lp16_s2f || mode != 1 || !d) return false;
    if (d->kh != 3 || d->kw != 3 || d->stride != 2 || d->pad != 1 || d->pad_mode != MMH_PAD_ZERO) return false;
    if (d->dtype != MMH_BF16 && d->dtype != MMH_FP16) return false;
    if (d->Cin != 64 || d->Cout != 128) return false;
    if (d->H % 2 || d->W % 2 || d->Ho != d->H / 2 || d->Wo != d->W / 2 || d->Ho % TH || d->Wo % TW) return false;
    if ((long long)d->B * d->H * d->W >= (1ll << 31) || (long long)d->B * d->Ho * d->Wo * d->y_cs >= (1ll << 31)) return false;
    return true;
}

int launch_conv_s2d(const mmh_conv_desc* d, const void* g16, const void* w16, const void* bias, void* dx, int dx_is16,
                    int act, const void* zeros, hipStream_t st) {
    S2DKP p{};
    p.g = static_cast<const char*>(g16);
    p.w = static_cast<const char*>(w16);
    p.zeros = static_cast<const char*>(zeros);
    if (dx_is16) p.y16 = static_cast<char*>(dx); else p.y = static_cast<float*>(dx);
    p.bias = static_cast<const float*>(bias);
    p.B = d->B; p.Ho = d->Ho; p.Wo = d->Wo; p.cs = d->y_cs; p.y_cs = d->x_cs; p.act = act;
    p.TX = p.Wo / TW; p.TY = p.Ho / TH; p.tiles = p.B * p.TX * p.TY;
    if (!g_cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        g_cus = n;
    }
    const int per_xcd = (p.tiles + 7) / 8;
    p.lists = 8 * std::max(1, std::min(g_cus / 8, per_xcd));
    const dim3 grid(p.lists);
    const bool h16 = d->dtype == MMH_FP16;
    static bool ready = false;
    if (!ready) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_s2d_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, DLDS_B);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_s2d_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, DLDS_B);
        if (e != hipSuccess) return fail("hipFuncSetAttribute(conv_s2d): %s", hipGetErrorString(e));
        ready = true;
    }
    if (h16) hipLaunchKernelGGL(conv_s2d_kernel<true>, grid, dim3(512), DLDS_B, st, p);
    else hipLaunchKernelGGL(conv_s2d_kernel<false>, grid, dim3(512), DLDS_B, st, p);
    return check_launch("conv_s2d_kernel");
}

int launch_conv_s2f(const mmh_conv_desc* d, const void* x16, const void* w16, const void* bias, void* y, int y_is16,
                    int act, const void* zeros, hipStream_t st, float* stats, int mode) {
    S2KP p{};
    if (d->stride == 1) {       // 64 -> 64: fprop, or (mode 1) the input gradient = the same conv of dy with the taps mirrored
        p.x = static_cast<const char*>(x16);
        p.w = static_cast<const char*>(w16);
        p.zeros = static_cast<const char*>(zeros);
        if (y_is16) p.y16 = static_cast<char*>(y); else p.y = static_cast<float*>(y);
        p.bias = static_cast<const float*>(bias);
        p.B = d->B; p.H = d->H; p.W = d->W; p.cs = mode == 0 ? d->x_cs : d->y_cs; p.C = 64;
        p.Ho = d->H; p.Wo = d->W; p.N = 64; p.y_cs = mode == 0 ? d->y_cs : d->x_cs; p.act = act;
        p.TX = p.Wo / TW; p.TY = p.Ho / TH; p.tiles = p.B * p.TX * p.TY;
        p.nsplit = 1; p.rev = mode == 1; p.dbg = g_lp16_dbg;
        if (!g_cus) {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
                n = 256;
            g_cus = n;
        }
        const int per_xcd1 = (p.tiles + 7) / 8;
        p.lists = 8 * std::max(1, std::min(g_cus / 8, per_xcd1));      // one per CU (206 registers: one workgroup is resident)
        const dim3 grid1(p.lists);
        constexpr int lds1 = HaloGeom<1>::LDS_B;
        if (d->dtype == MMH_FP16) hipLaunchKernelGGL((conv_s2f_kernel<true, 1, 4, 2, false, 1, 64>), grid1, dim3(512), lds1, st, p);
        else hipLaunchKernelGGL((conv_s2f_kernel<false, 1, 4, 2, false, 1, 64>), grid1, dim3(512), lds1, st, p);
        return check_launch("conv_s2f_kernel<stride 1>");
    }
    p.x = static_cast<const char*>(x16);
    p.w = static_cast<const char*>(w16);
    p.zeros = static_cast<const char*>(zeros);
    if (y_is16) p.y16 = static_cast<char*>(y); else p.y = static_cast<float*>(y);
    p.bias = static_cast<const float*>(bias);
    p.B = d->B; p.H = d->H; p.W = d->W; p.cs = d->x_cs; p.C = d->Cin;
    p.Ho = d->Ho; p.Wo = d->Wo; p.N = d->Cout; p.y_cs = d->y_cs; p.act = act;
    p.TX = p.Wo / TW; p.TY = p.Ho / TH; p.tiles = p.B * p.TX * p.TY;
    p.nsplit = p.N / 128;
    p.dbg = g_lp16_dbg;
    p.stats = stats;
    if (!g_cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        g_cus = n;
    }
    // one workgroup per CU (148 KB of LDS each): lists = CUs / nsplit, a multiple of 8, no more than the tiles an XCD has
    const int per_xcd = (p.tiles + 7) / 8;
    int lists_x = std::max(1, std::min(g_cus / 8 / p.nsplit, per_xcd));
    p.lists = 8 * lists_x;
    const dim3 grid(p.lists * p.nsplit);
    const bool h16 = d->dtype == MMH_FP16;
#define MMH_S2F(H16, KC, MW, NJ) { if (stats) MMH_S2F_(H16, KC, MW, NJ, true) else MMH_S2F_(H16, KC, MW, NJ, false) }
#define MMH_S2F_(H16, KC, MW, NJ, ST)                                                                              \
    {                                                                                                              \
        auto kfn = conv_s2f_kernel<H16, KC, MW, NJ, ST>;                                                           \
        static bool ready = false;                                                                                 \
        if (!ready) {                                                                                              \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),                                 \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, HaloGeom<2>::LDS_B);    \
            if (e != hipSuccess) return fail("hipFuncSetAttribute(conv_s2f): %s", hipGetErrorString(e));           \
            ready = true;                                                                                          \
        }                                                                                                          \
        hipLaunchKernelGGL(kfn, grid, dim3(512), HaloGeom<2>::LDS_B, st, p);                                       \
    }
    if (p.C == 64) { if (h16) MMH_S2F(true, 1, 2, 2) else MMH_S2F(false, 1, 2, 2) }
    else { if (h16) MMH_S2F(true, 2, 1, 1) else MMH_S2F(false, 2, 1, 1) }
#undef MMH_S2F
#undef MMH_S2F_
    return check_launch("conv_s2f_kernel");
}

}  // namespace mmh
