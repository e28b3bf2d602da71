// 16-bit wgrad of the 3x3 / stride-1 / pad-1 stack with ALL NINE TAPS resident: gfx950, third generation.
//
// dw[tap][ci][co] = sum over output pixels p of x[p + off(tap)][ci] * dy[p][co]   (models/Generator.py:40-113:
// the weight gradient of every PATBlock / ResnetBlock convolution; 54 + 2 x 6 launches per training step).
//
// wgrad_lp16r_kernel (conv_lp16.hip) gives a workgroup ONE tap of a 256 x 256 (ci, co) tile: per 64-pixel k-step it
// stages 32 KiB of x and 32 KiB of dy for 8.4 MFLOP - 128 FLOP per staged byte, 32 B/clk/CU at the MFMA rate - and
// the nine taps of a tile re-stage the same dy and (shifted by one pixel) the same x: measured DMA-bound (its DMA
// stream alone takes 64 % of the kernel), FETCH + WRITE 1.83 x the algorithmic bytes.
//
// Here a workgroup owns (64 ci) x (128 co) x (9 taps) = 73,728 outputs (144 accumulator VGPRs per lane) and walks
// blocks of 4 x 16 output pixels.  Per block it stages, ONCE, the block's 6 x 18 input halo [halo pixel][64 ci] (the
// padding - reflect or zero - folded into the DMA's per-lane source address) and its dy tile [64 pixels][128 co]:
// 31 KiB for 9.4 MFLOP = 304 FLOP per staged byte, 13.5 B/clk/CU.  The nine taps read the SAME halo image at nine
// row offsets: tap (kh, kw) of the k16-step of block row kk contracts halo rows (kk + kh) * 20 + kw + 0..15.
//
// LDS images (both operands are k-major in HBM - the contraction index, the pixel, is the slow index of NHWC - so
// both MFMA operands are read TRANSPOSED with ds_read_b64_tr_b16; a transposed 4 x 16 block touches 4 consecutive
// k rows x 64 B):
//   x halo  [6 halo rows x pitch 20][128 B]: rows r and r + 2 would share banks -> the 64-byte halves of a row are
//           swapped when bit 1 of the row index is set.  The pitch is a multiple of 4, so that bit - and with it the
//           lane's swizzled address - depends on the tap only through kw: three address registers serve all nine
//           taps, everything else is an immediate offset;
//   dy tile [64][256 B]: the four rows of a block share banks -> 16-byte chunk index XOR (row & 3) << 2.
// As the LDS-DMA writes linearly (lane i -> base + 16 i) the swizzles are applied to the per-lane GLOBAL source
// address and again on the read address (MI355X guide: both sides or neither).
//
// Pipeline: ring of four 32-KiB stages; a stage is issued three blocks ahead and waited for (vmcnt(4): every wave
// issues exactly four loads per stage, also past the end of its range, where they read the zero page) one k16-step
// before its first use, at the one barrier per block.  Fragments: the three A fragments of the next filter row are
// requested while the current row's three MFMAs (32x32x16) run.  The LDS-DMA is issued as inline asm (mmh::lds_dma16:
// hipcc would otherwise drain the ring behind every issue).
//
// Split-K over block ranges (one round of workgroups, split-major work list so that the tiles of one split sit on
// one XCD and share its L2), fp32 slabs [split][tap][Cin][Cout], fixed-order reduction (lp16_slab_reduce_kernel).
#include <algorithm>
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_vp;

constexpr int TCI = 64, TCO = 128;          // tile: input channels x output channels (x 9 taps)
constexpr int BC = 16;                      // pixel block columns = one k16-step
constexpr int XROWB = TCI * 2;              // 128 B per halo pixel
constexpr int DROWB = TCO * 2;              // 256 B per dy pixel
constexpr int TSTAGE = 32768;               // x halo + dy tile of a block: 32 KiB at either stride
constexpr int RING = 4;
// Stride 1: blocks of 4 x 16 output pixels, halo 6 x 18 at pitch 20 (16 KiB: 16 DMA instructions), dy tile 16 KiB.
// Stride 2 (the downsampling convs and, with the roles of the tensors swapped, ConvTranspose2d: models/Generator.py:192-253;
// zero padding): blocks of 2 x 16 output pixels, halo 5 x 33 at pitch 36 (24 KiB: 24 instructions), dy tile 8 KiB.  Tap
// (kh, kw) of the k16-step of block row kk contracts halo row 2 kk + kh at the pixels 2 t + kw: every second pixel, so all
// k rows of a transposed read would start 256 bytes apart - in the same banks.  The halo image is therefore stored with
// its pixel columns PERMUTED, column c in slot c ^ ((c >> 2) & 1): for either parity of kw the eight k rows of a read then
// spread over both 128-byte bank halves, and with the 64-byte half swap on bit 1 of the slot (as at stride 1) over all
// four quarters.  The permutation leaves bits 2.. of the column alone, so the second read of a fragment (+ 4 k rows = + 8
// pixels) is + 8 slots, and it is applied - like the swizzles - to the DMA's per-lane source address and the read address.
template <int ST> struct Geo {
    static constexpr int BR = ST == 1 ? 4 : 2;              // block rows = k16-steps per block
    static constexpr int HP = ST == 1 ? 20 : 36;            // halo pitch, a multiple of 4
    static constexpr int HROWS = ST * (BR - 1) + 3;         // 6 | 5 halo rows
    static constexpr int HCOLS = ST * (BC - 1) + 3;         // 18 | 33 halo columns
    static constexpr int XJ = ST == 1 ? 2 : 3;              // x DMA instructions per wave and stage
    static constexpr int DJ = ST == 1 ? 2 : 1;              // dy DMA instructions per wave and stage
    static constexpr int XSTAGE = 8 * XJ * 1024;            // 16 | 24 KiB
    static constexpr int DSTAGE = BR * BC * DROWB;          // 16 | 8 KiB
    static_assert(XSTAGE + DSTAGE == TSTAGE && HROWS * HP <= 8 * XJ * 8 && XJ + DJ == 4, "stage layout");
};
__device__ __forceinline__ int slot_of_col(int c, int st) { return st == 1 ? c : (c ^ ((c >> 2) & 1)); }

struct LpWgradTP {
    const char* x;          // 16-bit activations [B][H][W][Cin], pixel stride x_cs elements
    const char* dy;         // 16-bit output gradients [B][Ho][Wo][Cout], pixel stride dy_cs
    const char* zeros;      // >= 256 zero bytes
    float* slab;            // [S][9][Cin][Cout]
    int B, H, W, Ho, Wo, Cin, Cout, x_cs, dy_cs;
    int reflect;
    int stagger;            // the two waves of a SIMD issue their DMA at different points of a block
    int TR, TC;             // blocks per image: rows, columns
    int nblk, bps;          // blocks in all, blocks per split
    int CT, NT, S, items;
};

template <bool H16>
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    if (H16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0,
                                                      0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// 8 consecutive k rows (k = 8 h .. 8 h + 7 of the k16-step) of one column per lane: two transposed reads 4 rows apart
template <int ROWBYTES>
__device__ __forceinline__ bf16x8 tr_frag(const char* a) {
    typedef s16x4 __attribute__((address_space(3))) * lds_p;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a + 4 * ROWBYTES));
    struct { s16x4 a, b; } both = {lo, hi};
    return __builtin_bit_cast(bf16x8, both);
}

template <bool H16, int ST>
__global__ void __launch_bounds__(512, 2) wgrad_lp16t_kernel(const LpWgradTP p) {
    typedef Geo<ST> G;
    constexpr int BR = G::BR, HP = G::HP, XSTAGE = G::XSTAGE, XJ = G::XJ, DJ = G::DJ;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wr = wave >> 2, wc = wave & 3;
    const bool stagger = p.stagger != 0;
    const int per_xcd = (p.items + 7) / 8;
    int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (item >= p.items) return;
    const int nt = item % p.NT; item /= p.NT;
    const int ct = item % p.CT;
    const int split = item / p.CT;
    const int blk0 = split * p.bps;
    const int blk1 = min(p.nblk, blk0 + p.bps);
    const int nsteps = blk1 - blk0;

    // ---- DMA roles.  x: instruction xi fills halo slots 8 xi + lane / 8 (wave w: xi = w, w + 8, ..); dy: instruction
    // dj fills tile rows 4 dj + lane / 16 (wave w: dj = DJ w ..).
    int x_hy[XJ], x_hx[XJ];
    unsigned x_coff[XJ];
    bool x_row[XJ];
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
        const int hr = 8 * (wave + 8 * j) + (lane >> 3);
        x_hy[j] = hr / HP;
        x_hx[j] = slot_of_col(hr - x_hy[j] * HP, ST);       // the halo column this slot holds (an involution)
        x_row[j] = hr < G::HROWS * HP && x_hx[j] < G::HCOLS;
        const unsigned lc = (unsigned)(lane & 7) ^ ((unsigned)((hr >> 1) & 1) << 2);
        x_coff[j] = (unsigned)(ct * TCI) * 2u + lc * 16u;
    }
    int d_py[DJ], d_px[DJ];
    unsigned d_coff[DJ];
#pragma unroll
    for (int j = 0; j < DJ; ++j) {
        const int dr = 4 * (DJ * wave + j) + (lane >> 4);
        d_py[j] = dr >> 4;
        d_px[j] = dr & 15;
        const unsigned lc = (unsigned)(lane & 15) ^ ((unsigned)(dr & 3) << 2);
        d_coff[j] = (unsigned)(nt * TCO) * 2u + lc * 16u;
    }
    // the block the next issue() stages: (image, block row, block column), advanced once per issue
    int nb = blk0;
    int nb_img = nb / (p.TR * p.TC);
    int nb_tr = (nb - nb_img * p.TR * p.TC) / p.TC;
    int nb_tc = nb - (nb_img * p.TR + nb_tr) * p.TC;
    int slot_next = 0;
    const unsigned lds0 = mmh::lds_addr_of(smem);
    const unsigned xdst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);                   // + 8 KiB per j
    const unsigned ddst = __builtin_amdgcn_readfirstlane(lds0 + XSTAGE + (unsigned)(DJ * wave) * 1024u);   // + 1 KiB per j
    auto issue = [&]() {
        const unsigned sbase = (unsigned)slot_next * TSTAGE;
        slot_next = (slot_next + 1) & (RING - 1);
        const bool live = nb < blk1;
        const int r0 = nb_tr * BR, c0 = nb_tc * BC;
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            int ih = ST * r0 + x_hy[j] - 1, iw = ST * c0 + x_hx[j] - 1;
            if (ST == 1 && p.reflect) {
                ih = ih < 0 ? -ih : ih;
                iw = iw < 0 ? -iw : iw;
                ih = ih >= p.H ? 2 * (p.H - 1) - ih : ih;
                iw = iw >= p.W ? 2 * (p.W - 1) - iw : iw;
            }
            const bool ok = live && x_row[j] && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            const char* g = ok ? p.x + (size_t)((nb_img * p.H + ih) * p.W + iw) * (size_t)(p.x_cs * 2) + x_coff[j]
                               : p.zeros + (lane & 7) * 16;
            mmh::lds_dma16(g, xdst + sbase + (unsigned)j * 8192u);
        }
#pragma unroll
        for (int j = 0; j < DJ; ++j) {
            const int oh = r0 + d_py[j], ow = c0 + d_px[j];
            const bool ok = live && oh < p.Ho && ow < p.Wo;
            const char* g = ok ? p.dy + (size_t)((nb_img * p.Ho + oh) * p.Wo + ow) * (size_t)(p.dy_cs * 2) + d_coff[j]
                               : p.zeros + (lane & 15) * 16;
            mmh::lds_dma16(g, ddst + sbase + (unsigned)j * 1024u);
        }
        ++nb;
        if (++nb_tc == p.TC) {
            nb_tc = 0;
            if (++nb_tr == p.TR) { nb_tr = 0; ++nb_img; }
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ---- fragment addresses.  Lane: k row t = 8 h + q of the k16-step (second read: + 4), columns 16 G1 + 4 p2 .. + 3 of
    // the wave's 32 (ci for A, co for B)
    const int G1 = (lane >> 4) & 1, q = (lane & 15) >> 2, p2 = lane & 3;
    const int tk = 8 * h + q;
    unsigned a_base[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int row = slot_of_col(ST * tk + kw, ST);              // + (ST kk + kh) * HP rows: an immediate
        const int col = wr * 32 + 16 * G1 + 4 * p2;                 // element of the 64-channel halo row
        const unsigned chunk = (unsigned)(col >> 3) ^ ((unsigned)((row >> 1) & 1) << 2);
        a_base[kw] = (unsigned)row * XROWB + (chunk << 4) + (unsigned)(col & 4) * 2u;
    }
    unsigned b_base;
    {
        const int col = wc * 32 + 16 * G1 + 4 * p2;                 // element of the 128-channel dy row
        const unsigned chunk = (unsigned)(col >> 3) ^ ((unsigned)q << 2);
        b_base = (unsigned)tk * DROWB + (chunk << 4) + (unsigned)(col & 4) * 2u;
    }

    if (nsteps > 0) {
        issue(); issue(); issue();
        __builtin_amdgcn_s_waitcnt(0x0070 | 8);         // vmcnt(8): stage 0 has landed (stages 1, 2 may be in flight)
        __syncthreads();

        bf16x8 af[2][3], bfr[2];
        {
            const char* sX = smem;
            bfr[0] = tr_frag<DROWB>(sX + XSTAGE + b_base);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) af[0][kw] = tr_frag<ST * XROWB>(sX + a_base[kw]);
        }
        int slot = 0;
        for (int s = 0; s < nsteps; ++s) {
            const char* sX = smem + slot * TSTAGE;
            const int slot_n = (slot + 1) & (RING - 1);
            const char* sXn = smem + slot_n * TSTAGE;
#pragma unroll
            for (int kk = 0; kk < BR; ++kk) {
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int g = kk * 3 + kh;                      // group of this step, 0..11
                    const int cur = g & 1, nxt = cur ^ 1;
                    if (kk == BR - 1 && kh == 0) {
                        // stage s+1 (issued two blocks ago) must have landed before the last filter row of this
                        // block prefetches from it; every wave is past block s-1, whose slot stage s+3 takes
                        __builtin_amdgcn_s_waitcnt(0x0070 | 4);     // vmcnt(4) lgkmcnt(0)
                        __syncthreads();
                        // the two waves of a SIMD (wr = 0 / 1) issue the stage three blocks ahead at different points of the
                        // last k16-step (mmh_set_option("lp16_wgrad_ring", 3): both here): a DMA instruction costs its wave
                        // 60-180 cycles of issue time
                        if (wr == 0 || !stagger) issue();
                    }
                    if (kk == BR - 1 && kh == 2 && wr != 0 && stagger) issue();
                    // request the next group's fragments
                    if (kh < 2) {
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw)
                            af[nxt][kw] = tr_frag<ST * XROWB>(sX + a_base[kw] + (ST * kk + kh + 1) * (HP * XROWB));
                    } else if (kk < BR - 1) {
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw)
                            af[nxt][kw] = tr_frag<ST * XROWB>(sX + a_base[kw] + ST * (kk + 1) * (HP * XROWB));
                        bfr[(kk + 1) & 1] = tr_frag<DROWB>(sX + XSTAGE + b_base + (kk + 1) * (16 * DROWB));
                    } else if (s + 1 < nsteps) {
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) af[nxt][kw] = tr_frag<ST * XROWB>(sXn + a_base[kw]);
                        bfr[0] = tr_frag<DROWB>(sXn + XSTAGE + b_base);
                    }
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
                        acc[kh * 3 + kw] = mfma32<H16>(af[cur][kw], bfr[kk & 1], acc[kh * 3 + kw]);
                }
            }
            slot = slot_n;
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0070);     // drain the stages issued past the range before the LDS is released

    float* slab = p.slab + (size_t)split * 9 * p.Cin * p.Cout;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ct * TCI + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int co = nt * TCO + wc * 32 + l31;
            slab[((size_t)t * p.Cin + ci) * p.Cout + co] = acc[t][r];
        }
}

}  // namespace

namespace mmh {

int wgrad_lp16t_splits(const mmh_conv_desc* d) {
    const int tiles = (d->Cin / TCI) * (d->Cout / TCO);
    const int br = d->stride == 2 ? Geo<2>::BR : Geo<1>::BR;
    const long long nblk = (long long)d->B * ((d->Ho + br - 1) / br) * ((d->Wo + BC - 1) / BC);
    int S = std::max(1, 256 / tiles);                   // one round of workgroups, one per CU
    S = (int)std::min<long long>(S, std::max<long long>(1, nblk / 4));
    const long long bps = (nblk + S - 1) / S;
    return (int)((nblk + bps - 1) / bps);
}

// stride 1 ('same' output, zero or reflect padding) or stride 2 (even H, W, zero padding: Ho = H / 2)
bool wgrad_lp16t_supported(const mmh_conv_desc* d) {
    if (!d || d->kh != 3 || d->kw != 3 || d->pad != 1 || d->Cin % TCI || d->Cout % TCO) return false;
    if (d->dtype != MMH_BF16 && d->dtype != MMH_FP16) return false;
    if (d->stride == 1)
        return d->Ho == d->H && d->Wo == d->W && (d->pad_mode != MMH_PAD_REFLECT || (d->H >= 2 && d->W >= 2));
    return d->stride == 2 && d->pad_mode != MMH_PAD_REFLECT && d->H % 2 == 0 && d->W % 2 == 0 && d->Ho == d->H / 2 &&
           d->Wo == d->W / 2;
}

// slab: [S][9][Cin][Cout] fp32 with S = wgrad_lp16t_splits(d)
int launch_wgrad_lp16t(const mmh_conv_desc* d, const void* x16, const void* dy16, float* slab, const void* zeros,
                       hipStream_t st) {
    LpWgradTP p{};
    p.x = static_cast<const char*>(x16); p.dy = static_cast<const char*>(dy16);
    p.zeros = static_cast<const char*>(zeros);
    p.slab = slab;
    p.B = d->B; p.H = d->H; p.W = d->W; p.Ho = d->Ho; p.Wo = d->Wo;
    p.Cin = d->Cin; p.Cout = d->Cout; p.x_cs = d->x_cs; p.dy_cs = d->y_cs;
    p.reflect = d->pad_mode == MMH_PAD_REFLECT ? 1 : 0;
    p.stagger = g_lp16_wgrad_ring != 3;
    const int br = d->stride == 2 ? Geo<2>::BR : Geo<1>::BR;
    p.TR = (d->Ho + br - 1) / br; p.TC = (d->Wo + BC - 1) / BC;
    p.nblk = d->B * p.TR * p.TC;
    p.S = wgrad_lp16t_splits(d);
    p.bps = (p.nblk + p.S - 1) / p.S;
    p.CT = d->Cin / TCI; p.NT = d->Cout / TCO;
    p.items = p.S * p.CT * p.NT;
    constexpr int lds = RING * TSTAGE;
    static int ready = -1;
    if (ready != 0) {
        hipError_t e = hipSuccess;
        const void* fs[] = {reinterpret_cast<const void*>(wgrad_lp16t_kernel<false, 1>),
                            reinterpret_cast<const void*>(wgrad_lp16t_kernel<true, 1>),
                            reinterpret_cast<const void*>(wgrad_lp16t_kernel<false, 2>),
                            reinterpret_cast<const void*>(wgrad_lp16t_kernel<true, 2>)};
        for (const void* f : fs)
            if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        ready = e == hipSuccess ? 0 : fail("wgrad_lp16t_kernel: %s", hipGetErrorString(e));
    }
    if (ready != 0) return ready;
    const int per_xcd = (p.items + 7) / 8;
    const bool h16 = d->dtype == MMH_FP16;
    if (d->stride == 2) {
        if (h16) hipLaunchKernelGGL((wgrad_lp16t_kernel<true, 2>), dim3(8 * per_xcd), dim3(512), lds, st, p);
        else hipLaunchKernelGGL((wgrad_lp16t_kernel<false, 2>), dim3(8 * per_xcd), dim3(512), lds, st, p);
    } else {
        if (h16) hipLaunchKernelGGL((wgrad_lp16t_kernel<true, 1>), dim3(8 * per_xcd), dim3(512), lds, st, p);
        else hipLaunchKernelGGL((wgrad_lp16t_kernel<false, 1>), dim3(8 * per_xcd), dim3(512), lds, st, p);
    }
    return check_launch("wgrad_lp16t_kernel");
}

}  // namespace mmh
