// fp32 wgrad of the 3x3 / stride-2 / zero-pad-1 downsampling convs (models/Generator.py:192-199: nn.Conv2d(ngf m, 2 ngf m, 3, 2, 1)
// of the three streams, 64 -> 128 @256x256 and 128 -> 256 @128x128; Cin % 64 == 0, Cout % 128 == 0) as a STREAM down a
// column strip of dy, with every accumulator of the strip's filter block in registers.
//
// The generic route (conv_wgrad_kernel) is an implicit GEMM [9 Cin x pixels] . [pixels x Cout] in 128-row tiles: each row
// tile gathers its own im2col panel of x (every x pixel 2.25 times), stages it through registers into LDS, and splits the
// pixel range over slabs: 0.59 - 0.63 of the fp32 MFMA peak.  Here a work-group owns a strip of 16 dy positions x a range
// of dy rows x (64 input channels x 128 output channels) and walks down the rows.  dy row ph needs x rows 2 ph - 1, 2 ph,
// 2 ph + 1 (33 pixels each): per row ONE ring entry arrives by LDS-DMA - x rows 2 ph and 2 ph + 1 and dy row ph, 24.5 KiB,
// a plain copy of the global layout - two rows ahead of its use.  512 threads = 8 waves = 2 (input-channel tiles) x 4
// (output-channel tiles); a wave holds dw[9 taps][32 ci][32 co] = nine 32x32 accumulator tiles (144 registers) for the whole
// strip.  The contraction runs over positions, two per v_mfma_f32_32x32x2_f32: the x operand of lane (ci, k) is ONE float
// at [pixel 2 (2 ks + k) + kw][ci] - lanes along ci read 128 contiguous bytes, no swizzle, no transposed staging - and
// one dy read serves the nine taps: 10 ds_read_b32 (immediate offsets) and 9 independent MFMAs per k-step, 8 k-steps and
// one barrier per row.  Each work-group writes its block of dw once, into its slab; a fixed-order reduction sums the slabs
// (slab_reduce.hip).
#include <algorithm>
#include "common.h"

namespace mmh { int g_wgrad_s2_strip = 1; }

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const float __attribute__((address_space(3))) * lds_f_p;
__device__ __forceinline__ float lds_f(unsigned addr, int imm) { return *reinterpret_cast<lds_f_p>((size_t)(addr + (unsigned)imm)); }

constexpr int NT = 512;
constexpr int SW = 16;                          // dy positions per strip row
constexpr int XPX = 2 * SW + 1;                 // 33 x pixels per row
constexpr int CIB = 64, COB = 128;              // channels per work-group
constexpr int XROW_B = XPX * CIB * 4;           // 8448
constexpr int ROUNDS = 4;                       // DMA instructions per thread and entry
constexpr int ENT_B = ROUNDS * NT * 16;         // 32768: x row 2 ph | x row 2 ph + 1 | dy row ph | unused tail
constexpr int NENT = 4;                         // ring: rows ph - 1, ph in use, ph + 1, ph + 2 in flight
constexpr int LDS_B = NENT * ENT_B;             // 131072
constexpr int U_XA = XPX * (CIB / 4), U_XB = 2 * U_XA, U_DY = U_XB + SW * (COB / 4);      // 528, 1056, 1568 DMA units

__device__ char g_zero_line[128];               // DMA source of the zero padding

struct WgradS2KP {
    const float* x;         // [B][H][W][x_cs]
    const float* dy;        // [B][Ho][Wo][dy_cs]
    float* slab;            // [Z][9][Cin][Cout]
    int B, H, W, Ho, Wo, x_cs, dy_cs, Cin, Cout;
    int strips, rowsplit, rows_per, cichunks, cochunks, blocks;
};

__global__ void __launch_bounds__(NT, 1) wgrad_s2_kernel(const WgradS2KP p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, k = lane >> 5;
    const int ct = wave >> 2, nt = wave & 3;
    const unsigned lds0 = mmh::lds_addr_of(smem);
    const unsigned wdst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);

    // XCD x works on a contiguous range of units; the channel blocks of one (image, strip, rows) unit are neighbours
    const int L = (p.blocks & 7) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (p.blocks >> 3) + (blockIdx.x >> 3));
    const int chunks = p.cichunks * p.cochunks;
    const int z = L / chunks, cc = L - z * chunks;
    const int ci0 = (cc / p.cochunks) * CIB, co0 = (cc % p.cochunks) * COB;
    const int rs = z % p.rowsplit, t = z / p.rowsplit;
    const int strip = t % p.strips, b = t / p.strips;
    const int pw0 = strip * SW;
    const int r0 = rs * p.rows_per, r1 = min(p.Ho, r0 + p.rows_per);

    // DMA roles: unit u = round * 512 + tid of an entry: x row A | x row B | dy row | idle (reads the zero line)
    unsigned d_off[ROUNDS];         // float offset from the row base of its tensor
    int d_kind[ROUNDS];             // 0 x (row A or B), 1 dy, 2 idle / out of range in w
#pragma unroll
    for (int rr = 0; rr < ROUNDS; ++rr) {
        const int u = rr * NT + tid;
        if (u < U_XB) {
            const int v = u < U_XA ? u : u - U_XA;
            const int px = v >> 4, chk = v & 15;
            const int iw = 2 * pw0 - 1 + px;
            d_kind[rr] = (iw >= 0 && iw < p.W) ? 0 : 2;
            d_off[rr] = (unsigned)((iw + (u < U_XA ? 0 : p.W)) * p.x_cs + ci0 + 4 * chk);
        } else if (u < U_DY) {
            const int v = u - U_XB;
            const int pos = v >> 5, chk = v & 31;
            d_kind[rr] = pw0 + pos < p.Wo ? 1 : 2;
            d_off[rr] = (unsigned)((pw0 + pos) * p.dy_cs + co0 + 4 * chk);
        } else {
            d_kind[rr] = 2;
            d_off[rr] = 0;
        }
    }
    const void* const zero = g_zero_line + (lane & 7) * 16;
    auto issue_entry = [&](int ph) {
        const bool okx = ph >= 0 && ph < r1;            // x rows 2 ph, 2 ph + 1 exist (ph < Ho) and are used (ph < r1)
        const bool okd = ph >= r0 && ph < r1;
        const float* xb = p.x + ((size_t)(b * p.H + 2 * ph) * p.W) * (size_t)p.x_cs;       // uniform
        const float* db = p.dy + ((size_t)(b * p.Ho + ph) * p.Wo) * (size_t)p.dy_cs;
        const unsigned dst = wdst + (unsigned)((ph & (NENT - 1)) * ENT_B);
#pragma unroll
        for (int rr = 0; rr < ROUNDS; ++rr) {
            const void* g = zero;
            if (d_kind[rr] == 0 && okx) g = xb + d_off[rr];
            if (d_kind[rr] == 1 && okd) g = db + d_off[rr];
            mmh::lds_dma16(g, dst + (unsigned)(rr * NT * 16));
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[tp][i] = 0.f;

    const unsigned lane_a = (unsigned)(k * 512 + (32 * ct + m) * 4);        // position 2 ks + k -> pixel 2 (2 ks + k) + kw
    const unsigned lane_b = (unsigned)(2 * XROW_B + k * 512 + (32 * nt + m) * 4);
    if (r0 < r1) {
        issue_entry(r0 - 1);
        issue_entry(r0);
        issue_entry(r0 + 1);
        for (int ph = r0; ph < r1; ++ph) {
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0x0070 | ROUNDS);        // vmcnt(4): entries ph - 1 and ph landed, ph + 1 may fly
            __builtin_amdgcn_s_barrier();                       // ... everybody's; entry ph - 2 is no longer read
            asm volatile("" ::: "memory");
            issue_entry(ph + 2);
            const unsigned e_cur = lds0 + (unsigned)((ph & (NENT - 1)) * ENT_B);
            const unsigned e_prev = lds0 + (unsigned)(((ph - 1) & (NENT - 1)) * ENT_B);
            unsigned a0 = e_prev + XROW_B + lane_a, a1 = e_cur + lane_a, a2 = e_cur + XROW_B + lane_a, bb = e_cur + lane_b;
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(bb));
#pragma unroll
            for (int ks = 0; ks < SW / 2; ++ks) {
                const float bv = lds_f(bb, ks * 1024);
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float x0 = lds_f(a0, (4 * ks + kw) * 256), x1 = lds_f(a1, (4 * ks + kw) * 256),
                                x2 = lds_f(a2, (4 * ks + kw) * 256);
                    acc[kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, bv, acc[kw], 0, 0, 0);
                    acc[3 + kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, bv, acc[3 + kw], 0, 0, 0);
                    acc[6 + kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(x2, bv, acc[6 + kw], 0, 0, 0);
                }
            }
        }
    }
    // acc[tap][i]: column co0 + 32 nt + m, row ci0 + 32 ct + (i & 3) + 8 (i >> 2) + 4 k
    float* const sl = p.slab + (size_t)z * 9 * p.Cin * p.Cout + (size_t)(ci0 + 32 * ct + 4 * k) * p.Cout + co0 + 32 * nt + m;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int i = 0; i < 16; ++i)
            sl[(size_t)(tp * p.Cin + (i & 3) + 8 * (i >> 2)) * p.Cout] = acc[tp][i];
}

struct Plan { int strips, rowsplit, rows_per, Z; };
Plan plan(const mmh_conv_desc* d) {
    Plan pl;
    pl.strips = (d->Wo + SW - 1) / SW;
    const int chunks = (d->Cin / CIB) * (d->Cout / COB);
    // split the rows until about one work-group per CU exists (a work-group streams at least 8 rows)
    pl.rowsplit = 1;
    while (d->B * pl.strips * pl.rowsplit * chunks < 256 && d->Ho / (pl.rowsplit * 2) >= 8) pl.rowsplit *= 2;
    pl.rows_per = (d->Ho + pl.rowsplit - 1) / pl.rowsplit;
    pl.Z = d->B * pl.strips * pl.rowsplit;
    return pl;
}

}  // namespace

namespace mmh {

bool wgrad_s2_strip_ok(const mmh_conv_desc* d) {
    return g_wgrad_s2_strip && d->dtype == MMH_F32 && d->kh == 3 && d->kw == 3 && d->stride == 2 && d->pad == 1 &&
           d->pad_mode == MMH_PAD_ZERO && d->Cin % CIB == 0 && d->Cout % COB == 0 && d->H == 2 * d->Ho && d->W == 2 * d->Wo &&
           d->x_cs % 4 == 0 && d->y_cs % 4 == 0 && (size_t)d->B * d->H * d->W * d->x_cs < (1ull << 31) &&
           (size_t)d->B * d->Ho * d->Wo * d->y_cs < (1ull << 31);
}

size_t wgrad_s2_strip_ws_bytes(const mmh_conv_desc* d) {
    return (size_t)plan(d).Z * 9 * d->Cin * d->Cout * sizeof(float);
}

int launch_wgrad_s2_strip(const mmh_conv_desc* d, const void* x, const void* dy, void* dw, void* ws, size_t ws_bytes,
                          int accumulate, hipStream_t st) {
    const Plan pl = plan(d);
    MMH_REQUIRE(ws_bytes >= wgrad_s2_strip_ws_bytes(d), "wgrad_s2: workspace too small");
    WgradS2KP p{};
    p.x = static_cast<const float*>(x);
    p.dy = static_cast<const float*>(dy);
    p.slab = static_cast<float*>(ws);
    p.B = d->B; p.H = d->H; p.W = d->W; p.Ho = d->Ho; p.Wo = d->Wo; p.x_cs = d->x_cs; p.dy_cs = d->y_cs;
    p.Cin = d->Cin; p.Cout = d->Cout;
    p.strips = pl.strips; p.rowsplit = pl.rowsplit; p.rows_per = pl.rows_per;
    p.cichunks = d->Cin / CIB; p.cochunks = d->Cout / COB;
    p.blocks = pl.Z * p.cichunks * p.cochunks;
    static int ready = -1;
    if (ready != 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_s2_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
        if (e != hipSuccess) return fail("wgrad_s2: %s", hipGetErrorString(e));
        ready = 0;
    }
    hipLaunchKernelGGL(wgrad_s2_kernel, dim3(p.blocks), dim3(NT), LDS_B, st, p);
    if (int rc = check_launch("wgrad_s2_kernel")) return rc;
    const int64_t n4 = (int64_t)9 * d->Cin * d->Cout / 4;
    return launch_slab_reduce(p.slab, static_cast<float*>(dw), n4, pl.Z, accumulate, n4, st);
}

}  // namespace mmh
