// Probe, second build of the six-bf16-product fp32 GEMM (first build: split6_fast.hip, 1.47x the native GEMM and
// DMA-bound by construction: 72 KiB of LDS-DMA per 12.6 MFLOP because two 32-deep stages of a 256 x 128 block fill the LDS).
//
// What changed: the contraction is staged 16 deep and multiplied with MFMA 32x32x16 - a stage of a 256 x 256 block is then
// 3 terms x 512 rows x 32 B = 48 KiB, THREE of them fit (144 KiB), and a stage carries the same 12.6 MFLOP: 48 KiB instead of
// 72 per 12.6 MFLOP, a third less DMA, and with three stages one is always in flight (issued at the top of step g for step
// g + 2, awaited in the middle of step g + 1).
//
//   C[p] = A[p] (M x K) . B[p] (K x N),  x = x0 + x1 + x2,  a.b ~ a2b0 + a0b2 + a1b1 + a1b0 + a0b1 + a0b0 chained from C = 0 per
//   16-deep k-step (small terms first) and folded into the fp32 total by one add.
//   A3 [3][P][K/16][M][16] bf16 (term-major, k-blocked by 16), B3 [3][P][K/16][N][16] bf16, C [P][M][N] fp32.
//
// 512 threads = 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = 4 x 2 MFMA 32x32x16 tiles (128 accumulator registers); the
// weight-side fragment is the FIRST operand, so a lane ends up with 4 consecutive n of one m = one 16-byte store.  LDS image of
// a 32-row group of one term: [k half][row][16 B] - the MFMA's lane order (lane = 32 (k half) + row), so a fragment read is
// `base + 16 lane`, 1 KiB contiguous, conflict-free whatever the lane grouping; the LDS-DMA writes linearly, so the same
// permutation sits in its per-lane global source address (lane L fetches row L % 32, k half L / 32 of a 1 KiB contiguous run).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const bf16x8 __attribute__((address_space(3))) * lds_frag_p;

namespace {

constexpr int BM = 256, BN = 256, BK = 16;
constexpr int A_T = BM * BK * 2, B_T = BN * BK * 2;     // one term of one stage: 8 KiB each
constexpr int STAGE = 3 * (A_T + B_T);                  // 49152
constexpr int B_OFF = 3 * A_T;
constexpr int NSTAGE = 3;

struct SplitKP {
    const char* A3;     // [3][P][K/16][M][16] bf16
    const char* B3;     // [3][P][K/16][N][16] bf16
    float* C;           // [P][M][N]
    long long M;
    int K, N, P;
    int MT, NT;
    int dbg;            // timing-only: 1 no DMA after the prologue, 2 no fragment reads, 4 no MFMAs, 8 no stores
};

__device__ __forceinline__ bf16x8 lds_frag(unsigned a) { return *reinterpret_cast<lds_frag_p>(a); }
// LDS-DMA as inline asm (the kernel orders it against the reads itself: counted vmcnt + barrier; through the builtin the compiler
// is free to wait for everything outstanding in front of a later LDS read).  M0 = wave-uniform LDS base.
__device__ __forceinline__ void lds_dma16(const void* g, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_base) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p; }

__global__ void __launch_bounds__(512, 2) split6_gemm2_kernel(const SplitKP p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;
    // work list as in the first build: XCD x owns the planes x, x + 8, ..; its (plane, row tile, column tile) items, column tile
    // fastest, dealt round-robin to its workgroups (neighbouring items of ONE plane at any time: panels shared through L2)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wpx = gridDim.x >> 3;
    const int planes_x = (p.P - xcd + 7) / 8;
    const int per_plane = p.MT * p.NT;
    const int items_x = planes_x * per_plane;
    if (slot >= items_x) return;
    const int my_items = (items_x - slot + wpx - 1) / wpx;
    const int KS = p.K / BK;
    const int nsteps = my_items * KS;
    const size_t a_term = (size_t)p.P * p.M * p.K * 2, b_term = (size_t)p.P * p.N * p.K * 2;

    // DMA roles: instruction q (0..47) of a stage, wave w issues q = w + 8 r: term q / 16; (q % 16) < 8: A rows (q % 8) * 32 ..,
    // else B rows.  Lane L: row L % 32 of the 32, k half L / 32.
    const unsigned dsrc = (unsigned)(l31 * 32 + h * 16);
    const unsigned lds0 = lds_addr_of(smem);
    auto issue_stage = [&](int g) {
        const int item = slot + wpx * (g / KS), ks = g - (g / KS) * KS;
        const int pl = xcd + 8 * (item / per_plane), rem = item % per_plane;
        const int mt = rem / p.NT, nt = rem - mt * p.NT;
        const unsigned sbase = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(g % NSTAGE) * STAGE);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int q = wave + 8 * r;
            const int t = q >> 4, sub = q & 15;
            if (sub < 8) {
                const int rb = sub * 32;
                long long m = (long long)mt * BM + rb + l31;
                m = m < p.M ? m : p.M - 1;          // ragged last row tile: re-read the last row (never stored)
                const char* src = p.A3 + t * a_term + ((((size_t)pl * KS + ks) * p.M + m) * BK) * 2 + h * 16;
                lds_dma16(src, sbase + (unsigned)(t * A_T + rb * 32));
            } else {
                const int rb = (sub - 8) * 32;
                const char* src = p.B3 + t * b_term + ((((size_t)pl * KS + ks) * p.N + nt * BN + rb) * BK) * 2 + dsrc;
                lds_dma16(src, sbase + (unsigned)(B_OFF + t * B_T + rb * 32));
            }
        }
    };

    // fragment addresses: 32-row group (wm * 4 + i) of A / (wn * 2 + j) of B, + 16 lane; term: immediate
    const unsigned fa = lds0 + (unsigned)(wm * 4 * 1024) + (unsigned)lane * 16u;
    const unsigned fb = lds0 + B_OFF + (unsigned)(wn * 2 * 1024) + (unsigned)lane * 16u;

    f32x16 acc[4][2];
    bf16x8 af[2][3], bfr[2][3];            // A: double-buffered over i; B: refilled in place behind its last use of a step
    auto load_a = [&](int buf, int i, unsigned st) {
        if (p.dbg & 2) return;
#pragma unroll
        for (int t = 0; t < 3; ++t) af[buf][t] = lds_frag(fa + st + (unsigned)(t * A_T + i * 1024));
    };
    auto load_b = [&](int j, unsigned st) {
        if (p.dbg & 2) return;
#pragma unroll
        for (int t = 0; t < 3; ++t) bfr[j][t] = lds_frag(fb + st + (unsigned)(t * B_T + j * 1024));
    };
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            af[b][t] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
            bfr[b][t] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }

    // the six products of one output tile, chained from zero, small terms first (a2b0, a0b2, a1b1, a1b0, a0b1, a0b0): D[n][m].
    // refill: the B fragments of column group j are requested for the NEXT step right behind their last multiply of this one
    // (group 0's latency runs under group 1's chain, group 1's under the next step's first chain, which needs only group 0)
    auto six2 = [&](int i, int ab, bool refill, unsigned stn) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (!(p.dbg & 4)) {
                f32x16 t;
#pragma unroll
                for (int e = 0; e < 16; ++e) t[e] = 0.f;
                t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j][0], af[ab][2], t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j][2], af[ab][0], t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j][1], af[ab][1], t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j][0], af[ab][1], t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j][1], af[ab][0], t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j][0], af[ab][0], t, 0, 0, 0);
                acc[i][j] += t;
            }
            if (refill) load_b(j, stn);
        }
    };

    issue_stage(0);
    if (nsteps > 1) issue_stage(1);
    if (nsteps > 1) __builtin_amdgcn_s_waitcnt(0x0070 | 6);        // stage 0 landed (stage 1's six may fly)
    else __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();
    load_b(0, 0); load_b(1, 0);
    load_a(0, 0, 0);

    int item = slot, ks = 0;
    for (int g = 0; g < nsteps; ++g) {
        const unsigned st = (unsigned)(g % NSTAGE) * STAGE, stn = (unsigned)((g + 1) % NSTAGE) * STAGE;
        if (ks == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        }
        // ---- top: everybody is past step g - 1 (whose slot stage g + 2 takes)
        if (g > 0) __syncthreads();
        if (g + 2 < nsteps && !(p.dbg & 1)) issue_stage(g + 2);
        load_a(1, 1, st);
        six2(0, 0, false, 0);
        load_a(0, 2, st);
        six2(1, 1, false, 0);
        // ---- middle: stage g + 1 (issued at the top of step g - 1) must have landed before the next step's operands are
        // requested from it; stage g + 2's six instructions may still fly
        if (g + 2 < nsteps && !(p.dbg & 1)) __builtin_amdgcn_s_waitcnt(0x0070 | 6);
        else __builtin_amdgcn_s_waitcnt(0x0070);
        __syncthreads();
        load_a(1, 3, st);
        six2(2, 0, false, 0);
        const bool more = g + 1 < nsteps;
        if (more) load_a(0, 0, stn);
        six2(3, 1, more, stn);
        if (++ks == KS) {       // the item is complete: store its 128 x 64 wave tile, 16 bytes per lane and 4-row group
            const int pl = xcd + 8 * (item / per_plane), rem = item % per_plane;
            const int mt = rem / p.NT, nt = rem - mt * p.NT;
            if (!(p.dbg & 8)) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const long long m = (long long)mt * BM + wm * 128 + i * 32 + l31;
                    if (m < p.M) {
                        float* crow = p.C + ((size_t)pl * p.M + m) * p.N + nt * BN + wn * 64 + 4 * h;
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int rg = 0; rg < 4; ++rg)
                                *reinterpret_cast<f32x4*>(crow + j * 32 + rg * 8) =
                                    (f32x4){acc[i][j][4 * rg], acc[i][j][4 * rg + 1], acc[i][j][4 * rg + 2], acc[i][j][4 * rg + 3]};
                    }
                }
            }
            ks = 0; item += wpx;
        }
    }
}

}  // namespace

extern "C" int split6_gemm_fast2(const void* A3, const void* B3, float* C, long long M, int K, int N, int P, int dbg, hipStream_t st) {
    if (K % BK || N % BN || P < 1 || M < 1) return 1;
    static bool ready = false;
    if (!ready) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(split6_gemm2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * STAGE) != hipSuccess) return 2;
        ready = true;
    }
    SplitKP p{};
    p.A3 = static_cast<const char*>(A3); p.B3 = static_cast<const char*>(B3); p.C = C;
    p.M = M; p.K = K; p.N = N; p.P = P;
    p.MT = (int)((M + BM - 1) / BM); p.NT = N / BN; p.dbg = dbg;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipLaunchKernelGGL(split6_gemm2_kernel, dim3(8 * (cus / 8)), dim3(512), NSTAGE * STAGE, st, p);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
