// Probe, the SPEED half of VERDICT r4 #4: the Winograd-domain fp32 GEMMs  C[p] = A[p] (M x K) . B[p] (K x N)  from six bf16 MFMA
// products per fp32 product, as a real kernel (tools/probes/split6_gemm.hip is the accuracy half; its variant 6 is what this
// kernel computes).  x = x0 + x1 + x2 (three bf16 terms), a.b ~ a2b0 + a0b2 + a1b1 + a1b0 + a0b1 + a0b0; per 32-deep k-step the
// six products of an output tile are chained from C = 0 (small terms first) and folded into the fp32 total by one add: two-level
// summation with ONE accumulator set - measured closer to fp64 than the product's native two-level fp32 MFMA GEMM
// (profiles/r05_split6_accuracy.txt: 8.3e-8 against 1.25e-7 relative L1 on real F(6x6,3x3) operands).
//
// Operands are PRE-SPLIT and K-BLOCKED (split3_kernel below + a permute; in a product the producing transforms would write them):
//   A3 [3][P][K/32][M][32] bf16 (term-major), B3 [3][P][K/32][N][32] bf16, C [P][M][N] fp32.
// K-blocked: the 16 rows x 64 bytes one DMA instruction moves are 1 KiB CONTIGUOUS in memory - eight whole 128-byte lines.  With
// plain [M][K] planes the same instruction touched sixteen lines, half of each, and the other halves a k-step later: the kernel
// was DMA-bound at 31 GB/s per CU (600 us without its MFMAs against 466 us with nothing but them).
// Machine: as the 16-bit conv kernels - 512 threads = 8 waves as 4 (M) x 2 (N), block tile 256 x 128, wave tile 64 x 64 = 4 x 4
// MFMA 16x16x32 tiles (weight-side fragment first: a lane ends up with 4 consecutive n of one m = one 16-byte store);
// both operands global -> LDS by LDS-DMA, 32-deep stages of 64-byte rows (chunk swizzle c ^ ((-(row >> 2)) & 3): conflict-free
// under ds_read_b128's real lane groups), stage = 3 x (256 + 128) rows x 64 B = 72 KiB, two stages; one barrier per k-step,
// in its middle; every fragment register refilled in place one half-step ahead of its use (A rows 2, 3 and, behind the last
// row's multiplies, the B fragments of the next step).  Persistent: one workgroup per CU walks 16 items (plane, 4 row tiles x
// 4 column tiles); the stage ring runs across item boundaries.  XCD x owns the planes x, x + 8, ...
#include <hip/hip_runtime.h>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const bf16x8 __attribute__((address_space(3))) * lds_frag_p;

namespace {

constexpr int BM = 256, BN = 128, BK = 32;
constexpr int ROWB = BK * 2;                        // 64 bytes per LDS row
constexpr int A_T = BM * ROWB, B_T = BN * ROWB;     // one term of one stage: 16 KiB, 8 KiB
constexpr int STAGE = 3 * (A_T + B_T);              // 73728
constexpr int B_OFF = 3 * A_T;

struct SplitKP {
    const char* A3;     // [3][P][K/32][M][32] bf16
    const char* B3;     // [3][P][K/32][N][32] bf16
    float* C;           // [P][M][N]
    long long M;
    int K, N, P;
    int MT, NT;         // row / column tiles per plane
    int items;          // P * MT * NT
    int dbg;            // timing-only: 1 no DMA after the prologue, 2 no fragment reads, 4 no MFMAs, 8 no stores
};

__device__ __forceinline__ bf16x8 lds_frag(unsigned a) { return *reinterpret_cast<lds_frag_p>(a); }
__device__ __forceinline__ unsigned lds_addr_of(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p; }

__global__ void __launch_bounds__(512, 2) split6_gemm_kernel(const SplitKP p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    // work list: XCD x owns the planes x, x + 8, ...; its list of (plane, row tile, column tile) items (column tile fastest) is
    // dealt ROUND-ROBIN to its workgroups: at any time the XCD's workgroups sit on neighbouring items of ONE plane - the four
    // column tiles of a row tile read the same A panel, the row tiles the same B panel, so all but the first read of a line
    // hit that XCD's L2 (contiguous runs per workgroup had every workgroup stream its own panels: 24 GB/s per CU, the rate of
    // an HBM sweep, and the kernel was DMA-bound: 793 us without its MFMAs against 462 us with nothing but them)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wpx = gridDim.x >> 3;
    const int planes_x = (p.P - xcd + 7) / 8;
    const int per_plane = p.MT * p.NT;
    const int items_x = planes_x * per_plane;
    if (slot >= items_x) return;
    const int my_items = (items_x - slot + wpx - 1) / wpx;
    const int KS = p.K / BK;
    const int nsteps = my_items * KS;
    const size_t a_term = (size_t)p.P * p.M * p.K * 2, b_term = (size_t)p.P * p.N * p.K * 2;

    // DMA roles: instruction q (0..71) of a stage; wave w issues q = w, w + 8, ...: q < 48: A term q / 16, rows (q % 16) * 16 ..;
    // else B term (q - 48) / 8, rows ((q - 48) % 8) * 16 ..  Lane L: row L / 4 of the 16, physical chunk L % 4 = global chunk
    // (L % 4) ^ key(row)
    const int drow = lane >> 2;
    const unsigned dkey = (unsigned)((-(drow >> 2)) & 3);
    const unsigned dchunk = ((unsigned)(lane & 3) ^ dkey) * 16u;
    const unsigned lds0 = lds_addr_of(smem);
    auto issue_stage = [&](int g) {         // global step g = item * KS + ks -> stage buffer g & 1
        const int item = slot + wpx * (g / KS), ks = g - (g / KS) * KS;
        const int pl = xcd + 8 * (item / per_plane), rem = item % per_plane;
        const int mt = rem / p.NT, nt = rem - mt * p.NT;
        const unsigned sbase = lds0 + (unsigned)(g & 1) * STAGE;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const int q = wave + 8 * r;             // r < 6: A (q < 48), else B
            if (r < 6) {
                const int t = q >> 4, rb = (q & 15) * 16;
                long long m = (long long)mt * BM + rb + drow;
                m = m < p.M ? m : p.M - 1;          // ragged last row tile: re-read the last row (never stored)
                const char* src = p.A3 + t * a_term + ((((size_t)pl * KS + ks) * p.M + m) * BK) * 2 + dchunk;
                __builtin_amdgcn_global_load_lds(src, (lds_vp)(smem + (sbase - lds0) + t * A_T + rb * ROWB), 16, 0, 0);
            } else {
                const int qb = q - 48, t = qb >> 3, rb = (qb & 7) * 16;
                const char* src = p.B3 + t * b_term + ((((size_t)pl * KS + ks) * p.N + nt * BN + rb + drow) * BK) * 2 + dchunk;
                __builtin_amdgcn_global_load_lds(src, (lds_vp)(smem + (sbase - lds0) + B_OFF + t * B_T + rb * ROWB), 16, 0, 0);
            }
        }
    };

    // fragment addresses: row (wm * 64 + i * 16 + l15) / (wn * 64 + j * 16 + l15), chunk g4 ^ key(l15); i, j, term: immediates
    const unsigned fkey = (unsigned)((-(l15 >> 2)) & 3);
    const unsigned fa = lds0 + (unsigned)(wm * 64 + l15) * ROWB + ((((unsigned)g4) ^ fkey) << 4);
    const unsigned fb = lds0 + B_OFF + (unsigned)(wn * 64 + l15) * ROWB + ((((unsigned)g4) ^ fkey) << 4);

    f32x4 acc[4][4];
    bf16x8 af[4][3], bf[4][3];
    auto load_a = [&](int i, unsigned st) {
        if (p.dbg & 2) return;
#pragma unroll
        for (int t = 0; t < 3; ++t) af[i][t] = lds_frag(fa + st + (unsigned)(t * A_T + i * 16 * ROWB));
    };
    auto load_b = [&](int j, unsigned st) {
        if (p.dbg & 2) return;
#pragma unroll
        for (int t = 0; t < 3; ++t) bf[j][t] = lds_frag(fb + st + (unsigned)(t * B_T + j * 16 * ROWB));
    };
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 3; ++t) { af[i][t] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0}; bf[i][t] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0}; }

    // one output tile's six products, chained from zero, small terms first (a2b0, a0b2, a1b1, a1b0, a0b1, a0b0): D[n][m]
    auto six = [&](int i, int j) -> f32x4 {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if (p.dbg & 4) return t;
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][0], af[i][2], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][2], af[i][0], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][1], af[i][1], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][0], af[i][1], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][1], af[i][0], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][0], af[i][0], t, 0, 0, 0);
        return t;
    };

    // prologue: stages 0 and 1, then the fragments the first step starts with (all of B, A rows 0 and 1)
    issue_stage(0);
    if (nsteps > 1) issue_stage(1);
    if (nsteps > 1) __builtin_amdgcn_s_waitcnt(0x0070 | 9);        // stage 0 landed (stage 1's nine may fly)
    else __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) load_b(j, 0);
    load_a(0, 0); load_a(1, 0);

    int item = slot, ks = 0;
    for (int g = 0; g < nsteps; ++g) {
        const unsigned st = (unsigned)(g & 1) * STAGE, stn = STAGE - st;
        if (ks == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // ---- half 0: rows 0, 1; rows 2, 3 of this step stream in (the last reads of stage g)
        load_a(2, st); load_a(3, st);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] += six(i, j);
        // ---- middle: stage g is read; stage g + 1 (issued one step ago) must have landed before anybody reads it
        __builtin_amdgcn_s_waitcnt(0x0070);
        __syncthreads();
        // the DMA of stage g + 2: a DMA instruction costs its wave's issue slot tens of cycles, so the two waves of a SIMD (w, w + 4)
        // issue their nine at different points of half 1 - one wave's issue runs under the other's multiplies
        const bool dma = g + 2 < nsteps && !(p.dbg & 1);
        const bool early = wave < 4 || (p.dbg & 16);
        if (dma && early) issue_stage(g + 2);
        // ---- half 1: rows 2, 3; rows 0, 1 and all of B of the NEXT step stream in (in place, behind their last use)
        const bool more = g + 1 < nsteps;
        if (more) { load_a(0, stn); load_a(1, stn); }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[2][j] += six(2, j);
        if (dma && !early) issue_stage(g + 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[3][j] += six(3, j);
            if (more) load_b(j, stn);
        }
        if (++ks == KS) {       // the item is complete: store its 64 x 64 wave tile, 16 bytes per lane
            const int pl = xcd + 8 * (item / per_plane), rem = item % per_plane;
            const int mt = rem / p.NT, nt = rem - mt * p.NT;
            if (!(p.dbg & 8)) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const long long m = (long long)mt * BM + wm * 64 + i * 16 + l15;
                    if (m < p.M) {
                        float* crow = p.C + ((size_t)pl * p.M + m) * p.N + nt * BN + wn * 64 + 4 * g4;
#pragma unroll
                        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(crow + j * 16) = acc[i][j];
                    }
                }
            }
            ks = 0; item += wpx;
        }
    }
}

// x [rows][K] fp32 -> three bf16 term planes [3][rows][K]; 8 elements per thread
__global__ void split3_kernel(const float* __restrict__ x, __bf16* __restrict__ out, long long n8, long long plane) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const f32x4 v0 = reinterpret_cast<const f32x4*>(x)[2 * i], v1 = reinterpret_cast<const f32x4*>(x)[2 * i + 1];
        bf16x8 t0, t1, t2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = e < 4 ? v0[e] : v1[e - 4];
            const __bf16 h0 = (__bf16)v;
            const float r1 = v - (float)h0;
            const __bf16 h1 = (__bf16)r1;
            const float r2 = r1 - (float)h1;
            t0[e] = h0; t1[e] = h1; t2[e] = (__bf16)r2;
        }
        reinterpret_cast<bf16x8*>(out)[i] = t0;
        reinterpret_cast<bf16x8*>(out + plane)[i] = t1;
        reinterpret_cast<bf16x8*>(out + 2 * plane)[i] = t2;
    }
}

}  // namespace

extern "C" int split3(const float* x, void* out, long long n, hipStream_t st) {
    if (n % 8) return 1;
    hipLaunchKernelGGL(split3_kernel, dim3(4096), dim3(256), 0, st, x, static_cast<__bf16*>(out), n / 8, n);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int split6_gemm_fast(const void* A3, const void* B3, float* C, long long M, int K, int N, int P, int dbg, hipStream_t st) {
    if (K % BK || N % BN || P < 1 || M < 1) return 1;
    static bool ready = false;
    if (!ready) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(split6_gemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE) != hipSuccess) return 2;
        ready = true;
    }
    SplitKP p{};
    p.A3 = static_cast<const char*>(A3); p.B3 = static_cast<const char*>(B3); p.C = C;
    p.M = M; p.K = K; p.N = N; p.P = P;
    p.MT = (int)((M + BM - 1) / BM); p.NT = N / BN; p.items = P * p.MT * p.NT; p.dbg = dbg;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipLaunchKernelGGL(split6_gemm_kernel, dim3(8 * (cus / 8)), dim3(512), 2 * STAGE, st, p);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
