// fp32 Winograd-domain wgrad GEMMs  dU[p] = V[p]^T . Yh[p]  (P planes; V [P][T][Cin], Yh [P][T][Cout], dU [P][Cin][Cout]; the
// contraction runs over the T tiles) as a two-stage LDS-DMA ring, as wgrad_s2.hip streams its strips: both operands are
// k-major with the channel index contiguous, so a stage is a plain copy of 32 rows of each (32 + 32 KiB) and both MFMA
// fragments are ds_read_b32 of 32 consecutive floats at immediate offsets - no transposed staging, no swizzle, no address
// arithmetic in the loop.  (wino_wgrad_gemm_kernel stages through registers with two barriers per k-step at three
// work-groups per CU: 0.78 of the fp32 MFMA peak.)
//
// Work-group = 512 threads = 8 waves, output block 256 ci x 256 co of one plane (wave: 128 x 64 = eight 32x32 accumulator
// tiles), over a range of tiles.  Per 32-row stage and wave: 16 k-steps x (4 + 2 ds_read_b32, 8 independent
// v_mfma_f32_32x32x2_f32); two stages (64 KiB each), one barrier per stage; the DMA of stage s + 1 is issued during stage s -
// by the two waves of a SIMD at opposite ends of it.  512 x 512 channels: 64 planes x 2 x 2 blocks = 256 work-groups, each the
// whole contraction - no split-K, no slab pass; fewer blocks than CUs (256 x 256: 64) split the tile range and sum slabs in a
// fixed order (slab_reduce.hip).
#include <algorithm>
#include "common.h"

namespace mmh { int g_wino_wgrad_dma = 1; }

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const float __attribute__((address_space(3))) * lds_f_p;
__device__ __forceinline__ float lds_f(unsigned addr, int imm) { return *reinterpret_cast<lds_f_p>((size_t)(addr + (unsigned)imm)); }

constexpr int NT = 512;
constexpr int BMC = 256, BNC = 256;             // block: input channels x output channels
constexpr int KS = 32;                          // tiles (rows of the contraction) per stage
constexpr int A_B = KS * BMC * 4;               // 32768
constexpr int B_B = KS * BNC * 4;               // 32768
constexpr int ST_B = A_B + B_B;                 // 65536
constexpr int NST = 2;
constexpr int LDS_B = NST * ST_B;               // 131072
constexpr int AR = A_B / (NT * 16), BR = B_B / (NT * 16);      // 4 + 4 DMA instructions per thread and stage

__device__ char g_zero_line[128];

__device__ __forceinline__ void dma16_s(const void* sbase, unsigned voff, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_base) : "memory", "m0");
}

struct WinoWgradDmaKP {
    const float* V;         // [P][T][Cin]
    const float* Y;         // [P][T][Cout]
    float* out;             // dU [P][Cin][Cout] (S == 1) or slabs [P][S][Cin][Cout]
    int T, Cin, Cout, P, S;
    int st_per;             // stages per split
    int MT, NT_, blocks;
};

__global__ void __launch_bounds__(NT, 1) wino_wgrad_dma_kernel(const WinoWgradDmaKP p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, kk = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;
    const unsigned lds0 = mmh::lds_addr_of(smem);
    const unsigned wdst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);

    // XCD x works on a contiguous range; the blocks of one (plane, split) read the same two panels
    const int L = (p.blocks & 7) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (p.blocks >> 3) + (blockIdx.x >> 3));
    const int bn = L % p.NT_, t1 = L / p.NT_;
    const int bm = t1 % p.MT, t2 = t1 / p.MT;
    const int split = t2 % p.S, pl = t2 / p.S;
    const int ci0 = bm * BMC, co0 = bn * BNC;
    const int nst_all = (p.T + KS - 1) / KS;
    const int s_beg = split * p.st_per, s_end = min(nst_all, s_beg + p.st_per);

    // DMA lane offsets (bytes) inside a stage of 32 rows: A and B rows of 256 floats (64 chunks)
    unsigned a_off[AR], b_off[BR];
#pragma unroll
    for (int r = 0; r < AR; ++r) {
        const int u = r * NT + tid;
        a_off[r] = (unsigned)((u >> 6) * p.Cin * 4 + (u & 63) * 16);
    }
#pragma unroll
    for (int r = 0; r < BR; ++r) {
        const int u = r * NT + tid;
        b_off[r] = (unsigned)((u >> 6) * p.Cout * 4 + (u & 63) * 16);
    }
    const float* const Vp = p.V + (size_t)pl * p.T * p.Cin + ci0;
    const float* const Yp = p.Y + (size_t)pl * p.T * p.Cout + co0;
    const void* const zero = g_zero_line + (lane & 7) * 16;
    // stage s (rows 32 s ..) -> slot s % 2; a stage past the range or with rows past T reads zeros for those rows
    auto issue = [&](int s, int slot) {
        const unsigned dst = wdst + (unsigned)(slot * ST_B);
        const int row0 = s * KS;
        if (s < s_end && row0 + KS <= p.T) {
            const float* va = Vp + (size_t)row0 * p.Cin;
            const float* ya = Yp + (size_t)row0 * p.Cout;
#pragma unroll
            for (int r = 0; r < AR; ++r) dma16_s(va, a_off[r], dst + (unsigned)(r * NT * 16));
#pragma unroll
            for (int r = 0; r < BR; ++r) dma16_s(ya, b_off[r], dst + (unsigned)(A_B + r * NT * 16));
        } else {
            const int rows = s < s_end ? p.T - row0 : 0;
#pragma unroll
            for (int r = 0; r < AR; ++r) {
                const int u = r * NT + tid;
                mmh::lds_dma16((u >> 6) < rows ? (const void*)((const char*)(Vp + (size_t)row0 * p.Cin) + a_off[r]) : zero,
                               dst + (unsigned)(r * NT * 16));
            }
#pragma unroll
            for (int r = 0; r < BR; ++r) {
                const int u = r * NT + tid;
                mmh::lds_dma16((u >> 6) < rows ? (const void*)((const char*)(Yp + (size_t)row0 * p.Cout) + b_off[r]) : zero,
                               dst + (unsigned)(A_B + r * NT * 16));
            }
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const unsigned a_lane = lds0 + (unsigned)(kk * BMC * 4 + (128 * wm + m) * 4);
    const unsigned b_lane = lds0 + (unsigned)(A_B + kk * BNC * 4 + (64 * wn + m) * 4);
    if (s_beg < s_end) {
        issue(s_beg, 0);
        int slot = 0;
        for (int s = s_beg; s < s_end; ++s) {
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0x0070);                     // vmcnt(0): stage s landed
            __builtin_amdgcn_s_barrier();                           // ... everybody's; the other slot (stage s - 1) is free
            asm volatile("" ::: "memory");
            // the two waves of a SIMD (wm = 0 / 1) issue the next stage's DMA at opposite ends of the stage: a DMA instruction
            // costs its wave 60-180 cycles of issue time, which the other wave's MFMAs cover
            if (wm == 0) issue(s + 1, slot ^ 1);
            unsigned ab = a_lane + (unsigned)(slot * ST_B), bb = b_lane + (unsigned)(slot * ST_B);
            asm volatile("" : "+v"(ab), "+v"(bb));
#pragma unroll
            for (int ks = 0; ks < KS / 2; ++ks) {
                float a[4], b[2];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = lds_f(ab, ks * 2 * BMC * 4 + 128 * i);
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = lds_f(bb, ks * 2 * BNC * 4 + 128 * j);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
                if (ks == KS / 4 - 1 && wm != 0) issue(s + 1, slot ^ 1);
            }
            slot ^= 1;
        }
    }
    // acc[i][j][e]: row ci0 + 128 wm + 32 i + (e & 3) + 8 (e >> 2) + 4 kk, column co0 + 64 wn + 32 j + m
    float* const o = p.out + ((size_t)(pl * p.S + split) * p.Cin + ci0 + 128 * wm + 4 * kk) * p.Cout + co0 + 64 * wn + m;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float* const r = o + (size_t)(32 * i + (e & 3) + 8 * (e >> 2)) * p.Cout;
            r[0] = acc[i][0][e];
            r[32] = acc[i][1][e];
        }
}

int splits_for(int64_t tiles, int Cin, int Cout, int nbatch) {
    const int blocks0 = nbatch * (Cin / BMC) * (Cout / BNC);
    const int nst = (int)((tiles + KS - 1) / KS);
    int S = 1;
    while (blocks0 * S * 2 <= 256 && nst / (S * 2) >= 8) S *= 2;
    return S;
}

}  // namespace

namespace mmh {

bool wino_wgrad_dma_ok(int64_t tiles, int Cin, int Cout, int nbatch) {
    return g_wino_wgrad_dma && Cin % BMC == 0 && Cout % BNC == 0 && tiles >= KS && nbatch > 0 &&
           tiles * (int64_t)std::max(Cin, Cout) < (1ll << 29);
}

size_t wino_wgrad_dma_ws_bytes(int64_t tiles, int Cin, int Cout, int nbatch) {
    const int S = splits_for(tiles, Cin, Cout, nbatch);
    return S > 1 ? (size_t)nbatch * S * Cin * Cout * sizeof(float) : 0;
}

int launch_wino_wgrad_dma(const float* V, const float* Yh, int64_t tiles, int Cin, int Cout, int nbatch, float* ws,
                          float* dU, hipStream_t st) {
    WinoWgradDmaKP p{};
    p.V = V; p.Y = Yh;
    p.T = (int)tiles; p.Cin = Cin; p.Cout = Cout; p.P = nbatch;
    p.S = splits_for(tiles, Cin, Cout, nbatch);
    const int nst = (p.T + KS - 1) / KS;
    p.st_per = (nst + p.S - 1) / p.S;
    p.out = p.S > 1 ? ws : dU;
    p.MT = Cin / BMC; p.NT_ = Cout / BNC;
    p.blocks = nbatch * p.S * p.MT * p.NT_;
    static int ready = -1;
    if (ready != 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wino_wgrad_dma_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
        if (e != hipSuccess) return fail("wino_wgrad_dma: %s", hipGetErrorString(e));
        ready = 0;
    }
    hipLaunchKernelGGL(wino_wgrad_dma_kernel, dim3(p.blocks), dim3(NT), LDS_B, st, p);
    if (int rc = check_launch("wino_wgrad_dma_kernel")) return rc;
    if (p.S == 1) return 0;
    const int64_t n4 = (int64_t)Cin * Cout / 4;
    return launch_slab_reduce(ws, dU, nbatch * n4, p.S, 0, n4, st);
}

}  // namespace mmh
