// Probe: an fp32 GEMM from SIX bf16 MFMA products.  x = x0 + x1 + x2 with three bf16 terms (8 + 8 + 8 significand
// bits), a.b ~ a0b0 + a0b1 + a1b0 + a0b2 + a1b1 + a2b0 (the dropped terms are <= 2^-24 of the product), every product exact in the
// MFMA's fp32 accumulator.  gfx950's fp32 MFMA runs at 1/16 of the bf16 rate, so six bf16 MFMAs per fp32 product are 2.67x the
// native rate - IF the sum they deliver is as close to fp64 as the native fp32 chain's.  This file answers that with the real
// instruction (v_mfma_f32_16x16x32_bf16: its internal summation and rounding are not documented) on the Winograd-domain GEMM of
// the dominant conv, C[p] = A[p] (M x K, K contiguous) . B[p] (K x N, N contiguous).  One wave per 16 x 16 output tile, operands
// straight from global memory: an ACCURACY probe, slow by construction.  Built by tools/probes/split6_probe.py's recipe:
//     hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/probes/split6_gemm.hip -o tools/probes/build/split6_gemm.so
#include <hip/hip_runtime.h>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(const float* v, bf16x8& t0, bf16x8& t1, bf16x8& t2) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h0 = (__bf16)v[e];             // round to nearest even
        const float r1 = v[e] - (float)h0;          // exact
        const __bf16 h1 = (__bf16)r1;
        const float r2 = r1 - (float)h1;            // exact
        t0[e] = h0; t1[e] = h1; t2[e] = (__bf16)r2;
    }
}

// variant 0: six products into ONE accumulator, small terms first within a k-step
//         1: a0b0 into `hi`, the five others into `lo`, hi + lo at the end
//         2: three products (a0b0 + a0b1 + a1b0): the lower-precision form (no credit; for the table)
//         3: one product (plain bf16 operands)
//         4: six products, one accumulator, LARGE term first
//         5: as 1, and both accumulators restarted every 128 k and folded into fp32 totals (two-level, as the native kernel's)
//         6: ONE accumulator set: per k32 step the six products chained from C = 0 (small terms first, a0b0 last) into a
//            transient, folded into the total by one fp32 add - two-level summation with a 32-deep inner chain
//         7: as 6 with the transient spanning TWO k32 steps (64-deep inner chain)
template <int VARIANT>
__global__ void __launch_bounds__(64) split_gemm_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                         int64_t M, int K, int N) {
    const int lane = threadIdx.x, l15 = lane & 15, g4 = lane >> 4;
    const int64_t m0 = (int64_t)blockIdx.x * 16;
    const int n0 = blockIdx.y * 16;
    const int p = blockIdx.z;
    A += (int64_t)p * M * K; B += (int64_t)p * K * N; C += (int64_t)p * M * N;
    f32x4 hi = {0.f, 0.f, 0.f, 0.f}, lo = {0.f, 0.f, 0.f, 0.f}, tot = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 32) {
        float av[8], bv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            av[e] = A[(m0 + l15) * K + k0 + 8 * g4 + e];               // A fragment: row l15, k = 8 g4 + e
            bv[e] = B[(int64_t)(k0 + 8 * g4 + e) * N + n0 + l15];      // B fragment: column l15, k = 8 g4 + e
        }
        bf16x8 a0, a1, a2, b0, b1, b2;
        split3(av, a0, a1, a2);
        split3(bv, b0, b1, b2);
#define MM(acc, x, y) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc, 0, 0, 0)
        if (VARIANT == 0) {
            MM(hi, a2, b0); MM(hi, a0, b2); MM(hi, a1, b1); MM(hi, a1, b0); MM(hi, a0, b1); MM(hi, a0, b0);
        } else if (VARIANT == 1 || VARIANT == 5) {
            MM(lo, a2, b0); MM(lo, a0, b2); MM(lo, a1, b1); MM(lo, a1, b0); MM(lo, a0, b1); MM(hi, a0, b0);
            if (VARIANT == 5 && ((k0 + 32) % 128 == 0)) {
                tot += hi + lo;
                hi = (f32x4){0.f, 0.f, 0.f, 0.f}; lo = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        } else if (VARIANT == 6 || VARIANT == 7) {
            if (VARIANT == 6 || (k0 & 32) == 0) lo = (f32x4){0.f, 0.f, 0.f, 0.f};
            MM(lo, a2, b0); MM(lo, a0, b2); MM(lo, a1, b1); MM(lo, a1, b0); MM(lo, a0, b1); MM(lo, a0, b0);
            if (VARIANT == 6 || (k0 & 32) != 0) tot += lo;
        } else if (VARIANT == 2) {
            MM(hi, a1, b0); MM(hi, a0, b1); MM(hi, a0, b0);
        } else if (VARIANT == 3) {
            MM(hi, a0, b0);
        } else {
            MM(hi, a0, b0); MM(hi, a0, b1); MM(hi, a1, b0); MM(hi, a1, b1); MM(hi, a0, b2); MM(hi, a2, b0);
        }
#undef MM
    }
    const f32x4 out = VARIANT == 5 ? tot + (hi + lo) : (VARIANT == 1 ? hi + lo : ((VARIANT == 6 || VARIANT == 7) ? tot : hi));
    // D: column (second operand's row) = l15, row (first operand's row) = 4 g4 + r
#pragma unroll
    for (int r = 0; r < 4; ++r) C[(m0 + 4 * g4 + r) * N + n0 + l15] = out[r];
}

// the native chains for the same table: fp32 fmaf in k order (what v_mfma_f32_32x32x2_f32 computes, bit for bit: MI355X guide),
// one level, and two levels with 32-deep inner chains (the product's wino_gemm_kernel<128, 2>)
template <int LEVELS>
__global__ void __launch_bounds__(256) chain_gemm_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                          int64_t M, int K, int N) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int p = blockIdx.z;
    if (idx >= M * N) return;
    const int64_t m = idx / N; const int n = (int)(idx - m * N);
    A += (int64_t)p * M * K; B += (int64_t)p * K * N; C += (int64_t)p * M * N;
    float tot = 0.f, acc = 0.f;
    for (int k = 0; k < K; ++k) {
        acc = __builtin_fmaf(A[m * K + k], B[(int64_t)k * N + n], acc);
        if (LEVELS == 2 && (k & 31) == 31) { tot += acc; acc = 0.f; }
    }
    C[m * N + n] = LEVELS == 2 ? tot + acc : acc;
}

extern "C" int split6_gemm(const float* A, const float* B, float* C, int64_t M, int K, int N, int P, int variant, hipStream_t st) {
    if (M % 16 || N % 16 || K % 32) return 1;
    const dim3 grid((unsigned)(M / 16), N / 16, P);
    switch (variant) {
        case 0: hipLaunchKernelGGL(split_gemm_kernel<0>, grid, dim3(64), 0, st, A, B, C, M, K, N); break;
        case 1: hipLaunchKernelGGL(split_gemm_kernel<1>, grid, dim3(64), 0, st, A, B, C, M, K, N); break;
        case 2: hipLaunchKernelGGL(split_gemm_kernel<2>, grid, dim3(64), 0, st, A, B, C, M, K, N); break;
        case 3: hipLaunchKernelGGL(split_gemm_kernel<3>, grid, dim3(64), 0, st, A, B, C, M, K, N); break;
        case 4: hipLaunchKernelGGL(split_gemm_kernel<4>, grid, dim3(64), 0, st, A, B, C, M, K, N); break;
        case 5: hipLaunchKernelGGL(split_gemm_kernel<5>, grid, dim3(64), 0, st, A, B, C, M, K, N); break;
        case 6: hipLaunchKernelGGL(split_gemm_kernel<6>, grid, dim3(64), 0, st, A, B, C, M, K, N); break;
        case 7: hipLaunchKernelGGL(split_gemm_kernel<7>, grid, dim3(64), 0, st, A, B, C, M, K, N); break;
        case 10: hipLaunchKernelGGL(chain_gemm_kernel<1>, dim3((unsigned)((M * N + 255) / 256), 1, P), dim3(256), 0, st, A, B, C, M, K, N); break;
        case 11: hipLaunchKernelGGL(chain_gemm_kernel<2>, dim3((unsigned)((M * N + 255) / 256), 1, P), dim3(256), 0, st, A, B, C, M, K, N); break;
        default: return 2;
    }
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
