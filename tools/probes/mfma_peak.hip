// What the matrix cores of THIS box sustain: bare MFMA loops (operands in registers, random data, 2 waves per SIMD,
// every CU busy), timed over ~1 s after a warm-up so that the clock has settled under load.  The numbers DESIGN.md
// prices the conv kernels against - the nominal 2.5 PFLOP/s (bf16) assumes 2.4 GHz, which a dense MFMA stream does not
// hold (MI355X guide, DVFS give-back).
//     hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ void __launch_bounds__(512, 2) k(const float* in, float* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0.f;
    if (KIND == 0) {            // v_mfma_f32_16x16x32_bf16, 32 independent accumulators (the conv kernels' wave tile)
        bf16x8 a[8], b[4];
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)in[(tid * 8 + i * 8 + e) & 4095];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) b[i][e] = (__bf16)in[(tid * 4 + i * 8 + e + 17) & 4095];
        f32x4 acc[8][4];
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    } else if (KIND == 1) {     // v_mfma_f32_32x32x16_bf16, 8 accumulators
        bf16x8 a[4], b[2];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)in[(tid * 8 + i * 8 + e) & 4095];
        for (int i = 0; i < 2; ++i) for (int e = 0; e < 8; ++e) b[i][e] = (__bf16)in[(tid * 4 + i * 8 + e + 17) & 4095];
        f32x16 acc[4][2];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
    } else {                    // v_mfma_f32_32x32x2_f32, 4 accumulators
        float a[2], b[2];
        for (int i = 0; i < 2; ++i) { a[i] = in[(tid + i * 64) & 4095]; b[i] = in[(tid + i * 64 + 9) & 4095]; }
        f32x16 acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
    }
    out[tid] = s;
}

int main() {
    float *in, *out;
    std::vector<float> h(4096);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 512 * 512 * 4);
    hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"v_mfma_f32_16x16x32_bf16 (32 accumulators)", "v_mfma_f32_32x32x16_bf16 (8 accumulators)",
                            "v_mfma_f32_32x32x2_f32 (4 accumulators)"};
    const double flop_per_iter_wave[3] = {32.0 * 2 * 16 * 16 * 32, 8.0 * 2 * 32 * 32 * 16, 32.0 * 2 * 32 * 32 * 2};
    for (int kind = 0; kind < 3; ++kind) {
        const int iters = kind == 2 ? 20000 : 40000, grid = 512;       // 512 workgroups of 8 waves: 2 per CU-slot rounds
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int l = 0; l < 4; ++l) {
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, in, out, iters);
                else if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, in, out, iters);
                else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, in, out, iters);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double tf = 4.0 * grid * 8 * iters * flop_per_iter_wave[kind] / (ms * 1e-3) / 1e12;
            if (rep == 2) printf("%-48s %8.1f TFLOP/s sustained (%.0f ms)\n", names[kind], tf, ms);
        }
    }
    return 0;
}
