// How fast do the epilogue store shapes of the 16-bit conv kernels write?  [M pixels][C = 256 channels] bf16 (512 B per pixel),
// 8 waves per workgroup as in conv_lp16h2_kernel (wave = (wr, wc): 8 rows x 16 pixels x 64 channels each), one 16 x 16-pixel
// tile per workgroup iteration:
//   A  8 B per lane (store4 of an MFMA 16x16x32 accumulator): per instruction 16 pixels x 32 contiguous bytes
//   B  16 B per lane (two accumulators traded between lane pairs): 16 pixels x 64 contiguous bytes
//   C  16 B per lane, 1 KiB contiguous per instruction (tile transposed through LDS first - the LDS pass is NOT timed here)
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(512) k(char* y, int tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, g4 = lane >> 4, wr = wave >> 2, wc = wave & 3;
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        char* base = y + (size_t)t * 256 * 512;       // 256 pixels x 512 B
        if (MODE == 0) {
            for (int i = 0; i < 8; ++i)
                for (int j = 0; j < 4; ++j) {
                    uint2 v = {(unsigned)t, (unsigned)(i * 4 + j)};
                    *reinterpret_cast<uint2*>(base + (size_t)((wr * 8 + i) * 16 + l15) * 512 + (wc * 64 + j * 16 + 4 * g4) * 2) = v;
                }
        } else if (MODE == 1) {
            for (int i = 0; i < 8; ++i)
                for (int j = 0; j < 2; ++j) {
                    uint4 v = {(unsigned)t, (unsigned)i, (unsigned)j, 0u};
                    *reinterpret_cast<uint4*>(base + (size_t)((wr * 8 + i) * 16 + l15) * 512 + (wc * 64 + j * 32 + 8 * g4) * 2) = v;
                }
        } else {
            for (int i = 0; i < 16; ++i) {
                uint4 v = {(unsigned)t, (unsigned)i, 0u, 0u};
                *reinterpret_cast<uint4*>(base + (size_t)(wave * 16 + i) * 1024 + lane * 16) = v;
            }
        }
    }
}
int main() {
    const int tiles = 512;  // 32 x 64 x 64 pixels / 256 = the 256-channel conv of the step; x2 for 512 channels
    char* y; hipMalloc(&y, (size_t)tiles * 256 * 512);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {512, 256}) for (int mode = 0; mode < 3; ++mode) {
        auto run = [&]() { if (mode == 0) k<0><<<grid, 512>>>(y, tiles); else if (mode == 1) k<1><<<grid, 512>>>(y, tiles); else k<2><<<grid, 512>>>(y, tiles); };
        for (int i = 0; i < 3; ++i) run();
        hipEventRecord(e0); for (int i = 0; i < 20; ++i) run(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms / 20 * 1e3, gb = (double)tiles * 256 * 512 / 1e9;
        printf("grid %d mode %c: %.1f us for %.0f MB = %.2f TB/s\n", grid, "ABC"[mode], us, gb * 1e3, gb / us * 1e3);
    }
    return 0;
}
