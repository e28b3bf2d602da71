#include <hip/hip_runtime.h>
__global__ void k(unsigned* o) {
    unsigned a = threadIdx.x, b = threadIdx.x + 1000;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 512); k<<<1, 64>>>(d); unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; i += 8) printf("lane %2d: r0 %4u r1 %4u\n", i, h[i], h[64 + i]);
    return 0;
}
