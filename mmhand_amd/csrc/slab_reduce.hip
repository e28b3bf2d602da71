// Fixed-order reduction of split-K slabs, shared by the wgrad kernels: dw[b][i] (+)= sum_z slab[b][z][i].
// One thread per float4 column summing its slabs one after the other (the round-1 kernel) is latency-bound as soon as there
// are many slabs and few columns: 256 slabs x 295 KB (the 7x7 stems, the stride-2 wgrads) ran at 1.3 TB/s.  Here a
// work-group covers 64 float4 columns x RZ groups of slabs: group g sums slabs g, g + RZ, ... with four loads in flight,
// and the RZ partial sums are added in order.  The order depends on (splits, RZ) only: deterministic.
#include "common.h"

namespace {

template <int RZ>
__global__ void __launch_bounds__(64 * RZ) slab_reduce_par_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                                  int64_t n4_total, int splits, int accumulate, int64_t n4) {
    __shared__ float4 part[RZ][64];
    const int col = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + col;
    float4 s[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4_total) {
        const int64_t b = i / n4, r = i - b * n4;
        const float4* base = reinterpret_cast<const float4*>(slab) + b * splits * n4 + r;
        int z = g;
        for (; z + 3 * RZ < splits; z += 4 * RZ) {
            float4 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = base[(int64_t)(z + u * RZ) * n4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { s[u].x += t[u].x; s[u].y += t[u].y; s[u].z += t[u].z; s[u].w += t[u].w; }
        }
        for (; z < splits; z += RZ) {
            const float4 t = base[(int64_t)z * n4];
            s[0].x += t.x; s[0].y += t.y; s[0].z += t.z; s[0].w += t.w;
        }
    }
    part[g][col] = make_float4((s[0].x + s[1].x) + (s[2].x + s[3].x), (s[0].y + s[1].y) + (s[2].y + s[3].y),
                               (s[0].z + s[1].z) + (s[2].z + s[3].z), (s[0].w + s[1].w) + (s[2].w + s[3].w));
    __syncthreads();
    if (g == 0 && i < n4_total) {
        float4 r = part[0][col];
#pragma unroll
        for (int q = 1; q < RZ; ++q) { const float4 t = part[q][col]; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
        float4* o = reinterpret_cast<float4*>(dw) + i;
        if (accumulate) { const float4 t = *o; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
        *o = r;
    }
}

// few slabs: one thread per column
__global__ void slab_reduce_seq_kernel(const float* __restrict__ slab, float* __restrict__ dw, int64_t n4_total, int splits,
                                       int accumulate, int64_t n4) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n4_total; i += stride) {
        const int64_t b = i / n4, r = i - b * n4;
        const float4* base = reinterpret_cast<const float4*>(slab) + b * splits * n4 + r;
        float4 s = base[0];
        for (int z = 1; z < splits; ++z) {
            const float4 t = base[(int64_t)z * n4];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        float4* o = reinterpret_cast<float4*>(dw) + i;
        if (accumulate) { const float4 t = *o; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
        *o = s;
    }
}

}  // namespace

namespace mmh {

int g_slab_reduce_par = 1;

// n4 = float4 elements per batch; batch b's slabs are [b * splits, (b + 1) * splits); n4_total = batches * n4
int launch_slab_reduce(const float* slab, float* dw, int64_t n4_total, int splits, int accumulate, int64_t n4, hipStream_t st) {
    if (g_slab_reduce_par && splits >= 32)
        hipLaunchKernelGGL(slab_reduce_par_kernel<8>, dim3((unsigned)cdiv(n4_total, 64)), dim3(512), 0, st, slab, dw, n4_total,
                           splits, accumulate, n4);
    else if (g_slab_reduce_par && splits >= 8)
        hipLaunchKernelGGL(slab_reduce_par_kernel<4>, dim3((unsigned)cdiv(n4_total, 64)), dim3(256), 0, st, slab, dw, n4_total,
                           splits, accumulate, n4);
    else
        hipLaunchKernelGGL(slab_reduce_seq_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n4_total, 256), 4096)), dim3(256), 0, st,
                           slab, dw, n4_total, splits, accumulate, n4);
    return check_launch("slab_reduce");
}

}  // namespace mmh
