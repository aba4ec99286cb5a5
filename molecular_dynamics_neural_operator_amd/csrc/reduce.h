// Sum of fixed-order partial results: out[i] (+)= sum over slices of part[slice][i].  Shared by the training
// translation units (column sums, root / bias gradients, K-sliced A^T.B): the reductions that make every gradient
// bitwise reproducible without float atomics.
//
// A workgroup owns 64 outputs; its four waves each add one contiguous quarter of the slices in slice order, eight
// loads in flight, and the four quarter sums are combined in quarter order — a fixed association, whatever the
// launch.  (Round 3: one thread per output walking all slices with one load in flight — 128 slices cost 38 us for
// 4,096 outputs, 26 us for 1,024; eight such launches per training batch.)
#pragma once
#include <hip/hip_runtime.h>

namespace mdno {

__global__ __launch_bounds__(256) static void reduce_slices_q4_kernel(const float* __restrict__ part, int slices,
                                                                      long long count, float* __restrict__ out,
                                                                      int accumulate) {
    __shared__ float comb[4][64];
    const int c = threadIdx.x & 63, qt = threadIdx.x >> 6;
    const long long id = (long long)blockIdx.x * 64 + c;
    const int per = (slices + 3) / 4;
    const int k0 = qt * per;
    int k1 = k0 + per;
    if (k1 > slices) k1 = slices;
    float s = 0.f;
    if (id < count) {
        int k = k0;
        for (; k + 8 <= k1; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(k + u) * count + id];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < k1; ++k) s += part[(size_t)k * count + id];
    }
    comb[qt][c] = s;
    __syncthreads();
    if (qt == 0 && id < count) {
        const float t = (comb[0][c] + comb[1][c]) + (comb[2][c] + comb[3][c]);
        out[id] = accumulate ? out[id] + t : t;
    }
}

inline void launch_reduce_slices(const float* part, int slices, long long count, float* out, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(reduce_slices_q4_kernel, dim3((unsigned)((count + 63) / 64)), dim3(256), 0, s, part, slices, count, out,
                       accumulate);
}

}  // namespace mdno
