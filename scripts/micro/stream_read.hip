// Dev microbenchmark: pure streaming read of B bytes, to calibrate the per-launch fixed cost and the
// steady-state HBM read rate against which the conv kernel is judged.  hipcc --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ p, size_t n16, float* out, size_t per_wave16) {
    // each wave streams a contiguous run of per_wave16 float4 (like one conv segment: 16 KiB x 16)
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    size_t base = wave * per_wave16;
    if (base >= n16) return;
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t i = 0; i < per_wave16; i += 64 * 16) {
        float4 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = p[base + i + j * 64 + lane];
#pragma unroll
        for (int j = 0; j < 16; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

int main() {
    const size_t maxb = 8ull << 30;
    float4* buf; float* out;
    hipMalloc(&buf, maxb); hipMalloc(&out, 64);
    hipMemset(buf, 0, maxb);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (size_t per_wave_kib : {256, 1024}) {
        for (size_t mb : {256, 512, 1024, 2048, 4096, 8192}) {
            const size_t bytes = mb << 20, n16 = bytes / 16, pw16 = per_wave_kib * 1024 / 16;
            const size_t waves = (n16 + pw16 - 1) / pw16;
            const int blocks = (int)((waves + 3) / 4);
            for (int i = 0; i < 3; ++i) read_kernel<<<blocks, 256>>>(buf, n16, out, pw16);
            hipDeviceSynchronize();
            const int reps = 20;
            hipEventRecord(a);
            for (int i = 0; i < reps; ++i) read_kernel<<<blocks, 256>>>(buf, n16, out, pw16);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
            printf("per-wave %4zu KiB  %5zu MB  %8.1f us  %7.1f GB/s  waves %zu\n", per_wave_kib, mb, ms * 1e3, bytes / ms / 1e6, waves);
        }
    }
    return 0;
}
