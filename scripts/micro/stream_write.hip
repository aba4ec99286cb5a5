// Dev microbenchmark: what store rate does the chip sustain (a) for a plain contiguous fill and
// (b) for the epilogue pattern of a 256x128 GEMM tile (each wave stores a 32x32 fp32 accumulator:
// 16 dword stores of 2 x 128 B, rows `ld` floats apart), as a function of the row stride?
// hipcc --offload-arch=gfx950 -O3 scripts/micro/stream_write.hip -o scripts/micro/stream_write
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void fill_kernel(float4* __restrict__ p, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

// grid = tiles_n * tiles_m workgroups of 512 threads; tile (tm, tn) = rows tm*256.., cols tn*128..;
// m_fastest tile order as in the Y GEMM; element (m, n) at p[m*ld + n] (blocked = 0) or at
// p[((n>>11)*rows + m)*2048 + (n&2047)] (blocked = 1: one K-tile block per node contiguous).
__global__ __launch_bounds__(512) void tile_store_kernel(float* __restrict__ p, int rows, int N, int tiles_m, int blocked) {
    const int tile = blockIdx.x;
    const int bm = (tile % tiles_m) * 256, bn = (tile / tiles_m) * 128;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = bn + wn * 64 + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = bm + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < rows) {
                    const size_t o = blocked ? ((size_t)(n >> 11) * rows + m) * 2048 + (n & 2047) : (size_t)m * N + n;
                    p[o] = (float)(m + n);
                }
            }
    }
}

int main() {
    const size_t maxb = 4ull << 30;
    float* buf;
    hipMalloc(&buf, maxb);
    hipMemset(buf, 0, maxb);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int reps = 20;
    for (size_t mb : {128, 512, 2048}) {
        const size_t bytes = mb << 20, n16 = bytes / 16;
        for (int blocks : {2048, 8192}) {
            for (int i = 0; i < 3; ++i) fill_kernel<<<blocks, 256>>>((float4*)buf, n16);
            hipDeviceSynchronize();
            hipEventRecord(a);
            for (int i = 0; i < reps; ++i) fill_kernel<<<blocks, 256>>>((float4*)buf, n16);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
            printf("fill %5zu MB  blocks %5d  %8.1f us  %7.1f GB/s\n", mb, blocks, ms * 1e3, bytes / ms / 1e6);
        }
    }
    for (int rows : {504, 4032}) {
        for (int N : {65536, 65536 + 2048, 32768}) {
            for (int blocked = 0; blocked < 2; ++blocked) {
                const int tiles_m = (rows + 255) / 256, tiles_n = N / 128;
                const size_t bytes = (size_t)rows * N * 4;
                for (int i = 0; i < 3; ++i) tile_store_kernel<<<tiles_m * tiles_n, 512>>>(buf, rows, N, tiles_m, blocked);
                hipDeviceSynchronize();
                hipEventRecord(a);
                for (int i = 0; i < reps; ++i) tile_store_kernel<<<tiles_m * tiles_n, 512>>>(buf, rows, N, tiles_m, blocked);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
                printf("tile-store rows %5d N %6d blocked %d  %8.1f us  %7.1f GB/s\n", rows, N, blocked, ms * 1e3, bytes / ms / 1e6);
            }
        }
    }
    return 0;
}
