// Dev microbenchmark (round 4): does a store-bound kernel beside a read-bound kernel get more out
// of the memory system than the two one after the other?  R: streaming non-temporal read of RB bytes; Wr: streaming
// write of WB bytes into a buffer that fits the Infinity Cache (the Y chunk: 128 MiB).  Timed: R alone, Wr alone,
// R then Wr on one stream, R beside Wr on two streams.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void read_kernel(const f32x4* __restrict__ p, size_t n16, float* out, size_t per_wave16) {
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const size_t base = wave * per_wave16;
    if (base >= n16) return;
    f32x4 acc = {0, 0, 0, 0};
    for (size_t i = 0; i < per_wave16; i += 64 * 8) {
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(p + base + i + j * 64 + lane);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

__global__ __launch_bounds__(256) void write_kernel(f32x4* __restrict__ p, size_t n16, size_t per_wave16, float val) {
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const size_t base = wave * per_wave16;
    if (base >= n16) return;
    const f32x4 v = {val, val, val, val};
    for (size_t i = 0; i < per_wave16; i += 64) p[base + i + lane] = v;
}

int main() {
    const size_t RB = 280ull << 20, WB = 128ull << 20;   // what one chunk's per-source GEMM pulls from HBM / one Y chunk
    f32x4 *rbuf, *wbuf; float* out;
    hipMalloc(&rbuf, 4 * RB); hipMalloc(&wbuf, 2 * WB); hipMalloc(&out, 64);
    hipMemset(rbuf, 0, 4 * RB);
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t a, b, e2; hipEventCreate(&a); hipEventCreate(&b); hipEventCreateWithFlags(&e2, hipEventDisableTiming);
    const size_t rn16 = RB / 16, wn16 = WB / 16, pw16 = 256 * 1024 / 16;
    const int rblocks = (int)((rn16 / pw16 + 3) / 4), wblocks = (int)((wn16 / pw16 + 3) / 4);
    auto R = [&](hipStream_t s, int i) { read_kernel<<<rblocks, 256, 0, s>>>(rbuf + (size_t)(i & 3) * rn16, rn16, out, pw16); };
    auto Wr = [&](hipStream_t s, int i) { write_kernel<<<wblocks, 256, 0, s>>>(wbuf + (size_t)(i & 1) * wn16, wn16, pw16, 1.f); };
    const int reps = 40;
    float ms;
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = 0; i < 4; ++i) { R(s1, i); Wr(s1, i); }
        hipDeviceSynchronize();
        hipEventRecord(a, s1); for (int i = 0; i < reps; ++i) R(s1, i); hipEventRecord(b, s1); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b); const float tr = ms / reps;
        hipEventRecord(a, s1); for (int i = 0; i < reps; ++i) Wr(s1, i); hipEventRecord(b, s1); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b); const float tw = ms / reps;
        hipEventRecord(a, s1); for (int i = 0; i < reps; ++i) { R(s1, i); Wr(s1, i); } hipEventRecord(b, s1); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b); const float ts = ms / reps;
        hipDeviceSynchronize();
        hipEventRecord(a, s1);
        for (int i = 0; i < reps; ++i) { R(s1, i); Wr(s2, i); }
        hipEventRecord(e2, s2); hipStreamWaitEvent(s1, e2, 0);
        hipEventRecord(b, s1); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b); const float tp = ms / reps;
        printf("read %zu MB alone %.1f us (%.0f GB/s) | write %zu MB alone %.1f us (%.0f GB/s) | serial %.1f us | two streams %.1f us (%.0f GB/s combined)\n",
               RB >> 20, tr * 1e3, RB / tr / 1e6, WB >> 20, tw * 1e3, WB / tw / 1e6, ts * 1e3, tp * 1e3, (RB + WB) / tp / 1e6);
    }
    return 0;
}
