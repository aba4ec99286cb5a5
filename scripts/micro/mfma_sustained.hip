// Dev microbenchmark: what the chip SUSTAINS on v_mfma_f32_32x32x16_f16 when nothing but MFMAs runs, and how the
// clock moves when work is added beside them — the context for "0.43 of the 2.5 PF peak" on the hidden GEMM
// (EXPERIMENTS.md §00.5: that kernel sits at the power limit).  One 512-thread workgroup per CU (two waves per SIMD, as
// the GEMMs run), every wave a chain-free stream of MFMAs on register operands.  Variants:
//   0  MFMAs only, four independent accumulators
//   1  + one ds_read_b128 (1 KiB per wave) per MFMA, results folded into an operand (the GEMMs read 0.67 per MFMA)
//   2  + one 1-KiB global_load_lds piece per EVERY MFMAs (12, 6, 4, 3, 2) from a buffer of 2 MiB per XCD-sized footprint
//      (L2-resident) or 64 MiB (beyond L2); the hidden GEMM issues one piece per 4 MFMAs and wave, gemm_bf16.hip one per 4
// Prints time, executed TFLOP/s, and the shader clock = s_memtime ticks of one wave / kernel time.
// hipcc --offload-arch=gfx950 -O3 -o mfma_sustained mfma_sustained.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) unsigned char lds_u8;

template <int VARIANT, int EVERY = 3>
__global__ __launch_bounds__(512) void mfma_kernel(int iters, const unsigned char* __restrict__ src, float* out,
                                                   unsigned long long* ticks, unsigned mask) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    for (int i = tid; i < 32768 / 16; i += 512) reinterpret_cast<float4*>(lds)[i] = make_float4(1e-3f, 2e-3f, 3e-3f, 4e-3f);
    __syncthreads();
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    const unsigned char* lp = lds + ((wave * 64 + lane) * 16) % 32768;
    const unsigned gofs = (unsigned)(((size_t)blockIdx.x * 8 + wave) * 65536 + lane * 16);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j & 3], 0, 0, 0);
            if (VARIANT >= 1) {
                const f16x8 r = *reinterpret_cast<const f16x8*>(lp + ((j * 1024) & 16383));
                a[j & 7] = r[0];
            }
            if (VARIANT == 2 && (j % EVERY) == EVERY - 1)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((gofs + (unsigned)(it * (12 / EVERY) + j / EVERY) * 1024u) & mask)),
                                                 (__attribute__((address_space(3))) void*)(lds + 32768 + wave * 1024), 16, 0, 0);
        }
        if (VARIANT == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][7];
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 7 && tid == 0) ticks[0] = (unsigned long long)(t1 - t0);
}

template <int VARIANT, int EVERY = 3>
static void run(const char* name, int cus, const unsigned char* src, float* out, unsigned long long* ticks, unsigned mask = (64u << 20) - 1) {
    const int iters = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_kernel<VARIANT, EVERY>), hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) mfma_kernel<VARIANT, EVERY><<<cus, 512, 49152>>>(iters, src, out, ticks, mask);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        mfma_kernel<VARIANT, EVERY><<<cus, 512, 49152>>>(iters, src, out, ticks, mask);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long t = 0; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        const double flops = (double)cus * 8 * iters * 12 * 32768.0;
        printf("%-58s %8.1f us  %7.1f TFLOP/s executed  (%.3f of 2.5 PF)  s_memtime %llu ticks\n", name, ms * 1e3,
               flops / ms / 1e9, flops / ms / 1e9 / 2500.0, t);
    }
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    unsigned char* src; float* out; unsigned long long* ticks;
    hipMalloc(&src, 64u << 20); hipMemset(src, 0, 64u << 20); hipMalloc(&out, 64); hipMalloc(&ticks, 64);
    printf("%s, %d CUs, one 512-thread workgroup per CU\n", p.name, cus);
    run<0>("MFMA only", cus, src, out, ticks);
    run<1>("MFMA + 1 ds_read_b128 per MFMA", cus, src, out, ticks);
    run<2, 12>("MFMA + LDS reads + 1 KiB LDS-DMA per 12 MFMA, 64 MiB", cus, src, out, ticks);
    run<2, 6>("MFMA + LDS reads + 1 KiB LDS-DMA per 6 MFMA, 64 MiB", cus, src, out, ticks);
    run<2, 4>("MFMA + LDS reads + 1 KiB LDS-DMA per 4 MFMA, 64 MiB", cus, src, out, ticks);
    run<2, 3>("MFMA + LDS reads + 1 KiB LDS-DMA per 3 MFMA, 64 MiB", cus, src, out, ticks);
    run<2, 2>("MFMA + LDS reads + 1 KiB LDS-DMA per 2 MFMA, 64 MiB", cus, src, out, ticks);
    run<2, 4>("MFMA + LDS reads + 1 KiB LDS-DMA per 4 MFMA, 2 MiB (L2)", cus, src, out, ticks, (2u << 20) - 1);
    run<2, 2>("MFMA + LDS reads + 1 KiB LDS-DMA per 2 MFMA, 2 MiB (L2)", cus, src, out, ticks, (2u << 20) - 1);
    return 0;
}
