// Exact-fp32 MFMA tile helpers shared by the fp32 GEMM kernels (edge_mlp.hip, moment.hip, train.hip):
// v_mfma_f32_32x32x2_f32 over K-tiles of 32 staged in LDS as rows of 36 floats (144 B: conflict-free
// ds_read_b128).  Bit-for-bit an fmaf chain in k order.
#pragma once
#include <hip/hip_runtime.h>

namespace mdno {
namespace f32mma {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BK = 32, LD = BK + 4;

__device__ __forceinline__ void mma4(const float4& a, const float4& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
}

// One K-tile for a wave owning a 64x64 sub-tile (2x2 MFMA tiles)
__device__ __forceinline__ void mma_64x64(f32x16 (&acc)[2][2], const float* __restrict__ ab,
                                          const float* __restrict__ bb) {
#pragma unroll
    for (int t = 0; t < BK / 8; ++t) {
        const float4 a0 = *reinterpret_cast<const float4*>(ab + 8 * t);
        const float4 a1 = *reinterpret_cast<const float4*>(ab + 32 * LD + 8 * t);
        const float4 b0 = *reinterpret_cast<const float4*>(bb + 8 * t);
        const float4 b1 = *reinterpret_cast<const float4*>(bb + 32 * LD + 8 * t);
        mma4(a0, b0, acc[0][0]); mma4(a0, b1, acc[0][1]);
        mma4(a1, b0, acc[1][0]); mma4(a1, b1, acc[1][1]);
    }
}

// One K-tile for a wave owning a 32x64 sub-tile (1x2 MFMA tiles)
__device__ __forceinline__ void mma_32x64(f32x16& acc0, f32x16& acc1, const float* __restrict__ ab,
                                          const float* __restrict__ bb) {
#pragma unroll
    for (int t = 0; t < BK / 8; ++t) {
        const float4 a0 = *reinterpret_cast<const float4*>(ab + 8 * t);
        const float4 b0 = *reinterpret_cast<const float4*>(bb + 8 * t);
        const float4 b1 = *reinterpret_cast<const float4*>(bb + 32 * LD + 8 * t);
        mma4(a0, b0, acc0); mma4(a0, b1, acc1);
    }
}

}  // namespace f32mma
}  // namespace mdno
