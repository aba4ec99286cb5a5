// Shared by the kernels that produce or consume the split operand images (edge_mlp_split.hip owns the layout;
// gemm_bf16.hip's fp32 training products use the fp16 row scales).
#pragma once
#include <hip/hip_runtime.h>

namespace mdno {

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

// Byte offset of element (row, kcol) of plane p in the tiled layout; nkt = K/16.  A plane tile is
// 128 rows of 32 B (16 k); the two 16-B halves of a row are swapped when (row>>3)&1, which spreads
// the ds_read_b128 fragment reads of a 16-lane group over all 16 slots of a 256-B bank row.
__device__ __forceinline__ size_t tiled_off(long long row, int kcol, int nkt, int p) {
    const long long rt = row >> 7;
    const int r = (int)(row & 127), kt = kcol >> 4, c = (kcol >> 3) & 1, e = kcol & 7;
    return ((size_t)((rt * nkt + kt) * 3 + p) << 12) + r * 32 + ((c ^ ((r >> 3) & 1)) << 4) + e * 2;
}

// ---- two fp16 planes (SPLIT_F16, see edge_mlp_split.hip): x = hi + 2^-11 lo', exact for |x| < 65504
constexpr float F16_LO_SCALE = 2048.f, F16_LO_UNSCALE = 1.f / 2048.f, F16_MAX = 65504.f;
// fp16 has 30 binades.  Error of the two-plane form: |x - (hi + 2^-11 lo')| <= max(2^-23 |x|, 2^-36) — relative
// down to |x| = 2^-13, ABSOLUTE below (hi, then lo', go subnormal).  Two rules keep the absolute floor out of
// sight: (a) a WEIGHT row (one output unit) is multiplied by a power of two that puts its largest entry in
// [2^13, 2^14) before the split, and the product's column is multiplied back in the GEMM epilogue — exact
// both ways, so inside a row the floor sits 2^-50 below the largest weight; (b) an ACTIVATION tensor is
// accepted only if it holds at least one value >= F16_ACT_MIN = 2^-10 (floor <= 2^-26 of the tensor's
// largest value); otherwise — and on |x| >= 65504 — the product runs on the bf16 planes.
constexpr float F16_ACT_MIN = 1.f / 1024.f;
constexpr int F16_SEEN_OFF = 64;      // node features (factored conv): "seen" word = range word + 64 ints

// power of two s with max*s in [2^13, 2^14); 1 for a zero / subnormal / non-finite max
__device__ __forceinline__ float f16_row_scale(float row_max) {
    const int ex = (int)((__builtin_bit_cast(unsigned, row_max) >> 23) & 0xffu);
    if (ex == 0 || ex == 255) return 1.f;
    int e = (127 + 13) - ex;
    e = e < -126 ? -126 : (e > 126 ? 126 : e);      // s and 1/s both normal floats
    return __builtin_bit_cast(float, (unsigned)(127 + e) << 23);
}

__device__ __forceinline__ void split2h(float x, _Float16& h, _Float16& l) {
    h = (_Float16)x;
    l = (_Float16)((x - (float)h) * F16_LO_SCALE);
}

// Byte offset of element (row, kcol) of plane p in the two-plane image (same 4 KiB plane tiles and
// swizzle as tiled_off, two planes per k-step instead of three).
__device__ __forceinline__ size_t tiled_off2(long long row, int kcol, int nkt, int p) {
    const long long rt = row >> 7;
    const int r = (int)(row & 127), kt = kcol >> 4, c = (kcol >> 3) & 1, e = kcol & 7;
    return ((size_t)((rt * nkt + kt) * 2 + p) << 12) + r * 32 + ((c ^ ((r >> 3) & 1)) << 4) + e * 2;
}

}  // namespace mdno
