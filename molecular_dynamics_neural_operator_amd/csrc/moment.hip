// Factored conv, destination-side form ("edge moment"): the same function as
//     y_t = act( mean_{e -> t} x_src(e) . reshape(W3 h_e + b3, [Cin,Cout]) + x_t . root + bias )     (graph_kernel.py:194-209)
// with the sum over a destination's in-edges taken BEFORE the contraction with W3:
//     S_t[i][c] = sum_{e -> t} x_src(e)[i] * h_e[c]                  (K1)  one GEMM per destination over its own edges,
//                                                                          contraction over the EDGES: [64 x deg] . [deg x k]
//     z_t[o]    = sum_{i,c} S_t[i][c] * W3[i*64 + o][c] + sum_i s0_t[i] * B3[i][o]      (K2)  one tall GEMM
//                 [R, 64 k + 64] x [64 k + 64, 64], K-sliced;  s0_t = sum_{e -> t} x_src(e), B3 = reshape(b3, [64, 64])
//     y_t       = act( z_t / max(deg_t, 1) + x_t . root + bias )                                   (K3)
// Against the source-side form of rounds 1-3 (Y_j = x_j . W3T per source, M_j = H_j . Y_j^T per source, gather of the
// 256-B messages per destination; removed in round 5) this needs no reverse-edge index and no symmetric graph, has no
// 128-row tiles of a node's edges (a destination's edges are the contraction length: any degree, H read exactly once
// — the source-side form re-read Y_j once per 128 edges of a source: 1.33x the algorithmic bytes at degree 362), writes
// no per-edge partial messages (62 MB per application at N = 504) and ends in a per-node epilogue instead of a
// gather: HBM bytes per application  E k 4 (H) + 2 R 64 k 4 (S out, S in) + 64 k 64 4 (W3)  — 529 MB at N = 504,
// E = 60.6k, k = 1024 against 692 MB measured for the three source-side kernels.
//
// The products run on the 16-bit matrix pipe at fp32 accuracy, every fp32 operand split into planes on the way to LDS:
//   gemm_mode SPLIT_BF16  three bf16 planes (x = hi + mid + lo, exact for any fp32 x), six leading plane products;
//   gemm_mode SPLIT_F16   two fp16 planes (x = hi + lo), three products — half the matrix work, which in K1 and K2 is
//                         ADDED to the stream time rather than hidden under it (EXPERIMENTS 00.10, 00.11).  fp16 has 30
//                         binades, so operands are put high in its range first by exact powers of two (K2: every row
//                         of S by its own maximum, every column of W3R by its own; K1: the feature rows by the largest
//                         |feature| among the destination's own neighbours — no input of these is out of range —, H
//                         by 2^5, and a workgroup whose staged |H| leave [2^-7, 2047) reruns its destination on the
//                         bf16 planes: every decision depends on that destination's own edges only);
//   gemm_mode F32         the fp32 MFMA (moment_f32_kernel, project_f32_kernel): reference arithmetic.
//   K1  moment_kernel     workgroup = (destination t, 256 of the k hidden units); stage = 16 edges: H rows (fp32,
//                         k-tiled image written by the hidden GEMM, streamed once per application, non-temporal but
//                         for the first 64 MiB, kept in the Infinity Cache) and the neighbours' fp32 feature rows
//                         (gathered from L2) are split into planes on the way to LDS, edge-major, i.e. with the
//                         contraction index SLOWEST, and the MFMA fragments come out of gfx950's transposing read
//                         ds_read_b64_tr_b16; register prefetch of the next stage under the MFMAs.  One more workgroup
//                         per destination sums the neighbours' features (s0, the b3 term).
//   K2  project_kernel / project_f16_kernel
//                         workgroup = (256 destinations, 1/128 of the 64 k contraction): S tiles fp32 -> planes on the
//                         fly, W3R's likewise (bf16) or pre-split (fp16); partial sums per K slice.
//   K3  finish_kernel     per destination: K slices added in slice order, root / bias / mean / ReLU.
// Fixed summation orders everywhere: a destination's result depends on its own edges only (bitwise the same alone or
// in any batch), no float atomics.
#include <type_traits>

#include "kernels.h"
#include "mfma_f32.h"
#include "split_layout.h"

namespace mdno {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// (a, b) -> packed bf16 pair (one v_cvt_pk_bf16_f32) and the pair's values back in fp32
__device__ __forceinline__ unsigned pack_bf16(float a, float b, float& fa, float& fb) {
    const f32x2 v = {a, b};
    const bf16x2 p = __builtin_convertvector(v, bf16x2);
    const unsigned u = __builtin_bit_cast(unsigned, p);
    fa = __builtin_bit_cast(float, u << 16);
    fb = __builtin_bit_cast(float, u & 0xffff0000u);
    return u;
}

// four fp32 -> 3 x four bf16 (hi, mid, lo), 8 bytes per plane at dst + p * plane_bytes
__device__ __forceinline__ void split_store4(const float4 v, unsigned char* dst, int plane_bytes) {
    float h0, h1, h2, h3, m0, m1, m2, m3, t0, t1;
    uint2 hi, mid, lo;
    hi.x = pack_bf16(v.x, v.y, h0, h1);
    hi.y = pack_bf16(v.z, v.w, h2, h3);
    const float r0 = v.x - h0, r1 = v.y - h1, r2 = v.z - h2, r3 = v.w - h3;
    mid.x = pack_bf16(r0, r1, m0, m1);
    mid.y = pack_bf16(r2, r3, m2, m3);
    lo.x = pack_bf16(r0 - m0, r1 - m1, t0, t1);
    lo.y = pack_bf16(r2 - m2, r3 - m3, t0, t1);
    *reinterpret_cast<uint2*>(dst) = hi;
    *reinterpret_cast<uint2*>(dst + plane_bytes) = mid;
    *reinterpret_cast<uint2*>(dst + 2 * plane_bytes) = lo;
}

// the six leading plane products of (a0 + a1 + a2)(b0 + b1 + b2), smallest first
#define MDNO_MMA6(A, B, ACC)                                                        \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[1], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[2], B[0], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[2], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[0], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[1], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[0], ACC, 0, 0, 0);

// ---- two fp16 planes (gemm_mode SPLIT_F16; split_layout.h): x = hi + lo with hi = fp16(x), lo = fp16(x - hi), the lo
// plane NOT scaled here, so that the three leading products share one accumulator (K1 has no registers for a second
// one at three workgroups per CU).  |x - (hi + lo)| <= max(2^-23 |x|, 2^-25): the absolute floor is kept out of sight by
// putting the operands high in fp16's range first, with exact powers of two that are taken out again afterwards —
// K2: every row of S by its own maximum (K1 records it) and every column of W3R by its own, to [2^13, 2^14): floor
// 2^-38 of the row's / column's largest entry; K1: the feature rows by the largest |feature| among the destination's
// neighbours (likewise), H by 2^5 (floor 2^-30; a workgroup whose H holds a value >= 2047 or none >= 2^-7 redoes its
// destination on the bf16 planes — a decision that depends on that destination's own edges only).
// (a, b) -> packed fp16 pair (one v_cvt_pk_f16_f32 on gfx950) and the pair's values back in fp32
__device__ __forceinline__ unsigned pack_f16(float a, float b, float& fa, float& fb) {
    const f32x2 v = {a, b};
    const f16x2 p = __builtin_convertvector(v, f16x2);
    fa = (float)p.x;
    fb = (float)p.y;
    return __builtin_bit_cast(unsigned, p);
}

// four fp32 (already scaled) -> 2 x four fp16 (hi, lo), 8 bytes per plane at dst + p * plane_bytes
__device__ __forceinline__ void split2_store4(const float4 v, unsigned char* dst, int plane_bytes) {
    float h0, h1, h2, h3, t0, t1;
    uint2 hi, lo;
    hi.x = pack_f16(v.x, v.y, h0, h1);
    hi.y = pack_f16(v.z, v.w, h2, h3);
    lo.x = pack_f16(v.x - h0, v.y - h1, t0, t1);
    lo.y = pack_f16(v.z - h2, v.w - h3, t0, t1);
    *reinterpret_cast<uint2*>(dst) = hi;
    *reinterpret_cast<uint2*>(dst + plane_bytes) = lo;
}

// the three leading plane products of (a0 + a1)(b0 + b1), smallest first
#define MDNO_MMA3H(A, B, ACC)                                                       \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1], B[0], ACC, 0, 0, 0);         \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], B[1], ACC, 0, 0, 0);         \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], B[0], ACC, 0, 0, 0);

// ---------------------------------------------------------------- W3 [64*64, k] -> W3R tiled [64k/32][64 o][32]
// W3R[kappa][o] = W3[(i*64 + o)*k + c] with kappa = i*k + c (and B3[i][o] at kappa = 64 k + i): the B operand of K2,
// one 8 KiB run per 32 kappa
__global__ __launch_bounds__(256) void w3_moment_kernel(const float* __restrict__ w3, const float* __restrict__ b3, int k,
                                                        float* __restrict__ w3r) {
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;      // over (kappa tile, o, kappa & 31)
    const long long total = (long long)(64 * k + 64) * 64;
    if (id >= total) return;
    const int kl = (int)(id & 31), o = (int)((id >> 5) & 63);
    const long long kappa = (id >> 11) * 32 + kl;
    if (kappa >= (long long)64 * k) {      // the s0 rows: B3[i][o] = b3[i*64 + o]
        w3r[id] = b3[(kappa - (long long)64 * k) * 64 + o];
        return;
    }
    const int i = (int)(kappa / k), c = (int)(kappa - (long long)i * k);
    w3r[id] = w3[((size_t)i * 64 + o) * k + c];
}

// W3R on two fp16 planes (gemm_mode SPLIT_F16): column o (one output channel: 64 k + 64 entries) times the power of two
// that puts its largest entry in [2^13, 2^14), split; tile [kappa/32][plane][64 o][32 kappa] of fp16, 8 KiB like the fp32
// tile.  colinv[o] = the power of two that takes the scale out again (K3).
__global__ __launch_bounds__(256) void w3_colmax_kernel(const float* __restrict__ w3r, long long ntiles, int* __restrict__ colmax_bits) {
    __shared__ int red[4][64];
    const int o = threadIdx.x & 63, sub = threadIdx.x >> 6;
    float m = 0.f;
    for (long long t = (long long)blockIdx.x * 4 + sub; t < ntiles; t += (long long)gridDim.x * 4) {
        const float4* row = reinterpret_cast<const float4*>(w3r + (t * 64 + o) * 32);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 v = row[j];
            // (not fmaxf: a NaN must win AND stay — once m is NaN no comparison below replaces it — so that a column with
            // a non-finite weight gets scale 1 (f16_row_scale) and the value itself reaches the output through the planes,
            // as in the other GEMM modes)
            const float a = fabsf(v.x), b = fabsf(v.y), c = fabsf(v.z), d = fabsf(v.w);
            m = (a > m || a != a) ? a : m; m = (b > m || b != b) ? b : m;
            m = (c > m || c != c) ? c : m; m = (d > m || d != d) ? d : m;
        }
    }
    red[sub][o] = __builtin_bit_cast(int, m);      // non-negative floats (and NaN above Inf) order like their bits
    __syncthreads();
    if (sub == 0) {
        int v = red[0][o];
        for (int u = 1; u < 4; ++u) v = red[u][o] > v ? red[u][o] : v;
        atomicMax(colmax_bits + o, v);
    }
}

__global__ __launch_bounds__(256) void w3_planes_f16_kernel(const float* __restrict__ w3r, long long total,
                                                            const int* __restrict__ colmax_bits, _Float16* __restrict__ w3h,
                                                            float* __restrict__ colinv) {
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;      // over (kappa tile, o, kappa & 31), as w3r
    if (id < 64) colinv[id] = 1.f / f16_row_scale(__builtin_bit_cast(float, colmax_bits[id]));
    if (id >= total) return;
    const int kl = (int)(id & 31), o = (int)((id >> 5) & 63);
    const long long tile = id >> 11;
    const float v = w3r[id] * f16_row_scale(__builtin_bit_cast(float, colmax_bits[o]));
    const _Float16 h = (_Float16)v;
    w3h[(tile * 2 + 0) * 2048 + o * 32 + kl] = h;
    w3h[(tile * 2 + 1) * 2048 + o * 32 + kl] = (_Float16)(v - (float)h);
}

// ---------------------------------------------------------------- destinations by decreasing degree
// One workgroup per S chunk of CH destinations: order[chunk*CH + rank] = index inside the chunk of the destination
// with that rank (degree descending, ties by index).  Once per graph: all workgroups of a K1 launch are resident
// together, and a CU then gets one destination from every quarter of the sorted list instead of four of any size.
template <int CH>
__global__ __launch_bounds__(CH) void degree_order_kernel(const int* __restrict__ row_ptr, int num_rows,
                                                          int* __restrict__ order) {
    __shared__ __attribute__((aligned(16))) int key[CH];      // degree * CH + (CH - 1 - index): all distinct
    const int base = blockIdx.x * CH, t = threadIdx.x;
    const int cnt = num_rows - base < CH ? num_rows - base : CH;
    int dg = t < cnt ? row_ptr[base + t + 1] - row_ptr[base + t] : 0;
    dg = dg < (1 << 20) ? dg : (1 << 20);      // (the key must fit an int; beyond that the order does not matter)
    const int mine = t < cnt ? dg * CH + (CH - 1 - t) : -1;
    key[t] = mine;
    __syncthreads();
    if (t >= cnt) return;
    int rank = 0;
#pragma unroll 4
    for (int u = 0; u < CH; u += 4) {
        const int4 k4 = *reinterpret_cast<const int4*>(&key[u]);
        rank += (k4.x > mine) + (k4.y > mine) + (k4.z > mine) + (k4.w > mine);
    }
    order[base + rank] = t;
}

// ---------------------------------------------------------------- K1: S_t = X_N(t)^T . H_t
// k-tiles (32 kappa) of a destination's row of the S image: 64 k / 32 for S_t itself + 2 for s0_t (kappa = 64 k + i)
__host__ __device__ constexpr size_t moment_nkt(int K) { return (size_t)64 * K / 32 + 2; }
constexpr int MO_EDGES = 16;                 // edges per stage = one MFMA k-step
// The first bytes of H are loaded with the default cache policy, the rest non-temporally: the 2 x depth applications of
// a forward stream the same H, and what of it stays in the 256 MiB Infinity Cache between two of them is read faster
// (scripts/micro/mall_partial_residency.hip: 248 MB at 5.5 TB/s all non-temporal, 6.3 TB/s with 64 MiB kept — and no
// further gain from more, while S (written by K1, read back by K2 out of the same cache) and W3R need their share).
constexpr size_t kMomentCachedBytes = (size_t)64 << 20;
constexpr int MO_CQ = 256;                   // hidden units per workgroup
constexpr int MO_HROW = MO_CQ * 2 + 64;      // LDS bytes per edge row of an H plane (64 B of padding: the four rows of a
                                             // transposing read's block fall into four different 64-B bank quarters)
constexpr int MO_XROW = 64 * 2 + 64;         // the same for the 64 feature columns
constexpr int MO_HPLANE = MO_EDGES * MO_HROW, MO_XPLANE = MO_EDGES * MO_XROW;
constexpr int MO_LDS = 3 * MO_HPLANE + 3 * MO_XPLANE;      // 36,864 B

// One more workgroup per destination in K1's launch: s0_t[i] = sum_{e -> t} x_src(e)[i] (the b3 term's operand: the last
// MLP layer's bias seen through the summed neighbours), stored as kappa = 64 K + i of the same image, so that K2
// multiplies it with B3 like any other slice.  16 chains x 16 lanes (4 features each), four edges in flight per chain,
// chains added in order: a fixed order.
// the exponent e of a power of two 2^e held in a normal float
__device__ __forceinline__ int f32_exponent(float p2) { return (int)((__builtin_bit_cast(unsigned, p2) >> 23) & 0xffu) - 127; }

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

__device__ __forceinline__ void moment_s0(const int* __restrict__ row_ptr, const int* __restrict__ src,
                                          const float* __restrict__ x, float* __restrict__ S, int K, int t, int tl,
                                          float* __restrict__ rowmax_slot, float scale) {
    __shared__ float4 sred[16][16];
    const int tid = threadIdx.x;
    const int beg = row_ptr[t], end = row_ptr[t + 1];
    const int es = tid >> 4, q = tid & 15;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = beg + es; p < end; p += 64) {
        int sj[4];
        bool on[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            on[u] = p + 16 * u < end;
            sj[u] = on[u] ? src[p + 16 * u] : 0;
        }
        float4 g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            g[u] = on[u] ? *reinterpret_cast<const float4*>(x + (size_t)sj[u] * 64 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) { s0.x += g[u].x; s0.y += g[u].y; s0.z += g[u].z; s0.w += g[u].w; }
    }
    sred[es][q] = s0;
    __syncthreads();
    float4 s0r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (es == 0) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < 16; ++c) { const float4 v = sred[c][q]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
        a.x *= scale; a.y *= scale; a.z *= scale; a.w *= scale;      // (fp16 planes: the row's power of two, as K1's blocks)
        s0r = a;
        const int i = 4 * q;       // features 4q..4q+3: k-tile 64K/32 + (i >> 5), columns i & 31 ..
        float* d = S + ((size_t)(tl >> 7) * moment_nkt(K) + (size_t)64 * K / 32 + (i >> 5)) * 4096 + (tl & 127) * 32 + (i & 31);
        *reinterpret_cast<float4*>(d) = a;
    }
    if (rowmax_slot != nullptr && tid < 64) {      // wave 0: its first 16 lanes hold s0
        float m = es == 0 ? fmaxf(fmaxf(fabsf(s0r.x), fabsf(s0r.y)), fmaxf(fabsf(s0r.z), fabsf(s0r.w))) : 0.f;
        m = wave_max(m);
        if (tid == 0) *rowmax_slot = m;
    }
}

// Grid: (destination within the chunk, visited by decreasing degree) x (k / 256).  S chunk layout: the fp32 k-tiled
// image K2 streams, [node/128][64k/32][128][32] with kappa = i*k + c.
// F16 (gemm_mode SPLIT_F16): the stage loop runs on two fp16 planes (three products instead of six: the matrix work of
// this kernel is added to its stream time, not hidden under it — EXPERIMENTS 00.10, 00.11).  The feature rows are scaled
// by the power of two that puts the largest |x| among THIS destination's neighbours in [2^13, 2^14) (xm: every node's
// largest |feature|, written by K3 — by row_absmax_kernel for a forward's first application — and gathered over the
// edge list before the first stage: no x is out of range, floor 2^-38 of that maximum) and H by 2^5; the S image of this
// mode holds 2^(ex + 5) S — K2 scales every row by its own maximum anyway and takes the row's ex + 5 (recorded next to
// the row's maxima) out with its own scale.  The workgroup checks afterwards what it has staged: if an |H| reached 2047
// or none reached 2^-7 (floor 2^-30: split_layout.h's rule for activations — at most 2^-23 of the largest value), it runs
// the loop again on the three bf16 planes (of the same scaled operands: exact scalings).  Both decisions are the
// workgroup's own — one destination's edges —, so a destination's bits are the same alone or in any batch.
// counters (MomentWs::counters; a rollout plan zeroes them per run and reads them back: mdno_rollout_plan_fallback_counts)
constexpr int MOMENT_CNT_K1_RERUN = 0;      // K1 workgroups (destination x 256 hidden units) that ran their stage loop again on bf16 planes
constexpr int MOMENT_CNT_X_PLAIN = 1;       // destinations whose neighbours have no feature in [2^-100, 2^100): unscaled operands
constexpr int MO_F16_PRE_H_EXP = 5;      // H times 2^5: a workgroup's largest |H| may lie in [2^-7, 2047] (typical: 1 .. 10)
constexpr float MO_F16_PRE_H = 32.f, MO_F16_H_LIM = F16_MAX / 32.f, MO_F16_H_MIN = 0.25f / 32.f;

template <bool F16>
__global__ __launch_bounds__(256, 3) void moment_kernel(const float* __restrict__ Hm,
                                                        const int* __restrict__ row_ptr, const int* __restrict__ src,
                                                        const int* __restrict__ order, float* __restrict__ S, int K,
                                                        int row0, int cnt, const float* __restrict__ x, int cache_e,
                                                        float* __restrict__ rowmax, const float* __restrict__ xm,
                                                        int* __restrict__ counters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[MO_LDS];
    __shared__ float wg_red[12];
    // workgroup ids b, b+8, b+16, .. share an XCD: a destination's k/256 column blocks (and its s0 workgroup) run there
    // back to back, so that the neighbours' feature rows they all gather come through that L2 once (at N = 50,000 the
    // features are 12.8 MB — beyond an XCD's 4 MiB — and each block fetched them again through the fabric)
    const int nq = (K + MO_CQ - 1) / MO_CQ + 1;
    const int xcd = blockIdx.x & 7, rr = blockIdx.x >> 3;
    const int cq = rr % nq, ti = (rr / nq) * 8 + xcd;      // ti: rank of the destination (decreasing degree) in the chunk
    if (ti >= cnt) return;
    const int tl = order[row0 + ti];               // index inside this launch's chunk of destinations
    const int t = row0 + tl;
    const int beg = row_ptr[t], end = row_ptr[t + 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nkt = K >> 5;
    // rowmax: nq maxima per destination of the chunk and, behind them, the exponent ex + 5 of the row's scale
    float xsc = 1.f, hsc = 1.f;      // (F16) the powers of two the neighbours' features and H are multiplied with
    bool x_plain = false;
    if constexpr (F16) {
        float m = 0.f;
        for (int e = beg + tid; e < end; e += 256) m = fmaxf(m, xm[src[e]]);
        m = wave_max(m);
        if (lane == 0) wg_red[wave] = m;
        __syncthreads();
        m = fmaxf(fmaxf(wg_red[0], wg_red[1]), fmaxf(wg_red[2], wg_red[3]));
        // (no neighbour feature, or one beyond 2^+-100: operands as they are — S would overflow scaled —, and the bf16
        // planes, exact for any x; a property of the destination, so all its column blocks agree on the row's scale)
        x_plain = !(m >= 0x1p-100f && m < 0x1p100f);
        xsc = x_plain ? 1.f : __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, f16_row_scale(m))));
        hsc = x_plain ? 1.f : MO_F16_PRE_H;
        __syncthreads();      // (wg_red is free again)
        if (cq == 0 && tid == 0) {
            rowmax[(size_t)tl * (nq + 1) + nq] = x_plain ? 0.f : (float)(f32_exponent(xsc) + MO_F16_PRE_H_EXP);
            if (x_plain) atomicAdd(counters + MOMENT_CNT_X_PLAIN, 1);
        }
    }
    if (cq == nq - 1) {
        moment_s0(row_ptr, src, x, S, K, t, tl, rowmax ? rowmax + (size_t)tl * (nq + 1) + cq : nullptr, xsc * hsc);
        return;
    }
    // ---- staging roles
    // H: thread (edge er = tid >> 4, column group cc = tid & 15) loads four float4: hidden units cq*256 + u*64 + cc*4 ..
    const int er = tid >> 4, cc = tid & 15;
    // X: thread (edge xe = tid >> 4, xs = tid & 15): four fp32 features 4*xs .. of the edge's source row (from L2: the
    // features of a member are 129 KB), split into the planes on the way to LDS like H
    const int xe = tid >> 4, xs = tid & 15;
    auto h_ptr = [&](int e, int u) {      // element (edge e, hidden unit cq*256 + u*64 + cc*4) of the k-tiled H
        const int c = cq * MO_CQ + u * 64 + cc * 4;
        return Hm + ((size_t)(e >> 7) * nkt + (c >> 5)) * 4096 + (e & 127) * 32 + (c & 31);
    };
    float4 rh[4], rx;
    int sidx = 0;      // source node of this thread's X edge in the stage being loaded
#define MO_NT(DST, P)                                                                               \
    {                                                                                               \
        const f32x4 t_ = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P));             \
        DST = make_float4(t_.x, t_.y, t_.z, t_.w);                                                  \
    }
    auto load_stage = [&](int e0) {       // edges e0 .. e0+15 of this row -> registers (zeros past the end)
        const int e = e0 + er;
        if (e0 + MO_EDGES <= cache_e) {    // the part of H kept in the Infinity Cache (moment_conv): default-policy loads
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                rh[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < end && cq * MO_CQ + u * 64 < K) rh[u] = *reinterpret_cast<const float4*>(h_ptr(e, u));
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {      // (u: the 64 hidden units of wave u; none past k — k % 64 == 0)
                rh[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < end && cq * MO_CQ + u * 64 < K) MO_NT(rh[u], h_ptr(e, u))
            }
        }
        rx = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e0 + xe < end) rx = *reinterpret_cast<const float4*>(x + (size_t)sidx * 64 + 4 * xs);
    };
    // ---- fragment addresses (ds_read_b64_tr_b16): 16-lane group gq, lane = 4*qq + pp inside it reads row 8*(gq>>1) + qq
    // (+4 for the second half), 8 B at column 16*(gq&1) + 4*pp of the block; lane i of the group receives column i
    const int gq = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    const int frow = 8 * (gq >> 1) + qq, fcol = 16 * (gq & 1) + 4 * pp;
    const unsigned char* hb = lds + frow * MO_HROW + (wave * 64 + fcol) * 2;                  // + cb*64 B, + plane
    const unsigned char* xb = lds + 3 * MO_HPLANE + frow * MO_XROW + fcol * 2;               // + ih*64 B, + plane
    auto tr_frag = [&](const unsigned char* p_, int row_bytes) {
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p_));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p_ + 4 * row_bytes));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    f32x16 acc[2][2];      // [feature half ih][hidden block cb of this wave's 64]
    const int stages = (end - beg + MO_EDGES - 1) / MO_EDGES;
    const bool wave_live = cq * MO_CQ + wave * 64 < K;      // this wave's 64 hidden units exist
    float hmax = 0.f;      // (F16) the largest |H| this thread has staged
    auto max4 = [](const float4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); };
    auto pre = [](const float4 v, float sc) { return make_float4(v.x * sc, v.y * sc, v.z * sc, v.w * sc); };
    // workgroup-wide maxima of two per-thread values in one round (wg_red is touched by nothing else)
    auto wg_max2 = [&](float a, float b, float& ra, float& rb) {
        a = wave_max(a); b = wave_max(b);
        if (lane == 0) { wg_red[wave] = a; wg_red[4 + wave] = b; }
        __syncthreads();
        ra = fmaxf(fmaxf(wg_red[0], wg_red[1]), fmaxf(wg_red[2], wg_red[3]));
        rb = fmaxf(fmaxf(wg_red[4], wg_red[5]), fmaxf(wg_red[6], wg_red[7]));
    };

    // the stage loop on two fp16 planes (HALF) or three bf16 planes
    auto run = [&](auto half_tag) {
        constexpr bool HALF = decltype(half_tag)::value;
        auto store_stage = [&]() {
            if constexpr (HALF) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    hmax = fmaxf(hmax, max4(rh[u]));
                    split2_store4(pre(rh[u], hsc), lds + er * MO_HROW + (u * 64 + cc * 4) * 2, MO_HPLANE);
                }
                split2_store4(pre(rx, xsc), lds + 3 * MO_HPLANE + xe * MO_XROW + xs * 8, MO_XPLANE);
            } else if constexpr (F16) {      // the rerun: the same scaled operands on three bf16 planes
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    split_store4(pre(rh[u], hsc), lds + er * MO_HROW + (u * 64 + cc * 4) * 2, MO_HPLANE);
                split_store4(pre(rx, xsc), lds + 3 * MO_HPLANE + xe * MO_XROW + xs * 8, MO_XPLANE);
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    split_store4(rh[u], lds + er * MO_HROW + (u * 64 + cc * 4) * 2, MO_HPLANE);
                split_store4(rx, lds + 3 * MO_HPLANE + xe * MO_XROW + xs * 8, MO_XPLANE);
            }
        };
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        if (stages == 0) return;
        sidx = 0;
        if (beg + xe < end) sidx = src[beg + xe];
        load_stage(beg);
        if (beg + MO_EDGES + xe < end) sidx = src[beg + MO_EDGES + xe];
        store_stage();
        __syncthreads();
        for (int st = 0; st < stages; ++st) {
            const int e_next = beg + (st + 1) * MO_EDGES;
            if (st + 1 < stages) {
                load_stage(e_next);
                if (e_next + MO_EDGES + xe < end) sidx = src[e_next + MO_EDGES + xe];
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the prefetch above the MFMAs
            if (wave_live) {
                if constexpr (HALF) {
                    f16x8 a[2][2], b[2];
#pragma unroll
                    for (int ih = 0; ih < 2; ++ih)
#pragma unroll
                        for (int p = 0; p < 2; ++p)
                            a[ih][p] = __builtin_bit_cast(f16x8, tr_frag(xb + p * MO_XPLANE + ih * 64, MO_XROW));
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
                        for (int p = 0; p < 2; ++p) b[p] = __builtin_bit_cast(f16x8, tr_frag(hb + p * MO_HPLANE + cb * 64, MO_HROW));
                        MDNO_MMA3H(a[0], b, acc[0][cb])
                        MDNO_MMA3H(a[1], b, acc[1][cb])
                    }
                } else {
                    bf16x8 a[2][3], b[3];
#pragma unroll
                    for (int ih = 0; ih < 2; ++ih)
#pragma unroll
                        for (int p = 0; p < 3; ++p)
                            a[ih][p] = __builtin_bit_cast(bf16x8, tr_frag(xb + p * MO_XPLANE + ih * 64, MO_XROW));
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
                        for (int p = 0; p < 3; ++p) b[p] = __builtin_bit_cast(bf16x8, tr_frag(hb + p * MO_HPLANE + cb * 64, MO_HROW));
                        MDNO_MMA6(a[0], b, acc[0][cb])
                        MDNO_MMA6(a[1], b, acc[1][cb])
                    }
                }
            }
            if (st + 1 < stages) {
                __syncthreads();
                store_stage();
                __syncthreads();
            }
        }
    };
    if constexpr (F16) {
        run(std::true_type{});
        float mh, d;
        wg_max2(hmax, 0.f, mh, d);
        if (stages != 0 && (x_plain || !(mh < MO_F16_H_LIM && mh >= MO_F16_H_MIN))) {
            if (tid == 0) atomicAdd(counters + MOMENT_CNT_K1_RERUN, 1);
            __syncthreads();      // (every wave has read wg_red; the stage buffers were free already)
            run(std::false_type{});
        }
    } else {
        run(std::false_type{});
    }
#undef MO_NT
    // ---- S_t[i][c] -> the k-tiled image: kappa = i*K + c, tile (t >> 7, kappa >> 5), row t & 127, column kappa & 31
    // (a lane holds one hidden unit c = column l31 of the block, 16 feature rows: 128-B runs per half wave)
    float smax = 0.f;      // the largest |S_t[i][c]| this thread holds
    if (wave_live) {
        const int l31 = lane & 31, h = lane >> 5;
        float* Sb = S + (size_t)(tl >> 7) * moment_nkt(K) * 4096 + (tl & 127) * 32 + l31;
#pragma unroll
        for (int ih = 0; ih < 2; ++ih)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const int c0 = cq * MO_CQ + wave * 64 + cb * 32;      // multiple of 32
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = ih * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    const size_t kt = ((size_t)i * K + c0) >> 5;
                    const float v = acc[ih][cb][e];
                    Sb[kt * 4096] = v;
                    smax = fmaxf(smax, fabsf(v));
                }
            }
    }
    // K2 on fp16 planes scales a row of S by its own maximum: this block's share of it, behind the stores
    if (rowmax != nullptr) {
        __syncthreads();      // (wg_red is free again)
        float m, d;
        wg_max2(smax, 0.f, m, d);
        if (tid == 0) rowmax[(size_t)tl * (nq + 1) + cq] = m;
    }
}

// ---------------------------------------------------------------- K1, exact fp32 (gemm_mode F32)
// The same workgroup shape and the same S image, on v_mfma_f32_32x32x2_f32 (an fp32 fmaf chain over a destination's
// edges in edge order): H is the same k-tiled fp32 image (the fp32 hidden GEMM writes it too, edge_mlp.hip), staged as
// fp32 rows of 256 + 32 floats (the two k rows of an MFMA step fall into different bank halves), the neighbours' feature
// rows next to it.  Not a tuned path — F32 is the reference-arithmetic mode, ~16x the matrix-pipe time of the split
// modes — but the same formulation on any destination-sorted graph.
constexpr int MF_HLD = MO_CQ + 32, MF_XLD = 64 + 32;

__global__ __launch_bounds__(256) void moment_f32_kernel(const float* __restrict__ Hm, const int* __restrict__ row_ptr,
                                                         const int* __restrict__ src, const int* __restrict__ order,
                                                         float* __restrict__ S, int K, int row0, int cnt,
                                                         const float* __restrict__ x) {
    __shared__ __attribute__((aligned(16))) float hs[MO_EDGES * MF_HLD];
    __shared__ __attribute__((aligned(16))) float xs_[MO_EDGES * MF_XLD];
    const int nq = (K + MO_CQ - 1) / MO_CQ + 1;
    const int xcd = blockIdx.x & 7, rr = blockIdx.x >> 3;
    const int cq = rr % nq, ti = (rr / nq) * 8 + xcd;
    if (ti >= cnt) return;
    const int tl = order[row0 + ti];
    const int t = row0 + tl;
    if (cq == nq - 1) {
        moment_s0(row_ptr, src, x, S, K, t, tl, nullptr, 1.f);
        return;
    }
    const int beg = row_ptr[t], end = row_ptr[t + 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int er = tid >> 4, cc = tid & 15;       // staging: edge er of the stage, columns u*64 + cc*4 (H) / cc*4 (X)
    const int l31 = lane & 31, h = lane >> 5;
    const bool wave_live = cq * MO_CQ + wave * 64 < K;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int e0 = beg; e0 < end; e0 += MO_EDGES) {
        const int e = e0 + er;
        float4 rh[4], rx = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            rh[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int c = cq * MO_CQ + u * 64 + cc * 4;
            if (e < end && c < K)
                rh[u] = *reinterpret_cast<const float4*>(Hm + ((size_t)(e >> 7) * (K >> 5) + (c >> 5)) * 4096 + (e & 127) * 32 + (c & 31));
        }
        if (e < end) rx = *reinterpret_cast<const float4*>(x + (size_t)src[e] * 64 + 4 * cc);
        __syncthreads();      // (the previous stage's fragment reads are done)
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<float4*>(&hs[er * MF_HLD + u * 64 + cc * 4]) = rh[u];
        *reinterpret_cast<float4*>(&xs_[er * MF_XLD + cc * 4]) = rx;
        __syncthreads();
        if (wave_live) {
#pragma unroll
            for (int ks = 0; ks < MO_EDGES / 2; ++ks) {      // MFMA k-step = edges 2 ks + h
                const float a0 = xs_[(2 * ks + h) * MF_XLD + l31], a1 = xs_[(2 * ks + h) * MF_XLD + 32 + l31];
                const float b0 = hs[(2 * ks + h) * MF_HLD + wave * 64 + l31], b1 = hs[(2 * ks + h) * MF_HLD + wave * 64 + 32 + l31];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
    }
    if (!wave_live) return;
    float* Sb = S + (size_t)(tl >> 7) * moment_nkt(K) * 4096 + (tl & 127) * 32 + l31;
#pragma unroll
    for (int ih = 0; ih < 2; ++ih)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int c0 = cq * MO_CQ + wave * 64 + cb * 32;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = ih * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const size_t kt = ((size_t)i * K + c0) >> 5;
                Sb[kt * 4096] = acc[ih][cb][e];
            }
        }
}

// ---------------------------------------------------------------- K2: z = S . W3R, K-sliced
// Workgroup (256 destinations = two row tiles of the S image, slice): partial[slice][row][64] = S[rows][kappa in slice] . W3R[kappa][:].
// fp32 K-tiles of both operands (S rows, W3R rows) are split into three bf16 planes on the fly and staged in
// XOR-swizzled 64-B LDS rows; six plane products per pair.  PJ_SLICES is a constant and a slice's k-tiles a function of k alone: the association of a destination's
// sum does not depend on the launch.  The last slice also takes the two s0 k-tiles (x B3).
constexpr int PJ_SLICES = 128;
// 256 destinations per workgroup: two 128-row tiles of the S image against ONE W3R tile —
                                   // a CU streams at ~23.5 GB/s whatever the source, and with 128 rows a third of what
                                   // went through it was W3R (L2 hits): 34 us; (64 rows: 43 us)
constexpr int PJ_B_PLANE = 64 * 64;

template <int PJ_ROWS>
__global__ __launch_bounds__(PJ_ROWS * 2) void project_kernel(const float* __restrict__ S, const float* __restrict__ w3r,
                                                         float* __restrict__ part, int K, int cnt, int row0,
                                                         long long part_stride) {
    constexpr int PJ_A_PLANE = PJ_ROWS * 64, PJ_B_BASE = 3 * PJ_A_PLANE, NT = PJ_ROWS / 128;
    constexpr int RQ = PJ_ROWS / 4;       // staging: thread (srow < RQ, 4 columns) takes A rows srow + RQ j and B rows srow (+ RQ)
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * PJ_A_PLANE + 3 * PJ_B_PLANE];
    // The row groups of one K slice stream the same W3R tiles: they sit on ONE XCD (workgroup ids b, b+8, .. share an
    // XCD) next to each other, so that a slice's 8 KiB tiles come through that XCD's L2 once.
    const int ngrp = gridDim.x / PJ_SLICES;
    const int xcd = blockIdx.x & 7, rr = blockIdx.x >> 3;
    const int rg = rr % ngrp, slice = (rr / ngrp) * 8 + xcd;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // rows 32*wave .. +31 of the workgroup's
    const int l31 = lane & 31, h = lane >> 5;
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    const int nkt = (int)moment_nkt(K), per = (nkt - 2) / PJ_SLICES;
    const int kt0 = slice * per, nk = per + (slice == PJ_SLICES - 1 ? 2 : 0);
    const int first = rg * PJ_ROWS;
    const int rows_here = cnt - first < PJ_ROWS ? cnt - first : PJ_ROWS;
    const int live = (rows_here + 31) >> 5;       // 32-row groups holding at least one destination
    // row r of the workgroup = row r & 127 of tile NT rg + (r >> 7)
    auto a_ptr = [&](int r) {
        return S + ((size_t)(NT * rg + (r >> 7)) * nkt + kt0) * 4096 + (r & 127) * 32 + scol;
    };
    // rows past the last destination (the tail of its 32-row group is multiplied but never stored, the groups behind it
    // are not multiplied: wave >= live) re-read the LAST destination's row — written by K1, always inside the image — so
    // that every load and LDS store below is unconditional (straight-line code whose vmcnt waits the compiler can count)
    // and nothing K1 did not write goes through the matrix pipe
    auto a_row = [&](int r) { return r < rows_here ? r : rows_here - 1; };
    const float* A0 = a_ptr(a_row(srow));
    const float* A1 = a_ptr(a_row(srow + RQ));
    const float* A2 = a_ptr(a_row(srow + 2 * RQ));
    const float* A3 = a_ptr(a_row(srow + 3 * RQ));
    const float* Bg = w3r + (size_t)kt0 * 2048 + srow * 32 + scol;
    // TWO K-tiles in flight per thread (register sets P and Q, alternating): with one, a workgroup — one per CU at one
    // member — had 40 KiB on its way at a time and the kernel ran at what one round trip per tile gives
    struct Tile { float4 a0, a1, a2, a3, b0, b1; };
    auto load = [&](Tile& r, int kt) {
        r.a0 = *reinterpret_cast<const float4*>(A0 + (size_t)kt * 4096);
        r.a1 = *reinterpret_cast<const float4*>(A1 + (size_t)kt * 4096);
        r.a2 = *reinterpret_cast<const float4*>(A2 + (size_t)kt * 4096);
        r.a3 = *reinterpret_cast<const float4*>(A3 + (size_t)kt * 4096);
        r.b0 = *reinterpret_cast<const float4*>(Bg + (size_t)kt * 2048);
        if (RQ < 64) r.b1 = *reinterpret_cast<const float4*>(Bg + (size_t)kt * 2048 + RQ * 32);
    };
    auto st_off = [&](int row) { return row * 64 + ((((tid & 7) >> 1) ^ ((row >> 2) & 3)) << 4) + (tid & 1) * 8; };
    unsigned char* a_st0 = lds + st_off(srow);
    unsigned char* a_st1 = lds + st_off(srow + RQ);
    unsigned char* a_st2 = lds + st_off(srow + 2 * RQ);
    unsigned char* a_st3 = lds + st_off(srow + 3 * RQ);
    unsigned char* b_st0 = lds + PJ_B_BASE + st_off(srow);
    unsigned char* b_st1 = lds + PJ_B_BASE + st_off(srow + RQ);
    auto store = [&](const Tile& r) {
        split_store4(r.a0, a_st0, PJ_A_PLANE);
        split_store4(r.a1, a_st1, PJ_A_PLANE);
        split_store4(r.a2, a_st2, PJ_A_PLANE);
        split_store4(r.a3, a_st3, PJ_A_PLANE);
        split_store4(r.b0, b_st0, PJ_B_PLANE);
        if (RQ < 64) split_store4(r.b1, b_st1, PJ_B_PLANE);
    };
    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    const int arow = wave * 32 + l31, brow0 = l31, brow1 = 32 + l31;
    const int a_sw = (arow >> 2) & 3, b_sw0 = (brow0 >> 2) & 3, b_sw1 = (brow1 >> 2) & 3;
    const unsigned char* a_rd = lds + arow * 64;
    const unsigned char* b_rd0 = lds + PJ_B_BASE + brow0 * 64;
    const unsigned char* b_rd1 = lds + PJ_B_BASE + brow1 * 64;
#define PJ_MMA_TILE()                                                                                    \
    _Pragma("unroll") for (int st = 0; st < 2; ++st) {                                                   \
        bf16x8 a[3], b0[3], b1[3];                                                                       \
        _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                  \
            a[p] = *reinterpret_cast<const bf16x8*>(a_rd + p * PJ_A_PLANE + (((2 * st + h) ^ a_sw) << 4));   \
            b0[p] = *reinterpret_cast<const bf16x8*>(b_rd0 + p * PJ_B_PLANE + (((2 * st + h) ^ b_sw0) << 4)); \
            b1[p] = *reinterpret_cast<const bf16x8*>(b_rd1 + p * PJ_B_PLANE + (((2 * st + h) ^ b_sw1) << 4)); \
        }                                                                                                \
        MDNO_MMA6(a, b0, acc0) MDNO_MMA6(a, b1, acc1)                                                    \
    }
    const bool rows_live = wave < live;
    Tile P, Q;
    // nk is even (k % 128 == 0: k/64 k-tiles per slice, + the 2 of s0).  Every load below is issued whatever kt is
    // (past the end: the last tile again, an L2 hit that is never stored) so that the number of loads in flight at
    // each LDS store is a constant the compiler's vmcnt waits can rely on: 5 newer ones stay in flight.
    const int last = nk - 1;
    load(P, 0);
    load(Q, 1);
    store(P);
    __syncthreads();
    // LDS holds tile kt, the other set tile kt+1 (landed or landing), the set just stored is reloaded with tile kt+2
    for (int kt = 0;; kt += 2) {
        load(P, min(kt + 2, last));
        __builtin_amdgcn_sched_barrier(0);
        if (rows_live) { PJ_MMA_TILE() }
        __syncthreads();
        store(Q);
        __syncthreads();
        load(Q, min(kt + 3, last));
        __builtin_amdgcn_sched_barrier(0);
        if (rows_live) { PJ_MMA_TILE() }
        if (kt + 2 >= nk) break;
        __syncthreads();
        store(P);
        __syncthreads();
    }
#undef PJ_MMA_TILE
    float* Po = part + (size_t)slice * part_stride;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = first + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m < cnt) {
            Po[(size_t)(row0 + m) * 64 + l31] = acc0[e];
            Po[(size_t)(row0 + m) * 64 + 32 + l31] = acc1[e];
        }
    }
}

// ---------------------------------------------------------------- K2 on two fp16 planes (gemm_mode SPLIT_F16)
// project_kernel's loop with half the matrix work: S's rows times the power of two that puts the row's largest entry
// (rowmax: K1's record, one value per column block of the row) in [2^13, 2^14), split into two fp16 planes on the way
// to LDS; W3R comes pre-split (w3_planes_f16_kernel: 16 B per thread and plane tile, no vector work); three plane
// products per pair in ONE accumulator; the epilogue takes both scales out again (exact powers of two).  The same slices, the
// same partials layout, the same fixed association as project_kernel.
__global__ __launch_bounds__(512) void project_f16_kernel(const float* __restrict__ S, const unsigned short* __restrict__ w3h_bits,
                                                          float* __restrict__ part, int K, int cnt, int row0,
                                                          long long part_stride, const float* __restrict__ rowmax, int nq,
                                                          const float* __restrict__ colinv) {
    // (w3h_bits: the fp16 planes as raw 16-bit words — a _Float16 in the signature leaves the name mangled in traces)
    const _Float16* __restrict__ w3h = reinterpret_cast<const _Float16*>(w3h_bits);
    constexpr int PJ_ROWS = 256, PJ_A_PLANE = PJ_ROWS * 64, PJ_B_BASE = 2 * PJ_A_PLANE, NT = 2, RQ = 64;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * PJ_A_PLANE + 2 * PJ_B_PLANE];      // 40 KiB
    __shared__ int rowexp[PJ_ROWS];      // exponent of each row's scale (K1's and this kernel's): taken out again in the epilogue
    const int ngrp = gridDim.x / PJ_SLICES;
    const int xcd = blockIdx.x & 7, rr = blockIdx.x >> 3;
    const int rg = rr % ngrp, slice = (rr / ngrp) * 8 + xcd;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    const int nkt = (int)moment_nkt(K), per = (nkt - 2) / PJ_SLICES;
    const int kt0 = slice * per, nk = per + (slice == PJ_SLICES - 1 ? 2 : 0);
    const int first = rg * PJ_ROWS;
    const int rows_here = cnt - first < PJ_ROWS ? cnt - first : PJ_ROWS;
    const int live = (rows_here + 31) >> 5;
    auto a_row = [&](int r) { return r < rows_here ? r : rows_here - 1; };      // (as project_kernel: the last live row again)
    auto a_ptr = [&](int r) {
        return S + ((size_t)(NT * rg + (r >> 7)) * nkt + kt0) * 4096 + (r & 127) * 32 + scol;
    };
    // the scale of a row: from the maxima of its nq column blocks (rows past the last destination take the last one's)
    auto row_scale = [&](int r) {
        const float* m = rowmax + (size_t)(first + r) * (nq + 1);
        float v = 0.f;
        for (int q = 0; q < nq; ++q) v = fmaxf(v, m[q]);
        return f16_row_scale(v);
    };
    auto row_exp = [&](int r) { return (int)rowmax[(size_t)(first + r) * (nq + 1) + nq]; };      // K1's ex + 5 of the row
    const int r0_ = a_row(srow), r1_ = a_row(srow + RQ), r2_ = a_row(srow + 2 * RQ), r3_ = a_row(srow + 3 * RQ);
    const float* A0 = a_ptr(r0_);
    const float* A1 = a_ptr(r1_);
    const float* A2 = a_ptr(r2_);
    const float* A3 = a_ptr(r3_);
    const float sc0 = row_scale(r0_), sc1 = row_scale(r1_), sc2 = row_scale(r2_), sc3 = row_scale(r3_);
    if ((tid & 7) == 0) {      // (read in the epilogue, behind the loop's barriers)
        rowexp[srow] = f32_exponent(sc0) + row_exp(r0_); rowexp[srow + RQ] = f32_exponent(sc1) + row_exp(r1_);
        rowexp[srow + 2 * RQ] = f32_exponent(sc2) + row_exp(r2_); rowexp[srow + 3 * RQ] = f32_exponent(sc3) + row_exp(r3_);
    }
    // W3R's plane tiles: thread (plane bp, row bo, 16-B chunk bc)
    const int bp = tid >> 8, bo = (tid >> 2) & 63, bc = tid & 3;
    const _Float16* Bg = w3h + ((size_t)kt0 * 2 + bp) * 2048 + bo * 32 + bc * 8;
    struct Tile { float4 a0, a1, a2, a3; uint4 b; };
    auto load = [&](Tile& r, int kt) {
        r.a0 = *reinterpret_cast<const float4*>(A0 + (size_t)kt * 4096);
        r.a1 = *reinterpret_cast<const float4*>(A1 + (size_t)kt * 4096);
        r.a2 = *reinterpret_cast<const float4*>(A2 + (size_t)kt * 4096);
        r.a3 = *reinterpret_cast<const float4*>(A3 + (size_t)kt * 4096);
        r.b = *reinterpret_cast<const uint4*>(Bg + (size_t)kt * 4096);
    };
    auto st_off = [&](int row) { return row * 64 + ((((tid & 7) >> 1) ^ ((row >> 2) & 3)) << 4) + (tid & 1) * 8; };
    unsigned char* a_st0 = lds + st_off(srow);
    unsigned char* a_st1 = lds + st_off(srow + RQ);
    unsigned char* a_st2 = lds + st_off(srow + 2 * RQ);
    unsigned char* a_st3 = lds + st_off(srow + 3 * RQ);
    unsigned char* b_st = lds + PJ_B_BASE + bp * PJ_B_PLANE + bo * 64 + ((bc ^ ((bo >> 2) & 3)) << 4);
    auto scaled = [](const float4 v, float sc) { return make_float4(v.x * sc, v.y * sc, v.z * sc, v.w * sc); };
    auto store = [&](const Tile& r) {
        split2_store4(scaled(r.a0, sc0), a_st0, PJ_A_PLANE);
        split2_store4(scaled(r.a1, sc1), a_st1, PJ_A_PLANE);
        split2_store4(scaled(r.a2, sc2), a_st2, PJ_A_PLANE);
        split2_store4(scaled(r.a3, sc3), a_st3, PJ_A_PLANE);
        *reinterpret_cast<uint4*>(b_st) = r.b;
    };
    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    const int arow = wave * 32 + l31, brow0 = l31, brow1 = 32 + l31;
    const int a_sw = (arow >> 2) & 3, b_sw0 = (brow0 >> 2) & 3, b_sw1 = (brow1 >> 2) & 3;
    const unsigned char* a_rd = lds + arow * 64;
    const unsigned char* b_rd0 = lds + PJ_B_BASE + brow0 * 64;
    const unsigned char* b_rd1 = lds + PJ_B_BASE + brow1 * 64;
#define PJ_MMA_TILE()                                                                                    \
    _Pragma("unroll") for (int st = 0; st < 2; ++st) {                                                   \
        f16x8 a[2], b0[2], b1[2];                                                                        \
        _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                                  \
            a[p] = *reinterpret_cast<const f16x8*>(a_rd + p * PJ_A_PLANE + (((2 * st + h) ^ a_sw) << 4));    \
            b0[p] = *reinterpret_cast<const f16x8*>(b_rd0 + p * PJ_B_PLANE + (((2 * st + h) ^ b_sw0) << 4)); \
            b1[p] = *reinterpret_cast<const f16x8*>(b_rd1 + p * PJ_B_PLANE + (((2 * st + h) ^ b_sw1) << 4)); \
        }                                                                                                \
        MDNO_MMA3H(a, b0, acc0) MDNO_MMA3H(a, b1, acc1)                                                  \
    }
    const bool rows_live = wave < live;
    Tile P, Q;
    // (as project_kernel: nk even, every load unconditional, two K-tiles in flight per thread)
    const int last = nk - 1;
    load(P, 0);
    load(Q, 1);
    store(P);
    __syncthreads();
    for (int kt = 0;; kt += 2) {
        load(P, min(kt + 2, last));
        __builtin_amdgcn_sched_barrier(0);
        if (rows_live) { PJ_MMA_TILE() }
        __syncthreads();
        store(Q);
        __syncthreads();
        load(Q, min(kt + 3, last));
        __builtin_amdgcn_sched_barrier(0);
        if (rows_live) { PJ_MMA_TILE() }
        if (kt + 2 >= nk) break;
        __syncthreads();
        store(P);
        __syncthreads();
    }
#undef PJ_MMA_TILE
    // the row's and the column's powers of two out again — as ONE exponent per element (ldexp: exact over the whole range;
    // the two factors one after the other could overflow on the way): the partials K3 adds are in S . W3R's own units
    const int ec0 = f32_exponent(colinv[l31]), ec1 = f32_exponent(colinv[32 + l31]);
    float* Po = part + (size_t)slice * part_stride;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int ml = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, m = first + ml;
        if (m < cnt) {
            const int er = rowexp[ml];
            Po[(size_t)(row0 + m) * 64 + l31] = ldexpf(acc0[e], ec0 - er);
            Po[(size_t)(row0 + m) * 64 + 32 + l31] = ldexpf(acc1[e], ec1 - er);
        }
    }
}

// ---------------------------------------------------------------- K2, exact fp32 (gemm_mode F32)
// Workgroup (one 128-row tile of the S image, K slice): the same slices and the same partials as project_kernel, the
// products on v_mfma_f32_32x32x2_f32 (mfma_f32.h: fp32 K-tiles as LDS rows of 36 floats).
__global__ __launch_bounds__(256) void project_f32_kernel(const float* __restrict__ S, const float* __restrict__ w3r,
                                                          float* __restrict__ part, int K, int cnt, int row0,
                                                          long long part_stride) {
    using namespace f32mma;
    __shared__ __attribute__((aligned(16))) float As[128 * LD];
    __shared__ __attribute__((aligned(16))) float Bs[64 * LD];
    const int ngrp = gridDim.x / PJ_SLICES;
    const int xcd = blockIdx.x & 7, rr = blockIdx.x >> 3;
    const int rg = rr % ngrp, slice = (rr / ngrp) * 8 + xcd;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    const int nkt = (int)moment_nkt(K), per = (nkt - 2) / PJ_SLICES;
    const int kt0 = slice * per, nk = per + (slice == PJ_SLICES - 1 ? 2 : 0);
    const int first = rg * 128;
    const float* Ag = S + ((size_t)rg * nkt + kt0) * 4096 + srow * 32 + scol;      // rows srow + 32 j
    const float* Bg = w3r + (size_t)kt0 * 2048 + srow * 32 + scol;                 // rows srow, srow + 32
    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    for (int kt = 0; kt < nk; ++kt) {
        float4 a[4], b[2];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const float4*>(Ag + (size_t)kt * 4096 + j * 32 * 32);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const float4*>(Bg + (size_t)kt * 2048 + j * 32 * 32);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(&As[(srow + 32 * j) * LD + scol]) = a[j];
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<float4*>(&Bs[(srow + 32 * j) * LD + scol]) = b[j];
        __syncthreads();
        mma_32x64(acc0, acc1, &As[(wave * 32 + l31) * LD + 4 * h], &Bs[l31 * LD + 4 * h]);
    }
    float* Po = part + (size_t)slice * part_stride;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = first + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m < cnt) {
            Po[(size_t)(row0 + m) * 64 + l31] = acc0[e];
            Po[(size_t)(row0 + m) * 64 + 32 + l31] = acc1[e];
        }
    }
}

// ---------------------------------------------------------------- K3: slices + root + bias + mean + act
// One workgroup (32 chains x 16 lanes) per destination: chain es adds K slices es, es+32, .. in that order, the 32
// chains are added in chain order through LDS; the root product is split over the chains the same way.  Every load
// depends on the destination's index only (one round trip).
constexpr int FN_CHAINS = 32;

__global__ __launch_bounds__(FN_CHAINS * 16) void finish_kernel(const float* __restrict__ part, long long part_stride,
                                                                const int* __restrict__ row_ptr, const float* __restrict__ x,
                                                                const float* __restrict__ root, const float* __restrict__ bias,
                                                                float* __restrict__ y, int row0, int aggr, int relu,
                                                                float* __restrict__ xm_out) {
    constexpr int CPT = 64 / FN_CHAINS;
    __shared__ float4 red[FN_CHAINS][16];
    __shared__ float4 red2[FN_CHAINS][16];
    const int tid = threadIdx.x, es = tid >> 4, q = tid & 15;
    const int t = row0 + blockIdx.x;
    const int deg = row_ptr[t + 1] - row_ptr[t];
    float4 rootv[CPT], biasv = make_float4(0.f, 0.f, 0.f, 0.f);
    float xin[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        rootv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        xin[i] = 0.f;
        if (root != nullptr) {
            xin[i] = x[(size_t)t * 64 + CPT * es + i];
            rootv[i] = *reinterpret_cast<const float4*>(root + (CPT * es + i) * 64 + 4 * q);
        }
    }
    if (bias != nullptr && es == 0) biasv = *reinterpret_cast<const float4*>(bias + 4 * q);
    float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        float4 v[PJ_SLICES / FN_CHAINS];
#pragma unroll
        for (int u = 0; u < PJ_SLICES / FN_CHAINS; ++u)
            v[u] = *reinterpret_cast<const float4*>(part + (size_t)(es + u * FN_CHAINS) * part_stride + (size_t)t * 64 + 4 * q);
#pragma unroll
        for (int u = 0; u < PJ_SLICES / FN_CHAINS; ++u) { z.x += v[u].x; z.y += v[u].y; z.z += v[u].z; z.w += v[u].w; }
    }
    float4 racc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        racc.x = fmaf(xin[i], rootv[i].x, racc.x); racc.y = fmaf(xin[i], rootv[i].y, racc.y);
        racc.z = fmaf(xin[i], rootv[i].z, racc.z); racc.w = fmaf(xin[i], rootv[i].w, racc.w);
    }
    red[es][q] = z;
    red2[es][q] = racc;
    __syncthreads();
    if (es == 0) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f), rs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < FN_CHAINS; ++c) {
            const float4 a = red[c][q], b = red2[c][q];
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            rs.x += b.x; rs.y += b.y; rs.z += b.z; rs.w += b.w;
        }
        if (aggr == MDNO_AGGR_MEAN) {
            const float inv = (float)(deg > 1 ? deg : 1);
            s.x /= inv; s.y /= inv; s.z /= inv; s.w /= inv;
        }
        if (root != nullptr) { s.x += rs.x; s.y += rs.y; s.z += rs.z; s.w += rs.w; }
        if (bias != nullptr) { s.x += biasv.x; s.y += biasv.y; s.z += biasv.z; s.w += biasv.w; }
        if (relu) { s.x = relu_f(s.x); s.y = relu_f(s.y); s.z = relu_f(s.z); s.w = relu_f(s.w); }
        *reinterpret_cast<float4*>(y + (size_t)t * 64 + 4 * q) = s;
        if (xm_out != nullptr) {      // the row's largest |feature|: K1 on fp16 planes scales a destination's neighbours by it
            float m = fmaxf(fmaxf(fabsf(s.x), fabsf(s.y)), fmaxf(fabsf(s.z), fabsf(s.w)));
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));      // (es == 0: lanes 0..15 of wave 0)
            if (q == 0) xm_out[t] = m;
        }
    }
}

// every node's largest |feature| for a forward's FIRST conv application (the later ones get it from K3)
__global__ __launch_bounds__(256) void row_absmax_kernel(const float* __restrict__ x, int num_rows, float* __restrict__ xm) {
    const int r = blockIdx.x * 16 + (threadIdx.x >> 4), q = threadIdx.x & 15;
    float m = 0.f;
    if (r < num_rows) {
        const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * 64 + 4 * q);
        m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (q == 0 && r < num_rows) xm[r] = m;
}


}  // namespace

// ---------------------------------------------------------------- host side
constexpr int kMomentChunkRows = 512;      // destinations per S chunk: 512 x 64 k x 4 B = 128 MiB at k = 1024, written by K1
                                           // and read back by K2 while still in the 256 MiB Infinity Cache
static int moment_chunk_rows(int num_rows) {
    const int padded = (num_rows + 127) / 128 * 128;
    return padded < kMomentChunkRows ? padded : kMomentChunkRows;
}

// k % 128: what the hidden GEMM that writes H tiles by, and what makes a K2 slice an even number of k-tiles
// (64 k / 32 = 2 k k-tiles over 128 slices = k / 64 each; K2's loop takes them two at a time)
bool moment_supported(int width, int ker_width) { return width == 64 && ker_width >= 128 && ker_width % 128 == 0; }
static int moment_nq(int ker_width) { return (ker_width + MO_CQ - 1) / MO_CQ + 1; }      // K1's column blocks + the s0 block
static size_t s_chunk_floats(int num_rows, int ker_width) { return (size_t)(moment_chunk_rows(num_rows) / 128) * moment_nkt(ker_width) * 4096; }

// ONE carve for the size and for the pointers (Carver(nullptr) only counts): the two cannot drift apart.  (Up to round 5
// the size took xm as one run of 2 R floats while the carve took two 256-B-aligned runs of R: for R floats not a multiple
// of 64 the second run's tail — and whatever was carved behind it — lay up to 256 B past the region.)
static MomentWs moment_carve_impl(Carver& cv, int num_rows, int ker_width) {
    MomentWs f{};
    f.w3r = cv.take<float>((size_t)(64 * ker_width + 64) * 64);                       // W3R (+ the B3 rows)
    f.s = cv.take<float>(s_chunk_floats(num_rows, ker_width));                         // S (+ s0), one chunk of destinations
    f.part = cv.take<float>((size_t)PJ_SLICES * num_rows * 64);                        // K-slice partials of z
    f.part_stride = (long long)num_rows * 64;
    f.order = cv.take<int>((size_t)num_rows);                                          // destinations of each chunk by decreasing degree
    f.w3h = cv.take<_Float16>((size_t)(64 * ker_width + 64) * 64 * 2);                 // W3R on two fp16 planes (SPLIT_F16)
    f.colinv = cv.take<float>(64);                                                     // its columns' inverse scales
    f.colmax_bits = cv.take<int>(64);                                                  // (their maxima, as bits)
    f.rowmax = cv.take<float>((size_t)(moment_chunk_rows(num_rows) + 256) * (moment_nq(ker_width) + 1));   // row maxima of the S chunk (+ the rows' scale exponents)
    f.xm[0] = cv.take<float>((size_t)num_rows);                                        // every node's largest |feature|: in,
    f.xm[1] = cv.take<float>((size_t)num_rows);                                        // out
    f.counters = cv.take<int>(64);                                                     // MOMENT_CNT_*
    return f;
}

size_t moment_workspace_bytes(int num_rows, int ker_width) {
    Carver cv(nullptr);
    (void)moment_carve_impl(cv, num_rows, ker_width);
    return cv.used();
}

MomentWs moment_carve(void* ws, int num_rows, int ker_width) {
    Carver cv(ws);
    return moment_carve_impl(cv, num_rows, ker_width);
}

int moment_prepare_weights(const float* w3, const float* b3, int ker_width, const MomentWs& f, hipStream_t s, int gemm_mode) {
    const long long total = (long long)(64 * ker_width + 64) * 64;
    hipLaunchKernelGGL(w3_moment_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w3, b3, ker_width, f.w3r);
    if (gemm_mode == MDNO_GEMM_SPLIT_F16) {
        MDNO_HIP(hipMemsetAsync(f.colmax_bits, 0, 64 * sizeof(int), s));
        hipLaunchKernelGGL(w3_colmax_kernel, dim3(256), dim3(256), 0, s, (const float*)f.w3r, total / 2048, f.colmax_bits);
        hipLaunchKernelGGL(w3_planes_f16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)f.w3r, total,
                           (const int*)f.colmax_bits, f.w3h, f.colinv);
    }
    return check_launch("w3_moment_kernel");
}

int moment_prepare_graph(const int* row_ptr, int num_rows, const MomentWs& f, hipStream_t s) {
    TimedSection ts(KID_GRAPH, s);
    hipLaunchKernelGGL(degree_order_kernel<kMomentChunkRows>, dim3((num_rows + kMomentChunkRows - 1) / kMomentChunkRows),
                       dim3(kMomentChunkRows), 0, s, row_ptr, num_rows, f.order);
    return check_launch("degree_order_kernel");
}

int moment_row_absmax(const float* x, int num_rows, const MomentWs& f, hipStream_t s) {
    hipLaunchKernelGGL(row_absmax_kernel, dim3((num_rows + 15) / 16), dim3(256), 0, s, x, num_rows, f.xm[0]);
    return check_launch("row_absmax_kernel");
}

int moment_conv(const float* x, const float* h2, const int* row_ptr, const int* src, int num_rows, int ker_width,
                const float* root, const float* bias, int aggr, int relu, float* y, const MomentWs& f, hipStream_t s,
                int gemm_mode, int application) {
    MDNO_REQUIRE(moment_supported(64, ker_width), MDNO_EUNSUPPORTED, "moment conv: ker_width=%d (x128)", ker_width);
    const bool exact_f32 = gemm_mode == MDNO_GEMM_F32, f16 = gemm_mode == MDNO_GEMM_SPLIT_F16;
    const int nq = moment_nq(ker_width);
    // (fp16 planes) the nodes' largest |feature|: application a reads xm[a & 1] (x's) and writes xm[(a + 1) & 1] (y's)
    const float* xm_in = f.xm[application & 1];
    float* xm_out = f16 ? f.xm[(application + 1) & 1] : nullptr;
    size_t cached_bytes = kMomentCachedBytes;
#ifdef MDNO_EXP_CACHE_ENV
    if (const char* v = getenv("MDNO_EXP_CACHE_MIB")) cached_bytes = (size_t)atoi(v) << 20;
#endif
    const int cache_e = (int)(cached_bytes / ((size_t)ker_width * sizeof(float))) & ~127;      // whole 128-edge tiles of the image
    for (int r0 = 0; r0 < num_rows; r0 += kMomentChunkRows) {
        const int cnt = num_rows - r0 < kMomentChunkRows ? num_rows - r0 : kMomentChunkRows;
        {   // K1: the chunk's destinations x the k/256 column blocks
            TimedSection ts(KID_NNCONV, s);
            const dim3 grid((unsigned)(((cnt + 7) / 8) * 8 * nq));
            if (exact_f32)      // (h2: the k-tiled image in every GEMM mode)
                hipLaunchKernelGGL(moment_f32_kernel, grid, dim3(256), 0, s, h2, row_ptr, src, (const int*)f.order, f.s,
                                   ker_width, r0, cnt, x);
            else if (f16)
                hipLaunchKernelGGL(moment_kernel<true>, grid, dim3(256), 0, s, h2, row_ptr, src, (const int*)f.order, f.s, ker_width,
                                   r0, cnt, x, cache_e, f.rowmax, xm_in, f.counters);
            else
                hipLaunchKernelGGL(moment_kernel<false>, grid, dim3(256), 0, s, h2, row_ptr, src, (const int*)f.order, f.s, ker_width,
                                   r0, cnt, x, cache_e, (float*)nullptr, (const float*)nullptr, (int*)nullptr);
        }
        {   // K2: groups of row tiles x K slices
            TimedSection ts(KID_FACT_Y, s);
            if (exact_f32)
                hipLaunchKernelGGL(project_f32_kernel, dim3(((cnt + 127) / 128) * PJ_SLICES), dim3(256), 0, s, (const float*)f.s,
                                   (const float*)f.w3r, f.part, ker_width, cnt, r0, f.part_stride);
            else if (f16)
                hipLaunchKernelGGL(project_f16_kernel, dim3(((cnt + 255) / 256) * PJ_SLICES), dim3(512), 0, s, (const float*)f.s,
                                   reinterpret_cast<const unsigned short*>(f.w3h), f.part, ker_width, cnt, r0, f.part_stride,
                                   (const float*)f.rowmax, nq, (const float*)f.colinv);
            else
                hipLaunchKernelGGL(project_kernel<256>, dim3(((cnt + 255) / 256) * PJ_SLICES), dim3(512), 0, s, (const float*)f.s,
                                   (const float*)f.w3r, f.part, ker_width, cnt, r0, f.part_stride);
        }
        {   // K3
            TimedSection ts(KID_NNCONV_COMBINE, s);
            hipLaunchKernelGGL(finish_kernel, dim3(cnt), dim3(FN_CHAINS * 16), 0, s, (const float*)f.part, f.part_stride,
                               row_ptr, x, root, bias, y, r0, aggr, relu, xm_out);
        }
    }
    return check_launch("moment_conv");
}

}  // namespace mdno
