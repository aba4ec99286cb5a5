// Training (BASELINE.json configs[3]): backward of the kernel-integral block — 2*depth conv
// applications sharing one edge-MLP — as individually callable ops.  Replaces what autograd +
// torch_geometric do for graph_kernel.py:445-474 (train) on the path :299-302 / :194-209 / :239-242.
//
// Forward (materialised) per application a = 1..L:  z_a = mean_{e->t} x_{a-1}[src e] . W_e + x_{a-1}.root + bias,
// x_a = relu(z_a), W_e = reshape(L2(relu(L1(relu(L0 attr_e))))).  Given g_a = dLoss/dx_a:
//     gz_a      = g_a * (x_a > 0)                          gs_a[t] = gz_a[t] / max(deg_t, 1)
//     g_{a-1}   = gz_a . root^T + sum_{e: src e = r} W_e . gs_a[dst e]                 (nnconv_bwd_x)
//     d root    = sum_a x_{a-1}^T . gz_a,   d bias = sum_a colsum(gz_a)               (nnconv_bwd_root)
//     d W_e     = sum_a x_{a-1}[src e] (x) gs_a[dst e]                                (nnconv_bwd_we)
// and through the edge-MLP with  C = A . Bt^T  (linear_fwd),  C = A^T . B over rows (gemm_atb), column sums
// and ReLU masks.  Everything is fp32; reductions over rows/edges use fixed-order partial sums
// (no float atomics), so gradients are bitwise reproducible.
#include "kernels.h"
#include "mfma_f32.h"
#include "reduce.h"

namespace mdno {
namespace {

using f32mma::f32x16;
using f32mma::mma_64x64;
constexpr int BK = f32mma::BK, LD = f32mma::LD;

// ---------------------------------------------------------------- C = act(A . Bt^T + bias), any M
// A [M,K], Bt [N,K], C [M,N]; N % 128 == 0, K % 32 == 0 (MFMA path).  128x128x32 tile, 4 waves.
template <bool RELU>
__global__ __launch_bounds__(256, 2) void linear_mfma_kernel(const float* __restrict__ A, const float* __restrict__ Bt,
                                                             const float* __restrict__ bias, float* __restrict__ Cm,
                                                             int rows, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * 128 * LD;
    const int bm = blockIdx.y * 128, bn = blockIdx.x * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    const size_t ldk = (size_t)K;
    auto arow = [&](int r) { const int rr = bm + r; return (size_t)(rr < rows ? rr : rows - 1); };
    const float* A0 = A + arow(srow) * ldk + scol;
    const float* A1 = A + arow(srow + 32) * ldk + scol;
    const float* A2 = A + arow(srow + 64) * ldk + scol;
    const float* A3 = A + arow(srow + 96) * ldk + scol;
    const float* Bg = Bt + (size_t)(bn + srow) * ldk + scol;
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define MDNO_LOAD(KOFF)                                                  \
    ra0 = *reinterpret_cast<const float4*>(A0 + (KOFF));                 \
    ra1 = *reinterpret_cast<const float4*>(A1 + (KOFF));                 \
    ra2 = *reinterpret_cast<const float4*>(A2 + (KOFF));                 \
    ra3 = *reinterpret_cast<const float4*>(A3 + (KOFF));                 \
    rb0 = *reinterpret_cast<const float4*>(Bg + (KOFF));                 \
    rb1 = *reinterpret_cast<const float4*>(Bg + 32 * ldk + (KOFF));      \
    rb2 = *reinterpret_cast<const float4*>(Bg + 64 * ldk + (KOFF));      \
    rb3 = *reinterpret_cast<const float4*>(Bg + 96 * ldk + (KOFF));
    float* a_st = As + srow * LD + scol;
    float* b_st = Bs + srow * LD + scol;
#define MDNO_STORE(BUF)                                                            \
    *reinterpret_cast<float4*>(a_st + (BUF) * 128 * LD) = ra0;                     \
    *reinterpret_cast<float4*>(a_st + (BUF) * 128 * LD + 32 * LD) = ra1;           \
    *reinterpret_cast<float4*>(a_st + (BUF) * 128 * LD + 64 * LD) = ra2;           \
    *reinterpret_cast<float4*>(a_st + (BUF) * 128 * LD + 96 * LD) = ra3;           \
    *reinterpret_cast<float4*>(b_st + (BUF) * 128 * LD) = rb0;                     \
    *reinterpret_cast<float4*>(b_st + (BUF) * 128 * LD + 32 * LD) = rb1;           \
    *reinterpret_cast<float4*>(b_st + (BUF) * 128 * LD + 64 * LD) = rb2;           \
    *reinterpret_cast<float4*>(b_st + (BUF) * 128 * LD + 96 * LD) = rb3;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const float* a_rd = As + (wm * 64 + l31) * LD + 4 * h;
    const float* b_rd = Bs + (wn * 64 + l31) * LD + 4 * h;
    const int nk = K / BK;
    MDNO_LOAD(0)
    MDNO_STORE(0)
    __syncthreads();
    for (int kt = 0; kt < nk - 1; ++kt) {
        MDNO_LOAD((size_t)(kt + 1) * BK)
        mma_64x64(acc, a_rd + (kt & 1) * 128 * LD, b_rd + (kt & 1) * 128 * LD);
        MDNO_STORE((kt & 1) ^ 1)
        __syncthreads();
    }
    mma_64x64(acc, a_rd + ((nk - 1) & 1) * 128 * LD, b_rd + ((nk - 1) & 1) * 128 * LD);
#undef MDNO_LOAD
#undef MDNO_STORE
    // bias loaded once and pinned (sunk into the predicated store blocks it serialises the stores)
    float bv0 = 0.f, bv1 = 0.f;
    if (bias) {
        bv0 = bias[bn + wn * 64 + l31];
        bv1 = bias[bn + wn * 64 + 32 + l31];
    }
    asm volatile("" : "+v"(bv0), "+v"(bv1));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = bn + wn * 64 + j * 32 + l31;
        const float bv = j ? bv1 : bv0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = bm + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < rows) {
                    float v = acc[i][j][e] + bv;
                    if (RELU) v = relu_f(v);
                    Cm[(size_t)m * N + n] = v;
                }
            }
    }
}

// any shape (small fixtures, K = 6 first layer): one thread per output element
template <bool RELU>
__global__ __launch_bounds__(256) void linear_generic_kernel(const float* __restrict__ A, const float* __restrict__ Bt,
                                                             const float* __restrict__ bias, float* __restrict__ Cm,
                                                             long long rows, int N, int K) {
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= rows * N) return;
    const long long m = id / N;
    const int n = (int)(id % N);
    const float* a = A + m * K;
    const float* b = Bt + (size_t)n * K;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(a[k], b[k], s);
    s += bias ? bias[n] : 0.f;
    if (RELU) s = relu_f(s);
    Cm[id] = s;
}

// ---------------------------------------------------------------- C[N1,N2] = A^T . B over rows
// A [K,N1], B [K,N2] row-major (K = number of rows, e.g. edges), partial sums per K-slice:
// part[slice][N1][N2]; a second kernel adds the slices in order.  Tile 128x128, k-tile 32 rows.
// LDS holds the tiles k-major ([k][n]); MFMA fragments are 4-byte LDS reads (lanes along n).
constexpr int LDT = 128 + 4;
__global__ __launch_bounds__(256, 2) void gemm_atb_mfma_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                               float* __restrict__ part, long long K, int N1, int N2,
                                                               long long kslice) {
    __shared__ __attribute__((aligned(16))) float As[BK * LDT];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDT];
    const int bm = blockIdx.y * 128, bn = blockIdx.x * 128, slice = blockIdx.z;
    const long long k0 = (long long)slice * kslice;
    long long k1 = k0 + kslice;
    if (k1 > K) k1 = K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    // staging: 32 rows x 128 floats per operand = 1024 float4 -> 4 per thread; thread covers row tid>>5 + 8*i
    const int srow = tid >> 5, scol = (tid & 31) * 4;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (long long kt = k0; kt < k1; kt += BK) {
        float4 ra[4], rb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long r = kt + srow + 8 * i;
            if (r < k1) {
                ra[i] = *reinterpret_cast<const float4*>(A + r * N1 + bm + scol);
                rb[i] = *reinterpret_cast<const float4*>(B + r * N2 + bn + scol);
            } else {
                ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        __syncthreads();   // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4*>(As + (srow + 8 * i) * LDT + scol) = ra[i];
            *reinterpret_cast<float4*>(Bs + (srow + 8 * i) * LDT + scol) = rb[i];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            const float* ar = As + (2 * s + h) * LDT + wm * 64 + l31;
            const float* br = Bs + (2 * s + h) * LDT + wn * 64 + l31;
            const float a0 = ar[0], a1 = ar[32], b0 = br[0], b1 = br[32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    float* P = part + (size_t)slice * N1 * N2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = bn + wn * 64 + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = bm + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                P[(size_t)m * N2 + n] = acc[i][j][e];
            }
    }
}

// any N1, N2 (e.g. N2 = 6): thread per output element, loops over its K-slice
__global__ __launch_bounds__(256) void gemm_atb_generic_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                               float* __restrict__ part, long long K, int N1, int N2,
                                                               long long kslice) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= N1 * N2) return;
    const int m = id / N2, n = id % N2, slice = blockIdx.y;
    const long long k0 = (long long)slice * kslice;
    long long k1 = k0 + kslice;
    if (k1 > K) k1 = K;
    float s = 0.f;
    for (long long r = k0; r < k1; ++r) s = fmaf(A[r * N1 + m], B[r * N2 + n], s);
    part[(size_t)slice * N1 * N2 + id] = s;
}

// N2 <= 8 (the first edge-MLP layer's weight gradient: N2 = 6 edge attributes): thread per column of A, so that a
// row of A is read as whole lines (the kernel above reads each element of A from N2 threads and 172 B of every
// KiB of a row per wave); four rows in flight per thread.  Each (m, n) sum runs over its slice's rows in ascending
// order, as above.
__global__ __launch_bounds__(256) void gemm_atb_smalln_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                              float* __restrict__ part, long long K, int N1, int N2,
                                                              long long kslice) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= N1) return;
    const int slice = blockIdx.y;
    const long long k0 = (long long)slice * kslice;
    long long k1 = k0 + kslice;
    if (k1 > K) k1 = K;
    float s[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) s[n] = 0.f;
    long long r = k0;
    for (; r + 4 <= k1; r += 4) {
        float a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = A[(r + u) * N1 + m];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int n = 0; n < 8; ++n)
                if (n < N2) s[n] = fmaf(a[u], B[(r + u) * N2 + n], s[n]);
    }
    for (; r < k1; ++r) {
        const float a = A[r * N1 + m];
#pragma unroll
        for (int n = 0; n < 8; ++n)
            if (n < N2) s[n] = fmaf(a, B[r * N2 + n], s[n]);
    }
#pragma unroll
    for (int n = 0; n < 8; ++n)
        if (n < N2) part[((size_t)slice * N1 + m) * N2 + n] = s[n];
}

// column sums of A [K,N] in fixed order: part[slice][N] then reduce_slices
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ A, float* __restrict__ part, long long K,
                                                     int N, long long kslice) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int slice = blockIdx.y;
    const long long k0 = (long long)slice * kslice;
    long long k1 = k0 + kslice;
    if (k1 > K) k1 = K;
    float s = 0.f;
    for (long long r = k0; r < k1; ++r) s += A[r * N + n];
    part[(size_t)slice * N + n] = s;
}

// the same sums (each column over its slice's rows, ascending) with four adjacent columns per thread — 16-B loads —
// and four rows in flight: N % 4 == 0, 16-byte aligned A
__global__ __launch_bounds__(256) void colsum4_kernel(const float* __restrict__ A, float* __restrict__ part, long long K,
                                                      int N, long long kslice) {
    const int n = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (n >= N) return;
    const int slice = blockIdx.y;
    const long long k0 = (long long)slice * kslice;
    long long k1 = k0 + kslice;
    if (k1 > K) k1 = K;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    long long r = k0;
    for (; r + 4 <= k1; r += 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(A + (size_t)(r + u) * N + n);
#pragma unroll
        for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; r < k1; ++r) {
        const float4 v = *reinterpret_cast<const float4*>(A + (size_t)r * N + n);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(part + (size_t)slice * N + n) = s;
}

// out = g * (y > 0) [* scale[row]]
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                       const float* __restrict__ row_scale, float* __restrict__ out,
                                                       long long rows, int N) {
    const long long id = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (id >= rows * N) return;
    const float4 gv = *reinterpret_cast<const float4*>(g + id);
    const float4 yv = *reinterpret_cast<const float4*>(y + id);
    const float sc = row_scale ? row_scale[id / N] : 1.f;
    float4 o;
    o.x = yv.x > 0.f ? gv.x * sc : 0.f;
    o.y = yv.y > 0.f ? gv.y * sc : 0.f;
    o.z = yv.z > 0.f ? gv.z * sc : 0.f;
    o.w = yv.w > 0.f ? gv.w * sc : 0.f;
    *reinterpret_cast<float4*>(out + id) = o;
}

// both at once: gz = g * (y > 0) and gs = gz * scale[row] — what the conv backward needs of one application
// (d root, d bias read gz; the input and edge-weight gradients read gs = gz / max(deg,1)); same arithmetic as two calls
__global__ __launch_bounds__(256) void relu_bwd2_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                        const float* __restrict__ row_scale, float* __restrict__ gz,
                                                        float* __restrict__ gs, long long rows, int N) {
    const long long id = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (id >= rows * N) return;
    const float4 gv = *reinterpret_cast<const float4*>(g + id);
    const float4 yv = *reinterpret_cast<const float4*>(y + id);
    const float sc = row_scale[id / N];
    float4 a, b;
    a.x = yv.x > 0.f ? gv.x : 0.f; b.x = yv.x > 0.f ? gv.x * sc : 0.f;
    a.y = yv.y > 0.f ? gv.y : 0.f; b.y = yv.y > 0.f ? gv.y * sc : 0.f;
    a.z = yv.z > 0.f ? gv.z : 0.f; b.z = yv.z > 0.f ? gv.z * sc : 0.f;
    a.w = yv.w > 0.f ? gv.w : 0.f; b.w = yv.w > 0.f ? gv.w * sc : 0.f;
    *reinterpret_cast<float4*>(gz + id) = a;
    *reinterpret_cast<float4*>(gs + id) = b;
}

__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ A, float* __restrict__ At, int R, int Cc) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int i = ty; i < 32; i += 8)
        if (by + i < R && bx + tx < Cc) tile[i][tx] = A[(size_t)(by + i) * Cc + bx + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (bx + i < Cc && by + tx < R) At[(size_t)(bx + i) * R + by + tx] = tile[tx][i];
}

__global__ __launch_bounds__(256) void inv_degree_kernel(const int* __restrict__ row_ptr, int rows, int mean,
                                                         float* __restrict__ inv) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const int d = row_ptr[r + 1] - row_ptr[r];
    inv[r] = mean ? 1.0f / (float)(d > 1 ? d : 1) : 1.0f;
}

// ---------------------------------------------------------------- conv backward: input gradient
// g_prev[r] = gz[r] . root^T + sum_{p in out-edges of r} W_e[eid[p]] . gs[dst[p]]   (64x64 only)
// One workgroup (4 waves) per source row r; its out-edges (positions in the dst-sorted edge array)
// come from the src-sorted CSR (row_ptr_s, eid_s, dst_s).  Lane (g, q) owns rows 16g..16g+15 x
// columns 4q..4q+3 of W_e as in the forward kernel; per-lane partial dot products are summed over
// ALL the row's edges first and reduced across the 16 q-lanes once per row.
// STREAM: the matrix is an edge's W_e — read once per application, far larger than the caches: non-temporal loads
template <bool STREAM>
__device__ __forceinline__ void wg_accumulate(float (&acc)[16], const float* __restrict__ wmat,
                                              const float* __restrict__ gvec, int g, int q) {
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    const float4 gq = *reinterpret_cast<const float4*>(gvec + 4 * q);
    const float* wp = wmat + (16 * g) * 64 + 4 * q;
    float4 w[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (STREAM) {
            const f32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(wp + r * 64));
            w[r] = make_float4(t.x, t.y, t.z, t.w);
        } else {
            w[r] = *reinterpret_cast<const float4*>(wp + r * 64);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
        acc[r] = fmaf(w[r].x, gq.x, fmaf(w[r].y, gq.y, fmaf(w[r].z, gq.z, fmaf(w[r].w, gq.w, acc[r]))));
}

// y_below != NULL: gz_below / gs_below of the application below are written instead of g_prev (mdno_relu_bwd2 fused)
__global__ __launch_bounds__(256) void nnconv_bwd_x_kernel(const float* __restrict__ gz, const float* __restrict__ gs,
                                                           const int* __restrict__ row_ptr_s,
                                                           const int* __restrict__ eid_s, const int* __restrict__ dst_s,
                                                           const float* __restrict__ w_e, const float* __restrict__ root,
                                                           float* __restrict__ g_prev, int num_rows,
                                                           const float* __restrict__ y_below = nullptr,
                                                           const float* __restrict__ inv_deg = nullptr,
                                                           float* __restrict__ gz_below = nullptr,
                                                           float* __restrict__ gs_below = nullptr) {
    __shared__ float red[4][64];
    const int row = blockIdx.x;
    if (row >= num_rows) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, q = lane & 15;
    const int beg = row_ptr_s[row], end = row_ptr_s[row + 1];
    float acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int p = beg + wave; p < end; p += 4)
        wg_accumulate<true>(acc, w_e + (size_t)eid_s[p] * 4096, gs + (size_t)dst_s[p] * 64, g, q);
    if (root != nullptr && wave == ((end - beg) & 3)) wg_accumulate<false>(acc, root, gz + (size_t)row * 64, g, q);
    // reduce over the 16 q-lanes of each group: xor 1,2,4,8
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = acc[r];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8);
        acc[r] = v;
    }
    if (q == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][16 * g + r] = acc[r];
    }
    __syncthreads();
    if (tid < 64) {
        const float v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        const size_t at = (size_t)row * 64 + tid;
        if (y_below != nullptr) {
            const bool on = y_below[at] > 0.f;
            gz_below[at] = on ? v : 0.f;
            gs_below[at] = on ? v * inv_deg[row] : 0.f;
        } else {
            g_prev[at] = v;
        }
    }
}

// ---------------------------------------------------------------- conv backward: d root, d bias
// d root[i][o] (+)= sum_{l, r} x_l[r][i] * gz_l[r][o];  d bias[o] (+)= sum_{l, r} gz_l[r][o]
// x, gz: [L, R, 64] stacked layers.  Block b takes a slice of the L*R rows -> partials, then reduce.
__device__ __forceinline__ void bwd_root_slice(const float* __restrict__ x, const float* __restrict__ gz, long long r0, long long r1,
                                               int slot, float* __restrict__ part_root, float* __restrict__ part_bias) {
    __shared__ float xs[64][65], gsx[64][65];
    const int tid = threadIdx.x;
    const int i0 = (tid >> 4) * 4, o0 = (tid & 15) * 4;    // 4x4 outputs per thread
    float acc[4][4] = {};
    float bsum = 0.f;
    for (long long rb = r0; rb < r1; rb += 64) {
        __syncthreads();
        for (int t = tid; t < 64 * 64; t += 256) {
            const int rr = t >> 6, c = t & 63;
            const bool ok = rb + rr < r1;
            xs[rr][c] = ok ? x[(rb + rr) * 64 + c] : 0.f;
            gsx[rr][c] = ok ? gz[(rb + rr) * 64 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int rr = 0; rr < 64; ++rr) {
            float xv[4], gv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { xv[a] = xs[rr][i0 + a]; gv[a] = gsx[rr][o0 + a]; }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = fmaf(xv[a], gv[b], acc[a][b]);
        }
        if (tid < 64)
            for (int rr = 0; rr < 64; ++rr) bsum += gsx[rr][tid];
    }
    float* pr = part_root + (size_t)slot * 4096;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) pr[(i0 + a) * 64 + o0 + b] = acc[a][b];
    if (tid < 64) part_bias[(size_t)slot * 64 + tid] = bsum;
}

__global__ __launch_bounds__(256) void nnconv_bwd_root_kernel(const float* __restrict__ x, const float* __restrict__ gz,
                                                              long long rows, long long slice_rows,
                                                              float* __restrict__ part_root,
                                                              float* __restrict__ part_bias) {
    const long long r0 = (long long)blockIdx.x * slice_rows;
    long long r1 = r0 + slice_rows;
    if (r1 > rows) r1 = rows;
    bwd_root_slice(x, gz, r0, r1, blockIdx.x, part_root, part_bias);
}

__global__ __launch_bounds__(256) void nnconv_bwd_root_pair_kernel(const float* __restrict__ x, const float* __restrict__ gz,
                                                                   long long rows_each, long long slice_rows, int per_half,
                                                                   float* __restrict__ part_root, float* __restrict__ part_bias) {
    const int half = (int)blockIdx.x / per_half, b = (int)blockIdx.x - half * per_half;
    const long long base = (long long)half * rows_each;
    const long long r0 = base + (long long)b * slice_rows;
    long long r1 = r0 + slice_rows;
    if (r1 > base + rows_each) r1 = base + rows_each;
    bwd_root_slice(x, gz, r0, r1, blockIdx.x, part_root, part_bias);
}

// ---------------------------------------------------------------- conv backward: d W_e
// dW_e[p][i][o] (+)= sum_l x_l[src[p]][i] * gs_l[dst[p]][o];  x, gs: [L, R, 64].  One wave per edge,
// lane (g, q) owns rows 16g..16g+15 x columns 4q..4q+3 and writes them as 16 coalesced 16-B stores.
__global__ __launch_bounds__(256) void nnconv_bwd_we_kernel(const float* __restrict__ x, const float* __restrict__ gs,
                                                            const int* __restrict__ src, const int* __restrict__ dst,
                                                            long long E, int L, long long layer_stride,
                                                            float* __restrict__ dwe, int accumulate) {
    const int lane = threadIdx.x & 63, g = lane >> 4, q = lane & 15;
    const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= E) return;
    const float* xs = x + (size_t)src[p] * 64 + 16 * g;
    const float* gq = gs + (size_t)dst[p] * 64 + 4 * q;
    float4 acc[16];
    float* out = dwe + (size_t)p * 4096 + (16 * g) * 64 + 4 * q;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        acc[r] = accumulate ? *reinterpret_cast<const float4*>(out + r * 64) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = 0; l < L; ++l) {
        const float4 gv = *reinterpret_cast<const float4*>(gq + (size_t)l * layer_stride);
        const float* xl = xs + (size_t)l * layer_stride;
        const float4 x0 = *reinterpret_cast<const float4*>(xl), x1 = *reinterpret_cast<const float4*>(xl + 4);
        const float4 x2 = *reinterpret_cast<const float4*>(xl + 8), x3 = *reinterpret_cast<const float4*>(xl + 12);
        const float xv[16] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w,
                              x2.x, x2.y, x2.z, x2.w, x3.x, x3.y, x3.z, x3.w};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[r].x = fmaf(xv[r], gv.x, acc[r].x);
            acc[r].y = fmaf(xv[r], gv.y, acc[r].y);
            acc[r].z = fmaf(xv[r], gv.z, acc[r].z);
            acc[r].w = fmaf(xv[r], gv.w, acc[r].w);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) *reinterpret_cast<float4*>(out + r * 64) = acc[r];
}

constexpr int kSlices = 16;      // fixed K-split of the A^T.B reductions (partials added in slice order)
constexpr int kColSlices = 128;  // column sums: many more slices (the reduction is a pure stream)

}  // namespace
}  // namespace mdno

using namespace mdno;

extern "C" int mdno_linear_fwd(const float* a, const float* w, const float* bias, int64_t rows, int n, int k,
                               int relu, float* c, void* stream) {
    MDNO_REQUIRE(a && w && c && rows > 0 && n > 0 && k > 0, MDNO_EINVAL, "mdno_linear_fwd: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool mfma = n % 128 == 0 && k % 32 == 0 && rows < (1ll << 31) &&
                      ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(w)) & 15) == 0;
    if (mfma) {
        const size_t lds = sizeof(float) * 2 * 256 * LD;
        static std::atomic<unsigned long long> raised_relu{0}, raised_plain{0};
        MDNO_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(&linear_mfma_kernel<true>), (int)lds, raised_relu));
        MDNO_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(&linear_mfma_kernel<false>), (int)lds, raised_plain));
        dim3 grid(n / 128, (unsigned)((rows + 127) / 128));
        if (relu) hipLaunchKernelGGL(linear_mfma_kernel<true>, grid, dim3(256), lds, s, a, w, bias, c, (int)rows, n, k);
        else hipLaunchKernelGGL(linear_mfma_kernel<false>, grid, dim3(256), lds, s, a, w, bias, c, (int)rows, n, k);
    } else {
        const unsigned nb = (unsigned)((rows * n + 255) / 256);
        if (relu) hipLaunchKernelGGL(linear_generic_kernel<true>, dim3(nb), dim3(256), 0, s, a, w, bias, c, (long long)rows, n, k);
        else hipLaunchKernelGGL(linear_generic_kernel<false>, dim3(nb), dim3(256), 0, s, a, w, bias, c, (long long)rows, n, k);
    }
    return check_launch("mdno_linear_fwd");
}

extern "C" size_t mdno_linear_split_workspace_bytes(int64_t rows, int n, int k) {
    return rows > 0 && n > 0 && k > 0 ? split_linear_workspace_bytes((long long)rows, n, k) : 0;
}

extern "C" int mdno_linear_split_fwd(const float* a, const float* w, const float* bias, int64_t rows, int n, int k,
                                     int relu, float* c, void* workspace, size_t workspace_bytes, void* stream) {
    MDNO_REQUIRE(a && w && c && workspace && rows > 0 && n > 0 && k > 0, MDNO_EINVAL,
                 "mdno_linear_split_fwd: bad arguments");
    MDNO_REQUIRE(split_linear_supported((long long)rows, n, k) &&
                     ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(w)) & 15) == 0,
                 MDNO_EUNSUPPORTED, "mdno_linear_split_fwd: needs k %% 32 == 0, n %% 128 == 0, 16-byte aligned operands "
                 "(rows=%lld n=%d k=%d); use mdno_linear_fwd", (long long)rows, n, k);
    MDNO_REQUIRE(workspace_bytes >= split_linear_workspace_bytes((long long)rows, n, k), MDNO_EWORKSPACE,
                 "mdno_linear_split_fwd: workspace");
    return split_linear(a, w, bias, (long long)rows, n, k, relu, c, workspace, static_cast<hipStream_t>(stream));
}

extern "C" size_t mdno_linear_split_f16_workspace_bytes(int64_t rows, int n, int k) {
    return rows > 0 && n > 0 && k > 0 ? split_linear_f16_workspace_bytes((long long)rows, n, k) : 0;
}

extern "C" int mdno_linear_split_f16_fwd(const float* a, const float* w, const float* bias, int64_t rows, int n, int k,
                                         int relu, float* c, void* workspace, size_t workspace_bytes, void* stream) {
    MDNO_REQUIRE(a && w && c && workspace && rows > 0 && n > 0 && k > 0, MDNO_EINVAL,
                 "mdno_linear_split_f16_fwd: bad arguments");
    MDNO_REQUIRE(split_linear_supported((long long)rows, n, k) &&
                     ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(w)) & 15) == 0,
                 MDNO_EUNSUPPORTED, "mdno_linear_split_f16_fwd: needs k %% 32 == 0, n %% 128 == 0, 16-byte aligned "
                 "operands (rows=%lld n=%d k=%d); use mdno_linear_fwd", (long long)rows, n, k);
    MDNO_REQUIRE(workspace_bytes >= split_linear_f16_workspace_bytes((long long)rows, n, k), MDNO_EWORKSPACE,
                 "mdno_linear_split_f16_fwd: workspace");
    return split_linear_f16(a, w, bias, (long long)rows, n, k, relu, c, workspace, static_cast<hipStream_t>(stream));
}

extern "C" int mdno_gemm_atb_split_f16_supported(int64_t rows, int n1, int n2) {
    return gemm_atb_f16_supported((long long)rows, n1, n2) ? 1 : 0;
}

extern "C" size_t mdno_gemm_atb_split_f16_workspace_bytes(int64_t rows, int n1, int n2) {
    return gemm_atb_f16_supported((long long)rows, n1, n2) ? gemm_atb_f16_workspace_bytes((long long)rows, n1, n2) : 0;
}

extern "C" int mdno_gemm_atb_split_f16(const float* a, const float* b, int64_t rows, int n1, int n2, float* c, int accumulate,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    MDNO_REQUIRE(a && b && c && workspace && rows > 0, MDNO_EINVAL, "mdno_gemm_atb_split_f16: bad arguments");
    MDNO_REQUIRE(gemm_atb_f16_supported((long long)rows, n1, n2) &&
                     ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0,
                 MDNO_EUNSUPPORTED, "mdno_gemm_atb_split_f16: needs n1 %% 256 == 0, n2 %% 256 == 0, 16-byte aligned operands "
                 "(rows=%lld n1=%d n2=%d); use mdno_gemm_atb", (long long)rows, n1, n2);
    MDNO_REQUIRE(workspace_bytes >= gemm_atb_f16_workspace_bytes((long long)rows, n1, n2), MDNO_EWORKSPACE,
                 "mdno_gemm_atb_split_f16: workspace");
    return gemm_atb_f16(a, b, (long long)rows, n1, n2, c, accumulate, workspace, static_cast<hipStream_t>(stream));
}

extern "C" size_t mdno_reduce_workspace_bytes(int n1, int n2) {
    if (n2 <= 1) return align_up((size_t)kColSlices * (size_t)n1 * sizeof(float), 256);
    if (n2 <= 8) return align_up((size_t)kColSlices * (size_t)n1 * (size_t)n2 * sizeof(float), 256);   // gemm_atb_smalln_kernel
    return align_up((size_t)kSlices * (size_t)n1 * (size_t)n2 * sizeof(float), 256);
}

extern "C" int mdno_gemm_atb(const float* a, const float* b, int64_t rows, int n1, int n2, float* c, int accumulate,
                             void* workspace, size_t workspace_bytes, void* stream) {
    MDNO_REQUIRE(a && b && c && workspace && rows > 0 && n1 > 0 && n2 > 0, MDNO_EINVAL, "mdno_gemm_atb: bad arguments");
    MDNO_REQUIRE(workspace_bytes >= mdno_reduce_workspace_bytes(n1, n2), MDNO_EWORKSPACE, "mdno_gemm_atb: workspace");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(workspace);
    long long kslice = (rows + kSlices - 1) / kSlices;
    kslice = (kslice + BK - 1) / BK * BK;
    const bool mfma = n1 % 128 == 0 && n2 % 128 == 0 &&
                      ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
    if (!mfma && n2 <= 8) {      // few columns of B: a stream over A, many more slices
        const long long ks = (rows + kColSlices - 1) / kColSlices;
        hipLaunchKernelGGL(gemm_atb_smalln_kernel, dim3((n1 + 255) / 256, kColSlices), dim3(256), 0, s, a, b, part,
                           (long long)rows, n1, n2, ks);
        const long long cnt = (long long)n1 * n2;
        launch_reduce_slices((const float*)part, kColSlices, cnt, c, accumulate, s);
        return check_launch("mdno_gemm_atb");
    }
    if (mfma)
        hipLaunchKernelGGL(gemm_atb_mfma_kernel, dim3(n2 / 128, n1 / 128, kSlices), dim3(256), 0, s, a, b, part,
                           (long long)rows, n1, n2, kslice);
    else
        hipLaunchKernelGGL(gemm_atb_generic_kernel, dim3((n1 * n2 + 255) / 256, kSlices), dim3(256), 0, s, a, b, part,
                           (long long)rows, n1, n2, kslice);
    const long long count = (long long)n1 * n2;
    launch_reduce_slices((const float*)part, kSlices, count, c, accumulate, s);
    return check_launch("mdno_gemm_atb");
}

extern "C" int mdno_colsum(const float* a, int64_t rows, int n, float* out, int accumulate, void* workspace,
                           size_t workspace_bytes, void* stream) {
    MDNO_REQUIRE(a && out && workspace && rows > 0 && n > 0, MDNO_EINVAL, "mdno_colsum: bad arguments");
    MDNO_REQUIRE(workspace_bytes >= mdno_reduce_workspace_bytes(n, 1), MDNO_EWORKSPACE, "mdno_colsum: workspace");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(workspace);
    const long long kslice = (rows + kColSlices - 1) / kColSlices;
    if (n % 4 == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0)
        hipLaunchKernelGGL(colsum4_kernel, dim3((n / 4 + 255) / 256, kColSlices), dim3(256), 0, s, a, part, (long long)rows, n,
                           kslice);
    else
        hipLaunchKernelGGL(colsum_kernel, dim3((n + 255) / 256, kColSlices), dim3(256), 0, s, a, part, (long long)rows, n,
                           kslice);
    launch_reduce_slices((const float*)part, kColSlices, (long long)n, out, accumulate, s);
    return check_launch("mdno_colsum");
}

extern "C" int mdno_relu_bwd(const float* g, const float* y, const float* row_scale, int64_t rows, int n, float* out,
                             void* stream) {
    MDNO_REQUIRE(g && y && out && rows > 0 && n > 0 && n % 4 == 0, MDNO_EINVAL, "mdno_relu_bwd: bad arguments (n % 4)");
    const long long quads = rows * n / 4;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), g, y, row_scale, out, (long long)rows, n);
    return check_launch("mdno_relu_bwd");
}

extern "C" int mdno_relu_bwd2(const float* g, const float* y, const float* row_scale, int64_t rows, int n, float* gz,
                              float* gs, void* stream) {
    MDNO_REQUIRE(g && y && row_scale && gz && gs && rows > 0 && n > 0 && n % 4 == 0, MDNO_EINVAL,
                 "mdno_relu_bwd2: bad arguments (n % 4)");
    const long long quads = rows * n / 4;
    hipLaunchKernelGGL(relu_bwd2_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), g, y, row_scale, gz, gs, (long long)rows, n);
    return check_launch("mdno_relu_bwd2");
}

extern "C" int mdno_transpose(const float* a, int rows, int cols, float* at, void* stream) {
    MDNO_REQUIRE(a && at && rows > 0 && cols > 0, MDNO_EINVAL, "mdno_transpose: bad arguments");
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a, at, rows, cols);
    return check_launch("mdno_transpose");
}

extern "C" int mdno_inv_degree(const int32_t* row_ptr, int rows, int aggr, float* inv, void* stream) {
    MDNO_REQUIRE(row_ptr && inv && rows > 0, MDNO_EINVAL, "mdno_inv_degree: bad arguments");
    hipLaunchKernelGGL(inv_degree_kernel, dim3((rows + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       row_ptr, rows, aggr == MDNO_AGGR_MEAN ? 1 : 0, inv);
    return check_launch("mdno_inv_degree");
}

extern "C" int mdno_nnconv_bwd_x(const float* gz, const float* gs, const int32_t* row_ptr_s, const int32_t* eid_s,
                                 const int32_t* dst_s, int num_rows, const float* w_e, const float* root,
                                 int Cin, int Cout, float* g_prev, void* stream) {
    MDNO_REQUIRE(gz && gs && row_ptr_s && eid_s && dst_s && w_e && g_prev && num_rows > 0, MDNO_EINVAL,
                 "mdno_nnconv_bwd_x: bad arguments");
    MDNO_REQUIRE(Cin == 64 && Cout == 64, MDNO_EUNSUPPORTED, "mdno_nnconv_bwd_x: only 64x64 channels");
    hipLaunchKernelGGL(nnconv_bwd_x_kernel, dim3(num_rows), dim3(256), 0, static_cast<hipStream_t>(stream), gz, gs,
                       row_ptr_s, eid_s, dst_s, w_e, root, g_prev, num_rows);
    return check_launch("mdno_nnconv_bwd_x");
}

// rows per workgroup of nnconv_bwd_root_kernel: 256 (four 64-row passes) — 1,024 left cfg4's 21,504 stacked
// rows to 21 workgroups on 256 CUs (174 us per call); 128 moved the time into the serial slice sums
constexpr long long kRootSliceRows = 256;

extern "C" size_t mdno_nnconv_bwd_root_workspace_bytes(int64_t rows) {
    const long long blocks = (rows + kRootSliceRows - 1) / kRootSliceRows;
    return align_up((size_t)blocks * (4096 + 64) * sizeof(float), 256);
}

extern "C" int mdno_nnconv_bwd_root(const float* x, const float* gz, int64_t rows, int Cin, int Cout, float* d_root,
                                    float* d_bias, int accumulate, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    MDNO_REQUIRE(x && gz && rows > 0 && workspace, MDNO_EINVAL, "mdno_nnconv_bwd_root: bad arguments");
    MDNO_REQUIRE(Cin == 64 && Cout == 64, MDNO_EUNSUPPORTED, "mdno_nnconv_bwd_root: only 64x64 channels");
    MDNO_REQUIRE(workspace_bytes >= mdno_nnconv_bwd_root_workspace_bytes(rows), MDNO_EWORKSPACE,
                 "mdno_nnconv_bwd_root: workspace");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long slice_rows = kRootSliceRows;
    const int blocks = (int)((rows + slice_rows - 1) / slice_rows);
    float* part_root = static_cast<float*>(workspace);
    float* part_bias = part_root + (size_t)blocks * 4096;
    hipLaunchKernelGGL(nnconv_bwd_root_kernel, dim3(blocks), dim3(256), 0, s, x, gz, (long long)rows, slice_rows,
                       part_root, part_bias);
    if (d_root)
        launch_reduce_slices((const float*)part_root, blocks, 4096ll, d_root, accumulate, s);
    if (d_bias)
        launch_reduce_slices((const float*)part_bias, blocks, 64ll, d_bias, accumulate, s);
    return check_launch("mdno_nnconv_bwd_root");
}

// conv1's and conv2's root / bias gradients in ONE launch: x, gz [2 * rows_each, 64], the first rows_each rows conv1's
// stacked layers, the rest conv2's (they are adjacent in the training step's layer stack).  The slices of a half never
// cross into the other, and each half's partial sums are the ones mdno_nnconv_bwd_root forms for it alone (same slice
// boundaries, same order): bitwise the two single calls, one 36 us launch less per batch.
extern "C" size_t mdno_nnconv_bwd_root_pair_workspace_bytes(int64_t rows_each) {
    const long long blocks = 2 * ((rows_each + kRootSliceRows - 1) / kRootSliceRows);
    return align_up((size_t)blocks * (4096 + 64) * sizeof(float), 256);
}

extern "C" int mdno_nnconv_bwd_root_pair(const float* x, const float* gz, int64_t rows_each, float* d_root1, float* d_bias1,
                                         float* d_root2, float* d_bias2, void* workspace, size_t workspace_bytes, void* stream) {
    MDNO_REQUIRE(x && gz && rows_each > 0 && workspace && d_root1 && d_bias1 && d_root2 && d_bias2, MDNO_EINVAL,
                 "mdno_nnconv_bwd_root_pair: bad arguments");
    MDNO_REQUIRE(workspace_bytes >= mdno_nnconv_bwd_root_pair_workspace_bytes(rows_each), MDNO_EWORKSPACE,
                 "mdno_nnconv_bwd_root_pair: workspace");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int per_half = (int)((rows_each + kRootSliceRows - 1) / kRootSliceRows);
    float* part_root = static_cast<float*>(workspace);
    float* part_bias = part_root + (size_t)2 * per_half * 4096;
    hipLaunchKernelGGL(nnconv_bwd_root_pair_kernel, dim3(2 * per_half), dim3(256), 0, s, x, gz, (long long)rows_each,
                       (long long)kRootSliceRows, per_half, part_root, part_bias);
    launch_reduce_slices((const float*)part_root, per_half, 4096ll, d_root1, 0, s);
    launch_reduce_slices((const float*)part_root + (size_t)per_half * 4096, per_half, 4096ll, d_root2, 0, s);
    launch_reduce_slices((const float*)part_bias, per_half, 64ll, d_bias1, 0, s);
    launch_reduce_slices((const float*)part_bias + (size_t)per_half * 64, per_half, 64ll, d_bias2, 0, s);
    return check_launch("mdno_nnconv_bwd_root_pair");
}

extern "C" int mdno_nnconv_bwd_we(const float* x, const float* gs, const int32_t* src, const int32_t* dst, int64_t E,
                                  int layers, int64_t layer_stride, int Cin, int Cout, float* d_we, int accumulate,
                                  void* stream) {
    MDNO_REQUIRE(x && gs && src && dst && d_we && E > 0 && layers > 0, MDNO_EINVAL, "mdno_nnconv_bwd_we: bad arguments");
    MDNO_REQUIRE(Cin == 64 && Cout == 64, MDNO_EUNSUPPORTED, "mdno_nnconv_bwd_we: only 64x64 channels");
    hipLaunchKernelGGL(nnconv_bwd_we_kernel, dim3((unsigned)((E + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, gs, src, dst, (long long)E, layers, (long long)layer_stride,
                       d_we, accumulate);
    return check_launch("mdno_nnconv_bwd_we");
}

// ---------------------------------------------------------------- the conv applications of a training step as ONE call
extern "C" int mdno_nnconv_chain_fwd(float* x_layers, const int32_t* row_ptr, const int32_t* src, int num_rows,
                                     const float* w_e, const float* root1, const float* bias1, const float* root2,
                                     const float* bias2, int depth, void* stream) {
    MDNO_REQUIRE(x_layers && row_ptr && src && w_e && num_rows > 0 && depth > 0, MDNO_EINVAL,
                 "mdno_nnconv_chain_fwd: bad arguments");
    const size_t stride = (size_t)num_rows * 64;
    for (int a = 1; a <= 2 * depth; ++a)
        MDNO_TRY(mdno_nnconv_fwd(x_layers + (a - 1) * stride, row_ptr, src, num_rows, w_e, a <= depth ? root1 : root2,
                                 a <= depth ? bias1 : bias2, 64, 64, MDNO_AGGR_MEAN, 1, x_layers + a * stride, stream));
    return MDNO_OK;
}

extern "C" int mdno_nnconv_chain_bwd(const float* g_out, const float* x_layers, const float* inv_deg,
                                     const int32_t* row_ptr_s, const int32_t* eid_s, const int32_t* dst_s, int num_rows,
                                     const float* w_e, const float* root1, const float* root2, int depth, float* gz,
                                     float* gs, float* g_in, void* stream) {
    MDNO_REQUIRE(g_out && x_layers && inv_deg && row_ptr_s && eid_s && dst_s && w_e && gz && gs && g_in && num_rows > 0 &&
                     depth > 0, MDNO_EINVAL, "mdno_nnconv_chain_bwd: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int L = 2 * depth;
    const size_t stride = (size_t)num_rows * 64;
    MDNO_TRY(mdno_relu_bwd2(g_out, x_layers + L * stride, inv_deg, num_rows, 64, gz + (L - 1) * stride, gs + (L - 1) * stride,
                            stream));
    for (int a = L; a >= 1; --a) {
        const float* root = a <= depth ? root1 : root2;
        if (a > 1)
            hipLaunchKernelGGL(nnconv_bwd_x_kernel, dim3(num_rows), dim3(256), 0, s, gz + (a - 1) * stride, gs + (a - 1) * stride,
                               row_ptr_s, eid_s, dst_s, w_e, root, (float*)nullptr, num_rows, x_layers + (a - 1) * stride,
                               inv_deg, gz + (a - 2) * stride, gs + (a - 2) * stride);
        else
            hipLaunchKernelGGL(nnconv_bwd_x_kernel, dim3(num_rows), dim3(256), 0, s, gz, gs, row_ptr_s, eid_s, dst_s, w_e, root,
                               g_in, num_rows, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr);
    }
    return check_launch("mdno_nnconv_chain_bwd");
}
