// Factored evaluation of the kernel-integral conv: the same function as
//     m_e = x_j . reshape(W3 h_e + b3, [Cin,Cout])      (graph_kernel.py:201-202, W3 = net.layers.4)
// without ever forming the [E, Cin*Cout] edge weights.  With h_e = relu(L2(relu(L1(attr_e)))) in R^k,
//     m_e[o] = sum_c h_e[c] * Y_j[o,c] + q_j[o],   Y_j[o,c] = sum_i x_j[i] W3[i*Cout+o, c],
//                                                  q_j[o]   = sum_i x_j[i] b3[i*Cout+o],
// where j = source of edge e.  Y and q depend on the NODE only, so per conv application
//     (1) Y = X . W3T            one GEMM   [R, Cin] x [Cin, Cout*k]          2*R*Cin*Cout*k flop
//     (2) M_j = H_j . Y_j^T      one GEMM per source j over its own edges      2*E*k*Cout flop
//     (3) y_t = act(mean_{e->t} m_e + x_t.root + bias)                         gather of 256-B rows
// = 12 GFLOP per application at N=504, E=60.6k, k=1024 — against 518 GFLOP once for W_e plus a
// 1 GB stream per application in the materialised form.  This is a reassociation of the reference's
// sums (the contraction over c is done before the one over i), so values agree to fp32 rounding.
//
// Requirements: a SYMMETRIC graph in CSR with ascending columns (radius graphs): row r is read as
// "source r -> destinations col[p]", edges of one source are contiguous (what (2) needs), and the
// in-edges of node t are the reverses of row t's entries, found once per graph (rev[p] = position of
// r inside row col[p], binary search).  H (= the edge-MLP's last hidden activation, fp32 [E,k]) is
// produced in this source-major order by evaluating the MLP on attr = [pos[row], pos[col]].
//
// This file is the SOURCE-SIDE form on the exact fp32 MFMA (v_mfma_f32_32x32x2_f32, bit-for-bit an fmaf chain):
// what gemm_mode F32 runs — (1) gemm_rows_guarded_kernel, (2) gemm_per_source_kernel, (3) aggregate_rev_kernel.
// The split GEMM modes (the default) run the DESTINATION-SIDE form of moment.hip, which superseded this file's
// bf16-plane kernels in round 4 (same sums, fewer bytes, no reverse-edge index).
// (2) is bound by the H / Y stream (4 KiB of H per edge per application) next to the fp32 matrix rate.
#include "kernels.h"
#include "mfma_f32.h"

namespace mdno {
namespace {

using f32mma::f32x16;
using f32mma::mma_32x64;
using f32mma::mma_64x64;
constexpr int BK = f32mma::BK, LD = f32mma::LD;   // LDS rows padded to 36 floats (conflict-free ds_read_b128)

// ---------------------------------------------------------------- W3 [Cin*Cout, k] -> W3T [n', i]
// Row order n' = (c/32)*(C*32) + o*32 + c%32: the GEMM output row  Y[node][n']  is then already the
// k-tiled image  [k_tile][o][32]  that step (2) streams as contiguous 8 KiB pieces.
__global__ __launch_bounds__(256) void w3_transpose_kernel(const float* __restrict__ w3, int C, int k,
                                                           float* __restrict__ w3t) {
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;   // over (n', i), i fastest
    const long long total = (long long)C * k * C;
    if (id >= total) return;
    const int i = (int)(id % C);
    const long long n = id / C;
    const int cc = (int)(n % 32), o = (int)((n / 32) % C), kt = (int)(n / (32 * C));
    w3t[id] = w3[((size_t)i * C + o) * k + kt * 32 + cc];
}

// q[r][o] = sum_i x[r][i] * b3[i*64 + o]: the last MLP layer's bias seen through x_j (added to every
// message of source j by step (2)).  Thread = (row, o); used when the fused split+q kernel is not.
__global__ __launch_bounds__(256) void node_bias_kernel(const float* __restrict__ x, const float* __restrict__ b3,
                                                        int rows, float* __restrict__ q) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), o = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * 64;
    float acc = 0.f;
#pragma unroll 16
    for (int i = 0; i < 64; ++i) acc = fmaf(xr[i], b3[i * 64 + o], acc);
    q[(size_t)row * 64 + o] = acc;
}

// ---------------------------------------------------------------- (1) Y = X . W3T^T, rows guarded
// C[m][n] = sum_kk A[m][kk] * Bt[n][kk];  A [rows, K], Bt [N, K], C [rows, N]; N % 128 == 0, K % 32 == 0.
// 128x128x32 tile, 4 waves (2x2 of 64x64), register staging, double-buffered LDS (as edge_mlp.hip).
__global__ __launch_bounds__(256, 2) void gemm_rows_guarded_kernel(const float* __restrict__ A,
                                                                   const float* __restrict__ Bt,
                                                                   float* __restrict__ Cm, int rows, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * 128 * LD;
    const int bm = blockIdx.y * 128, bn = blockIdx.x * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    const size_t ldk = (size_t)K;
    // rows past the end re-read the last valid row (never stored)
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
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = bn + wn * 64 + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = bm + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < rows) Cm[(size_t)m * N + n] = acc[i][j][e];
            }
    }
}

// ---------------------------------------------------------------- (2) grouped: M_j = H_j . Y_j^T + q_j
// Workgroup (m-tile, source j, k-slice): rows beg_j + 128*mt .. of H [E,k] against Y_j [64,k] over
// k in [slice*k/KS, (slice+1)*k/KS) -> partial sums Mp[slice][E][64] (added in fixed order by the
// aggregation kernel; slice 0 carries q_j).  Both operands are stored k-tiled — H as
// [e/128][k/32][128][32], Y_j as [k/32][64][32] — so a K-tile is one contiguous 16 KiB / 8 KiB run
// (row-major H cost 128-B granules at a 4 KiB stride: 1.5 TB/s).  Tile 128 x 64 x 32, 4 waves, wave w owns rows
// 32w..32w+31 and both 32-column halves.  The GEMM reads every H row and every Y_j once and needs
// 32 flop per H byte, so with the exact fp32 MFMA it sits at the corner of HBM and the fp32 matrix
// rate (123 us at shape B: 40 % of either): many small workgroups (single LDS buffer, 27 KiB, 5 per
// CU; k split KS ways) keep enough loads in flight.
constexpr int KS = 2;

__global__ __launch_bounds__(256, 4) void gemm_per_source_kernel(const float* __restrict__ Hm,
                                                                 const float* __restrict__ Y,
                                                                 const float* __restrict__ Q,
                                                                 const int* __restrict__ row_ptr,
                                                                 float* __restrict__ Mp, long long part_stride,
                                                                 int K, int row0, int* __restrict__ status) {
    __shared__ __attribute__((aligned(16))) float As[128 * LD];
    __shared__ __attribute__((aligned(16))) float Bs[64 * LD];
    // source j is the fastest grid dimension: workgroups are dealt round-robin over the 8 XCDs by
    // linear id, and with the m-tile fastest (most sources have one tile) 3/4 of the work landed on
    // two XCDs.  The launch covers sources row0 .. row0+gridDim.x-1; Y holds that chunk only.
    const int j = row0 + blockIdx.x, slice = blockIdx.z;
    const int beg = row_ptr[j], end = row_ptr[j + 1];
    const int r0 = beg + blockIdx.y * 128;
    if (blockIdx.y == gridDim.y - 1 && slice == 0 && threadIdx.x == 0 && end - beg > (int)gridDim.y * 128 && status)
        atomicOr(status, MDNO_STATUS_DEGREE_OVERFLOW);   // max_degree bound too small: edges would be dropped
    if (r0 >= end) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    const int nkt = K / BK, nk = nkt / KS, kt0 = slice * nk;
    // element (e, kt) of the k-tiled H: ((e>>7)*nkt + kt)*4096 + (e&127)*32 floats; rows past the
    // group's end re-read its last row (never stored)
    auto aptr = [&](int r) {
        const int rr = r0 + r, e = rr < end ? rr : end - 1;
        return Hm + ((size_t)(e >> 7) * nkt + kt0) * 4096 + (e & 127) * 32 + scol;
    };
    const float* A0 = aptr(srow);
    const float* A1 = aptr(srow + 32);
    const float* A2 = aptr(srow + 64);
    const float* A3 = aptr(srow + 96);
    const float* Bg = Y + (size_t)(j - row0) * 64 * K + (size_t)kt0 * 2048 + srow * 32 + scol;
    float4 ra0, ra1, ra2, ra3, rb0, rb1;
#define MDNO_LOAD(KT)                                                         \
    ra0 = *reinterpret_cast<const float4*>(A0 + (size_t)(KT) * 4096);         \
    ra1 = *reinterpret_cast<const float4*>(A1 + (size_t)(KT) * 4096);         \
    ra2 = *reinterpret_cast<const float4*>(A2 + (size_t)(KT) * 4096);         \
    ra3 = *reinterpret_cast<const float4*>(A3 + (size_t)(KT) * 4096);         \
    rb0 = *reinterpret_cast<const float4*>(Bg + (size_t)(KT) * 2048);         \
    rb1 = *reinterpret_cast<const float4*>(Bg + (size_t)(KT) * 2048 + 1024);
    float* a_st = As + srow * LD + scol;
    float* b_st = Bs + srow * LD + scol;
#define MDNO_STORE()                                                  \
    *reinterpret_cast<float4*>(a_st) = ra0;                           \
    *reinterpret_cast<float4*>(a_st + 32 * LD) = ra1;                 \
    *reinterpret_cast<float4*>(a_st + 64 * LD) = ra2;                 \
    *reinterpret_cast<float4*>(a_st + 96 * LD) = ra3;                 \
    *reinterpret_cast<float4*>(b_st) = rb0;                           \
    *reinterpret_cast<float4*>(b_st + 32 * LD) = rb1;
    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    // q_j (bias of the last MLP layer seen through x_j) rides on the first k-slice.  Loaded before
    // the K loop and pinned: sunk into the predicated store blocks of the epilogue, the load would put
    // an s_waitcnt vmcnt(0) — a full store round trip — in front of every store.
    float q0 = 0.f, q1 = 0.f;
    if (slice == 0) {
        q0 = Q[(size_t)j * 64 + l31];
        q1 = Q[(size_t)j * 64 + 32 + l31];
    }
    asm volatile("" : "+v"(q0), "+v"(q1));
    const float* a_rd = As + (wave * 32 + l31) * LD + 4 * h;
    const float* b_rd = Bs + l31 * LD + 4 * h;
    MDNO_LOAD(0)
    MDNO_STORE()
    __syncthreads();
    // a wave whose 32 rows all lie past the group's end stages data but issues no MFMAs (the
    // matrix pipe of its SIMD goes to other workgroups): padding costs per 32 rows, not per 128
    const bool rows_live = __builtin_amdgcn_readfirstlane(r0 + wave * 32) < end;
    for (int kt = 0; kt < nk - 1; ++kt) {
        MDNO_LOAD(kt + 1)
        __builtin_amdgcn_sched_barrier(0);      // keep the prefetch above the MFMAs
        if (rows_live) mma_32x64(acc0, acc1, a_rd, b_rd);
        __syncthreads();
        MDNO_STORE()
        __syncthreads();
    }
    if (rows_live) mma_32x64(acc0, acc1, a_rd, b_rd);
#undef MDNO_LOAD
#undef MDNO_STORE
    float* Mo = Mp + (size_t)slice * part_stride;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = r0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m < end) {
            Mo[(size_t)m * 64 + l31] = acc0[e] + q0;
            Mo[(size_t)m * 64 + 32 + l31] = acc1[e] + q1;
        }
    }
}

// ---------------------------------------------------------------- reverse-edge index
// rev[p] for entry p = (row r, col c): position of r inside row c (exists iff the graph is symmetric);
// a missing reverse sets the status bit and points rev[p] at p.
__global__ __launch_bounds__(256) void reverse_edges_kernel(const int* __restrict__ row_ptr,
                                                            const int* __restrict__ col,
                                                            const int* __restrict__ rowid, int num_rows,
                                                            int* __restrict__ rev, int* __restrict__ status) {
    const int E = row_ptr[num_rows];
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= E) return;
    const int r = rowid[p], c = col[p];
    int lo = row_ptr[c], hi = row_ptr[c + 1];
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (col[mid] < r) lo = mid + 1; else hi = mid;
    }
    int at = lo;
    if (!(lo < row_ptr[c + 1] && col[lo] == r)) {
        at = p;
        if (status) atomicOr(status, MDNO_STATUS_ASYMMETRIC_GRAPH);
    }
    rev[p] = at;
}

// ---------------------------------------------------------------- (3) aggregate + root + bias + act
// One workgroup (32 chains x 16 lanes) per destination row; thread = (es, q): thirty-two in-edges are gathered per
// step (es = tid>>4), 16 B of the 64-float message per thread (q = tid&15), four steps in flight.
// The rows gathered are 256 B each at random positions of M, so the kernel is bound by how many
// loads are outstanding, not by bytes.  Each es-chain adds its edges in row order (and an edge's KS k-slice partials
// in slice order); the 32 chains are then added in es order through LDS — a fixed order, so the
// result is deterministic.  The root term x_t.root is accumulated the same way (es picks 2 of the
// 64 input channels).
constexpr int AGG_CHAINS = 32;      // summation chains per destination row (es), 16 threads (q) each

__global__ __launch_bounds__(AGG_CHAINS * 16) void aggregate_rev_kernel(const float* __restrict__ Mp, long long part_stride,
                                                            const int* __restrict__ rev,
                                                            const int* __restrict__ row_ptr,
                                                            const float* __restrict__ x,
                                                            const float* __restrict__ root,
                                                            const float* __restrict__ bias, float* __restrict__ y,
                                                            int num_rows, int aggr, int relu) {
    constexpr int CPT = 64 / AGG_CHAINS;     // input channels per thread in the root product
    __shared__ float4 part[AGG_CHAINS][16];
    const int tid = threadIdx.x, es = tid >> 4, q = tid & 15;
    const int t = blockIdx.x;
    const int beg = row_ptr[t], end = row_ptr[t + 1];
    const int deg = end - beg;
    // the root block, the row's input features and the bias are fetched NOW: their latency hides behind the message loop
    float4 rootv[CPT], biasv = make_float4(0.f, 0.f, 0.f, 0.f);
    float xin[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) { rootv[i] = make_float4(0.f, 0.f, 0.f, 0.f); xin[i] = 0.f; }
    if (root != nullptr) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            xin[i] = x[(size_t)t * 64 + CPT * es + i];
            rootv[i] = *reinterpret_cast<const float4*>(root + (CPT * es + i) * 64 + 4 * q);
        }
    }
    if (bias != nullptr && es == 0) biasv = *reinterpret_cast<const float4*>(bias + 4 * q);
    struct Msg { float4 v[KS]; };
    auto fetch = [&](int rp) {
        Msg g;
        const float* m = Mp + (size_t)rp * 64 + 4 * q;
#pragma unroll
        for (int k = 0; k < KS; ++k) g.v[k] = *reinterpret_cast<const float4*>(m + (size_t)k * part_stride);
        return g;
    };
    auto total = [](const Msg& g) {
        float4 e = g.v[0];
#pragma unroll
        for (int k = 1; k < KS; ++k) { e.x += g.v[k].x; e.y += g.v[k].y; e.z += g.v[k].z; e.w += g.v[k].w; }
        return e;
    };
    // Chain es adds entries beg+es, beg+es+32, ... in that order.  Batches of four: the four rev[] words,
    // then every plane of the four messages, are in flight together.
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = beg + es; p < end; p += 4 * AGG_CHAINS) {
        int rp[4];
        bool on[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            on[k] = p + k * AGG_CHAINS < end;
            rp[k] = on[k] ? rev[p + k * AGG_CHAINS] : 0;
        }
        Msg g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (on[k]) g[k] = fetch(rp[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (on[k]) {
                const float4 e = total(g[k]);
                acc.x += e.x; acc.y += e.y; acc.z += e.z; acc.w += e.w;
            }
    }
    part[es][q] = acc;
    __syncthreads();
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (es == 0) {
#pragma unroll
        for (int c = 0; c < AGG_CHAINS; ++c) {
            const float4 v = part[c][q];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (aggr == MDNO_AGGR_MEAN) {
            const float inv = (float)(deg > 1 ? deg : 1);
            s.x /= inv; s.y /= inv; s.z /= inv; s.w /= inv;
        }
    }
    if (root != nullptr) {
        __syncthreads();
        float4 racc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            racc.x = fmaf(xin[i], rootv[i].x, racc.x); racc.y = fmaf(xin[i], rootv[i].y, racc.y);
            racc.z = fmaf(xin[i], rootv[i].z, racc.z); racc.w = fmaf(xin[i], rootv[i].w, racc.w);
        }
        part[es][q] = racc;
        __syncthreads();
        if (es == 0) {
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int c = 0; c < AGG_CHAINS; ++c) {
                const float4 v = part[c][q];
                r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w;
            }
            s.x += r.x; s.y += r.y; s.z += r.z; s.w += r.w;
        }
    }
    if (es == 0) {
        if (bias != nullptr) { s.x += biasv.x; s.y += biasv.y; s.z += biasv.z; s.w += biasv.w; }
        if (relu) { s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f); }
        *reinterpret_cast<float4*>(y + (size_t)t * 64 + 4 * q) = s;
    }
}

}  // namespace

// ---------------------------------------------------------------- host side
bool factored_supported(int width, int ker_width) { return width == 64 && ker_width % (KS * BK) == 0; }

// Steps (1) and (2) run chunk by chunk over the sources: a chunk's Y (512 rows x 256 KiB = 128 MiB at
// k = 1024) is written by the Y GEMM and read back by the per-source GEMM right behind it, while it is
// still in the 256 MiB Infinity Cache — and always at the same addresses, so an ensemble of any size
// keeps ONE chunk of Y alive.  Chunks are cut at multiples of the GEMM's row tile, not at member boundaries; a
// source's arithmetic does not depend on the chunk it is in.
constexpr int kYChunkRows = 512;
static int y_chunk_rows(int num_rows) { return num_rows < kYChunkRows ? num_rows : kYChunkRows; }

size_t factored_workspace_bytes(int num_rows, int ker_width, long long edge_cap) {
    Carver cv(nullptr);
    cv.take<float>((size_t)64 * ker_width * 64);               // W3T
    cv.take<float>((size_t)y_chunk_rows(num_rows) * 64 * ker_width);   // Y, one chunk of sources
    cv.take<float>((size_t)num_rows * 64);                     // q
    cv.take<float>((size_t)KS * edge_cap * 64);                // M: k-slice partials, one plane each
    cv.take<int>((size_t)edge_cap);                            // rev
    return cv.used();
}

FactoredWs factored_carve(void* ws, int num_rows, int ker_width, long long edge_cap) {
    FactoredWs f{};
    Carver cv(ws);
    f.w3t = cv.take<float>((size_t)64 * ker_width * 64);
    f.y = cv.take<float>((size_t)y_chunk_rows(num_rows) * 64 * ker_width);
    f.q = cv.take<float>((size_t)num_rows * 64);
    f.m = cv.take<float>((size_t)KS * edge_cap * 64);
    f.part_stride = (long long)edge_cap * 64;
    f.rev = cv.take<int>((size_t)edge_cap);
    return f;
}

int factored_prepare_weights(const float* w3, int ker_width, const FactoredWs& f, hipStream_t s) {
    const long long total = (long long)64 * ker_width * 64;
    hipLaunchKernelGGL(w3_transpose_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w3, 64, ker_width,
                       f.w3t);
    return check_launch("w3_transpose_kernel");
}

int factored_prepare_graph(const int* row_ptr, const int* col, const int* rowid, int num_rows, long long edge_cap,
                           const FactoredWs& f, int* status, hipStream_t s) {
    TimedSection ts(KID_GRAPH, s);
    hipLaunchKernelGGL(reverse_edges_kernel, dim3((unsigned)((edge_cap + 255) / 256)), dim3(256), 0, s, row_ptr, col,
                       rowid, num_rows, f.rev, status);
    return check_launch("reverse_edges_kernel");
}

int factored_conv(const float* x, const float* h2, const int* row_ptr, int num_rows, int max_degree, int ker_width,
                  const float* b3, const float* root, const float* bias, int aggr, int relu, float* y,
                  const FactoredWs& f, int* status, hipStream_t s) {
    const size_t lds1 = sizeof(float) * 2 * 256 * LD;   // 73,728 B
    static_assert(kYChunkRows % 128 == 0, "Y chunk must be a multiple of the GEMM row tile");
    const int ncols = 64 * ker_width;
    static std::atomic<unsigned long long> lds_raised{0};
    MDNO_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(&gemm_rows_guarded_kernel), (int)lds1, lds_raised));
    {
        TimedSection ts(KID_FACT_Y, s);
        hipLaunchKernelGGL(node_bias_kernel, dim3((num_rows + 3) / 4), dim3(256), 0, s, x, b3, num_rows, f.q);
    }
    const int mtiles = (max_degree + 127) / 128;
    for (int r0 = 0; r0 < num_rows; r0 += kYChunkRows) {
        const int cnt = num_rows - r0 < kYChunkRows ? num_rows - r0 : kYChunkRows;
        {   // (1) Y of sources r0 .. r0+cnt-1
            TimedSection ts(KID_FACT_Y, s);
            hipLaunchKernelGGL(gemm_rows_guarded_kernel, dim3(ncols / 128, (cnt + 127) / 128), dim3(256), lds1, s,
                               x + (size_t)r0 * 64, (const float*)f.w3t, f.y, cnt, ncols, 64);
        }
        {   // (2) the per-source GEMMs of the same sources
            TimedSection ts(KID_NNCONV, s);
            hipLaunchKernelGGL(gemm_per_source_kernel, dim3(cnt, mtiles, KS), dim3(256), 0, s, h2, (const float*)f.y,
                               (const float*)f.q, row_ptr, f.m, f.part_stride, ker_width, r0, status);
        }
    }
    {
        TimedSection ts(KID_NNCONV_COMBINE, s);
        hipLaunchKernelGGL(aggregate_rev_kernel, dim3(num_rows), dim3(AGG_CHAINS * 16), 0, s, (const float*)f.m,
                           f.part_stride, (const int*)f.rev, row_ptr, x, root, bias, y, num_rows, aggr, relu);
    }
    return check_launch("factored_conv");
}

}  // namespace mdno
