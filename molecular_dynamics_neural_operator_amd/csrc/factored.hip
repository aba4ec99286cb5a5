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
// Two sets of kernels, chosen by gemm_mode (same sums, both fp32-accurate):
//   SPLIT_BF16 (default)  (1) = the split-bf16 GEMM of edge_mlp_split.hip on bf16 plane images of x and
//                         W3T; (2) = gemm_per_source_split_kernel, which splits the fp32 K-tiles of H
//                         and Y_j into three bf16 planes on the fly and runs the six plane products on
//                         the bf16 matrix pipe; (3) also prepares the next application's operands.
//   F32                   (1) gemm_rows_guarded_kernel, (2) gemm_per_source_kernel: exact fp32 MFMA
//                         (v_mfma_f32_32x32x2_f32), bit-for-bit an fmaf chain.
// (2) is bound by the H / Y stream (4 KiB of H per edge per application), not by the matrix pipe.
#include "kernels.h"
#include "mfma_f32.h"
#include "split_layout.h"

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
// CU; k split KS ways) keep enough loads in flight.  gemm_mode F32 only; the bf16 kernel below is the
// default.
constexpr int KS = 2;
// The bf16 kernel cuts k 2 ways for a source's first 128-row tile and 4 ways for its later tiles:
// those hold the few rows past 128 (a third of the sources at shape B), are dispatched last, and as
// 16-iteration workgroups they formed a 43 %-full second round as long as the first (an iteration
// costs 2.3-2.8 us whatever the number of live rows).  With 8 iterations the tail is half as long.
// The number of partials an entry has depends only on its own position in its row, so the result does
// not depend on what else is in the batch; the aggregation reads it from the top bits of rev[].
// Only done for small members (N*KS <= 2048 workgroups in the first tiles: one or two rounds of the
// chip), where the tail matters; a function of the member size alone, so batches stay bit-identical
// to their members run alone.
constexpr int KS_TAIL = 4, MAX_PLANES = 4, REV_SHIFT = 28;
constexpr int REV_MASK = (1 << REV_SHIFT) - 1;

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

// ---------------------------------------------------------------- (2) on the bf16 matrix pipe
// Same workgroup shape and the same fp32 operands in HBM as gemm_per_source_kernel, but each staged
// K-tile is split on the fly into three bf16 planes (x = hi + mid + lo exactly, edge_mlp_split.hip)
// and multiplied as the six leading plane products with fp32 accumulation: 12 cycles of matrix pipe
// per k instead of 32, for ~6 VALU ops per staged element.  That moves the kernel off the fp32 MFMA
// rate (where it sat at 40 %, next to its HBM time) and leaves the H/Y stream as the one bound.
// LDS: plane p of A = 128 rows x 64 B (32 k), of B = 64 rows x 64 B; the four 16-B chunks of a row
// are XOR-swizzled with (row>>2)&3, which makes both the ds_write_b64 of the staging threads and the
// ds_read_b128 fragment reads conflict-free without padding (36 KiB, 4 workgroups per CU).
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SPL_A_PLANE = 128 * 64, SPL_B_PLANE = 64 * 64, SPL_B_BASE = 3 * SPL_A_PLANE;

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// (a, b) -> packed bf16 pair (one v_cvt_pk_bf16_f32) and the pair's values back in fp32 (shift / mask)
__device__ __forceinline__ unsigned pack_bf16(float a, float b, float& fa, float& fb) {
    const f32x2 v = {a, b};
    const bf16x2 p = __builtin_convertvector(v, bf16x2);
    const unsigned u = __builtin_bit_cast(unsigned, p);
    fa = __builtin_bit_cast(float, u << 16);
    fb = __builtin_bit_cast(float, u & 0xffff0000u);
    return u;
}

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

__global__ __launch_bounds__(256, 4) void gemm_per_source_split_kernel(const float* __restrict__ Hm,
                                                                       const float* __restrict__ Y,
                                                                       const float* __restrict__ Q,
                                                                       const int* __restrict__ row_ptr,
                                                                       float* __restrict__ Mp, long long part_stride,
                                                                       int K, int ks_tail, int slots, int row0,
                                                                       int* __restrict__ status,
                                                                       const int* __restrict__ order) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * SPL_A_PLANE + 3 * SPL_B_PLANE];
    // slots of a source: [tile 0: slices 0..KS-1 | tile 1: slices 0..ks_tail-1 | tile 2 ...].
    // Small members: grid (source, slot), source fastest (XCD balance; first tiles before later ones).
    // Many rounds of workgroups (gridDim.y == 1: large members or many members): source-major — a source's workgroups are dispatched
    // together, so the second and third read of Y_j meet it in the Infinity Cache instead of HBM; the
    // slot is rotated by the source index so that the heavy first-tile slots visit all eight XCDs.
    int j = blockIdx.x, slot = blockIdx.y;
    if (gridDim.y == 1) {
        j = (int)(blockIdx.x / (unsigned)slots);
        slot = (int)((blockIdx.x % (unsigned)slots + (unsigned)j) % (unsigned)slots);
    }
    // the chunk's sources are visited in order of decreasing degree (source_order_kernel): workgroup ids
    // b, b+256, b+512, ... land on one CU, which then gets one source from every quarter of the sorted list
    // instead of four of any size — all workgroups of a launch are resident at once, so nothing else evens
    // out what a CU has to stream
    j = order[row0 + j];
    const int jl = j;      // index inside this launch's chunk of sources (Y holds the chunk only)
    j += row0;
    const int mt = slot < KS ? 0 : 1 + (slot - KS) / ks_tail;
    const int slice = slot < KS ? slot : (slot - KS) % ks_tail;
    const int ks = mt == 0 ? KS : ks_tail;
    const int beg = row_ptr[j], end = row_ptr[j + 1];
    const int r0 = beg + mt * 128;
    if (slot == slots - 1 && threadIdx.x == 0 && end - beg > (mt + 1) * 128 && status)
        atomicOr(status, MDNO_STATUS_DEGREE_OVERFLOW);
    if (r0 >= end) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    const int nkt = K / BK, nk = nkt / ks, kt0 = slice * nk;
    // 32-row groups that hold at least one of this source's rows: the others are neither loaded nor
    // split nor multiplied (their LDS rows feed only the wave that skips its MFMAs)
    const int live = (end - r0 + 31) >> 5;
    auto aptr = [&](int r) {
        const int rr = r0 + r, e = rr < end ? rr : end - 1;
        return Hm + ((size_t)(e >> 7) * nkt + kt0) * 4096 + (e & 127) * 32 + scol;
    };
    const float* A0 = aptr(srow);
    const float* A1 = aptr(srow + 32);
    const float* A2 = aptr(srow + 64);
    const float* A3 = aptr(srow + 96);
    const float* Bg = Y + (size_t)jl * 64 * K + (size_t)kt0 * 2048 + srow * 32 + scol;
    float4 ra0, ra1 = make_float4(0.f, 0.f, 0.f, 0.f), ra2 = ra1, ra3 = ra1, rb0, rb1;
    // H is streamed once per application: non-temporal loads (global_load ... nt) leave the caches to
    // Y_j and the partial sums (-4.5 % kernel time)
#define MDNO_NT(DST, P)                                                                             \
    {                                                                                               \
        const f32x4 t_ = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P));             \
        DST = make_float4(t_.x, t_.y, t_.z, t_.w);                                                  \
    }
#define MDNO_LOAD(KT)                                                                       \
    MDNO_NT(ra0, A0 + (size_t)(KT) * 4096)                                                  \
    if (live > 1) MDNO_NT(ra1, A1 + (size_t)(KT) * 4096)                                    \
    if (live > 2) MDNO_NT(ra2, A2 + (size_t)(KT) * 4096)                                    \
    if (live > 3) MDNO_NT(ra3, A3 + (size_t)(KT) * 4096)                                    \
    rb0 = *reinterpret_cast<const float4*>(Bg + (size_t)(KT) * 2048);                       \
    rb1 = *reinterpret_cast<const float4*>(Bg + (size_t)(KT) * 2048 + 1024);
    // staging thread (row, 4 k at scol): 8 bytes of chunk (scol>>3) of the row, swizzled
    auto st_off = [&](int row) { return row * 64 + ((((tid & 7) >> 1) ^ ((row >> 2) & 3)) << 4) + (tid & 1) * 8; };
    unsigned char* a_st0 = lds + st_off(srow);
    unsigned char* a_st1 = lds + st_off(srow + 32);
    unsigned char* a_st2 = lds + st_off(srow + 64);
    unsigned char* a_st3 = lds + st_off(srow + 96);
    unsigned char* b_st0 = lds + SPL_B_BASE + st_off(srow);
    unsigned char* b_st1 = lds + SPL_B_BASE + st_off(srow + 32);
#define MDNO_STORE()                                              \
    split_store4(ra0, a_st0, SPL_A_PLANE);                        \
    if (live > 1) split_store4(ra1, a_st1, SPL_A_PLANE);          \
    if (live > 2) split_store4(ra2, a_st2, SPL_A_PLANE);          \
    if (live > 3) split_store4(ra3, a_st3, SPL_A_PLANE);          \
    split_store4(rb0, b_st0, SPL_B_PLANE);                        \
    split_store4(rb1, b_st1, SPL_B_PLANE);
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
    // fragment reads: row r, k-step s (16 k), half h -> chunk 2s+h, swizzled
    const int arow = wave * 32 + l31, brow0 = l31, brow1 = 32 + l31;
    const int a_sw = (arow >> 2) & 3, b_sw0 = (brow0 >> 2) & 3, b_sw1 = (brow1 >> 2) & 3;
    const unsigned char* a_rd = lds + arow * 64;
    const unsigned char* b_rd0 = lds + SPL_B_BASE + brow0 * 64;
    const unsigned char* b_rd1 = lds + SPL_B_BASE + brow1 * 64;
#define MDNO_MMA6(A, B, ACC)                                                        \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[1], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[2], B[0], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[2], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[0], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[1], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[0], ACC, 0, 0, 0);
#define MDNO_MMA_TILE()                                                                                  \
    _Pragma("unroll") for (int st = 0; st < 2; ++st) {                                                   \
        bf16x8 a[3], b0[3], b1[3];                                                                       \
        _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                  \
            a[p] = *reinterpret_cast<const bf16x8*>(a_rd + p * SPL_A_PLANE + (((2 * st + h) ^ a_sw) << 4));   \
            b0[p] = *reinterpret_cast<const bf16x8*>(b_rd0 + p * SPL_B_PLANE + (((2 * st + h) ^ b_sw0) << 4)); \
            b1[p] = *reinterpret_cast<const bf16x8*>(b_rd1 + p * SPL_B_PLANE + (((2 * st + h) ^ b_sw1) << 4)); \
        }                                                                                                \
        MDNO_MMA6(a, b0, acc0) MDNO_MMA6(a, b1, acc1)                                                    \
    }
    MDNO_LOAD(0)
    MDNO_STORE()
    __syncthreads();
    const bool rows_live = wave < live;
    for (int kt = 0; kt < nk - 1; ++kt) {
        MDNO_LOAD(kt + 1)
        __builtin_amdgcn_sched_barrier(0);      // keep the prefetch above the MFMAs
        if (rows_live) { MDNO_MMA_TILE() }
        __syncthreads();
        MDNO_STORE()
        __syncthreads();
    }
    if (rows_live) { MDNO_MMA_TILE() }
#undef MDNO_LOAD
#undef MDNO_NT
#undef MDNO_STORE
#undef MDNO_MMA_TILE
#undef MDNO_MMA6
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

// ---------------------------------------------------------------- sources by decreasing degree
// One workgroup per Y chunk of kYChunkRows sources: order[chunk*kYChunkRows + rank] = index inside the
// chunk of the source with that rank (degree descending, ties by index).  Once per graph.
template <int CH>
__global__ __launch_bounds__(CH) void source_order_kernel(const int* __restrict__ row_ptr, int num_rows,
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

// ---------------------------------------------------------------- reverse-edge index
// rev[p] for entry p = (row r, col c): position of r inside row c (exists iff the graph is symmetric);
// a missing reverse sets the status bit and points rev[p] at p.
__global__ __launch_bounds__(256) void reverse_edges_kernel(const int* __restrict__ row_ptr,
                                                            const int* __restrict__ col,
                                                            const int* __restrict__ rowid, int num_rows,
                                                            int* __restrict__ rev, int* __restrict__ status,
                                                            int tail_planes, int* __restrict__ f16_x_flags) {
    // once per forward, before any conv application: no node feature has been seen out of fp16 range
    // yet (a kernel's stores rather than a memset node: the captured step stays a chain of kernels)
    if (blockIdx.x == 0 && threadIdx.x <= kMaxF16Applications) {
        f16_x_flags[threadIdx.x] = 0;
        f16_x_flags[F16_SEEN_OFF + threadIdx.x] = 0;
    }
    const int E = row_ptr[num_rows];
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= E) return;
    const int r = rowid[p], c = col[p];
    const int cbeg = row_ptr[c];
    int lo = cbeg, hi = row_ptr[c + 1];
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (col[mid] < r) lo = mid + 1; else hi = mid;
    }
    int at = lo, abeg = cbeg;
    if (!(lo < row_ptr[c + 1] && col[lo] == r)) {
        at = p;
        abeg = row_ptr[r];
        if (status) atomicOr(status, MDNO_STATUS_ASYMMETRIC_GRAPH);
    }
    // top bits: how many k-slice partials step (2) writes for that entry (first tile of its row or later)
    rev[p] = at | ((at - abeg < 128 ? KS : tail_planes) << REV_SHIFT);
}

// ---------------------------------------------------------------- (3) aggregate + root + bias + act
// One workgroup (8 waves) per destination row; thread = (es, q): thirty-two in-edges are gathered per
// step (es = tid>>4), 16 B of the 64-float message per thread (q = tid&15), four steps in flight.
// The rows gathered are 256 B each at random positions of M, so the kernel is bound by how many
// loads are outstanding, not by bytes: one wave per row (4 chains) took 35 us per application,
// this shape 3x less.  Each es-chain adds its edges in row order (and an edge's k-slice partials
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
                                                            int num_rows, int aggr, int relu,
                                                            const float* __restrict__ next_b3,
                                                            float* __restrict__ next_q,
                                                            unsigned char* __restrict__ next_xp,
                                                            unsigned char* __restrict__ next_xh,
                                                            int* __restrict__ next_flag) {
    constexpr int CPT = 64 / AGG_CHAINS;     // input channels per thread in the root / B3 products
    __shared__ float4 part[AGG_CHAINS][16];
    const int tid = threadIdx.x, es = tid >> 4, q = tid & 15;
    const int t = blockIdx.x;
    const int beg = row_ptr[t], end = row_ptr[t + 1];
    const int deg = end - beg;
    // Everything the tail of this kernel needs from memory besides the messages is fetched NOW, so that
    // its latency hides behind the message loop instead of forming a chain behind it: the thread's
    // CPT x 4 block of root and of the next application's B3 (input channels CPT*es.., output columns
    // 4q..4q+3), its input features and the bias.
    float4 rootv[CPT], b3v[CPT], biasv = make_float4(0.f, 0.f, 0.f, 0.f);
    float xin[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) { rootv[i] = b3v[i] = make_float4(0.f, 0.f, 0.f, 0.f); xin[i] = 0.f; }
    if (root != nullptr) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            xin[i] = x[(size_t)t * 64 + CPT * es + i];
            rootv[i] = *reinterpret_cast<const float4*>(root + (CPT * es + i) * 64 + 4 * q);
        }
    }
    if (next_b3 != nullptr) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) b3v[i] = *reinterpret_cast<const float4*>(next_b3 + (CPT * es + i) * 64 + 4 * q);
    }
    if (bias != nullptr && es == 0) biasv = *reinterpret_cast<const float4*>(bias + 4 * q);
    // Every entry has KS partials; an entry in a later tile of its row has MAX_PLANES (top bits of
    // rev[]; rare).  All loads of a step are issued before any is consumed — the extra planes under a
    // predicate, zero when absent — so a step costs one memory round trip either way; partials are
    // added in plane order.
    struct Msg { float4 v[MAX_PLANES]; };
    auto fetch = [&](int rp) {
        Msg g;
        const float* m = Mp + (size_t)(rp & REV_MASK) * 64 + 4 * q;
        const bool tail = (rp >> REV_SHIFT) > KS;
#pragma unroll
        for (int k = 0; k < MAX_PLANES; ++k) {
            g.v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < KS || tail) g.v[k] = *reinterpret_cast<const float4*>(m + (size_t)k * part_stride);
        }
        return g;
    };
    auto total = [](const Msg& g) {
        float4 e = g.v[0];
#pragma unroll
        for (int k = 1; k < MAX_PLANES; ++k) { e.x += g.v[k].x; e.y += g.v[k].y; e.z += g.v[k].z; e.w += g.v[k].w; }
        return e;
    };
    // Chain es adds entries beg+es, beg+es+32, ... in that order.  Batches of four: the four rev[] words,
    // then every plane of the four messages, are in flight together, so a batch costs two dependent
    // round trips whatever its size — and a row of up to 128 entries is ONE batch per chain (the first
    // version, 16 chains and a 4-2-1 ladder of loops, took three batches for a row of 120).
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
    // What the NEXT conv application needs from this row, while it is at hand (saves a launch per
    // application): q = y_t . B3 (bias of the last MLP layer seen through the node) and the plane
    // images of y_t (operand of the Y GEMM).  q is summed like the root term — sixteen 4-term partial
    // products (es picks the input channels) added in es order — from the B3 block fetched up front;
    // the first version ran 64 dependent-latency loads per thread behind everything else.
    if (next_b3 != nullptr) {
        __syncthreads();
        if (es == 0) part[0][q] = s;
        __syncthreads();
        const float* row = reinterpret_cast<const float*>(&part[0][0]);
        float4 qacc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const float yi = row[CPT * es + i];
            qacc.x = fmaf(yi, b3v[i].x, qacc.x); qacc.y = fmaf(yi, b3v[i].y, qacc.y);
            qacc.z = fmaf(yi, b3v[i].z, qacc.z); qacc.w = fmaf(yi, b3v[i].w, qacc.w);
        }
        // the plane images are cut from `row` by threads 64..71 while the partial products are parked
        // in a second LDS array
        __shared__ float4 qpart[AGG_CHAINS][16];
        qpart[es][q] = qacc;
        __syncthreads();
        if (es == 0) {
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int c = 0; c < AGG_CHAINS; ++c) {
                const float4 v = qpart[c][q];
                r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w;
            }
            *reinterpret_cast<float4*>(next_q + (size_t)t * 64 + 4 * q) = r;
        }
        if (tid >= 64 && tid < 72) {
            const int c = tid - 64;
            __bf16 pl[3][8];
#pragma unroll
            for (int j2 = 0; j2 < 8; ++j2) split3(row[8 * c + j2], pl[0][j2], pl[1][j2], pl[2][j2]);
#pragma unroll
            for (int p = 0; p < 3; ++p)
                *reinterpret_cast<uint4*>(next_xp + tiled_off(t, 8 * c, 4, p)) = *reinterpret_cast<const uint4*>(pl[p]);
            if (next_xh != nullptr) {
                _Float16 ph[2][8];
                bool bad = false, seen = false;
#pragma unroll
                for (int j2 = 0; j2 < 8; ++j2) {
                    bad |= !(fabsf(row[8 * c + j2]) < F16_MAX);
                    seen |= fabsf(row[8 * c + j2]) >= F16_ACT_MIN;
                    split2h(row[8 * c + j2], ph[0][j2], ph[1][j2]);
                }
                if (bad) atomicOr(next_flag, 1);
                if (seen) next_flag[F16_SEEN_OFF] = 1;
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    *reinterpret_cast<uint4*>(next_xh + tiled_off2(t, 8 * c, 4, p)) = *reinterpret_cast<const uint4*>(ph[p]);
            }
        }
    }
}

}  // namespace

// ---------------------------------------------------------------- host side
bool factored_supported(int width, int ker_width) { return width == 64 && ker_width % (KS_TAIL * BK) == 0; }

// Steps (1) and (2) run chunk by chunk over the sources: a chunk's Y (512 rows x 256 KiB = 128 MiB at
// k = 1024) is written by the Y GEMM and read back by the per-source GEMM right behind it, while it is
// still in the 256 MiB Infinity Cache — and always at the same addresses, so an ensemble of any size
// keeps ONE chunk of Y alive instead of one Y per member (8 members: 1 GB, evicted between producer and
// consumer).  Chunks are cut at multiples of the GEMM's 256-row tile, not at member boundaries; a
// source's arithmetic does not depend on the chunk it is in.
constexpr int kYChunkRows = 512;      // (256: 5 % slower at 8 members, 1024: the same)
static int y_chunk_rows(int num_rows) { return num_rows < kYChunkRows ? num_rows : kYChunkRows; }

size_t factored_workspace_bytes(int num_rows, int ker_width, long long edge_cap) {
    Carver cv(nullptr);
    cv.take<float>((size_t)64 * ker_width * 64);               // W3T
    cv.take<float>((size_t)y_chunk_rows(num_rows) * 64 * ker_width);   // Y, one chunk of sources
    cv.take<float>((size_t)num_rows * 64);                     // q
    cv.take<float>((size_t)MAX_PLANES * edge_cap * 64);        // M: k-slice partials, one plane each
    cv.take<int>((size_t)edge_cap);                            // rev
    cv.take<char>(split_planes_bytes((long long)64 * ker_width, 64));   // W3T as bf16 planes
    cv.take<char>(split_planes_bytes(num_rows, 64));                    // X as bf16 planes
    cv.take<char>(split_planes_f16_bytes((long long)64 * ker_width, 64));   // W3T as fp16 planes
    cv.take<char>(split_planes_f16_bytes(num_rows, 64));                    // X as fp16 planes
    cv.take<int>(128);                                                  // fp16 range flags + "seen" words
    cv.take<float>((size_t)64 * ker_width);                             // unscale factors of the fp16 W3T rows
    cv.take<int>((size_t)num_rows);                                     // sources of each Y chunk by decreasing degree
    return cv.used();
}

FactoredWs factored_carve(void* ws, int num_rows, int ker_width, long long edge_cap) {
    FactoredWs f{};
    Carver cv(ws);
    f.w3t = cv.take<float>((size_t)64 * ker_width * 64);
    f.y = cv.take<float>((size_t)y_chunk_rows(num_rows) * 64 * ker_width);
    f.q = cv.take<float>((size_t)num_rows * 64);
    f.m = cv.take<float>((size_t)MAX_PLANES * edge_cap * 64);
    f.part_stride = (long long)edge_cap * 64;
    f.rev = cv.take<int>((size_t)edge_cap);
    f.w3tp = cv.take<char>(split_planes_bytes((long long)64 * ker_width, 64));
    f.xp = cv.take<char>(split_planes_bytes(num_rows, 64));
    f.w3th = cv.take<char>(split_planes_f16_bytes((long long)64 * ker_width, 64));
    f.xh = cv.take<char>(split_planes_f16_bytes(num_rows, 64));
    f.f16_flags = cv.take<int>(128);
    f.w3tus = cv.take<float>((size_t)64 * ker_width);
    f.order = cv.take<int>((size_t)num_rows);
    return f;
}

int factored_prepare_weights(const float* w3, int ker_width, int gemm_mode, const FactoredWs& f, hipStream_t s) {
    const long long total = (long long)64 * ker_width * 64;
    hipLaunchKernelGGL(w3_transpose_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w3, 64, ker_width,
                       f.w3t);
    MDNO_TRY(check_launch("w3_transpose_kernel"));
    if (gemm_mode == MDNO_GEMM_SPLIT_BF16) {
        MDNO_TRY(split_planes(f.w3t, 64 * ker_width, 64, f.w3tp, s));
        MDNO_TRY(fill_ints(f.f16_flags, 1, 0, s));
        MDNO_TRY(split_planes_f16(f.w3t, 64 * ker_width, 64, f.w3th, f.w3tus, f.f16_flags, s));
    }
    return MDNO_OK;
}

// k slices of a source's second and later row tiles (see KS_TAIL)
static int tail_slices(int gemm_mode, int rows_per_member) {
    return gemm_mode == MDNO_GEMM_SPLIT_BF16 && (long long)rows_per_member * KS <= 2048 ? KS_TAIL : KS;
}

int degree_order_chunks(const int* row_ptr, int num_rows, int chunk_rows, int* order, hipStream_t s) {
    MDNO_REQUIRE(chunk_rows == kYChunkRows, MDNO_EINVAL, "degree_order_chunks: chunk of %d rows (only %d)", chunk_rows, kYChunkRows);
    hipLaunchKernelGGL(source_order_kernel<kYChunkRows>, dim3((num_rows + kYChunkRows - 1) / kYChunkRows), dim3(kYChunkRows), 0, s,
                       row_ptr, num_rows, order);
    return check_launch("source_order_kernel");
}

int factored_prepare_graph(const int* row_ptr, const int* col, const int* rowid, int num_rows, int rows_per_member,
                           int gemm_mode, long long edge_cap, const FactoredWs& f, int* status, hipStream_t s) {
    MDNO_REQUIRE(edge_cap <= REV_MASK, MDNO_EUNSUPPORTED, "factored conv: edge_cap %lld exceeds %d", edge_cap, REV_MASK);
    TimedSection ts(KID_GRAPH, s);
    hipLaunchKernelGGL(reverse_edges_kernel, dim3((unsigned)((edge_cap + 255) / 256)), dim3(256), 0, s, row_ptr, col,
                       rowid, num_rows, f.rev, status, tail_slices(gemm_mode, rows_per_member), f.f16_flags + 1);
    static_assert(kYChunkRows <= 1024, "source_order_kernel ranks one chunk per workgroup");
    hipLaunchKernelGGL(source_order_kernel<kYChunkRows>, dim3((num_rows + kYChunkRows - 1) / kYChunkRows), dim3(kYChunkRows), 0, s, row_ptr,
                       num_rows, f.order);
    return check_launch("reverse_edges_kernel");
}

int factored_conv(const float* x, const float* h2, const int* row_ptr, int num_rows, int rows_per_member,
                  int max_degree, int ker_width,
                  int gemm_mode, const float* b3, const float* root, const float* bias, int aggr, int relu, float* y,
                  const FactoredWs& f, int* status, hipStream_t s, bool x_prepared, const float* next_b3,
                  int application) {
    const size_t lds1 = sizeof(float) * 2 * 256 * LD;   // 73,728 B
    MDNO_REQUIRE(kYChunkRows % 256 == 0, MDNO_EINVAL, "Y chunk must be a multiple of the GEMM row tile");
    const int ncols = 64 * ker_width;
    const bool split = gemm_mode == MDNO_GEMM_SPLIT_BF16;
    const bool y_f16 = split && application >= 0;
    MDNO_REQUIRE(application < kMaxF16Applications, MDNO_EUNSUPPORTED, "factored conv: more than %d applications",
                 kMaxF16Applications);
    int* flag_x = y_f16 ? f.f16_flags + 1 + application : nullptr;
    if (split) {
        // X -> planes and q = X . B3: done by the previous application's aggregation when x_prepared
        if (!x_prepared) {
            TimedSection ts(KID_FACT_Y, s);
            MDNO_TRY(split_planes_bias64(x, num_rows, b3, f.q, f.xp, s, y_f16 ? f.xh : nullptr, flag_x));
        }
    } else {
        static std::atomic<unsigned long long> lds_raised{0};
        MDNO_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(&gemm_rows_guarded_kernel), (int)lds1, lds_raised));
        TimedSection ts(KID_FACT_Y, s);
        hipLaunchKernelGGL(node_bias_kernel, dim3((num_rows + 3) / 4), dim3(256), 0, s, x, b3, num_rows, f.q);
    }
    const int mtiles = (max_degree + 127) / 128;
    const int kt = tail_slices(gemm_mode, rows_per_member);
    const int slots = KS + (mtiles - 1) * kt;
    for (int r0 = 0; r0 < num_rows; r0 += kYChunkRows) {
        const int cnt = num_rows - r0 < kYChunkRows ? num_rows - r0 : kYChunkRows;
        {   // (1) Y of sources r0 .. r0+cnt-1
            TimedSection ts(KID_FACT_Y, s);
            if (split) {
                // 6 bf16 plane products (fp32-level accuracy, edge_mlp_split.hip): the matrix work drops
                // under the store of Y, which is what bounds this step.  r0 is a multiple of the 128-row
                // plane tiles: the chunk's planes start at tile r0/128 (K = 64: 4 k-steps x 12 KiB each)
                const unsigned char* xp = static_cast<const unsigned char*>(f.xp) + (size_t)(r0 >> 7) * 4 * 3 * 4096;
                if (y_f16) {   // two fp16 planes, 32 KiB per 128-row tile (bf16 images ride along for the fallback)
                    const unsigned char* xh = static_cast<const unsigned char*>(f.xh) + ((size_t)(r0 >> 7) << 15);
                    MDNO_TRY(split_gemm_rows_k64_f16(xh, f.w3th, xp, f.w3tp, f.f16_flags, flag_x, f.w3tus, cnt, ncols, f.y, s));
                } else {
                    MDNO_TRY(split_gemm_rows(xp, f.w3tp, cnt, ncols, 64, f.y, s));
                }
            } else {
                hipLaunchKernelGGL(gemm_rows_guarded_kernel, dim3(ncols / 128, (cnt + 127) / 128), dim3(256), lds1, s,
                                   x + (size_t)r0 * 64, (const float*)f.w3t, f.y, cnt, ncols, 64);
            }
        }
        {   // (2) the per-source GEMMs of the same sources
            TimedSection ts(KID_NNCONV, s);
            if (split) {
                // many rounds of workgroups (large members): source-major
                const long long nwg = (long long)cnt * slots;
                const bool source_major = nwg >= 8192 && nwg < (1ll << 31);
                hipLaunchKernelGGL(gemm_per_source_split_kernel,
                                   source_major ? dim3((unsigned)(cnt * slots), 1) : dim3(cnt, slots), dim3(256), 0, s,
                                   h2, (const float*)f.y, (const float*)f.q, row_ptr, f.m, f.part_stride, ker_width,
                                   kt, slots, r0, status, (const int*)f.order);
            } else {
                hipLaunchKernelGGL(gemm_per_source_kernel, dim3(cnt, mtiles, KS), dim3(256), 0, s, h2,
                                   (const float*)f.y, (const float*)f.q, row_ptr, f.m, f.part_stride, ker_width, r0,
                                   status);
            }
        }
    }
    const bool split_next = gemm_mode == MDNO_GEMM_SPLIT_BF16 && next_b3 != nullptr;
    {
        TimedSection ts(KID_NNCONV_COMBINE, s);
        hipLaunchKernelGGL(aggregate_rev_kernel, dim3(num_rows), dim3(AGG_CHAINS * 16), 0, s, (const float*)f.m,
                           f.part_stride, (const int*)f.rev, row_ptr, x, root, bias, y, num_rows, aggr, relu,
                           split_next ? next_b3 : nullptr, f.q, static_cast<unsigned char*>(f.xp),
                           y_f16 ? static_cast<unsigned char*>(f.xh) : nullptr, y_f16 ? flag_x + 1 : nullptr);
    }
    return check_launch("factored_conv");
}

}  // namespace mdno
