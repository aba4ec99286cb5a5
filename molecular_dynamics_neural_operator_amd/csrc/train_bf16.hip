// Training in bf16 (BASELINE.json configs[3] names bf16; reference: graph_kernel.py:445-474, 541-547).
//
// The kernel-integral block's large tensors — the edge-MLP activations h1, h2 [E,k], the edge weights
// W_e [E, Cin*Cout] and their gradient dW_e — are STORED in bf16 (row-major), and every GEMM of the
// block is a single bf16 x bf16 MFMA product with fp32 accumulation; the parameters stay fp32 (master
// weights, cast per call), as do the node features, the conv outputs and every reduction.  That halves
// the two E x 16 KiB tensors of the fp32 path (train.hip), takes the weight-gradient products A^T.B
// off the fp32 MFMA (1/16 of the bf16 rate) and halves what the conv kernels stream.
//
//   cast_bf16            fp32 -> bf16 (RNE), row-major
//   gemm_nt_bf16         C = act(A . W^T + b)      A bf16 [rows,K], W bf16 [N,K]      -> bf16 or fp32
//   gemm_tn_bf16         C = A^T . B over rows     A bf16 [rows,n1], B bf16 [rows,n2] -> fp32 [n1,n2]
//   nnconv64_bf16w_*     the conv forward / input-gradient kernels of nnconv.hip / train.hip reading bf16 W_e
//   nnconv_bwd_we_bf16   dW_e written as bf16
//   relu_bwd_bf16, colsum_bf16   the elementwise / reduction ops on bf16 operands
// Reductions keep fixed-order partial sums (no float atomics): gradients are bitwise reproducible.
#include "kernels.h"
#include "reduce.h"

namespace mdno {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }
__device__ __forceinline__ float4 ld4_bf16(const __bf16* p) {      // 4 consecutive bf16 -> float4 (8-B load)
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                       __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u));
}
__device__ __forceinline__ uint2 pack4_bf16(float a, float b, float c, float d) {
    const bf16x4 v = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
    return __builtin_bit_cast(uint2, v);
}

// ---------------------------------------------------------------- cast
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ in, long long n4,
                                                        uint2* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 v = reinterpret_cast<const float4*>(in)[i];
    out[i] = pack4_bf16(v.x, v.y, v.z, v.w);
}

// ---------------------------------------------------------------- first layer: K <= 8, bf16 out
// c[r][n] = bf16(act(sum_k a[r][k] w[n][k] + b[n])), a fp32 [rows,K] (the edge attributes), w fp32 [n,K].
// A store with a few FMAs in front: thread = (row, 8 consecutive columns) -> one 16-B store; the sums are
// the fmaf chains of linear_generic_kernel (k ascending, bias added last), so the result is the one the
// fp32 kernel followed by mdno_cast_bf16 gave — without the fp32 round trip (200 + 12 us -> 2x us at cfg4).
// Workgroup = SK_ROWS rows x all columns, in passes of 128 column groups (1,024 columns); thread = (group,
// row parity).  The weights of the pass sit in LDS as [k][j][group] (group = 8 consecutive columns, j =
// column inside it), so the lanes of a wave — consecutive groups — read consecutive words.  (With every
// thread fetching its 48 weights from global memory, 192 B apart from its neighbour's, the kernel took
// 531 us; the fp32 kernel 200.)
constexpr int SK_ROWS = 32, SK_GROUPS = 128;
template <bool RELU>
__global__ __launch_bounds__(256) void linear_smallk_bf16_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                                 const float* __restrict__ bias, long long rows, int N,
                                                                 int K, uint4* __restrict__ C) {
    __shared__ float wt[8 * 8 * SK_GROUPS];       // [k][j][group], 32 KiB
    __shared__ float bs[8 * SK_GROUPS];           // [j][group]
    __shared__ float as[SK_ROWS * 8];
    const int tid = threadIdx.x, groups = N >> 3;
    const int grp = tid & (SK_GROUPS - 1), par = tid / SK_GROUPS;      // 128 groups x 2 row parities
    const long long r0 = (long long)blockIdx.x * SK_ROWS;
    for (int t = tid; t < SK_ROWS * 8; t += 256) {
        const long long r = r0 + (t >> 3);
        const int k = t & 7;
        as[t] = (r < rows && k < K) ? A[r * K + k] : 0.f;
    }
    for (int g0 = 0; g0 < groups; g0 += SK_GROUPS) {
        const int ng = groups - g0 < SK_GROUPS ? groups - g0 : SK_GROUPS;
        __syncthreads();
        for (int t = tid; t < ng * 8 * K; t += 256) {       // coalesced over (column, k)
            const int col = t / K, k = t - col * K;
            wt[(k * 8 + (col & 7)) * SK_GROUPS + (col >> 3)] = W[(size_t)g0 * 8 * K + t];
        }
        for (int t = tid; t < ng * 8; t += 256) bs[(t & 7) * SK_GROUPS + (t >> 3)] = bias ? bias[g0 * 8 + t] : 0.f;
        __syncthreads();
        if (grp < ng) {
            float wr[8][8], br[8];      // the thread's 8 columns x K weights stay in registers across the rows
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                br[j] = bs[j * SK_GROUPS + grp];
#pragma unroll
                for (int k = 0; k < 8; ++k) wr[k][j] = k < K ? wt[(k * 8 + j) * SK_GROUPS + grp] : 0.f;
            }
            for (int rr = par; rr < SK_ROWS; rr += 2) {
                const long long r = r0 + rr;
                if (r >= rows) break;
                float av[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) av[k] = as[rr * 8 + k];
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float s = 0.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (k < K) s = fmaf(av[k], wr[k][j], s);
                    s += br[j];
                    v[j] = RELU ? relu_f(s) : s;
                }
                const uint2 lo = pack4_bf16(v[0], v[1], v[2], v[3]), hi = pack4_bf16(v[4], v[5], v[6], v[7]);
                C[(size_t)r * groups + g0 + grp] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        }
    }
}

// ---------------------------------------------------------------- C = act(A . W^T + b)
// 128 x 128 block tile, 4 waves of 64 x 64 (2 x 2 v_mfma_f32_32x32x16_bf16), 32 k per stage, register
// staging into double-buffered LDS.  Rows of 32 bf16 are padded to 80 B: 16-B aligned fragment reads,
// five-quad row stride.
constexpr int NK = 32, NLD = 40;     // k per stage; LDS row length in bf16 (80 B)

__device__ __forceinline__ void mma_bf16_tile(f32x16 (&acc)[2][2], const __bf16* as, const __bf16* bs, int l31, int h) {
#pragma unroll
    for (int kk = 0; kk < NK / 16; ++kk) {
        bf16x8 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a[i] = *reinterpret_cast<const bf16x8*>(as + (i * 32 + l31) * NLD + kk * 16 + 8 * h);
            b[i] = *reinterpret_cast<const bf16x8*>(bs + (i * 32 + l31) * NLD + kk * 16 + 8 * h);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}

template <bool RELU, bool OUT_BF16>
__global__ __launch_bounds__(256) void gemm_nt_bf16_kernel(const __bf16* __restrict__ A, const __bf16* __restrict__ W,
                                                           const float* __restrict__ bias, void* __restrict__ Cv,
                                                           long long rows, int N, int K) {
    __shared__ __attribute__((aligned(16))) __bf16 As[2][128 * NLD];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2][128 * NLD];
    const long long bm = (long long)blockIdx.y * 128;
    const int bn = blockIdx.x * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    const int srow = tid >> 2, sk = (tid & 3) * 8;          // rows srow, srow + 64; 8 k each
    auto arow = [&](int r) { const long long rr = bm + r; return (size_t)(rr < rows ? rr : rows - 1); };
    const __bf16* A0 = A + arow(srow) * K + sk;
    const __bf16* A1 = A + arow(srow + 64) * K + sk;
    const __bf16* B0 = W + (size_t)(bn + srow) * K + sk;
    const __bf16* B1 = W + (size_t)(bn + srow + 64) * K + sk;
    uint4 ra0, ra1, rb0, rb1;
#define MDNO_LD(KO)                                              \
    ra0 = *reinterpret_cast<const uint4*>(A0 + (KO));            \
    ra1 = *reinterpret_cast<const uint4*>(A1 + (KO));            \
    rb0 = *reinterpret_cast<const uint4*>(B0 + (KO));            \
    rb1 = *reinterpret_cast<const uint4*>(B1 + (KO));
#define MDNO_ST(BUF)                                                                  \
    *reinterpret_cast<uint4*>(&As[BUF][srow * NLD + sk]) = ra0;                       \
    *reinterpret_cast<uint4*>(&As[BUF][(srow + 64) * NLD + sk]) = ra1;                \
    *reinterpret_cast<uint4*>(&Bs[BUF][srow * NLD + sk]) = rb0;                       \
    *reinterpret_cast<uint4*>(&Bs[BUF][(srow + 64) * NLD + sk]) = rb1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float bv0 = 0.f, bv1 = 0.f;        // before the K loop and pinned (see edge_mlp_split.hip, epilogue stores)
    if (bias) {
        bv0 = bias[bn + wn * 64 + l31];
        bv1 = bias[bn + wn * 64 + 32 + l31];
    }
    asm volatile("" : "+v"(bv0), "+v"(bv1));
    const int nk = K / NK;
    MDNO_LD(0)
    MDNO_ST(0)
    __syncthreads();
    for (int kt = 0; kt < nk - 1; ++kt) {
        MDNO_LD((size_t)(kt + 1) * NK)
        mma_bf16_tile(acc, &As[kt & 1][wm * 64 * NLD], &Bs[kt & 1][wn * 64 * NLD], l31, h);
        MDNO_ST((kt & 1) ^ 1)
        __syncthreads();
    }
    mma_bf16_tile(acc, &As[(nk - 1) & 1][wm * 64 * NLD], &Bs[(nk - 1) & 1][wn * 64 * NLD], l31, h);
#undef MDNO_LD
#undef MDNO_ST
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = bn + wn * 64 + j * 32 + l31;
        const float bv = j ? bv1 : bv0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const long long m = bm + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < rows) {
                    float v = acc[i][j][e] + bv;
                    if (RELU) v = relu_f(v);
                    if (OUT_BF16) static_cast<__bf16*>(Cv)[(size_t)m * N + n] = (__bf16)v;
                    else static_cast<float*>(Cv)[(size_t)m * N + n] = v;
                }
            }
    }
}

// ---------------------------------------------------------------- C = A^T . B over rows (one K-slice)
// A [rows, n1], B [rows, n2] row-major: the contraction index (the edge) is the SLOW index of both
// operands, the opposite of what an MFMA fragment wants (8 consecutive k per lane).  The staged 16-B
// pieces go to LDS as they come — image [k 32][128 columns], 256-B rows, one ds_write_b128 each — and the
// fragments are read with gfx950's transposing LDS read: ds_read_b64_tr_b16 hands each lane of a 16-lane
// group one COLUMN of a 4-row x 16-column block, i.e. 4 consecutive k of its own m; two of them make the
// 8-k operand of v_mfma_f32_32x32x16_bf16.  The 16-B chunks of a row are XOR-swizzled with
// ((row&3)<<2 | (row>>2)&3), which keeps both the row writes and the transposed reads off each other's
// banks (cdna_hip_programming.md T10, image (b)).  (The first version transposed on the way IN, with
// eight 2-byte LDS writes per staged piece: 467 us per product at cfg4's batch against 253 now.)
// Rows are cut into `slices` equal runs (blockIdx.z), partial products go to part[slice][n1][n2] and are
// added in slice order by reduce_slices_bf16path_kernel.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ int tn_off(int row, int ch) {      // byte offset of 16-B chunk ch of row `row`
    return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

__device__ __forceinline__ bf16x8 tn_frag(const unsigned char* base, int off0, int off1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off1));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

constexpr int TN_STAGE = 32 * 256;      // one operand, one stage: 32 k x 128 columns of bf16

__global__ __launch_bounds__(256) void gemm_tn_bf16_kernel(const __bf16* __restrict__ A, const __bf16* __restrict__ B,
                                                           float* __restrict__ part, long long rows, int n1, int n2,
                                                           long long slice_rows) {
    __shared__ __attribute__((aligned(16))) unsigned char As[2][TN_STAGE];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2][TN_STAGE];
    const int bm = blockIdx.y * 128, bn = blockIdx.x * 128;
    const long long r0 = (long long)blockIdx.z * slice_rows;
    long long r1 = r0 + slice_rows;
    if (r1 > rows) r1 = rows;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    // staging: pieces tid and tid + 256 of the stage's 512: row = piece >> 4, 16-B chunk = piece & 15
    const int srow = tid >> 4, sch = tid & 15;
    const int st0 = tn_off(srow, sch), st1 = tn_off(srow + 16, sch);
    uint4 ra0, ra1, rb0, rb1;
    const uint4 zero = make_uint4(0, 0, 0, 0);
#define MDNO_LD(E0)                                                                                         \
    {                                                                                                       \
        const long long e0_ = (E0) + srow, e1_ = (E0) + srow + 16;                                          \
        ra0 = e0_ < r1 ? *reinterpret_cast<const uint4*>(A + (size_t)e0_ * n1 + bm + sch * 8) : zero;       \
        ra1 = e1_ < r1 ? *reinterpret_cast<const uint4*>(A + (size_t)e1_ * n1 + bm + sch * 8) : zero;       \
        rb0 = e0_ < r1 ? *reinterpret_cast<const uint4*>(B + (size_t)e0_ * n2 + bn + sch * 8) : zero;       \
        rb1 = e1_ < r1 ? *reinterpret_cast<const uint4*>(B + (size_t)e1_ * n2 + bn + sch * 8) : zero;       \
    }
#define MDNO_ST(BUF)                                              \
    *reinterpret_cast<uint4*>(As[BUF] + st0) = ra0;               \
    *reinterpret_cast<uint4*>(As[BUF] + st1) = ra1;               \
    *reinterpret_cast<uint4*>(Bs[BUF] + st0) = rb0;               \
    *reinterpret_cast<uint4*>(Bs[BUF] + st1) = rb1;
    // transposed reads: 16-lane group g = lane>>4 takes the 4-row x 16-column block at rows
    // kk*16 + 8*(g>>1) + 4*r (r = 0, 1: the two reads of a fragment), columns m0 = tile + 16*(g&1); lane
    // 4q+p of the group supplies row q, columns 4p..4p+3 of the block
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    int a_off[2][2][2], b_off[2][2][2];      // [m or n tile i][k-step kk][read r]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = kk * 16 + 8 * (g >> 1) + 4 * r + q;
                a_off[i][kk][r] = tn_off(row, (wm * 64 + i * 32 + 16 * (g & 1)) / 8 + (p >> 1)) + 8 * (p & 1);
                b_off[i][kk][r] = tn_off(row, (wn * 64 + i * 32 + 16 * (g & 1)) / 8 + (p >> 1)) + 8 * (p & 1);
            }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    auto mma = [&](int buf) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = tn_frag(As[buf], a_off[i][kk][0], a_off[i][kk][1]);
                b[i] = tn_frag(Bs[buf], b_off[i][kk][0], b_off[i][kk][1]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    const long long nst = (r1 - r0 + 31) / 32;      // (uniform per workgroup: EXEC is all ones at the reads)
    if (nst > 0) {
        MDNO_LD(r0)
        MDNO_ST(0)
        __syncthreads();
        for (long long st = 0; st < nst - 1; ++st) {
            MDNO_LD(r0 + (st + 1) * 32)
            mma((int)(st & 1));
            MDNO_ST((int)((st & 1) ^ 1))
            __syncthreads();
        }
        mma((int)((nst - 1) & 1));
    }
#undef MDNO_LD
#undef MDNO_ST
    float* P = part + (size_t)blockIdx.z * n1 * n2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = bn + wn * 64 + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = bm + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                P[(size_t)m * n2 + n] = acc[i][j][e];
            }
    }
}

// ---------------------------------------------------------------- conv forward, bf16 W_e
// nnconv64_row_kernel (nnconv.hip) with 8-B loads of four bf16: same lane map (lane (g,q) owns input rows
// 16g..16g+15 x output columns 4q..4q+3), same 16 summation chains, fp32 accumulation.
__device__ __forceinline__ void fma4(float4& a, float s, const float4& w) {
    a.x = fmaf(s, w.x, a.x); a.y = fmaf(s, w.y, a.y); a.z = fmaf(s, w.z, a.z); a.w = fmaf(s, w.w, a.w);
}
__device__ __forceinline__ float4 reduce_over_g(float4 a) {
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
        a.x += __shfl_xor(a.x, o); a.y += __shfl_xor(a.y, o); a.z += __shfl_xor(a.z, o); a.w += __shfl_xor(a.w, o);
    }
    return a;
}
template <class WT>
__device__ __forceinline__ float4 ldw4(const WT* p);
template <>
__device__ __forceinline__ float4 ldw4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
// (W_e is read once per conv application and is far larger than the caches: streamed past them)
template <>
__device__ __forceinline__ float4 ldw4<__bf16>(const __bf16* p) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 u = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p));
    return make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                       __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u));
}

// (W first: its address does not wait for src[p], which the x row's does; all twenty loads in flight before the
// first FMA waits — a batch row has ~12 edges, one per wave, so a workgroup's life is its chain of round trips)
template <class WT>
__device__ __forceinline__ void edge_acc64(float4& acc, const float* __restrict__ xrow, const WT* __restrict__ wmat,
                                           int g, int q) {
    const WT* wp = wmat + (16 * g) * 64 + 4 * q;
    float4 w[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) w[r] = ldw4<WT>(wp + r * 64);
    const float* xp = xrow + 16 * g;
    const float4 x0 = *reinterpret_cast<const float4*>(xp), x1 = *reinterpret_cast<const float4*>(xp + 4);
    const float4 x2 = *reinterpret_cast<const float4*>(xp + 8), x3 = *reinterpret_cast<const float4*>(xp + 12);
    __builtin_amdgcn_sched_barrier(0);
    fma4(acc, x0.x, w[0]);  fma4(acc, x0.y, w[1]);  fma4(acc, x0.z, w[2]);  fma4(acc, x0.w, w[3]);
    fma4(acc, x1.x, w[4]);  fma4(acc, x1.y, w[5]);  fma4(acc, x1.z, w[6]);  fma4(acc, x1.w, w[7]);
    fma4(acc, x2.x, w[8]);  fma4(acc, x2.y, w[9]);  fma4(acc, x2.z, w[10]); fma4(acc, x2.w, w[11]);
    fma4(acc, x3.x, w[12]); fma4(acc, x3.y, w[13]); fma4(acc, x3.z, w[14]); fma4(acc, x3.w, w[15]);
}

// WAVES = 16: a wave per summation chain; WAVES = 4: a wave owns chains w, w+4, w+8, w+12, one after the other (same
// chains, same order of additions: same bits) — four times as many workgroups resident per CU
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void nnconv64_bf16w_kernel(const float* __restrict__ x, const int* __restrict__ row_ptr,
                                                              const int* __restrict__ src,
                                                              const __bf16* __restrict__ w_e,
                                                              const float* __restrict__ root,
                                                              const float* __restrict__ bias, float* __restrict__ y,
                                                              int num_rows, int aggr, int relu) {
    __shared__ float red[16][64];
    __shared__ float rootred[64];
    const int row = blockIdx.x;
    if (row >= num_rows) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, q = lane & 15;
    const int beg = row_ptr[row], end = row_ptr[row + 1], deg = end - beg;
#pragma unroll
    for (int u = 0; u < 16 / WAVES; ++u) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int p = beg + wave + u * WAVES; p < end; p += 16)
            edge_acc64<__bf16>(acc, x + (size_t)src[p] * 64, w_e + (size_t)p * 4096, g, q);
        acc = reduce_over_g(acc);
        if (lane < 16) *reinterpret_cast<float4*>(&red[wave + u * WAVES][4 * lane]) = acc;
    }
    const bool root_wave = root != nullptr && wave == (deg % WAVES);
    float4 racc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (root_wave) edge_acc64<float>(racc, x + (size_t)row * 64, root, g, q);
    racc = reduce_over_g(racc);
    if (root_wave && lane < 16) *reinterpret_cast<float4*>(&rootred[4 * lane]) = racc;
    __syncthreads();
    if (tid < 64) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) s += red[c][tid];
        if (aggr == MDNO_AGGR_MEAN) s = s / (float)(deg > 1 ? deg : 1);
        if (root != nullptr) s += rootred[tid];
        if (bias != nullptr) s += bias[tid];
        if (relu) s = relu_f(s);
        y[(size_t)row * 64 + tid] = s;
    }
}

// ---------------------------------------------------------------- conv backward wrt the input, bf16 W_e
// nnconv_bwd_x_kernel (train.hip): g_prev[r] = gz[r] . root^T + sum_{e: src e = r} W_e . gs[dst e]
template <class WT>
__device__ __forceinline__ void wg_acc(float (&acc)[16], const WT* __restrict__ wmat, const float* __restrict__ gvec,
                                       int g, int q) {
    const float4 gq = *reinterpret_cast<const float4*>(gvec + 4 * q);
    const WT* wp = wmat + (16 * g) * 64 + 4 * q;
    float4 w[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) w[r] = ldw4<WT>(wp + r * 64);
#pragma unroll
    for (int r = 0; r < 16; ++r)
        acc[r] = fmaf(w[r].x, gq.x, fmaf(w[r].y, gq.y, fmaf(w[r].z, gq.z, fmaf(w[r].w, gq.w, acc[r]))));
}

// y_below != NULL: the gradient leaves through the ReLU of the application below (whose output is y_below) —
// gz_below = g_prev * (y_below > 0), gs_below = gz_below * inv_deg[row] are written instead of g_prev: what
// mdno_relu_bwd2 would make of g_prev in a launch of its own, same arithmetic
__global__ __launch_bounds__(256) void nnconv_bwd_x_bf16w_kernel(const float* __restrict__ gz, const float* __restrict__ gs,
                                                                 const int* __restrict__ row_ptr_s,
                                                                 const int* __restrict__ eid_s,
                                                                 const int* __restrict__ dst_s,
                                                                 const __bf16* __restrict__ w_e,
                                                                 const float* __restrict__ root,
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
        wg_acc<__bf16>(acc, w_e + (size_t)eid_s[p] * 4096, gs + (size_t)dst_s[p] * 64, g, q);
    if (root != nullptr && wave == ((end - beg) & 3)) wg_acc<float>(acc, root, gz + (size_t)row * 64, g, q);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = acc[r];
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
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

// ---------------------------------------------------------------- column sums + A^T.B for a few fp32 columns
// colsum[n] = sum_r a[r][n] and atb[n][j] = sum_r a[r][n] * b[r][j] (j < KB <= 8) in ONE pass over a bf16 [rows,n]:
// the bias and weight gradient of the edge-MLP's FIRST layer (b = the fp32 edge attributes, 6 columns), which were a
// column-sum pass plus a 128-column zero-padded bf16 A^T.B over the same 89 MB.  Same slicing as colsum_bf16_kernel.
template <int KB>
__global__ __launch_bounds__(256) void colsum_atb_bf16_kernel(const __bf16* __restrict__ a, const float* __restrict__ b,
                                                              float* __restrict__ part, long long rows, int n,
                                                              long long slice_rows) {
    __shared__ float red[4][64][8];
    const int cg = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.x * 512 + cg * 8;
    const long long r0 = (long long)blockIdx.y * slice_rows;
    long long r1 = r0 + slice_rows;
    if (r1 > rows) r1 = rows;
    float s[KB + 1][8];
#pragma unroll
    for (int m = 0; m <= KB; ++m)
#pragma unroll
        for (int j = 0; j < 8; ++j) s[m][j] = 0.f;
    if (col < n) {
        auto add = [&](const uint4& u, const float (&bv)[KB]) {
            const unsigned w[4] = {u.x, u.y, u.z, u.w};
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[2 * j] = __builtin_bit_cast(float, w[j] << 16);
                v[2 * j + 1] = __builtin_bit_cast(float, w[j] & 0xffff0000u);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s[0][j] += v[j];
#pragma unroll
                for (int m = 0; m < KB; ++m) s[m + 1][j] = fmaf(v[j], bv[m], s[m + 1][j]);
            }
        };
        for (long long r = r0 + rl; r < r1; r += 8) {       // two of this wave's rows in flight
            const bool two = r + 4 < r1;
            const uint4 u0 = *reinterpret_cast<const uint4*>(a + (size_t)r * n + col);
            const uint4 u1 = two ? *reinterpret_cast<const uint4*>(a + (size_t)(r + 4) * n + col) : make_uint4(0, 0, 0, 0);
            float b0[KB], b1[KB];
#pragma unroll
            for (int m = 0; m < KB; ++m) {
                b0[m] = b[(size_t)r * KB + m];
                b1[m] = two ? b[(size_t)(r + 4) * KB + m] : 0.f;
            }
            add(u0, b0);
            if (two) add(u1, b1);
        }
    }
    // four row phases combined in phase order, one quantity at a time (8 KiB of LDS)
    for (int m = 0; m <= KB; ++m) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) red[rl][cg][j] = s[m][j];
        __syncthreads();
        if (rl == 0 && col < n) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                part[((size_t)blockIdx.y * (KB + 1) + m) * n + col + j] =
                    (red[0][cg][j] + red[1][cg][j]) + (red[2][cg][j] + red[3][cg][j]);
        }
    }
}

// out[n][KB] <- reduced [KB][n] (columns of atb), colsum <- row 0
template <int KB>
__global__ __launch_bounds__(256) void colsum_atb_finish_kernel(const float* __restrict__ red, int n, float* __restrict__ colsum,
                                                                float* __restrict__ atb) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    colsum[c] = red[c];
#pragma unroll
    for (int m = 0; m < KB; ++m) atb[(size_t)c * KB + m] = red[(size_t)(m + 1) * n + c];
}

// ---------------------------------------------------------------- d W_e as bf16
// nnconv_bwd_we_kernel (train.hip) with the result rounded once, at the end, and written as 8-B stores
__global__ __launch_bounds__(256) void nnconv_bwd_we_bf16_kernel(const float* __restrict__ x, const float* __restrict__ gs,
                                                                 const int* __restrict__ src, const int* __restrict__ dst,
                                                                 long long E, int L, long long layer_stride,
                                                                 __bf16* __restrict__ dwe) {
    const int lane = threadIdx.x & 63, g = lane >> 4, q = lane & 15;
    const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= E) return;
    const float* xs = x + (size_t)src[p] * 64 + 16 * g;
    const float* gq = gs + (size_t)dst[p] * 64 + 4 * q;
    float4 acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = 0; l < L; ++l) {
        const float4 gv = *reinterpret_cast<const float4*>(gq + (size_t)l * layer_stride);
        const float* xl = xs + (size_t)l * layer_stride;
        const float4 x0 = *reinterpret_cast<const float4*>(xl), x1 = *reinterpret_cast<const float4*>(xl + 4);
        const float4 x2 = *reinterpret_cast<const float4*>(xl + 8), x3 = *reinterpret_cast<const float4*>(xl + 12);
        const float xv[16] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w,
                              x2.x, x2.y, x2.z, x2.w, x3.x, x3.y, x3.z, x3.w};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[r].x = fmaf(xv[r], gv.x, acc[r].x); acc[r].y = fmaf(xv[r], gv.y, acc[r].y);
            acc[r].z = fmaf(xv[r], gv.z, acc[r].z); acc[r].w = fmaf(xv[r], gv.w, acc[r].w);
        }
    }
    __bf16* out = dwe + (size_t)p * 4096 + (16 * g) * 64 + 4 * q;
#pragma unroll
    for (int r = 0; r < 16; ++r) *reinterpret_cast<uint2*>(out + r * 64) = pack4_bf16(acc[r].x, acc[r].y, acc[r].z, acc[r].w);
}

// ---------------------------------------------------------------- d W_e as bf16 + its column sums, on the matrix pipe
// dW_e[p] = sum_l gs_l[dst p] (x) x_l[src p] is a [64 x L] . [L x 64] product per edge: with L <= 16 ONE k-step of
// v_mfma_f32_32x32x16_bf16 per 32 x 32 quadrant.  The kernel above spends 768 FMAs per lane and edge on it (115 us at
// cfg4, twice what writing the 358 MB takes); here both fp32 operands are split exactly into three bf16 planes in
// registers and the six leading plane products accumulated in fp32 (fp32 accuracy, as everywhere in this library):
// 24 MFMAs per edge.  One wave per edge at a time, edges p = wave, wave + W, ..: the next edge's 32 operand words are
// fetched before this edge's MFMAs.
//   A = G (rows o): lane (l31, h) holds gs_l[dst][32 ob + l31], l = 8 h .. 8 h + 7;  B = X (columns i): x_l[src][32 ib + l31]
//   acc[ob][ib][e] = dW_e[i = 32 ib + l31][o = 32 ob + (e & 3) + 8 (e >> 2) + 4 h]: four consecutive o -> one 8-B LDS write
// and the rounded tile goes out through LDS row by row: 16 B per lane, 1 KiB contiguous per store instruction.
// The column sums (the last layer's bias gradient: sum over edges of the ROUNDED dW_e, what mdno_colsum_bf16 computes
// from the stored tensor in a second pass over its 358 MB) are taken on the way: every lane owns 64 fixed (i, o)
// positions of the tile, adds each edge's rounded values in edge order, the four waves of a workgroup are added in wave
// order through LDS and the workgroups by reduce_slices: fixed association, no atomics.
// BF16 = false: the fp32 training path's dW_e (train.hip's nnconv_bwd_we_kernel: 180 us + a 96 us column-sum pass over
// 716 MB at cfg4) through the same loop, the tile staged as fp32 and its column sums taken from the stored values.
constexpr int WE_WGS = 512;                              // workgroups of the launch (whatever E: the association of the sums is fixed)
template <bool BF16> struct WeTile {
    static constexpr int ROW = BF16 ? 136 : 272;         // LDS bytes per row of 64 outputs (+8 / +16: the 32 rows a write touches spread over the banks)
    static constexpr int BYTES = 64 * ROW;               // 8,704 / 17,408 B per wave
};

template <bool BF16>
__global__ __launch_bounds__(256, 2) void nnconv_bwd_we_mfma_kernel(const float* __restrict__ x, const float* __restrict__ gs,
                                                                   const int* __restrict__ src, const int* __restrict__ dst,
                                                                   long long E, int L, long long layer_stride,
                                                                   void* __restrict__ dwe_, float* __restrict__ part) {
    constexpr int ROW = WeTile<BF16>::ROW, TILE = WeTile<BF16>::BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * TILE > 4096 * 4 ? 4 * TILE : 4096 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    // read-back role: 16 B per lane; bf16: 8 rows x 128 B per pass (8 passes), fp32: 4 rows x 256 B (16 passes)
    constexpr int PASSES = BF16 ? 8 : 16, RPP = 64 / PASSES, PER = BF16 ? 8 : 4;
    const int rr = BF16 ? lane >> 3 : lane >> 4, rc = BF16 ? lane & 7 : lane & 15;
    unsigned char* tile = lds + wave * TILE;
    const long long W = (long long)gridDim.x * 4;
    float cs[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) cs[j] = 0.f;
    float ga[2][8], xb[2][8];                            // this edge's operand words; next edge's while the MFMAs run
    // (every load unconditional — a layer past L re-reads layer 0 and is zeroed by a select — so that the loop body is
    // straight-line code: with `on ? load : 0` the compiler built a branch around each of the 32 loads)
    auto fetch = [&](long long p, float (&g_)[2][8], float (&x_)[2][8]) {
        const float* gq = gs + (size_t)dst[p] * 64 + l31;
        const float* xq = x + (size_t)src[p] * 64 + l31;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int l = 8 * h + j;
            const size_t off = (size_t)(l < L ? l : 0) * layer_stride;
            g_[0][j] = gq[off];
            g_[1][j] = gq[off + 32];
            x_[0][j] = xq[off];
            x_[1][j] = xq[off + 32];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (8 * h + j >= L) { g_[0][j] = 0.f; g_[1][j] = 0.f; x_[0][j] = 0.f; x_[1][j] = 0.f; }
    };
    auto split3 = [](const float (&v)[8], bf16x8 (&pl)[3]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const __bf16 hi = (__bf16)v[j];
            const float r1 = v[j] - (float)hi;
            const __bf16 mid = (__bf16)r1;
            const __bf16 lo = (__bf16)(r1 - (float)mid);
            pl[0][j] = hi; pl[1][j] = mid; pl[2][j] = lo;
        }
    };
    long long p = (long long)blockIdx.x * 4 + wave;
    if (p < E) fetch(p, ga, xb);
    for (; p < E; p += W) {
        bf16x8 a[2][3], b[2][3];
        split3(ga[0], a[0]); split3(ga[1], a[1]);
        split3(xb[0], b[0]); split3(xb[1], b[1]);
        if (p + W < E) fetch(p + W, ga, xb);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc[2][2];
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                acc[ob][ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ob][1], b[ib][1], zero, 0, 0, 0);      // (C = inline 0)
                acc[ob][ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ob][2], b[ib][0], acc[ob][ib], 0, 0, 0);
                acc[ob][ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ob][0], b[ib][2], acc[ob][ib], 0, 0, 0);
                acc[ob][ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ob][1], b[ib][0], acc[ob][ib], 0, 0, 0);
                acc[ob][ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ob][0], b[ib][1], acc[ob][ib], 0, 0, 0);
                acc[ob][ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ob][0], b[ib][0], acc[ob][ib], 0, 0, 0);
            }
        // tile -> LDS [i][o] (a wave's own tile: no workgroup barrier); four consecutive o per write
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x16& c = acc[ob][ib];
                    unsigned char* wp = tile + (32 * ib + l31) * ROW + (32 * ob + 8 * g + 4 * h) * (BF16 ? 2 : 4);
                    if (BF16) *reinterpret_cast<uint2*>(wp) = pack4_bf16(c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]);
                    else *reinterpret_cast<float4*>(wp) = make_float4(c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]);
                }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const unsigned char* rp = tile + (RPP * ps + rr) * ROW + rc * 16;
            const size_t at = (size_t)p * 4096 + (RPP * ps + rr) * 64 + PER * rc;
            if (BF16) {
                const uint2 u0 = *reinterpret_cast<const uint2*>(rp), u1 = *reinterpret_cast<const uint2*>(rp + 8);
                *reinterpret_cast<uint4*>(static_cast<__bf16*>(dwe_) + at) = make_uint4(u0.x, u0.y, u1.x, u1.y);
                const unsigned w4[4] = {u0.x, u0.y, u1.x, u1.y};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    cs[8 * ps + 2 * j] += __builtin_bit_cast(float, w4[j] << 16);
                    cs[8 * ps + 2 * j + 1] += __builtin_bit_cast(float, w4[j] & 0xffff0000u);
                }
            } else {
                const float4 v = *reinterpret_cast<const float4*>(rp);
                *reinterpret_cast<float4*>(static_cast<float*>(dwe_) + at) = v;
                cs[4 * ps] += v.x; cs[4 * ps + 1] += v.y; cs[4 * ps + 2] += v.z; cs[4 * ps + 3] += v.w;
            }
        }
        __builtin_amdgcn_wave_barrier();      // the tile is rewritten by the next edge
    }
    // column sums: the four waves in wave order through LDS -> part[workgroup][4096]
    float* red = reinterpret_cast<float*>(lds);
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps)
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    const int idx = (RPP * ps + rr) * 64 + PER * rc + j;
                    red[idx] = w == 0 ? cs[PER * ps + j] : red[idx] + cs[PER * ps + j];
                }
        }
        __syncthreads();
    }
    float* po = part + (size_t)blockIdx.x * 4096;
    for (int i = threadIdx.x; i < 1024; i += 256)
        reinterpret_cast<float4*>(po)[i] = reinterpret_cast<const float4*>(red)[i];
}

// ---------------------------------------------------------------- elementwise / reductions on bf16
// out = g * (y > 0): g fp32, y bf16 (the saved activation), out bf16 or fp32
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void relu_bwd_bf16_kernel(const float* __restrict__ g, const __bf16* __restrict__ y,
                                                            long long rows, int n, void* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;       // over groups of 4 elements
    if (i >= rows * (n / 4)) return;
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    const float4 yv = ld4_bf16(y + 4 * i);
    const float4 r = make_float4(yv.x > 0.f ? gv.x : 0.f, yv.y > 0.f ? gv.y : 0.f, yv.z > 0.f ? gv.z : 0.f,
                                 yv.w > 0.f ? gv.w : 0.f);
    if (OUT_BF16) static_cast<uint2*>(out)[i] = pack4_bf16(r.x, r.y, r.z, r.w);
    else static_cast<float4*>(out)[i] = r;
}

// column sums of a bf16 matrix: block = (512 columns, row slice); thread = 8 consecutive columns (one 16-B load per
// row: a wave reads a contiguous KiB) x one of four row phases -> part[slice][n]; phases added in order, slices
// reduced in slice order.  (One 2-byte load per thread and row, 128 B per wave-load, took 94 us for the
// 358 MB of dW_e; this shape streams.)
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const __bf16* __restrict__ a, float* __restrict__ part,
                                                          long long rows, int n, long long slice_rows) {
    __shared__ float red[4][64][8];
    const int cg = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.x * 512 + cg * 8;
    const long long r0 = (long long)blockIdx.y * slice_rows;
    long long r1 = r0 + slice_rows;
    if (r1 > rows) r1 = rows;
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    if (col < n) {
        auto add = [&](const uint4& u) {
            const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[2 * j] += __builtin_bit_cast(float, w[j] << 16);
                s[2 * j + 1] += __builtin_bit_cast(float, w[j] & 0xffff0000u);
            }
        };
        long long r = r0 + rl;
        for (; r + 12 < r1; r += 16) {       // four of this thread's rows in flight (same rows, same order)
            uint4 u[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) u[v] = *reinterpret_cast<const uint4*>(a + (size_t)(r + 4 * v) * n + col);
#pragma unroll
            for (int v = 0; v < 4; ++v) add(u[v]);
        }
        for (; r < r1; r += 4) add(*reinterpret_cast<const uint4*>(a + (size_t)r * n + col));
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[rl][cg][j] = s[j];
    __syncthreads();
    if (rl == 0 && col < n) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            part[(size_t)blockIdx.y * n + col + j] = (red[0][cg][j] + red[1][cg][j]) + (red[2][cg][j] + red[3][cg][j]);
    }
}

constexpr int kTnSlices = 16, kColSlicesB = 128;

}  // namespace
}  // namespace mdno

using namespace mdno;

extern "C" int mdno_cast_bf16(const float* in, int64_t count, void* out, void* stream) {
    MDNO_REQUIRE(in && out && count >= 0 && count % 4 == 0, MDNO_EINVAL, "mdno_cast_bf16: count=%lld (multiple of 4)",
                 (long long)count);
    if (count == 0) return MDNO_OK;
    const long long n4 = count / 4;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       in, n4, static_cast<uint2*>(out));
    return check_launch("cast_bf16_kernel");
}

extern "C" size_t mdno_linear_bf16_workspace_bytes(int n, int k) {
    return align_up((size_t)n * k * sizeof(__bf16), 256);
}

extern "C" int mdno_linear_bf16_fwd(const void* a, const float* w, const float* bias, int64_t rows, int n, int k,
                                    int relu, int out_bf16, void* c, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    MDNO_REQUIRE(a && w && c && workspace && rows > 0, MDNO_EINVAL, "mdno_linear_bf16_fwd: bad arguments");
    MDNO_REQUIRE(n % 128 == 0 && k % 32 == 0, MDNO_EUNSUPPORTED, "mdno_linear_bf16_fwd: n=%d (x128) k=%d (x32)", n, k);
    MDNO_REQUIRE(workspace_bytes >= mdno_linear_bf16_workspace_bytes(n, k), MDNO_EWORKSPACE,
                 "mdno_linear_bf16_fwd: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    MDNO_TRY(mdno_cast_bf16(w, (int64_t)n * k, workspace, stream));      // master weights -> bf16, every call
    if (gemm_nt_pp_supported(rows, n, k))      // 256 x 256 tiles, LDS-DMA ring, two wave groups a phase apart
        return gemm_nt_pp(a, workspace, bias, (long long)rows, n, k, relu, out_bf16, c, s);
    const __bf16* A = static_cast<const __bf16*>(a);
    const __bf16* W = static_cast<const __bf16*>(workspace);
    // (an XCD-aware order — all column tiles of a row tile on one XCD — was measured: 617 -> 640 us on the
    // K = 1024, N = 4096 product, no change on the others; the plain grid stays)
    const dim3 grid(n / 128, (unsigned)((rows + 127) / 128));
#define MDNO_GO(R, O) hipLaunchKernelGGL((gemm_nt_bf16_kernel<R, O>), grid, dim3(256), 0, s, A, W, bias, c, (long long)rows, n, k)
    if (relu) { if (out_bf16) MDNO_GO(true, true); else MDNO_GO(true, false); }
    else      { if (out_bf16) MDNO_GO(false, true); else MDNO_GO(false, false); }
#undef MDNO_GO
    return check_launch("gemm_nt_bf16_kernel");
}

extern "C" int mdno_linear_smallk_bf16_fwd(const float* a, const float* w, const float* bias, int64_t rows, int n, int k,
                                           int relu, void* c, void* stream) {
    MDNO_REQUIRE(a && w && c && rows > 0, MDNO_EINVAL, "mdno_linear_smallk_bf16_fwd: bad arguments");
    MDNO_REQUIRE(k >= 1 && k <= 8 && n % 8 == 0 && (reinterpret_cast<uintptr_t>(c) & 15) == 0, MDNO_EUNSUPPORTED,
                 "mdno_linear_smallk_bf16_fwd: k=%d (1..8) n=%d (x8), c 16-byte aligned", k, n);
    const long long blocks = (rows + SK_ROWS - 1) / SK_ROWS;
    MDNO_REQUIRE(blocks < (1ll << 31), MDNO_EUNSUPPORTED, "mdno_linear_smallk_bf16_fwd: too many rows");
    const dim3 grid((unsigned)blocks);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (relu) hipLaunchKernelGGL(linear_smallk_bf16_kernel<true>, grid, dim3(256), 0, s, a, w, bias, (long long)rows, n, k,
                                 static_cast<uint4*>(c));
    else hipLaunchKernelGGL(linear_smallk_bf16_kernel<false>, grid, dim3(256), 0, s, a, w, bias, (long long)rows, n, k,
                            static_cast<uint4*>(c));
    return check_launch("linear_smallk_bf16_kernel");
}

extern "C" size_t mdno_gemm_atb_bf16_workspace_bytes(int n1, int n2) {
    const size_t old_path = align_up((size_t)kTnSlices * n1 * n2 * sizeof(float), 256);
    if (!gemm_tn_pp_supported(1, n1, n2)) return old_path;
    const size_t pp = gemm_tn_pp_workspace_bytes(1, n1, n2);      // (32 slabs: what any row count up to 2^31 / widest * 32 needs)
    return pp > old_path ? pp : old_path;
}

extern "C" int mdno_linear_bf16_masked_supported(int64_t rows, int n, int k) { return gemm_nt_pp_supported(rows, n, k) ? 1 : 0; }

extern "C" int mdno_linear_bf16_masked(const void* a, const float* w, const void* y, int64_t rows, int n, int k, void* c,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    MDNO_REQUIRE(a && w && y && c && workspace && rows > 0, MDNO_EINVAL, "mdno_linear_bf16_masked: bad arguments");
    MDNO_REQUIRE(gemm_nt_pp_supported(rows, n, k), MDNO_EUNSUPPORTED, "mdno_linear_bf16_masked: n=%d (x256) k=%d (x32, >= 64)", n, k);
    MDNO_REQUIRE(workspace_bytes >= mdno_linear_bf16_workspace_bytes(n, k), MDNO_EWORKSPACE,
                 "mdno_linear_bf16_masked: workspace too small");
    MDNO_TRY(mdno_cast_bf16(w, (int64_t)n * k, workspace, stream));      // master weights -> bf16, every call
    return gemm_nt_pp_masked(a, workspace, y, (long long)rows, n, k, c, static_cast<hipStream_t>(stream));
}

extern "C" int mdno_gemm_atb_bf16(const void* a, const void* b, int64_t rows, int n1, int n2, float* c,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    MDNO_REQUIRE(a && b && c && workspace && rows > 0, MDNO_EINVAL, "mdno_gemm_atb_bf16: bad arguments");
    MDNO_REQUIRE(n1 % 128 == 0 && n2 % 128 == 0, MDNO_EUNSUPPORTED, "mdno_gemm_atb_bf16: n1=%d n2=%d (x128)", n1, n2);
    MDNO_REQUIRE(workspace_bytes >= mdno_gemm_atb_bf16_workspace_bytes(n1, n2), MDNO_EWORKSPACE,
                 "mdno_gemm_atb_bf16: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (gemm_tn_pp_supported(rows, n1, n2)) {
        // the slab count grows past 32 once a slice's operand panel would reach 2 GiB (rows >= ~8.4M at n = 4096);
        // the row-independent workspace query does not cover that: refuse rather than write past the buffer
        MDNO_REQUIRE(workspace_bytes >= gemm_tn_pp_workspace_bytes((long long)rows, n1, n2), MDNO_EWORKSPACE,
                     "mdno_gemm_atb_bf16: %lld rows need more slabs than mdno_gemm_atb_bf16_workspace_bytes provides",
                     (long long)rows);
        return gemm_tn_pp(a, b, (long long)rows, n1, n2, c, workspace, s);
    }
    const long long slice_rows = ((rows + kTnSlices - 1) / kTnSlices + 31) / 32 * 32;
    hipLaunchKernelGGL(gemm_tn_bf16_kernel, dim3(n2 / 128, n1 / 128, kTnSlices), dim3(256), 0, s,
                       static_cast<const __bf16*>(a), static_cast<const __bf16*>(b), static_cast<float*>(workspace),
                       (long long)rows, n1, n2, slice_rows);
    const long long count = (long long)n1 * n2;
    launch_reduce_slices(static_cast<const float*>(workspace), kTnSlices, count, c, 0, s);
    return check_launch("gemm_tn_bf16_kernel");
}

extern "C" int mdno_nnconv_bf16w_fwd(const float* x, const int32_t* row_ptr, const int32_t* src, int num_rows,
                                     const void* w_e, const float* root, const float* bias, int aggr, int relu,
                                     float* y, void* stream) {
    MDNO_REQUIRE(x && row_ptr && src && w_e && y && num_rows > 0, MDNO_EINVAL, "mdno_nnconv_bf16w_fwd: bad arguments");
    MDNO_REQUIRE(aggr == MDNO_AGGR_ADD || aggr == MDNO_AGGR_MEAN, MDNO_EUNSUPPORTED, "mdno_nnconv_bf16w_fwd: aggr %d", aggr);
    // many short rows (a training batch: 3,584 rows of ~12 edges): four waves per row keep four times as many rows
    // resident per CU — 30.4k instead of 29.5k samples/s on cfg4; few rows: a wave per chain
    if (num_rows >= 2048)
        hipLaunchKernelGGL(nnconv64_bf16w_kernel<4>, dim3(num_rows), dim3(256), 0, static_cast<hipStream_t>(stream), x, row_ptr,
                           src, static_cast<const __bf16*>(w_e), root, bias, y, num_rows, aggr, relu);
    else
        hipLaunchKernelGGL(nnconv64_bf16w_kernel<16>, dim3(num_rows), dim3(1024), 0, static_cast<hipStream_t>(stream), x,
                           row_ptr, src, static_cast<const __bf16*>(w_e), root, bias, y, num_rows, aggr, relu);
    return check_launch("nnconv64_bf16w_kernel");
}

extern "C" int mdno_nnconv_bwd_x_bf16w(const float* gz, const float* gs, const int32_t* row_ptr_s, const int32_t* eid_s,
                                       const int32_t* dst_s, int num_rows, const void* w_e, const float* root,
                                       float* g_prev, void* stream) {
    MDNO_REQUIRE(gz && gs && row_ptr_s && eid_s && dst_s && w_e && g_prev && num_rows > 0, MDNO_EINVAL,
                 "mdno_nnconv_bwd_x_bf16w: bad arguments");
    hipLaunchKernelGGL(nnconv_bwd_x_bf16w_kernel, dim3(num_rows), dim3(256), 0, static_cast<hipStream_t>(stream), gz, gs,
                       row_ptr_s, eid_s, dst_s, static_cast<const __bf16*>(w_e), root, g_prev, num_rows);
    return check_launch("nnconv_bwd_x_bf16w_kernel");
}

extern "C" int mdno_nnconv_bwd_we_bf16(const float* x, const float* gs, const int32_t* src, const int32_t* dst, int64_t E,
                                       int L, int64_t layer_stride, void* d_we, void* stream) {
    MDNO_REQUIRE(x && gs && src && dst && d_we && E >= 0 && L > 0, MDNO_EINVAL, "mdno_nnconv_bwd_we_bf16: bad arguments");
    if (E == 0) return MDNO_OK;
    hipLaunchKernelGGL(nnconv_bwd_we_bf16_kernel, dim3((unsigned)((E + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, gs, src, dst, (long long)E, L, (long long)layer_stride,
                       static_cast<__bf16*>(d_we));
    return check_launch("nnconv_bwd_we_bf16_kernel");
}

extern "C" size_t mdno_nnconv_bwd_we_colsum_workspace_bytes(void) { return align_up((size_t)WE_WGS * 4096 * sizeof(float), 256); }
extern "C" size_t mdno_nnconv_bwd_we_bf16_colsum_workspace_bytes(void) { return mdno_nnconv_bwd_we_colsum_workspace_bytes(); }

template <bool BF16>
static int bwd_we_colsum(const char* what, const float* x, const float* gs, const int32_t* src, const int32_t* dst, int64_t E, int L,
                         int64_t layer_stride, void* d_we, float* colsum, void* workspace, size_t workspace_bytes, void* stream) {
    MDNO_REQUIRE(x && gs && src && dst && d_we && colsum && workspace && E >= 0 && L > 0, MDNO_EINVAL, "%s: bad arguments", what);
    MDNO_REQUIRE(L <= 16, MDNO_EUNSUPPORTED, "%s: %d conv applications (one MFMA k-step holds 16)", what, L);
    MDNO_REQUIRE(workspace_bytes >= mdno_nnconv_bwd_we_colsum_workspace_bytes(), MDNO_EWORKSPACE, "%s: workspace too small", what);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(nnconv_bwd_we_mfma_kernel<BF16>, dim3(WE_WGS), dim3(256), 0, s, x, gs, src, dst, (long long)E, L,
                       (long long)layer_stride, d_we, static_cast<float*>(workspace));
    launch_reduce_slices(static_cast<const float*>(workspace), WE_WGS, 4096, colsum, 0, s);
    return check_launch(what);
}

extern "C" int mdno_nnconv_bwd_we_bf16_colsum(const float* x, const float* gs, const int32_t* src, const int32_t* dst, int64_t E,
                                              int L, int64_t layer_stride, void* d_we, float* colsum, void* workspace,
                                              size_t workspace_bytes, void* stream) {
    return bwd_we_colsum<true>("mdno_nnconv_bwd_we_bf16_colsum", x, gs, src, dst, E, L, layer_stride, d_we, colsum, workspace,
                               workspace_bytes, stream);
}

extern "C" int mdno_nnconv_bwd_we_colsum(const float* x, const float* gs, const int32_t* src, const int32_t* dst, int64_t E, int L,
                                         int64_t layer_stride, float* d_we, float* colsum, void* workspace, size_t workspace_bytes,
                                         void* stream) {
    return bwd_we_colsum<false>("mdno_nnconv_bwd_we_colsum", x, gs, src, dst, E, L, layer_stride, d_we, colsum, workspace,
                                workspace_bytes, stream);
}

extern "C" int mdno_relu_bwd_bf16(const float* g, const void* y, int64_t rows, int n, int out_bf16, void* out,
                                  void* stream) {
    MDNO_REQUIRE(g && y && out && rows >= 0 && n % 4 == 0, MDNO_EINVAL, "mdno_relu_bwd_bf16: bad arguments");
    const long long groups = (long long)rows * (n / 4);
    if (groups == 0) return MDNO_OK;
    const dim3 grid((unsigned)((groups + 255) / 256));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (out_bf16)
        hipLaunchKernelGGL(relu_bwd_bf16_kernel<true>, grid, dim3(256), 0, s, g, static_cast<const __bf16*>(y),
                           (long long)rows, n, out);
    else
        hipLaunchKernelGGL(relu_bwd_bf16_kernel<false>, grid, dim3(256), 0, s, g, static_cast<const __bf16*>(y),
                           (long long)rows, n, out);
    return check_launch("relu_bwd_bf16_kernel");
}

extern "C" size_t mdno_colsum_bf16_workspace_bytes(int n) { return align_up((size_t)kColSlicesB * n * sizeof(float), 256); }

extern "C" int mdno_colsum_bf16(const void* a, int64_t rows, int n, float* out, void* workspace, size_t workspace_bytes,
                                void* stream) {
    MDNO_REQUIRE(a && out && workspace && rows > 0 && n > 0, MDNO_EINVAL, "mdno_colsum_bf16: bad arguments");
    MDNO_REQUIRE(workspace_bytes >= mdno_colsum_bf16_workspace_bytes(n), MDNO_EWORKSPACE, "mdno_colsum_bf16: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long slice_rows = (rows + kColSlicesB - 1) / kColSlicesB;
    MDNO_REQUIRE(n % 8 == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0, MDNO_EUNSUPPORTED, "mdno_colsum_bf16: n=%d (x8)", n);
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3((n + 511) / 512, kColSlicesB), dim3(256), 0, s, static_cast<const __bf16*>(a),
                       static_cast<float*>(workspace), (long long)rows, n, slice_rows);
    launch_reduce_slices(static_cast<const float*>(workspace), kColSlicesB, (long long)n, out, 0, s);
    return check_launch("colsum_bf16_kernel");
}

// ---------------------------------------------------------------- the conv applications of a training step as ONE call
extern "C" int mdno_nnconv_chain_bf16w_fwd(float* x_layers, const int32_t* row_ptr, const int32_t* src, int num_rows,
                                           const void* w_e, const float* root1, const float* bias1, const float* root2,
                                           const float* bias2, int depth, void* stream) {
    MDNO_REQUIRE(x_layers && row_ptr && src && w_e && num_rows > 0 && depth > 0, MDNO_EINVAL,
                 "mdno_nnconv_chain_bf16w_fwd: bad arguments");
    const size_t stride = (size_t)num_rows * 64;
    for (int a = 1; a <= 2 * depth; ++a)
        MDNO_TRY(mdno_nnconv_bf16w_fwd(x_layers + (a - 1) * stride, row_ptr, src, num_rows, w_e, a <= depth ? root1 : root2,
                                       a <= depth ? bias1 : bias2, MDNO_AGGR_MEAN, 1, x_layers + a * stride, stream));
    return MDNO_OK;
}

extern "C" int mdno_nnconv_chain_bf16w_bwd(const float* g_out, const float* x_layers, const float* inv_deg,
                                           const int32_t* row_ptr_s, const int32_t* eid_s, const int32_t* dst_s,
                                           int num_rows, const void* w_e, const float* root1, const float* root2, int depth,
                                           float* gz, float* gs, float* g_in, void* stream) {
    MDNO_REQUIRE(g_out && x_layers && inv_deg && row_ptr_s && eid_s && dst_s && w_e && gz && gs && g_in && num_rows > 0 &&
                     depth > 0, MDNO_EINVAL, "mdno_nnconv_chain_bf16w_bwd: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int L = 2 * depth;
    const size_t stride = (size_t)num_rows * 64;
    // the gradient enters through the ReLU of application L; every later ReLU is the epilogue of the kernel above it
    MDNO_TRY(mdno_relu_bwd2(g_out, x_layers + L * stride, inv_deg, num_rows, 64, gz + (L - 1) * stride, gs + (L - 1) * stride,
                            stream));
    for (int a = L; a >= 1; --a) {
        const float* root = a <= depth ? root1 : root2;
        if (a > 1)
            hipLaunchKernelGGL(nnconv_bwd_x_bf16w_kernel, dim3(num_rows), dim3(256), 0, s, gz + (a - 1) * stride,
                               gs + (a - 1) * stride, row_ptr_s, eid_s, dst_s, static_cast<const __bf16*>(w_e), root,
                               (float*)nullptr, num_rows, x_layers + (a - 1) * stride, inv_deg, gz + (a - 2) * stride,
                               gs + (a - 2) * stride);
        else
            hipLaunchKernelGGL(nnconv_bwd_x_bf16w_kernel, dim3(num_rows), dim3(256), 0, s, gz, gs, row_ptr_s, eid_s, dst_s,
                               static_cast<const __bf16*>(w_e), root, g_in, num_rows, (const float*)nullptr,
                               (const float*)nullptr, (float*)nullptr, (float*)nullptr);
    }
    return check_launch("mdno_nnconv_chain_bf16w_bwd");
}

extern "C" size_t mdno_colsum_atb_bf16_workspace_bytes(int n, int kb) {
    return align_up((size_t)(kColSlicesB + 1) * (size_t)(kb + 1) * n * sizeof(float), 256);
}

extern "C" int mdno_colsum_atb_bf16(const void* a, const float* b, int64_t rows, int n, int kb, float* colsum, float* atb,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    MDNO_REQUIRE(a && b && colsum && atb && workspace && rows > 0 && n > 0, MDNO_EINVAL, "mdno_colsum_atb_bf16: bad arguments");
    MDNO_REQUIRE(n % 8 == 0 && (kb == 6 || kb == 8) && (reinterpret_cast<uintptr_t>(a) & 15) == 0, MDNO_EUNSUPPORTED,
                 "mdno_colsum_atb_bf16: n=%d (x8), kb=%d (6 or 8)", n, kb);
    MDNO_REQUIRE(workspace_bytes >= mdno_colsum_atb_bf16_workspace_bytes(n, kb), MDNO_EWORKSPACE,
                 "mdno_colsum_atb_bf16: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long slice_rows = (rows + kColSlicesB - 1) / kColSlicesB;
    float* part = static_cast<float*>(workspace);
    float* red = part + (size_t)kColSlicesB * (kb + 1) * n;
    const dim3 grid((n + 511) / 512, kColSlicesB);
    if (kb == 6)
        hipLaunchKernelGGL(colsum_atb_bf16_kernel<6>, grid, dim3(256), 0, s, static_cast<const __bf16*>(a), b, part,
                           (long long)rows, n, slice_rows);
    else
        hipLaunchKernelGGL(colsum_atb_bf16_kernel<8>, grid, dim3(256), 0, s, static_cast<const __bf16*>(a), b, part,
                           (long long)rows, n, slice_rows);
    launch_reduce_slices(part, kColSlicesB, (long long)(kb + 1) * n, red, 0, s);
    if (kb == 6)
        hipLaunchKernelGGL(colsum_atb_finish_kernel<6>, dim3((n + 255) / 256), dim3(256), 0, s, (const float*)red, n, colsum, atb);
    else
        hipLaunchKernelGGL(colsum_atb_finish_kernel<8>, dim3((n + 255) / 256), dim3(256), 0, s, (const float*)red, n, colsum, atb);
    return check_launch("mdno_colsum_atb_bf16");
}
