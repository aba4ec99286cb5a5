// K1+K2: edge attributes -> edge-MLP -> W_e, written in CSR edge order.
//
// Replaces DenseNet.forward (graph_kernel.py:239-242) as called per conv application from
// NNConv_old.message (:201), and the per-edge Python loop that builds edge_attr (:372-379).
//   layer 0  [E,ker_in] x [ker_in,k]   + bias, ReLU   (VALU; K = 6, fused with the attr gather)
//   layer 1  [E,k]      x [k,k]        + bias, ReLU   (fp32 MFMA GEMM)
//   layer 2  [E,k]      x [k,Cin*Cout] + bias         (fp32 MFMA GEMM; 80 % of the model's flops)
//
// Roofline: MFMA (v_mfma_f32_32x32x2_f32, exact fp32, 157 TF peak).  Flops per edge =
// 2*(ker_in*k + k*k + k*Cin*Cout) = 10,498,048 at k=1024, 64x64.
//
// GEMM shape: C[M,N] = act(A[M,K] . Bt[N,K]^T + bias[N]) with BOTH operands K-contiguous (torch
// Linear keeps weight as [out,in]), which is the natural operand order for MFMA: lane l feeds
// A[row l&31][k] and Bt[col l&31][k].  Block tile 128x128x32, 4 waves as 2x2, each wave 64x64 =
// 2x2 MFMA tiles (4 independent accumulators keep the matrix pipe issuing back-to-back).  Tiles go
// global -> registers -> LDS (rows padded to 36 floats so ds_read_b128 is bank-conflict-free),
// double-buffered, one barrier per K-tile.  Each lane reads its fragment as one 16-B LDS load that
// covers FOUR k-steps: lane half h reads k = 8t+4h..8t+4h+3; MFMA step c of that group contracts
// k = 8t+4h+c in both operands, which is a permutation of the k order and leaves the sum's value
// set unchanged (fp32 rounding order differs from a left-to-right dot product, as any blocked GEMM).
//
// The number of valid rows (edges) lives in device memory (the graph is rebuilt on the device every
// rollout step), so grids are sized by capacity and whole tiles beyond *num_edges exit at once.
#include "kernels.h"
#include "mfma_f32.h"

namespace mdno {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32, LDS_LD = BK + 4;  // 36 floats = 144 B rows

// ---------------------------------------------------------------- layer 0 (+ attr gather)
// Block = 256 threads handles EB consecutive edges; thread c-strided over the k hidden units.
constexpr int EB = 16;
constexpr int MAX_F = 8;

__global__ __launch_bounds__(256) void edge_l0_kernel(
    const float* __restrict__ frames, int frame, const int* __restrict__ t_dev, int rows_per_frame,
    const int* __restrict__ src, const int* __restrict__ dst,
    const float* __restrict__ edge_attr, const int* __restrict__ perm, const int* __restrict__ num_edges,
    long long e_begin, int e_count, int F, int k, const float* __restrict__ w0, const float* __restrict__ b0,
    float* __restrict__ h) {
    __shared__ float attr[EB][MAX_F];
    const long long E = *num_edges;
    const long long e0 = e_begin + (long long)blockIdx.x * EB;
    if (e0 >= E || (long long)blockIdx.x * EB >= e_count) return;
    const int tid = threadIdx.x;
    if (tid < EB * MAX_F) {
        const int le = tid / MAX_F, f = tid % MAX_F;
        const long long e = e0 + le;
        float v = 0.f;
        if (e < E && f < F) {
            if (frames != nullptr) {  // attr = [pos[src], pos[dst]]   (graph_kernel.py:372-379)
                const float* edge_pos = frames + (size_t)(frame + (t_dev ? *t_dev : 0)) * rows_per_frame * 3;
                const int node = (f < 3) ? src[e] : dst[e];
                v = edge_pos[(size_t)node * 3 + (f % 3)];
            } else {
                const long long pe = perm ? (long long)perm[e] : e;
                v = edge_attr[pe * F + f];
            }
        }
        attr[le][f] = v;
    }
    __syncthreads();
    for (int c = tid; c < k; c += 256) {
        float w[MAX_F];
#pragma unroll
        for (int f = 0; f < MAX_F; ++f) w[f] = (f < F) ? w0[(size_t)c * F + f] : 0.f;
        const float bc = b0[c];
#pragma unroll 4
        for (int le = 0; le < EB; ++le) {
            const long long e = e0 + le;
            if (e >= E || e - e_begin >= e_count) break;
            float s = 0.f;
#pragma unroll
            for (int f = 0; f < MAX_F; ++f) s = fmaf(attr[le][f], w[f], s);
            s += bc;
            h[(size_t)(e - e_begin) * k + c] = relu_f(s);
        }
    }
}

// ---------------------------------------------------------------- fp32 MFMA GEMM (TN)
struct GemmArgs {
    const float* A;      // [rows, K] rows local to the chunk
    const float* Bt;     // [N, K]
    const float* bias;   // [N]
    float* C;            // [rows, N] rows local to the chunk
    const int* num_edges;
    long long row_begin;  // global edge index of local row 0
    int rows;             // chunk rows (capacity)
    int N, K;
    int tiled_out;        // 1: C written k-tiled [rows/128][N/32][128][32] (what csrc/moment.hip K1 streams)
};

// One K-tile of MFMA work for a wave: 2x2 tiles of 32x32 (mfma_f32.h)
static_assert(LDS_LD == f32mma::LD && BK == f32mma::BK, "tile helpers assume 36-float LDS rows");
__device__ __forceinline__ void mma_tile(f32x16 (&acc)[2][2], const float* __restrict__ a_base,
                                         const float* __restrict__ b_base) {
    f32mma::mma_64x64(acc, a_base, b_base);
}

template <bool RELU>
__global__ __launch_bounds__(256, 2) void gemm_tn_mfma_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [2][BM][LDS_LD]
    float* Bs = smem + 2 * BM * LDS_LD;     // [2][BN][LDS_LD]

    long long valid = (long long)(*g.num_edges) - g.row_begin;
    if (valid > g.rows) valid = g.rows;
    const int bm = blockIdx.y * BM;
    if (bm >= valid) return;
    const int bn = blockIdx.x * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;

    // global -> register staging: thread covers rows (tid>>3) + 32*i, 16-B column chunk (tid&7).
    // Named registers (no arrays, no lambdas): anything the compiler cannot keep in VGPRs here
    // goes to scratch and serialises the loads.
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    const size_t ldk = (size_t)g.K;
    const float* Ag = g.A + (size_t)(bm + srow) * ldk + scol;
    const float* Bg = g.Bt + (size_t)(bn + srow) * ldk + scol;
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define MDNO_LOAD_TILE(KOFF)                                                        \
    ra0 = *reinterpret_cast<const float4*>(Ag + (KOFF));                            \
    ra1 = *reinterpret_cast<const float4*>(Ag + 32 * ldk + (KOFF));                 \
    ra2 = *reinterpret_cast<const float4*>(Ag + 64 * ldk + (KOFF));                 \
    ra3 = *reinterpret_cast<const float4*>(Ag + 96 * ldk + (KOFF));                 \
    rb0 = *reinterpret_cast<const float4*>(Bg + (KOFF));                            \
    rb1 = *reinterpret_cast<const float4*>(Bg + 32 * ldk + (KOFF));                 \
    rb2 = *reinterpret_cast<const float4*>(Bg + 64 * ldk + (KOFF));                 \
    rb3 = *reinterpret_cast<const float4*>(Bg + 96 * ldk + (KOFF));
    float* a_st = As + srow * LDS_LD + scol;
    float* b_st = Bs + srow * LDS_LD + scol;
#define MDNO_STORE_TILE(BUF)                                                                    \
    *reinterpret_cast<float4*>(a_st + (BUF) * BM * LDS_LD) = ra0;                               \
    *reinterpret_cast<float4*>(a_st + (BUF) * BM * LDS_LD + 32 * LDS_LD) = ra1;                 \
    *reinterpret_cast<float4*>(a_st + (BUF) * BM * LDS_LD + 64 * LDS_LD) = ra2;                 \
    *reinterpret_cast<float4*>(a_st + (BUF) * BM * LDS_LD + 96 * LDS_LD) = ra3;                 \
    *reinterpret_cast<float4*>(b_st + (BUF) * BN * LDS_LD) = rb0;                               \
    *reinterpret_cast<float4*>(b_st + (BUF) * BN * LDS_LD + 32 * LDS_LD) = rb1;                 \
    *reinterpret_cast<float4*>(b_st + (BUF) * BN * LDS_LD + 64 * LDS_LD) = rb2;                 \
    *reinterpret_cast<float4*>(b_st + (BUF) * BN * LDS_LD + 96 * LDS_LD) = rb3;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const float* a_rd = As + (wm * 64 + l31) * LDS_LD + 4 * h;
    const float* b_rd = Bs + (wn * 64 + l31) * LDS_LD + 4 * h;
    const int nk = g.K / BK;
    MDNO_LOAD_TILE(0)
    MDNO_STORE_TILE(0)
    __syncthreads();
    // steady state: prefetch tile kt+1 into registers, multiply tile kt from LDS, publish kt+1
    for (int kt = 0; kt < nk - 1; ++kt) {
        const int buf = kt & 1;
        MDNO_LOAD_TILE((size_t)(kt + 1) * BK)
        mma_tile(acc, a_rd + buf * BM * LDS_LD, b_rd + buf * BN * LDS_LD);
        MDNO_STORE_TILE(buf ^ 1)
        __syncthreads();
    }
    {
        const int buf = (nk - 1) & 1;
        mma_tile(acc, a_rd + buf * BM * LDS_LD, b_rd + buf * BN * LDS_LD);
    }
#undef MDNO_LOAD_TILE
#undef MDNO_STORE_TILE

    // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
    // The bias values are loaded once and pinned (see edge_mlp_split.hip: sunk into the predicated
    // store blocks, the load puts an s_waitcnt vmcnt(0) in front of every store).
    float bv0 = 0.f, bv1 = 0.f;
    if (g.bias) {
        bv0 = g.bias[bn + wn * 64 + l31];
        bv1 = g.bias[bn + wn * 64 + 32 + l31];
    }
    asm volatile("" : "+v"(bv0), "+v"(bv1));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = bn + wn * 64 + j * 32 + l31;
        const float bv = j ? bv1 : bv0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = bm + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < valid) {
                    float v = acc[i][j][r] + bv;
                    if (RELU) v = relu_f(v);
                    if (g.tiled_out)
                        g.C[((size_t)(m >> 7) * (g.N >> 5) + (n >> 5)) * 4096 + (m & 127) * 32 + (n & 31)] = v;
                    else
                        g.C[(size_t)m * g.N + n] = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------- any-shape fallback (fixtures)
template <bool RELU>
__global__ __launch_bounds__(256) void gemm_tn_generic_kernel(GemmArgs g) {
    __shared__ float As[16][17], Bs[16][17];
    long long valid = (long long)(*g.num_edges) - g.row_begin;
    if (valid > g.rows) valid = g.rows;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m = blockIdx.y * 16 + ty, n = blockIdx.x * 16 + tx;
    if ((long long)blockIdx.y * 16 >= valid) return;
    float s = 0.f;
    for (int k0 = 0; k0 < g.K; k0 += 16) {
        const int am = blockIdx.y * 16 + ty, bn_ = blockIdx.x * 16 + ty;
        As[ty][tx] = (am < valid && k0 + tx < g.K) ? g.A[(size_t)am * g.K + k0 + tx] : 0.f;
        Bs[ty][tx] = (bn_ < g.N && k0 + tx < g.K) ? g.Bt[(size_t)bn_ * g.K + k0 + tx] : 0.f;
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) s = fmaf(As[ty][kk], Bs[tx][kk], s);
        __syncthreads();
    }
    if (m < valid && n < g.N) {
        float v = s + (g.bias ? g.bias[n] : 0.f);
        if (RELU) v = relu_f(v);
        g.C[(size_t)m * g.N + n] = v;
    }
}

template <bool RELU>
int launch_gemm(const GemmArgs& g, hipStream_t s) {
    TimedSection ts(RELU ? KID_GEMM_L1 : KID_GEMM_L2, s);
    const bool mfma_ok = (g.K % BK == 0) && (g.N % BN == 0) && (g.rows % BM == 0) &&
                         ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.Bt)) & 15) == 0;
    if (mfma_ok) {
        const size_t lds = sizeof(float) * 2 * (BM + BN) * LDS_LD;  // 73,728 B
        static std::atomic<unsigned long long> lds_raised{0};   // one per <RELU> instantiation
        MDNO_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(&gemm_tn_mfma_kernel<RELU>), (int)lds, lds_raised));
        dim3 grid(g.N / BN, g.rows / BM);
        hipLaunchKernelGGL(gemm_tn_mfma_kernel<RELU>, grid, dim3(256), lds, s, g);
    } else {
        dim3 grid((g.N + 15) / 16, (g.rows + 15) / 16);
        hipLaunchKernelGGL(gemm_tn_generic_kernel<RELU>, grid, dim3(256), 0, s, g);
    }
    return check_launch("edge-MLP GEMM");
}

constexpr long long kMaxChunkRows = 262144;

long long chunk_rows_for(long long edge_cap) {
    long long r = (edge_cap + 255) / 256 * 256;   // multiple of both GEMM kernels' row tiles (128, 256)
    return r < kMaxChunkRows ? r : kMaxChunkRows;
}

}  // namespace
}  // namespace mdno

using namespace mdno;

extern "C" size_t mdno_edge_mlp_workspace_bytes(int ker_width, int out_dim, int64_t edge_cap, int gemm_mode) {
    if (ker_width <= 0 || out_dim <= 0 || edge_cap <= 0) return 0;
    const long long chunk = chunk_rows_for(edge_cap);
    if (gemm_mode != MDNO_GEMM_F32 && edge_mlp_split_supported(ker_width, out_dim))
        return edge_mlp_split_workspace_bytes(ker_width, out_dim, chunk);
    return align_up(2 * (size_t)chunk * (size_t)ker_width * sizeof(float) + 512, 256);
}

int* mdno::edge_mlp_activation_flags(void* workspace, int ker_width, int out_dim, long long edge_cap, int gemm_mode) {
    if (gemm_mode != MDNO_GEMM_SPLIT_F16 || !edge_mlp_split_supported(ker_width, out_dim) || ker_width % 32 != 0 ||
        edge_cap <= 0)
        return nullptr;
    return edge_mlp_split_activation_flags(workspace, ker_width, out_dim, chunk_rows_for(edge_cap));
}

int mdno::edge_mlp(const float* frames, int frame, const int* t_dev, int rows_per_frame, const int* src,
                   const int* dst, const float* edge_attr, const int* perm, const int* num_edges,
                   long long edge_cap, int ker_in, int ker_width, int out_dim, int gemm_mode,
                   const EdgeMlpWeights& w, float* w_e, void* workspace, size_t workspace_bytes, hipStream_t s,
                   int phase) {
    MDNO_REQUIRE(num_edges && w.w0 && w.b0 && w.w1 && w.b1 && w.w2 && w.b2 && w_e && workspace, MDNO_EINVAL,
                 "edge_mlp: null pointer");
    MDNO_REQUIRE((frames && src && dst) || edge_attr, MDNO_EINVAL, "edge_mlp: need (edge_pos, src, dst) or edge_attr");
    MDNO_REQUIRE(edge_cap > 0 && ker_width > 0 && out_dim > 0, MDNO_EINVAL, "edge_mlp: bad sizes");
    MDNO_REQUIRE(ker_in > 0 && ker_in <= MAX_F, MDNO_EUNSUPPORTED, "edge_mlp: ker_in=%d (1..%d)", ker_in, MAX_F);
    MDNO_REQUIRE(edge_attr || ker_in == 6, MDNO_EINVAL, "edge_mlp: position-derived attributes need ker_in == 6");
    MDNO_REQUIRE(gemm_mode == MDNO_GEMM_SPLIT_BF16 || gemm_mode == MDNO_GEMM_F32 || gemm_mode == MDNO_GEMM_SPLIT_F16,
                 MDNO_EINVAL, "edge_mlp: gemm_mode=%d", gemm_mode);
    const size_t need = mdno_edge_mlp_workspace_bytes(ker_width, out_dim, edge_cap, gemm_mode);
    MDNO_REQUIRE(workspace_bytes >= need, MDNO_EWORKSPACE, "edge_mlp: workspace %zu < %zu", workspace_bytes, need);
    const long long chunk = chunk_rows_for(edge_cap);
    if (gemm_mode != MDNO_GEMM_F32 && edge_mlp_split_supported(ker_width, out_dim))
        return edge_mlp_split(frames, frame, t_dev, rows_per_frame, src, dst, edge_attr, perm, num_edges, edge_cap,
                              chunk, ker_in, ker_width, out_dim, w, w_e, workspace, s, phase,
                              gemm_mode == MDNO_GEMM_SPLIT_F16 && ker_width % 32 == 0);
    if ((phase & WP_PHASE_MASK) == WP_PREPARE_ONLY) return MDNO_OK;   // the fp32 GEMMs read the weights as they are
    Carver cv(workspace);
    float* h1 = cv.take<float>((size_t)chunk * ker_width);
    float* h2 = cv.take<float>((size_t)chunk * ker_width);
    const float* pos_mode = edge_attr ? nullptr : frames;
    for (long long e0 = 0; e0 < edge_cap; e0 += chunk) {
        const int cnt = (int)((edge_cap - e0) < chunk ? (edge_cap - e0) : chunk);
        {
            TimedSection ts(KID_EDGE_L0, s);
            hipLaunchKernelGGL(edge_l0_kernel, dim3((cnt + EB - 1) / EB), dim3(256), 0, s, pos_mode, frame, t_dev,
                               rows_per_frame, src, dst, edge_attr, perm, num_edges, e0, cnt, ker_in, ker_width,
                               w.w0, w.b0, h1);
        }
        MDNO_TRY(check_launch("edge_l0_kernel"));
        GemmArgs g1{h1, w.w1, w.b1, h2, num_edges, e0, (int)chunk, ker_width, ker_width, 0};
        MDNO_TRY(launch_gemm<true>(g1, s));
        // the last layer writes straight into W_e; rows past *num_edges are masked by `valid`
        GemmArgs g2{h2, w.w2, w.b2, w_e + (size_t)e0 * out_dim, num_edges, e0, (int)chunk, out_dim, ker_width, 0};
        MDNO_TRY(launch_gemm<false>(g2, s));
    }
    return MDNO_OK;
}

int mdno::edge_mlp_hidden(const float* frames, int frame, const int* t_dev, int rows_per_frame, const int* src,
                          const int* dst, const float* edge_attr, const int* perm, const int* num_edges,
                          long long edge_cap, int ker_in, int ker_width, int gemm_mode, const EdgeMlpWeights& w,
                          float* h_out, void* workspace, size_t workspace_bytes, hipStream_t s, int phase) {
    MDNO_REQUIRE(num_edges && w.w0 && w.b0 && w.w1 && w.b1 && h_out && workspace, MDNO_EINVAL,
                 "edge_mlp_hidden: null pointer");
    MDNO_REQUIRE((frames && src && dst) || edge_attr, MDNO_EINVAL, "edge_mlp_hidden: need positions+CSR or edge_attr");
    MDNO_REQUIRE(ker_in > 0 && ker_in <= MAX_F, MDNO_EUNSUPPORTED, "edge_mlp: ker_in=%d (1..%d)", ker_in, MAX_F);
    MDNO_REQUIRE(edge_attr || ker_in == 6, MDNO_EINVAL, "edge_mlp: position-derived attributes need ker_in == 6");
    const size_t need = mdno_edge_mlp_workspace_bytes(ker_width, ker_width, edge_cap, gemm_mode);
    MDNO_REQUIRE(workspace_bytes >= need, MDNO_EWORKSPACE, "edge_mlp_hidden: workspace %zu < %zu", workspace_bytes, need);
    const long long chunk = chunk_rows_for(edge_cap);
    if (gemm_mode != MDNO_GEMM_F32 && edge_mlp_split_supported(ker_width, ker_width))
        return edge_mlp_split_hidden(frames, frame, t_dev, rows_per_frame, src, dst, edge_attr, perm, num_edges,
                                     edge_cap, chunk, ker_in, ker_width, w, h_out, workspace, s, phase,
                                     gemm_mode == MDNO_GEMM_SPLIT_F16);
    if ((phase & WP_PHASE_MASK) == WP_PREPARE_ONLY) return MDNO_OK;
    MDNO_REQUIRE(ker_width % BN == 0 && (reinterpret_cast<uintptr_t>(w.w1) & 15) == 0, MDNO_EUNSUPPORTED,
                 "edge_mlp_hidden: ker_width=%d must be a multiple of %d", ker_width, BN);
    Carver cv(workspace);
    float* h1 = cv.take<float>((size_t)chunk * ker_width);
    const float* pos_mode = edge_attr ? nullptr : frames;
    for (long long e0 = 0; e0 < edge_cap; e0 += chunk) {
        const int cnt = (int)((edge_cap - e0) < chunk ? (edge_cap - e0) : chunk);
        {
            TimedSection ts(KID_EDGE_L0, s);
            hipLaunchKernelGGL(edge_l0_kernel, dim3((cnt + EB - 1) / EB), dim3(256), 0, s, pos_mode, frame, t_dev,
                               rows_per_frame, src, dst, edge_attr, perm, num_edges, e0, cnt, ker_in, ker_width,
                               w.w0, w.b0, h1);
        }
        MDNO_TRY(check_launch("edge_l0_kernel"));
        GemmArgs g1{h1, w.w1, w.b1, h_out + (size_t)e0 * ker_width, num_edges, e0, (int)chunk, ker_width, ker_width, 1};
        MDNO_TRY(launch_gemm<true>(g1, s));
    }
    return MDNO_OK;
}

extern "C" int mdno_edge_mlp_fwd(const float* edge_pos, const int32_t* src, const int32_t* dst,
                                 const float* edge_attr, const int32_t* perm, const int32_t* num_edges,
                                 int64_t edge_cap, int ker_in, int ker_width, int out_dim, int gemm_mode,
                                 const float* w0, const float* b0, const float* w1, const float* b1,
                                 const float* w2, const float* b2, float* w_e, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    EdgeMlpWeights w{w0, b0, w1, b1, w2, b2};
    return mdno::edge_mlp(edge_pos, 0, nullptr, 0, src, dst, edge_attr, perm, num_edges, (long long)edge_cap, ker_in,
                          ker_width, out_dim, gemm_mode, w, w_e, workspace, workspace_bytes,
                          static_cast<hipStream_t>(stream));
}
