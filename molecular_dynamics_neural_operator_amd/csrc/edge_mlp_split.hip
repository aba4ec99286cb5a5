// K2 fast path: the edge-MLP GEMMs on the bf16 matrix pipe at fp32-level accuracy.
//
// gfx950 has no TF32/xf32; its exact fp32 MFMA runs at 1/16 of the bf16 rate (157 TF vs ~2.5 PF).
// Every fp32 operand is split exactly into three bf16 planes  x = x_hi + x_mid + x_lo
// (x_hi = bf16(x), x_mid = bf16(x - x_hi), x_lo = bf16(x - x_hi - x_mid); the subtractions are exact
// in fp32) and the product is accumulated in fp32 from the six leading plane products
//     a.b ~= a_hi b_hi + (a_hi b_mid + a_mid b_hi) + (a_hi b_lo + a_lo b_hi + a_mid b_mid),
// each a bf16 x bf16 MFMA with fp32 accumulation (products of bf16 pairs are exact in fp32).  The
// dropped terms are <= 2^-24 |a b|: the result is as close to the fp64 product as a plain fp32 GEMM
// (measured rms 9e-8 vs 2.4e-7 for an fp32 GEMM at K=1024; tests/test_gpu_parity.py).  6 MFMAs at 16x
// the fp32-MFMA rate = 2.67x fewer matrix-pipe cycles than v_mfma_f32_32x32x2_f32.
//
// Operand layout in HBM = the LDS image, tile by tile.  A [rows,K] operand is stored as
//     [row_tile = rows/128][k_tile = K/32][plane 0..2][128 rows][32 k]  bf16,
// each 8 KiB plane tile holding its rows as 64-B lines whose four 16-B chunks are XOR-swizzled by
// (row>>2)&3 — exactly what the fragment reads want in LDS.  Staging a K-tile is then a straight,
// fully coalesced 24 KiB copy per operand (16 B per lane, 1 KiB per wave-instruction) and a
// workgroup walks one contiguous run of K/32 * 24 KiB.  We own both producers and consumers of
// these buffers, so nothing else ever sees the layout:
//   weights      fp32 [N,K] --split_planes_kernel--> tiled planes            (once per forward, 30 MB)
//   layer 0      edge attrs -> relu(linear)  --split--> H1 tiled planes      (edge_l0_split_kernel)
//   layer 1      H1 x W1 -> relu -> split  ---------> H2 tiled planes        (epilogue emits planes)
//   layer 2      H2 x W2 + b  -> fp32 W_e[E, Cin*Cout]  (row-major, what the conv streams)
// so no fp32 activation is ever stored.
//
// GEMM kernel: 128x128x32 block tile, 4 waves (2x2), wave tile 64x64 = 2x2 v_mfma_f32_32x32x16_bf16
// tiles, 48 MFMAs per K-tile per wave.  Each staged operand fragment feeds 2-3 of the six products,
// so LDS and global traffic per MFMA are half those of an ordinary bf16 GEMM.  Staging is
// global -> registers -> LDS with the next tile's loads in flight during the MFMAs; the swizzle
// makes the ds_read_b128 fragment reads bank-conflict-free.  Workgroups are numbered so that each
// XCD owns a contiguous range of tiles (neighbouring tiles share the A row-panel through that
// XCD's L2).
#include "kernels.h"

namespace mdno {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int TM = 128, TN = 128, TK = 32;
constexpr int PLANE_BYTES = TM * TK * 2;       // 8 KiB per operand plane tile
constexpr int LDS_BYTES = 6 * PLANE_BYTES;     // A[3] + B[3] = 48 KiB

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

// Byte offset of element (row, kcol) of plane p in the tiled layout; nkt = K/32.
__device__ __forceinline__ size_t tiled_off(long long row, int kcol, int nkt, int p) {
    const long long rt = row >> 7;
    const int r = (int)(row & 127), kt = kcol >> 5, c = (kcol >> 3) & 3, e = kcol & 7;
    return ((size_t)((rt * nkt + kt) * 3 + p) << 13) + r * 64 + ((c ^ ((r >> 2) & 3)) << 4) + e * 2;
}

// ---------------------------------------------------------------- fp32 [rows,K] -> tiled planes
// thread = one 16-B chunk (8 consecutive k of one row)
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ w, int rows, int K,
                                                           unsigned char* __restrict__ planes) {
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    const int chunks_per_row = K >> 3;
    if (id >= (long long)rows * chunks_per_row) return;
    const int row = (int)(id / chunks_per_row), k0 = (int)(id % chunks_per_row) * 8;
    const float4 v0 = *reinterpret_cast<const float4*>(w + (size_t)row * K + k0);
    const float4 v1 = *reinterpret_cast<const float4*>(w + (size_t)row * K + k0 + 4);
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    __bf16 o[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) split3(x[j], o[0][j], o[1][j], o[2][j]);
    const int nkt = K >> 5;
#pragma unroll
    for (int p = 0; p < 3; ++p)
        *reinterpret_cast<uint4*>(planes + tiled_off(row, k0, nkt, p)) = *reinterpret_cast<const uint4*>(o[p]);
}

// ---------------------------------------------------------------- layer 0 (+ attr gather) -> planes
// Block = 256 threads, EB edges.  Thread = one 16-B chunk (8 hidden units) of one edge at a time:
// k/8 threads cover an edge, the block walks its edges in groups of 256/(k/8).
constexpr int EB = 32;
constexpr int MAX_F = 8;

__global__ __launch_bounds__(256) void edge_l0_split_kernel(
    const float* __restrict__ frames, int frame, const int* __restrict__ t_dev, int rows_per_frame,
    const int* __restrict__ src, const int* __restrict__ dst, const float* __restrict__ edge_attr,
    const int* __restrict__ perm, const int* __restrict__ num_edges, long long e_begin, int e_count, int F, int k,
    const float* __restrict__ w0, const float* __restrict__ b0, unsigned char* __restrict__ hp) {
    __shared__ float attr[EB][MAX_F];
    const long long E = *num_edges;
    const long long e0 = e_begin + (long long)blockIdx.x * EB;
    if (e0 >= E || (long long)blockIdx.x * EB >= e_count) return;
    const int tid = threadIdx.x;
    {
        const int le = tid / MAX_F, f = tid % MAX_F;   // 256 threads = 32 edges x 8 features
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
    const int cpr = k >> 3;                 // chunks per edge row
    const int nkt = k >> 5;
    for (int c0 = (tid % cpr) * 8, le0 = tid / cpr; c0 < k; c0 += 256 * 8) {   // one pass when k/8 <= 256
        float w[8][MAX_F], bc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int f = 0; f < MAX_F; ++f) w[j][f] = (f < F) ? w0[(size_t)(c0 + j) * F + f] : 0.f;
            bc[j] = b0[c0 + j];
        }
        const int estep = (cpr >= 256) ? 1 : 256 / cpr;
        for (int le = le0; le < EB; le += estep) {
            const long long e = e0 + le;
            if (e >= E || e - e_begin >= e_count) break;
            __bf16 o[3][8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float s = 0.f;
#pragma unroll
                for (int f = 0; f < MAX_F; ++f) s = fmaf(attr[le][f], w[j][f], s);
                split3(fmaxf(s + bc[j], 0.f), o[0][j], o[1][j], o[2][j]);
            }
#pragma unroll
            for (int p = 0; p < 3; ++p)
                *reinterpret_cast<uint4*>(hp + tiled_off(e - e_begin, c0, nkt, p)) = *reinterpret_cast<const uint4*>(o[p]);
        }
    }
}

// ---------------------------------------------------------------- split-bf16 GEMM
struct SplitGemmArgs {
    const unsigned char* Ap;   // tiled planes of A  [rows/128][K/32][3][8 KiB]
    const unsigned char* Bp;   // tiled planes of Bt [N/128][K/32][3][8 KiB]
    const float* bias;         // [N]
    float* C;                  // fp32 [rows][N] row-major            (OUT_PLANES = false)
    unsigned char* Cp;         // tiled planes [rows/128][N/32][3][8 KiB], ReLU applied (OUT_PLANES = true)
    const int* num_edges;
    long long row_begin;
    int rows, N, K;
    int tiles_n, tiles_m;
};

// One K-tile (32) for a wave: 2 k-steps x (2x2 tiles) x 6 plane products = 48 MFMAs.
__device__ __forceinline__ void mma_split_tile(f32x16 (&acc)[2][2], const unsigned char* lds, int a_rd, int b_rd,
                                               int fsw, int h) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int coff = ((2 * s + h) ^ fsw) * 16;
        bf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[i][p] = *reinterpret_cast<const bf16x8*>(lds + p * PLANE_BYTES + a_rd + i * 32 * 64 + coff);
                b[i][p] = *reinterpret_cast<const bf16x8*>(lds + p * PLANE_BYTES + b_rd + i * 32 * 64 + coff);
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // smallest terms first
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
            }
    }
}

template <bool OUT_PLANES>
__global__ __launch_bounds__(256, 2) void gemm_split_bf16_kernel(SplitGemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    long long valid = (long long)(*g.num_edges) - g.row_begin;
    if (valid > g.rows) valid = g.rows;
    if (valid <= 0) return;
    // XCD-aware tile order over the tiles that hold valid rows: workgroups b, b+8, ... share an XCD
    // (round-robin dispatch); give each XCD a contiguous range of tiles.  Bijective for any count.
    const int nwg = g.tiles_n * (int)((valid + TM - 1) / TM);
    const int orig = blockIdx.x;
    if (orig >= nwg) return;
    const int xcd = orig & 7, q = nwg >> 3, r8 = nwg & 7;
    const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
    const int bm = (tile / g.tiles_n) * TM;
    const int bn = (tile % g.tiles_n) * TN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;

    // ---- staging: the operand's K-tile is a contiguous 24 KiB LDS image (3 planes x 8 KiB);
    // thread copies 16 B at tid*16 + j*4 KiB, j = 0..5, for A and for B.
    // Named registers only: arrays here end up in scratch and serialise the loads.
    const int nkt = g.K / TK;
    const unsigned char* a_src = g.Ap + ((size_t)(bm >> 7) * nkt * 3 << 13) + tid * 16;
    const unsigned char* b_src = g.Bp + ((size_t)(bn >> 7) * nkt * 3 << 13) + tid * 16;
    const int st_off = tid * 16;
    uint4 a0, a1, a2, a3, a4, a5, b0, b1, b2, b3, b4, b5;
#define MDNO_LD16(P) (*reinterpret_cast<const uint4*>(P))
#define MDNO_SPLIT_LOAD(KT)                                                                   \
    {                                                                                         \
        const unsigned char* pa = a_src + (size_t)(KT) * (3 * PLANE_BYTES);                   \
        const unsigned char* pb = b_src + (size_t)(KT) * (3 * PLANE_BYTES);                   \
        a0 = MDNO_LD16(pa);          a1 = MDNO_LD16(pa + 4096);  a2 = MDNO_LD16(pa + 8192);   \
        a3 = MDNO_LD16(pa + 12288);  a4 = MDNO_LD16(pa + 16384); a5 = MDNO_LD16(pa + 20480);  \
        b0 = MDNO_LD16(pb);          b1 = MDNO_LD16(pb + 4096);  b2 = MDNO_LD16(pb + 8192);   \
        b3 = MDNO_LD16(pb + 12288);  b4 = MDNO_LD16(pb + 16384); b5 = MDNO_LD16(pb + 20480);  \
    }
#define MDNO_ST16(OFF, V) *reinterpret_cast<uint4*>(lds + (OFF)) = (V)
#define MDNO_SPLIT_STORE()                                                                              \
    MDNO_ST16(st_off, a0);          MDNO_ST16(st_off + 4096, a1);   MDNO_ST16(st_off + 8192, a2);        \
    MDNO_ST16(st_off + 12288, a3);  MDNO_ST16(st_off + 16384, a4);  MDNO_ST16(st_off + 20480, a5);       \
    MDNO_ST16(st_off + 24576, b0);  MDNO_ST16(st_off + 28672, b1);  MDNO_ST16(st_off + 32768, b2);       \
    MDNO_ST16(st_off + 36864, b3);  MDNO_ST16(st_off + 40960, b4);  MDNO_ST16(st_off + 45056, b5);

    // ---- fragment read map: row = w*64 + i*32 + l31, logical chunk = 2*s + h, swizzled by (row>>2)&3
    const int fsw = (l31 >> 2) & 3;
    const int a_rd = (wm * 64 + l31) * 64;
    const int b_rd = 3 * PLANE_BYTES + (wn * 64 + l31) * 64;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = nkt;
    MDNO_SPLIT_LOAD(0)
    MDNO_SPLIT_STORE()
    __syncthreads();
    // steady state: next tile's global loads are in flight while this tile is multiplied
    for (int kt = 0; kt < nk - 1; ++kt) {
        MDNO_SPLIT_LOAD(kt + 1)
        mma_split_tile(acc, lds, a_rd, b_rd, fsw, h);
        __syncthreads();                       // every wave is done reading this tile
        MDNO_SPLIT_STORE()
        __syncthreads();
    }
    mma_split_tile(acc, lds, a_rd, b_rd, fsw, h);
#undef MDNO_SPLIT_LOAD
#undef MDNO_SPLIT_STORE
#undef MDNO_LD16
#undef MDNO_ST16

    // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = bn + wn * 64 + j * 32 + l31;
        const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = bm + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < valid) {
                    const float v = acc[i][j][e] + bv;
                    if (OUT_PLANES) {
                        __bf16 ph, pm, pl;
                        split3(fmaxf(v, 0.f), ph, pm, pl);
                        const size_t o = tiled_off(m, n, g.N >> 5, 0);
                        *reinterpret_cast<__bf16*>(g.Cp + o) = ph;
                        *reinterpret_cast<__bf16*>(g.Cp + o + PLANE_BYTES) = pm;
                        *reinterpret_cast<__bf16*>(g.Cp + o + 2 * PLANE_BYTES) = pl;
                    } else {
                        g.C[(size_t)m * g.N + n] = v;
                    }
                }
            }
        }
    }
}

template <bool OUT_PLANES>
int launch_split_gemm(SplitGemmArgs g, int kid, hipStream_t s) {
    TimedSection ts(kid, s);
    static bool attr_set[2] = {false, false};
    if (!attr_set[OUT_PLANES]) {
        MDNO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_bf16_kernel<OUT_PLANES>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set[OUT_PLANES] = true;
    }
    g.tiles_n = g.N / TN;
    g.tiles_m = g.rows / TM;
    hipLaunchKernelGGL(gemm_split_bf16_kernel<OUT_PLANES>, dim3(g.tiles_n * g.tiles_m), dim3(256), LDS_BYTES, s, g);
    return check_launch("split-bf16 GEMM");
}

}  // namespace

bool edge_mlp_split_supported(int ker_width, int out_dim) {
    return ker_width % TK == 0 && ker_width % TN == 0 && out_dim % TN == 0;
}

size_t edge_mlp_split_workspace_bytes(int ker_width, int out_dim, long long chunk) {
    Carver cv(nullptr);
    cv.take<__bf16>(3 * (size_t)chunk * ker_width);
    cv.take<__bf16>(3 * (size_t)chunk * ker_width);
    cv.take<__bf16>(3 * (size_t)ker_width * ker_width);
    cv.take<__bf16>(3 * (size_t)out_dim * ker_width);
    return cv.used();
}

int edge_mlp_split(const float* frames, int frame, const int* t_dev, int rows_per_frame, const int* src,
                   const int* dst, const float* edge_attr, const int* perm, const int* num_edges,
                   long long edge_cap, long long chunk, int ker_in, int ker_width, int out_dim,
                   const EdgeMlpWeights& w, float* w_e, void* workspace, hipStream_t s) {
    MDNO_REQUIRE(ker_in > 0 && ker_in <= MAX_F, MDNO_EUNSUPPORTED, "edge_mlp: ker_in=%d (1..%d)", ker_in, MAX_F);
    MDNO_REQUIRE(((reinterpret_cast<uintptr_t>(w.w1) | reinterpret_cast<uintptr_t>(w.w2)) & 15) == 0, MDNO_EINVAL,
                 "edge_mlp: weight pointers must be 16-byte aligned");
    const int k = ker_width;
    Carver cv(workspace);
    unsigned char* h1p = reinterpret_cast<unsigned char*>(cv.take<__bf16>(3 * (size_t)chunk * k));
    unsigned char* h2p = reinterpret_cast<unsigned char*>(cv.take<__bf16>(3 * (size_t)chunk * k));
    unsigned char* w1p = reinterpret_cast<unsigned char*>(cv.take<__bf16>(3 * (size_t)k * k));
    unsigned char* w2p = reinterpret_cast<unsigned char*>(cv.take<__bf16>(3 * (size_t)out_dim * k));
    {
        TimedSection ts(KID_EDGE_L0, s);
        const long long c1 = (long long)k * (k / 8), c2 = (long long)out_dim * (k / 8);
        hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((c1 + 255) / 256)), dim3(256), 0, s, w.w1, k, k, w1p);
        hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((c2 + 255) / 256)), dim3(256), 0, s, w.w2, out_dim, k,
                           w2p);
    }
    MDNO_TRY(check_launch("split_planes_kernel"));
    const float* pos_mode = edge_attr ? nullptr : frames;
    for (long long e0 = 0; e0 < edge_cap; e0 += chunk) {
        const int cnt = (int)((edge_cap - e0) < chunk ? (edge_cap - e0) : chunk);
        {
            TimedSection ts(KID_EDGE_L0, s);
            hipLaunchKernelGGL(edge_l0_split_kernel, dim3((cnt + EB - 1) / EB), dim3(256), 0, s, pos_mode, frame, t_dev,
                               rows_per_frame, src, dst, edge_attr, perm, num_edges, e0, cnt, ker_in, k, w.w0, w.b0,
                               h1p);
        }
        MDNO_TRY(check_launch("edge_l0_split_kernel"));
        SplitGemmArgs g1{h1p, w1p, w.b1, nullptr, h2p, num_edges, e0, (int)chunk, k, k, 0, 0};
        MDNO_TRY(launch_split_gemm<true>(g1, KID_GEMM_L1, s));
        SplitGemmArgs g2{h2p, w2p, w.b2, w_e + (size_t)e0 * out_dim, nullptr, num_edges, e0, (int)chunk, out_dim, k,
                         0, 0};
        MDNO_TRY(launch_split_gemm<false>(g2, KID_GEMM_L2, s));
    }
    return MDNO_OK;
}

}  // namespace mdno
