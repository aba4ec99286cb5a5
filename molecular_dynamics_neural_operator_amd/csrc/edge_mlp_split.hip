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
// Data flow (all planes are bf16, plane-major [3][rows][K]):
//   weights      fp32 [N,K] --split_planes_kernel--> Bp[3][N][K]          (once per forward, 40 MB)
//   layer 0      edge attrs -> relu(linear)  --split--> H1p[3][chunk][k]   (edge_l0_split_kernel)
//   layer 1      H1p x W1p -> relu -> split  -----> H2p[3][chunk][k]       (epilogue emits planes)
//   layer 2      H2p x W2p + b  -> fp32 W_e[E, Cin*Cout]
// so no fp32 activation is ever stored.
//
// GEMM kernel: 128x128x32 block tile, 4 waves (2x2), wave tile 64x64 = 2x2 v_mfma_f32_32x32x16_bf16
// tiles, 48 MFMAs per K-tile per wave.  Each staged operand fragment feeds 2-3 of the six products,
// so LDS and global traffic per MFMA are half those of an ordinary bf16 GEMM.  Staging is
// global -> registers -> LDS with the next tile's loads in flight during the MFMAs; LDS rows are
// 64 B (32 bf16) with the 16-B chunk position XOR-swizzled by (row>>2)&3, which makes the
// ds_read_b128 fragment reads bank-conflict-free.  Workgroups are numbered so that each XCD owns a
// contiguous range of tiles (neighbouring tiles share the A row-panel through that XCD's L2).
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

// ---------------------------------------------------------------- fp32 [rows,cols] -> 3 bf16 planes
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ w, long long count,
                                                           __bf16* __restrict__ planes) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= count) return;
    const float4 v = *reinterpret_cast<const float4*>(w + i);
    const float x[4] = {v.x, v.y, v.z, v.w};
    __bf16 o[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split3(x[j], o[0][j], o[1][j], o[2][j]);
#pragma unroll
    for (int p = 0; p < 3; ++p)
        *reinterpret_cast<uint2*>(planes + (size_t)p * count + i) = *reinterpret_cast<const uint2*>(o[p]);
}

// ---------------------------------------------------------------- layer 0 (+ attr gather) -> planes
constexpr int EB = 16;
constexpr int MAX_F = 8;

__global__ __launch_bounds__(256) void edge_l0_split_kernel(
    const float* __restrict__ frames, int frame, const int* __restrict__ t_dev, int rows_per_frame,
    const int* __restrict__ src, const int* __restrict__ dst, const float* __restrict__ edge_attr,
    const int* __restrict__ perm, const int* __restrict__ num_edges, long long e_begin, int e_count, int F, int k,
    const float* __restrict__ w0, const float* __restrict__ b0, __bf16* __restrict__ hp, long long plane_stride) {
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
            s = fmaxf(s + bc, 0.f);
            __bf16 h, m, l;
            split3(s, h, m, l);
            const size_t o = (size_t)(e - e_begin) * k + c;
            hp[o] = h;
            hp[o + plane_stride] = m;
            hp[o + 2 * plane_stride] = l;
        }
    }
}

// ---------------------------------------------------------------- split-bf16 GEMM
struct SplitGemmArgs {
    const __bf16* Ap;        // [3][a_rows][K]  (a_rows = chunk capacity)
    const __bf16* Bp;        // [3][N][K]
    const float* bias;       // [N]
    float* C;                // fp32 [rows][N]            (OUT_PLANES = false)
    __bf16* Cp;              // [3][a_rows][N] planes     (OUT_PLANES = true, ReLU applied)
    const int* num_edges;
    long long row_begin;
    long long a_plane_stride;  // a_rows * K
    long long b_plane_stride;  // N * K
    long long c_plane_stride;  // a_rows * N
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

    // ---- staging map: plane tile = 128 rows x 4 chunks(16 B); thread covers chunks tid and tid+256
    const int srow0 = tid >> 2, sc = tid & 3;                  // rows srow0 and srow0+64
    const size_t ldk = (size_t)g.K;
    const __bf16* a_src = g.Ap + (size_t)(bm + srow0) * ldk + sc * 8;
    const __bf16* b_src = g.Bp + (size_t)(bn + srow0) * ldk + sc * 8;
    const int sw0 = (sc ^ ((srow0 >> 2) & 3)) * 16;            // (row+64)>>2 has the same low 2 bits
    const int st_off = srow0 * 64 + sw0;                       // second row: + 64*64 B
    // named registers only: arrays here end up in scratch and serialise the loads
    uint4 a00, a01, a10, a11, a20, a21, b00, b01, b10, b11, b20, b21;
    const __bf16* a_src1 = a_src + g.a_plane_stride;
    const __bf16* a_src2 = a_src + 2 * g.a_plane_stride;
    const __bf16* b_src1 = b_src + g.b_plane_stride;
    const __bf16* b_src2 = b_src + 2 * g.b_plane_stride;
    const size_t row64 = 64 * ldk;
#define MDNO_LD16(P) (*reinterpret_cast<const uint4*>(P))
#define MDNO_SPLIT_LOAD(KOFF)                                                      \
    a00 = MDNO_LD16(a_src + (KOFF));  a01 = MDNO_LD16(a_src + row64 + (KOFF));      \
    a10 = MDNO_LD16(a_src1 + (KOFF)); a11 = MDNO_LD16(a_src1 + row64 + (KOFF));     \
    a20 = MDNO_LD16(a_src2 + (KOFF)); a21 = MDNO_LD16(a_src2 + row64 + (KOFF));     \
    b00 = MDNO_LD16(b_src + (KOFF));  b01 = MDNO_LD16(b_src + row64 + (KOFF));      \
    b10 = MDNO_LD16(b_src1 + (KOFF)); b11 = MDNO_LD16(b_src1 + row64 + (KOFF));     \
    b20 = MDNO_LD16(b_src2 + (KOFF)); b21 = MDNO_LD16(b_src2 + row64 + (KOFF));
#define MDNO_ST16(OFF, V) *reinterpret_cast<uint4*>(lds + (OFF)) = (V)
#define MDNO_SPLIT_STORE()                                                                                  \
    MDNO_ST16(st_off, a00);                    MDNO_ST16(st_off + 4096, a01);                                \
    MDNO_ST16(PLANE_BYTES + st_off, a10);      MDNO_ST16(PLANE_BYTES + st_off + 4096, a11);                  \
    MDNO_ST16(2 * PLANE_BYTES + st_off, a20);  MDNO_ST16(2 * PLANE_BYTES + st_off + 4096, a21);              \
    MDNO_ST16(3 * PLANE_BYTES + st_off, b00);  MDNO_ST16(3 * PLANE_BYTES + st_off + 4096, b01);              \
    MDNO_ST16(4 * PLANE_BYTES + st_off, b10);  MDNO_ST16(4 * PLANE_BYTES + st_off + 4096, b11);              \
    MDNO_ST16(5 * PLANE_BYTES + st_off, b20);  MDNO_ST16(5 * PLANE_BYTES + st_off + 4096, b21);

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

    const int nk = g.K / TK;
    MDNO_SPLIT_LOAD(0)
    MDNO_SPLIT_STORE()
    __syncthreads();
    // steady state: next tile's global loads are in flight while this tile is multiplied
    for (int kt = 0; kt < nk - 1; ++kt) {
        MDNO_SPLIT_LOAD((size_t)(kt + 1) * TK)
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
                        const size_t o = (size_t)m * g.N + n;
                        g.Cp[o] = ph;
                        g.Cp[o + g.c_plane_stride] = pm;
                        g.Cp[o + 2 * g.c_plane_stride] = pl;
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
    __bf16* h1p = cv.take<__bf16>(3 * (size_t)chunk * k);
    __bf16* h2p = cv.take<__bf16>(3 * (size_t)chunk * k);
    __bf16* w1p = cv.take<__bf16>(3 * (size_t)k * k);
    __bf16* w2p = cv.take<__bf16>(3 * (size_t)out_dim * k);
    {
        TimedSection ts(KID_EDGE_L0, s);
        const long long c1 = (long long)k * k, c2 = (long long)out_dim * k;
        hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((c1 / 4 + 255) / 256)), dim3(256), 0, s, w.w1, c1, w1p);
        hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((c2 / 4 + 255) / 256)), dim3(256), 0, s, w.w2, c2, w2p);
    }
    MDNO_TRY(check_launch("split_planes_kernel"));
    const float* pos_mode = edge_attr ? nullptr : frames;
    const long long hstride = chunk * k;
    for (long long e0 = 0; e0 < edge_cap; e0 += chunk) {
        const int cnt = (int)((edge_cap - e0) < chunk ? (edge_cap - e0) : chunk);
        {
            TimedSection ts(KID_EDGE_L0, s);
            hipLaunchKernelGGL(edge_l0_split_kernel, dim3((cnt + EB - 1) / EB), dim3(256), 0, s, pos_mode, frame, t_dev,
                               rows_per_frame, src, dst, edge_attr, perm, num_edges, e0, cnt, ker_in, k, w.w0, w.b0,
                               h1p, hstride);
        }
        MDNO_TRY(check_launch("edge_l0_split_kernel"));
        SplitGemmArgs g1{h1p, w1p, w.b1, nullptr, h2p, num_edges, e0, hstride, (long long)k * k, hstride,
                         (int)chunk, k, k, 0, 0};
        MDNO_TRY(launch_split_gemm<true>(g1, KID_GEMM_L1, s));
        SplitGemmArgs g2{h2p, w2p, w.b2, w_e + (size_t)e0 * out_dim, nullptr, num_edges, e0, hstride,
                         (long long)out_dim * k, 0, (int)chunk, out_dim, k, 0, 0};
        MDNO_TRY(launch_split_gemm<false>(g2, KID_GEMM_L2, s));
    }
    return MDNO_OK;
}

}  // namespace mdno
