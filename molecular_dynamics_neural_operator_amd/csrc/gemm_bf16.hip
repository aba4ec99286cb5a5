// bf16 GEMMs of the training path (BASELINE configs[3]; reference: graph_kernel.py:445-474 — the products
// autograd runs for DenseNet's three Linear layers, forward and backward), on the gfx950 matrix pipe.
//
//   gemm_nt_pp   C = act(A . W^T + b)   A bf16 [rows,K] row-major, W bf16 [N,K] row-major -> bf16 or fp32
//
// Structure ("ping-pong", one 256 x 256 output tile per 8-wave workgroup, one workgroup per CU):
//   * waves 0-3 own rows 0-127 of the tile, waves 4-7 rows 128-255; wave tile 128 x 64 = 4 x 2
//     v_mfma_f32_32x32x16_bf16 tiles (128 accumulator registers), so a 16-k step is 8 MFMAs fed by
//     6 fragment reads — 0.75 ds_read_b128 per MFMA, against 1.0 for a 64 x 64 wave tile;
//   * a 256 x 256 tile needs 16 KiB of A and 16 KiB of B per 32 k: 256 B of operand per MFMA, half of
//     what a 256 x 128 tile pulls from L2 for a single-product (not plane-split) GEMM — that, not the
//     matrix pipe, is what a 128-wide tile runs into;
//   * K advances in STAGES of 32 k.  A stage lives in one of three 32 KiB LDS slots, filled by LDS-DMA
//     (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPRs) two stages ahead of its first use;
//   * the two wave groups run the same program ONE PHASE APART: while waves 0-3 multiply stage t out of
//     registers (16 MFMAs back to back, nothing else in the stream), waves 4-7 read their fragments of
//     stage t from LDS and issue their share of the DMA of stage t+2, and vice versa.  The two waves of a
//     SIMD are always one of each group (waves w and w+4 share a SIMD), so the matrix pipe of every SIMD
//     sees an MFMA phase at all times, and the LDS reads / DMA issue of one wave sit beside the MFMAs of
//     the other instead of in front of its own.  Phases are separated by raw s_barrier instructions
//     (never __syncthreads(): its fence drains vmcnt and with it the DMA in flight) and a counted
//     s_waitcnt vmcnt(4) — a wave's four pieces of the newest stage stay in flight across the barrier.
//
// LDS image of a stage: rows of 32 k = 64 B; the four 16-B chunks of a row are XOR-swizzled with
// (row >> 2) & 3, which makes the ds_read_b128 fragment reads (lane = row, 16 B at k-half h) conflict
// free.  LDS-DMA writes lane-linearly, so the swizzle is applied to the per-lane SOURCE address: a piece
// (64 lanes x 16 B) covers 16 rows, lane l lands in LDS slot (row l>>2, chunk l&3) and fetches global
// chunk (l&3) ^ ((l>>4)&3) of that row.
//
// Ordering of LDS-DMA writes against fragment reads (MI355X_MICROARCH.md: nothing orders a ds_read behind
// a pending LDS-DMA but the issuing wave's vmcnt plus a barrier the reader has passed): a wave waits for
// its pieces of stage t+1 at the END of its load phase of stage t, before that phase's barrier; the first
// read of stage t+1 by anybody is at least one barrier later.  Slot reuse: stage t+2 goes to the slot of
// stage t-1, whose last reads (waves 4-7, load phase t-1) completed — lgkmcnt(0) — before the barrier in
// front of the earliest DMA issue into it.
#include "kernels.h"

namespace mdno {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((address_space(1))) const unsigned char glb_u8;

constexpr int PP_T = 256;                       // tile is PP_T x PP_T
constexpr int PP_BK = 32;                       // k per stage
constexpr int PP_ROW_BYTES = PP_BK * 2;         // 64 B
constexpr int PP_OPERAND_BYTES = PP_T * PP_ROW_BYTES;     // 16 KiB
constexpr int PP_STAGE_BYTES = 2 * PP_OPERAND_BYTES;      // 32 KiB
constexpr int PP_RING = 3;
constexpr int PP_LDS_BYTES = PP_RING * PP_STAGE_BYTES;    // 96 KiB
constexpr int PP_PIECES_PER_WAVE = PP_STAGE_BYTES / 1024 / 8;   // 4

struct NtArgs {
    const __bf16* A;      // [rows, K]
    const __bf16* W;      // [N, K]
    const float* bias;    // [N] or null
    void* C;              // [rows, N] bf16 or fp32
    long long rows;
    int N, K;
    int tiles_n;
    long long tiles_m;
};

template <bool RELU, bool OUT_BF16>
__global__ __launch_bounds__(512, 2) void gemm_nt_pp_kernel(NtArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // XCD-aware tile order (blocks b, b+8, ... share an XCD): each XCD gets a contiguous range of tiles, n
    // fastest, so that the tiles sharing an A row panel run on one L2.  Bijective for any count.
    const long long nwg = g.tiles_m * g.tiles_n;
    const long long orig = blockIdx.x;
    const long long xcd = orig & 7, q = nwg >> 3, r8 = nwg & 7;
    const long long tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
    const long long bm = (tile / g.tiles_n) * PP_T;
    const int bn = (int)(tile % g.tiles_n) * PP_T;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                  // 0: waves 0-3 (rows 0-127), 1: waves 4-7 (rows 128-255)
    const int wn = wave & 3;                    // 64-column strip
    const int l31 = lane & 31, h = lane >> 5;

    // ---- LDS-DMA: wave w moves pieces w, w+8 (A rows 16w.., 128+16w..) and w+16, w+24 (B rows likewise)
    const int pr = lane >> 2;                                   // row inside a piece
    const int pc = (lane & 3) ^ ((lane >> 4) & 3);              // global chunk fetched into LDS chunk lane&3
    const size_t ldk = (size_t)g.K * 2;                         // bytes per operand row
    const unsigned char* psrc[PP_PIECES_PER_WAVE];
#pragma unroll
    for (int t = 0; t < PP_PIECES_PER_WAVE; ++t) {
        const int piece = wave + 8 * t;                          // 0..31; < 16: A, else B
        const int row = (piece & 15) * 16 + pr;
        if (piece < 16) {
            long long ar = bm + row;
            if (ar >= g.rows) ar = g.rows - 1;                   // past the end: re-read the last row (never stored)
            psrc[t] = reinterpret_cast<const unsigned char*>(g.A) + (size_t)ar * ldk + pc * 16;
        } else {
            psrc[t] = reinterpret_cast<const unsigned char*>(g.W) + (size_t)(bn + row) * ldk + pc * 16;
        }
    }
    // LDS offset of piece t inside a slot: A pieces at piece*1024, B pieces behind the A operand
    auto piece_off = [&](int t) { return (wave + 8 * t) * 1024; };      // (pieces 16.. are B: 16 KiB + ...: same formula)
#define MDNO_PP_DMA(ST)                                                                                         \
    {                                                                                                           \
        const int st_ = (ST), slot_ = st_ % PP_RING;       /* (ST may name the caller's loop variable) */       \
        _Pragma("unroll") for (int pi_ = 0; pi_ < PP_PIECES_PER_WAVE; ++pi_)                                    \
            __builtin_amdgcn_global_load_lds((glb_u8*)(psrc[pi_] + (size_t)st_ * PP_ROW_BYTES),                \
                                             (lds_u8*)(lds + slot_ * PP_STAGE_BYTES + piece_off(pi_)), 16, 0, 0); \
    }

    // ---- fragment read offsets (row-swizzled 16-B chunks)
    int a_off[4], b_off[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = grp * 128 + i * 32 + l31;
        a_off[i] = row * PP_ROW_BYTES;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = wn * 64 + j * 32 + l31;
        b_off[j] = PP_OPERAND_BYTES + row * PP_ROW_BYTES;
    }
    const int sw = (l31 >> 2) & 3;              // (row >> 2) & 3: every row above is l31 plus a multiple of 32
    const int c0 = ((0 + h) ^ sw) << 4, c1 = ((2 + h) ^ sw) << 4;     // k-step 0: chunks h, k-step 1: chunks 2 + h

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float bv0 = 0.f, bv1 = 0.f;      // fetched before the K loop and pinned (edge_mlp_split.hip, epilogue stores)
    if (g.bias) {
        bv0 = g.bias[bn + wn * 64 + l31];
        bv1 = g.bias[bn + wn * 64 + 32 + l31];
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bv0), "+v"(bv1));      // the counted waits below must see DMA pieces only

    const int T = g.K / PP_BK;
    bf16x8 fa[2][4], fb[2][2];
#define MDNO_PP_LOAD(ST)                                                                          \
    {                                                                                             \
        const unsigned char* sb_ = lds + ((ST) % PP_RING) * PP_STAGE_BYTES;                       \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
            fa[0][i] = *reinterpret_cast<const bf16x8*>(sb_ + a_off[i] + c0);                     \
            fa[1][i] = *reinterpret_cast<const bf16x8*>(sb_ + a_off[i] + c1);                     \
        }                                                                                         \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                           \
            fb[0][j] = *reinterpret_cast<const bf16x8*>(sb_ + b_off[j] + c0);                     \
            fb[1][j] = *reinterpret_cast<const bf16x8*>(sb_ + b_off[j] + c1);                     \
        }                                                                                         \
    }
#define MDNO_PP_MMA()                                                                             \
    {                                                                                             \
        __builtin_amdgcn_s_setprio(1);                                                            \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                          \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                         \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                     \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                            \
    }
#define MDNO_PP_BARRIER()                            \
    __builtin_amdgcn_sched_barrier(0);               \
    __builtin_amdgcn_s_barrier();                    \
    asm volatile("" ::: "memory");                   \
    __builtin_amdgcn_sched_barrier(0);

    // prologue: stages 0 and 1 in flight, stage 0 landed for everybody
    MDNO_PP_DMA(0)
    if (T > 1) { MDNO_PP_DMA(1) }
    if (T > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MDNO_PP_BARRIER()
    if (grp == 1) { MDNO_PP_BARRIER() }          // the stagger: waves 4-7 run one phase behind
    for (int t = 0; t < T; ++t) {
        // ---- load phase of stage t (the other group multiplies meanwhile)
        MDNO_PP_LOAD(t)
        if (t + 2 < T) {
            MDNO_PP_DMA(t + 2)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // my pieces of stage t+1 have landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // my fragments are in registers (slot t is free of me)
        MDNO_PP_BARRIER()
        // ---- multiply phase of stage t
        MDNO_PP_MMA()
        // (the last barrier of waves 4-7 — the one the stagger added to their count — is left out: nothing follows
        // it but the epilogue, which waves 0-3 then start while waves 4-7 are still multiplying)
        if (grp == 0 || t + 1 < T) { MDNO_PP_BARRIER() }
    }
#undef MDNO_PP_DMA
#undef MDNO_PP_LOAD
#undef MDNO_PP_MMA
#undef MDNO_PP_BARRIER

    // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5): a lane holds
    // ONE column of 16 rows, so storing from the accumulators is 2-byte (or 4-byte) pieces 2*N apart — 128 store
    // instructions per lane and 16.7 us per tile, a third of the K = 1024 products.  Instead every wave turns its
    // tile through a private LDS patch (the ring is idle now: 8 KiB per wave), 32 rows at a time: bias / ReLU /
    // rounding on the way in (one element per ds_write), whole rows on the way out — 16 B per lane, a 128-B line
    // per 8 (bf16) or 4 (fp32) lanes.  LDS operations of one wave execute in order: no wait between the two.
    constexpr int ESZ = OUT_BF16 ? 2 : 4;
    constexpr int PATCH_ROW = 64 * ESZ;                    // bytes per patch row (the wave's 64 columns)
    unsigned char* patch = lds + wave * (32 * 64 * 4);    // 8 KiB apart (fp32 size) for either type
    const size_t ldc = (size_t)g.N * ESZ;
    unsigned char* cbase = static_cast<unsigned char*>(g.C) + (size_t)(bn + wn * 64) * ESZ;
    constexpr int LANES_PER_ROW = PATCH_ROW / 16;          // 8 or 16
    constexpr int ROWS_PER_INSTR = 64 / LANES_PER_ROW;     // 8 or 4
    const int orow = lane / LANES_PER_ROW, ochunk = lane % LANES_PER_ROW;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float bv = j ? bv1 : bv0;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = (e & 3) + 8 * (e >> 2) + 4 * h;
                float v = acc[i][j][e] + bv;
                if (RELU) v = fmaxf(v, 0.f);
                unsigned char* dst = patch + r * PATCH_ROW + (j * 32 + l31) * ESZ;
                if (OUT_BF16) *reinterpret_cast<__bf16*>(dst) = (__bf16)v;
                else *reinterpret_cast<float*>(dst) = v;
            }
        }
        const long long m0 = bm + grp * 128 + i * 32;
#pragma unroll
        for (int rr = 0; rr < 32; rr += ROWS_PER_INSTR) {
            const uint4 v = *reinterpret_cast<const uint4*>(patch + (rr + orow) * PATCH_ROW + ochunk * 16);
            const long long m = m0 + rr + orow;
            if (m < g.rows) *reinterpret_cast<uint4*>(cbase + (size_t)m * ldc + ochunk * 16) = v;
        }
    }
}

}  // namespace

bool gemm_nt_pp_supported(long long rows, int N, int K) {
    return rows > 0 && N % PP_T == 0 && K % PP_BK == 0 && K >= 2 * PP_BK;
}

// C = act(A . W^T + b): A bf16 [rows,K], W bf16 [N,K] (both row-major), C bf16 or fp32 [rows,N]
int gemm_nt_pp(const void* A, const void* W, const float* bias, long long rows, int N, int K, int relu, int out_bf16,
               void* C, hipStream_t s) {
    MDNO_REQUIRE(gemm_nt_pp_supported(rows, N, K), MDNO_EUNSUPPORTED, "gemm_nt_pp: rows=%lld N=%d K=%d", rows, N, K);
    NtArgs g{static_cast<const __bf16*>(A), static_cast<const __bf16*>(W), bias, C, rows, N, K, N / PP_T,
             (rows + PP_T - 1) / PP_T};
    const long long nwg = g.tiles_m * g.tiles_n;
    MDNO_REQUIRE(nwg < (1ll << 31), MDNO_EUNSUPPORTED, "gemm_nt_pp: too many tiles");
    static std::atomic<unsigned long long> raised[4] = {};
#define MDNO_GO(R, O, IDX)                                                                                         \
    {                                                                                                              \
        MDNO_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(&gemm_nt_pp_kernel<R, O>), PP_LDS_BYTES, raised[IDX])); \
        hipLaunchKernelGGL((gemm_nt_pp_kernel<R, O>), dim3((unsigned)nwg), dim3(512), PP_LDS_BYTES, s, g);          \
    }
    if (relu) { if (out_bf16) MDNO_GO(true, true, 0) else MDNO_GO(true, false, 1) }
    else      { if (out_bf16) MDNO_GO(false, true, 2) else MDNO_GO(false, false, 3) }
#undef MDNO_GO
    return check_launch("gemm_nt_pp_kernel");
}

}  // namespace mdno
