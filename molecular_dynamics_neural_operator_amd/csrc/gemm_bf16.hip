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
#include "split_layout.h"

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

// EPI_BIAS: C = act(acc + bias) as bf16 or fp32.  EPI_MASK: C = bf16(Y > 0 ? acc : 0) — the ReLU backward of the
// layer whose stored output is Y, fused (graph_kernel.py:239-242 differentiated: d pre = d h * (h > 0)).
// EPI_SLAB: fp32 partial products of a K slice into slab z (the A^T.B products; added in slice order afterwards).
enum { EPI_BIAS = 0, EPI_MASK = 1, EPI_SLAB = 2 };

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
// one 32x32x16 MFMA on 16-bit fragments held as bf16x8 registers: bf16, or (F16) the same bits read as fp16
template <bool F16>
__device__ __forceinline__ f32x16 mma16(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

struct PpArgs {
    const __bf16* A;      // NT: [rows, K] row-major.   TN: [rows, n1]
    const __bf16* W;      // NT: [N, K] row-major.      TN: [rows, n2]
    const float* bias;    // EPI_BIAS: [N] or null
    const __bf16* Y;      // EPI_MASK: [rows, N]
    void* C;              // NT: [rows, N] bf16 or fp32.  TN: slabs [slices][n1][n2] fp32
    long long rows;       // NT: output rows.  TN: contraction length
    int N, K;             // NT: output columns, contraction length.  TN: n2, n1
    int tiles_n;
    long long tiles_m;
    long long slice_rows; // TN: rows per K slice (multiple of 32)
    // F16 (TN only): the operands are fp16 planes and blockIdx.z picks one of the three plane products of
    // x = hi + 2^-11 lo' — z = 0: A . W, 1: A2 . W, 2: A . W2 — each into its own set of slabs
    const __bf16* A2 = nullptr;
    const __bf16* W2 = nullptr;
};

// TN = false: C = A . W^T (contraction index fastest in both operands: LDS rows of 32 k, ds_read_b128 fragments).
// TN = true:  C = A^T . B over rows (contraction index SLOWEST in both operands): a stage is 32 rows of 256
//   columns per operand, copied as it lies — LDS rows of 512 B, the 16-B chunks XOR-swizzled by ((row & 3) << 2) —
//   and the fragments come out of gfx950's transposing read ds_read_b64_tr_b16 (a 16-lane group gets a 4-row x
//   16-column block column-major: 4 consecutive k of its own m; two reads make the 8-k MFMA operand).  The four
//   rows of a block sit in four different 64-B quarters of the 256-B bank row: conflict free.
// PERSIST (NT with bf16 output only): the grid is one workgroup per CU (a multiple of 8) and a workgroup walks the
// tiles of its XCD's range j, j + G/8, j + 2 G/8, ..; the first two stages of the NEXT tile are put in flight by every
// wave right after its last multiply phase, under its epilogue stores (the epilogue's LDS patches live in ring slot 2,
// which the next tile does not touch before the barrier that ends its prologue), so that a tile no longer starts with
// every CU's first 64 KB arriving at once (EXPERIMENTS.md §0.3: 6.7k of a K = 1,024 tile's 64.5k cycles).  Same MFMAs
// in the same order per output element: bit-identical results.
template <bool TN, int EPI, bool RELU, bool OUT_BF16, bool F16 = false, bool PERSIST = false>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(PpArgs g) {
    static_assert(!PERSIST || (!TN && OUT_BF16 && EPI != EPI_SLAB && !F16), "persistent form: A . W^T with bf16 output");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const __bf16* const opA = (F16 && blockIdx.z == 1) ? g.A2 : g.A;
    const __bf16* const opW = (F16 && blockIdx.z == 2) ? g.W2 : g.W;
    // XCD-aware tile order (blocks b, b+8, ... share an XCD): each XCD gets a contiguous range of tiles, n
    // fastest, so that the tiles sharing an A row panel run on one L2.  Bijective for any count.
    const long long nwg = g.tiles_m * g.tiles_n;
    const long long orig = blockIdx.x;
    const long long xcd = orig & 7, q = nwg >> 3, r8 = nwg & 7;
    const long long xcd_first = xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;      // this XCD's first tile
    const long long xcd_count = q + (xcd < r8 ? 1 : 0);
    const long long per_xcd = PERSIST ? (long long)(gridDim.x >> 3) : 0;      // workgroups per XCD (stride of the walk)
    long long loc = orig >> 3;                          // index inside the XCD's range
    if (PERSIST && loc >= xcd_count) return;
    long long tile = xcd_first + loc;
    long long bm = (tile / g.tiles_n) * PP_T;
    int bn = (int)(tile % g.tiles_n) * PP_T;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                  // 0: waves 0-3 (rows 0-127), 1: waves 4-7 (rows 128-255)
    const int wn = wave & 3;                    // 64-column strip
    const int l31 = lane & 31, h = lane >> 5;

    // ---- contraction range and stage count
    long long k0 = 0, k1 = g.K;                 // NT: k; TN: rows of this slice
    if (TN) {
        k0 = (long long)blockIdx.y * g.slice_rows;
        k1 = k0 + g.slice_rows;
        if (k1 > g.rows) k1 = g.rows;
    }
    const int T = (int)((k1 - k0 + PP_BK - 1) / PP_BK);

    // ---- LDS-DMA sources.  Wave w moves pieces w, w+8 (operand A) and w+16, w+24 (operand B) of every stage.
    //   NT: a piece is 16 rows x 64 B; lane l -> row l>>2, LDS chunk l&3 <- global chunk (l&3) ^ ((l>>4)&3)
    //   TN: a piece is 2 rows x 512 B; lane l -> row l>>5, LDS chunk l&31 <- global chunk (l&31) ^ ((row&3)<<2)
    // Issued as buffer loads (buffer_load_dwordx4 ... offen lds): one descriptor per operand whose base is this
    // tile's (and, TN, this K slice's) first byte, a per-lane 32-bit offset that never changes, and the stage as
    // the scalar offset — no 64-bit address arithmetic per piece, and whatever lies past the descriptor's size
    // reads as ZERO: the rows of A past the end of the matrix (NT; their outputs are never stored) and the rows
    // of the last stage past the end of the contraction (TN; they must contribute nothing).
    unsigned voff[PP_PIECES_PER_WAVE];
    unsigned stride_a, stride_b;                 // bytes from one stage to the next, per operand
    __amdgpu_buffer_rsrc_t rsrc_a, rsrc_b;
    // (NT) the two descriptors of the tile at (row bm_, column bn_)
    auto nt_descriptors = [&](long long bm_, int bn_) {
        const size_t ldk = (size_t)g.K * 2;
        long long live = g.rows - bm_;
        if (live > PP_T) live = PP_T;
        rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const unsigned char*>(g.A) + (size_t)bm_ * ldk), 0,
                                                   (int)(live * (long long)ldk), 0x00020000);
        rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const unsigned char*>(g.W) + (size_t)bn_ * ldk), 0,
                                                   (int)(PP_T * ldk), 0x00020000);
    };
    if (!TN) {
        const int pr = lane >> 2, pc = (lane & 3) ^ ((lane >> 4) & 3);
        const size_t ldk = (size_t)g.K * 2;
        stride_a = stride_b = PP_ROW_BYTES;
        nt_descriptors(bm, bn);
#pragma unroll
        for (int t = 0; t < PP_PIECES_PER_WAVE; ++t) {
            const int piece = wave + 8 * t;                      // 0..31; < 16: A, else B
            voff[t] = (unsigned)(((piece & 15) * 16 + pr) * ldk + pc * 16);
        }
    } else {
        stride_a = (unsigned)((size_t)PP_BK * g.K * 2);
        stride_b = (unsigned)((size_t)PP_BK * g.N * 2);
        const long long live = k1 - k0;          // rows of this slice
        rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const unsigned char*>(opA) + ((size_t)k0 * g.K + bm) * 2), 0,
                                                   live > 0 ? (unsigned)(((live - 1) * g.K + PP_T) * 2) : 0, 0x00020000);
        rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const unsigned char*>(opW) + ((size_t)k0 * g.N + bn) * 2), 0,
                                                   live > 0 ? (unsigned)(((live - 1) * g.N + PP_T) * 2) : 0, 0x00020000);
#pragma unroll
        for (int t = 0; t < PP_PIECES_PER_WAVE; ++t) {
            const int piece = wave + 8 * t;
            const int row = (piece & 15) * 2 + (lane >> 5);      // 0..31
            const int pc = (lane & 31) ^ ((row & 3) << 2);
            voff[t] = (unsigned)((size_t)row * (piece < 16 ? g.K : g.N) * 2 + pc * 16);
        }
    }
    const int piece_base = wave * 1024;          // piece t of this wave sits at piece_base + t * 8 KiB inside a slot
#define MDNO_PP_DMA(ST)                                                                                         \
    {                                                                                                           \
        const int st_ = (ST), slot_ = st_ % PP_RING;       /* (ST may name the caller's loop variable) */       \
        _Pragma("unroll") for (int pi_ = 0; pi_ < PP_PIECES_PER_WAVE; ++pi_)                                    \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(pi_ < 2 ? rsrc_a : rsrc_b,                                 \
                                                     (lds_u8*)(lds + slot_ * PP_STAGE_BYTES + piece_base + pi_ * 8192), 16, \
                                                     voff[pi_], (unsigned)st_ * (pi_ < 2 ? stride_a : stride_b), 0, 0); \
    }

    // ---- fragment read offsets
    int a_off[4], b_off[2];
    int c0 = 0, c1 = 0;
    if (!TN) {       // rows of 64 B, 16-B chunks swizzled by (row >> 2) & 3; lane = (row l31, k-half h)
#pragma unroll
        for (int i = 0; i < 4; ++i) a_off[i] = (grp * 128 + i * 32 + l31) * PP_ROW_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j) b_off[j] = PP_OPERAND_BYTES + (wn * 64 + j * 32 + l31) * PP_ROW_BYTES;
        const int sw = (l31 >> 2) & 3;
        c0 = ((0 + h) ^ sw) << 4;
        c1 = ((2 + h) ^ sw) << 4;
    } else {         // rows of 512 B (k index), chunk ^= (row & 3) << 2; 16-lane group gq = lane>>4, lane = 4q+p in it
        const int gq = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
        const int rowb = (8 * (gq >> 1) + qq) * 512 + 8 * (pp & 1);          // + kk*16 rows + r*4 rows as immediates
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = (grp * 128 + i * 32 + 16 * (gq & 1)) / 8 + (pp >> 1);
            a_off[i] = rowb + ((ch ^ (qq << 2)) << 4);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ch = (wn * 64 + j * 32 + 16 * (gq & 1)) / 8 + (pp >> 1);
            b_off[j] = PP_OPERAND_BYTES + rowb + ((ch ^ (qq << 2)) << 4);
        }
    }

    f32x16 acc[4][2];
    float bv0 = 0.f, bv1 = 0.f;      // fetched before the K loop and pinned (edge_mlp_split.hip, epilogue stores)
    if (EPI == EPI_BIAS && g.bias) {
        bv0 = g.bias[bn + wn * 64 + l31];
        bv1 = g.bias[bn + wn * 64 + 32 + l31];
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bv0), "+v"(bv1));      // the counted waits below must see DMA pieces only

    bf16x8 fa[2][4], fb[2][2];
    // The transposing reads (k rows +0..3, +4..7 of both k-steps of a stage: four per operand block) as inline assembly:
    // behind an LDS-DMA the compiler puts a full vmcnt(0) in front of every ds_read_tr builtin (it cannot tell what the DMA
    // wrote from what the read reads), i.e. in front of each load phase — after the counted wait and the barrier that
    // already ordered them (EXPERIMENTS.md §0.2a).  The results are complete after the lgkmcnt(0) that ends the load phase.
    const unsigned lds_addr = (unsigned)(size_t)(lds_u8*)lds;
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    // (the waits sit inside the blocks: what the compiler does with the result registers after a block — the moves that
    // pair two 64-bit results into one operand — must find them complete)
#define MDNO_TR4(O, A) "ds_read_b64_tr_b16 %" #O ", %" #A "\n ds_read_b64_tr_b16 %" #O "+1, %" #A " offset:2048\n"
    auto tr_frags_a = [&](unsigned a0, unsigned a1, unsigned a2, unsigned a3, bf16x8 (&f)[2][4]) {
        i32x2 r[16];
        asm volatile("ds_read_b64_tr_b16 %0, %16\n ds_read_b64_tr_b16 %1, %16 offset:2048\n"
                     "ds_read_b64_tr_b16 %2, %16 offset:8192\n ds_read_b64_tr_b16 %3, %16 offset:10240\n"
                     "ds_read_b64_tr_b16 %4, %17\n ds_read_b64_tr_b16 %5, %17 offset:2048\n"
                     "ds_read_b64_tr_b16 %6, %17 offset:8192\n ds_read_b64_tr_b16 %7, %17 offset:10240\n"
                     "ds_read_b64_tr_b16 %8, %18\n ds_read_b64_tr_b16 %9, %18 offset:2048\n"
                     "ds_read_b64_tr_b16 %10, %18 offset:8192\n ds_read_b64_tr_b16 %11, %18 offset:10240\n"
                     "ds_read_b64_tr_b16 %12, %19\n ds_read_b64_tr_b16 %13, %19 offset:2048\n"
                     "ds_read_b64_tr_b16 %14, %19 offset:8192\n ds_read_b64_tr_b16 %15, %19 offset:10240\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]),
                       "=&v"(r[8]), "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15])
                     : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[0][i] = __builtin_bit_cast(bf16x8, (i32x4){r[4 * i].x, r[4 * i].y, r[4 * i + 1].x, r[4 * i + 1].y});
            f[1][i] = __builtin_bit_cast(bf16x8, (i32x4){r[4 * i + 2].x, r[4 * i + 2].y, r[4 * i + 3].x, r[4 * i + 3].y});
        }
    };
    auto tr_frags_b = [&](unsigned b0, unsigned b1, bf16x8 (&f)[2][2]) {
        i32x2 r[8];
        asm volatile("ds_read_b64_tr_b16 %0, %8\n ds_read_b64_tr_b16 %1, %8 offset:2048\n"
                     "ds_read_b64_tr_b16 %2, %8 offset:8192\n ds_read_b64_tr_b16 %3, %8 offset:10240\n"
                     "ds_read_b64_tr_b16 %4, %9\n ds_read_b64_tr_b16 %5, %9 offset:2048\n"
                     "ds_read_b64_tr_b16 %6, %9 offset:8192\n ds_read_b64_tr_b16 %7, %9 offset:10240\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                     : "v"(b0), "v"(b1) : "memory");
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f[0][j] = __builtin_bit_cast(bf16x8, (i32x4){r[4 * j].x, r[4 * j].y, r[4 * j + 1].x, r[4 * j + 1].y});
            f[1][j] = __builtin_bit_cast(bf16x8, (i32x4){r[4 * j + 2].x, r[4 * j + 2].y, r[4 * j + 3].x, r[4 * j + 3].y});
        }
    };
#undef MDNO_TR4
#define MDNO_PP_LOAD(ST)                                                                          \
    {                                                                                             \
        const unsigned char* sb_ = lds + ((ST) % PP_RING) * PP_STAGE_BYTES;                       \
        if (!TN) {                                                                                \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                       \
                fa[0][i] = *reinterpret_cast<const bf16x8*>(sb_ + a_off[i] + c0);                 \
                fa[1][i] = *reinterpret_cast<const bf16x8*>(sb_ + a_off[i] + c1);                 \
            }                                                                                     \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                       \
                fb[0][j] = *reinterpret_cast<const bf16x8*>(sb_ + b_off[j] + c0);                 \
                fb[1][j] = *reinterpret_cast<const bf16x8*>(sb_ + b_off[j] + c1);                 \
            }                                                                                     \
        } else {                                                                                  \
            const unsigned sa_ = lds_addr + ((ST) % PP_RING) * PP_STAGE_BYTES;                    \
            tr_frags_a(sa_ + a_off[0], sa_ + a_off[1], sa_ + a_off[2], sa_ + a_off[3], fa);       \
            tr_frags_b(sa_ + b_off[0], sa_ + b_off[1], fb);                                       \
        }                                                                                         \
    }
#define MDNO_PP_MMA()                                                                             \
    {                                                                                             \
        __builtin_amdgcn_s_setprio(1);                                                            \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                          \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                         \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                     \
                    acc[i][j] = mma16<F16>(fa[ks][i], fb[ks][j], acc[i][j]);                      \
        __builtin_amdgcn_s_setprio(0);                                                            \
    }
#define MDNO_PP_BARRIER()                            \
    __builtin_amdgcn_sched_barrier(0);               \
    __builtin_amdgcn_s_barrier();                    \
    asm volatile("" ::: "memory");                   \
    __builtin_amdgcn_sched_barrier(0);

    // prologue: stages 0 and 1 in flight, stage 0 landed for everybody
    if (T > 0) { MDNO_PP_DMA(0) }
    if (T > 1) { MDNO_PP_DMA(1) }
    bool first_tile = true;
tile_loop:      // (PERSIST: one pass per tile of this workgroup; otherwise a single pass)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // a later tile's first two stages went out under the previous epilogue, with that epilogue's stores behind them in
    // the queue: everything is waited for (stage 1 landed long ago: an epilogue is longer than a stage's flight)
    if (T > 1 && first_tile) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(bv0), "+v"(bv1) :: "memory");      // (and this tile's bias is pinned here)
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
    if (T == 0 && grp == 0) { MDNO_PP_BARRIER() }   // (an empty slice: pair the stagger barrier)

    // ---- PERSIST: this wave's share of the next tile's stages 0 and 1, now (every fragment read of the tile just
    // multiplied completed before the last barrier this wave passed: any slot may be overwritten; slot 2 is left to
    // the epilogue's patches).  The next tile's bias first — older in the queue than the pieces, pinned after its K loop.
    const long long cur_bm = bm;
    const int cur_bn = bn;
    float bn0 = 0.f, bn1 = 0.f;
    bool more_tiles = false;
    if (PERSIST) {
        loc += per_xcd;
        more_tiles = loc < xcd_count;
        if (more_tiles) {
            tile = xcd_first + loc;
            bm = (tile / g.tiles_n) * PP_T;
            bn = (int)(tile % g.tiles_n) * PP_T;
            if (EPI == EPI_BIAS && g.bias) {
                bn0 = g.bias[bn + wn * 64 + l31];
                bn1 = g.bias[bn + wn * 64 + 32 + l31];
            }
            nt_descriptors(bm, bn);
            __builtin_amdgcn_sched_barrier(0);
            MDNO_PP_DMA(0)
            if (T > 1) { MDNO_PP_DMA(1) }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5): a lane holds
    // ONE column of 16 rows, so storing from the accumulators is 2-byte (or 4-byte) pieces 2*N apart — 128 store
    // instructions per lane and 16.7 us per tile, a third of the K = 1024 products.  Instead every wave turns its
    // tile through a private LDS patch (the ring is idle now: 8 KiB per wave), 32 rows at a time: bias / ReLU /
    // rounding on the way in (one element per ds_write), whole rows on the way out — 16 B per lane, a 128-B line
    // per 8 (bf16) or 4 (fp32) lanes.  LDS operations of one wave execute in order: no wait between the two.
    constexpr bool OBF = OUT_BF16 && EPI != EPI_SLAB;
    constexpr int ESZ = OBF ? 2 : 4;
    constexpr int PATCH_ROW = 64 * ESZ;                    // bytes per patch row (the wave's 64 columns)
    // (PERSIST: bf16 patches of 4 KiB per wave in ring slot 2 — slots 0 and 1 are receiving the next tile)
    unsigned char* patch = PERSIST ? lds + 2 * PP_STAGE_BYTES + wave * (32 * 64 * 2) : lds + wave * (32 * 64 * 4);    // else 8 KiB apart (fp32 size) for either type
    const size_t ldc = (size_t)g.N * ESZ;
    unsigned char* cbase = static_cast<unsigned char*>(g.C) + (size_t)(cur_bn + wn * 64) * ESZ;
    // slab (product z, slice y) = [n1 = K][n2 = N] fp32
    if (EPI == EPI_SLAB) cbase += ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * (size_t)g.K * g.N * 4;
    const long long out_rows = TN ? (long long)g.K : g.rows;
    constexpr int LANES_PER_ROW = PATCH_ROW / 16;          // 8 or 16
    constexpr int ROWS_PER_INSTR = 64 / LANES_PER_ROW;     // 8 or 4
    const int orow = lane / LANES_PER_ROW, ochunk = lane % LANES_PER_ROW;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long m0 = cur_bm + grp * 128 + i * 32;
        // EPI_MASK: this lane's 16 B of the stored activation Y, one load per output store, issued first
        uint4 yv[32 / ROWS_PER_INSTR];
        if (EPI == EPI_MASK) {
#pragma unroll
            for (int rr = 0; rr < 32; rr += ROWS_PER_INSTR) {
                long long m = m0 + rr + orow;
                if (m >= out_rows) m = out_rows - 1;
                yv[rr / ROWS_PER_INSTR] = *reinterpret_cast<const uint4*>(
                    reinterpret_cast<const unsigned char*>(g.Y) + ((size_t)m * g.N + cur_bn + wn * 64) * 2 + ochunk * 16);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float bv = j ? bv1 : bv0;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = (e & 3) + 8 * (e >> 2) + 4 * h;
                float v = acc[i][j][e] + bv;
                if (RELU) v = relu_f(v);
                unsigned char* dst = patch + r * PATCH_ROW + (j * 32 + l31) * ESZ;
                if (OBF) *reinterpret_cast<__bf16*>(dst) = (__bf16)v;
                else *reinterpret_cast<float*>(dst) = v;
            }
        }
        // The read-backs by inline assembly, all of the block's at once: in front of a plain LDS load behind an LDS-DMA
        // the compiler puts s_waitcnt vmcnt(0) — in this loop a wait for every store issued so far (a store round trip
        // per KiB stored: what made the epilogue 8.1k cycles of a K = 1,024 tile, EXPERIMENTS.md §00.5) and, in the
        // persistent form, for the next tile's pieces just sent.  (EPI_MASK: the Y loads above are waited for by the
        // compiler where `keep` first uses them — once per block.)
        uint4 pv[32 / ROWS_PER_INSTR];
        {
            const unsigned pa = (unsigned)(size_t)(lds_u8*)patch + (unsigned)(orow * PATCH_ROW + ochunk * 16);
            constexpr int RS = ROWS_PER_INSTR * PATCH_ROW;      // bytes between the rows of two consecutive read-backs
            asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:%5\n ds_read_b128 %2, %4 offset:%6\n"
                         "ds_read_b128 %3, %4 offset:%7"
                         : "=&v"(pv[0]), "=&v"(pv[1]), "=&v"(pv[2]), "=&v"(pv[3])
                         : "v"(pa), "n"(RS), "n"(2 * RS), "n"(3 * RS) : "memory");
            if constexpr (ROWS_PER_INSTR == 4)
                asm volatile("ds_read_b128 %0, %4 offset:%5\n ds_read_b128 %1, %4 offset:%6\n ds_read_b128 %2, %4 offset:%7\n"
                             "ds_read_b128 %3, %4 offset:%8"
                             : "=&v"(pv[(32 / ROWS_PER_INSTR) - 4]), "=&v"(pv[(32 / ROWS_PER_INSTR) - 3]),
                               "=&v"(pv[(32 / ROWS_PER_INSTR) - 2]), "=&v"(pv[(32 / ROWS_PER_INSTR) - 1])
                             : "v"(pa), "n"(4 * RS), "n"(5 * RS), "n"(6 * RS), "n"(7 * RS) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int rr = 0; rr < 32; rr += ROWS_PER_INSTR) {
            uint4 v = pv[rr / ROWS_PER_INSTR];
            if (EPI == EPI_MASK) {      // eight bf16 per lane: keep where the stored activation is positive
                const uint4 y = yv[rr / ROWS_PER_INSTR];
                auto keep = [](unsigned vv, unsigned yy) {
                    // bf16 > 0  <=>  sign clear and not zero (NaN activations cannot come out of a ReLU)
                    const unsigned lo = ((yy & 0x8000u) == 0 && (yy & 0x7fffu) != 0) ? (vv & 0xffffu) : 0u;
                    const unsigned hi = ((yy & 0x80000000u) == 0 && (yy & 0x7fff0000u) != 0) ? (vv & 0xffff0000u) : 0u;
                    return lo | hi;
                };
                v.x = keep(v.x, y.x); v.y = keep(v.y, y.y); v.z = keep(v.z, y.z); v.w = keep(v.w, y.w);
            }
            const long long m = m0 + rr + orow;
            if (m < out_rows) *reinterpret_cast<uint4*>(cbase + (size_t)m * ldc + ochunk * 16) = v;
        }
    }
    if (PERSIST && more_tiles) {
        bv0 = bn0;
        bv1 = bn1;
        first_tile = false;
        // this wave's patch reads are done before anybody's stage-2 pieces can reach slot 2: they are issued after the
        // prologue barrier of the next tile, which this wave enters only after its LDS operations have been issued in
        // order (lgkmcnt(0) makes it explicit)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        goto tile_loop;
    }
#undef MDNO_PP_DMA
#undef MDNO_PP_LOAD
#undef MDNO_PP_MMA
#undef MDNO_PP_BARRIER
}

// out[i] = sum over slices (in order) of slab[z][i]
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slab, int slices, long long count,
                                                           float* __restrict__ out) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= count) return;
    float4 s = *reinterpret_cast<const float4*>(slab + i);
    int z = 1;
    for (; z + 4 <= slices; z += 4) {        // four slabs in flight, added in slab order
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(slab + (size_t)(z + u) * count + i);
#pragma unroll
        for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; z < slices; ++z) {
        const float4 v = *reinterpret_cast<const float4*>(slab + (size_t)z * count + i);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(out + i) = s;
}

}  // namespace

bool gemm_nt_pp_supported(long long rows, int N, int K) {
    return rows > 0 && N % PP_T == 0 && K % PP_BK == 0 && K >= 2 * PP_BK;
}

// compute units of the current device (a persistent launch is one workgroup per CU)
static int device_cus() {
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

// MDNO_GEMM_PP_PERSIST=0: one workgroup per tile for the bf16-output products too (A/B switch of the persistent form)
static bool pp_persist_enabled() {
    static const bool on = !(getenv("MDNO_GEMM_PP_PERSIST") && getenv("MDNO_GEMM_PP_PERSIST")[0] == '0');
    return on;
}

template <bool TN, int EPI, bool RELU, bool OUT_BF16, bool F16 = false>
static int launch_pp(const PpArgs& g, unsigned slices, hipStream_t s) {
    const long long nwg = g.tiles_m * g.tiles_n;
    MDNO_REQUIRE(nwg < (1ll << 31), MDNO_EUNSUPPORTED, "gemm_pp: too many tiles");
    if constexpr (!TN && OUT_BF16 && EPI != EPI_SLAB && !F16) {
        // more tiles than CUs: the persistent form (one workgroup per CU walking its XCD's tiles)
        const int cus = device_cus() / 8 * 8;
        if (pp_persist_enabled() && cus >= 8 && nwg > cus) {
            static std::atomic<unsigned long long> raised_p{0};
            MDNO_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(&gemm_pp_kernel<TN, EPI, RELU, OUT_BF16, F16, true>),
                                       PP_LDS_BYTES, raised_p));
            hipLaunchKernelGGL((gemm_pp_kernel<TN, EPI, RELU, OUT_BF16, F16, true>), dim3((unsigned)cus), dim3(512), PP_LDS_BYTES,
                               s, g);
            return check_launch("gemm_pp_kernel (persistent)");
        }
    }
    static std::atomic<unsigned long long> raised{0};
    MDNO_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(&gemm_pp_kernel<TN, EPI, RELU, OUT_BF16, F16>), PP_LDS_BYTES, raised));
    hipLaunchKernelGGL((gemm_pp_kernel<TN, EPI, RELU, OUT_BF16, F16>), dim3((unsigned)nwg, slices, F16 ? 3 : 1), dim3(512),
                       PP_LDS_BYTES, s, g);
    return check_launch("gemm_pp_kernel");
}

// C = act(A . W^T + b): A bf16 [rows,K], W bf16 [N,K] (both row-major), C bf16 or fp32 [rows,N]
int gemm_nt_pp(const void* A, const void* W, const float* bias, long long rows, int N, int K, int relu, int out_bf16,
               void* C, hipStream_t s) {
    MDNO_REQUIRE(gemm_nt_pp_supported(rows, N, K), MDNO_EUNSUPPORTED, "gemm_nt_pp: rows=%lld N=%d K=%d", rows, N, K);
    PpArgs g{static_cast<const __bf16*>(A), static_cast<const __bf16*>(W), bias, nullptr, C, rows, N, K, N / PP_T,
             (rows + PP_T - 1) / PP_T, 0};
    if (relu) return out_bf16 ? launch_pp<false, EPI_BIAS, true, true>(g, 1, s) : launch_pp<false, EPI_BIAS, true, false>(g, 1, s);
    return out_bf16 ? launch_pp<false, EPI_BIAS, false, true>(g, 1, s) : launch_pp<false, EPI_BIAS, false, false>(g, 1, s);
}

// C bf16 [rows,N] = (Y > 0) ? A . W^T : 0 — the input gradient of a Linear+ReLU layer whose stored output is Y
int gemm_nt_pp_masked(const void* A, const void* W, const void* Y, long long rows, int N, int K, void* C, hipStream_t s) {
    MDNO_REQUIRE(gemm_nt_pp_supported(rows, N, K) && Y, MDNO_EUNSUPPORTED, "gemm_nt_pp_masked: rows=%lld N=%d K=%d", rows, N, K);
    PpArgs g{static_cast<const __bf16*>(A), static_cast<const __bf16*>(W), nullptr, static_cast<const __bf16*>(Y), C,
             rows, N, K, N / PP_T, (rows + PP_T - 1) / PP_T, 0};
    return launch_pp<false, EPI_MASK, false, true>(g, 1, s);
}

// ---- C [n1,n2] fp32 = A^T . B over rows: A bf16 [rows,n1], B bf16 [rows,n2]
bool gemm_tn_pp_supported(long long rows, int n1, int n2) { return rows > 0 && n1 % PP_T == 0 && n2 % PP_T == 0; }

static int tn_slices(long long rows, int n1, int n2) {
    const long long tiles = (long long)(n1 / PP_T) * (n2 / PP_T);
    long long sl = 256 / tiles;                    // one workgroup per CU when the tiles alone do not fill the chip
    if (sl < 1) sl = 1;
    if (sl > 32) sl = 32;
    const long long max_sl = (rows + 4 * PP_BK - 1) / (4 * PP_BK);      // at least a few stages per slice
    if (sl > max_sl) sl = max_sl;
    // a slice's operand panels are addressed through 32-bit buffer offsets: keep them under 2 GiB
    const long long widest = n1 > n2 ? n1 : n2;
    while ((rows + sl - 1) / sl * widest * 2 >= (1ll << 31)) ++sl;
    return (int)sl;
}

size_t gemm_tn_pp_workspace_bytes(long long rows, int n1, int n2) {
    // slabs (none when one slice writes C directly), sized for the largest slice count the rule above picks
    long long sl = tn_slices(rows, n1, n2);
    if (sl < 32) sl = 32;
    return align_up((size_t)sl * n1 * n2 * sizeof(float), 256);
}

int gemm_tn_pp(const void* A, const void* B, long long rows, int n1, int n2, float* C, void* workspace, hipStream_t s) {
    MDNO_REQUIRE(gemm_tn_pp_supported(rows, n1, n2), MDNO_EUNSUPPORTED, "gemm_tn_pp: rows=%lld n1=%d n2=%d", rows, n1, n2);
    const int slices = tn_slices(rows, n1, n2);
    const long long slice_rows = ((rows + slices - 1) / slices + PP_BK - 1) / PP_BK * PP_BK;
    float* slabs = slices == 1 ? C : static_cast<float*>(workspace);
    PpArgs g{static_cast<const __bf16*>(A), static_cast<const __bf16*>(B), nullptr, nullptr, slabs, rows, n2, n1,
             n2 / PP_T, n1 / PP_T, slice_rows};
    MDNO_TRY((launch_pp<true, EPI_SLAB, false, false>(g, (unsigned)slices, s)));
    if (slices > 1) {
        const long long count = (long long)n1 * n2;
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((count / 4 + 255) / 256)), dim3(256), 0, s,
                           static_cast<const float*>(workspace), slices, count, C);
    }
    return check_launch("gemm_tn_pp");
}

// ---------------------------------------------------------------- fp32 A^T . B on two fp16 planes (fp32 training)
// C [n1,n2] (+)= A^T . B for fp32 A [rows,n1], B [rows,n2] — the weight gradients of the fp32 training path, which
// ran on the exact fp32 MFMA (157 TF peak: 2.0 ms for the 4096 x 1024 x 43.7k product).  Here: every COLUMN of A
// and of B (a row of A^T / B^T) is scaled by its own power of two (largest entry into [2^13, 2^14), as
// split_layout.h does for weight rows), split into two fp16 planes kept row-major, and the three plane products run
// on the transposing ping-pong kernel above (blockIdx.z = product), each K slice into its own slab; the slabs are
// then added in a fixed order, the cross terms scaled by 2^-11 once, and the column scales undone — all exact
// except the fp32 additions.
namespace {
__global__ __launch_bounds__(256) void zero_u32_kernel(unsigned* p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0u;
}

// bits of max |x[r][c]| over the rows of this block's slice -> atomicMax (non-negative floats order as integers);
// a thread owns four adjacent columns (16-B loads), four rows in flight
__global__ __launch_bounds__(256) void colmax_kernel(const float* __restrict__ x, long long rows, int n, long long slice_rows,
                                                     unsigned* __restrict__ maxbits) {
    const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= n) return;
    const long long r0 = (long long)blockIdx.y * slice_rows;
    long long r1 = r0 + slice_rows;
    if (r1 > rows) r1 = rows;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 m = {0u, 0u, 0u, 0u};
    auto take = [&](const u32x4 v) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned b = v[j] & 0x7fffffffu;
            m[j] = b > m[j] ? b : m[j];
        }
    };
    long long r = r0;
    for (; r + 4 <= r1; r += 4) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const u32x4*>(x + (size_t)(r + u) * n + c);
#pragma unroll
        for (int u = 0; u < 4; ++u) take(v[u]);
    }
    for (; r < r1; ++r) take(*reinterpret_cast<const u32x4*>(x + (size_t)r * n + c));
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (m[j]) atomicMax(&maxbits[c + j], m[j]);
}

// x [rows,n] fp32 -> hi, lo' [rows,n] fp16 of x * scale(column); 8 columns per thread
__global__ __launch_bounds__(256) void split_cols_f16_kernel(const float* __restrict__ x, long long count8, int n,
                                                             const unsigned* __restrict__ maxbits,
                                                             _Float16* __restrict__ hi, _Float16* __restrict__ lo) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= count8) return;
    const int c = (int)((i * 8) % n);
    const float4 v0 = *reinterpret_cast<const float4*>(x + i * 8), v1 = *reinterpret_cast<const float4*>(x + i * 8 + 4);
    const float xs[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    _Float16 oh[8], ol[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) split2h(xs[j] * f16_row_scale(__builtin_bit_cast(float, maxbits[c + j])), oh[j], ol[j]);
    *reinterpret_cast<uint4*>(hi + i * 8) = *reinterpret_cast<const uint4*>(oh);
    *reinterpret_cast<uint4*>(lo + i * 8) = *reinterpret_cast<const uint4*>(ol);
}

// out[i][j] (+)= (S0 + 2^-11 (S1 + S2)) / (scale_a[i] scale_b[j]),  S_p = slabs of product p added in slice order
__global__ __launch_bounds__(256) void reduce_f16_products_kernel(const float* __restrict__ slab, int slices, long long count,
                                                                  int n2, const unsigned* __restrict__ amax,
                                                                  const unsigned* __restrict__ bmax, int accumulate,
                                                                  float* __restrict__ out) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= count) return;
    float4 sum[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const float* sp = slab + (size_t)p * slices * count + i;
        float4 s = *reinterpret_cast<const float4*>(sp);
        for (int z = 1; z < slices; ++z) {
            const float4 v = *reinterpret_cast<const float4*>(sp + (size_t)z * count);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        sum[p] = s;
    }
    const int row = (int)(i / n2), col = (int)(i % n2);
    const float ua = 1.f / f16_row_scale(__builtin_bit_cast(float, amax[row]));
    float r[4] = {sum[0].x + (sum[1].x + sum[2].x) * F16_LO_UNSCALE, sum[0].y + (sum[1].y + sum[2].y) * F16_LO_UNSCALE,
                  sum[0].z + (sum[1].z + sum[2].z) * F16_LO_UNSCALE, sum[0].w + (sum[1].w + sum[2].w) * F16_LO_UNSCALE};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[j] = r[j] * ua * (1.f / f16_row_scale(__builtin_bit_cast(float, bmax[col + j])));
        if (accumulate) r[j] += out[i + j];
    }
    *reinterpret_cast<float4*>(out + i) = make_float4(r[0], r[1], r[2], r[3]);
}

struct AtbF16Ws {
    _Float16 *ah, *al, *bh, *bl;
    unsigned *amax, *bmax;
    float* slabs;
    int slices;
    size_t total;
};

AtbF16Ws carve_atb_f16(void* ws, long long rows, int n1, int n2) {
    AtbF16Ws w{};
    Carver cv(ws);
    w.ah = cv.take<_Float16>((size_t)rows * n1);
    w.al = cv.take<_Float16>((size_t)rows * n1);
    w.bh = cv.take<_Float16>((size_t)rows * n2);
    w.bl = cv.take<_Float16>((size_t)rows * n2);
    w.amax = cv.take<unsigned>((size_t)n1);
    w.bmax = cv.take<unsigned>((size_t)n2);
    // three products share the chip: a third of the slices the bf16 product would take, at least one
    w.slices = (tn_slices(rows, n1, n2) + 2) / 3;
    w.slabs = cv.take<float>((size_t)3 * w.slices * n1 * n2);
    w.total = cv.used();
    return w;
}
}  // namespace

bool gemm_atb_f16_supported(long long rows, int n1, int n2) {
    return gemm_tn_pp_supported(rows, n1, n2) && rows * (long long)(n1 > n2 ? n1 : n2) < (1ll << 40);
}

size_t gemm_atb_f16_workspace_bytes(long long rows, int n1, int n2) { return carve_atb_f16(nullptr, rows, n1, n2).total; }

int gemm_atb_f16(const float* a, const float* b, long long rows, int n1, int n2, float* c, int accumulate, void* workspace,
                 hipStream_t s) {
    MDNO_REQUIRE(gemm_atb_f16_supported(rows, n1, n2), MDNO_EUNSUPPORTED, "gemm_atb_f16: rows=%lld n1=%d n2=%d", rows, n1, n2);
    const AtbF16Ws w = carve_atb_f16(workspace, rows, n1, n2);
    hipLaunchKernelGGL(zero_u32_kernel, dim3((n1 + n2 + 255) / 256), dim3(256), 0, s, w.amax, n1 + n2);   // (amax | bmax contiguous)
    const int cm_slices = 512;
    const long long cm_rows = (rows + cm_slices - 1) / cm_slices;
    hipLaunchKernelGGL(colmax_kernel, dim3((n1 + 1023) / 1024, cm_slices), dim3(256), 0, s, a, rows, n1, cm_rows, w.amax);
    hipLaunchKernelGGL(colmax_kernel, dim3((n2 + 1023) / 1024, cm_slices), dim3(256), 0, s, b, rows, n2, cm_rows, w.bmax);
    const long long ca = rows * n1 / 8, cb = rows * n2 / 8;
    hipLaunchKernelGGL(split_cols_f16_kernel, dim3((unsigned)((ca + 255) / 256)), dim3(256), 0, s, a, ca, n1, w.amax, w.ah, w.al);
    hipLaunchKernelGGL(split_cols_f16_kernel, dim3((unsigned)((cb + 255) / 256)), dim3(256), 0, s, b, cb, n2, w.bmax, w.bh, w.bl);
    MDNO_TRY(check_launch("gemm_atb_f16: split"));
    const long long slice_rows = ((rows + w.slices - 1) / w.slices + PP_BK - 1) / PP_BK * PP_BK;
    PpArgs g{reinterpret_cast<const __bf16*>(w.ah), reinterpret_cast<const __bf16*>(w.bh), nullptr, nullptr, w.slabs, rows, n2, n1,
             n2 / PP_T, n1 / PP_T, slice_rows, reinterpret_cast<const __bf16*>(w.al), reinterpret_cast<const __bf16*>(w.bl)};
    MDNO_TRY((launch_pp<true, EPI_SLAB, false, false, true>(g, (unsigned)w.slices, s)));
    const long long count = (long long)n1 * n2;
    hipLaunchKernelGGL(reduce_f16_products_kernel, dim3((unsigned)((count / 4 + 255) / 256)), dim3(256), 0, s, w.slabs, w.slices,
                       count, n2, w.amax, w.bmax, accumulate, c);
    return check_launch("gemm_atb_f16");
}

}  // namespace mdno
