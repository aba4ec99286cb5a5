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
//     [row_tile = rows/128][k_tile = K/16][plane 0..2][128 rows][16 k]  bf16,
// each 4 KiB plane tile holding its rows as 32-B lines whose two 16-B halves are swapped when
// (row>>3)&1 — exactly what the fragment reads want in LDS.  Staging a k-step is then a straight
// 12 KiB copy per operand, done by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction,
// no VGPRs, no ds_write), and a workgroup walks one contiguous run of K/16 * 12 KiB.  We own both producers and consumers of
// these buffers, so nothing else ever sees the layout:
//   weights      fp32 [N,K] --split_planes_kernel--> tiled planes   (once per forward, or once per
//                                                                        mdno_rollout_plan_run; 30 MB)
//   layer 0      edge attrs -> relu(linear)  --split--> H1 tiled planes      (edge_l0_split_kernel)
//   layer 1      H1 x W1 -> relu -> split  ---------> H2 tiled planes        (epilogue emits planes)
//   layer 2      H2 x W2 + b  -> fp32 W_e[E, Cin*Cout]  (row-major, what the conv streams)
// so no fp32 activation is ever stored (materialized conv).  The factored conv stops after layer 1,
// whose epilogue then writes H as k-tiled fp32 (OUT 2: what csrc/moment.hip streams)
// (split_gemm_rows); the training ops use it through split_linear (OUT 0 / 3).
//
// GEMM kernel: 256x128 block tile, k-step 16 per stage, 8 waves (4x2), wave tile 64x64 = 2x2
// v_mfma_f32_32x32x16_bf16 tiles, 24 MFMAs + 12 fragment reads per stage per wave.  Each staged
// fragment feeds 2-3 of the six products, so LDS traffic per MFMA is half that of an ordinary bf16
// GEMM; the 256-row tile keeps the L2->LDS fill (36 KiB per 192 MFMAs) under the ~70 GB/s a CU
// can pull from L2 (a 128x128 tile needed 52 GB/s per CU at 65 % MFMA utilisation and stalled
// there).  Two LDS stage buffers (72 KiB, 2 workgroups = 16 waves per CU): the DMA of stage k+1
// runs under the MFMAs of stage k, one barrier per stage; the swizzle makes the ds_read_b128
// fragment reads bank-conflict-free.  Workgroups are numbered so that each
// XCD owns a contiguous range of tiles (neighbouring tiles share the A row-panel through that
// XCD's L2).
//
// SPLIT_F16 (gemm_mode 2) runs the k x k hidden layer with TWO fp16 planes instead:
//     x = x_hi + 2^-11 x_lo',   x_hi = fp16(x),  x_lo' = fp16((x - x_hi) * 2^11)      (22-23 mantissa bits)
//     a.b ~= a_hi b_hi + 2^-11 (a_hi b_lo' + a_lo' b_hi)                               (dropped: 2^-22 a_lo' b_lo')
// — three fp16 x fp16 MFMAs (exact products, fp32 accumulation, the cross terms in their own
// accumulator so that the 2^-11 is applied once, exactly) instead of six bf16 ones: half the matrix
// work and two thirds of the staged bytes for an error vs fp64 still below a plain fp32 GEMM's
// (rel. rms 7e-8 before accumulation error vs 2.4e-7; tests/test_gpu_parity.py).  fp16 has a 5-bit
// exponent: the scheme is exact only while |x| < 65504, so the producers of the planes raise a device
// flag when a value is out of range, the fp16 GEMM then exits at once and the bf16 kernels — launched
// right behind it, and exiting at once when the flag is clear — redo the chunk.  Values below 2^-14
// lose relative, never absolute, accuracy (x_hi goes subnormal, x_lo' still carries 11 more bits).
#include "kernels.h"
#include "split_layout.h"

namespace mdno {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// flags[0]: a weight is not finite (set when the weight planes are built); flags[1]: an activation of the
// current forward is out of fp16 range; flags[2], flags[3]: h1 / h2 of the current forward hold at least one
// value >= F16_ACT_MIN (split_layout.h: below that the two-plane form has only an absolute error bound).
// `need`: which of flags[2], flags[3] (bit 0, bit 1) the product's operands depend on.  Any failure sends
// the chunk down the bf16 path.  flags[F16_FALLBACK_COUNT] counts the products (GEMM launches) that took it
// (never reset by the library's forwards: a rollout plan zeroes it per run and reads it back, include/mdno.h).
constexpr int F16_FALLBACK_COUNT = 8;
// grid of a bf16 FALLBACK launch (one that exits at once unless a flag is up): a launch that is not needed then costs a
// kernel boundary, not the dispatch of a capacity-sized grid (thousands of workgroups); when it IS needed its workgroups
// walk the tiles with this stride
constexpr int kFallbackGemmGrid = 512, kFallbackL0RowTiles = 32;

__device__ __forceinline__ bool f16_blocked(const int* __restrict__ flags, int need) {
    bool b = (__builtin_nontemporal_load(flags) | __builtin_nontemporal_load(flags + 1)) != 0;
    if (need & 1) b |= __builtin_nontemporal_load(flags + 2) == 0;
    if (need & 2) b |= __builtin_nontemporal_load(flags + 3) == 0;
    return b;
}

constexpr int TN = 128, TK = 16;               // block tile is TM x 128 (TM = 128 or 256); one MFMA k-step per stage
constexpr int PLANE_BYTES = 128 * TK * 2;      // 4 KiB per (128-row) operand plane tile
// stage = A: TM/128 row tiles x 3 planes, B: 3 planes.  TM=256: 36 KiB (x2 buffers = 72 KiB, 2 workgroups
// of 8 waves per CU); TM=128: 24 KiB (48 KiB, 3 workgroups of 4 waves per CU)
constexpr int stage_bytes(int tm) { return (3 * tm / 128 + 3) * PLANE_BYTES; }

typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((address_space(1))) const unsigned char glb_u8;

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
    const int nkt = K >> 4;
#pragma unroll
    for (int p = 0; p < 3; ++p)
        *reinterpret_cast<uint4*>(planes + tiled_off(row, k0, nkt, p)) = *reinterpret_cast<const uint4*>(o[p]);
}

// Four rows per workgroup, one wave each for the first pass: the row's largest magnitude, then the power of two that
// lifts it to [2^13, 2^14) (f16_row_scale; exact).  unscale[row] = 1/scale is what the GEMM epilogue multiplies the
// row's output column (or row) by.  Second pass, all four waves over the four rows together: lane = (k-tile of 16,
// row, 8-k half), so that a store instruction writes the 4 rows x 32 B of a k-tile that lie next to each other in a
// plane tile — 128-B runs instead of the 32-B pieces a wave-per-row store scatters 8 KiB apart (the activations of a
// training batch are 0.2-0.7 GB per split: their stores are what this kernel takes).
__global__ __launch_bounds__(256) void split_planes_f16_kernel(const float* __restrict__ w, int rows, int K,
                                                               unsigned char* __restrict__ planes,
                                                               float* __restrict__ unscale,
                                                               int* __restrict__ range_flag) {
    __shared__ float scale_s[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row0 = blockIdx.x * 4;
    const int chunks = K >> 3;
    {
        const int row = row0 + wave;
        float mx = 0.f;
        if (row < rows) {
            const float* wr = w + (size_t)row * K;
            for (int c = lane; c < chunks; c += 64) {
                const float4 v0 = *reinterpret_cast<const float4*>(wr + 8 * c);
                const float4 v1 = *reinterpret_cast<const float4*>(wr + 8 * c + 4);
                mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v0.x), fabsf(v0.y)), fmaxf(fabsf(v0.z), fabsf(v0.w))));
                mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v1.x), fabsf(v1.y)), fmaxf(fabsf(v1.z), fabsf(v1.w))));
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        const float sc = f16_row_scale(mx);
        if (lane == 0) {
            scale_s[wave] = sc;
            if (row < rows) unscale[row] = 1.f / sc;       // (power of two: exact)
        }
    }
    __syncthreads();
    const int nkt = K >> 4;
    const int r = (threadIdx.x >> 1) & 3, half = threadIdx.x & 1;      // thread -> (k-tile, row, half)
    const int row = row0 + r;
    const float sc = scale_s[r];
    bool bad = false;
    if (row < rows) {
        const float* wr = w + (size_t)row * K;
        for (int kt = threadIdx.x >> 3; kt < nkt; kt += 32) {
            const int k0 = kt * 16 + half * 8;
            const float4 v0 = *reinterpret_cast<const float4*>(wr + k0);
            const float4 v1 = *reinterpret_cast<const float4*>(wr + k0 + 4);
            const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            _Float16 o[2][8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xs = x[j] * sc;
                bad |= !(fabsf(xs) < F16_MAX);        // (inf / NaN only: the row's max sits below 2^14)
                split2h(xs, o[0][j], o[1][j]);
            }
#pragma unroll
            for (int p = 0; p < 2; ++p)
                *reinterpret_cast<uint4*>(planes + tiled_off2(row, k0, nkt, p)) = *reinterpret_cast<const uint4*>(o[p]);
        }
    }
    if (bad) atomicOr(range_flag, 1);
}

// ---------------------------------------------------------------- layer 0 (+ attr gather) -> planes
// Workgroup = one 128-row tile of the chunk x 128 hidden units (8 k-steps of the tiled image).  Lanes
// run over ROWS: lane = (row = lane>>1 within the wave's 32 rows, half = lane&1), so every wave store
// is one contiguous KiB of a plane tile (32 rows x 32 B) and the workgroup fills whole 4 KiB plane
// tiles.  (The first version ran lanes over the hidden units of one edge: each store instruction
// scattered 32-B pieces 12 KiB apart and the kernel wrote its 372 MB at 3.8 TB/s; this shape reaches
// the ~6 TB/s the chip stores at.)  The edge's attributes stay in registers; the block's slice of W0
// and b0 (3.5 KiB) sits in LDS and is read as broadcasts.
constexpr int MAX_F = 8;
constexpr int L0_ROWS = 128, L0_UNITS = 128, L0_UNITS_SMALL = 32;

// MODE 1: emit the two fp16 planes (and raise f16_flags[1] on a value out of fp16 range) instead of the
// three bf16 planes; MODE 0 given f16_flags runs only when a flag is up (fallback); MODE 2: both images
// (launches of a few rows, whose GEMMs pick their operand image themselves: gemm_split_f16_small_kernel).
// (MODE 2 takes 32 hidden units per workgroup instead of 128: a launch of a few hundred rows is three row tiles, and
// with 64 outputs per thread its 24 workgroups computed for 4 us; 96 workgroups of 16 outputs per thread do not)
template <int FT, int MODE>   // FT = compile-time ker_in (6 for position-derived attributes), 0 = run-time F
__global__ __launch_bounds__(256) void edge_l0_split_kernel(
    const float* __restrict__ frames, int frame, const int* __restrict__ t_dev, int rows_per_frame,
    const int* __restrict__ src, const int* __restrict__ dst, const float* __restrict__ edge_attr,
    const int* __restrict__ perm, const int* __restrict__ num_edges, long long e_begin, int e_count, int F, int k,
    const float* __restrict__ w0, const float* __restrict__ b0, unsigned char* __restrict__ hp,
    int* __restrict__ f16_flags, int f16_need, unsigned char* __restrict__ hp_f16) {
    constexpr bool F16 = MODE != 0, BF16 = MODE != 1;
    constexpr int UNITS = MODE == 2 ? L0_UNITS_SMALL : L0_UNITS;
    __shared__ __attribute__((aligned(16))) float wsh[UNITS * MAX_F];
    __shared__ __attribute__((aligned(16))) float bsh[UNITS];
    if (MODE == 0 && f16_flags != nullptr && !f16_blocked(f16_flags, f16_need)) return;   // fallback launch, not needed
    unsigned char* const hph = MODE == 2 ? hp_f16 : hp;
    const int Fn = FT ? FT : F;
    const long long E = *num_edges;
    const long long tile0 = (long long)blockIdx.x * L0_ROWS;   // first row of this tile inside the chunk
    if (e_begin + tile0 >= E || tile0 >= e_count) return;
    const int u0 = blockIdx.y * UNITS, tid = threadIdx.x;
    // operands from which layer 0 can produce a NaN (not finite, or so large that a product overflows: |a w| <= 1e36
    // below the bound, six of them and a bias stay finite): the fp16 image of such a chunk is not used (F16 modes)
    bool operands_wild = false;
    constexpr float L0_OPERAND_BOUND = 1e18f;
    for (int i = tid; i < UNITS * Fn; i += 256) {
        const float wv = w0[(size_t)u0 * Fn + i];
        wsh[i] = wv;
        if (F16) operands_wild |= !(fabsf(wv) < L0_OPERAND_BOUND);
    }
    if (tid < UNITS) {
        const float bv = b0[u0 + tid];
        bsh[tid] = bv;
        if (F16) operands_wild |= !(fabsf(bv) < L0_OPERAND_BOUND);
    }
    const int lane = tid & 63, r = (tid >> 6) * 32 + (lane >> 1), half = lane & 1;
    const long long le = tile0 + r, e = e_begin + le;
    const bool valid = e < E && le < e_count;
    float attr[MAX_F];
#pragma unroll
    for (int f = 0; f < MAX_F; ++f) attr[f] = 0.f;
    if (valid) {
        if (frames != nullptr) {  // attr = [pos[src], pos[dst]]   (graph_kernel.py:372-379)
            const float* edge_pos = frames + (size_t)(frame + (t_dev ? *t_dev : 0)) * rows_per_frame * 3;
            const float* ps = edge_pos + (size_t)src[e] * 3;
            const float* pd = edge_pos + (size_t)dst[e] * 3;
            attr[0] = ps[0]; attr[1] = ps[1]; attr[2] = ps[2];
            attr[3] = pd[0]; attr[4] = pd[1]; attr[5] = pd[2];
        } else {
            const long long pe = perm ? (long long)perm[e] : e;
#pragma unroll
            for (int f = 0; f < MAX_F; ++f)
                if (f < Fn) attr[f] = edge_attr[pe * Fn + f];
        }
    }
    if (F16) {
#pragma unroll
        for (int f = 0; f < MAX_F; ++f) operands_wild |= !(fabsf(attr[f]) < L0_OPERAND_BOUND);
        if (operands_wild) atomicOr(f16_flags + 1, 1);
    }
    __syncthreads();
    if (!valid) return;
    const int nkt = k >> 4;
    bool bad = false, seen = false;
#pragma unroll 2
    for (int t = 0; t < UNITS / 16; ++t) {
        const int c = t * 16 + half * 8;       // this thread's 8 hidden units, relative to u0
        __bf16 o[3][8];
        _Float16 oh[2][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float sum = 0.f;
#pragma unroll
            for (int f = 0; f < MAX_F; ++f)
                if (f < Fn) sum = fmaf(attr[f], wsh[(c + j) * Fn + f], sum);
            // (v_max_f32 turns a NaN into 0 where torch's relu passes it on — and this loop is at its vector-issue rate, its
            // schedule does not survive another instruction: 61 -> 120 registers.  A NaN can only come from operands that
            // are not finite or large enough for a product to overflow; those raise the range flag BEFORE the loop
            // (operands_wild below), and the chunk is then redone by the bf16 kernels, whose ReLU is relu_f.)
            // (MODE 0, 2: the bf16 image is the one used then — relu_f's compare-and-select, written as instructions: given
            // the expression the compiler re-vectorises the loop to 122 registers)
            float v = sum + bsh[c + j];
            if (MODE == 1) v = fmaxf(v, 0.f);
            else asm("v_cmp_le_f32 vcc, %0, 0\n\tv_cndmask_b32 %0, %0, 0, vcc" : "+v"(v) : : "vcc");
            if (F16) {
                bad |= !(v < F16_MAX);
                seen |= v >= F16_ACT_MIN;
                split2h(v, oh[0][j], oh[1][j]);
            }
            if (BF16) split3(v, o[0][j], o[1][j], o[2][j]);
        }
        if (F16) {
#pragma unroll
            for (int p = 0; p < 2; ++p)
                *reinterpret_cast<uint4*>(hph + tiled_off2(le, u0 + c, nkt, p)) = *reinterpret_cast<const uint4*>(oh[p]);
        }
        if (BF16) {
#pragma unroll
            for (int p = 0; p < 3; ++p)
                *reinterpret_cast<uint4*>(hp + tiled_off(le, u0 + c, nkt, p)) = *reinterpret_cast<const uint4*>(o[p]);
        }
    }
    if (F16 && bad) atomicOr(f16_flags + 1, 1);
    if (F16 && seen) f16_flags[2] = 1;      // (the same value from whoever stores it: no atomic)
}

// The bf16 image of a chunk for the FALLBACK of gemm_mode SPLIT_F16 (MODE 0's arithmetic): launched behind the fp16
// kernels with a small grid (kFallbackL0RowTiles row tiles x k/128), it exits at once unless a range flag is up — a
// kernel boundary instead of the dispatch of a capacity-sized grid — and otherwise its workgroups walk the row tiles
// with stride gridDim.x.
template <int FT>
__global__ __launch_bounds__(256) void edge_l0_split_fallback_kernel(
    const float* __restrict__ frames, int frame, const int* __restrict__ t_dev, int rows_per_frame,
    const int* __restrict__ src, const int* __restrict__ dst, const float* __restrict__ edge_attr,
    const int* __restrict__ perm, const int* __restrict__ num_edges, long long e_begin, int e_count, int F, int k,
    const float* __restrict__ w0, const float* __restrict__ b0, unsigned char* __restrict__ hp,
    const int* __restrict__ f16_flags, int f16_need) {
    __shared__ __attribute__((aligned(16))) float wsh[L0_UNITS * MAX_F];
    __shared__ __attribute__((aligned(16))) float bsh[L0_UNITS];
    if (!f16_blocked(f16_flags, f16_need)) return;
    const int Fn = FT ? FT : F;
    const long long E = *num_edges;
    const int u0 = blockIdx.y * L0_UNITS, tid = threadIdx.x;
    for (int i = tid; i < L0_UNITS * Fn; i += 256) wsh[i] = w0[(size_t)u0 * Fn + i];
    if (tid < L0_UNITS) bsh[tid] = b0[u0 + tid];
    __syncthreads();
    const int lane = tid & 63, r = (tid >> 6) * 32 + (lane >> 1), half = lane & 1;
    const int nkt = k >> 4;
    for (long long tile0 = (long long)blockIdx.x * L0_ROWS; e_begin + tile0 < E && tile0 < e_count;
         tile0 += (long long)gridDim.x * L0_ROWS) {
        const long long le = tile0 + r, e = e_begin + le;
        if (!(e < E && le < e_count)) continue;
        float attr[MAX_F];
#pragma unroll
        for (int f = 0; f < MAX_F; ++f) attr[f] = 0.f;
        if (frames != nullptr) {
            const float* edge_pos = frames + (size_t)(frame + (t_dev ? *t_dev : 0)) * rows_per_frame * 3;
            const float* ps = edge_pos + (size_t)src[e] * 3;
            const float* pd = edge_pos + (size_t)dst[e] * 3;
            attr[0] = ps[0]; attr[1] = ps[1]; attr[2] = ps[2];
            attr[3] = pd[0]; attr[4] = pd[1]; attr[5] = pd[2];
        } else {
            const long long pe = perm ? (long long)perm[e] : e;
#pragma unroll
            for (int f = 0; f < MAX_F; ++f)
                if (f < Fn) attr[f] = edge_attr[pe * Fn + f];
        }
#pragma unroll 2
        for (int t = 0; t < L0_UNITS / 16; ++t) {
            const int c = t * 16 + half * 8;
            __bf16 o[3][8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float sum = 0.f;
#pragma unroll
                for (int f = 0; f < MAX_F; ++f)
                    if (f < Fn) sum = fmaf(attr[f], wsh[(c + j) * Fn + f], sum);
                float v = sum + bsh[c + j];
                asm("v_cmp_le_f32 vcc, %0, 0\n\tv_cndmask_b32 %0, %0, 0, vcc" : "+v"(v) : : "vcc");      // relu_f (see edge_l0_split_kernel)
                split3(v, o[0][j], o[1][j], o[2][j]);
            }
#pragma unroll
            for (int p = 0; p < 3; ++p)
                *reinterpret_cast<uint4*>(hp + tiled_off(le, u0 + c, nkt, p)) = *reinterpret_cast<const uint4*>(o[p]);
        }
    }
}

// f16 = true: fp16 planes + range flag; f16 = false with flags: the bf16 fallback (runs if a flag is up)
static int launch_edge_l0_split(const float* pos_mode, int frame, const int* t_dev, int rows_per_frame, const int* src,
                                const int* dst, const float* edge_attr, const int* perm, const int* num_edges,
                                long long e0, int cnt, int F, int k, const float* w0, const float* b0,
                                unsigned char* hp, hipStream_t s, bool f16 = false, int* f16_flags = nullptr,
                                int f16_need = 0, unsigned char* hp_f16 = nullptr) {
    const int row_tiles = (cnt + L0_ROWS - 1) / L0_ROWS;
    if (!f16 && f16_flags != nullptr && hp_f16 == nullptr) {      // the fallback launch behind the fp16 kernels
        const dim3 fgrid(row_tiles < kFallbackL0RowTiles ? row_tiles : kFallbackL0RowTiles, k / L0_UNITS);
        if (F == 6)
            hipLaunchKernelGGL((edge_l0_split_fallback_kernel<6>), fgrid, dim3(256), 0, s, pos_mode, frame, t_dev, rows_per_frame,
                               src, dst, edge_attr, perm, num_edges, e0, cnt, F, k, w0, b0, hp, (const int*)f16_flags, f16_need);
        else
            hipLaunchKernelGGL((edge_l0_split_fallback_kernel<0>), fgrid, dim3(256), 0, s, pos_mode, frame, t_dev, rows_per_frame,
                               src, dst, edge_attr, perm, num_edges, e0, cnt, F, k, w0, b0, hp, (const int*)f16_flags, f16_need);
        return check_launch("edge_l0_split_fallback_kernel");
    }
    const dim3 grid(row_tiles, k / (hp_f16 ? L0_UNITS_SMALL : L0_UNITS));
    // hp_f16 given: both images (bf16 planes -> hp, fp16 planes -> hp_f16)
#define MDNO_L0(FT, MODE)                                                                                             \
    hipLaunchKernelGGL((edge_l0_split_kernel<FT, MODE>), grid, dim3(256), 0, s, pos_mode, frame, t_dev, rows_per_frame, \
                       src, dst, edge_attr, perm, num_edges, e0, cnt, F, k, w0, b0, hp, f16_flags, f16_need, hp_f16)
    if (F == 6) { if (hp_f16) MDNO_L0(6, 2); else if (f16) MDNO_L0(6, 1); else MDNO_L0(6, 0); }
    else        { if (hp_f16) MDNO_L0(0, 2); else if (f16) MDNO_L0(0, 1); else MDNO_L0(0, 0); }
#undef MDNO_L0
    return check_launch("edge_l0_split_kernel");
}

// ---------------------------------------------------------------- split-bf16 GEMM
struct SplitGemmArgs {
    const unsigned char* Ap;   // tiled planes of A  [rows/128][K/32][3][8 KiB]
    const unsigned char* Bp;   // tiled planes of Bt [N/128][K/32][3][8 KiB]
    const float* bias;         // [N]
    float* C;                  // fp32 [rows][N] row-major            (OUT_PLANES = false)
    unsigned char* Cp;         // tiled planes [rows/128][N/32][3][8 KiB], ReLU applied (OUT_PLANES = true)
    const int* num_edges;      // device count of valid rows (rows past it are neither computed nor stored) ...
    long long row_begin;       // ... relative to this first row; NULL: rows_valid below is used instead
    int rows, N, K;
    int tiles_n, tiles_m;
    int rows_valid;
    int m_fastest;             // tile order: 0 = n fastest (neighbours share the A row-panel), 1 = m fastest (share B)
    const int* f16_flags = nullptr;   // SPLIT_F16: the fp16 kernel runs while no flag is up, the bf16 one (given
                                      // the flags) only when one is; NULL: unconditional
    int f16_need = 0;                 // which "seen" words the operands depend on (f16_blocked)
    const float* b_unscale = nullptr; // fp16 kernel: per-column factor undoing the weight rows' power-of-two scale
    // gemm_split_f16_small_kernel only: the bf16 images of the same operands (and of the output, OUT 4).  Given
    // these the kernel multiplies them itself when a flag is up — no fallback launch behind it
    const unsigned char* Ap_b = nullptr;
    const unsigned char* Bp_b = nullptr;
    unsigned char* Cp_b = nullptr;
    // fp16 kernels: per-ROW factor undoing a power-of-two scale of A's rows (split_linear_f16: activations and
    // gradients of any magnitude, scaled row by row like the weights); [rows] floats, NULL = none
    const float* a_unscale = nullptr;
};

// One stage (k-step of 16) for a wave: (2x2 tiles) x 6 plane products = 24 MFMAs, 12 fragment reads.
__device__ __forceinline__ void mma_split_stage(f32x16 (&acc)[2][2], const unsigned char* st, int a_rd, int b_rd) {
    bf16x8 a[2][3], b[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            a[i][p] = *reinterpret_cast<const bf16x8*>(st + p * PLANE_BYTES + a_rd + i * 32 * 32);
            b[i][p] = *reinterpret_cast<const bf16x8*>(st + p * PLANE_BYTES + b_rd + i * 32 * 32);
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

// OUT: 0 = fp32 row-major (W_e), 1 = tiled bf16 planes after ReLU (next GEMM's operand),
//      2 = fp32 k-tiled [rows/128][N/32][128][32] after ReLU (the hidden activation the factored conv streams)
//      3 = fp32 row-major after ReLU (training: mdno_linear_split_fwd)
// one TM x 128 tile of the product: tile `orig` of `nwg` in the XCD-aware order
template <int TM, int OUT>
__device__ __forceinline__ void gemm_split_bf16_tile(const SplitGemmArgs& g, unsigned char* lds, long long valid, int nwg, int orig) {
    constexpr int WAVES = TM / 32;                       // (TM/64) x 2 waves of 64x64
    constexpr int STAGE_BYTES = stage_bytes(TM);
    constexpr int PIECES = STAGE_BYTES / 1024;           // 1 KiB DMA pieces per stage: 24 or 36
    constexpr int A_PIECES = PIECES - 12;
    constexpr int PPW = (PIECES + WAVES - 1) / WAVES;    // pieces per wave: 6 or 5
    // XCD-aware tile order over the tiles that hold valid rows: workgroups b, b+8, ... share an XCD
    // (round-robin dispatch); give each XCD a contiguous range of tiles.  Bijective for any count.
    const int xcd = orig & 7, q = nwg >> 3, r8 = nwg & 7;
    const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
    const int tiles_mv = nwg / g.tiles_n;
    const int bm = (g.m_fastest ? tile % tiles_mv : tile / g.tiles_n) * TM;
    const int bn = (g.m_fastest ? tile / tiles_mv : tile % g.tiles_n) * TN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
    const int wm = wave >> 1, wn = wave & 1;                      // (TM/64) x 2 waves, 64x64 each
    const int l31 = lane & 31, h = lane >> 5;

    // ---- staging by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, no VGPRs, no
    // ds_write).  A stage is PIECES pieces of 1 KiB, already in LDS-image order in HBM: 12 per A
    // row tile (planes 0-2, contiguous), then 12 for the B row tile.  Wave w moves pieces
    // w, w+WAVES, w+2*WAVES, ...
    const int nkt = g.K / TK;
    const size_t stage_stride = 3 * PLANE_BYTES;   // 12 KiB per k-step per 128-row tile
    const unsigned char* a_base = g.Ap + ((size_t)(bm >> 7) * nkt * 3 << 12);
    const unsigned char* b_base = g.Bp + ((size_t)(bn >> 7) * nkt * 3 << 12);
    const size_t a_tile_stride = (size_t)nkt * 3 << 12;
    const unsigned char* psrc[PPW];
#pragma unroll
    for (int t = 0; t < PPW; ++t) {
        int qq = wave + t * WAVES;
        if (qq >= PIECES) qq = wave;               // unused slot (guarded below), keep the pointer valid
        psrc[t] = (qq < A_PIECES ? a_base + (size_t)(qq / 12) * a_tile_stride + (qq % 12) * 1024
                                 : b_base + (qq - A_PIECES) * 1024) + lane * 16;
    }
    const int d0 = wave * 1024;
#define MDNO_DMA_STAGE(KT, BUF)                                                                       \
    {                                                                                                 \
        const size_t ko = (size_t)(KT) * stage_stride;                                                \
        lds_u8* ldst = (lds_u8*)(lds + (BUF) * STAGE_BYTES + d0);                                     \
        _Pragma("unroll") for (int t = 0; t < PPW; ++t)                                               \
            if (PIECES % WAVES == 0 || t < PPW - 1 || wave + t * WAVES < PIECES)                      \
                __builtin_amdgcn_global_load_lds((glb_u8*)(psrc[t] + ko), ldst + t * WAVES * 1024, 16, 0, 0); \
    }

    // ---- fragment read map: 32-B rows, 16-B half h swapped by (row>>3)&1.  A rows of wave wm live
    // in row-tile wm>>1 at (wm&1)*64 + i*32 + l31; B rows at wn*64 + i*32 + l31 behind the A planes.
    const int hsw = (h ^ ((l31 >> 3) & 1)) << 4;
    const int a_rd = (wm >> 1) * 3 * PLANE_BYTES + ((wm & 1) * 64 + l31) * 32 + hsw;
    const int b_rd = A_PIECES * 1024 + (wn * 64 + l31) * 32 + hsw;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // The bias is fetched HERE, before the K loop, and pinned: left to itself the compiler sinks the
    // load into each of the 64 predicated store blocks of the epilogue, and every store then sits
    // behind an s_waitcnt vmcnt(0) that also waits for the previous store's acknowledgement
    // (7.7 us of serialized round trips per workgroup, measured with s_memrealtime stamps).
    float bv0 = 0.f, bv1 = 0.f;
    if (g.bias) {
        bv0 = g.bias[bn + wn * 64 + l31];
        bv1 = g.bias[bn + wn * 64 + 32 + l31];
    }
    asm volatile("" : "+v"(bv0), "+v"(bv1));

    // Pipeline: stage kt+1 is in flight (DMA) while stage kt is multiplied.  __syncthreads() waits
    // for this wave's outstanding DMA (vmcnt(0)) and then for every wave: after it, stage kt is
    // complete in LDS and nobody still reads the buffer stage kt+1 is about to overwrite.
    // The wait for this wave's own DMA is written out: a workgroup-scope barrier is not required by the
    // memory model to imply vmcnt(0) (today's compiler emits it anyway; the explicit wait costs nothing).
#define MDNO_DMA_BARRIER()                                 \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       \
    __syncthreads();
    MDNO_DMA_STAGE(0, 0)
    for (int kt = 0; kt < nkt; kt += 2) {      // K is a multiple of 32: stages come in pairs
        MDNO_DMA_BARRIER()
        MDNO_DMA_STAGE(kt + 1, 1)
        mma_split_stage(acc, lds, a_rd, b_rd);
        MDNO_DMA_BARRIER()
        if (kt + 2 < nkt) MDNO_DMA_STAGE(kt + 2, 0)
        mma_split_stage(acc, lds + STAGE_BYTES, a_rd, b_rd);
    }
#undef MDNO_DMA_STAGE
#undef MDNO_DMA_BARRIER

    // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = bn + wn * 64 + j * 32 + l31;
        const float bv = j ? bv1 : bv0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = bm + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < valid) {
                    const float v = acc[i][j][e] + bv;
                    if (OUT == 2) {   // k-tiled fp32 image [m/128][n/32][128][32] (csrc/moment.hip K1)
                        g.C[((size_t)(m >> 7) * (g.N >> 5) + (n >> 5)) * 4096 + (m & 127) * 32 + (n & 31)] = relu_f(v);
                    } else if (OUT == 1) {
                        __bf16 ph, pm, pl;
                        split3(relu_f(v), ph, pm, pl);
                        const size_t o = tiled_off(m, n, g.N >> 4, 0);
                        *reinterpret_cast<__bf16*>(g.Cp + o) = ph;
                        *reinterpret_cast<__bf16*>(g.Cp + o + PLANE_BYTES) = pm;
                        *reinterpret_cast<__bf16*>(g.Cp + o + 2 * PLANE_BYTES) = pl;
                    } else {
                        g.C[(size_t)m * g.N + n] = OUT == 3 ? relu_f(v) : v;
                    }
                }
            }
        }
    }
}

// STRIDE: a FALLBACK launch of the SPLIT_F16 mode (g.f16_flags given): it exits at once unless a range flag is up, so it
// gets a small grid (kFallbackGemmGrid) whose workgroups walk the tiles, and it counts itself (F16_FALLBACK_COUNT).
template <int TM, int OUT, bool STRIDE = false>
// (second launch bound = waves per SIMD, not workgroups per CU; the LDS footprint decides the latter)
__global__ __launch_bounds__(TM * 2, 2) void gemm_split_bf16_kernel(SplitGemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    if (g.f16_flags != nullptr && !f16_blocked(g.f16_flags, g.f16_need)) return;   // bf16 fallback launch, not needed
    if (g.f16_flags != nullptr && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(const_cast<int*>(g.f16_flags) + F16_FALLBACK_COUNT, 1);
    long long valid = g.num_edges ? (long long)(*g.num_edges) - g.row_begin : (long long)g.rows_valid;
    if (valid > g.rows) valid = g.rows;
    if (valid <= 0) return;
    const int nwg = g.tiles_n * (int)((valid + TM - 1) / TM);
    if constexpr (STRIDE) {
        for (int orig = blockIdx.x; orig < nwg; orig += gridDim.x) {
            gemm_split_bf16_tile<TM, OUT>(g, lds, valid, nwg, orig);
            __syncthreads();      // (the next tile's first DMA lands in the buffer the slowest wave may still read)
        }
    } else {
        if ((int)blockIdx.x < nwg) gemm_split_bf16_tile<TM, OUT>(g, lds, valid, nwg, blockIdx.x);
    }
}


// ---------------------------------------------------------------- two-plane fp16 GEMM (SPLIT_F16)
// Same 256x128 block tile, wave tile, LDS image and tile order as gemm_split_bf16_kernel, with two
// planes per operand and two accumulator sets (hi.hi, and the cross terms that carry the 2^-11).  The
// second set costs 64 VGPRs (170 in all), which leaves one 8-wave workgroup per CU instead of two, so
// the latency of the LDS-DMA can no longer hide behind a neighbour workgroup: stages are twice as long
// (two k-steps = 32 k: 24 MFMAs + 16 fragment reads per wave, one barrier) and live in a ring of three
// (3 x 48 KiB), two stages ahead of the MFMAs, with a COUNTED wait — `s_waitcnt vmcnt(6)` lets the
// newest stage's six pieces stay in flight (loads return in order; the K loop issues no stores).
template <int MI>   // MI 32-row tiles x 2 32-column tiles per wave
__device__ __forceinline__ void mma_f16_kstep(f32x16 (&acc)[MI][2], f32x16 (&accx)[MI][2], const unsigned char* st,
                                              int a_rd, int b_rd) {
    f16x8 a[MI][2], b[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i][p] = *reinterpret_cast<const f16x8*>(st + p * PLANE_BYTES + a_rd + i * 32 * 32);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j][p] = *reinterpret_cast<const f16x8*>(st + p * PLANE_BYTES + b_rd + j * 32 * 32);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[j][0], accx[i][j], 0, 0, 0);
            accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][1], accx[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
        }
}

// The same k-step with something to do between its (i, j) groups of three MFMAs: fill(g), g = 0 .. 2*MI-1, is
// called after group g has been issued and pinned there — the LDS-DMA pieces of the stage after next go
// out one per group, in the shadow of MFMAs already in the pipe, instead of as a burst of six in front of
// the stage's first fragment read (an LDS-DMA piece costs the issuing wave 60-180 cycles, more inside a
// burst: MI355X_MICROARCH.md, cycle constants).
template <int MI, class F>
__device__ __forceinline__ void mma_f16_kstep_fill(f32x16 (&acc)[MI][2], f32x16 (&accx)[MI][2], const unsigned char* st,
                                                   int a_rd, int b_rd, F&& fill) {
    f16x8 a[MI][2], b[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i][p] = *reinterpret_cast<const f16x8*>(st + p * PLANE_BYTES + a_rd + i * 32 * 32);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j][p] = *reinterpret_cast<const f16x8*>(st + p * PLANE_BYTES + b_rd + j * 32 * 32);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[j][0], accx[i][j], 0, 0, 0);
            accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][1], accx[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            fill(i * 2 + j);
            __builtin_amdgcn_sched_barrier(0);
        }
}

// one LDS-DMA piece (64 lanes x 16 B) as a buffer load: descriptor {base, bytes}, per-lane offset, scalar offset
__device__ __forceinline__ void dma_piece_buffer(const unsigned char* base, int bytes, lds_u8* dst, unsigned voffset,
                                                 unsigned soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000), dst, 16,
                                             voffset, soffset, 0, 0);
}

constexpr int F16_TM = 256, F16_RING = 3;
constexpr int F16_TILE_BYTES = 2 * 2 * PLANE_BYTES;       // one 128-row tile: two k-steps x two planes = 16 KiB
constexpr int F16_STAGE_BYTES = 3 * F16_TILE_BYTES;       // A tile 0 | A tile 1 | B = 48 KiB

// OUT: 2 = fp32 k-tiled after ReLU (the factored conv's H), 0 = fp32 row-major, 3 = fp32 row-major after ReLU,
//      4 = two fp16 planes after ReLU (the next GEMM's operand; raises f16_flags[1] on a value out of range)
// MI = 32-row tiles per wave: 2 -> 8 waves of 64x64 (162 VGPRs, two waves per SIMD).  The LDS pipe, not
// the matrix pipe, is what this shape runs into (176 KiB of LDS traffic per 1,536 matrix-pipe cycles at
// 128 B/clk); MI = 4 — 4 waves of 128x64, a third fewer fragment reads per MFMA, but 2 x 128 accumulator
// registers and hence ONE wave per SIMD — was measured at 452 us against 392: nothing hides the
// barrier and the fragment-read latency any more.  Only MI = 2 is instantiated.
// ROW_SCALE: the rows of A carry their own power-of-two scale as well (g.a_unscale; split_linear_f16) — its own
// instantiation: the 32 extra registers and multiplies in the epilogue cost the inference kernels 12 % (353 -> 400 us
// for the hidden GEMM at N = 504) when they sat in the common one
template <int OUT, int MI, bool ROW_SCALE = false>
__global__ __launch_bounds__(1024 / MI) void gemm_split_f16_kernel(SplitGemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int TM = F16_TM, WAVES = 16 / MI;
    constexpr int PPW = F16_STAGE_BYTES / 1024 / WAVES;   // one-KiB DMA pieces per wave per stage: 6 or 12
    if (g.f16_flags != nullptr && f16_blocked(g.f16_flags, g.f16_need)) return;   // out of fp16 range: the bf16 launch behind us runs
    long long valid = g.num_edges ? (long long)(*g.num_edges) - g.row_begin : (long long)g.rows_valid;
    if (valid > g.rows) valid = g.rows;
    if (valid <= 0) return;
    const int nwg = g.tiles_n * (int)((valid + TM - 1) / TM);
    const int orig = blockIdx.x;
    if (orig >= nwg) return;
    const int xcd = orig & 7, q = nwg >> 3, r8 = nwg & 7;
    const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
    const int tiles_mv = nwg / g.tiles_n;
    const int bm = (g.m_fastest ? tile % tiles_mv : tile / g.tiles_n) * TM;
    const int bn = (g.m_fastest ? tile / tiles_mv : tile % g.tiles_n) * TN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;              // (8 / MI) x 2 waves
    const int l31 = lane & 31, h = lane >> 5;

    // piece qq of a stage: block qq/16 (A row tile 0, A row tile 1, B), KiB qq%16 of that block's 16 KiB;
    // in HBM a row tile's k-steps are contiguous (8 KiB each), so a stage is one 16 KiB run per block
    const int nkt = g.K / TK, nst = g.K / 32;
    // pieces go out as buffer loads (buffer_load_dwordx4 ... offen lds): one descriptor per operand panel based at
    // this tile's first byte, a per-lane 32-bit offset that never changes, the stage as the scalar offset — no
    // 64-bit address arithmetic per piece (gemm_bf16.hip measured 8 % on its DMA-bound loop)
    const size_t a_tile_stride = (size_t)nkt * 2 << 12;
    const unsigned char* const a_panel = g.Ap + (size_t)(bm >> 7) * a_tile_stride;      // two row tiles
    const unsigned char* const b_panel = g.Bp + (size_t)(bn >> 7) * a_tile_stride;
    const int a_bytes = (int)(2 * a_tile_stride), b_bytes = (int)a_tile_stride;
    unsigned voff[PPW];
#pragma unroll
    for (int t = 0; t < PPW; ++t) {
        const int qq = wave + t * WAVES, blk = qq >> 4;
        voff[t] = (unsigned)((blk == 1 ? a_tile_stride : 0) + (qq & 15) * 1024 + lane * 16);
    }
    // (piece t of a wave belongs to block (wave + t * WAVES) >> 4: with 8 waves, t = 0, 1 -> A tile 0, t = 2, 3 -> A tile 1,
    // t = 4, 5 -> B; with 4 waves twice as many each)
#define MDNO_PIECE_DMA(T, DST, KO)                                                              \
    dma_piece_buffer((((T) * WAVES) >> 4) < 2 ? a_panel : b_panel, (((T) * WAVES) >> 4) < 2 ? a_bytes : b_bytes, DST, voff[T], KO)
    const int d0 = wave * 1024;
#define MDNO_DMA_STAGE(ST, SLOT)                                                                       \
    {                                                                                                  \
        const unsigned ko = (unsigned)(ST) * F16_TILE_BYTES;                                           \
        lds_u8* ldst = (lds_u8*)(lds + (SLOT) * F16_STAGE_BYTES + d0);                                 \
        _Pragma("unroll") for (int t = 0; t < PPW; ++t)                                                \
            MDNO_PIECE_DMA(t, ldst + t * WAVES * 1024, ko);                                        \
    }

    const int hsw = (h ^ ((l31 >> 3) & 1)) << 4;
    const int arow = wm * (MI * 32) + l31;                // first of the wave's rows inside the 256-row block
    const int a_rd = (arow >> 7) * F16_TILE_BYTES + (arow & 127) * 32 + hsw;
    const int b_rd = 2 * F16_TILE_BYTES + (wn * 64 + l31) * 32 + hsw;

    f32x16 acc[MI][2], accx[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; accx[i][j][e] = 0.f; }
    float bv0 = 0.f, bv1 = 0.f;      // fetched before the K loop and pinned (see gemm_split_bf16_kernel)
    float us0 = 1.f, us1 = 1.f;      // undo the power-of-two scale of the weight rows behind these two columns
    if (g.bias) {
        bv0 = g.bias[bn + wn * 64 + l31];
        bv1 = g.bias[bn + wn * 64 + 32 + l31];
    }
    if (g.b_unscale) {
        us0 = g.b_unscale[bn + wn * 64 + l31];
        us1 = g.b_unscale[bn + wn * 64 + 32 + l31];
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bv0), "+v"(bv1), "+v"(us0), "+v"(us1));   // the counted waits below must see DMA pieces only

    MDNO_DMA_STAGE(0, 0)
    if (nst > 1) MDNO_DMA_STAGE(1, 1)
    int slot = 0, slot_in = 2;       // slot being multiplied; slot the next DMA goes to
    for (int st = 0; st < nst; ++st) {
        // this wave's pieces of stage st have landed (the next stage's PPW pieces may still be in flight) ...
        if (st + 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... and so have everybody's; nobody still reads the slot stage st+2 goes to (multiplied at st-1:
        // a wave's fragment reads feed its MFMAs, so they have returned before it gets here).  The bare
        // barrier instruction: __syncthreads() carries a fence that the compiler lowers to vmcnt(0),
        // which would wait for the stage just put in flight
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned char* sb = lds + slot * F16_STAGE_BYTES;
        const bool more = st + 2 < nst;
        const unsigned ko = (unsigned)(st + 2) * F16_TILE_BYTES;
        lds_u8* ldst = (lds_u8*)(lds + slot_in * F16_STAGE_BYTES + d0);
        // pieces of stage st+2: PPW / 2 behind the groups of each k-step (MI = 2: 3 + 3 of the 4 + 4 groups)
        constexpr int HALF = (PPW + 1) / 2;
        // (written out rather than passed as a functor to mma_f16_kstep_fill: a lambda that touches a buffer
        // descriptor makes the host pass drop this kernel's stub)
#define MDNO_F16_KSTEP_DMA(SB, P0, NP)                                                                                \
        {                                                                                                                 \
            f16x8 a_[MI][2], b_[2][2];                                                                                    \
            _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                                               \
                _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                            \
                    a_[i][p] = *reinterpret_cast<const f16x8*>((SB) + p * PLANE_BYTES + a_rd + i * 32 * 32);              \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                             \
                    b_[j][p] = *reinterpret_cast<const f16x8*>((SB) + p * PLANE_BYTES + b_rd + j * 32 * 32);              \
            }                                                                                                             \
            _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                                \
                _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                           \
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_[i][1], b_[j][0], accx[i][j], 0, 0, 0);         \
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_[i][0], b_[j][1], accx[i][j], 0, 0, 0);         \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_[i][0], b_[j][0], acc[i][j], 0, 0, 0);           \
                    __builtin_amdgcn_sched_barrier(0);                                                                    \
                    if (more && i * 2 + j < (NP))                                                                         \
                        MDNO_PIECE_DMA((P0) + i * 2 + j, ldst + ((P0) + i * 2 + j) * WAVES * 1024, ko);                   \
                    __builtin_amdgcn_sched_barrier(0);                                                                    \
                }                                                                                                         \
        }
        MDNO_F16_KSTEP_DMA(sb, 0, HALF)
        MDNO_F16_KSTEP_DMA(sb + 2 * PLANE_BYTES, HALF, PPW - HALF)
#undef MDNO_F16_KSTEP_DMA
        slot = slot == F16_RING - 1 ? 0 : slot + 1;
        slot_in = slot_in == F16_RING - 1 ? 0 : slot_in + 1;
    }
#undef MDNO_DMA_STAGE
#undef MDNO_PIECE_DMA

    // the rows' own scale factors, all fetched before the first store (a load inside the predicated store blocks
    // would put a full wait in front of every store)
    float ua[ROW_SCALE ? MI : 1][16];
    if (ROW_SCALE) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                ua[ROW_SCALE ? i : 0][e] = g.a_unscale[bm + wm * (MI * 32) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h];
    }
    bool bad = false, seen = false;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = bn + wn * 64 + j * 32 + l31;
        const float bv = j ? bv1 : bv0, us = j ? us1 : us0;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = bm + wm * (MI * 32) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < valid) {
                    float v = (acc[i][j][e] + accx[i][j][e] * F16_LO_UNSCALE) * us;
                    if (ROW_SCALE) v *= ua[ROW_SCALE ? i : 0][e];
                    v += bv;
                    if (OUT == 2) {
                        g.C[((size_t)(m >> 7) * (g.N >> 5) + (n >> 5)) * 4096 + (m & 127) * 32 + (n & 31)] = relu_f(v);
                    } else if (OUT == 4) {
                        const float rv = relu_f(v);
                        bad |= !(rv < F16_MAX);
                        seen |= rv >= F16_ACT_MIN;
                        _Float16 ph, pl;
                        split2h(rv, ph, pl);
                        const size_t o = tiled_off2(m, n, g.N >> 4, 0);
                        *reinterpret_cast<_Float16*>(g.Cp + o) = ph;
                        *reinterpret_cast<_Float16*>(g.Cp + o + PLANE_BYTES) = pl;
                    } else {
                        g.C[(size_t)m * g.N + n] = OUT == 3 ? relu_f(v) : v;
                    }
                }
            }
        }
    }
    if (OUT == 4 && bad) atomicOr(const_cast<int*>(g.f16_flags) + 1, 1);
    if (OUT == 4 && seen) const_cast<int*>(g.f16_flags)[3] = 1;
}

// ---- the same product for a FEW rows (a 28-atom chain has 330 edges): (32 WM) x 64 tiles, WM x 2 waves of 32 x 32.
// At E = 330 the kernel above launches 2 x 8 (hidden layer) or 2 x 32 (last layer) workgroups whose K loops
// take 35-40 us whatever the row count — 24 MFMAs per wave per stage, 32 stages.  Here a wave has 6 MFMAs per
// stage, and what bounds the K loop is how many KiB of operand planes ONE CU has to pull through its L2 -> LDS
// path (~64 B/clk): (32 WM + 64) x K x 4 B per workgroup.  WM is chosen so that the launch has about as many
// workgroups as the chip has CUs: the last layer (64 column tiles) takes 128 x 64 tiles (768 KiB per workgroup),
// the hidden layer (16 column tiles) 32 x 64 (384 KiB).  The stages (A: WM KiB per k-step and plane, B: 2 KiB)
// sit in a ring of six with five in flight.  Per output element the MFMAs are the same instructions in the
// same order as in gemm_split_f16_kernel (k-steps ascending; cross terms a_lo b_hi, a_hi b_lo, then
// a_hi b_hi), so all these kernels give the same bits and the choice between them — made from the launch's row
// capacity — never shows in a result (an ensemble member's trajectory is the same alone and in a batch).
constexpr int F16S_TN = 64, F16S_RING = 6;
constexpr int f16s_stage_bytes(int wm) { return (4 * wm + 8) * 1024; }      // 24 / 16 / 12 KiB
constexpr int f16s_lds_bytes(int wm) {
    return F16S_RING * f16s_stage_bytes(wm) > 2 * wm * 32 * 40 * 4 ? F16S_RING * f16s_stage_bytes(wm) : 2 * wm * 32 * 40 * 4;
}

template <int OUT, int WM>
__global__ __launch_bounds__(128 * WM) void gemm_split_f16_small_kernel(SplitGemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int TM = 32 * WM, WAVES = 2 * WM, PPW = 2 + 4 / WM, STAGE = f16s_stage_bytes(WM);
    constexpr int A_BYTES = 4 * WM * 1024;                // A part of a stage: 4 runs (k-step, plane) of WM KiB
    const bool blocked = g.f16_flags != nullptr && f16_blocked(g.f16_flags, g.f16_need);
    if (blocked && g.Ap_b == nullptr) return;      // (the bf16 launch behind this one redoes the chunk)
    if (blocked && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(const_cast<int*>(g.f16_flags) + F16_FALLBACK_COUNT, 1);
    long long valid = g.num_edges ? (long long)(*g.num_edges) - g.row_begin : (long long)g.rows_valid;
    if (valid > g.rows) valid = g.rows;
    if (valid <= 0) return;
    const int tiles_mv = (int)((valid + TM - 1) / TM);
    const int nwg = g.tiles_n * tiles_mv;
    const int orig = blockIdx.x;
    if (orig >= nwg) return;
    // XCD x owns a contiguous run of tiles, row tiles fastest: the row tiles of one B panel are neighbours
    const int xcd = orig & 7, q = nwg >> 3, r8 = nwg & 7;
    const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
    const int bm = (tile % tiles_mv) * TM, bn = (tile / tiles_mv) * F16S_TN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;              // WM x 2 waves of 32 x 32
    const int l31 = lane & 31, h = lane >> 5;
    const int nkt = g.K / TK, nst = g.K / 32;
    const size_t tile_stride = (size_t)nkt * 2 << 12;     // bytes of one 128-row tile of an operand image
    const unsigned char* const a_panel = g.Ap + (size_t)(bm >> 7) * tile_stride;
    const unsigned char* const b_panel = g.Bp + (size_t)(bn >> 7) * tile_stride;
    const int panel_bytes = (int)tile_stride;
    const int a_row0 = (bm & 127) * 32;                   // this tile's rows inside a 4 KiB plane tile
    const int b_row0 = ((bn >> 6) & 1) * 2048;
    // piece wave + t * WAVES of a stage lands at that many KiB of the slot.  t = 0, 1: A — run (k-step, plane)
    // q / WM of 4 KiB in HBM, of which this tile's rows are WM KiB; t >= 2: B — run b >> 1, the tile's 64 columns
    // are one 2 KiB half of it
    unsigned voff[PPW];
#pragma unroll
    for (int t = 0; t < PPW; ++t) {
        if (t < 2) {
            const int qa = wave + t * WAVES;
            voff[t] = (unsigned)((qa / WM) * 4096 + a_row0 + (qa % WM) * 1024 + lane * 16);
        } else {
            const int b = wave + (t - 2) * WAVES;
            voff[t] = (unsigned)((b >> 1) * 4096 + b_row0 + (b & 1) * 1024 + lane * 16);
        }
    }
#define MDNO_DMA_STAGE(ST, SLOT)                                                                         \
    {                                                                                                    \
        const unsigned ko = (unsigned)(ST) * F16_TILE_BYTES;                                             \
        lds_u8* ldst = (lds_u8*)(lds + (SLOT) * STAGE + wave * 1024);                                    \
        _Pragma("unroll") for (int t = 0; t < PPW; ++t)                                                  \
            dma_piece_buffer(t < 2 ? a_panel : b_panel, panel_bytes, ldst + t * WAVES * 1024, voff[t], ko); \
    }
    const int hsw = (h ^ ((l31 >> 3) & 1)) << 4;
    const int a_rd = (wm * 32 + l31) * 32 + hsw;                          // + (ks * 2 + p) * WM KiB
    const int b_rd = A_BYTES + (wn * 32 + l31) * 32 + hsw;                // + (ks * 2 + p) * 2 KiB

    f32x16 acc, accx;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[e] = 0.f; accx[e] = 0.f; }
    float bv = 0.f, us = 1.f;
    if (g.bias) bv = g.bias[bn + wn * 32 + l31];
    if (g.b_unscale) us = g.b_unscale[bn + wn * 32 + l31];
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bv), "+v"(us));   // the counted waits below must see DMA pieces only

    if (blocked) {
        // An operand is out of fp16 range (or too small for it): the same tile from the three-plane bf16
        // images, six products per k-step in gemm_split_bf16_kernel's order.  A rare path (exploding
        // activations of an untrained net): one k-step at a time, no ring.
        us = 1.f;
        const size_t tile_stride_b = (size_t)nkt * 3 << 12;
        const unsigned char* a_src = g.Ap_b + (size_t)(bm >> 7) * tile_stride_b + a_row0 + lane * 16;
        const unsigned char* b_src = g.Bp_b + (size_t)(bn >> 7) * tile_stride_b + b_row0 + lane * 16;
        const int a_rb = (wm * 32 + l31) * 32 + hsw, b_rb = 3 * WM * 1024 + (wn * 32 + l31) * 32 + hsw;
        for (int kt = 0; kt < nkt; ++kt) {
            __syncthreads();      // the previous k-step's fragment reads are done
            for (int qq = wave; qq < 3 * WM + 6; qq += WAVES) {      // A: three planes of WM KiB, B: three 2 KiB halves
                const unsigned char* src = qq < 3 * WM
                                               ? a_src + (size_t)kt * 12288 + (qq / WM) * 4096 + (qq % WM) * 1024
                                               : b_src + (size_t)kt * 12288 + ((qq - 3 * WM) >> 1) * 4096 + ((qq - 3 * WM) & 1) * 1024;
                __builtin_amdgcn_global_load_lds((glb_u8*)src, (lds_u8*)(lds + qq * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            bf16x8 a[3], b[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[p] = *reinterpret_cast<const bf16x8*>(lds + p * (WM * 1024) + a_rb);
                b[p] = *reinterpret_cast<const bf16x8*>(lds + p * 2048 + b_rb);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < F16S_RING - 1; ++t)
        if (!blocked && t < nst) MDNO_DMA_STAGE(t, t)
    int slot = 0, slot_in = F16S_RING - 1;
    for (int st = 0; st < (blocked ? 0 : nst); ++st) {
        // stages still in flight behind stage st: min(RING - 2, nst - 1 - st) groups of PPW pieces
        const int behind = nst - 1 - st;
        if (behind >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PPW) : "memory");
        else if (behind == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PPW) : "memory");
        else if (behind == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
        else if (behind == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // everybody's pieces of stage st have landed, and nobody still reads the slot multiplied at st - 1
        // (a wave's fragment reads feed its MFMAs, so they have returned before it gets here)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (st + F16S_RING - 1 < nst) MDNO_DMA_STAGE(st + F16S_RING - 1, slot_in)
        const unsigned char* sb = lds + slot * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const f16x8 a0 = *reinterpret_cast<const f16x8*>(sb + (ks * 2) * (WM * 1024) + a_rd);
            const f16x8 a1 = *reinterpret_cast<const f16x8*>(sb + (ks * 2 + 1) * (WM * 1024) + a_rd);
            const f16x8 b0 = *reinterpret_cast<const f16x8*>(sb + (ks * 2) * 2048 + b_rd);
            const f16x8 b1 = *reinterpret_cast<const f16x8*>(sb + (ks * 2 + 1) * 2048 + b_rd);
            accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, accx, 0, 0, 0);
            accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, accx, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
        }
        slot = slot == F16S_RING - 1 ? 0 : slot + 1;
        slot_in = slot_in == F16S_RING - 1 ? 0 : slot_in + 1;
    }
#undef MDNO_DMA_STAGE

    float ua[16];      // (fetched before the first store: see gemm_split_f16_kernel)
#pragma unroll
    for (int e = 0; e < 16; ++e)
        ua[e] = (g.a_unscale && !blocked) ? g.a_unscale[bm + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h] : 1.f;
    bool bad = false, seen = false;
    const int n = bn + wn * 32 + l31;
    if (OUT == 4) {
        // A lane holds ONE column of 16 rows: stored from the accumulators, the up to five plane images would take 80
        // two-byte stores per lane.  Instead the wave turns its 32 x 32 tile through a private LDS patch (the ring is
        // idle once everybody has left the K loop) and each lane converts and stores 8 adjacent columns of a row:
        // 16 B per plane and store.  Rows of 40 floats: the two half-waves (rows r, r + 4) hit disjoint banks.
        __syncthreads();
        float* patch = reinterpret_cast<float*>(lds) + wave * (32 * 40);
#pragma unroll
        for (int e = 0; e < 16; ++e)
            patch[((e & 3) + 8 * (e >> 2) + 4 * h) * 40 + l31] = relu_f((acc[e] + accx[e] * F16_LO_UNSCALE) * us * ua[e] + bv);
        // (LDS operations of one wave execute in order: no wait between its writes and its reads)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int row = pr * 16 + (lane >> 2), ch = lane & 3;
            const int m = bm + wm * 32 + row, n0 = bn + wn * 32 + ch * 8;
            const float4 v0 = *reinterpret_cast<const float4*>(patch + row * 40 + ch * 8);
            const float4 v1 = *reinterpret_cast<const float4*>(patch + row * 40 + ch * 8 + 4);
            if (m < valid) {
                const float rv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                _Float16 oh[2][8];
                __bf16 ob[3][8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    bad |= !(rv[j] < F16_MAX);
                    seen |= rv[j] >= F16_ACT_MIN;
                    split2h(rv[j], oh[0][j], oh[1][j]);
                    split3(rv[j], ob[0][j], ob[1][j], ob[2][j]);
                }
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    *reinterpret_cast<uint4*>(g.Cp + tiled_off2(m, n0, g.N >> 4, p)) = *reinterpret_cast<const uint4*>(oh[p]);
                if (g.Cp_b != nullptr) {      // and the bf16 image, for a consumer that finds a flag up
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        *reinterpret_cast<uint4*>(g.Cp_b + tiled_off(m, n0, g.N >> 4, p)) = *reinterpret_cast<const uint4*>(ob[p]);
                }
            }
        }
        if (bad) atomicOr(const_cast<int*>(g.f16_flags) + 1, 1);
        if (seen) const_cast<int*>(g.f16_flags)[3] = 1;
        return;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = bm + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m < valid) {
            const float v = (acc[e] + accx[e] * F16_LO_UNSCALE) * us * ua[e] + bv;
            g.C[(size_t)m * g.N + n] = OUT == 3 ? relu_f(v) : v;
        }
    }
}

// few rows: the 256-row kernel would leave more than half of the CUs without a tile
constexpr int F16S_MAX_BIG_TILES = 128;

template <int OUT, int WM>
int launch_split_f16_gemm_small_wm(SplitGemmArgs g, hipStream_t s) {
    static std::atomic<unsigned long long> lds_raised{0};     // one per <OUT, WM> instantiation
    MDNO_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(&gemm_split_f16_small_kernel<OUT, WM>), f16s_lds_bytes(WM), lds_raised));
    g.tiles_n = g.N / F16S_TN;
    g.tiles_m = g.rows / (32 * WM);
    hipLaunchKernelGGL((gemm_split_f16_small_kernel<OUT, WM>), dim3(g.tiles_n * g.tiles_m), dim3(128 * WM), f16s_lds_bytes(WM), s, g);
    return check_launch("split-f16 GEMM (few rows)");
}

// the tallest tile that still gives the launch (by its row capacity) about one workgroup per CU
template <int OUT>
int launch_split_f16_gemm_small(const SplitGemmArgs& g, hipStream_t s) {
    const long long col_tiles = g.N / F16S_TN;
    if (col_tiles * (g.rows / 128) >= 256) return launch_split_f16_gemm_small_wm<OUT, 4>(g, s);
    if (col_tiles * (g.rows / 64) >= 256) return launch_split_f16_gemm_small_wm<OUT, 2>(g, s);
    return launch_split_f16_gemm_small_wm<OUT, 1>(g, s);
}

template <int OUT, int MI>
int launch_split_f16_gemm(SplitGemmArgs g, hipStream_t s) {
    constexpr int lds_bytes = F16_RING * F16_STAGE_BYTES;     // 147,456 B: one workgroup per CU
    static std::atomic<unsigned long long> lds_raised{0};
    MDNO_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(&gemm_split_f16_kernel<OUT, MI>), lds_bytes, lds_raised));
    MDNO_REQUIRE(g.K % 32 == 0 && g.N % TN == 0 && g.rows % F16_TM == 0, MDNO_EUNSUPPORTED,
                 "split-f16 GEMM: rows=%d N=%d K=%d", g.rows, g.N, g.K);
    if constexpr (OUT != 2) {      // (the factored conv's k-tiled H is written by the 256-row kernel only)
        if ((g.N / TN) * (g.rows / F16_TM) <= F16S_MAX_BIG_TILES) return launch_split_f16_gemm_small<OUT>(g, s);
    }
    g.tiles_n = g.N / TN;
    g.tiles_m = g.rows / F16_TM;
    if (g.a_unscale != nullptr) {
        static std::atomic<unsigned long long> lds_raised_rs{0};
        MDNO_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(&gemm_split_f16_kernel<OUT, MI, true>), lds_bytes, lds_raised_rs));
        hipLaunchKernelGGL((gemm_split_f16_kernel<OUT, MI, true>), dim3(g.tiles_n * g.tiles_m), dim3(1024 / MI), lds_bytes, s, g);
        return check_launch("split-f16 GEMM");
    }
    hipLaunchKernelGGL((gemm_split_f16_kernel<OUT, MI>), dim3(g.tiles_n * g.tiles_m), dim3(1024 / MI), lds_bytes, s, g);
    return check_launch("split-f16 GEMM");
}

template <int TM, int OUT>
int launch_split_gemm_tm(SplitGemmArgs g, hipStream_t s) {
    constexpr int lds_bytes = 2 * stage_bytes(TM);
    static std::atomic<unsigned long long> lds_raised{0};   // one per <TM, OUT> instantiation
    MDNO_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(&gemm_split_bf16_kernel<TM, OUT>), lds_bytes, lds_raised));
    g.tiles_n = g.N / TN;
    g.tiles_m = g.rows / TM;
    if (g.f16_flags != nullptr) {      // a fallback launch of SPLIT_F16: small grid, its workgroups walk the tiles
        static std::atomic<unsigned long long> lds_raised_f{0};
        MDNO_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(&gemm_split_bf16_kernel<TM, OUT, true>), lds_bytes, lds_raised_f));
        const int tiles = g.tiles_n * g.tiles_m;
        hipLaunchKernelGGL((gemm_split_bf16_kernel<TM, OUT, true>), dim3(tiles < kFallbackGemmGrid ? tiles : kFallbackGemmGrid),
                           dim3(TM * 2), lds_bytes, s, g);
        return check_launch("split-bf16 GEMM (fallback)");
    }
    hipLaunchKernelGGL((gemm_split_bf16_kernel<TM, OUT>), dim3(g.tiles_n * g.tiles_m), dim3(TM * 2), lds_bytes,
                       s, g);
    return check_launch("split-bf16 GEMM");
}

// 256-row tiles need 1.5x fewer staged bytes per MFMA; 128-row tiles give twice the workgroups.
// The wide last layer (N = Cin*Cout = 4096) has tiles to spare, the k x k middle layer does not.
template <int OUT>
int launch_split_gemm(const SplitGemmArgs& g, int kid, hipStream_t s) {
    TimedSection ts(kid, s);
    return g.N >= 2048 ? launch_split_gemm_tm<256, OUT>(g, s) : launch_split_gemm_tm<128, OUT>(g, s);
}

}  // namespace

namespace {
__global__ void fill_ints_kernel(int* p, int n, int v) {
    if ((int)threadIdx.x < n) p[threadIdx.x] = v;
}
}  // namespace

// n <= 256 ints set by a kernel (not a memset node: a captured step stays a plain chain of kernel nodes)
int fill_ints(int* p, int n, int value, hipStream_t s) {
    MDNO_REQUIRE(p && n > 0 && n <= 256, MDNO_EINVAL, "fill_ints: n=%d", n);
    hipLaunchKernelGGL(fill_ints_kernel, dim3(1), dim3(256), 0, s, p, n, value);
    return check_launch("fill_ints_kernel");
}

// ---------------------------------------------------------------- generic pieces
size_t split_planes_bytes(long long rows, int K) {
    return (size_t)3 * ((rows + 255) / 256 * 256) * K * sizeof(__bf16);   // whole 256-row GEMM tiles
}

int split_planes(const float* a, int rows, int K, void* planes, hipStream_t s) {
    MDNO_REQUIRE(K % 16 == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0, MDNO_EINVAL, "split_planes: K=%d", K);
    const long long chunks = (long long)rows * (K / 8);
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, s, a, rows, K,
                       static_cast<unsigned char*>(planes));
    return check_launch("split_planes_kernel");
}

size_t split_planes_f16_bytes(long long rows, int K) {
    return (size_t)2 * ((rows + 255) / 256 * 256) * K * sizeof(_Float16);
}

int split_planes_f16(const float* a, int rows, int K, void* planes, float* unscale, int* range_flag, hipStream_t s) {
    MDNO_REQUIRE(K % 16 == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0 && range_flag && unscale, MDNO_EINVAL,
                 "split_planes_f16: K=%d", K);
    hipLaunchKernelGGL(split_planes_f16_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, a, rows, K,
                       static_cast<unsigned char*>(planes), unscale, range_flag);
    return check_launch("split_planes_f16_kernel");
}

// act(A . W^T + b) for the training ops: A [rows,K] and W [N,K] are split here, every call
size_t split_linear_workspace_bytes(long long rows, int N, int K) {
    return align_up(split_planes_bytes(rows, K), 256) + align_up(split_planes_bytes(N, K), 256);
}

bool split_linear_supported(long long rows, int N, int K) {
    return K % 32 == 0 && N % TN == 0 && rows > 0 && rows < (1ll << 31) - 256;
}

int split_linear(const float* a, const float* w, const float* bias, long long rows, int N, int K, int relu, float* c,
                 void* workspace, hipStream_t s) {
    Carver cv(workspace);
    unsigned char* ap = reinterpret_cast<unsigned char*>(cv.take<char>(split_planes_bytes(rows, K)));
    unsigned char* wp = reinterpret_cast<unsigned char*>(cv.take<char>(split_planes_bytes(N, K)));
    MDNO_TRY(split_planes(a, (int)rows, K, ap, s));
    MDNO_TRY(split_planes(w, N, K, wp, s));
    SplitGemmArgs g{ap, wp, bias, c, nullptr, nullptr, 0, (int)((rows + 255) / 256 * 256), N, K, 0, 0, (int)rows, 0};
    if (relu) return N >= 2048 ? launch_split_gemm_tm<256, 3>(g, s) : launch_split_gemm_tm<128, 3>(g, s);
    return N >= 2048 ? launch_split_gemm_tm<256, 0>(g, s) : launch_split_gemm_tm<128, 0>(g, s);
}

// The same product on two fp16 planes per operand (three plane products instead of six): every row of A and of W
// is multiplied by its own power of two first (f16_row_scale: the row's largest entry lands in [2^13, 2^14), so
// nothing leaves fp16's range whatever the magnitudes — gradients of 1e-9 as well as activations of 1e6) and the
// output element (m, n) is multiplied back by both in the epilogue, exactly.  Inside a row, entries more than
// ~2^29 below the row's largest lose relative accuracy (split_layout.h); their share of a dot product is below
// fp32 rounding of the sum.
struct SplitF16Ws {
    unsigned char *ap, *wp;
    float *aus, *wus;
    int* flag;
    size_t total;
};

static SplitF16Ws carve_split_f16(void* ws, long long rows, int N, int K) {
    SplitF16Ws w{};
    Carver cv(ws);
    const long long rows_pad = (rows + 255) / 256 * 256;
    w.ap = reinterpret_cast<unsigned char*>(cv.take<char>(split_planes_f16_bytes(rows, K)));
    w.wp = reinterpret_cast<unsigned char*>(cv.take<char>(split_planes_f16_bytes(N, K)));
    w.aus = cv.take<float>((size_t)rows_pad);
    w.wus = cv.take<float>((size_t)N);
    w.flag = cv.take<int>(64);
    w.total = cv.used();
    return w;
}

size_t split_linear_f16_workspace_bytes(long long rows, int N, int K) { return carve_split_f16(nullptr, rows, N, K).total; }

int split_linear_f16(const float* a, const float* w, const float* bias, long long rows, int N, int K, int relu, float* c,
                     void* workspace, hipStream_t s) {
    const SplitF16Ws sw = carve_split_f16(workspace, rows, N, K);
    // (the flag word collects "non-finite input": such rows give non-finite outputs, as an fp32 product would)
    MDNO_TRY(split_planes_f16(a, (int)rows, K, sw.ap, sw.aus, sw.flag, s));
    MDNO_TRY(split_planes_f16(w, N, K, sw.wp, sw.wus, sw.flag, s));
    SplitGemmArgs g{sw.ap, sw.wp, bias, c, nullptr, nullptr, 0, (int)((rows + 255) / 256 * 256), N, K, 0, 0, (int)rows, 0};
    g.b_unscale = sw.wus;
    g.a_unscale = sw.aus;
    return relu ? launch_split_f16_gemm<3, 2>(g, s) : launch_split_f16_gemm<0, 2>(g, s);
}

bool edge_mlp_split_supported(int ker_width, int out_dim) {
    return ker_width % 32 == 0 && ker_width % TN == 0 && out_dim % TN == 0;
}

// One layout for both entry points: [h1 planes][h2 planes][W1 planes][W2 planes][W1 fp16 planes][flags][W2 fp16 planes]
// [W1, W2 unscale][h1, h2 fp16 planes of a few-row launch]
struct SplitWs {
    unsigned char *h1p, *h2p, *w1p, *w2p, *w1h, *w2h;
    unsigned char *h1h, *h2h;   // few rows (both_gemms_small): the fp16 images beside the bf16 ones in h1p, h2p
    float *w1us, *w2us;      // per-row unscale factors of the fp16 weight images
    int* f16_flags;
    size_t total;
};

// Both GEMMs of the full edge-MLP on the few-rows kernel: then every kernel of the chain carries both operand
// images and picks one itself, and no fallback launch follows (at that size a launch costs more than a GEMM)
static bool both_gemms_small(int k, int out_dim, long long chunk) {
    const long long tm = chunk / F16_TM;
    return chunk % F16_TM == 0 && (k / TN) * tm <= F16S_MAX_BIG_TILES && (out_dim / TN) * tm <= F16S_MAX_BIG_TILES;
}

static SplitWs carve_split(void* ws, int k, int out_dim, long long chunk) {
    SplitWs w{};
    Carver cv(ws);
    w.h1p = reinterpret_cast<unsigned char*>(cv.take<__bf16>(3 * (size_t)chunk * k));
    w.h2p = reinterpret_cast<unsigned char*>(cv.take<__bf16>(3 * (size_t)chunk * k));
    w.w1p = reinterpret_cast<unsigned char*>(cv.take<__bf16>(3 * (size_t)k * k));
    w.w2p = reinterpret_cast<unsigned char*>(cv.take<__bf16>(3 * (size_t)out_dim * k));
    w.w1h = reinterpret_cast<unsigned char*>(cv.take<_Float16>(2 * (size_t)k * k));
    w.f16_flags = cv.take<int>(64);
    w.w2h = reinterpret_cast<unsigned char*>(cv.take<_Float16>(2 * (size_t)out_dim * k));
    w.w1us = cv.take<float>((size_t)k);
    w.w2us = cv.take<float>((size_t)out_dim);
    const size_t small = both_gemms_small(k, out_dim, chunk) ? 2 * (size_t)chunk * k : 0;
    w.h1h = reinterpret_cast<unsigned char*>(cv.take<_Float16>(small));
    w.h2h = reinterpret_cast<unsigned char*>(cv.take<_Float16>(small));
    w.total = cv.used();
    return w;
}

size_t edge_mlp_split_workspace_bytes(int ker_width, int out_dim, long long chunk) {
    return carve_split(nullptr, ker_width, out_dim, chunk).total;
}

// (the word behind the flags that counts the products redone on bf16 planes: activation flags + 7)
int* edge_mlp_split_activation_flags(void* workspace, int ker_width, int out_dim, long long chunk) {
    return carve_split(workspace, ker_width, out_dim, chunk).f16_flags + 1;
}

int edge_mlp_split(const float* frames, int frame, const int* t_dev, int rows_per_frame, const int* src,
                   const int* dst, const float* edge_attr, const int* perm, const int* num_edges,
                   long long edge_cap, long long chunk, int ker_in, int ker_width, int out_dim,
                   const EdgeMlpWeights& w, float* w_e, void* workspace, hipStream_t s, int phase_in, bool f16) {
    const int phase = phase_in & WP_PHASE_MASK;
    const bool flags_zeroed = (phase_in & WP_FLAGS_ZEROED) != 0;
    MDNO_REQUIRE(ker_in > 0 && ker_in <= MAX_F, MDNO_EUNSUPPORTED, "edge_mlp: ker_in=%d (1..%d)", ker_in, MAX_F);
    MDNO_REQUIRE(((reinterpret_cast<uintptr_t>(w.w1) | reinterpret_cast<uintptr_t>(w.w2)) & 15) == 0, MDNO_EINVAL,
                 "edge_mlp: weight pointers must be 16-byte aligned");
    const int k = ker_width;
    const SplitWs sw = carve_split(workspace, k, out_dim, chunk);
    unsigned char *h1p = sw.h1p, *h2p = sw.h2p, *w1p = sw.w1p, *w2p = sw.w2p;
    if (phase != WP_RUN_ONLY) {
        TimedSection ts(KID_EDGE_L0, s);
        const long long c1 = (long long)k * (k / 8), c2 = (long long)out_dim * (k / 8);
        hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((c1 + 255) / 256)), dim3(256), 0, s, w.w1, k, k, w1p);
        hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((c2 + 255) / 256)), dim3(256), 0, s, w.w2, out_dim, k,
                           w2p);
        if (f16) {   // fp16 images of W1, W2 + their range flag (flags[0]); the bf16 images serve the fallback
            MDNO_TRY(fill_ints(sw.f16_flags, 1, 0, s));
            MDNO_TRY(split_planes_f16(w.w1, k, k, sw.w1h, sw.w1us, sw.f16_flags, s));
            MDNO_TRY(split_planes_f16(w.w2, out_dim, k, sw.w2h, sw.w2us, sw.f16_flags, s));
        }
    }
    MDNO_TRY(check_launch("split_planes_kernel"));
    if (phase == WP_PREPARE_ONLY) return MDNO_OK;
    const float* pos_mode = edge_attr ? nullptr : frames;
    // activation flags of THIS forward (range, h1 seen, h2 seen)
    if (f16 && !flags_zeroed) MDNO_TRY(fill_ints(sw.f16_flags + 1, kEdgeMlpActivationFlags, 0, s));
    for (long long e0 = 0; e0 < edge_cap; e0 += chunk) {
        const int cnt = (int)((edge_cap - e0) < chunk ? (edge_cap - e0) : chunk);
        if (f16 && both_gemms_small(k, out_dim, chunk)) {
            // a few hundred rows: three launches — every kernel writes / finds both operand images and the
            // GEMMs take the bf16 ones themselves when a flag is up (gemm_split_f16_small_kernel)
            {
                TimedSection ts(KID_EDGE_L0, s);
                MDNO_TRY(launch_edge_l0_split(pos_mode, frame, t_dev, rows_per_frame, src, dst, edge_attr, perm, num_edges,
                                              e0, cnt, ker_in, k, w.w0, w.b0, h1p, s, true, sw.f16_flags, 0, sw.h1h));
            }
            {
                TimedSection ts(KID_GEMM_L1, s);
                SplitGemmArgs gh{sw.h1h, sw.w1h, w.b1, nullptr, sw.h2h, num_edges, e0, (int)chunk, k, k, 0, 0, 0, 0,
                                 sw.f16_flags, 1, sw.w1us, h1p, w1p, h2p};
                MDNO_TRY((launch_split_f16_gemm<4, 2>(gh, s)));
            }
            {
                TimedSection ts(KID_GEMM_L2, s);
                SplitGemmArgs gh{sw.h2h, sw.w2h, w.b2, w_e + (size_t)e0 * out_dim, nullptr, num_edges, e0, (int)chunk, out_dim,
                                 k, 0, 0, 0, 0, sw.f16_flags, 3, sw.w2us, h2p, w2p, nullptr};
                MDNO_TRY((launch_split_f16_gemm<0, 2>(gh, s)));
            }
            continue;
        }
        if (f16) {
            // SPLIT_F16: layer 0 -> fp16 planes, hidden layer -> fp16 planes of h2 (its epilogue checks the
            // range), last layer -> W_e; then the same chunk on the bf16 kernels, which exit at their first
            // instruction unless a range flag is up
            {
                TimedSection ts(KID_EDGE_L0, s);
                MDNO_TRY(launch_edge_l0_split(pos_mode, frame, t_dev, rows_per_frame, src, dst, edge_attr, perm, num_edges,
                                              e0, cnt, ker_in, k, w.w0, w.b0, h1p, s, true, sw.f16_flags));
            }
            {
                TimedSection ts(KID_GEMM_L1, s);   // reads h1 (need 1), writes h2 planes + their range / "seen" words
                SplitGemmArgs gh{h1p, sw.w1h, w.b1, nullptr, h2p, num_edges, e0, (int)chunk, k, k, 0, 0, 0, 0, sw.f16_flags,
                                 1, sw.w1us};
                MDNO_TRY((launch_split_f16_gemm<4, 2>(gh, s)));
            }
            {
                TimedSection ts(KID_GEMM_L2, s);   // reads h2 (need 3: h1 and h2 both fit fp16)
                SplitGemmArgs gh{h2p, sw.w2h, w.b2, w_e + (size_t)e0 * out_dim, nullptr, num_edges, e0, (int)chunk, out_dim,
                                 k, 0, 0, 0, 0, sw.f16_flags, 3, sw.w2us};
                MDNO_TRY((launch_split_f16_gemm<0, 2>(gh, s)));
                // the bf16 trio runs iff the fp16 chain did not go all the way through (same test: need 3)
                MDNO_TRY(launch_edge_l0_split(pos_mode, frame, t_dev, rows_per_frame, src, dst, edge_attr, perm, num_edges,
                                              e0, cnt, ker_in, k, w.w0, w.b0, h1p, s, false, sw.f16_flags, 3));
                SplitGemmArgs g1{h1p, w1p, w.b1, nullptr, h2p, num_edges, e0, (int)chunk, k, k, 0, 0, 0, 0, sw.f16_flags, 3};
                MDNO_TRY((launch_split_gemm_tm<128, 1>(g1, s)));
                SplitGemmArgs g2{h2p, w2p, w.b2, w_e + (size_t)e0 * out_dim, nullptr, num_edges, e0, (int)chunk, out_dim, k,
                                 0, 0, 0, 0, sw.f16_flags, 3};
                if (out_dim >= 2048) MDNO_TRY((launch_split_gemm_tm<256, 0>(g2, s)));
                else MDNO_TRY((launch_split_gemm_tm<128, 0>(g2, s)));
            }
            continue;
        }
        {
            TimedSection ts(KID_EDGE_L0, s);
            MDNO_TRY(launch_edge_l0_split(pos_mode, frame, t_dev, rows_per_frame, src, dst, edge_attr, perm, num_edges,
                                          e0, cnt, ker_in, k, w.w0, w.b0, h1p, s));
        }
        SplitGemmArgs g1{h1p, w1p, w.b1, nullptr, h2p, num_edges, e0, (int)chunk, k, k, 0, 0, 0, 0};
        MDNO_TRY(launch_split_gemm<1>(g1, KID_GEMM_L1, s));
        SplitGemmArgs g2{h2p, w2p, w.b2, w_e + (size_t)e0 * out_dim, nullptr, num_edges, e0, (int)chunk, out_dim, k,
                         0, 0, 0, 0};
        MDNO_TRY(launch_split_gemm<0>(g2, KID_GEMM_L2, s));
    }
    return MDNO_OK;
}

int edge_mlp_split_hidden(const float* frames, int frame, const int* t_dev, int rows_per_frame, const int* src,
                          const int* dst, const float* edge_attr, const int* perm, const int* num_edges,
                          long long edge_cap, long long chunk, int ker_in, int ker_width, const EdgeMlpWeights& w,
                          float* h_out, void* workspace, hipStream_t s, int phase_in, bool f16) {
    const int phase = phase_in & WP_PHASE_MASK;
    const bool flags_zeroed = (phase_in & WP_FLAGS_ZEROED) != 0;
    MDNO_REQUIRE(ker_in > 0 && ker_in <= MAX_F, MDNO_EUNSUPPORTED, "edge_mlp: ker_in=%d (1..%d)", ker_in, MAX_F);
    MDNO_REQUIRE((reinterpret_cast<uintptr_t>(w.w1) & 15) == 0, MDNO_EINVAL,
                 "edge_mlp: weight pointers must be 16-byte aligned");
    const int k = ker_width;
    const SplitWs sw = carve_split(workspace, k, k, chunk);      // (same layout as the full MLP with out_dim = k)
    unsigned char *h1p = sw.h1p, *w1p = sw.w1p;
    if (phase != WP_RUN_ONLY) {
        TimedSection ts(KID_EDGE_L0, s);
        const long long c1 = (long long)k * (k / 8);
        hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((c1 + 255) / 256)), dim3(256), 0, s, w.w1, k, k, w1p);
        if (f16) {   // fp16 image of W1 + its range flag (flags[0]); the bf16 image above serves the fallback
            MDNO_TRY(fill_ints(sw.f16_flags, 1, 0, s));
            MDNO_TRY(split_planes_f16(w.w1, k, k, sw.w1h, sw.w1us, sw.f16_flags, s));
        }
    }
    MDNO_TRY(check_launch("split_planes_kernel"));
    if (phase == WP_PREPARE_ONLY) return MDNO_OK;
    const float* pos_mode = edge_attr ? nullptr : frames;
    // activation flags of THIS forward (range, h1 seen, h2 seen)
    if (f16 && !flags_zeroed) MDNO_TRY(fill_ints(sw.f16_flags + 1, kEdgeMlpActivationFlags, 0, s));
    for (long long e0 = 0; e0 < edge_cap; e0 += chunk) {
        const int cnt = (int)((edge_cap - e0) < chunk ? (edge_cap - e0) : chunk);
        float* out = h_out + (size_t)e0 * k;      // chunk % 128 == 0: the k-tiled tile index continues across chunks
        if (f16) {
            {
                TimedSection ts(KID_EDGE_L0, s);
                MDNO_TRY(launch_edge_l0_split(pos_mode, frame, t_dev, rows_per_frame, src, dst, edge_attr, perm, num_edges,
                                              e0, cnt, ker_in, k, w.w0, w.b0, h1p, s, true, sw.f16_flags));
            }
            TimedSection ts(KID_GEMM_L1, s);
            SplitGemmArgs gh{h1p, sw.w1h, w.b1, out, nullptr, num_edges, e0, (int)chunk, k, k, 0, 0, 0, 0, sw.f16_flags, 1,
                             sw.w1us};
            MDNO_TRY((launch_split_f16_gemm<2, 2>(gh, s)));
            // the same chunk on the bf16 kernels: both exit at their first instruction unless a range flag is up
            MDNO_TRY(launch_edge_l0_split(pos_mode, frame, t_dev, rows_per_frame, src, dst, edge_attr, perm, num_edges,
                                          e0, cnt, ker_in, k, w.w0, w.b0, h1p, s, false, sw.f16_flags, 1));
            SplitGemmArgs g1{h1p, w1p, w.b1, out, nullptr, num_edges, e0, (int)chunk, k, k, 0, 0, 0, 0, sw.f16_flags, 1};
            MDNO_TRY((launch_split_gemm_tm<128, 2>(g1, s)));
            continue;
        }
        {
            TimedSection ts(KID_EDGE_L0, s);
            MDNO_TRY(launch_edge_l0_split(pos_mode, frame, t_dev, rows_per_frame, src, dst, edge_attr, perm, num_edges,
                                          e0, cnt, ker_in, k, w.w0, w.b0, h1p, s));
        }
        SplitGemmArgs g1{h1p, w1p, w.b1, out, nullptr, num_edges, e0, (int)chunk, k, k, 0, 0, 0, 0};
        MDNO_TRY(launch_split_gemm<2>(g1, KID_GEMM_L1, s));
    }
    return MDNO_OK;
}

}  // namespace mdno
