// K3-K6: gather -> per-edge matvec -> segmented aggregation -> root/bias/ReLU, one launch.
//
// Replaces NNConv_old.forward/message/update (graph_kernel.py:194-209) plus torch_geometric's
// index_select gather and scatter-mean, i.e. per conv application
//     y[r] = act( aggr_{p in row r} x[src[p]] . W_e[p]  +  x[r] . root + bias ).
//
// Roofline: HBM.  W_e is read exactly once (Cin*Cout*4 B per edge, 16 KiB at 64x64) at
// 2*Cin*Cout flop per edge = 0.5 flop/B; x (R*Cin*4 B) is L2-resident and re-gathered from cache.
// Algorithmic bytes per launch: E*(Cin*Cout*4 + 4) + (R+1)*4 + 2*R*C*4.
//
// Layout / mapping (64x64 specialisation): edges are sorted by destination, so a row's W_e block
// is ONE contiguous run of deg*16 KiB.  One workgroup owns one destination row; its 4 (many rows)
// or 16 (few rows) waves take the row's edges round-robin (16 summation chains either way).  Inside a wave, lane l = (g, q) with g = l>>4, q = l&15
// accumulates output columns 4q..4q+3 over input rows 16g..16g+15: every wave-instruction is a
// 16 B/lane load covering four whole 256-B rows of W_e[p] (fully coalesced), 16 such loads per
// edge, 64 FMAs per lane.  Partial sums stay in registers across ALL edges of the row; the
// reduction over g (2 shuffles) and over the waves (LDS) happens once per row, in a fixed order,
// so results are bitwise reproducible (no atomics).  The root term x[r].root is folded in as one
// more "edge" (weight matrix = root) with its own accumulator.
//
// Measured against a pure streaming read of the same bytes (scripts/micro/stream_read.hip: 1 GB in
// 178 us = 6.04 TB/s incl. ~7 us of ramp): 993 MB in 187 us at N=504 (92 % of that), 6.1 TB/s at 8
// members.  Splitting rows into 16-edge segments balanced over waves (two-pass, partial sums) was
// built and measured: no gain (192 us) — the kernel is bandwidth-, not balance-limited.
#include "kernels.h"

namespace mdno {
namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// W_e is read exactly once per application: stream it past the caches (global_load ... nt)
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_stream(const float* p) {
    const f32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}

__device__ __forceinline__ void fma4(float4& a, float s, const float4& w) {
    a.x = fmaf(s, w.x, a.x);
    a.y = fmaf(s, w.y, a.y);
    a.z = fmaf(s, w.z, a.z);
    a.w = fmaf(s, w.w, a.w);
}

// acc += x[16g..16g+15] . Wblk[16g..16g+15][4q..4q+3].  STREAM: W is read once per application and is far larger
// than the caches (nt loads leave them to x); !STREAM: all of W_e fits the L2s (a short chain: 330 edges = 5.4 MB
// over 8 x 4 MB), the 2 x depth applications of a forward re-read it, and a row's 12 x 16 KiB reach its ONE CU at
// the L2's 66-73 GB/s per CU instead of the Infinity Cache's 33 (MI355X_MICROARCH.md, gather rates): the
// application is bound by exactly that
template <bool STREAM = true>
__device__ __forceinline__ void edge_accumulate64(float4& acc, const float* __restrict__ xrow,
                                                  const float* __restrict__ wmat, int g, int q) {
    // the 16 KiB of W first: their address does not wait for src[p], which the x row's does
    const float* wp = wmat + (16 * g) * 64 + 4 * q;
    float4 w[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) w[r] = STREAM ? ld4_stream(wp + r * 64) : ld4(wp + r * 64);
    const float* xp = xrow + 16 * g;
    const float4 x0 = ld4(xp), x1 = ld4(xp + 4), x2 = ld4(xp + 8), x3 = ld4(xp + 12);
    // all twenty loads in flight before the first FMA waits for one (left alone, the scheduler waits for the
    // x row after nine of them and issues the rest behind that round trip)
    __builtin_amdgcn_sched_barrier(0);
    fma4(acc, x0.x, w[0]);  fma4(acc, x0.y, w[1]);  fma4(acc, x0.z, w[2]);  fma4(acc, x0.w, w[3]);
    fma4(acc, x1.x, w[4]);  fma4(acc, x1.y, w[5]);  fma4(acc, x1.z, w[6]);  fma4(acc, x1.w, w[7]);
    fma4(acc, x2.x, w[8]);  fma4(acc, x2.y, w[9]);  fma4(acc, x2.z, w[10]); fma4(acc, x2.w, w[11]);
    fma4(acc, x3.x, w[12]); fma4(acc, x3.y, w[13]); fma4(acc, x3.z, w[14]); fma4(acc, x3.w, w[15]);
}

__device__ __forceinline__ float4 reduce_over_g(float4 a) {
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
        a.x += __shfl_xor(a.x, o);
        a.y += __shfl_xor(a.y, o);
        a.z += __shfl_xor(a.z, o);
        a.w += __shfl_xor(a.w, o);
    }
    return a;
}

// A row's edges are dealt to CHAINS = 16 summation chains (edge i of the row -> chain i % 16) whatever
// the launch shape: with 16 waves a wave owns one chain, with 4 waves it owns chains w, w+4, w+8, w+12
// (one accumulator each).  The chains are then added in chain order, so the 4- and the 16-wave launch
// give the same bits and a row's result does not depend on how many rows it is batched with.
constexpr int CHAINS = 16;

template <int WAVES, bool STREAM = true>
__global__ __launch_bounds__(WAVES * 64) void nnconv64_row_kernel(
    const float* __restrict__ x, const int* __restrict__ row_ptr, const int* __restrict__ src,
    const float* __restrict__ w_e, const float* __restrict__ root, const float* __restrict__ bias,
    float* __restrict__ y, int num_rows, int aggr, int relu, FcTail fc) {
    constexpr int CPW = CHAINS / WAVES;   // chains per wave
    __shared__ float red[CHAINS][64];
    __shared__ float rootred[64];
    const int row = blockIdx.x;
    if (row >= num_rows) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int g = lane >> 4, q = lane & 15;
    const int beg = row_ptr[row], end = row_ptr[row + 1];
    const int deg = end - beg;

    // one chain after the other (a single accumulator and one edge's 16 loads in flight per wave, as
    // in the one-chain-per-wave shape; interleaving the chains made the compiler hoist the loads of
    // all CPW edges: 198 VGPRs, 2 waves per SIMD)
#pragma unroll
    for (int u = 0; u < CPW; ++u) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (aggr != MDNO_AGGR_MAX) {
            for (int p = beg + wave + u * WAVES; p < end; p += CHAINS)
                edge_accumulate64<STREAM>(acc, x + (size_t)src[p] * 64, w_e + (size_t)p * 4096, g, q);
            acc = reduce_over_g(acc);
        } else {      // every message in full, then the running maximum of the chain (-inf: a chain without edges)
            acc = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            for (int p = beg + wave + u * WAVES; p < end; p += CHAINS) {
                float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
                edge_accumulate64<STREAM>(m, x + (size_t)src[p] * 64, w_e + (size_t)p * 4096, g, q);
                m = reduce_over_g(m);
                acc = make_float4(fmaxf(acc.x, m.x), fmaxf(acc.y, m.y), fmaxf(acc.z, m.z), fmaxf(acc.w, m.w));
            }
        }
        if (lane < 16) *reinterpret_cast<float4*>(&red[wave + u * WAVES][4 * lane]) = acc;
    }
    // the root term x[r].root is one more "edge" with its own accumulator, taken by the wave that
    // got the fewest edges
    const bool root_wave = root != nullptr && wave == (deg % WAVES);
    float4 racc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (root_wave) edge_accumulate64<false>(racc, x + (size_t)row * 64, root, g, q);      // (every row reads root: cached)

    racc = reduce_over_g(racc);
    if (root_wave && lane < 16) *reinterpret_cast<float4*>(&rootred[4 * lane]) = racc;
    __syncthreads();
    if (tid < 64) {
        float s = 0.f;
        if (aggr != MDNO_AGGR_MAX) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) s += red[c][tid];   // fixed order: bitwise reproducible
        } else if (deg > 0) {
            s = red[0][tid];
#pragma unroll
            for (int c = 1; c < CHAINS; ++c) s = fmaxf(s, red[c][tid]);
        }
        if (aggr == MDNO_AGGR_MEAN) s = s / (float)(deg > 1 ? deg : 1);
        if (root != nullptr) s += rootred[tid];
        if (bias != nullptr) s += bias[tid];
        if (relu) s = relu_f(s);
        y[(size_t)row * 64 + tid] = s;
        if (fc.out != nullptr) {
            // the output layer on this row, as fc_out_kernel computes it (one product per lane, xor tree, bias)
            const int step = fc.t_dev ? *fc.t_dev : 0;
            float* o_ptr = fc.out + ((size_t)(fc.t_out + step) * num_rows + row) * fc.out_width;
            for (int o = 0; o < fc.out_width; ++o) {
                float v = fmaf(s, fc.w[(size_t)o * 64 + tid], 0.f);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
                if (tid == 0) o_ptr[o] = v + (fc.b ? fc.b[o] : 0.f);
            }
            // (this wave has its copy of the step counter; the last workgroup through here moves it on)
            if (fc.step.done != nullptr && tid == 0 && atomicAdd(fc.step.done, 1) == num_rows - 1) {
                *fc.step.done = 0;
                if (fc.step.edges_per_step) fc.step.edges_per_step[step] = *fc.step.num_edges;
                *fc.step.t_dev = step + 1;
            }
        }
    }
}

// Short chains (all of W_e in the L2s, a few dozen rows): what bounds an application is the 12 x 16 KiB of a row
// arriving at ONE CU (66-73 GB/s from L2).  So SPLIT workgroups share a row by OUTPUT COLUMNS: each reads its
// 64 / SPLIT columns of every edge's matrix (whole 128-B or 64-B lines of the 256-B rows), 64 / SPLIT lanes per
// edge and SPLIT edges per wave, and needs nothing from its partners — an output column's sum is formed exactly
// as in nnconv64_row_kernel (lane (g, q): input rows 16g.., columns 4q..; the same 16 chains, the same g tree,
// chains added in order), so the split does not change a bit.  Only the output layer riding on the LAST
// application needs the whole row: the workgroup that finishes a row last (agent-scope counter) reads it back
// and applies fc2 with fc_out_kernel's arithmetic.  Workgroups b, b+8, ... (one XCD) hold the parts of one row.
template <int SPLIT, bool STREAM>
__global__ __launch_bounds__(1024 / SPLIT) void nnconv64_colsplit_kernel(
    const float* __restrict__ x, const int* __restrict__ row_ptr, const int* __restrict__ src,
    const float* __restrict__ w_e, const float* __restrict__ root, const float* __restrict__ bias,
    float* __restrict__ y, int num_rows, int aggr, int relu, FcTail fc) {
    constexpr int QN = 16 / SPLIT;          // q values (groups of 4 columns) per workgroup
    constexpr int LPE = 4 * QN;             // lanes per edge: (g, ql)
    constexpr int COLS = 4 * QN;            // output columns of this workgroup
    __shared__ float red[CHAINS][COLS];
    __shared__ float rootred[COLS];
    const unsigned b = blockIdx.x, xcd = b & 7u, i = b >> 3;
    const int part = (int)(i % SPLIT), row = (int)(i / SPLIT) * 8 + (int)xcd;
    if (row >= num_rows) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int sub = lane / LPE, l = lane % LPE;
    const int g = l / QN, q = part * QN + l % QN;
    const int chain = wave * SPLIT + sub;   // 0..15
    const int beg = row_ptr[row], end = row_ptr[row + 1];
    const int deg = end - beg;
    auto reduce_g = [](float4 a) {          // the g tree of reduce_over_g: g's low bit first, then its high bit
#pragma unroll
        for (int o = QN; o <= 2 * QN; o <<= 1) {
            a.x += __shfl_xor(a.x, o);
            a.y += __shfl_xor(a.y, o);
            a.z += __shfl_xor(a.z, o);
            a.w += __shfl_xor(a.w, o);
        }
        return a;
    };
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = beg + chain; p < end; p += CHAINS)
        edge_accumulate64<STREAM>(acc, x + (size_t)src[p] * 64, w_e + (size_t)p * 4096, g, q);
    acc = reduce_g(acc);
    if (g == 0) *reinterpret_cast<float4*>(&red[chain][4 * (l % QN)]) = acc;
    // the root term: the chain that got the fewest edges takes it (as the wave deg % 16 does in the row kernel)
    const bool root_chain = root != nullptr && chain == (deg % CHAINS);
    float4 racc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (root_chain) edge_accumulate64<false>(racc, x + (size_t)row * 64, root, g, q);
    racc = reduce_g(racc);
    if (root_chain && g == 0) *reinterpret_cast<float4*>(&rootred[4 * (l % QN)]) = racc;
    __syncthreads();
    if (tid < 64) {      // wave 0: lanes < COLS own an output column each
        const int col = part * COLS + tid;
        if (tid < COLS) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) s += red[c][tid];   // fixed order: bitwise reproducible
            if (aggr == MDNO_AGGR_MEAN) s = s / (float)(deg > 1 ? deg : 1);
            if (root != nullptr) s += rootred[tid];
            if (bias != nullptr) s += bias[col];
            if (relu) s = relu_f(s);
            y[(size_t)row * 64 + col] = s;
        }
        if (fc.out != nullptr) {
            // the row's last part to get here applies the output layer to the whole row (release our columns,
            // acquire the partners': agent scope — the parts may sit on different CUs of the XCD)
            int last = 0;
            if (tid == 0)
                last = __hip_atomic_fetch_add(fc.row_done + row, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == SPLIT - 1;
            last = __shfl(last, 0);
            if (last) {
                if (tid == 0) fc.row_done[row] = 0;
                const float s = __builtin_nontemporal_load(y + (size_t)row * 64 + tid);      // (never from a stale L1 line)
                const int step = fc.t_dev ? *fc.t_dev : 0;
                float* o_ptr = fc.out + ((size_t)(fc.t_out + step) * num_rows + row) * fc.out_width;
                for (int o = 0; o < fc.out_width; ++o) {
                    float v = fmaf(s, fc.w[(size_t)o * 64 + tid], 0.f);
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
                    if (tid == 0) o_ptr[o] = v + (fc.b ? fc.b[o] : 0.f);
                }
                if (fc.step.done != nullptr && tid == 0 && atomicAdd(fc.step.done, 1) == num_rows - 1) {
                    *fc.step.done = 0;
                    if (fc.step.edges_per_step) fc.step.edges_per_step[step] = *fc.step.num_edges;
                    *fc.step.t_dev = step + 1;
                }
            }
        }
    }
}

// Any (Cin, Cout): one wave per destination row, lane = output column (strided), sequential edges.
// Used for the small-dimension fixtures; not a performance path.
__global__ __launch_bounds__(256) void nnconv_generic_kernel(
    const float* __restrict__ x, const int* __restrict__ row_ptr, const int* __restrict__ src,
    const float* __restrict__ w_e, const float* __restrict__ root, const float* __restrict__ bias,
    float* __restrict__ y, int num_rows, int Cin, int Cout, int aggr, int relu) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= num_rows) return;
    const int beg = row_ptr[row], end = row_ptr[row + 1];
    const int deg = end - beg;
    for (int o = lane; o < Cout; o += 64) {
        float s = 0.f;
        for (int p = beg; p < end; ++p) {
            const float* xj = x + (size_t)src[p] * Cin;
            const float* w = w_e + (size_t)p * Cin * Cout + o;
            float m = 0.f;
            for (int i = 0; i < Cin; ++i) m = fmaf(xj[i], w[(size_t)i * Cout], m);
            s = aggr != MDNO_AGGR_MAX ? s + m : (p == beg ? m : fmaxf(s, m));
        }
        if (aggr == MDNO_AGGR_MEAN) s = s / (float)(deg > 1 ? deg : 1);
        if (root != nullptr) {
            const float* xr = x + (size_t)row * Cin;
            float m = 0.f;
            for (int i = 0; i < Cin; ++i) m = fmaf(xr[i], root[(size_t)i * Cout + o], m);
            s += m;
        }
        if (bias != nullptr) s += bias[o];
        if (relu) s = relu_f(s);
        y[(size_t)row * Cout + o] = s;
    }
}

}  // namespace
}  // namespace mdno

int mdno::nnconv(const float* x, const int* row_ptr, const int* src, int num_rows, const float* w_e,
                 const float* root, const float* bias, int Cin, int Cout, int aggr, int relu, float* y,
                 hipStream_t s, const FcTail* fc, long long edge_cap) {
    MDNO_REQUIRE(x && row_ptr && src && w_e && y, MDNO_EINVAL, "nnconv: null pointer");
    // all of W_e within the L2s' reach (8 x 4 MiB): the applications of a forward find it there
    const bool w_e_cacheable = edge_cap > 0 && edge_cap * (long long)Cin * Cout * 4 <= (24ll << 20);
    MDNO_REQUIRE(num_rows > 0 && Cin > 0 && Cout > 0, MDNO_EINVAL, "nnconv: rows=%d Cin=%d Cout=%d", num_rows, Cin,
                 Cout);
    MDNO_REQUIRE(aggr == MDNO_AGGR_ADD || aggr == MDNO_AGGR_MEAN || aggr == MDNO_AGGR_MAX, MDNO_EUNSUPPORTED,
                 "nnconv: aggr %d not implemented (add, mean, max)", aggr);
    MDNO_REQUIRE(aggr != MDNO_AGGR_MAX || fc == nullptr, MDNO_EUNSUPPORTED, "nnconv: max aggregation inside the model");
    MDNO_REQUIRE(x != y, MDNO_EINVAL, "nnconv: y aliases x");
    const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w_e) |
                           reinterpret_cast<uintptr_t>(root)) & 15) == 0;
    MDNO_REQUIRE(!fc || (fc->w && fc->out && fc->out_width > 0 && fc->t_out >= 0 &&
                         (!fc->step.done || (fc->step.t_dev && fc->step.t_dev == fc->t_dev && fc->step.num_edges))),
                 MDNO_EINVAL, "nnconv: incomplete output-layer tail");
    const FcTail no_tail{};
    if (fc && !(Cin == 64 && Cout == 64 && aligned)) {      // the generic kernel has no tail: the layer gets its own launch
        MDNO_TRY(nnconv(x, row_ptr, src, num_rows, w_e, root, bias, Cin, Cout, aggr, relu, y, s, nullptr, edge_cap));
        return fc_out(y, fc->w, fc->b, num_rows, Cout, fc->out_width, fc->out, fc->t_out, fc->t_dev, s,
                      fc->step.done ? &fc->step : nullptr);
    }
    TimedSection ts(KID_NNCONV, s);
    if (Cin == 64 && Cout == 64 && aligned) {
        // waves per destination row: with only a few hundred rows (one ~500-atom trajectory) more
        // waves per row keep enough loads in flight on every CU.  Both shapes add a row's edges in the
        // same 16 chains, so the choice never changes a bit of the result
        if (num_rows >= 4096)
            hipLaunchKernelGGL(nnconv64_row_kernel<4>, dim3(num_rows), dim3(256), 0, s, x, row_ptr, src, w_e, root,
                               bias, y, num_rows, aggr, relu, fc ? *fc : no_tail);
        else if (aggr != MDNO_AGGR_MAX && num_rows <= 128 && (!fc || fc->row_done)) {
            // fewer rows than CUs: four workgroups per row, 16 output columns each — a row's edges reach one CU at
            // 66-73 GB/s from L2 and 33 GB/s from the Infinity Cache, so rows alone would leave most of the chip's
            // fetch paths idle (N = 28, same box: 0.111 ms/step with one workgroup per row, 0.102 with two, 0.098
            // with four)
            const unsigned rows8 = (unsigned)((num_rows + 7) / 8 * 8);
            if (num_rows > 64)      // (as many workgroups as CUs at most; two parts read whole 128-B lines)
                hipLaunchKernelGGL((nnconv64_colsplit_kernel<2, true>), dim3(rows8 * 2), dim3(512), 0, s, x, row_ptr, src,
                                   w_e, root, bias, y, num_rows, aggr, relu, fc ? *fc : no_tail);
            else if (w_e_cacheable)
                hipLaunchKernelGGL((nnconv64_colsplit_kernel<4, false>), dim3(rows8 * 4), dim3(256), 0, s, x, row_ptr, src,
                                   w_e, root, bias, y, num_rows, aggr, relu, fc ? *fc : no_tail);
            else
                hipLaunchKernelGGL((nnconv64_colsplit_kernel<4, true>), dim3(rows8 * 4), dim3(256), 0, s, x, row_ptr, src,
                                   w_e, root, bias, y, num_rows, aggr, relu, fc ? *fc : no_tail);
        } else if (w_e_cacheable)
            hipLaunchKernelGGL((nnconv64_row_kernel<16, false>), dim3(num_rows), dim3(1024), 0, s, x, row_ptr, src, w_e,
                               root, bias, y, num_rows, aggr, relu, fc ? *fc : no_tail);
        else
            hipLaunchKernelGGL(nnconv64_row_kernel<16>, dim3(num_rows), dim3(1024), 0, s, x, row_ptr, src, w_e, root,
                               bias, y, num_rows, aggr, relu, fc ? *fc : no_tail);
    } else {
        hipLaunchKernelGGL(nnconv_generic_kernel, dim3((num_rows + 3) / 4), dim3(256), 0, s, x, row_ptr, src, w_e,
                           root, bias, y, num_rows, Cin, Cout, aggr, relu);
    }
    return check_launch("nnconv");
}

extern "C" int mdno_nnconv_fwd(const float* x, const int32_t* row_ptr, const int32_t* src, int num_rows,
                               const float* w_e, const float* root, const float* bias, int Cin, int Cout,
                               int aggr, int relu, float* y, void* stream) {
    return mdno::nnconv(x, row_ptr, src, num_rows, w_e, root, bias, Cin, Cout, aggr, relu, y,
                        static_cast<hipStream_t>(stream));
}
