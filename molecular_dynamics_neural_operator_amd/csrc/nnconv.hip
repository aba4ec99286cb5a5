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

// acc += x[16g..16g+15] . Wblk[16g..16g+15][4q..4q+3]
__device__ __forceinline__ void edge_accumulate64(float4& acc, const float* __restrict__ xrow,
                                                  const float* __restrict__ wmat, int g, int q) {
    const float* xp = xrow + 16 * g;
    const float4 x0 = ld4(xp), x1 = ld4(xp + 4), x2 = ld4(xp + 8), x3 = ld4(xp + 12);
    const float* wp = wmat + (16 * g) * 64 + 4 * q;
    float4 w[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) w[r] = ld4_stream(wp + r * 64);
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

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void nnconv64_row_kernel(
    const float* __restrict__ x, const int* __restrict__ row_ptr, const int* __restrict__ src,
    const float* __restrict__ w_e, const float* __restrict__ root, const float* __restrict__ bias,
    float* __restrict__ y, int num_rows, int aggr, int relu) {
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
        for (int p = beg + wave + u * WAVES; p < end; p += CHAINS)
            edge_accumulate64(acc, x + (size_t)src[p] * 64, w_e + (size_t)p * 4096, g, q);
        acc = reduce_over_g(acc);
        if (lane < 16) *reinterpret_cast<float4*>(&red[wave + u * WAVES][4 * lane]) = acc;
    }
    // the root term x[r].root is one more "edge" with its own accumulator, taken by the wave that
    // got the fewest edges
    const bool root_wave = root != nullptr && wave == (deg % WAVES);
    float4 racc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (root_wave) edge_accumulate64(racc, x + (size_t)row * 64, root, g, q);

    racc = reduce_over_g(racc);
    if (root_wave && lane < 16) *reinterpret_cast<float4*>(&rootred[4 * lane]) = racc;
    __syncthreads();
    if (tid < 64) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) s += red[c][tid];   // fixed order: bitwise reproducible
        if (aggr == MDNO_AGGR_MEAN) s = s / (float)(deg > 1 ? deg : 1);
        if (root != nullptr) s += rootred[tid];
        if (bias != nullptr) s += bias[tid];
        if (relu) s = fmaxf(s, 0.f);
        y[(size_t)row * 64 + tid] = s;
    }
}

// ---------------------------------------------------------------- all conv applications of a SMALL graph in one launch
// The reference's own BBA (N = 28 C-alpha atoms, ~340 edges, bba_analysis.ipynb:1034) is launch-bound: W_e is
// 5 MB (L2-resident), a conv application is 2 us of work on 28 workgroups behind a ~9 us launch, twelve times
// per forward (graph_kernel.py:299-302).  Here ONE launch runs all 2*depth applications and fc2 (:305): workgroup
// w owns rows w, w + G, ... for the whole forward (G <= 256: every workgroup resident), and an application's
// output travels to its consumers — the workgroups holding a neighbour — as DATAFLOW, not through a grid
// barrier: a row is stored write-through (sc1: 8-byte agent-scope stores), the storing wave drains (vmcnt(0))
// and one lane then stores the row's flag = tag of this launch and application; a consuming wave polls the
// flag of the source row of ITS edge (sc1 loads) and then reads the row with sc1 loads, which bypass its CU's L1
// (cdna_hip_programming.md Guideline 16, R1 with sc1 loads in place of the acquire: one storing wave per row, the
// polling wave is the loading wave).  Tags are unique per launch (a generation word in the workspace, bumped by a
// one-thread kernel behind this one), so nothing has to be cleared.  Spins are bounded: a workgroup that waits
// ~1 s sets MDNO_STATUS_FUSED_TIMEOUT and goes on.
// The arithmetic is nnconv64_row_kernel<16>'s, instruction for instruction (same chains, same order), and
// fc_out_kernel's: results are bit-identical to the launch-per-application path (tested).
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ void ld_row16_sc1(const float* xrow16, float (&v)[16]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned long long u = __hip_atomic_load((gu64*)(xrow16 + 2 * k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v[2 * k] = __builtin_bit_cast(float, (unsigned)u);
        v[2 * k + 1] = __builtin_bit_cast(float, (unsigned)(u >> 32));
    }
}

// edge_accumulate64 with the 16 input features already in registers
__device__ __forceinline__ void edge_accumulate64_regs(float4& acc, const float (&xv)[16], const float* __restrict__ wmat,
                                                       int g, int q) {
    const float* wp = wmat + (16 * g) * 64 + 4 * q;
    float4 w[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) w[r] = ld4_stream(wp + r * 64);
#pragma unroll
    for (int r = 0; r < 16; ++r) fma4(acc, xv[r], w[r]);
}

struct FusedConvArgs {
    const float* x0;            // [R,64] input of the first application (plain: written by an earlier launch)
    const int* row_ptr;
    const int* src;
    const float* w_e;
    const float* root[2];
    const float* bias[2];
    int depth, blocks;          // applications = depth * blocks; block b uses root[b], bias[b]
    float* xg;                  // [applications][R][64] row payloads (sc1)
    unsigned* flags;            // [applications][R]
    const unsigned* gen;        // generation of this launch
    const float* fc2_w;         // [out_width, 64]
    const float* fc2_b;
    int out_width;
    float* out_frames;          // frame t_out + *t_dev of [T][R][out_width]
    int t_out;
    const int* t_dev;
    float* latent;              // [R,64] or null
    int R;
    int* status;
};

constexpr unsigned kFusedSpinLimit = 1u << 22;

__global__ __launch_bounds__(1024) void nnconv64_fused_kernel(FusedConvArgs a) {
    __shared__ float red[CHAINS][64];
    __shared__ float rootred[64];
    __shared__ float rowout[64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int g = lane >> 4, q = lane & 15;
    const int L = a.depth * a.blocks;
    const unsigned gen = *a.gen;
    bool timed_out = false;
    for (int app = 0; app < L; ++app) {
        const float* root = a.root[app / a.depth];
        const float* bias = a.bias[app / a.depth];
        const unsigned tag_in = (gen << 5) | (unsigned)app;           // tag of application app-1's rows (app >= 1)
        const unsigned tag_out = (gen << 5) | (unsigned)(app + 1);
        const float* xin = app == 0 ? a.x0 : a.xg + (size_t)(app - 1) * a.R * 64;
        const unsigned* fin = app == 0 ? nullptr : a.flags + (size_t)(app - 1) * a.R;
        float* xout = a.xg + (size_t)app * a.R * 64;
        for (int row = blockIdx.x; row < a.R; row += gridDim.x) {
            const int beg = a.row_ptr[row], end = a.row_ptr[row + 1];
            const int deg = end - beg;
            // wait for a source row of the previous application, then its 16 features of this lane's group
            auto fetch = [&](int srow, float (&xv)[16]) {
                if (fin != nullptr) {
                    unsigned spins = 0;
                    while (__hip_atomic_load((gu32*)(fin + srow), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag_in) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > kFusedSpinLimit) { timed_out = true; break; }
                    }
                }
                // (the first application's input was written by an earlier launch: the same loads, no flag)
                ld_row16_sc1(xin + (size_t)srow * 64 + 16 * g, xv);
            };
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int p = beg + wave; p < end; p += CHAINS) {
                float xv[16];
                fetch(a.src[p], xv);
                edge_accumulate64_regs(acc, xv, a.w_e + (size_t)p * 4096, g, q);
            }
            acc = reduce_over_g(acc);
            if (lane < 16) *reinterpret_cast<float4*>(&red[wave][4 * lane]) = acc;
            const bool root_wave = root != nullptr && wave == (deg % CHAINS);
            float4 racc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (root_wave) {
                float xv[16];
                fetch(row, xv);
                edge_accumulate64_regs(racc, xv, root, g, q);
            }
            racc = reduce_over_g(racc);
            if (root_wave && lane < 16) *reinterpret_cast<float4*>(&rootred[4 * lane]) = racc;
            __syncthreads();
            if (tid < 64) {
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) s += red[c][tid];
                s = s / (float)(deg > 1 ? deg : 1);            // mean aggregation (the model's; graph_kernel.py:272-273)
                if (root != nullptr) s += rootred[tid];
                if (bias != nullptr) s += bias[tid];
                s = fmaxf(s, 0.f);
                rowout[tid] = s;
                // publish: 8-byte write-through stores of the row, drain, then the flag (one wave: no barrier needed)
                if (tid < 32) {
                    const unsigned long long u = (unsigned long long)__builtin_bit_cast(unsigned, rowout[2 * tid]) |
                                                 ((unsigned long long)__builtin_bit_cast(unsigned, rowout[2 * tid + 1]) << 32);
                    __hip_atomic_store((gu64*)(xout + (size_t)row * 64 + 2 * tid), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (tid == 0)
                    __hip_atomic_store((gu32*)(a.flags + (size_t)app * a.R + row), tag_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (app == L - 1) {       // fc2 (fc_out_kernel's arithmetic) and the latent copy, from the row at hand
                    if (a.latent) a.latent[(size_t)row * 64 + tid] = s;
                    const int t = a.t_out + (a.t_dev ? *a.t_dev : 0);
                    float* o_ptr = a.out_frames + ((size_t)t * a.R + row) * a.out_width;
                    for (int o = 0; o < a.out_width; ++o) {
                        float v = fmaf(s, a.fc2_w[(size_t)o * 64 + tid], 0.f);
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
                        if (tid == 0) o_ptr[o] = v + (a.fc2_b ? a.fc2_b[o] : 0.f);
                    }
                }
            }
            __syncthreads();              // red / rootred / rowout are reused by the next row
        }
    }
    if (timed_out && a.status) atomicOr(a.status, MDNO_STATUS_FUSED_TIMEOUT);
}

__global__ void bump_generation_kernel(unsigned* gen) { *gen = *gen + 1; }

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
            s += m;
        }
        if (aggr == MDNO_AGGR_MEAN) s = s / (float)(deg > 1 ? deg : 1);
        if (root != nullptr) {
            const float* xr = x + (size_t)row * Cin;
            float m = 0.f;
            for (int i = 0; i < Cin; ++i) m = fmaf(xr[i], root[(size_t)i * Cout + o], m);
            s += m;
        }
        if (bias != nullptr) s += bias[o];
        if (relu) s = fmaxf(s, 0.f);
        y[(size_t)row * Cout + o] = s;
    }
}

}  // namespace
}  // namespace mdno

int mdno::nnconv(const float* x, const int* row_ptr, const int* src, int num_rows, const float* w_e,
                 const float* root, const float* bias, int Cin, int Cout, int aggr, int relu, float* y,
                 hipStream_t s) {
    MDNO_REQUIRE(x && row_ptr && src && w_e && y, MDNO_EINVAL, "nnconv: null pointer");
    MDNO_REQUIRE(num_rows > 0 && Cin > 0 && Cout > 0, MDNO_EINVAL, "nnconv: rows=%d Cin=%d Cout=%d", num_rows, Cin,
                 Cout);
    MDNO_REQUIRE(aggr == MDNO_AGGR_ADD || aggr == MDNO_AGGR_MEAN, MDNO_EUNSUPPORTED,
                 "nnconv: aggr %d not implemented (add, mean)", aggr);
    MDNO_REQUIRE(x != y, MDNO_EINVAL, "nnconv: y aliases x");
    const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w_e) |
                           reinterpret_cast<uintptr_t>(root)) & 15) == 0;
    TimedSection ts(KID_NNCONV, s);
    if (Cin == 64 && Cout == 64 && aligned) {
        // waves per destination row: with only a few hundred rows (one ~500-atom trajectory) more
        // waves per row keep enough loads in flight on every CU.  Both shapes add a row's edges in the
        // same 16 chains, so the choice never changes a bit of the result
        if (num_rows >= 4096)
            hipLaunchKernelGGL(nnconv64_row_kernel<4>, dim3(num_rows), dim3(256), 0, s, x, row_ptr, src, w_e, root,
                               bias, y, num_rows, aggr, relu);
        else
            hipLaunchKernelGGL(nnconv64_row_kernel<16>, dim3(num_rows), dim3(1024), 0, s, x, row_ptr, src, w_e, root,
                               bias, y, num_rows, aggr, relu);
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


// ---- fused small-graph path (engine.hip decides when)
namespace mdno {
size_t nnconv_fused_workspace_bytes(int R, int applications) {
    return align_up((size_t)applications * R * 64 * sizeof(float), 256) + align_up((size_t)applications * R * sizeof(unsigned), 256) + 256;
}

// rows up to 8 per workgroup at 256 workgroups; edge lists small enough that an application is launch-, not
// bandwidth-bound (W_e of <= 8,192 edges = 128 MiB streams in ~25 us)
bool nnconv_fused_applicable(int R, long long edge_cap, int width) { return width == 64 && R <= 2048 && edge_cap <= 8192; }

int nnconv_fused(const float* x0, const int* row_ptr, const int* src, int R, const float* w_e, const float* root1,
                 const float* bias1, const float* root2, const float* bias2, int depth, int blocks, const float* fc2_w,
                 const float* fc2_b, int out_width, float* out_frames, int t_out, const int* t_dev, float* latent,
                 void* workspace, int* status, hipStream_t s) {
    MDNO_REQUIRE(x0 && row_ptr && src && w_e && fc2_w && out_frames && workspace && R > 0 && depth > 0 && blocks >= 1 &&
                     blocks <= 2 && depth * blocks < 32,
                 MDNO_EINVAL, "nnconv_fused: bad arguments");
    const int L = depth * blocks;
    Carver cv(workspace);
    FusedConvArgs a{};
    a.x0 = x0; a.row_ptr = row_ptr; a.src = src; a.w_e = w_e;
    a.root[0] = root1; a.bias[0] = bias1; a.root[1] = root2; a.bias[1] = bias2;
    a.depth = depth; a.blocks = blocks;
    a.xg = cv.take<float>((size_t)L * R * 64);
    a.flags = cv.take<unsigned>((size_t)L * R);
    unsigned* gen = cv.take<unsigned>(64);
    a.gen = gen;
    a.fc2_w = fc2_w; a.fc2_b = fc2_b; a.out_width = out_width; a.out_frames = out_frames; a.t_out = t_out; a.t_dev = t_dev;
    a.latent = latent; a.R = R; a.status = status;
    {
        TimedSection ts(KID_NNCONV, s);
        hipLaunchKernelGGL(nnconv64_fused_kernel, dim3(R < 256 ? R : 256), dim3(1024), 0, s, a);
    }
    hipLaunchKernelGGL(bump_generation_kernel, dim3(1), dim3(1), 0, s, gen);
    return check_launch("nnconv64_fused_kernel");
}
}  // namespace mdno
