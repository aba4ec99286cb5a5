// K7/K8 node prologue and K9 output projection: the per-atom ends of KernelNN.forward.
//
// Prologue replaces graph_kernel.py:279-298 — W sequential nn.LSTM(3,3) single-step calls over the
// window with the atoms as the LSTM batch and zero initial (h, c), lstm_fc, nn.Embedding lookup,
// concat [emb, x], fc1, ReLU — in one launch.  B=1 semantics per member (members
// are independent; the reference's batched mode threads one LSTM state through the batch axis and
// is not reproduced).  Output projection replaces fc2 (:305) and writes the new frame straight
// into the trajectory buffer.  Both are negligible in time (R*~1e3 flop); they exist so that a
// rollout step never leaves the device.
#include "kernels.h"
#include "graph_small.h"

namespace mdno {
namespace {

constexpr int H = 3;            // x_position_dim == LSTM hidden size (graph_kernel.py:264)
constexpr int MAX_EMB = 16;

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

struct PrologueArgs {
    const float* frames;
    int t0;
    const int* t0_dev;
    int M, W, N;
    const long long* aa;
    int aa_per_member;
    const float *w_ih, *w_hh, *b_ih, *b_hh, *fc_w, *fc_b, *emb_w, *fc1_w, *fc1_b;
    int num_emb, emb_dim, width;
    float* x0;
    int* status;
};

// one wave per row r
__device__ __forceinline__ void node_prologue_row(const PrologueArgs& a, int r, int lane) {
    const int R = a.M * a.N;
    if (r >= R) return;
    const int m = r / a.N, n = r - m * a.N;
    const int t0 = a.t0 + (a.t0_dev ? *a.t0_dev : 0);
    const float* f0 = a.frames + ((size_t)t0 * R + r) * 3;  // frames are time-major [T, M*N, 3]

    float h[H] = {0.f, 0.f, 0.f};
    if (a.w_ih != nullptr) {
        // The W cells are a serial chain per atom, so the wave's lanes split each cell instead of repeating
        // it: lane g < 12 owns gate g (its row of W_ih / W_hh, its bias, its nonlinearity), the twelve
        // gate values are then broadcast and every lane updates (c, h) — still wave-uniform, and each
        // gate's sum is formed in the same order as before (a cell's critical path is one gate, not twelve).
        const int g = lane < 4 * H ? lane : 0;
        float wih[H], whh[H];
#pragma unroll
        for (int k = 0; k < H; ++k) {
            wih[k] = a.w_ih[g * H + k];
            whh[k] = a.w_hh[g * H + k];
        }
        const float bsum = a.b_ih[g] + a.b_hh[g];
        const bool is_tanh = g >= 2 * H && g < 3 * H;
        float c[H] = {0.f, 0.f, 0.f};
        for (int tb = 0; tb < a.W; tb += 8) {
            // the window's frames do not depend on the recurrence: eight steps' positions are fetched at
            // once, so the chain below waits for memory once per eight cells instead of once per cell
            float xs[8][H];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float* p = f0 + (size_t)(tb + u < a.W ? tb + u : a.W - 1) * R * 3;
                xs[u][0] = p[0]; xs[u][1] = p[1]; xs[u][2] = p[2];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (tb + u < a.W) {
                    float s = bsum;
#pragma unroll
                    for (int k = 0; k < H; ++k) s = fmaf(wih[k], xs[u][k], s);
#pragma unroll
                    for (int k = 0; k < H; ++k) s = fmaf(whh[k], h[k], s);
                    const float act = is_tanh ? tanhf(s) : sigmoidf_(s);
#pragma unroll
                    for (int k = 0; k < H; ++k) {
                        const float ig = __shfl(act, k), fg = __shfl(act, H + k);
                        const float gg = __shfl(act, 2 * H + k), og = __shfl(act, 3 * H + k);
                        c[k] = fg * c[k] + ig * gg;
                        h[k] = og * tanhf(c[k]);
                    }
                }
            }
        }
    }
    float feat[MAX_EMB + H];
    long long id = a.aa[a.aa_per_member ? r : n];
    if (id < 0 || id >= a.num_emb) {
        if (lane == 0 && a.status) atomicOr(a.status, MDNO_STATUS_BAD_AMINOACID);
        id = id < 0 ? 0 : a.num_emb - 1;
    }
#pragma unroll
    for (int e = 0; e < MAX_EMB; ++e) feat[e] = (e < a.emb_dim) ? a.emb_w[id * a.emb_dim + e] : 0.f;
    if (a.w_ih != nullptr) {
#pragma unroll
        for (int k = 0; k < H; ++k) {
            float s = a.fc_b[k];
#pragma unroll
            for (int j = 0; j < H; ++j) s = fmaf(a.fc_w[k * H + j], h[j], s);
            feat[MAX_EMB + k] = s;
        }
    } else {
        // notebook-era model (bba_analysis.ipynb:123-128: emb, fc1, conv1, fc2 only): the node
        // feature is the raw position of the newest window frame
        const float* p = f0 + (size_t)(a.W - 1) * R * 3;
#pragma unroll
        for (int k = 0; k < H; ++k) feat[MAX_EMB + k] = p[k];
    }
    const int in_w = a.emb_dim + H;
    for (int o = lane; o < a.width; o += 64) {
        const float* w = a.fc1_w + (size_t)o * in_w;
        float s = a.fc1_b[o];
        for (int e = 0; e < a.emb_dim; ++e) s = fmaf(w[e], feat[e], s);
#pragma unroll
        for (int k = 0; k < H; ++k) s = fmaf(w[a.emb_dim + k], feat[MAX_EMB + k], s);
        a.x0[(size_t)r * a.width + o] = relu_f(s);
    }
}

__global__ __launch_bounds__(256) void node_prologue_kernel(PrologueArgs a) {
    node_prologue_row(a, blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
}

// Head of a rollout step on a short chain: workgroup 0 builds the radius graph of the newest frame, the others
// run the node prologue (16 rows each) — the two do not depend on each other, and at this size each is shorter
// than the launch that would carry it.
__global__ __launch_bounds__(1024) void step_head_small_kernel(SmallGraphArgs g, PrologueArgs a) {
    if (blockIdx.x == 0) radius_graph_small_body(g);
    else node_prologue_row(a, (blockIdx.x - 1) * 16 + (threadIdx.x >> 6), threadIdx.x & 63);
}

__global__ __launch_bounds__(256) void fc_out_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ b, int R, int width,
                                                     int out_width, float* __restrict__ out, int t_out,
                                                     const int* __restrict__ t_dev, StepTail tail) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int step = t_dev ? *t_dev : 0;
    if (r < R) {
        float* o_ptr = out + ((size_t)(t_out + step) * R + r) * out_width;
        const float* xr = x + (size_t)r * width;
        for (int o = 0; o < out_width; ++o) {
            float s = 0.f;
            for (int c = lane; c < width; c += 64) s = fmaf(xr[c], w[(size_t)o * width + c], s);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
            if (lane == 0) o_ptr[o] = s + (b ? b[o] : 0.f);
        }
    }
    if (tail.done == nullptr) return;
    // the step counter moves on once every workgroup is past its read of it (each has `step` in a register
    // before it gets here; the next kernel of the stream starts after this one has drained)
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(tail.done, 1) == (int)gridDim.x - 1) {
        *tail.done = 0;
        if (tail.edges_per_step) tail.edges_per_step[step] = *tail.num_edges;
        *tail.t_dev = step + 1;
    }
}

}  // namespace
}  // namespace mdno

namespace mdno {
namespace {
int prologue_args(const mdno_kernelnn_params* p, const float* frames, int t0, const int* t_dev, int M, int W, int N,
                  const long long* aa, int aa_per_member, float* x0, int* status, PrologueArgs& a) {
    MDNO_REQUIRE(p && frames && aa && x0, MDNO_EINVAL, "node_prologue: null pointer");
    MDNO_REQUIRE(p->emb_w && p->fc1_w && p->fc1_b, MDNO_EINVAL, "node_prologue: null weight pointer");
    const bool lstm = p->lstm_w_ih != nullptr;
    MDNO_REQUIRE(!lstm || (p->lstm_w_hh && p->lstm_b_ih && p->lstm_b_hh && p->lstm_fc_w && p->lstm_fc_b), MDNO_EINVAL,
                 "node_prologue: partial LSTM weight set");
    MDNO_REQUIRE(M > 0 && W > 0 && N > 0 && t0 >= 0, MDNO_EINVAL, "node_prologue: M=%d W=%d N=%d", M, W, N);
    MDNO_REQUIRE(p->x_position_dim == H, MDNO_EUNSUPPORTED, "x_position_dim=%d (only 3)", p->x_position_dim);
    MDNO_REQUIRE(p->embedding_dim >= 0 && p->embedding_dim <= MAX_EMB, MDNO_EUNSUPPORTED, "embedding_dim=%d (0..%d)",
                 p->embedding_dim, MAX_EMB);
    MDNO_REQUIRE(p->in_width == p->embedding_dim + H, MDNO_EINVAL,
                 "in_width=%d must equal embedding_dim + 3 = %d (graph_kernel.py:296)", p->in_width,
                 p->embedding_dim + H);
    a = PrologueArgs{frames, t0, t_dev, M, W, N, aa, aa_per_member, p->lstm_w_ih, p->lstm_w_hh, p->lstm_b_ih,
                     p->lstm_b_hh, p->lstm_fc_w, p->lstm_fc_b, p->emb_w, p->fc1_w, p->fc1_b, p->num_embeddings,
                     p->embedding_dim, p->width, x0, status};
    return MDNO_OK;
}
}  // namespace
}  // namespace mdno

int mdno::node_prologue(const mdno_kernelnn_params* p, const float* frames, int t0, const int* t_dev, int M, int W,
                        int N, const long long* aa, int aa_per_member, float* x0, int* status, hipStream_t s) {
    PrologueArgs a;
    MDNO_TRY(prologue_args(p, frames, t0, t_dev, M, W, N, aa, aa_per_member, x0, status, a));
    const int R = M * N;
    TimedSection ts(KID_PROLOGUE, s);
    hipLaunchKernelGGL(node_prologue_kernel, dim3((R + 3) / 4), dim3(256), 0, s, a);
    return check_launch("node_prologue");
}

bool mdno::step_head_small_supported(int M, int N) { return small_graph_supported(M, N); }

int mdno::step_head_small(const mdno_kernelnn_params* p, const float* frames, int W, const int* t_dev, int M, int N,
                          const long long* aa, int aa_per_member, float* x0, double cutoff, int* row_ptr, int* src,
                          int* dst, long long edge_cap, int* num_edges, int* status, int* zero_words, int n_zero,
                          hipStream_t s) {
    MDNO_REQUIRE(step_head_small_supported(M, N), MDNO_EUNSUPPORTED, "step_head_small: M=%d N=%d", M, N);
    MDNO_REQUIRE(row_ptr && src && num_edges && edge_cap > 0 && t_dev, MDNO_EINVAL, "step_head_small: null pointer");
    MDNO_REQUIRE(n_zero >= 0 && n_zero <= 64 && (n_zero == 0 || zero_words), MDNO_EINVAL, "step_head_small: n_zero=%d", n_zero);
    PrologueArgs a;
    MDNO_TRY(prologue_args(p, frames, 0, t_dev, M, W, N, aa, aa_per_member, x0, status, a));
    const int R = M * N;
    // graph of the LAST window frame (W - 1 + t), prologue over the window starting at frame t
    const SmallGraphArgs g{frames, W - 1, t_dev, N, R, cutoff, edge_cap, row_ptr, src, dst, num_edges, status,
                           zero_words, n_zero};
    TimedSection ts(KID_GRAPH, s);
    hipLaunchKernelGGL(step_head_small_kernel, dim3(1 + (R + 15) / 16), dim3(1024), 0, s, g, a);
    return check_launch("step_head_small");
}

int mdno::fc_out(const float* x, const float* w, const float* b, int rows, int width, int out_width,
                 float* out_frames, int t_out, const int* t_dev, hipStream_t s, const StepTail* tail) {
    MDNO_REQUIRE(x && w && out_frames, MDNO_EINVAL, "fc_out: null pointer");
    MDNO_REQUIRE(rows > 0 && width > 0 && out_width > 0 && t_out >= 0, MDNO_EINVAL, "fc_out: bad sizes");
    MDNO_REQUIRE(!tail || (tail->t_dev && tail->t_dev == t_dev && tail->num_edges && tail->done), MDNO_EINVAL,
                 "fc_out: incomplete step tail");
    TimedSection ts(KID_FC_OUT, s);
    hipLaunchKernelGGL(fc_out_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, w, b, rows, width, out_width,
                       out_frames, t_out, t_dev, tail ? *tail : StepTail{nullptr, nullptr, nullptr, nullptr});
    return check_launch("fc_out");
}

extern "C" int mdno_node_prologue_fwd(const mdno_kernelnn_params* p, const float* frames, int M, int W, int N,
                                      const int64_t* x_aminoacid, int aa_per_member, float* x0, int32_t* status,
                                      void* stream) {
    return mdno::node_prologue(p, frames, 0, nullptr, M, W, N, (const long long*)x_aminoacid, aa_per_member, x0,
                               status, static_cast<hipStream_t>(stream));
}

extern "C" int mdno_fc_out_fwd(const float* x, const float* w, const float* b, int rows, int width, int out_width,
                               float* out, void* stream) {
    return mdno::fc_out(x, w, b, rows, width, out_width, out, 0, nullptr, static_cast<hipStream_t>(stream));
}
