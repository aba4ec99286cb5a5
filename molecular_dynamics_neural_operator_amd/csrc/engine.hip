// Whole forward and the on-device autoregressive rollout.
//
// forward  replaces KernelNN.forward            (graph_kernel.py:277-309)
// rollout  replaces recursive_propagation       (graph_kernel.py:396-413) and the notebook's
//          propogate (bba_analysis.ipynb:336-358): the reference crosses PCIe twice per step and
//          rebuilds the graph with scipy on the host; here a step is a fixed sequence of launches
//          on one stream, captured once into a hipGraph and replayed.
//
// One algorithmic change relative to the reference, value-preserving: conv1 and conv2 share ONE
// edge-MLP (graph_kernel.py:271-273) and edge_attr never changes inside a forward (:278-302), so
// all 2*depth conv applications use the same W_e = net(edge_attr).  It is evaluated once per
// forward instead of 2*depth times.
#include "kernels.h"

#include <string>
#include <vector>

namespace mdno {

struct Timer {
    struct Rec { hipEvent_t a, b; int kid; };
    std::vector<Rec> recs;
    size_t used = 0;
    size_t open_idx[KID_COUNT] = {};
};
thread_local Timer* g_active_timer = nullptr;

void timer_mark(int kid, bool start, hipStream_t s) {
    Timer* t = g_active_timer;
    if (!t) return;
    if (start) {
        if (t->used >= t->recs.size()) { t->open_idx[kid] = (size_t)-1; return; }
        const size_t i = t->used++;
        t->recs[i].kid = kid;
        t->open_idx[kid] = i;
        (void)hipEventRecord(t->recs[i].a, s);
    } else if (t->open_idx[kid] != (size_t)-1) {
        (void)hipEventRecord(t->recs[t->open_idx[kid]].b, s);
    }
}

namespace {
thread_local std::string g_last_error;
}

void set_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

namespace {

int validate_params(const mdno_kernelnn_params* p) {
    MDNO_REQUIRE(p != nullptr, MDNO_EINVAL, "params: null");
    MDNO_REQUIRE(p->width > 0 && p->ker_width > 0 && p->depth >= 0 && p->ker_in > 0 && p->out_width > 0, MDNO_EINVAL,
                 "params: width=%d ker_width=%d depth=%d ker_in=%d out_width=%d", p->width, p->ker_width, p->depth,
                 p->ker_in, p->out_width);
    MDNO_REQUIRE(p->k_w0 && p->k_b0 && p->k_w1 && p->k_b1 && p->k_w2 && p->k_b2 && p->conv1_root && p->conv1_bias &&
                     p->fc2_w && p->fc2_b,
                 MDNO_EINVAL, "params: null weight pointer");
    MDNO_REQUIRE(p->conv_mode >= MDNO_CONV_MATERIALIZED && p->conv_mode <= MDNO_CONV_AUTO, MDNO_EINVAL,
                 "params: conv_mode=%d", p->conv_mode);
    MDNO_REQUIRE((p->conv2_root == nullptr) == (p->conv2_bias == nullptr), MDNO_EINVAL,
                 "params: conv2_root and conv2_bias must both be set or both be NULL");
    return MDNO_OK;
}

bool separate_conv2_kernel(const mdno_kernelnn_params* p) {
    return p->k2_w0 && (p->k2_w0 != p->k_w0 || p->k2_w1 != p->k_w1 || p->k2_w2 != p->k_w2 || p->k2_b0 != p->k_b0 ||
                        p->k2_b1 != p->k_b1 || p->k2_b2 != p->k_b2);
}

struct FwdWs {
    float *xa, *xb, *w_e, *h2;
    void *mlp, *fact;
    size_t mlp_bytes, total;
    bool factored;      // destination-side moment form (moment.hip), in every GEMM mode
};

bool factored_available(const mdno_kernelnn_params* p) { return moment_supported(p->width, p->ker_width); }

// The factored conv applies to graphs the library builds itself (symmetric radius graphs) at width 64.
// AUTO adds a size rule: below ~8-10k edges the factored form's fixed cost per application (Y GEMM
// over 64*k columns whatever the row count, 16 pipeline iterations per source) loses to simply
// materialising W_e (N=28: 0.25 vs 0.52 ms/step, N=120: 0.27 vs 0.58; N=504: 5.3 vs 2.4).  The rule
// looks at the capacity PER MEMBER, so a member takes the same path — and gives the same bits —
// alone, in a batch of 8 or in a shard of 64.
constexpr long long kAutoFactoredMinEdgeCapPerMember = 24576;
// the rule on a COUNTED graph (mdno_conv_mode_for_graph): the factored form keeps a 256 KiB object per source node and
// application where the materialised one streams 16 KiB per edge — it pays from ~40 neighbours per atom on, once the
// graph fills its launches
constexpr long long kAutoFactoredMinDegree = 40;
constexpr long long kAutoFactoredMinEdgesPerMember = 16384;

bool use_factored(const mdno_kernelnn_params* p, int M, long long edge_cap, bool position_graph) {
    if (p->conv_mode == MDNO_CONV_MATERIALIZED || !factored_available(p)) return false;
    if (p->conv_mode == MDNO_CONV_FACTORED) return true;
    return position_graph && edge_cap / (M > 0 ? M : 1) >= kAutoFactoredMinEdgeCapPerMember;
}

FwdWs carve_fwd(void* ws, const mdno_kernelnn_params* p, int M, int N, long long edge_cap, bool factored) {
    FwdWs f{};
    Carver cv(ws);
    const size_t R = (size_t)M * N;
    f.xa = cv.take<float>(R * p->width);
    f.xb = cv.take<float>(R * p->width);
    f.factored = factored;
    if (f.factored) {
        // no W_e at all: the last hidden activation H [edge_cap, k] plus the per-node Y and per-edge M
        // k-tiled [e/128][k/32][128][32]: whole 128-row tiles, so the row count is rounded up
        f.h2 = cv.take<float>((size_t)((edge_cap + 127) / 128 * 128) * p->ker_width);
        f.fact = cv.take<char>(moment_workspace_bytes((int)R, p->ker_width));
        f.mlp_bytes = mdno_edge_mlp_workspace_bytes(p->ker_width, p->ker_width, edge_cap, p->gemm_mode);
    } else {
        f.mlp_bytes = mdno_edge_mlp_workspace_bytes(p->ker_width, p->width * p->width, edge_cap, p->gemm_mode);
    }
    // the materialised path also serves explicit-edge_attr calls when conv_mode asks for factored
    f.w_e = cv.take<float>(f.factored ? 0 : (size_t)edge_cap * p->width * p->width);
    f.mlp = cv.take<char>(f.mlp_bytes);
    f.total = cv.used();
    return f;
}

// device words counting the SPLIT_F16 fallbacks of a forward workspace (mdno_*_fallback_counts): the factored conv's
// (two ints) and the edge-MLP's (one int); NULL where the configuration has none
void fwd_counter_words(const mdno_kernelnn_params* p, const FwdWs& fw, int M, int N, long long edge_cap, int** conv, int** mlp) {
    *conv = nullptr;
    *mlp = nullptr;
    if (p->gemm_mode != MDNO_GEMM_SPLIT_F16) return;
    if (fw.factored) *conv = moment_carve(fw.fact, M * N, p->ker_width).counters;
    int* act = edge_mlp_activation_flags(fw.mlp, p->ker_width, fw.factored ? p->ker_width : p->width * p->width, edge_cap,
                                         p->gemm_mode);
    if (act) *mlp = act + 7;      // (edge_mlp_split.hip: flags[F16_FALLBACK_COUNT = 8]; act = flags + 1)
}

int zero_counter_words(int* conv, int* mlp, hipStream_t s) {
    if (conv) MDNO_HIP(hipMemsetAsync(conv, 0, 2 * sizeof(int), s));
    if (mlp) MDNO_HIP(hipMemsetAsync(mlp, 0, sizeof(int), s));
    return MDNO_OK;
}

int read_counter_words(const int* conv, const int* mlp, int64_t counts[4], hipStream_t s) {
    int host[3] = {0, 0, 0};
    if (conv) MDNO_HIP(hipMemcpyAsync(host, conv, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    if (mlp) MDNO_HIP(hipMemcpyAsync(host + 2, mlp, sizeof(int), hipMemcpyDeviceToHost, s));
    MDNO_HIP(hipStreamSynchronize(s));
    counts[0] = host[0]; counts[1] = host[1]; counts[2] = host[2]; counts[3] = 0;
    return MDNO_OK;
}

// frames/t0/t_dev address the window; edge_frames/edge_frame the frame the graph was built on.
int forward_impl(const mdno_kernelnn_params* p, const float* frames, int t0, const int* t_dev, int M, int W, int N,
                 const long long* aa, int aa_per_member, const int* row_ptr, const int* src, const int* dst,
                 const int* num_edges, long long edge_cap, const float* edge_frames, int edge_frame,
                 const float* edge_attr, const int* perm, float* out_frames, int t_out, float* latent,
                 const FwdWs& ws, int* status, hipStream_t s, int phase = WP_BOTH, const StepTail* tail = nullptr) {
    const int R = M * N, C = p->width;
    // WP_PREPARE_ONLY: just the weight-derived operands of the (single, shared) edge-MLP
    const bool prep_only = (phase & WP_PHASE_MASK) == WP_PREPARE_ONLY;
    if (!prep_only && !(phase & WP_PROLOGUE_DONE))
        MDNO_TRY(node_prologue(p, frames, t0, t_dev, M, W, N, aa, aa_per_member, ws.xa, status, s));
    float* cur = ws.xa;
    float* nxt = ws.xb;
    bool fc_done = false;      // the output layer went out with the last conv application
    const int blocks = p->conv2_root ? 2 : 1;   // notebook-era model: conv1 only (lstm_* NULL as well)
    if (ws.factored) {
        // destination-side form: row r = DESTINATION r with in-edges src[p] -> r, attributes [pos[src], pos[dst]]
        // as the reference has them (graph_kernel.py:372-379) or the caller's own edge_attr (+ perm): ANY graph in
        // destination-sorted CSR, no symmetry needed
        MDNO_REQUIRE((edge_frames && dst) || edge_attr, MDNO_EINVAL,
                     "factored conv needs edge attributes: positions (edge_pos, dst) or edge_attr");
        const MomentWs mw = moment_carve(ws.fact, R, p->ker_width);
        if (!prep_only) MDNO_TRY(moment_prepare_graph(row_ptr, R, mw, s));
        if (!prep_only && p->gemm_mode == MDNO_GEMM_SPLIT_F16) MDNO_TRY(moment_row_absmax(cur, R, mw, s));
        int application = 0;
        for (int block = 0; block < blocks; ++block) {
            const bool own = block == 1 && separate_conv2_kernel(p);
            if (block == 0 || own) {
                EdgeMlpWeights w = own ? EdgeMlpWeights{p->k2_w0, p->k2_b0, p->k2_w1, p->k2_b1, p->k2_w2, p->k2_b2}
                                       : EdgeMlpWeights{p->k_w0, p->k_b0, p->k_w1, p->k_b1, p->k_w2, p->k_b2};
                MDNO_TRY(edge_mlp_hidden(edge_frames, edge_frame, t_dev, R, src, dst, edge_attr, perm, num_edges, edge_cap,
                                         p->ker_in, p->ker_width, p->gemm_mode, w, ws.h2, ws.mlp, ws.mlp_bytes, s,
                                         block == 0 ? phase : (phase & ~WP_FLAGS_ZEROED)));
                if ((phase & WP_PHASE_MASK) != WP_RUN_ONLY) MDNO_TRY(moment_prepare_weights(w.w2, w.b2, p->ker_width, mw, s, p->gemm_mode));
            }
            if (prep_only) return MDNO_OK;
            const float* root = block == 0 ? p->conv1_root : p->conv2_root;
            const float* bias = block == 0 ? p->conv1_bias : p->conv2_bias;
            for (int d = 0; d < p->depth; ++d) {
                MDNO_TRY(moment_conv(cur, ws.h2, row_ptr, src, R, p->ker_width, root, bias, MDNO_AGGR_MEAN, /*relu=*/1, nxt,
                                     mw, s, p->gemm_mode, application++));
                float* t = cur; cur = nxt; nxt = t;
            }
        }
        if (latent) MDNO_HIP(hipMemcpyAsync(latent, cur, sizeof(float) * (size_t)R * C, hipMemcpyDeviceToDevice, s));
        MDNO_TRY(fc_out(cur, p->fc2_w, p->fc2_b, R, C, p->out_width, out_frames, t_out, t_dev, s, tail));
        return MDNO_OK;
    } else {
        for (int block = 0; block < blocks; ++block) {
            if (block == 0 || separate_conv2_kernel(p)) {
                EdgeMlpWeights w = (block == 0)
                                       ? EdgeMlpWeights{p->k_w0, p->k_b0, p->k_w1, p->k_b1, p->k_w2, p->k_b2}
                                       : EdgeMlpWeights{p->k2_w0, p->k2_b0, p->k2_w1, p->k2_b1, p->k2_w2, p->k2_b2};
                MDNO_TRY(edge_mlp(edge_frames, edge_frame, t_dev, R, src, dst, edge_attr, perm, num_edges, edge_cap,
                                  p->ker_in, p->ker_width, C * C, p->gemm_mode, w, ws.w_e, ws.mlp, ws.mlp_bytes, s,
                                  block == 0 ? phase : (phase & ~WP_FLAGS_ZEROED)));   // (the caller zeroed the flags once)
            }
            if (prep_only) return MDNO_OK;
            const float* root = block == 0 ? p->conv1_root : p->conv2_root;
            const float* bias = block == 0 ? p->conv1_bias : p->conv2_bias;
            for (int d = 0; d < p->depth; ++d) {
                // the forward's last application also applies the output layer to its rows (and ends the step)
                const bool last = block + 1 == blocks && d + 1 == p->depth;
                const FcTail fc{p->fc2_w, p->fc2_b, p->out_width, out_frames, t_out, t_dev,
                                tail ? *tail : StepTail{nullptr, nullptr, nullptr, nullptr}, tail ? tail->row_done : nullptr};
                MDNO_TRY(nnconv(cur, row_ptr, src, R, ws.w_e, root, bias, C, C, MDNO_AGGR_MEAN, /*relu=*/1, nxt, s,
                                last ? &fc : nullptr, edge_cap));
                fc_done = last;
                float* t = cur; cur = nxt; nxt = t;
            }
        }
    }
    if (latent) MDNO_HIP(hipMemcpyAsync(latent, cur, sizeof(float) * (size_t)R * C, hipMemcpyDeviceToDevice, s));
    if (!fc_done) MDNO_TRY(fc_out(cur, p->fc2_w, p->fc2_b, R, C, p->out_width, out_frames, t_out, t_dev, s, tail));
    return MDNO_OK;
}

struct RolloutWs {
    int *row_ptr, *src, *dst, *num_edges, *t_dev, *row_done;
    void *fwd, *graph_scratch;
    size_t fwd_bytes, graph_scratch_bytes, total;
};

RolloutWs carve_rollout(void* ws, const mdno_kernelnn_params* p, int M, int N, long long edge_cap) {
    RolloutWs r{};
    Carver cv(ws);
    const size_t R = (size_t)M * N;
    r.row_ptr = cv.take<int>(R + 1);
    r.src = cv.take<int>((size_t)edge_cap);
    r.dst = cv.take<int>((size_t)edge_cap);
    r.num_edges = cv.take<int>(64);   // counters on their own 256-B line
    r.t_dev = r.num_edges + 1;        // (+2: finished workgroups of the step's last kernel, StepTail::done)
    r.row_done = cv.take<int>(R <= 256 ? 256 : 0);      // short chains: parts of a row finished (FcTail::row_done)
    r.fwd_bytes = carve_fwd(nullptr, p, M, N, edge_cap, use_factored(p, M, edge_cap, true)).total;
    r.fwd = cv.take<char>(r.fwd_bytes);
    r.graph_scratch_bytes = radius_graph_scratch_bytes(M, N);      // cell list of large members (0 otherwise)
    r.graph_scratch = cv.take<char>(r.graph_scratch_bytes);
    r.total = cv.used();
    return r;
}

}  // namespace
}  // namespace mdno

using namespace mdno;

extern "C" int mdno_abi_version(void) { return MDNO_ABI_VERSION; }

#ifndef MDNO_BUILD_ID
#error "compile through csrc/build.sh: it defines MDNO_BUILD_ID (the content hash of the sources)"
#endif
extern "C" const char* mdno_build_id(void) { return MDNO_BUILD_ID; }

extern "C" const char* mdno_last_error(void) { return g_last_error.c_str(); }

extern "C" size_t mdno_kernelnn_workspace_bytes(const mdno_kernelnn_params* p, int M, int N, int64_t edge_cap) {
    if (!p || M <= 0 || N <= 0 || edge_cap <= 0) return 0;
    // AUTO resolves per call (explicit edge attributes always run materialised): room for either
    const size_t a = carve_fwd(nullptr, p, M, N, (long long)edge_cap, use_factored(p, M, edge_cap, true)).total;
    const size_t b = carve_fwd(nullptr, p, M, N, (long long)edge_cap, use_factored(p, M, edge_cap, false)).total;
    return a > b ? a : b;
}

extern "C" int mdno_resolve_conv_mode(const mdno_kernelnn_params* p, int M, int64_t edge_cap) {
    return p && use_factored(p, M, (long long)edge_cap, true) ? MDNO_CONV_FACTORED : MDNO_CONV_MATERIALIZED;
}

extern "C" int mdno_conv_mode_for_graph(const mdno_kernelnn_params* p, int M, int N, int64_t num_edges) {
    if (!p || M <= 0 || N <= 0 || !factored_available(p)) return MDNO_CONV_MATERIALIZED;
    const long long rows = (long long)M * N;
    const bool dense = num_edges >= kAutoFactoredMinDegree * rows && num_edges >= (long long)M * kAutoFactoredMinEdgesPerMember;
    return dense ? MDNO_CONV_FACTORED : MDNO_CONV_MATERIALIZED;
}

extern "C" int mdno_kernelnn_fwd(const mdno_kernelnn_params* p, const float* frames, int M, int W, int N,
                                 const int64_t* x_aminoacid, int aa_per_member, const int32_t* row_ptr,
                                 const int32_t* src, const int32_t* dst, const int32_t* num_edges,
                                 int64_t edge_cap, const float* edge_pos, const float* edge_attr,
                                 const int32_t* perm, float* out, float* latent, void* workspace,
                                 size_t workspace_bytes, int32_t* status, void* stream) {
    MDNO_TRY(validate_params(p));
    MDNO_REQUIRE(frames && x_aminoacid && row_ptr && src && num_edges && out && workspace, MDNO_EINVAL,
                 "mdno_kernelnn_fwd: null pointer");
    MDNO_REQUIRE(M > 0 && W > 0 && N > 0 && edge_cap > 0, MDNO_EINVAL, "mdno_kernelnn_fwd: M=%d W=%d N=%d", M, W, N);
    FwdWs ws = carve_fwd(workspace, p, M, N, (long long)edge_cap,
                         use_factored(p, M, (long long)edge_cap, edge_pos && !edge_attr && dst));
    MDNO_REQUIRE(workspace_bytes >= ws.total, MDNO_EWORKSPACE, "mdno_kernelnn_fwd: workspace %zu < %zu",
                 workspace_bytes, ws.total);
    {   // the fallback counters of this forward (mdno_kernelnn_fallback_counts)
        int *conv = nullptr, *mlp = nullptr;
        fwd_counter_words(p, ws, M, N, (long long)edge_cap, &conv, &mlp);
        MDNO_TRY(zero_counter_words(conv, mlp, static_cast<hipStream_t>(stream)));
    }
    return forward_impl(p, frames, 0, nullptr, M, W, N, (const long long*)x_aminoacid, aa_per_member, row_ptr, src,
                        dst, num_edges, (long long)edge_cap, edge_pos, 0, edge_attr, perm, out, 0, latent, ws,
                        status, static_cast<hipStream_t>(stream));
}

extern "C" int mdno_kernelnn_fallback_counts(const mdno_kernelnn_params* p, int M, int N, int64_t edge_cap,
                                            int position_graph, void* workspace, int64_t counts[4], void* stream) {
    MDNO_TRY(validate_params(p));
    MDNO_REQUIRE(workspace && counts && M > 0 && N > 0 && edge_cap > 0, MDNO_EINVAL, "mdno_kernelnn_fallback_counts: bad arguments");
    const FwdWs ws = carve_fwd(workspace, p, M, N, (long long)edge_cap, use_factored(p, M, (long long)edge_cap, position_graph != 0));
    int *conv = nullptr, *mlp = nullptr;
    fwd_counter_words(p, ws, M, N, (long long)edge_cap, &conv, &mlp);
    return read_counter_words(conv, mlp, counts, static_cast<hipStream_t>(stream));
}

extern "C" size_t mdno_rollout_workspace_bytes(const mdno_kernelnn_params* p, int M, int N, int64_t edge_cap) {
    if (!p || M <= 0 || N <= 0 || edge_cap <= 0) return 0;
    return carve_rollout(nullptr, p, M, N, (long long)edge_cap).total;
}

namespace mdno {
namespace {
__global__ void set_step_kernel(int* t_dev, int v, int* row_done, int n_rows) {
    if (threadIdx.x == 0) {
        t_dev[0] = v;
        t_dev[1] = 0;      // StepTail::done
    }
    if ((int)threadIdx.x < n_rows) row_done[threadIdx.x] = 0;
}
}  // namespace
}  // namespace mdno

struct mdno_rollout_plan {
    mdno_kernelnn_params p;
    float* traj;
    int M, W, N, max_steps;
    const long long* aa;
    int aa_per_member;
    double threshold;
    long long edge_cap;
    RolloutWs r;
    FwdWs fw;
    int* edges_per_step;
    int* status;
    hipGraph_t graph;
    hipGraphExec_t exec;
    hipGraph_t graph_n;        // kStepsPerGraph steps in one graph (short chains: a graph launch costs two kernel launches)
    hipGraphExec_t exec_n;
    Timer* timer;
    bool weights_cached;   // weight-derived operands are rebuilt per plan_run call, not per step
};

static void plan_counter_words(const mdno_rollout_plan* pl, int** conv, int** mlp) {
    fwd_counter_words(&pl->p, pl->fw, pl->M, pl->N, pl->edge_cap, conv, mlp);
}

// The bf16 plane images of W1 (/W2) and W3T depend on the weights only: one workspace slot each, so
// they can be kept across steps when conv1 and conv2 share one edge-MLP (always, for KernelNN).
static int plan_prepare_weights(mdno_rollout_plan* pl, hipStream_t s) {
    const int W = pl->W;
    return forward_impl(&pl->p, pl->traj, 0, pl->r.t_dev, pl->M, W, pl->N, pl->aa, pl->aa_per_member, pl->r.row_ptr,
                        pl->r.src, pl->r.dst, pl->r.num_edges, pl->edge_cap, pl->traj, W - 1, nullptr,
                        nullptr, pl->traj, W, nullptr, pl->fw, pl->status, s, WP_PREPARE_ONLY);
}

static int plan_enqueue_step(mdno_rollout_plan* pl, hipStream_t s) {
    const int W = pl->W;
    // graph + edge attributes of the LAST window frame (graph_kernel.py:363, :375): frame W-1+t
    // the graph kernels also clear the activation flags of this step's edge-MLP, and the step's last kernel
    // moves the step counter on: two launches fewer per step (a launch is ~4 us; a 28-atom step is ~30 of them)
    int* act_flags = edge_mlp_activation_flags(pl->fw.mlp, pl->p.ker_width,
                                               pl->fw.factored ? pl->p.ker_width : pl->p.width * pl->p.width,
                                               pl->edge_cap, pl->p.gemm_mode);
    const int n_zero = act_flags ? kEdgeMlpActivationFlags : 0;
    const bool head = step_head_small_supported(pl->M, pl->N);      // short chain: graph and node prologue in one launch
    if (head)
        MDNO_TRY(step_head_small(&pl->p, pl->traj, W, pl->r.t_dev, pl->M, pl->N, pl->aa, pl->aa_per_member, pl->fw.xa,
                                 pl->threshold, pl->r.row_ptr, pl->r.src, pl->r.dst, pl->edge_cap, pl->r.num_edges,
                                 pl->status, act_flags, n_zero, s));
    else
        MDNO_TRY(radius_graph(pl->traj, W - 1, pl->r.t_dev, pl->M, pl->N, pl->threshold, pl->r.row_ptr, pl->r.src,
                              pl->r.dst, pl->edge_cap, pl->r.num_edges, pl->status, s, act_flags, n_zero,
                              pl->r.graph_scratch, pl->r.graph_scratch_bytes));
    const StepTail tail{pl->r.t_dev, pl->r.num_edges, pl->edges_per_step, pl->r.t_dev + 1,
                        (long long)pl->M * pl->N <= 256 ? pl->r.row_done : nullptr};
    return forward_impl(&pl->p, pl->traj, 0, pl->r.t_dev, pl->M, W, pl->N, pl->aa, pl->aa_per_member, pl->r.row_ptr,
                        pl->r.src, pl->r.dst, pl->r.num_edges, pl->edge_cap, pl->traj, W - 1, nullptr,
                        nullptr, pl->traj, W, nullptr, pl->fw, pl->status, s,
                        (pl->weights_cached ? WP_RUN_ONLY : WP_BOTH) | (act_flags ? WP_FLAGS_ZEROED : 0) |
                            (head ? WP_PROLOGUE_DONE : 0),
                        &tail);
}

constexpr int kStepsPerGraph = 8;

// `n` consecutive steps captured on `s` -> executable graph
static int capture_steps(mdno_rollout_plan* pl, hipStream_t s, int n, hipGraph_t* graph, hipGraphExec_t* exec) {
    hipError_t eb = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    MDNO_REQUIRE(eb == hipSuccess, MDNO_ELAUNCH, "hipStreamBeginCapture: %s", hipGetErrorString(eb));
    int rc = MDNO_OK;
    for (int t = 0; t < n && rc == MDNO_OK; ++t) rc = plan_enqueue_step(pl, s);
    hipError_t ec = hipStreamEndCapture(s, graph);
    if (rc != MDNO_OK || ec != hipSuccess || !*graph) {
        if (rc == MDNO_OK) set_error("hipStreamEndCapture: %s", hipGetErrorString(ec));
        if (*graph) (void)hipGraphDestroy(*graph);
        *graph = nullptr;
        return rc != MDNO_OK ? rc : MDNO_ELAUNCH;
    }
    hipError_t ei = hipGraphInstantiate(exec, *graph, nullptr, nullptr, 0);
    if (ei != hipSuccess) {
        set_error("hipGraphInstantiate: %s", hipGetErrorString(ei));
        (void)hipGraphDestroy(*graph);
        *graph = nullptr;
        *exec = nullptr;
        return MDNO_ELAUNCH;
    }
    return MDNO_OK;
}

extern "C" int mdno_rollout_plan_create(mdno_rollout_plan** plan, const mdno_kernelnn_params* p, float* traj, int M,
                                        int W, int N, int max_steps, const int64_t* x_aminoacid, int aa_per_member,
                                        double threshold, int64_t edge_cap, void* workspace,
                                        size_t workspace_bytes, int32_t* edges_per_step, int32_t* status,
                                        int use_graph, void* stream) {
    MDNO_REQUIRE(plan != nullptr, MDNO_EINVAL, "mdno_rollout_plan_create: null plan pointer");
    *plan = nullptr;
    MDNO_TRY(validate_params(p));
    MDNO_REQUIRE(traj && x_aminoacid && workspace, MDNO_EINVAL, "mdno_rollout_plan_create: null pointer");
    MDNO_REQUIRE(M > 0 && W > 0 && N > 0 && max_steps > 0 && edge_cap > 0, MDNO_EINVAL,
                 "mdno_rollout_plan_create: M=%d W=%d N=%d max_steps=%d", M, W, N, max_steps);
    MDNO_REQUIRE(p->out_width == 3, MDNO_EINVAL,
                 "rollout: out_width=%d, the model output must be a frame [N,3] (graph_kernel.py:407-410)",
                 p->out_width);
    RolloutWs r = carve_rollout(workspace, p, M, N, (long long)edge_cap);
    MDNO_REQUIRE(workspace_bytes >= r.total, MDNO_EWORKSPACE, "rollout: workspace %zu < %zu", workspace_bytes, r.total);
    mdno_rollout_plan* pl = new mdno_rollout_plan{};
    pl->p = *p;
    pl->traj = traj;
    pl->M = M; pl->W = W; pl->N = N; pl->max_steps = max_steps;
    pl->aa = (const long long*)x_aminoacid;
    pl->aa_per_member = aa_per_member;
    pl->threshold = threshold;
    pl->edge_cap = (long long)edge_cap;
    pl->r = r;
    pl->fw = carve_fwd(r.fwd, p, M, N, (long long)edge_cap, use_factored(p, M, (long long)edge_cap, true));
    pl->edges_per_step = edges_per_step;
    pl->status = status;
    pl->weights_cached = !separate_conv2_kernel(p);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (use_graph && s != nullptr) {
        int rc = capture_steps(pl, s, 1, &pl->graph, &pl->exec);
        // a short chain's step is a few dozen launches of a few microseconds: several steps per graph launch
        if (rc == MDNO_OK && max_steps >= kStepsPerGraph && step_head_small_supported(M, N) &&
            capture_steps(pl, s, kStepsPerGraph, &pl->graph_n, &pl->exec_n) != MDNO_OK) {
            // the many-steps graph is an optimisation: without it the single-step graph above replays every step.  Said
            // once on stderr (a short chain then runs at a fraction of its rate) and visible to the caller through
            // mdno_rollout_plan_steps_per_launch; the plan itself is good, so the error string is not left behind
            fprintf(stderr, "libmdno: the %d-steps-per-launch graph could not be built (%s); replaying single steps\n",
                    kStepsPerGraph, mdno_last_error());
            set_error("%s", "");
            pl->graph_n = nullptr;
            pl->exec_n = nullptr;
        }
        if (rc != MDNO_OK) {
            mdno_rollout_plan_destroy(pl);
            return rc;
        }
    }
    *plan = pl;
    return MDNO_OK;
}

extern "C" int mdno_rollout_plan_steps_per_launch(mdno_rollout_plan* plan) {
    if (plan == nullptr) return 0;
    if (plan->exec_n != nullptr) return kStepsPerGraph;
    return plan->exec != nullptr ? 1 : 0;
}

extern "C" int mdno_rollout_plan_run(mdno_rollout_plan* pl, int start_step, int steps, void* stream) {
    MDNO_REQUIRE(pl != nullptr, MDNO_EINVAL, "mdno_rollout_plan_run: null plan");
    MDNO_REQUIRE(start_step >= 0 && steps >= 0 && start_step + steps <= pl->max_steps, MDNO_EINVAL,
                 "mdno_rollout_plan_run: start_step=%d steps=%d exceed max_steps=%d", start_step, steps, pl->max_steps);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (steps == 0) return MDNO_OK;
    hipLaunchKernelGGL(set_step_kernel, dim3(1), dim3(256), 0, s, pl->r.t_dev, start_step, pl->r.row_done,
                       (long long)pl->M * pl->N <= 256 ? 256 : 0);
    MDNO_TRY(check_launch("set_step"));
    {   // the fallback counters of this run
        int *conv = nullptr, *mlp = nullptr;
        plan_counter_words(pl, &conv, &mlp);
        MDNO_TRY(zero_counter_words(conv, mlp, s));
    }
    // the weights may have been updated in place since the last call: refresh their images once
    if (pl->weights_cached) MDNO_TRY(plan_prepare_weights(pl, s));
    for (int t = 0; t < steps; ++t) {
        if (pl->exec_n && !pl->timer && steps - t >= kStepsPerGraph) {
            hipError_t el = hipGraphLaunch(pl->exec_n, s);
            MDNO_REQUIRE(el == hipSuccess, MDNO_ELAUNCH, "hipGraphLaunch(steps %d..): %s", t, hipGetErrorString(el));
            t += kStepsPerGraph - 1;
        } else if (pl->exec && !pl->timer) {
            hipError_t el = hipGraphLaunch(pl->exec, s);
            MDNO_REQUIRE(el == hipSuccess, MDNO_ELAUNCH, "hipGraphLaunch(step %d): %s", t, hipGetErrorString(el));
        } else {  // plain launches (always when a timer is attached: events cannot sit inside a replay)
            g_active_timer = pl->timer;
            const int rc = plan_enqueue_step(pl, s);
            g_active_timer = nullptr;
            MDNO_TRY(rc);
        }
    }
    return MDNO_OK;
}

extern "C" int mdno_rollout_plan_timer_attach(mdno_rollout_plan* pl, int max_records) {
    MDNO_REQUIRE(pl != nullptr && max_records > 0, MDNO_EINVAL, "mdno_rollout_plan_timer_attach: bad arguments");
    if (pl->timer) return MDNO_OK;
    Timer* t = new Timer();
    t->recs.resize((size_t)max_records);
    for (auto& r : t->recs) {
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) {
            set_error("hipEventCreate failed");
            delete t;
            return MDNO_ELAUNCH;
        }
    }
    pl->timer = t;
    return MDNO_OK;
}

extern "C" int mdno_rollout_plan_timer_read(mdno_rollout_plan* pl, int kernel_id, double* total_ms, int64_t* count) {
    MDNO_REQUIRE(pl && pl->timer && total_ms && count && kernel_id >= 0 && kernel_id < KID_COUNT, MDNO_EINVAL,
                 "mdno_rollout_plan_timer_read: bad arguments");
    double tot = 0.0;
    int64_t n = 0;
    for (size_t i = 0; i < pl->timer->used; ++i) {
        const auto& r = pl->timer->recs[i];
        if (r.kid != kernel_id) continue;
        float ms = 0.f;
        hipError_t e = hipEventElapsedTime(&ms, r.a, r.b);
        MDNO_REQUIRE(e == hipSuccess, MDNO_ELAUNCH, "hipEventElapsedTime: %s (synchronise the stream first)",
                     hipGetErrorString(e));
        tot += ms;
        ++n;
    }
    *total_ms = tot;
    *count = n;
    return MDNO_OK;
}

extern "C" int mdno_rollout_plan_timer_detach(mdno_rollout_plan* pl) {
    if (!pl || !pl->timer) return MDNO_OK;
    for (auto& r : pl->timer->recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    delete pl->timer;
    pl->timer = nullptr;
    return MDNO_OK;
}

extern "C" int mdno_rollout_plan_fallback_counts(mdno_rollout_plan* pl, int64_t counts[4], void* stream) {
    MDNO_REQUIRE(pl != nullptr && counts != nullptr, MDNO_EINVAL, "mdno_rollout_plan_fallback_counts: null argument");
    int *conv = nullptr, *mlp = nullptr;
    plan_counter_words(pl, &conv, &mlp);
    return read_counter_words(conv, mlp, counts, static_cast<hipStream_t>(stream));
}

extern "C" int mdno_rollout_plan_destroy(mdno_rollout_plan* pl) {
    if (!pl) return MDNO_OK;
    (void)mdno_rollout_plan_timer_detach(pl);
    if (pl->exec) (void)hipGraphExecDestroy(pl->exec);
    if (pl->graph) (void)hipGraphDestroy(pl->graph);
    if (pl->exec_n) (void)hipGraphExecDestroy(pl->exec_n);
    if (pl->graph_n) (void)hipGraphDestroy(pl->graph_n);
    delete pl;
    return MDNO_OK;
}

extern "C" int mdno_rollout(const mdno_kernelnn_params* p, float* traj, int M, int W, int N, int steps,
                            const int64_t* x_aminoacid, int aa_per_member, double threshold, int64_t edge_cap,
                            void* workspace, size_t workspace_bytes, int32_t* edges_per_step, int32_t* status,
                            int use_graph, void* stream) {
    MDNO_REQUIRE(steps >= 0, MDNO_EINVAL, "mdno_rollout: steps=%d", steps);
    if (steps == 0) return MDNO_OK;
    mdno_rollout_plan* pl = nullptr;
    MDNO_TRY(mdno_rollout_plan_create(&pl, p, traj, M, W, N, steps, x_aminoacid, aa_per_member, threshold, edge_cap,
                                      workspace, workspace_bytes, edges_per_step, status, use_graph,
                                      stream));
    int rc = mdno_rollout_plan_run(pl, 0, steps, stream);
    if (pl->exec) {
        // The executable graph must outlive its enqueued launches; this convenience call has no
        // object to park it in, so it waits for the stream before releasing it (see mdno.h).
        hipError_t es = hipStreamSynchronize(static_cast<hipStream_t>(stream));
        if (es != hipSuccess && rc == MDNO_OK) {
            set_error("hipStreamSynchronize after rollout: %s", hipGetErrorString(es));
            rc = MDNO_ELAUNCH;
        }
    }
    mdno_rollout_plan_destroy(pl);
    return rc;
}
