// Internal launch functions behind the C ABI.  They take one extra pair of arguments the public
// entry points do not expose: a frame index plus an optional DEVICE-side step counter, so that a
// captured rollout step (hipGraph) can be replayed unchanged while the window advances.
//   frames : f32 [T, R, 3] time-major, R = M*N rows per frame; frame f starts at frames + f*R*3
//   frame  : host index; effective frame = frame + (t_dev ? *t_dev : 0)
#pragma once
#include "common.h"

namespace mdno {

// Optional per-kernel timing with HIP events on the launching stream (bench.py's roofline leg).
// A timer is attached to a rollout plan by the caller; while an eager (non-graph) run of that plan
// enqueues kernels on this thread, every launch site below brackets its kernel with two events.
enum KernelId { KID_NNCONV = 0, KID_GEMM_L1 = 1, KID_GEMM_L2 = 2, KID_EDGE_L0 = 3, KID_GRAPH = 4,
                KID_PROLOGUE = 5, KID_FC_OUT = 6, KID_NNCONV_COMBINE = 7, KID_FACT_Y = 8, KID_COUNT = 9 };
struct Timer;
extern thread_local Timer* g_active_timer;
void timer_mark(int kid, bool start, hipStream_t s);
struct TimedSection {
    int kid; hipStream_t s; bool on;
    TimedSection(int k, hipStream_t st) : kid(k), s(st), on(g_active_timer != nullptr) { if (on) timer_mark(kid, true, s); }
    ~TimedSection() { if (on) timer_mark(kid, false, s); }
};

// zero_words / n_zero (<= 64): ints the graph kernels also reset — a rollout step clears the edge-MLP's
// activation flags here instead of with a launch of its own
// scratch (radius_graph_scratch_bytes; 0 below 8,192 atoms per member): with it a large member's graph is built
// through a cell list instead of N^2 pair tests — the same edges in the same order
size_t radius_graph_scratch_bytes(int M, int N);
int radius_graph(const float* frames, int frame, const int* t_dev, int M, int N, double cutoff, int* row_ptr,
                 int* src, int* dst, long long edge_cap, int* num_edges, int* status, hipStream_t s,
                 int* zero_words = nullptr, int n_zero = 0, void* scratch = nullptr, size_t scratch_bytes = 0);

struct EdgeMlpWeights {
    const float *w0, *b0, *w1, *b1, *w2, *b2;
};

// Weight-derived operands (bf16 plane images, W3T) live in the caller's workspace.  A rollout plan
// builds them once per mdno_rollout_plan_run (WP_PREPARE_ONLY) and its steps reuse them (WP_RUN_ONLY);
// one-off calls do both (WP_BOTH).
// WP_FLAGS_ZEROED (OR-ed in): the caller has already reset the activation flags of this forward
// (edge_mlp_activation_flags), the edge-MLP launches no kernel for it.
// WP_PROLOGUE_DONE (OR-ed in, engine only): the node features of this forward are already in place.
enum WeightPhase { WP_BOTH = 0, WP_PREPARE_ONLY = 1, WP_RUN_ONLY = 2, WP_PHASE_MASK = 3, WP_FLAGS_ZEROED = 4,
                   WP_PROLOGUE_DONE = 8 };
constexpr int kEdgeMlpActivationFlags = 3;
// the three per-forward flag words of the SPLIT_F16 edge-MLP inside `workspace` (NULL in the other modes);
// out_dim = ker_width for edge_mlp_hidden
int* edge_mlp_activation_flags(void* workspace, int ker_width, int out_dim, long long edge_cap, int gemm_mode);
int* edge_mlp_split_activation_flags(void* workspace, int ker_width, int out_dim, long long chunk);

// gemm_mode: MDNO_GEMM_SPLIT_BF16 (default; falls back to exact fp32 when the shape is not tileable)
// or MDNO_GEMM_F32.
int edge_mlp(const float* frames, int frame, const int* t_dev, int rows_per_frame, const int* src, const int* dst,
             const float* edge_attr, const int* perm, const int* num_edges, long long edge_cap, int ker_in,
             int ker_width, int out_dim, int gemm_mode, const EdgeMlpWeights& w, float* w_e, void* workspace,
             size_t workspace_bytes, hipStream_t s, int phase = WP_BOTH);
bool edge_mlp_split_supported(int ker_width, int out_dim);
size_t edge_mlp_split_workspace_bytes(int ker_width, int out_dim, long long chunk);
int edge_mlp_split(const float* frames, int frame, const int* t_dev, int rows_per_frame, const int* src,
                   const int* dst, const float* edge_attr, const int* perm, const int* num_edges,
                   long long edge_cap, long long chunk, int ker_in, int ker_width, int out_dim,
                   const EdgeMlpWeights& w, float* w_e, void* workspace, hipStream_t s, int phase = WP_BOTH,
                   bool f16 = false);

// Edge-MLP up to its last hidden activation: H = relu(L1(relu(L0(attr)))) as fp32 [edge_cap, ker_width]
// row-major (what the factored conv consumes); same attr modes and gemm_mode as edge_mlp.
int edge_mlp_hidden(const float* frames, int frame, const int* t_dev, int rows_per_frame, const int* src,
                    const int* dst, const float* edge_attr, const int* perm, const int* num_edges,
                    long long edge_cap, int ker_in, int ker_width, int gemm_mode, const EdgeMlpWeights& w,
                    float* h_out, void* workspace, size_t workspace_bytes, hipStream_t s, int phase = WP_BOTH);
int edge_mlp_split_hidden(const float* frames, int frame, const int* t_dev, int rows_per_frame, const int* src,
                          const int* dst, const float* edge_attr, const int* perm, const int* num_edges,
                          long long edge_cap, long long chunk, int ker_in, int ker_width, const EdgeMlpWeights& w,
                          float* h_out, void* workspace, hipStream_t s, int phase = WP_BOTH, bool f16 = false);

// Pieces of the split-bf16 GEMM usable on their own (edge_mlp_split.hip): fp32 [rows,K] -> tiled bf16
// planes (buffer of split_planes_bytes), and C[rows,N] = A . Bt^T (fp32 row-major) from two such images.
size_t split_planes_bytes(long long rows, int K);
int split_planes(const float* a, int rows, int K, void* planes, hipStream_t s);
int fill_ints(int* p, int n, int value, hipStream_t s);     // n <= 256, by a kernel
// two-plane fp16 images (SPLIT_F16)
size_t split_planes_f16_bytes(long long rows, int K);
// C [n1,n2] (+)= A^T . B for fp32 operands on two fp16 planes each, columns scaled by powers of two (gemm_bf16.hip)
bool gemm_atb_f16_supported(long long rows, int n1, int n2);
size_t gemm_atb_f16_workspace_bytes(long long rows, int n1, int n2);
int gemm_atb_f16(const float* a, const float* b, long long rows, int n1, int n2, float* c, int accumulate, void* workspace,
                 hipStream_t s);
// act(A . W^T + b) on two fp16 planes per operand, rows of both scaled by powers of two (training: split_linear's
// shapes at half the matrix work)
size_t split_linear_f16_workspace_bytes(long long rows, int N, int K);
int split_linear_f16(const float* a, const float* w, const float* bias, long long rows, int N, int K, int relu, float* c,
                     void* workspace, hipStream_t s);
// (each row scaled by a power of two into fp16's upper binades; unscale[row] = the factor that undoes it)
int split_planes_f16(const float* a, int rows, int K, void* planes, float* unscale, int* range_flag, hipStream_t s);
// C = act(A . W^T + b) with both operands split on the way in (training ops)
size_t split_linear_workspace_bytes(long long rows, int N, int K);
bool split_linear_supported(long long rows, int N, int K);
int split_linear(const float* a, const float* w, const float* bias, long long rows, int N, int K, int relu, float* c,
                 void* workspace, hipStream_t s);

// Factored conv, destination-side form (moment.hip: S_t = sum_{e->t} x_src (x) h_e, then y_t = W3 : S_t), the one
// factored formulation: on three bf16 planes per operand for the split GEMM modes, on the fp32 MFMA for gemm_mode F32
// (`exact_f32`; h2 is the k-tiled fp32 image [e/128][k/32][128][32] either way).
struct MomentWs {
    float *w3r, *s, *part;
    int* order;               // destinations of each S chunk by decreasing degree
    long long part_stride;
    // gemm_mode SPLIT_F16: W3R on two fp16 planes with its columns' inverse scales, and the row maxima of the S chunk
    _Float16* w3h;
    float* colinv;
    int* colmax_bits;
    float* rowmax;
    float* xm[2];             // every node's largest |feature|: of a conv application's input and of its output
    int* counters;            // 64 ints: [0] K1 workgroups rerun on bf16 planes, [1] destinations with unscaled operands (SPLIT_F16)
};
bool moment_supported(int width, int ker_width);
size_t moment_workspace_bytes(int num_rows, int ker_width);
MomentWs moment_carve(void* ws, int num_rows, int ker_width);
int moment_prepare_weights(const float* w3, const float* b3, int ker_width, const MomentWs& f, hipStream_t s, int gemm_mode);
int moment_prepare_graph(const int* row_ptr, int num_rows, const MomentWs& f, hipStream_t s);
// (the last MLP layer's bias b3 is part of W3R: moment_prepare_weights)
int moment_conv(const float* x, const float* h2, const int* row_ptr, const int* src, int num_rows, int ker_width,
                const float* root, const float* bias, int aggr, int relu, float* y, const MomentWs& f, hipStream_t s,
                int gemm_mode, int application);
// SPLIT_F16, before a forward's application 0: every row's largest |x| (the later applications get it from the one before)
int moment_row_absmax(const float* x, int num_rows, const MomentWs& f, hipStream_t s);

// bf16 training GEMMs (gemm_bf16.hip): 256 x 256 tiles, 8 waves, two wave groups one phase apart over an
// LDS-DMA ring of 32-k stages.
//   gemm_nt_pp         C = act(A . W^T + b), A bf16 [rows,K], W bf16 [N,K], C bf16 or fp32 (N % 256 == 0, K % 32 == 0, K >= 64)
//   gemm_nt_pp_masked  C bf16 = (Y > 0) ? A . W^T : 0 — Linear+ReLU input gradient with the mask fused (Y bf16 [rows,N])
//   gemm_tn_pp         C fp32 [n1,n2] = A^T . B over rows (n1, n2 % 256 == 0); K slices added in fixed order
bool gemm_nt_pp_supported(long long rows, int N, int K);
int gemm_nt_pp(const void* A, const void* W, const float* bias, long long rows, int N, int K, int relu, int out_bf16,
               void* C, hipStream_t s);
int gemm_nt_pp_masked(const void* A, const void* W, const void* Y, long long rows, int N, int K, void* C, hipStream_t s);
bool gemm_tn_pp_supported(long long rows, int n1, int n2);
size_t gemm_tn_pp_workspace_bytes(long long rows, int n1, int n2);
int gemm_tn_pp(const void* A, const void* B, long long rows, int n1, int n2, float* C, void* workspace, hipStream_t s);

// End of a rollout step, done by the workgroup of the step's last kernel that finishes last (every workgroup has
// read *t_dev by then): edges_per_step[t] = *num_edges, *t_dev = t + 1.  `done` counts finished workgroups and is
// left at 0.
struct StepTail {
    int* t_dev;
    const int* num_edges;
    int* edges_per_step;
    int* done;
    int* row_done = nullptr;   // [rows] zeroed ints, or NULL (FcTail::row_done)
};
// The model's output layer on a conv application's rows (fc_out below, same arithmetic): the last application of a
// forward writes out[(t_out + *t_dev) * rows + r][0..out_width) = w . y[r] + b next to y itself.
struct FcTail {
    const float* w;
    const float* b;
    int out_width;
    float* out;
    int t_out;
    const int* t_dev;
    StepTail step;      // step.done == NULL: none
    int* row_done = nullptr;   // [rows] zeroed ints: lets a row be shared by several workgroups (nnconv64_colsplit_kernel)
};
int nnconv(const float* x, const int* row_ptr, const int* src, int num_rows, const float* w_e, const float* root,
           const float* bias, int Cin, int Cout, int aggr, int relu, float* y, hipStream_t s,
           const FcTail* fc = nullptr, long long edge_cap = -1);      // edge_cap: bound on the edge count, if known

int node_prologue(const mdno_kernelnn_params* p, const float* frames, int t0, const int* t_dev, int M, int W, int N,
                  const long long* aa, int aa_per_member, float* x0, int* status, hipStream_t s);

// Head of a rollout step on a short chain (M * N <= 128 rows): radius graph of frame W-1+t and node prologue of the
// window at frame t in ONE launch (they are independent; each is shorter than a launch).
bool step_head_small_supported(int M, int N);
int step_head_small(const mdno_kernelnn_params* p, const float* frames, int W, const int* t_dev, int M, int N,
                    const long long* aa, int aa_per_member, float* x0, double cutoff, int* row_ptr, int* src, int* dst,
                    long long edge_cap, int* num_edges, int* status, int* zero_words, int n_zero, hipStream_t s);

int fc_out(const float* x, const float* w, const float* b, int rows, int width, int out_width, float* out_frames,
           int t_out, const int* t_dev, hipStream_t s, const StepTail* tail = nullptr);

}  // namespace mdno
