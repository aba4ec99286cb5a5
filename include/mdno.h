/*
 * mdno.h — C ABI of libmdno.so: the MI355X (gfx950) implementation of the graph-kernel
 * neural-operator hot path of ramanathanlab/molecular_dynamics_neural_operator.
 *
 * The reference has NO native/FFI boundary for this path: it is a Python torch.nn.Module API
 * (graph_kernel.py) over implicit PyTorch / torch_geometric device ops.  This header is therefore
 * the boundary a maintainer binds with ctypes; every entry point names the
 * reference code it replaces as file:line under the reference root.
 *
 * Conventions (all entry points)
 *   - return 0 on success, a negative MDNO_E* code on failure; mdno_last_error() gives the message
 *     of the calling thread's most recent failure.
 *   - every pointer is a DEVICE pointer owned by the caller unless marked [host]; nothing is
 *     allocated or freed on the caller's behalf; sizes are explicit.
 *   - `stream` is a hipStream_t passed as void*; work is enqueued asynchronously and not
 *     synchronised (one exception: mdno_rollout with use_graph != 0 waits for `stream` before it
 *     returns, to release the captured graph).  Entry points are re-entrant; there is no global
 *     state besides the per-thread error string (and per-device "attribute already raised" caches).
 *   - graphs are CSR over DESTINATION rows: row r (= member*N + atom) lists, in ascending order,
 *     the SOURCE nodes j of its in-edges (j -> r), self-loop included.  "Edge p" is position p of
 *     that list; per-edge tensors (W_e) are stored in this order, so every row's edges are one
 *     contiguous run and aggregation needs no atomics (run-to-run bitwise reproducible).
 *   - float = IEEE fp32; positions are Angstrom, frames are float32 [n_atoms,3] (dataset.py:159).
 */
#ifndef MDNO_H
#define MDNO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDNO_ABI_VERSION 15

#define MDNO_OK            0
#define MDNO_EINVAL       -1   /* bad argument (null pointer, non-positive size, unsupported dim) */
#define MDNO_ELAUNCH      -2   /* HIP runtime error while enqueuing */
#define MDNO_EWORKSPACE   -3   /* workspace too small */
#define MDNO_EUNSUPPORTED -4   /* shape outside what the kernels implement */

#define MDNO_AGGR_ADD  0
#define MDNO_AGGR_MEAN 1
#define MDNO_AGGR_MAX  2   /* mdno_nnconv_fwd only (inference): per-channel max over a node's messages, 0 for no message */

/* How the two wide edge-MLP GEMMs are evaluated (fp32 in, fp32 out either way):
 *   SPLIT_BF16  every fp32 operand is split exactly into 3 bf16 planes and the product accumulated
 *               in fp32 from the 6 leading plane products on the bf16 matrix pipe; error vs fp64 at
 *               the level of a plain fp32 GEMM (dropped terms <= 2^-24 |a b|).  Shapes the
 *               tiles do not divide (ker_width % 128, out_dim % 128) silently use F32.
 *   F32         v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 fmaf chain. */
#define MDNO_GEMM_SPLIT_BF16 0
#define MDNO_GEMM_F32        1
/*   SPLIT_F16   as SPLIT_BF16, except that the wide edge-MLP GEMMs and the two products of the factored conv
 *               (csrc/moment.hip: K1, K2) run on TWO fp16 planes (22-23 mantissa bits) and three plane
 *               products with fp32 accumulation: half the matrix work, error vs fp64 still below a plain
 *               fp32 GEMM's.  fp16 has 30 binades, so every use keeps its operands in range by exact powers
 *               of two (weight rows / columns and the rows of the conv's moment image by their own maxima)
 *               or checks them on the device: an edge-MLP chunk with a value out of range is redone by the
 *               SPLIT_BF16 kernels inside the same forward, a conv workgroup reruns its own destination on
 *               the bf16 planes (no host involvement either way), so results are fp32-accurate for any
 *               input that fp32 itself can carry.  The Python host side's default. */
#define MDNO_GEMM_SPLIT_F16  2

/* How a conv application is evaluated inside mdno_kernelnn_fwd / the rollout (same function either way):
 *   MATERIALIZED  the reference's formulation: W_e = net(edge_attr) [E,Cin,Cout] is written once per
 *                 forward and streamed by every conv application (HBM-bound gather/matvec/scatter).
 *   FACTORED      S_t = sum_{e -> t} x_src(e) (x) h_e per destination, then y_t = W3 : S_t (csrc/moment.hip): a
 *                 reassociation of the same sums that never forms W_e.  Any destination-sorted graph at
 *                 width 64 and ker_width % 128 == 0, in every GEMM mode; everything else runs MATERIALIZED. */
#define MDNO_CONV_MATERIALIZED 0
#define MDNO_CONV_FACTORED     1
/*   AUTO          FACTORED where it applies and the graph is large enough to pay for its fixed cost
 *                 per application (the Y GEMM and a 16-iteration pipeline per source: edge capacity
 *                 per member >= 24,576), MATERIALIZED otherwise (small graphs: 2x faster at N=28..120).
 *                 The library sees a CAPACITY only: a caller that knows its graphs should pass FACTORED for
 *                 dense ones (mean degree >= ~40 and >= ~16k edges per member) and MATERIALIZED otherwise —
 *                 protein-like chains of any length; the Python RolloutEngine does so from the window it is
 *                 reset with. */
#define MDNO_CONV_AUTO         2

/* status word bits written by device code (read back by the caller after synchronising) */
#define MDNO_STATUS_EDGE_OVERFLOW 1   /* radius graph found more than edge_cap edges; list truncated */
#define MDNO_STATUS_BAD_AMINOACID 2   /* x_aminoacid outside [0, num_embeddings) */
/* (bits 4 and 8 belonged to the source-side factored kernels removed with ABI 14: never set) */
#define MDNO_STATUS_BAD_EDGE_INDEX  16 /* mdno_coo_to_csr: a node id outside [0, num_nodes) (clamped in bounds;
                                          the reference's index_select / scatter raise IndexError there) */

int         mdno_abi_version(void);
const char* mdno_last_error(void);
/* Identity of the sources this binary was compiled from: the first 16 hex digits of sha256 over the bytes of
 * the .hip, .h and .sh files of csrc/ and include/mdno.h in C-locale name order (csrc/build.sh computes it; the
 * Python binding recomputes it from the tree at load time and refuses a library that was built from other
 * sources — the built .so travels to the GPU box outside version control). */
const char* mdno_build_id(void);

/* ------------------------------------------------------------------------------------------
 * Parameters of KernelNN under the reference's state_dict key names (graph_kernel.py:246-275).
 * All weights fp32, torch layouts: Linear.weight [out,in]; LSTM weight_ih_l0/weight_hh_l0 [4H,H]
 * gate order i,f,g,o; NNConv_old.root [Cin,Cout].  conv1.net and conv2.net are ONE module in the
 * reference (graph_kernel.py:271-273): set k2_* = NULL to share k_* (edge weights then evaluated
 * once per forward instead of 2*depth times — same values, the inputs never change, :278-302).
 * Notebook-era variant (bba_analysis.ipynb:123-128: emb, fc1, conv1, fc2 only, window 1): set all
 * lstm_* pointers and conv2_root/conv2_bias to NULL — the node feature is then [emb, newest frame]
 * and only `depth` conv1 applications run.
 * ---------------------------------------------------------------------------------------- */
typedef struct mdno_kernelnn_params {
    int32_t width, ker_width, depth, ker_in, in_width, out_width;
    int32_t num_embeddings, embedding_dim, x_position_dim, gemm_mode;  /* gemm_mode: MDNO_GEMM_* */
    int32_t conv_mode, reserved0;                                       /* conv_mode: MDNO_CONV_* */
    const float *lstm_w_ih, *lstm_w_hh, *lstm_b_ih, *lstm_b_hh;     /* lstm.*_l0            */
    const float *lstm_fc_w, *lstm_fc_b;                             /* lstm_fc.{weight,bias} */
    const float *emb_w;                                             /* emb.weight [20,4]     */
    const float *fc1_w, *fc1_b;                                     /* fc1 [width,in_width]  */
    const float *k_w0, *k_b0, *k_w1, *k_b1, *k_w2, *k_b2;           /* conv1.net.layers.{0,2,4} */
    const float *k2_w0, *k2_b0, *k2_w1, *k2_b1, *k2_w2, *k2_b2;     /* conv2.net.* or NULL = shared */
    const float *conv1_root, *conv1_bias, *conv2_root, *conv2_bias; /* [width,width], [width] */
    const float *fc2_w, *fc2_b;                                     /* fc2 [out_width,width] */
} mdno_kernelnn_params;

/* ------------------------------------------------------------------------------------------
 * K0/K1  radius graph  — replaces construct_pairdata's scipy distance_matrix + coo_matrix +
 * per-edge Python loop (graph_kernel.py:362-379; notebook variant bba_analysis.ipynb:302-320).
 *   pos       f32 [M*N,3]  one frame per member (M independent members, no cross-member edges)
 *   cutoff    f64; pair kept iff sqrt(dx^2+dy^2+dz^2) < cutoff with the sum and sqrt in f64 on the
 *             f32 coordinates (scipy promotes to f64), strict <, self-loops kept
 *   row_ptr   i32 [M*N+1] out   src i32 [edge_cap] out   dst i32 [edge_cap] out (may be NULL)
 *   num_edges i32 [1] out (device)   status i32 [1] in/out (device, OR-ed; may be NULL)
 * Row-major COO of the reference == (dst, src) pairs of this CSR read in order (the contact map is
 * symmetric), i.e. reference edge_index = [expand(row_ptr); src].
 * ---------------------------------------------------------------------------------------- */
int mdno_radius_graph_csr(const float* pos, int M, int N, double cutoff,
                          int32_t* row_ptr, int32_t* src, int32_t* dst, int64_t edge_cap,
                          int32_t* num_edges, int32_t* status, void* stream);
/* The same graph with caller-provided scratch: from 8,192 atoms per member on, a cell list (cells of edge >= cutoff,
 * 27 cells tested per destination, sources read back in ascending order from an atom mask) replaces the N^2 pair
 * tests — same pair test, same edges, same order, bit for bit.  mdno_radius_graph_workspace_bytes is 0 below that
 * size (workspace may then be NULL: this entry equals mdno_radius_graph_csr). */
size_t mdno_radius_graph_workspace_bytes(int M, int N);
int mdno_radius_graph_csr_ws(const float* pos, int M, int N, double cutoff, int32_t* row_ptr, int32_t* src, int32_t* dst,
                             int64_t edge_cap, int32_t* num_edges, int32_t* status, void* workspace,
                             size_t workspace_bytes, void* stream);

/* General graphs: stable sort of a COO edge list by destination — what torch_geometric's
 * scatter over edge_index[1] implies (graph_kernel.py:198 -> MessagePassing.propagate).
 *   edge_index i64 [2,E] (row 0 = source, row 1 = target)
 *   row_ptr i32 [num_nodes+1], src i32 [E], dst i32 [E] (may be NULL), perm i32 [E]: CSR position
 *   p holds input edge perm[p].  Own counting sort (count, scan, slot, per-row rank sort of the
 *   unique edge ids): deterministic, no vendor sort.  Node ids outside [0, num_nodes) — where the
 *   reference's index_select / scatter raise — set MDNO_STATUS_BAD_EDGE_INDEX in `status`
 *   (i32 [1] device, OR-ed; may be NULL) and are clamped so that nothing is read out of bounds.
 *   num_edges i32 [1] out (device, may be NULL): E, for the entry points that take the count from the device.
 *   Workspace size from mdno_coo_to_csr_workspace_bytes. */
size_t mdno_coo_to_csr_workspace_bytes(int64_t E, int num_nodes);
int mdno_coo_to_csr(const int64_t* edge_index, int64_t E, int num_nodes,
                    int32_t* row_ptr, int32_t* src, int32_t* dst, int32_t* perm, int32_t* num_edges,
                    int32_t* status, void* workspace, size_t workspace_bytes, void* stream);
/* The edges of a destination-sorted CSR (its src / dst arrays, E entries) grouped by SOURCE — what the input
 * gradient of a conv application walks (training; autograd's scatter in the reference, graph_kernel.py:198
 * differentiated): row j lists, in ascending CSR position, the targets nbr[q] of source j; rowid[q] = j (may be
 * NULL); perm[q] = the position of that edge in the destination-sorted CSR (and in W_e).  Same sort, same
 * workspace size as mdno_coo_to_csr. */
int mdno_csr_by_source(const int32_t* csr_src, const int32_t* csr_dst, int64_t E, int num_nodes,
                       int32_t* row_ptr, int32_t* nbr, int32_t* rowid, int32_t* perm, int32_t* status,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * K2  edge-MLP — replaces DenseNet.forward (graph_kernel.py:239-242) as called from
 * NNConv_old.message (:201): W_e = L3(relu(L2(relu(L1(edge_attr))))) -> f32 [E, out_dim],
 * written in CSR edge order; row e viewed [Cin,Cout] row-major (:201 .view(-1,Cin,Cout)).
 * Edge attributes come from ONE of
 *   (a) edge_pos f32 [R,3] + CSR (src,dst): attr[p] = [pos[src[p]], pos[dst[p]]]   (:372-379)
 *   (b) edge_attr f32 [E,ker_in] (+ perm i32 [E] or NULL): attr[p] = edge_attr[perm[p]]
 *   num_edges  i32 [1] device (rows >= *num_edges are not computed); edge_cap bounds it.
 *   workspace: mdno_edge_mlp_workspace_bytes(ker_width, out_dim, edge_cap, gemm_mode).
 * ---------------------------------------------------------------------------------------- */
size_t mdno_edge_mlp_workspace_bytes(int ker_width, int out_dim, int64_t edge_cap, int gemm_mode);
int mdno_edge_mlp_fwd(const float* edge_pos, const int32_t* src, const int32_t* dst,
                      const float* edge_attr, const int32_t* perm,
                      const int32_t* num_edges, int64_t edge_cap, int ker_in, int ker_width, int out_dim,
                      int gemm_mode,
                      const float* w0, const float* b0, const float* w1, const float* b1,
                      const float* w2, const float* b2,
                      float* w_e, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * K3-K6  conv application — replaces NNConv_old.forward/message/update (graph_kernel.py:194-209)
 * and torch_geometric's gather + scatter-mean:
 *   y[r] = act( aggr_{p in row r} ( x[src[p]] . W_e[p] )  +  x[r] . root  +  bias )
 *   aggr = sum (MDNO_AGGR_ADD), sum / max(deg,1) (MDNO_AGGR_MEAN) or the per-channel maximum over the row's
 *   messages, 0 for a row without any (MDNO_AGGR_MAX — torch_geometric's "max", NNConv_old docstring
 *   graph_kernel.py:148-150); act = ReLU if relu != 0
 *   (the ReLU of graph_kernel.py:300/302 fused).  root / bias may be NULL.
 *   x f32 [R,Cin]  W_e f32 [E,Cin,Cout]  y f32 [R,Cout]; y must not alias x.
 * ---------------------------------------------------------------------------------------- */
int mdno_nnconv_fwd(const float* x, const int32_t* row_ptr, const int32_t* src, int num_rows,
                    const float* w_e, const float* root, const float* bias,
                    int Cin, int Cout, int aggr, int relu, float* y, void* stream);

/* ------------------------------------------------------------------------------------------
 * Frame buffers are TIME-MAJOR: frames f32 [T, M, N, 3]; frame t of all M members is one contiguous
 * [M*N, 3] block (for M = 1 this is exactly PairData.x_position [W,N,3], dataset.py:185/207).
 *
 * K7/K8  node prologue — replaces graph_kernel.py:279-298: W sequential LSTM(3,3) cells over the
 * window (batch = atoms, zero initial state), lstm_fc, Embedding lookup, concat, fc1, ReLU.
 *   frames f32 [W, M, N, 3]
 *   x_aminoacid i64 [N] (aa_per_member = 0: shared by all members) or [M*N] (aa_per_member = 1)
 *   x0 f32 [M*N, width] out
 * ---------------------------------------------------------------------------------------- */
int mdno_node_prologue_fwd(const mdno_kernelnn_params* p, const float* frames, int M, int W, int N,
                           const int64_t* x_aminoacid, int aa_per_member,
                           float* x0, int32_t* status, void* stream);

/* K9  output projection — replaces fc2 (graph_kernel.py:305): out[r,:] = x[r] . W^T + b,
 * x f32 [rows,width], w f32 [out_width,width], out f32 [rows,out_width]. */
int mdno_fc_out_fwd(const float* x, const float* w, const float* b, int rows, int width, int out_width,
                    float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Whole forward — replaces KernelNN.forward (graph_kernel.py:277-309) for M independent samples
 * (B=1 semantics each).  frames f32 [W,M,N,3].  Graph given as CSR over the M*N rows; edge
 * attributes by (a) edge_pos f32 [M*N,3] (the frame the graph was built on) or (b) edge_attr
 * (+perm), as in mdno_edge_mlp_fwd.  out f32 [M*N,out_width]; latent f32 [M*N,width] (the
 * return_latent=True output, :303) may be NULL.
 * Workspace: mdno_kernelnn_workspace_bytes(p, M, N, edge_cap).
 * ---------------------------------------------------------------------------------------- */
size_t mdno_kernelnn_workspace_bytes(const mdno_kernelnn_params* p, int M, int N, int64_t edge_cap);
/* The formulation (MDNO_CONV_MATERIALIZED / MDNO_CONV_FACTORED) a forward or rollout with these
 * parameters, M members and this total edge capacity runs on a position-derived radius graph
 * (resolves AUTO; the rule is on edge_cap / M, so it does not depend on the batch a member is in). */
int mdno_resolve_conv_mode(const mdno_kernelnn_params* p, int M, int64_t edge_cap);
/* The formulation AUTO should run for a radius graph the caller has COUNTED: M members of N atoms with num_edges
 * directed edges in total (self-loops included).  FACTORED for dense graphs — mean degree >= 40 AND >= 16,384 edges
 * per member (measured break-even, DESIGN.md table of deviations) — where the model's dimensions allow it;
 * MATERIALIZED otherwise (protein-like chains of any length: 10-30 neighbours within 8 A).  Pure host function, no
 * launch; p->conv_mode is ignored.  A caller with AUTO in its params and a known graph sets p->conv_mode to the
 * result before mdno_rollout_plan_create / mdno_kernelnn_fwd (the Python RolloutEngine does, on the window it is
 * reset with); mdno_resolve_conv_mode remains the capacity-only rule the library applies by itself.
 * Replaces nothing in the reference (graph_kernel.py:194-209 has one formulation). */
int mdno_conv_mode_for_graph(const mdno_kernelnn_params* p, int M, int N, int64_t num_edges);
int mdno_kernelnn_fwd(const mdno_kernelnn_params* p, const float* frames, int M, int W, int N,
                      const int64_t* x_aminoacid, int aa_per_member,
                      const int32_t* row_ptr, const int32_t* src, const int32_t* dst,
                      const int32_t* num_edges, int64_t edge_cap,
                      const float* edge_pos, const float* edge_attr, const int32_t* perm,
                      float* out, float* latent, void* workspace, size_t workspace_bytes,
                      int32_t* status, void* stream);

/* ------------------------------------------------------------------------------------------
 * Autoregressive rollout — replaces recursive_propagation (graph_kernel.py:396-413) and the
 * notebook's propogate (bba_analysis.ipynb:336-358) with the whole loop on the device:
 *   for s in 0..steps-1:  graph + edge attrs of frame s+W-1 (the last window frame, :363,:375)
 *                         -> forward on frames s..s+W-1 -> frame s+W       (no host crossing)
 *   traj f32 [W+steps, M, N, 3] in/out (time-major): caller fills frames 0..W-1 (the start window,
 *   :401); frames W.. are produced.  Members are independent (block-diagonal graph).
 *   edges_per_step i32 [steps] out (device, may be NULL): E of each step's graph.
 *   use_graph != 0 and stream != NULL: one step is stream-captured into a hipGraph and replayed
 *   `steps` times (a device-side step counter advances the window); otherwise plain launches.
 * ---------------------------------------------------------------------------------------- */
size_t mdno_rollout_workspace_bytes(const mdno_kernelnn_params* p, int M, int N, int64_t edge_cap);
int mdno_rollout(const mdno_kernelnn_params* p, float* traj, int M, int W, int N, int steps,
                 const int64_t* x_aminoacid, int aa_per_member, double threshold, int64_t edge_cap,
                 void* workspace, size_t workspace_bytes, int32_t* edges_per_step,
                 int32_t* status, int use_graph, void* stream);

/* The same loop as a reusable plan: the step is captured once at creation (on `stream`, nothing
 * executes), then any range of steps is replayed without re-capturing and without synchronising.
 *   traj f32 [W+max_steps, M, N, 3]; run(start_step, steps) produces frames W+start_step ..
 *   W+start_step+steps-1 from the frames before them.  The plan keeps a copy of *p (the weight
 *   pointers must stay valid) and must be destroyed only after its enqueued work has completed. */
typedef struct mdno_rollout_plan mdno_rollout_plan;
int mdno_rollout_plan_create(mdno_rollout_plan** plan, const mdno_kernelnn_params* p, float* traj,
                             int M, int W, int N, int max_steps,
                             const int64_t* x_aminoacid, int aa_per_member, double threshold,
                             int64_t edge_cap, void* workspace, size_t workspace_bytes,
                             int32_t* edges_per_step, int32_t* status, int use_graph, void* stream);
int mdno_rollout_plan_run(mdno_rollout_plan* plan, int start_step, int steps, void* stream);
/* How the plan replays: 0 = plain launches (no graph), 1 = one captured step per graph launch, 8 = eight steps per
 * graph launch (short chains, M*N <= 128 rows, whose step is a few dozen launches of a few microseconds; ranges that
 * are no multiple of 8 finish on the one-step graph).  If the eight-step capture fails at creation the plan falls
 * back to 1 and says so once on stderr. */
int mdno_rollout_plan_steps_per_launch(mdno_rollout_plan* plan);
int mdno_rollout_plan_destroy(mdno_rollout_plan* plan);

/* Measurement aid (bench.py roofline leg): with a timer attached, plan_run issues plain launches
 * (never the captured graph) and brackets every kernel with two HIP events on `stream`.
 * kernel_id: 0 conv (materialized: the gather/matvec/scatter kernel; factored: K1, the edge-moment kernel of
 * csrc/moment.hip), 1 edge-MLP GEMM layer 1, 2 edge-MLP GEMM layer 2 (materialized only), 3 edge-MLP layer 0
 * (+ weight splits when not cached), 4 radius graph (+ the factored conv's degree order), 5 node prologue, 6 fc2,
 * 7 factored conv: K3 (slices + root + bias + mean + ReLU), 8 factored conv: K2 (projection on W3).  Read after
 * synchronising the stream. */
int mdno_rollout_plan_timer_attach(mdno_rollout_plan* plan, int max_records);
int mdno_rollout_plan_timer_read(mdno_rollout_plan* plan, int kernel_id, double* total_ms, int64_t* count);
int mdno_rollout_plan_timer_detach(mdno_rollout_plan* plan);

/* Which path ran (ABI 15).  gemm_mode MDNO_GEMM_SPLIT_F16 takes its two-fp16-plane products where the operands' ranges
 * allow it and redoes a piece on three bf16 planes where they do not (same result to fp32 rounding, more matrix work):
 * the decisions are made on the device, per piece, and counted there.  The counters are zeroed by every
 * mdno_rollout_plan_run and read here (the call synchronises `stream`, the stream the plan was run on):
 *   counts[0]  K1 workgroups of the factored conv (one destination x 256 hidden units of one application) that ran
 *              their stage loop again on bf16 planes (csrc/moment.hip: |H| staged by the workgroup left [2^-7, 2047),
 *              or the destination's neighbours have no feature in [2^-100, 2^100))
 *   counts[1]  destinations (per application) of the second kind: operands taken unscaled
 *   counts[2]  edge-MLP products (one GEMM over one chunk of <= 262,144 edges) redone on bf16 planes
 *              (csrc/edge_mlp_split.hip: an activation >= 65504, none >= 2^-10, or a non-finite weight)
 *   counts[3]  reserved (0)
 * All zero: every product of the run since the last plan_run was taken on fp16 planes.  Other GEMM modes: all zero. */
int mdno_rollout_plan_fallback_counts(mdno_rollout_plan* plan, int64_t counts[4], void* stream);
/* The same counters for ONE mdno_kernelnn_fwd call (which zeroes them when it starts): `workspace`, M, N, edge_cap and
 * the params as given to that call; position_graph != 0 if it was given edge_pos and dst and no edge_attr (the
 * workspace is laid out by the conv formulation that call resolved to).  Synchronises `stream`. */
int mdno_kernelnn_fallback_counts(const mdno_kernelnn_params* p, int M, int N, int64_t edge_cap, int position_graph,
                                  void* workspace, int64_t counts[4], void* stream);

/* ------------------------------------------------------------------------------------------
 * The optimiser step of train() (graph_kernel.py:467 on torch.optim.Adam(lr, weight_decay), :541-543) for all parameter
 * tensors in ONE launch (ABI 15): g' = g + weight_decay p;  m += (g' - m)(1 - beta1);  v = beta2 v + (1 - beta2) g'^2;
 * p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps) — torch.optim.Adam's arithmetic (L2 weight
 * decay, no amsgrad), fp32, in place.  `tensors`: HOST array of `count` entries (device pointers, element counts);
 * step >= 1 is the number of this update.  Asynchronous on `stream`.
 * ---------------------------------------------------------------------------------------- */
typedef struct mdno_adam_tensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t numel;
} mdno_adam_tensor;
int mdno_adam_step(int count, const mdno_adam_tensor* tensors, double lr, double beta1, double beta2, double eps,
                   double weight_decay, int64_t step, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training ops (BASELINE configs[3]) — the backward of the kernel-integral block that autograd +
 * torch_geometric provide to train() (graph_kernel.py:445-474) for the path :299-302 / :194-209 /
 * :239-242.  All fp32; row/edge reductions use fixed-order partial sums (bitwise reproducible).
 * With gz = g * (x_out > 0) and gs[t] = gz[t] / max(deg_t,1) (mean) or gz[t] (add):
 *   mdno_linear_fwd       C = act(A . W^T + b)       A [rows,k], W [n,k] (torch Linear layout)
 *   mdno_linear_split_fwd the same on the bf16 matrix pipe: both operands split exactly into three
 *                         bf16 planes on the way in, 6 plane products, fp32 accumulation (the
 *                         MDNO_GEMM_SPLIT_BF16 arithmetic; needs k % 32 == 0, n % 128 == 0 and a
 *                         workspace of mdno_linear_split_workspace_bytes(rows, n, k))
 *   mdno_linear_split_f16_fwd  the same on two fp16 planes per operand (3 plane products; the
 *                         MDNO_GEMM_SPLIT_F16 arithmetic): every row of A and of W is scaled by its own power of
 *                         two before the split and the output scaled back, so no magnitude leaves fp16's range;
 *                         same shape rules, workspace mdno_linear_split_f16_workspace_bytes(rows, n, k)
 *   mdno_gemm_atb         C (+)= A^T . B             A [rows,n1], B [rows,n2] -> C [n1,n2] (weight grads)
 *   mdno_gemm_atb_split_f16  the same product on the 16-bit matrix pipe at fp32 accuracy: every column of A and of B
 *                         scaled by its own power of two, split into two fp16 planes, three plane products, K slices
 *                         added in a fixed order (needs n1 % 256 == 0, n2 % 256 == 0: _supported; workspace
 *                         mdno_gemm_atb_split_f16_workspace_bytes(rows, n1, n2))
 *   mdno_colsum           out (+)= column sums of A [rows,n]                       (bias grads)
 *   mdno_relu_bwd         out = g * (y > 0) [* row_scale[row]]
 *   mdno_relu_bwd2        gz = g * (y > 0) and gs = gz * row_scale[row] in one pass (same values as two calls)
 *   mdno_transpose        At [cols,rows] = A [rows,cols]^T
 *   mdno_inv_degree       inv[r] = 1/max(deg_r,1) (mean) or 1 (add)
 *   mdno_nnconv_bwd_x     g_prev[r] = gz[r].root^T + sum_{e: src e = r} W_e . gs[dst e]; edges grouped by
 *                         source: row_ptr_s [R+1], eid_s [E] (position of the edge in the dst-sorted
 *                         arrays / in W_e), dst_s [E]   (mdno_coo_to_csr on the swapped edge list)
 *   mdno_nnconv_bwd_root  d_root (+)= sum_rows x^T gz, d_bias (+)= colsum(gz); x, gz [rows,64] (layers stacked)
 *   mdno_nnconv_bwd_we    d_we[p] (+)= sum_l x_l[src p] (x) gs_l[dst p]; x, gs [layers, R, 64] (layer_stride floats)
 * Workspaces: mdno_reduce_workspace_bytes(n1, n2) for gemm_atb / colsum (n2 = 1),
 * mdno_nnconv_bwd_root_workspace_bytes(rows).  mdno_nnconv_bwd_root_pair: conv1's and conv2's gradients from ONE launch over
 * x, gz [2 * rows_each, 64] (first half conv1's stacked layers, second half conv2's): bitwise the two single calls.
 * ---------------------------------------------------------------------------------------- */
int mdno_linear_fwd(const float* a, const float* w, const float* bias, int64_t rows, int n, int k, int relu,
                    float* c, void* stream);
size_t mdno_linear_split_workspace_bytes(int64_t rows, int n, int k);
int mdno_linear_split_fwd(const float* a, const float* w, const float* bias, int64_t rows, int n, int k, int relu,
                          float* c, void* workspace, size_t workspace_bytes, void* stream);
size_t mdno_linear_split_f16_workspace_bytes(int64_t rows, int n, int k);
int mdno_linear_split_f16_fwd(const float* a, const float* w, const float* bias, int64_t rows, int n, int k, int relu,
                              float* c, void* workspace, size_t workspace_bytes, void* stream);
size_t mdno_reduce_workspace_bytes(int n1, int n2);
int mdno_gemm_atb(const float* a, const float* b, int64_t rows, int n1, int n2, float* c, int accumulate,
                  void* workspace, size_t workspace_bytes, void* stream);
int mdno_gemm_atb_split_f16_supported(int64_t rows, int n1, int n2);
size_t mdno_gemm_atb_split_f16_workspace_bytes(int64_t rows, int n1, int n2);
int mdno_gemm_atb_split_f16(const float* a, const float* b, int64_t rows, int n1, int n2, float* c, int accumulate,
                            void* workspace, size_t workspace_bytes, void* stream);
int mdno_colsum(const float* a, int64_t rows, int n, float* out, int accumulate,
                void* workspace, size_t workspace_bytes, void* stream);
int mdno_relu_bwd2(const float* g, const float* y, const float* row_scale, int64_t rows, int n, float* gz, float* gs,
                   void* stream);
int mdno_relu_bwd(const float* g, const float* y, const float* row_scale, int64_t rows, int n, float* out,
                  void* stream);
int mdno_transpose(const float* a, int rows, int cols, float* at, void* stream);
int mdno_inv_degree(const int32_t* row_ptr, int rows, int aggr, float* inv, void* stream);
int mdno_nnconv_bwd_x(const float* gz, const float* gs, const int32_t* row_ptr_s, const int32_t* eid_s,
                      const int32_t* dst_s, int num_rows, const float* w_e, const float* root,
                      int Cin, int Cout, float* g_prev, void* stream);
size_t mdno_nnconv_bwd_root_workspace_bytes(int64_t rows);
size_t mdno_nnconv_bwd_root_pair_workspace_bytes(int64_t rows_each);
int mdno_nnconv_bwd_root_pair(const float* x, const float* gz, int64_t rows_each, float* d_root1, float* d_bias1,
                              float* d_root2, float* d_bias2, void* workspace, size_t workspace_bytes, void* stream);
int mdno_nnconv_bwd_root(const float* x, const float* gz, int64_t rows, int Cin, int Cout,
                         float* d_root, float* d_bias, int accumulate,
                         void* workspace, size_t workspace_bytes, void* stream);
int mdno_nnconv_bwd_we(const float* x, const float* gs, const int32_t* src, const int32_t* dst, int64_t E,
                       int layers, int64_t layer_stride, int Cin, int Cout, float* d_we, int accumulate,
                       void* stream);

/* ------------------------------------------------------------------------------------------
 * Training ops, bf16 (BASELINE configs[3] names bf16; csrc/train_bf16.hip).  The block's large tensors —
 * h1, h2 [E,k], W_e and dW_e [E,64*64] — are bf16, row-major (pointers typed void*), every GEMM is one
 * bf16 x bf16 MFMA product with fp32 accumulation; parameters (cast per call from the fp32 masters), node
 * features, conv outputs and all reductions are fp32.  Same formulas as the fp32 ops above, width 64.
 *   mdno_cast_bf16          out[i] = bf16(in[i])  (round to nearest even), count % 4 == 0
 *   mdno_linear_smallk_bf16_fwd  c bf16 [rows,n] = act(a . w^T + b) for the FIRST edge-MLP layer (graph_kernel.py:239-242,
 *                           layers.0): a fp32 [rows,k] edge attributes, k <= 8, n % 8 == 0; the same fmaf chains
 *                           as mdno_linear_fwd's generic kernel, rounded once to bf16
 *   mdno_linear_bf16_fwd    c = act(a . w^T + b): a bf16 [rows,k], w fp32 [n,k]; c bf16 (out_bf16) or fp32;
 *                           n % 128 == 0, k % 32 == 0; workspace mdno_linear_bf16_workspace_bytes(n, k)
 *   mdno_linear_bf16_masked c bf16 [rows,n] = (y > 0) ? a . w^T : 0 — the input gradient of a Linear+ReLU layer whose
 *                           stored output is y bf16 [rows,n] (graph_kernel.py:239-242 differentiated), mask fused
 *                           into the GEMM's epilogue; n % 256 == 0, k % 32 == 0, k >= 64
 *                           (mdno_linear_bf16_masked_supported); workspace mdno_linear_bf16_workspace_bytes(n, k)
 *   mdno_gemm_atb_bf16      c [n1,n2] fp32 = a^T . b over rows, a bf16 [rows,n1], b bf16 [rows,n2], n1, n2 % 128 == 0;
 *                           fixed row slices added in order; workspace mdno_gemm_atb_bf16_workspace_bytes (covers any
 *                           rows < 2^31 * 32 / (2 * max(n1,n2)) — 8.4M at n = 4096; beyond: MDNO_EWORKSPACE)
 *   mdno_nnconv_bf16w_fwd   mdno_nnconv_fwd at 64x64 with w_e bf16 [E,4096]
 *   mdno_nnconv_bwd_x_bf16w mdno_nnconv_bwd_x with w_e bf16
 *   mdno_nnconv_bwd_we_bf16 d_we bf16 [E,4096] = sum_l x_l[src p] (x) gs_l[dst p] (rounded once, at the end)
 *   mdno_nnconv_bwd_we_bf16_colsum  the same d_we (layers <= 16: one MFMA k-step per 32 x 32 quadrant, both fp32 operands as three
 *                           bf16 planes, six plane products) AND colsum [4096] fp32 = its column sums, the sums of the ROUNDED
 *                           values as mdno_colsum_bf16 would take them from the stored tensor, without the second pass over it;
 *                           workspace mdno_nnconv_bwd_we_bf16_colsum_workspace_bytes()
 *   mdno_nnconv_bwd_we_colsum       the same with d_we and its column sums in fp32 (the fp32 training path: mdno_nnconv_bwd_we + mdno_colsum)
 *   mdno_relu_bwd_bf16      out = g * (y > 0): g fp32, y bf16, out bf16 (out_bf16) or fp32; n % 4 == 0
 *   mdno_colsum_bf16        out [n] fp32 = column sums of a bf16 [rows,n]; workspace mdno_colsum_bf16_workspace_bytes(n)
 * ---------------------------------------------------------------------------------------- */
int mdno_cast_bf16(const float* in, int64_t count, void* out, void* stream);
int mdno_linear_smallk_bf16_fwd(const float* a, const float* w, const float* bias, int64_t rows, int n, int k,
                                int relu, void* c, void* stream);
size_t mdno_linear_bf16_workspace_bytes(int n, int k);
int mdno_linear_bf16_fwd(const void* a, const float* w, const float* bias, int64_t rows, int n, int k, int relu,
                         int out_bf16, void* c, void* workspace, size_t workspace_bytes, void* stream);
int mdno_linear_bf16_masked_supported(int64_t rows, int n, int k);
int mdno_linear_bf16_masked(const void* a, const float* w, const void* y, int64_t rows, int n, int k, void* c,
                            void* workspace, size_t workspace_bytes, void* stream);
size_t mdno_gemm_atb_bf16_workspace_bytes(int n1, int n2);
int mdno_gemm_atb_bf16(const void* a, const void* b, int64_t rows, int n1, int n2, float* c,
                       void* workspace, size_t workspace_bytes, void* stream);
int mdno_nnconv_bf16w_fwd(const float* x, const int32_t* row_ptr, const int32_t* src, int num_rows, const void* w_e,
                          const float* root, const float* bias, int aggr, int relu, float* y, void* stream);
int mdno_nnconv_bwd_x_bf16w(const float* gz, const float* gs, const int32_t* row_ptr_s, const int32_t* eid_s,
                            const int32_t* dst_s, int num_rows, const void* w_e, const float* root, float* g_prev,
                            void* stream);
int mdno_nnconv_bwd_we_bf16(const float* x, const float* gs, const int32_t* src, const int32_t* dst, int64_t E,
                            int layers, int64_t layer_stride, void* d_we, void* stream);
size_t mdno_nnconv_bwd_we_colsum_workspace_bytes(void);
int mdno_nnconv_bwd_we_colsum(const float* x, const float* gs, const int32_t* src, const int32_t* dst, int64_t E, int layers,
                              int64_t layer_stride, float* d_we, float* colsum, void* workspace, size_t workspace_bytes,
                              void* stream);
size_t mdno_nnconv_bwd_we_bf16_colsum_workspace_bytes(void);
int mdno_nnconv_bwd_we_bf16_colsum(const float* x, const float* gs, const int32_t* src, const int32_t* dst, int64_t E,
                                   int layers, int64_t layer_stride, void* d_we, float* colsum, void* workspace,
                                   size_t workspace_bytes, void* stream);
int mdno_relu_bwd_bf16(const float* g, const void* y, int64_t rows, int n, int out_bf16, void* out, void* stream);
size_t mdno_colsum_bf16_workspace_bytes(int n);
int mdno_colsum_bf16(const void* a, int64_t rows, int n, float* out, void* workspace, size_t workspace_bytes,
                     void* stream);

/* ------------------------------------------------------------------------------------------
 * Training: the 2*depth conv applications of a step as ONE call each way (graph_kernel.py:299-302 and what autograd
 * does for them in train(), :445-474) — the same kernels in the same order as 2*depth mdno_nnconv_*_fwd calls and
 * 2*depth (mdno_relu_bwd2, mdno_nnconv_bwd_x*) pairs, with every ReLU backward below the top one done by the
 * input-gradient kernel above it (same arithmetic; 2*depth - 1 launches fewer).  Width 64, mean aggregation.
 *   x_layers f32 [2*depth+1, R, 64]: [0] = the block's input; forward writes [a] = relu(conv(x[a-1])), conv1's
 *            root/bias for a <= depth, conv2's above.  w_e f32 / bf16 [E,4096] in destination-sorted edge order.
 *   backward: g_out [R,64] = dLoss/dx[2*depth]; inv_deg [R] (mdno_inv_degree); edges grouped by source as for
 *            mdno_nnconv_bwd_x.  Writes gz, gs f32 [2*depth, R, 64] (gz[a-1] = g_a * (x[a] > 0), gs = gz * inv_deg:
 *            the operands of mdno_nnconv_bwd_root / mdno_nnconv_bwd_we*) and g_in [R,64] = dLoss/dx[0].
 *   mdno_colsum_atb_bf16  colsum [n] = column sums of a bf16 [rows,n] and atb [n,kb] = a^T . b for b fp32 [rows,kb]
 *            (kb = 6 or 8) in one pass: bias and weight gradient of the edge-MLP's first layer (b = edge_attr);
 *            n % 8 == 0; workspace mdno_colsum_atb_bf16_workspace_bytes(n, kb).
 * ---------------------------------------------------------------------------------------- */
int mdno_nnconv_chain_fwd(float* x_layers, const int32_t* row_ptr, const int32_t* src, int num_rows, const float* w_e,
                          const float* root1, const float* bias1, const float* root2, const float* bias2, int depth,
                          void* stream);
int mdno_nnconv_chain_bwd(const float* g_out, const float* x_layers, const float* inv_deg, const int32_t* row_ptr_s,
                          const int32_t* eid_s, const int32_t* dst_s, int num_rows, const float* w_e, const float* root1,
                          const float* root2, int depth, float* gz, float* gs, float* g_in, void* stream);
int mdno_nnconv_chain_bf16w_fwd(float* x_layers, const int32_t* row_ptr, const int32_t* src, int num_rows,
                                const void* w_e, const float* root1, const float* bias1, const float* root2,
                                const float* bias2, int depth, void* stream);
int mdno_nnconv_chain_bf16w_bwd(const float* g_out, const float* x_layers, const float* inv_deg,
                                const int32_t* row_ptr_s, const int32_t* eid_s, const int32_t* dst_s, int num_rows,
                                const void* w_e, const float* root1, const float* root2, int depth, float* gz, float* gs,
                                float* g_in, void* stream);
size_t mdno_colsum_atb_bf16_workspace_bytes(int n, int kb);
int mdno_colsum_atb_bf16(const void* a, const float* b, int64_t rows, int n, int kb, float* colsum, float* atb,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training: batch assembly on the device (csrc/collate.hip) — replaces, for a trajectory resident in HBM,
 * ContactMapDataset.__getitem__ per sample (dataset.py:180-227, incl. the per-edge attribute loop :194-201)
 * and torch_geometric's DataListLoader / Batch.from_data_list collation (graph_kernel.py:513-519, :454;
 * offset rule PairData.__inc__, dataset.py:41-45).
 *   pos   f32 [T,N,3]; rows, cols i32: the flat contact maps of all frames (dataset.py:114, 189)
 *   meta  i64 [3*B+1] (device): {first window frame of sample b} {first edge of that frame in rows/cols}
 *         {first edge of sample b in the batch; entry B = E}
 *   ->    x_position f32 [W,B*N,3] (time-major), y f32 [B*N,3] (frame idx+W+horizon-1),
 *         edge_index i64 [2,E] (sample b shifted by b*N), edge_attr f32 [E,6] = [pos[idx][row], pos[idx][col]]
 *   max_edges_per_sample only sizes the launch.
 * ---------------------------------------------------------------------------------------- */
int mdno_collate_samples(const float* pos, const int32_t* rows, const int32_t* cols, const int64_t* meta,
                         int B, int N, int W, int horizon, int max_edges_per_sample, float* x_position, float* y,
                         int64_t* edge_index, float* edge_attr, void* stream);
/* out[p][0..width) = in[perm[p]][0..width): per-edge rows (edge attributes) put into the destination-sorted order of a
 * graph built by mdno_coo_to_csr (its perm) — torch's index_select on the path edge_attr -> DenseNet (graph_kernel.py:200). */
int mdno_permute_rows(const float* in, const int32_t* perm, int64_t rows, int width, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training: the loss (csrc/loss.hip) — LpLoss.rel with p = 2 (graph_kernel.py:105-119; used :462, :547) and the batch
 * MSE train() logs beside it (:465), forward and backward in three small launches instead of ~15 ATen ones.
 *   out, y    f32 [batch, dim] (dim = atoms * 3: the flattened frame of a sample)
 *   stats     f32 [batch, 4] (16-B aligned), written by fwd, read by bwd: {ratio_b, ||out_b - y_b||^2, ||out_b - y_b||, ||y_b||}
 *   loss_mse  f32 [2]: loss = sum_b ratio_b (size_average != 0: / batch), mse = sum_b ||out_b - y_b||^2 / (batch dim)
 *   bwd       grad_out [batch, dim] = g * (out_b - y_b) / (||out_b - y_b|| ||y_b||) (/ batch if size_average), 0 for a
 *             sample with out_b == y_b (torch.norm's subgradient); grad_loss f32 [1] on the device, NULL = 1.
 * Fixed summation orders (bitwise reproducible).
 * ---------------------------------------------------------------------------------------- */
int mdno_lploss_rel_fwd(const float* out, const float* y, long long batch, int dim, int size_average, float* stats,
                        float* loss_mse, void* stream);
int mdno_lploss_rel_bwd(const float* out, const float* y, const float* stats, const float* grad_loss, long long batch,
                        int dim, int size_average, float* grad_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training: backward of the per-atom ends (csrc/train_nodes.hip) — the node prologue (graph_kernel.py:279-298;
 * forward = mdno_node_prologue_fwd) and fc2 (:305; forward = mdno_fc_out_fwd).  Fixed-order reductions.
 *   mdno_node_prologue_bwd   x0, g0 f32 [M*N,width]: the forward's output and dLoss/dx0.  Outputs (overwritten):
 *                            d_lstm f32 [108] = [w_ih 36 | w_hh 36 | b_ih 12 | lstm_fc.weight 9 | lstm_fc.bias 3 |
 *                            b_hh 12 (the same values as b_ih, a second copy: every parameter's gradient has its
 *                            own 12 floats)] (NULL for the notebook-era model), d_emb
 *                            [num_embeddings, embedding_dim], d_fc1_w [width, in_width], d_fc1_b [width].
 *                            window <= 16.
 *   mdno_fc_out_bwd          out = x . w^T + b:  dx [rows,width] = g . w,  d_w [out_width,width] = g^T . x,
 *                            d_b [out_width] = colsum(g);  g f32 [rows,out_width]
 * ---------------------------------------------------------------------------------------- */
size_t mdno_node_prologue_bwd_workspace_bytes(const mdno_kernelnn_params* p, int rows);
int mdno_node_prologue_bwd(const mdno_kernelnn_params* p, const float* frames, int M, int W, int N,
                           const int64_t* x_aminoacid, int aa_per_member, const float* x0, const float* g0,
                           float* d_lstm, float* d_emb, float* d_fc1_w, float* d_fc1_b,
                           void* workspace, size_t workspace_bytes, void* stream);
size_t mdno_fc_out_bwd_workspace_bytes(int rows, int width, int out_width);
int mdno_fc_out_bwd(const float* x, const float* w, const float* g, int rows, int width, int out_width,
                    float* dx, float* d_w, float* d_b, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MDNO_H */
