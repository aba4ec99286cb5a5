// Training: backward of the per-atom ends of KernelNN.forward — the node prologue (graph_kernel.py:279-298:
// W LSTM(3,3) cells over the window with the atoms as the batch, lstm_fc, Embedding, concat, fc1, ReLU;
// forward = node_prologue_kernel in node_ops.hip) and the output projection fc2 (:305).  With these the
// whole differentiable forward + backward of the model runs in libmdno; PyTorch keeps the parameters and
// the optimizer.  Replaces what autograd does for those modules in train() (graph_kernel.py:445-474).
//
// One thread per atom walks its own LSTM backward through time: the forward is replayed once keeping
// (h_t, c_t) of every window step in registers (6 floats x W <= 16), each backward step rebuilds its gates
// from x_t and h_{t-1}.  Parameter gradients are summed over atoms in a FIXED order — butterfly inside a
// wave, waves in order through LDS, workgroups in order by a second kernel — so they are bitwise
// reproducible (no float atomics); the embedding gradient, a scatter by residue type, is gathered per
// table entry in atom order.
#include "kernels.h"

namespace mdno {
namespace {

constexpr int H = 3, MAX_W = 16, MAX_EMB = 16, ROWS = 256;   // atoms per workgroup
constexpr int N_LSTM = 4 * H * H * 2 + 4 * H + H * H + H;     // w_ih 36 | w_hh 36 | bias 12 (b_ih and b_hh share it) | fc_w 9 | fc_b 3 = 96

__device__ __forceinline__ float sigm(float v) { return 1.0f / (1.0f + expf(-v)); }

// sum over the workgroup in a fixed order; the result is valid in thread 0
__device__ __forceinline__ float block_sum(float v, float* red /* [4] */) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

struct PrologueBwdArgs {
    const float* frames;      // [W, R, 3]
    int R, N, W;
    const long long* aa;
    int aa_per_member;
    const float *w_ih, *w_hh, *b_ih, *b_hh, *fc_w, *fc_b, *emb_w, *fc1_w, *fc1_b;   // lstm pointers NULL: notebook-era model
    int num_emb, emb_dim, width;
    const float* x0;          // [R, width] forward output (ReLU mask)
    const float* g0;          // [R, width] dLoss/dx0
    float* part;              // [blocks][stride] partial sums: lstm 96 | emb num_emb*emb_dim | fc1_w width*in_w | fc1_b width
    int stride;
};

__global__ __launch_bounds__(ROWS) void node_prologue_bwd_kernel(PrologueBwdArgs a) {
    __shared__ float feat_s[ROWS][MAX_EMB + H];        // the fc1 input of every atom of this workgroup
    __shared__ float dfeat_s[ROWS][MAX_EMB];           // its gradient wrt the embedding part
    __shared__ int aa_s[ROWS];
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int r = blockIdx.x * ROWS + tid;
    const bool live = r < a.R;
    const int in_w = a.emb_dim + H;
    const bool lstm = a.w_ih != nullptr;
    float* P = a.part + (size_t)blockIdx.x * a.stride;

    // ---- replay the forward of this atom
    float hs[MAX_W + 1][H], cs[MAX_W + 1][H];
#pragma unroll
    for (int k = 0; k < H; ++k) { hs[0][k] = 0.f; cs[0][k] = 0.f; }
    float wih[4 * H][H], whh[4 * H][H], bsum[4 * H];
    if (lstm) {
#pragma unroll
        for (int g = 0; g < 4 * H; ++g) {
#pragma unroll
            for (int k = 0; k < H; ++k) { wih[g][k] = a.w_ih[g * H + k]; whh[g][k] = a.w_hh[g * H + k]; }
            bsum[g] = a.b_ih[g] + a.b_hh[g];
        }
    }
    const float* f0 = a.frames + (size_t)(live ? r : 0) * 3;
    auto gates = [&](int t, float (&gi)[H], float (&gf)[H], float (&gg)[H], float (&go)[H]) {
        const float* p = f0 + (size_t)t * a.R * 3;
        const float x[H] = {p[0], p[1], p[2]};
        float pre[4 * H];
#pragma unroll
        for (int g = 0; g < 4 * H; ++g) {
            float s = bsum[g];
#pragma unroll
            for (int k = 0; k < H; ++k) s = fmaf(wih[g][k], x[k], s);
#pragma unroll
            for (int k = 0; k < H; ++k) s = fmaf(whh[g][k], hs[t][k], s);
            pre[g] = s;
        }
#pragma unroll
        for (int k = 0; k < H; ++k) {
            gi[k] = sigm(pre[k]); gf[k] = sigm(pre[H + k]); gg[k] = tanhf(pre[2 * H + k]); go[k] = sigm(pre[3 * H + k]);
        }
    };
    if (lstm) {
#pragma unroll
        for (int t = 0; t < MAX_W; ++t) {
            if (t < a.W) {
                float gi[H], gf[H], gg[H], go[H];
                gates(t, gi, gf, gg, go);
#pragma unroll
                for (int k = 0; k < H; ++k) {
                    cs[t + 1][k] = gf[k] * cs[t][k] + gi[k] * gg[k];
                    hs[t + 1][k] = go[k] * tanhf(cs[t + 1][k]);
                }
            }
        }
    }
    float hW[H] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t <= MAX_W; ++t)
        if (t == a.W) {
#pragma unroll
            for (int k = 0; k < H; ++k) hW[k] = hs[t][k];
        }
    long long id = live ? a.aa[a.aa_per_member ? r : r % a.N] : 0;
    id = id < 0 ? 0 : (id >= a.num_emb ? a.num_emb - 1 : id);          // (the forward flagged it)
    float feat[MAX_EMB + H];
#pragma unroll
    for (int e = 0; e < MAX_EMB; ++e) feat[e] = (e < a.emb_dim) ? a.emb_w[id * a.emb_dim + e] : 0.f;
    if (lstm) {
#pragma unroll
        for (int k = 0; k < H; ++k) {
            float s = a.fc_b[k];
#pragma unroll
            for (int j = 0; j < H; ++j) s = fmaf(a.fc_w[k * H + j], hW[j], s);
            feat[MAX_EMB + k] = s;
        }
    } else {
        const float* p = f0 + (size_t)(a.W - 1) * a.R * 3;
#pragma unroll
        for (int k = 0; k < H; ++k) feat[MAX_EMB + k] = p[k];
    }
    // ---- d feat = fc1_w^T . (g0 * (x0 > 0))
    float dfeat[MAX_EMB + H];
#pragma unroll
    for (int i = 0; i < MAX_EMB + H; ++i) dfeat[i] = 0.f;
    if (live) {
        const float* gr = a.g0 + (size_t)r * a.width;
        const float* xr = a.x0 + (size_t)r * a.width;
        for (int o = 0; o < a.width; ++o) {
            const float gz = xr[o] > 0.f ? gr[o] : 0.f;
            const float* w = a.fc1_w + (size_t)o * in_w;
            for (int e = 0; e < a.emb_dim; ++e) dfeat[e] = fmaf(gz, w[e], dfeat[e]);
#pragma unroll
            for (int k = 0; k < H; ++k) dfeat[MAX_EMB + k] = fmaf(gz, w[a.emb_dim + k], dfeat[MAX_EMB + k]);
        }
    }
#pragma unroll
    for (int i = 0; i < MAX_EMB + H; ++i) feat_s[tid][i] = live ? feat[i] : 0.f;
#pragma unroll
    for (int e = 0; e < MAX_EMB; ++e) dfeat_s[tid][e] = live ? dfeat[e] : 0.f;
    aa_s[tid] = live ? (int)id : -1;

    // ---- LSTM + lstm_fc backward for this atom; per-thread parameter gradients
    float glstm[N_LSTM];
#pragma unroll
    for (int i = 0; i < N_LSTM; ++i) glstm[i] = 0.f;
    if (lstm && live) {
        float dh[H] = {0.f, 0.f, 0.f}, dc[H] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < H; ++k) {        // feat[4+k] = fc_b[k] + sum_j fc_w[k][j] h_W[j]
            const float d = dfeat[MAX_EMB + k];
            glstm[93 + k] = d;
#pragma unroll
            for (int j = 0; j < H; ++j) {
                glstm[84 + k * H + j] = d * hW[j];
                dh[j] = fmaf(a.fc_w[k * H + j], d, dh[j]);
            }
        }
#pragma unroll
        for (int t = MAX_W - 1; t >= 0; --t) {
            if (t < a.W) {
                float gi[H], gf[H], gg[H], go[H];
                gates(t, gi, gf, gg, go);
                const float* p = f0 + (size_t)t * a.R * 3;
                const float x[H] = {p[0], p[1], p[2]};
                float dpre[4 * H];
#pragma unroll
                for (int k = 0; k < H; ++k) {
                    const float tc = tanhf(cs[t + 1][k]);
                    const float d_o = dh[k] * tc;
                    const float dct = dc[k] + dh[k] * go[k] * (1.f - tc * tc);
                    dpre[k] = dct * gg[k] * gi[k] * (1.f - gi[k]);                 // input gate
                    dpre[H + k] = dct * cs[t][k] * gf[k] * (1.f - gf[k]);          // forget gate
                    dpre[2 * H + k] = dct * gi[k] * (1.f - gg[k] * gg[k]);         // candidate
                    dpre[3 * H + k] = d_o * go[k] * (1.f - go[k]);                 // output gate
                    dc[k] = dct * gf[k];
                }
#pragma unroll
                for (int k = 0; k < H; ++k) dh[k] = 0.f;
#pragma unroll
                for (int g = 0; g < 4 * H; ++g) {
#pragma unroll
                    for (int k = 0; k < H; ++k) {
                        glstm[g * H + k] = fmaf(dpre[g], x[k], glstm[g * H + k]);                  // w_ih
                        glstm[36 + g * H + k] = fmaf(dpre[g], hs[t][k], glstm[36 + g * H + k]);    // w_hh
                        dh[k] = fmaf(whh[g][k], dpre[g], dh[k]);
                    }
                    glstm[72 + g] += dpre[g];                                                      // b_ih = b_hh
                }
            }
        }
    }
    // ---- reduce over the workgroup, fixed order
    if (lstm) {
#pragma unroll
        for (int i = 0; i < N_LSTM; ++i) {
            const float s = block_sum(glstm[i], red);
            if (tid == 0) P[i] = s;
        }
    }
    __syncthreads();
    // embedding: entry (row, col) adds the atoms of its residue type in atom order
    const int n_emb = a.num_emb * a.emb_dim;
    for (int j = tid; j < n_emb; j += ROWS) {
        const int row = j / a.emb_dim, col = j - row * a.emb_dim;
        float s = 0.f;
        for (int q = 0; q < ROWS; ++q)
            if (aa_s[q] == row) s += dfeat_s[q][col];
        P[N_LSTM + j] = s;
    }
    // fc1: thread (o, quarter) sums gz[r][o] * feat[r][:] over its 64 atoms; quarters added in order
    float* Pw = P + N_LSTM + n_emb;
    float* Pb = Pw + (size_t)a.width * in_w;
    for (int o0 = 0; o0 < a.width; o0 += 64) {
        const int o = o0 + (tid & 63), part = tid >> 6;
        float acc[MAX_EMB + H + 1];
#pragma unroll
        for (int i = 0; i <= MAX_EMB + H; ++i) acc[i] = 0.f;
        if (o < a.width) {
            for (int q = part * 64; q < part * 64 + 64; ++q) {
                const int rr = blockIdx.x * ROWS + q;
                if (rr >= a.R) break;
                const float gz = a.x0[(size_t)rr * a.width + o] > 0.f ? a.g0[(size_t)rr * a.width + o] : 0.f;
#pragma unroll
                for (int i = 0; i < MAX_EMB + H; ++i) acc[i] = fmaf(gz, feat_s[q][i], acc[i]);
                acc[MAX_EMB + H] += gz;
            }
        }
        // 4-way combine through LDS, one value per (quarter, o), quarters in order
        __shared__ float comb[4][64];
        for (int i = 0; i <= MAX_EMB + H; ++i) {
            const bool used = i < a.emb_dim || (i >= MAX_EMB && i < MAX_EMB + H) || i == MAX_EMB + H;
            if (!used) continue;                       // uniform
            __syncthreads();
            comb[part][tid & 63] = acc[i];
            __syncthreads();
            if (part == 0 && o < a.width) {
                const float s = (comb[0][tid] + comb[1][tid]) + (comb[2][tid] + comb[3][tid]);
                if (i == MAX_EMB + H) Pb[o] = s;
                else Pw[(size_t)o * in_w + (i < MAX_EMB ? i : a.emb_dim + (i - MAX_EMB))] = s;
            }
        }
    }
}

// out[j] = sum over workgroups (in order) of part[b][j]
__global__ __launch_bounds__(256) void reduce_blocks_kernel(const float* __restrict__ part, int blocks, int stride,
                                                            int count, float* __restrict__ out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= count) return;
    float s = 0.f;
    for (int b = 0; b < blocks; ++b) s += part[(size_t)b * stride + j];
    out[j] = s;
}

// ---------------------------------------------------------------- fc2 backward
// out = x . W^T + b  (x [R,width], W [ow,width]):  dx = g . W,  dW = g^T . x,  db = colsum(g)
__global__ __launch_bounds__(256) void fc_out_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ g, int R, int width, int ow,
                                                         float* __restrict__ dx, float* __restrict__ part, int stride) {
    __shared__ float comb[4][64];
    const int tid = threadIdx.x, c0 = tid & 63, part_id = tid >> 6;
    const int rbase = blockIdx.x * ROWS;
    float* P = part + (size_t)blockIdx.x * stride;
    // dx: thread (row quarter, column) walks its 64 rows
    for (int cb = 0; cb < width; cb += 64) {
        const int c = cb + c0;
        for (int q = part_id * 64; q < part_id * 64 + 64; ++q) {
            const int r = rbase + q;
            if (r >= R || c >= width) break;
            float s = 0.f;
            for (int o = 0; o < ow; ++o) s = fmaf(g[(size_t)r * ow + o], w[(size_t)o * width + c], s);
            dx[(size_t)r * width + c] = s;
        }
    }
    // dW[o][c], db[o]: quarters of the workgroup's rows, combined in order
    for (int o = 0; o < ow; ++o) {
        for (int cb = 0; cb < width; cb += 64) {
            const int c = cb + c0;
            float s = 0.f;
            if (c < width)
                for (int q = part_id * 64; q < part_id * 64 + 64; ++q) {
                    const int r = rbase + q;
                    if (r >= R) break;
                    s = fmaf(g[(size_t)r * ow + o], x[(size_t)r * width + c], s);
                }
            __syncthreads();
            comb[part_id][c0] = s;
            __syncthreads();
            if (part_id == 0 && c < width) P[o * width + c] = (comb[0][c0] + comb[1][c0]) + (comb[2][c0] + comb[3][c0]);
        }
        float sb = 0.f;
        const int r = rbase + tid;
        if (r < R) sb = g[(size_t)r * ow + o];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sb += __shfl_xor(sb, off);
        __syncthreads();
        if (c0 == 0) comb[0][part_id] = sb;
        __syncthreads();
        if (tid == 0) P[ow * width + o] = (comb[0][0] + comb[0][1]) + (comb[0][2] + comb[0][3]);
    }
}

}  // namespace
}  // namespace mdno

using namespace mdno;

static int prologue_counts(const mdno_kernelnn_params* p, int* n_emb, int* n_w, int* total) {
    *n_emb = p->num_embeddings * p->embedding_dim;
    *n_w = p->width * (p->embedding_dim + H);
    *total = N_LSTM + *n_emb + *n_w + p->width;
    return 0;
}

extern "C" size_t mdno_node_prologue_bwd_workspace_bytes(const mdno_kernelnn_params* p, int rows) {
    if (!p || rows <= 0) return 0;
    int ne, nw, tot;
    prologue_counts(p, &ne, &nw, &tot);
    return align_up((size_t)((rows + ROWS - 1) / ROWS) * tot * sizeof(float), 256);
}

extern "C" int mdno_node_prologue_bwd(const mdno_kernelnn_params* p, const float* frames, int M, int W, int N,
                                      const int64_t* x_aminoacid, int aa_per_member, const float* x0, const float* g0,
                                      float* d_lstm, float* d_emb, float* d_fc1_w, float* d_fc1_b, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    MDNO_REQUIRE(p && frames && x_aminoacid && x0 && g0 && d_emb && d_fc1_w && d_fc1_b && workspace, MDNO_EINVAL,
                 "mdno_node_prologue_bwd: null pointer");
    const bool lstm = p->lstm_w_ih != nullptr;
    MDNO_REQUIRE(!lstm || (p->lstm_w_hh && p->lstm_b_ih && p->lstm_b_hh && p->lstm_fc_w && p->lstm_fc_b && d_lstm),
                 MDNO_EINVAL, "mdno_node_prologue_bwd: partial LSTM set");
    MDNO_REQUIRE(M > 0 && N > 0 && W > 0 && W <= MAX_W, MDNO_EUNSUPPORTED, "mdno_node_prologue_bwd: window %d (1..%d)", W,
                 MAX_W);
    MDNO_REQUIRE(p->x_position_dim == H && p->embedding_dim >= 0 && p->embedding_dim <= MAX_EMB &&
                     p->in_width == p->embedding_dim + H,
                 MDNO_EUNSUPPORTED, "mdno_node_prologue_bwd: unsupported dims");
    const int R = M * N, blocks = (R + ROWS - 1) / ROWS;
    int ne, nw, tot;
    prologue_counts(p, &ne, &nw, &tot);
    MDNO_REQUIRE(workspace_bytes >= mdno_node_prologue_bwd_workspace_bytes(p, R), MDNO_EWORKSPACE,
                 "mdno_node_prologue_bwd: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(workspace);
    PrologueBwdArgs a{frames, R, N, W, (const long long*)x_aminoacid, aa_per_member, p->lstm_w_ih, p->lstm_w_hh,
                      p->lstm_b_ih, p->lstm_b_hh, p->lstm_fc_w, p->lstm_fc_b, p->emb_w, p->fc1_w, p->fc1_b,
                      p->num_embeddings, p->embedding_dim, p->width, x0, g0, part, tot};
    hipLaunchKernelGGL(node_prologue_bwd_kernel, dim3(blocks), dim3(ROWS), 0, s, a);
    auto reduce = [&](int off, int count, float* out) {
        hipLaunchKernelGGL(reduce_blocks_kernel, dim3((count + 255) / 256), dim3(256), 0, s, (const float*)part + off,
                           blocks, tot, count, out);
    };
    if (lstm) reduce(0, N_LSTM, d_lstm);
    reduce(N_LSTM, ne, d_emb);
    reduce(N_LSTM + ne, nw, d_fc1_w);
    reduce(N_LSTM + ne + nw, p->width, d_fc1_b);
    return check_launch("node_prologue_bwd");
}

extern "C" size_t mdno_fc_out_bwd_workspace_bytes(int rows, int width, int out_width) {
    if (rows <= 0) return 0;
    return align_up((size_t)((rows + ROWS - 1) / ROWS) * (size_t)(out_width * width + out_width) * sizeof(float), 256);
}

extern "C" int mdno_fc_out_bwd(const float* x, const float* w, const float* g, int rows, int width, int out_width,
                               float* dx, float* d_w, float* d_b, void* workspace, size_t workspace_bytes,
                               void* stream) {
    MDNO_REQUIRE(x && w && g && dx && d_w && d_b && workspace && rows > 0 && width > 0 && out_width > 0, MDNO_EINVAL,
                 "mdno_fc_out_bwd: bad arguments");
    MDNO_REQUIRE(workspace_bytes >= mdno_fc_out_bwd_workspace_bytes(rows, width, out_width), MDNO_EWORKSPACE,
                 "mdno_fc_out_bwd: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int blocks = (rows + ROWS - 1) / ROWS, stride = out_width * width + out_width;
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(fc_out_bwd_kernel, dim3(blocks), dim3(ROWS), 0, s, x, w, g, rows, width, out_width, dx, part, stride);
    hipLaunchKernelGGL(reduce_blocks_kernel, dim3((out_width * width + 255) / 256), dim3(256), 0, s, (const float*)part,
                       blocks, stride, out_width * width, d_w);
    hipLaunchKernelGGL(reduce_blocks_kernel, dim3((out_width + 255) / 256), dim3(256), 0, s,
                       (const float*)part + out_width * width, blocks, stride, out_width, d_b);
    return check_launch("fc_out_bwd");
}
