// Training: backward of the per-atom ends of KernelNN.forward — the node prologue (graph_kernel.py:279-298:
// W LSTM(3,3) cells over the window with the atoms as the batch, lstm_fc, Embedding, concat, fc1, ReLU;
// forward = node_prologue_kernel in node_ops.hip) and the output projection fc2 (:305).  With these the
// whole differentiable forward + backward of the model runs in libmdno; PyTorch keeps the parameters and
// the optimizer.  Replaces what autograd does for those modules in train() (graph_kernel.py:445-474).
//
// One thread per atom walks its own LSTM backward through time: the forward is replayed once keeping
// (h_t, c_t) of every window step in registers (6 floats x W <= 16), each backward step rebuilds its gates
// from x_t and h_{t-1}.  A workgroup is ONE wave = 64 atoms (round 4; it was 256 atoms = 14 workgroups for a
// cfg4 batch, each thread walking row-strided global loads one round trip at a time: 154 us): the wave's
// 64 x 64 tile of g0 * (x0 > 0) is brought to LDS by whole-row loads and every later loop reads LDS.
// Parameter gradients are summed over atoms in a FIXED order — butterfly inside the wave, workgroups in
// order by a second kernel — so they are bitwise reproducible (no float atomics); the embedding gradient, a
// scatter by residue type, is gathered per table entry in atom order.
#include "kernels.h"

namespace mdno {
namespace {

constexpr int H = 3, MAX_W = 16, MAX_EMB = 16, ROWS = 64;   // atoms (rows) per workgroup
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

template <int EMB>      // compile-time bound on emb_dim (a multiple of 4)
__global__ __launch_bounds__(ROWS) void node_prologue_bwd_kernel(PrologueBwdArgs a) {
    __shared__ float gz_s[ROWS][65];                   // g0 * (x0 > 0): this wave's atoms x 64 output columns of a pass
    __shared__ float w_s[64][EMB + H + 1];             // the pass's 64 rows of fc1_w, zero-padded
    __shared__ float red_s[ROWS][N_LSTM + 1];          // per-atom LSTM parameter gradients, summed by column
    __shared__ float feat_s[ROWS][MAX_EMB + H];        // the fc1 input of every atom of this workgroup
    __shared__ float dfeat_s[ROWS][MAX_EMB];           // its gradient wrt the embedding part
    __shared__ int aa_s[ROWS];
    const int tid = threadIdx.x;                       // = lane: the workgroup is one wave
    const int rbase = blockIdx.x * ROWS;
    const int r = rbase + tid;
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
#pragma unroll
    for (int i = 0; i < MAX_EMB + H; ++i) feat_s[tid][i] = live ? feat[i] : 0.f;
    aa_s[tid] = live ? (int)id : -1;

    // ---- fc1 backward, 64 output columns per pass: gz tile -> LDS (whole 256-B row segments, every load of the pass
    //      in flight at once), then d feat = fc1_w^T . gz (thread = atom) and d fc1_w, d fc1_b (thread = output
    //      column, atoms in order).  EMB (template) bounds emb_dim at compile time: the inner loops are straight code.
    float dfeat[EMB + H];
#pragma unroll
    for (int i = 0; i < EMB + H; ++i) dfeat[i] = 0.f;
    const int n_emb = a.num_emb * a.emb_dim;
    float* Pw = P + N_LSTM + n_emb;
    float* Pb = Pw + (size_t)a.width * in_w;
    for (int o0 = 0; o0 < a.width; o0 += 64) {
        const int o = o0 + tid;
        const int ncol = a.width - o0 < 64 ? a.width - o0 : 64;
        __syncthreads();
        // this pass's rows of fc1_w, zero-padded to EMB + H columns (broadcast reads below)
        {
            float wrow[EMB + H];
#pragma unroll
            for (int i = 0; i < EMB + H; ++i) wrow[i] = 0.f;
            if (o < a.width) {
#pragma unroll
                for (int e = 0; e < EMB; ++e)
                    if (e < a.emb_dim) wrow[e] = a.fc1_w[(size_t)o * in_w + e];
#pragma unroll
                for (int k = 0; k < H; ++k) wrow[EMB + k] = a.fc1_w[(size_t)o * in_w + a.emb_dim + k];
            }
#pragma unroll
            for (int i = 0; i < EMB + H; ++i) w_s[tid][i] = wrow[i];
        }
        if ((a.width & 3) == 0 && ncol == 64) {
            // lane (row group rg = lane >> 4, column quad c4 = lane & 15) x 16 rows: 32 float4 loads in flight
            const int rg = tid >> 4, c4 = tid & 15;
            float4 xv[16], gv[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int rr = rbase + rg + 4 * j;
                const size_t at = (size_t)(rr < a.R ? rr : a.R - 1) * a.width + o0 + 4 * c4;
                xv[j] = *reinterpret_cast<const float4*>(a.x0 + at);
                gv[j] = *reinterpret_cast<const float4*>(a.g0 + at);
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const bool in = rbase + rg + 4 * j < a.R;
                float* d = &gz_s[rg + 4 * j][4 * c4];
                d[0] = in && xv[j].x > 0.f ? gv[j].x : 0.f;
                d[1] = in && xv[j].y > 0.f ? gv[j].y : 0.f;
                d[2] = in && xv[j].z > 0.f ? gv[j].z : 0.f;
                d[3] = in && xv[j].w > 0.f ? gv[j].w : 0.f;
            }
        } else {
#pragma unroll 8
            for (int q = 0; q < ROWS; ++q) {
                const int rr = rbase + q;
                float v = 0.f;
                if (rr < a.R && o < a.width) {
                    const size_t at = (size_t)rr * a.width + o;
                    v = a.x0[at] > 0.f ? a.g0[at] : 0.f;
                }
                gz_s[q][tid] = v;
            }
        }
        __syncthreads();
#pragma unroll 4
        for (int oo = 0; oo < ncol; ++oo) {
            const float gz = gz_s[tid][oo];
#pragma unroll
            for (int i = 0; i < EMB + H; ++i) dfeat[i] = fmaf(gz, w_s[oo][i], dfeat[i]);
        }
        float acc[EMB + H + 1];
#pragma unroll
        for (int i = 0; i <= EMB + H; ++i) acc[i] = 0.f;
#pragma unroll 4
        for (int q = 0; q < ROWS; ++q) {
            const float gz = gz_s[q][tid];
#pragma unroll
            for (int i = 0; i < EMB; ++i) acc[i] = fmaf(gz, feat_s[q][i], acc[i]);
#pragma unroll
            for (int k = 0; k < H; ++k) acc[EMB + k] = fmaf(gz, feat_s[q][MAX_EMB + k], acc[EMB + k]);
            acc[EMB + H] += gz;
        }
        if (o < a.width) {
#pragma unroll
            for (int i = 0; i < EMB; ++i)
                if (i < a.emb_dim) Pw[(size_t)o * in_w + i] = acc[i];
#pragma unroll
            for (int k = 0; k < H; ++k) Pw[(size_t)o * in_w + a.emb_dim + k] = acc[EMB + k];
            Pb[o] = acc[EMB + H];
        }
    }
#pragma unroll
    for (int e = 0; e < MAX_EMB; ++e) dfeat_s[tid][e] = (live && e < EMB) ? dfeat[e < EMB ? e : 0] : 0.f;
    __syncthreads();
    // embedding: entry (row, col) adds the atoms of its residue type in atom order
    for (int j = tid; j < n_emb; j += ROWS) {
        const int row = j / a.emb_dim, col = j - row * a.emb_dim;
        float s = 0.f;
#pragma unroll 8
        for (int q = 0; q < ROWS; ++q) {
            const float v = dfeat_s[q][col];
            s += aa_s[q] == row ? v : 0.f;
        }
        P[N_LSTM + j] = s;
    }

    // ---- LSTM + lstm_fc backward for this atom; per-thread parameter gradients
    if (!lstm) return;
    float glstm[N_LSTM];
#pragma unroll
    for (int i = 0; i < N_LSTM; ++i) glstm[i] = 0.f;
    if (live) {
        float dh[H] = {0.f, 0.f, 0.f}, dc[H] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < H; ++k) {        // feat[4+k] = fc_b[k] + sum_j fc_w[k][j] h_W[j]
            const float d = dfeat[EMB + k];
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
    // ---- sum over the wave's atoms in atom order: lane i adds column i (and i + 64) of the per-atom gradients
#pragma unroll
    for (int i = 0; i < N_LSTM; ++i) red_s[tid][i] = glstm[i];
    __syncthreads();
    {
        float s0 = 0.f, s1 = 0.f;
        const int c1 = tid < N_LSTM - 64 ? 64 + tid : 64;
#pragma unroll 8
        for (int q = 0; q < ROWS; ++q) { s0 += red_s[q][tid]; s1 += red_s[q][c1]; }
        P[tid] = s0;
        if (tid < N_LSTM - 64) P[64 + tid] = s1;
    }
}

// out[j] = sum over workgroups (in order) of part[b][j]
__global__ __launch_bounds__(256) void reduce_blocks_kernel(const float* __restrict__ part, int blocks, int stride,
                                                            int count, float* __restrict__ out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= count) return;
    float s = 0.f;
#pragma unroll 8
    for (int b = 0; b < blocks; ++b) s += part[(size_t)b * stride + j];     // (independent loads, added in order)
    out[j] = s;
}

// ---------------------------------------------------------------- fc2 backward
// out = x . W^T + b  (x [R,width], W [ow,width]):  dx = g . W,  dW = g^T . x,  db = colsum(g)
__global__ __launch_bounds__(256) void fc_out_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ g, int R, int width, int ow,
                                                         float* __restrict__ dx, float* __restrict__ part, int stride) {
    // a workgroup owns ROWS = 64 rows; thread (quarter = its wave, column c) owns 16 of them: 16 independent row loads
    // in flight instead of a 64-deep chain of round trips (round 3: 256 rows per workgroup, 14 workgroups, 60 us)
    constexpr int QR = ROWS / 4;
    __shared__ float comb[4][64];
    const int tid = threadIdx.x, c0 = tid & 63, part_id = tid >> 6;
    const int rbase = blockIdx.x * ROWS, r0 = rbase + part_id * QR;
    float* P = part + (size_t)blockIdx.x * stride;
    for (int cb = 0; cb < width; cb += 64) {
        const int c = cb + c0;
        const bool col = c < width;
        float xv[QR];
#pragma unroll
        for (int q = 0; q < QR; ++q) xv[q] = (col && r0 + q < R) ? x[(size_t)(r0 + q) * width + c] : 0.f;
        float dxv[QR];
#pragma unroll
        for (int q = 0; q < QR; ++q) dxv[q] = 0.f;
        for (int o = 0; o < ow; ++o) {
            const float wv = col ? w[(size_t)o * width + c] : 0.f;
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < QR; ++q) {
                const float gv = r0 + q < R ? g[(size_t)(r0 + q) * ow + o] : 0.f;      // (uniform in the wave)
                dxv[q] = fmaf(gv, wv, dxv[q]);
                s = fmaf(gv, xv[q], s);
            }
            __syncthreads();
            comb[part_id][c0] = s;
            __syncthreads();
            if (part_id == 0 && col) P[o * width + c] = (comb[0][c0] + comb[1][c0]) + (comb[2][c0] + comb[3][c0]);
        }
#pragma unroll
        for (int q = 0; q < QR; ++q)
            if (col && r0 + q < R) dx[(size_t)(r0 + q) * width + c] = dxv[q];
    }
    // db[o]: the workgroup's rows, butterfly in wave 0
    if (part_id == 0) {
        for (int o = 0; o < ow; ++o) {
            float sb = rbase + c0 < R ? g[(size_t)(rbase + c0) * ow + o] : 0.f;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sb += __shfl_xor(sb, off);
            if (c0 == 0) P[ow * width + o] = sb;
        }
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
    if (p->embedding_dim <= 4) hipLaunchKernelGGL(node_prologue_bwd_kernel<4>, dim3(blocks), dim3(ROWS), 0, s, a);
    else if (p->embedding_dim <= 8) hipLaunchKernelGGL(node_prologue_bwd_kernel<8>, dim3(blocks), dim3(ROWS), 0, s, a);
    else hipLaunchKernelGGL(node_prologue_bwd_kernel<MAX_EMB>, dim3(blocks), dim3(ROWS), 0, s, a);
    auto reduce = [&](int off, int count, float* out) {
        hipLaunchKernelGGL(reduce_blocks_kernel, dim3((count + 255) / 256), dim3(256), 0, s, (const float*)part + off,
                           blocks, tot, count, out);
    };
    if (lstm) {
        reduce(0, N_LSTM, d_lstm);
        reduce(72, 12, d_lstm + N_LSTM);      // b_hh's gradient = b_ih's: its own 12 floats (no two .grad tensors alias)
    }
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
    hipLaunchKernelGGL(fc_out_bwd_kernel, dim3(blocks), dim3(256), 0, s, x, w, g, rows, width, out_width, dx, part, stride);
    hipLaunchKernelGGL(reduce_blocks_kernel, dim3((out_width * width + 255) / 256), dim3(256), 0, s, (const float*)part,
                       blocks, stride, out_width * width, d_w);
    hipLaunchKernelGGL(reduce_blocks_kernel, dim3((out_width + 255) / 256), dim3(256), 0, s,
                       (const float*)part + out_width * width, blocks, stride, out_width, d_b);
    return check_launch("fc_out_bwd");
}
