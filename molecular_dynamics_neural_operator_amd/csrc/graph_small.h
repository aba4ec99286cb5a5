// Radius graph of a short chain in one workgroup (graph.hip), as a device function so that a rollout step can
// run it beside the node prologue in one launch (node_ops.hip: step_head_small).
#pragma once
#include "kernels.h"

namespace mdno {

// the pair test, evaluated exactly as scipy's distance_matrix does on f32 coordinates (graph.hip)
__device__ __forceinline__ bool within(double xi, double yi, double zi, const float* __restrict__ pj,
                                       double cutoff) {
    const double dx = (double)pj[0] - xi, dy = (double)pj[1] - yi, dz = (double)pj[2] - zi;
    const double s = (dx * dx + dy * dy) + dz * dz;
    return sqrt(s) < cutoff;
}

// The three passes in ONE workgroup for a short chain (R <= 128 rows, N <= 128 atoms per member — the reference's
// 28-residue BBA): at that size a launch costs more than the pass it carries.  Same tests, same order: a row's
// neighbour masks are kept in LDS between the count and the fill.
constexpr int kSmallGraphRows = 128;

struct SmallGraphArgs {
    const float* frames;
    int frame;
    const int* t_dev;
    int N, R;
    double cutoff;
    long long cap;
    int *row_ptr, *src, *dst, *num_edges, *status, *zero_words;
    int n_zero;
};

// body of a 1,024-thread workgroup
__device__ __forceinline__ void radius_graph_small_body(const SmallGraphArgs& a) {
    const float* __restrict__ frames = a.frames;
    const int* __restrict__ t_dev = a.t_dev;
    const int frame = a.frame, N = a.N, R = a.R, n_zero = a.n_zero;
    const double cutoff = a.cutoff;
    const long long cap = a.cap;
    int* __restrict__ row_ptr = a.row_ptr;
    int* __restrict__ src = a.src;
    int* __restrict__ dst = a.dst;
    int* __restrict__ num_edges = a.num_edges;
    int* __restrict__ status = a.status;
    int* __restrict__ zero_words = a.zero_words;
    __shared__ unsigned long long mask_s[kSmallGraphRows][2];
    __shared__ int excl_s[kSmallGraphRows + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < n_zero) zero_words[tid] = 0;
    const float* pos = frames + (size_t)(frame + (t_dev ? *t_dev : 0)) * R * 3;
    for (int r = wave; r < R; r += 16) {
        const int m = r / N;
        const float* pm = pos + (size_t)m * N * 3;
        const float* pi = pos + (size_t)r * 3;
        const double xi = pi[0], yi = pi[1], zi = pi[2];
        for (int jb = 0; jb < 2; ++jb) {
            const int j = jb * 64 + lane;
            const bool in = (j < N) && within(xi, yi, zi, pm + (size_t)j * 3, cutoff);
            const unsigned long long mask = __ballot(in);
            if (lane == 0) mask_s[r][jb] = mask;
        }
    }
    __syncthreads();
    if (tid < 128) {      // exclusive scan of the in-degrees over two waves
        const int v = tid < R ? __popcll(mask_s[tid][0]) + __popcll(mask_s[tid][1]) : 0;
        int incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (tid == 63) excl_s[kSmallGraphRows] = incl;      // total of the first wave
        excl_s[tid] = incl - v;
    }
    __syncthreads();
    if (tid < 128) {
        const int e = excl_s[tid] + (tid >= 64 ? excl_s[kSmallGraphRows] : 0);
        if (tid < R) row_ptr[tid] = (int)(e < cap ? e : cap);
        if (tid == R - 1) {
            const long long total = (long long)e + __popcll(mask_s[tid][0]) + __popcll(mask_s[tid][1]);
            const long long ec = total < cap ? total : cap;
            row_ptr[R] = (int)ec;
            *num_edges = (int)ec;
            if (total > cap && status) atomicOr(status, MDNO_STATUS_EDGE_OVERFLOW);
        }
    }
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int r = wave; r < R; r += 16) {
        const int m = r / N;
        long long base = excl_s[r] + (r >= 64 ? excl_s[kSmallGraphRows] : 0);
        for (int jb = 0; jb < 2; ++jb) {
            const unsigned long long mask = mask_s[r][jb];
            if ((mask >> lane) & 1ull) {
                const long long p = base + __popcll(mask & lt);
                if (p < cap) {
                    src[p] = m * N + jb * 64 + lane;
                    if (dst) dst[p] = r;
                }
            }
            base += __popcll(mask);
        }
    }
}


inline bool small_graph_supported(int M, int N) { return (long long)M * N <= kSmallGraphRows && N <= 128; }

}  // namespace mdno
