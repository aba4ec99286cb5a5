// Graph construction on the device: radius graph -> destination-sorted CSR, and COO -> CSR.
//
// Replaces construct_pairdata's host path (graph_kernel.py:362-379): scipy distance_matrix
// (O(N^2) f64), coo_matrix, and a Python loop over edges — executed once per rollout step with two
// PCIe crossings (graph_kernel.py:406-410).  Here the frame never leaves HBM.
//
// Bit-exactness: the pair test is evaluated exactly as scipy does on f32 coordinates — differences,
// squares and the 3-term sum in f64 (squares of f32 differences are exact in f64, so FMA
// contraction cannot change the sum), correctly rounded f64 sqrt, strict `<` against the f64 cutoff.
#include "kernels.h"

#include <hipcub/hipcub.hpp>

namespace mdno {

namespace {

constexpr int kRowsPerBlock = 4;  // one wave per destination row

__device__ __forceinline__ bool within(double xi, double yi, double zi, const float* __restrict__ pj,
                                       double cutoff) {
    const double dx = (double)pj[0] - xi, dy = (double)pj[1] - yi, dz = (double)pj[2] - zi;
    const double s = (dx * dx + dy * dy) + dz * dz;
    return sqrt(s) < cutoff;
}

// Pass 1: in-degree of every row.  Lane l tests atoms j = l, l+64, ... of the row's own member.
__global__ __launch_bounds__(256) void radius_count_kernel(const float* __restrict__ frames, int frame,
                                                           const int* __restrict__ t_dev, int N, int R,
                                                           double cutoff, int* __restrict__ deg) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (r >= R) return;
    const float* pos = frames + (size_t)(frame + (t_dev ? *t_dev : 0)) * R * 3;
    const int m = r / N;
    const float* pm = pos + (size_t)m * N * 3;
    const float* pi = pos + (size_t)r * 3;
    const double xi = pi[0], yi = pi[1], zi = pi[2];
    int cnt = 0;
    for (int j0 = 0; j0 < N; j0 += 64) {
        const int j = j0 + lane;
        const bool in = (j < N) && within(xi, yi, zi, pm + (size_t)j * 3, cutoff);
        cnt += __popcll(__ballot(in));
    }
    if (lane == 0) deg[r] = cnt;
}

// Pass 2: exclusive scan of deg -> row_ptr, clipped at edge_cap (single workgroup; R is small
// next to the per-edge work that follows).
__global__ __launch_bounds__(1024) void scan_rows_kernel(const int* __restrict__ deg, int R, long long cap,
                                                         int* __restrict__ row_ptr, int* __restrict__ num_edges,
                                                         int* __restrict__ status) {
    __shared__ long long wsum[16];
    __shared__ long long carry_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < R; base += 1024) {
        const int i = base + tid;
        long long v = (i < R) ? deg[i] : 0;
        long long incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            long long t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        long long woff = 0;
        for (int k = 0; k < w; ++k) woff += wsum[k];
        const long long carry = carry_s;
        const long long excl = carry + woff + incl - v;
        if (i < R) row_ptr[i] = (int)(excl < cap ? excl : cap);
        __syncthreads();
        if (tid == 1023) carry_s = carry + woff + incl;
        __syncthreads();
    }
    if (tid == 0) {
        const long long total = carry_s;
        const long long e = total < cap ? total : cap;
        row_ptr[R] = (int)e;
        *num_edges = (int)e;
        if (total > cap && status) atomicOr(status, MDNO_STATUS_EDGE_OVERFLOW);
    }
}

// Pass 3: write each row's sources in ascending order (ballot + prefix popcount keeps the order
// deterministic) and, optionally, the destination of every edge.
__global__ __launch_bounds__(256) void radius_fill_kernel(const float* __restrict__ frames, int frame,
                                                          const int* __restrict__ t_dev, int N, int R,
                                                          double cutoff, const int* __restrict__ row_ptr,
                                                          long long cap, int* __restrict__ src,
                                                          int* __restrict__ dst) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (r >= R) return;
    const float* pos = frames + (size_t)(frame + (t_dev ? *t_dev : 0)) * R * 3;
    const int m = r / N;
    const float* pm = pos + (size_t)m * N * 3;
    const float* pi = pos + (size_t)r * 3;
    const double xi = pi[0], yi = pi[1], zi = pi[2];
    long long base = row_ptr[r];
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int j0 = 0; j0 < N; j0 += 64) {
        const int j = j0 + lane;
        const bool in = (j < N) && within(xi, yi, zi, pm + (size_t)j * 3, cutoff);
        const unsigned long long mask = __ballot(in);
        if (in) {
            const long long p = base + __popcll(mask & lt);
            if (p < cap) {
                src[p] = m * N + j;
                if (dst) dst[p] = r;
            }
        }
        base += __popcll(mask);
    }
}

// ---- COO -> CSR helpers
__global__ void coo_keys_kernel(const long long* __restrict__ edge_index, long long E, int* __restrict__ keys,
                                int* __restrict__ vals) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) {
        keys[e] = (int)edge_index[E + e];  // row 1 = target
        vals[e] = (int)e;
    }
}

__global__ void coo_gather_kernel(const long long* __restrict__ edge_index, long long E,
                                  const int* __restrict__ keys_sorted, const int* __restrict__ perm,
                                  int* __restrict__ src, int* __restrict__ dst) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < E) {
        src[p] = (int)edge_index[perm[p]];  // row 0 = source
        if (dst) dst[p] = keys_sorted[p];
    }
}

__global__ void row_ptr_lower_bound_kernel(const int* __restrict__ keys_sorted, long long E, int num_nodes,
                                           int* __restrict__ row_ptr) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > num_nodes) return;
    long long lo = 0, hi = E;  // first position with key >= r
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (keys_sorted[mid] < r) lo = mid + 1; else hi = mid;
    }
    row_ptr[r] = (int)lo;
}

struct CooWs {
    int *keys_in, *keys_out, *vals_in;
    void* cub;
    size_t cub_bytes, total;
};

CooWs carve_coo(void* ws, long long E) {
    CooWs c{};
    size_t cub_bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, cub_bytes, (const int*)nullptr, (int*)nullptr,
                                       (const int*)nullptr, (int*)nullptr, (int)E);
    Carver cv(ws);
    c.keys_in = cv.take<int>(E);
    c.keys_out = cv.take<int>(E);
    c.vals_in = cv.take<int>(E);
    c.cub = cv.take<char>(cub_bytes);
    c.cub_bytes = cub_bytes;
    c.total = cv.used();
    return c;
}

}  // namespace
}  // namespace mdno

int mdno::radius_graph(const float* frames, int frame, const int* t_dev, int M, int N, double cutoff, int* row_ptr,
                       int* src, int* dst, long long edge_cap, int* num_edges, int* status, hipStream_t s) {
    MDNO_REQUIRE(frames && row_ptr && src && num_edges, MDNO_EINVAL, "radius_graph: null pointer");
    MDNO_REQUIRE(M > 0 && N > 0 && edge_cap > 0 && frame >= 0, MDNO_EINVAL, "radius_graph: M=%d N=%d cap=%lld", M, N,
                 edge_cap);
    MDNO_REQUIRE((long long)M * N < (1ll << 31) - 1 && edge_cap < (1ll << 31) - 1, MDNO_EUNSUPPORTED,
                 "radius_graph: row or edge count exceeds int32 indexing");
    const int R = M * N;
    const int blocks = (R + kRowsPerBlock - 1) / kRowsPerBlock;
    // The in-degrees are staged in src[0..R) (needs edge_cap >= R); the fill pass overwrites them.
    MDNO_REQUIRE(edge_cap >= R, MDNO_EINVAL, "radius_graph: edge_cap (%lld) < rows (%d)", edge_cap, R);
    TimedSection ts(KID_GRAPH, s);
    hipLaunchKernelGGL(radius_count_kernel, dim3(blocks), dim3(256), 0, s, frames, frame, t_dev, N, R, cutoff, src);
    hipLaunchKernelGGL(scan_rows_kernel, dim3(1), dim3(1024), 0, s, (const int*)src, R, edge_cap, row_ptr,
                       num_edges, status);
    hipLaunchKernelGGL(radius_fill_kernel, dim3(blocks), dim3(256), 0, s, frames, frame, t_dev, N, R, cutoff,
                       (const int*)row_ptr, edge_cap, src, dst);
    return check_launch("radius_graph");
}

using namespace mdno;

extern "C" int mdno_radius_graph_csr(const float* pos, int M, int N, double cutoff, int32_t* row_ptr,
                                     int32_t* src, int32_t* dst, int64_t edge_cap, int32_t* num_edges,
                                     int32_t* status, void* stream) {
    return radius_graph(pos, 0, nullptr, M, N, cutoff, row_ptr, src, dst, (long long)edge_cap, num_edges, status,
                        static_cast<hipStream_t>(stream));
}

extern "C" size_t mdno_coo_to_csr_workspace_bytes(int64_t E, int num_nodes) {
    (void)num_nodes;
    if (E <= 0) return 256;
    return carve_coo(nullptr, E).total;
}

extern "C" int mdno_coo_to_csr(const int64_t* edge_index, int64_t E, int num_nodes, int32_t* row_ptr,
                               int32_t* src, int32_t* dst, int32_t* perm, void* workspace,
                               size_t workspace_bytes, void* stream) {
    MDNO_REQUIRE(row_ptr && num_nodes > 0 && E >= 0, MDNO_EINVAL, "mdno_coo_to_csr: bad arguments");
    MDNO_REQUIRE(E < (1ll << 31) - 1, MDNO_EUNSUPPORTED, "mdno_coo_to_csr: E exceeds int32 indexing");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (E == 0) {
        MDNO_HIP(hipMemsetAsync(row_ptr, 0, sizeof(int) * (size_t)(num_nodes + 1), s));
        return MDNO_OK;
    }
    MDNO_REQUIRE(edge_index && src && perm && workspace, MDNO_EINVAL, "mdno_coo_to_csr: null pointer");
    CooWs c = carve_coo(workspace, E);
    MDNO_REQUIRE(workspace_bytes >= c.total, MDNO_EWORKSPACE, "mdno_coo_to_csr: workspace %zu < %zu",
                 workspace_bytes, c.total);
    const int tb = 256;
    const int nb = (int)((E + tb - 1) / tb);
    hipLaunchKernelGGL(coo_keys_kernel, dim3(nb), dim3(tb), 0, s, (const long long*)edge_index, (long long)E,
                       c.keys_in, c.vals_in);
    int end_bit = 1;
    while ((1ll << end_bit) < (long long)num_nodes && end_bit < 31) ++end_bit;
    size_t cub_bytes = c.cub_bytes;
    MDNO_HIP(hipcub::DeviceRadixSort::SortPairs(c.cub, cub_bytes, (const int*)c.keys_in, c.keys_out,
                                                (const int*)c.vals_in, perm, (int)E, 0, end_bit, s));
    hipLaunchKernelGGL(coo_gather_kernel, dim3(nb), dim3(tb), 0, s, (const long long*)edge_index, (long long)E,
                       (const int*)c.keys_out, (const int*)perm, src, dst);
    hipLaunchKernelGGL(row_ptr_lower_bound_kernel, dim3((num_nodes + 1 + tb - 1) / tb), dim3(tb), 0, s,
                       (const int*)c.keys_out, (long long)E, num_nodes, row_ptr);
    return check_launch("mdno_coo_to_csr");
}
