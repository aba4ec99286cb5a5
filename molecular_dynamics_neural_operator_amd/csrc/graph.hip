// Graph construction on the device: radius graph -> destination-sorted CSR, and COO -> CSR.
//
// Replaces construct_pairdata's host path (graph_kernel.py:362-379): scipy distance_matrix
// (O(N^2) f64), coo_matrix, and a Python loop over edges — executed once per rollout step with two
// PCIe crossings (graph_kernel.py:406-410).  Here the frame never leaves HBM.  Three forms, one result: one workgroup
// for a short chain (graph_small.h), brute force per member (N^2 pair tests: 15 us at N = 504), and from 8,192 atoms
// per member on a cell list (below).
//
// Bit-exactness: the pair test is evaluated exactly as scipy does on f32 coordinates — differences,
// squares and the 3-term sum in f64 (squares of f32 differences are exact in f64, so FMA
// contraction cannot change the sum), correctly rounded f64 sqrt, strict `<` against the f64 cutoff.
#include "kernels.h"
#include "graph_small.h"

namespace mdno {

namespace {

constexpr int kRowsPerBlock = 4;  // one wave per destination row


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
                                                         int* __restrict__ status, int* __restrict__ zero_words,
                                                         int n_zero) {
    __shared__ long long wsum[16];
    __shared__ long long carry_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid < n_zero) zero_words[tid] = 0;
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

// ---------------------------------------------------------------- radius graph through a cell list (large members)
// The brute-force passes above test N^2 pairs per member: 15 us at N = 504, 8 ms at N = 50,000, 0.8 s at 500,000.
// From kCellMinAtoms atoms per member on, atoms are binned into cubic cells of edge >= cutoff (cell size chosen per
// member from its bounding box so that a member has at most kCellMax cells), and a destination tests the atoms of its
// 27 neighbouring cells only — nine contiguous runs of the cell-sorted atom list (x is the fastest cell index).  The
// PAIR TEST is the same function (`within`: f64 on the f32 coordinates), and every pair within the cutoff lies in
// neighbouring cells (the cell edge is the cutoff times 1 + 1e-6, so rounding in the cell index cannot separate them),
// so the result is the same set of edges.  ORDER: hits are recorded as bits of an atom mask kept in LDS (one wave per
// destination, 65,536 atoms per pass over the mask) and read back in ascending order — the sources of a row come out
// sorted whatever order the cells delivered them in, and the graph is bit-identical to the brute-force one (tested on
// the whole 50k-atom box).
constexpr int kCellMinAtoms = 8192;
constexpr int kCellMax = 32768;          // cells per member (<= 32 per axis)
constexpr int kMaskBits = 65536;         // atoms per pass over a wave's LDS mask (8 KiB)

struct CellWs {
    float* box;        // [M][8]: origin x, y, z, inverse cell edge; then (as ints) nx, ny, nz, nx*ny*nz
    int* cell_of;      // [R]
    int* sorted;       // [R] atoms of a member ordered by cell (index inside the member)
    int* start;        // [M][kCellMax + 1] exclusive scan of the cell populations (also used as counters)
    int* cursor;       // [M][kCellMax]
    size_t total;
};
static CellWs carve_cells(void* ws, int M, int N) {
    CellWs c{};
    Carver cv(ws);
    const size_t R = (size_t)M * N;
    c.box = cv.take<float>((size_t)M * 8);
    c.cell_of = cv.take<int>(R);
    c.sorted = cv.take<int>(R);
    c.start = cv.take<int>((size_t)M * (kCellMax + 1));
    c.cursor = cv.take<int>((size_t)M * kCellMax);
    c.total = cv.used();
    return c;
}

// one workgroup per member: bounding box -> cell grid; zeroes the member's cell counters
__global__ __launch_bounds__(1024) void cell_box_kernel(const float* __restrict__ frames, int frame, const int* __restrict__ t_dev,
                                                        int N, int R, double cutoff, float* __restrict__ box,
                                                        int* __restrict__ start, int* __restrict__ cursor) {
    __shared__ float red[6][16];
    const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* pm = frames + (size_t)(frame + (t_dev ? *t_dev : 0)) * R * 3 + (size_t)m * N * 3;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int j = tid; j < N; j += 1024)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float v = pm[(size_t)j * 3 + d];
            lo[d] = fminf(lo[d], v);
            hi[d] = fmaxf(hi[d], v);
        }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[d] = fminf(lo[d], __shfl_xor(lo[d], o));
            hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o));
        }
        if (lane == 0) { red[d][w] = lo[d]; red[3 + d][w] = hi[d]; }
    }
    for (int c = tid; c < kCellMax; c += 1024) { start[(size_t)m * (kCellMax + 1) + c] = 0; cursor[(size_t)m * kCellMax + c] = 0; }
    __syncthreads();
    if (tid == 0) {
        double ext = 0.0;
        float org[3];
        for (int d = 0; d < 3; ++d) {
            float a = red[d][0], b = red[3 + d][0];
            for (int k = 1; k < 16; ++k) { a = fminf(a, red[d][k]); b = fmaxf(b, red[3 + d][k]); }
            org[d] = a;
            if ((double)b - (double)a > ext) ext = (double)b - (double)a;
            red[d][0] = a; red[3 + d][0] = b;
        }
        // cell edge: the cutoff (+1e-6 relative), enlarged until no axis has more than 32 cells
        double edge = cutoff * (1.0 + 1.0e-6);
        if (!(edge > 0.0)) edge = 1.0;
        if (ext / edge >= 31.0) edge = ext / 31.0;
        int n[3];
        for (int d = 0; d < 3; ++d) {
            const double e = (double)red[3 + d][0] - (double)red[d][0];
            int c = (int)(e / edge) + 1;
            n[d] = c < 1 ? 1 : (c > 32 ? 32 : c);
        }
        float* b = box + (size_t)m * 8;
        b[0] = org[0]; b[1] = org[1]; b[2] = org[2];
        b[3] = (float)(1.0 / edge);
        int* bi = reinterpret_cast<int*>(b + 4);
        bi[0] = n[0]; bi[1] = n[1]; bi[2] = n[2]; bi[3] = n[0] * n[1] * n[2];
    }
}

__device__ __forceinline__ void cell_coords(const float* __restrict__ b, const float* __restrict__ p, int& cx, int& cy, int& cz) {
    const int* bi = reinterpret_cast<const int*>(b + 4);
    const double inv = (double)b[3];
    cx = (int)(((double)p[0] - (double)b[0]) * inv);
    cy = (int)(((double)p[1] - (double)b[1]) * inv);
    cz = (int)(((double)p[2] - (double)b[2]) * inv);
    cx = cx < 0 ? 0 : (cx >= bi[0] ? bi[0] - 1 : cx);
    cy = cy < 0 ? 0 : (cy >= bi[1] ? bi[1] - 1 : cy);
    cz = cz < 0 ? 0 : (cz >= bi[2] ? bi[2] - 1 : cz);
}

__global__ __launch_bounds__(256) void cell_count_kernel(const float* __restrict__ frames, int frame, const int* __restrict__ t_dev,
                                                         int N, int R, const float* __restrict__ box,
                                                         int* __restrict__ cell_of, int* __restrict__ start) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const float* pos = frames + (size_t)(frame + (t_dev ? *t_dev : 0)) * R * 3;
    const int m = r / N;
    const float* b = box + (size_t)m * 8;
    const int* bi = reinterpret_cast<const int*>(b + 4);
    int cx, cy, cz;
    cell_coords(b, pos + (size_t)r * 3, cx, cy, cz);
    const int c = (cz * bi[1] + cy) * bi[0] + cx;
    cell_of[r] = c;
    atomicAdd(&start[(size_t)m * (kCellMax + 1) + c], 1);       // (integer counts: deterministic)
}

// one workgroup per member: populations -> exclusive scan in place (start[ncell] = N)
__global__ __launch_bounds__(1024) void cell_scan_kernel(const float* __restrict__ box, int* __restrict__ start) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ncell = reinterpret_cast<const int*>(box + (size_t)m * 8 + 4)[3];
    int* st = start + (size_t)m * (kCellMax + 1);
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < ncell + 1; base += 1024) {
        const int i = base + tid;
        const int v = i < ncell ? st[i] : 0;
        int incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < w; ++k) woff += wsum[k];
        const int carry = carry_s;
        if (i <= ncell) st[i] = carry + woff + incl - v;
        __syncthreads();
        if (tid == 1023) carry_s = carry + woff + incl;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void cell_scatter_kernel(int N, int R, const int* __restrict__ cell_of,
                                                           const int* __restrict__ start, int* __restrict__ cursor,
                                                           int* __restrict__ sorted) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const int m = r / N, c = cell_of[r];
    const int slot = start[(size_t)m * (kCellMax + 1) + c] + atomicAdd(&cursor[(size_t)m * kCellMax + c], 1);
    sorted[(size_t)m * N + slot] = r - m * N;       // (order inside a cell is arbitrary: the mask below restores index order)
}

// One wave per destination row.  FILL = false: in-degree -> deg[r].  FILL = true: the row's sources, ascending.
template <bool FILL>
__global__ __launch_bounds__(256) void radius_cell_kernel(const float* __restrict__ frames, int frame,
                                                          const int* __restrict__ t_dev, int N, int R, double cutoff,
                                                          const float* __restrict__ box, const int* __restrict__ start,
                                                          const int* __restrict__ sorted, int* __restrict__ deg,
                                                          const int* __restrict__ row_ptr, long long cap,
                                                          int* __restrict__ src, int* __restrict__ dst) {
    __shared__ unsigned mask_s[kRowsPerBlock][kMaskBits / 32];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * kRowsPerBlock + wv;
    if (r >= R) return;
    const float* pos = frames + (size_t)(frame + (t_dev ? *t_dev : 0)) * R * 3;
    const int m = r / N;
    const float* pm = pos + (size_t)m * N * 3;
    const float* pi = pos + (size_t)r * 3;
    const double xi = pi[0], yi = pi[1], zi = pi[2];
    const float* b = box + (size_t)m * 8;
    const int* bi = reinterpret_cast<const int*>(b + 4);
    const int nx = bi[0], ny = bi[1], nz = bi[2];
    int cx, cy, cz;
    cell_coords(b, pi, cx, cy, cz);
    const int* st = start + (size_t)m * (kCellMax + 1);
    const int* so = sorted + (size_t)m * N;
    unsigned* mask = mask_s[wv];
    long long base = FILL ? row_ptr[r] : 0;
    int cnt = 0;
    for (int w0 = 0; w0 < N; w0 += kMaskBits) {          // windows of atom indices (one for N <= 65,536)
        const int wn = N - w0 < kMaskBits ? N - w0 : kMaskBits;
        const int words = (wn + 31) >> 5;
        for (int i = lane; i < words; i += 64) mask[i] = 0u;
        __builtin_amdgcn_wave_barrier();
        for (int dz = -1; dz <= 1; ++dz) {
            const int z = cz + dz;
            if (z < 0 || z >= nz) continue;
            for (int dy = -1; dy <= 1; ++dy) {
                const int y = cy + dy;
                if (y < 0 || y >= ny) continue;
                const int x0 = cx > 0 ? cx - 1 : 0, x1 = cx + 1 < nx ? cx + 1 : nx - 1;
                const int c0 = (z * ny + y) * nx + x0, c1 = (z * ny + y) * nx + x1;
                const int p0 = st[c0], p1 = st[c1 + 1];
                for (int p = p0 + lane; p < p1; p += 64) {
                    const int j = so[p];
                    if (j >= w0 && j < w0 + wn && within(xi, yi, zi, pm + (size_t)j * 3, cutoff))
                        atomicOr(&mask[(j - w0) >> 5], 1u << ((j - w0) & 31));
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // read the mask back in ascending order
        for (int i0 = 0; i0 < words; i0 += 64) {
            const int i = i0 + lane;
            const unsigned bits = i < words ? mask[i] : 0u;
            const int pc = __popc(bits);
            if (!FILL) {
                cnt += pc;
            } else {
                int incl = pc;
                for (int o = 1; o < 64; o <<= 1) {
                    const int t = __shfl_up(incl, o);
                    if (lane >= o) incl += t;
                }
                long long at = base + incl - pc;
                unsigned rest = bits;
                while (rest) {
                    const int bpos = __ffs(rest) - 1;
                    rest &= rest - 1;
                    if (at < cap) {
                        src[at] = m * N + w0 + i * 32 + bpos;
                        if (dst) dst[at] = r;
                    }
                    ++at;
                }
                base += __shfl(incl, 63);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (!FILL) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        if (lane == 0) deg[r] = cnt;
    }
}

// (the three passes in one workgroup for a short chain: graph_small.h)
__global__ __launch_bounds__(1024) void radius_graph_small_kernel(SmallGraphArgs a) { radius_graph_small_body(a); }

// ---- COO -> CSR: counting sort by destination (keys are node ids < num_nodes)
//   count   in-degree of every node (integer atomics: the counts are deterministic); node ids outside
//           [0, num_nodes) set a status bit and are clamped so that every later access stays in bounds
//   scan    row_ptr (scan_rows_kernel above)
//   slot    every edge takes a slot of its destination's row through an atomic cursor — any order
//   sort    each row's edge ids are put in ascending order: ids are unique, so the result is THE stable
//           order by destination whatever order the slots were handed out in, and the sort is
//           run-to-run deterministic.  Rank sort: rows are short (a contact-map row has ~10^2
//           entries); rows above kBigRow entries take a whole workgroup each.
constexpr int kBigRow = 2048;

__device__ __forceinline__ int clamp_node(long long v, int num_nodes, bool& bad) {
    if (v < 0 || v >= num_nodes) {
        bad = true;
        return v < 0 ? 0 : num_nodes - 1;
    }
    return (int)v;
}

// The edge list the sort reads: i64 [2,E] (row 0 = source, row 1 = target) as torch_geometric has it, or two i32
// arrays (mdno_csr_by_source: the arrays of an existing CSR with their roles swapped).
struct EdgeView {
    const long long* ei;      // i64 [2,E], or NULL
    const int *s32, *t32;     // i32 [E] each
    long long E;
    __device__ __forceinline__ long long source(long long e) const { return ei ? ei[e] : (long long)s32[e]; }
    __device__ __forceinline__ long long target(long long e) const { return ei ? ei[E + e] : (long long)t32[e]; }
};

__global__ __launch_bounds__(256) void coo_count_kernel(const EdgeView edge_index, long long E,
                                                        int num_nodes, int* __restrict__ deg,
                                                        int* __restrict__ status) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    bool bad = false;
    (void)clamp_node(edge_index.source(e), num_nodes, bad);
    const int d = clamp_node(edge_index.target(e), num_nodes, bad);
    if (bad && status) atomicOr(status, MDNO_STATUS_BAD_EDGE_INDEX);
    atomicAdd(&deg[d], 1);
}

__global__ __launch_bounds__(256) void coo_slot_kernel(const EdgeView edge_index, long long E,
                                                       int num_nodes, const int* __restrict__ row_ptr,
                                                       int* __restrict__ cursor, int* __restrict__ ids) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    bool bad = false;
    const int d = clamp_node(edge_index.target(e), num_nodes, bad);
    ids[row_ptr[d] + atomicAdd(&cursor[d], 1)] = (int)e;
}

// rank of an id = number of smaller ids in its row; `ids` is read-only here, results go to perm/src/dst
__device__ __forceinline__ void coo_place(const EdgeView& edge_index, int num_nodes, int row, int beg,
                                          int rank, int id, int* __restrict__ perm, int* __restrict__ src,
                                          int* __restrict__ dst) {
    bool bad = false;
    perm[beg + rank] = id;
    src[beg + rank] = clamp_node(edge_index.source(id), num_nodes, bad);
    if (dst) dst[beg + rank] = row;
}

__global__ __launch_bounds__(256) void coo_row_sort_kernel(const EdgeView edge_index, int num_nodes,
                                                           const int* __restrict__ row_ptr,
                                                           const int* __restrict__ ids, int* __restrict__ perm,
                                                           int* __restrict__ src, int* __restrict__ dst) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= num_nodes) return;
    const int beg = row_ptr[row], deg = row_ptr[row + 1] - beg;
    if (deg > kBigRow) return;                      // coo_big_row_sort_kernel
    for (int i = lane; i < deg; i += 64) {
        const int id = ids[beg + i];
        int rank = 0;
        for (int j = 0; j < deg; ++j) rank += ids[beg + j] < id;   // same address in every lane: one broadcast load
        coo_place(edge_index, num_nodes, row, beg, rank, id, perm, src, dst);
    }
}

__global__ __launch_bounds__(1024) void coo_big_row_sort_kernel(const EdgeView edge_index,
                                                                int num_nodes, const int* __restrict__ row_ptr,
                                                                const int* __restrict__ ids, int* __restrict__ perm,
                                                                int* __restrict__ src, int* __restrict__ dst) {
    __shared__ int tile[1024];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int beg = row_ptr[row], deg = row_ptr[row + 1] - beg;
    if (deg <= kBigRow) return;
    for (int i0 = 0; i0 < deg; i0 += 1024) {        // uniform trip counts: the barriers below are reached by all
        const int i = i0 + tid;
        const int id = i < deg ? ids[beg + i] : 0x7fffffff;
        int rank = 0;
        for (int j0 = 0; j0 < deg; j0 += 1024) {
            __syncthreads();
            tile[tid] = j0 + tid < deg ? ids[beg + j0 + tid] : 0x7fffffff;
            __syncthreads();
            const int n = deg - j0 < 1024 ? deg - j0 : 1024;
            for (int j = 0; j < n; ++j) rank += tile[j] < id;
        }
        if (i < deg) coo_place(edge_index, num_nodes, row, beg, rank, id, perm, src, dst);
    }
}

struct CooWs {
    int *deg, *cursor, *ids, *scratch;
    size_t total;
};

CooWs carve_coo(void* ws, long long E, int num_nodes) {
    CooWs c{};
    Carver cv(ws);
    c.deg = cv.take<int>((size_t)2 * num_nodes);   // deg | cursor: one memset
    c.cursor = c.deg ? c.deg + num_nodes : nullptr;
    c.ids = cv.take<int>((size_t)E);
    c.scratch = cv.take<int>(64);
    c.total = cv.used();
    return c;
}

}  // namespace
}  // namespace mdno

size_t mdno::radius_graph_scratch_bytes(int M, int N) {
    return (M > 0 && N >= kCellMinAtoms) ? carve_cells(nullptr, M, N).total : 0;
}

int mdno::radius_graph(const float* frames, int frame, const int* t_dev, int M, int N, double cutoff, int* row_ptr,
                       int* src, int* dst, long long edge_cap, int* num_edges, int* status, hipStream_t s,
                       int* zero_words, int n_zero, void* scratch, size_t scratch_bytes) {
    MDNO_REQUIRE(frames && row_ptr && src && num_edges, MDNO_EINVAL, "radius_graph: null pointer");
    MDNO_REQUIRE(M > 0 && N > 0 && edge_cap > 0 && frame >= 0, MDNO_EINVAL, "radius_graph: M=%d N=%d cap=%lld", M, N,
                 edge_cap);
    MDNO_REQUIRE((long long)M * N < (1ll << 31) - 1 && edge_cap < (1ll << 31) - 1, MDNO_EUNSUPPORTED,
                 "radius_graph: row or edge count exceeds int32 indexing");
    const int R = M * N;
    const int blocks = (R + kRowsPerBlock - 1) / kRowsPerBlock;
    // The in-degrees are staged in src[0..R) (needs edge_cap >= R); the fill pass overwrites them.
    MDNO_REQUIRE(edge_cap >= R, MDNO_EINVAL, "radius_graph: edge_cap (%lld) < rows (%d)", edge_cap, R);
    MDNO_REQUIRE(n_zero >= 0 && n_zero <= 64 && (n_zero == 0 || zero_words), MDNO_EINVAL, "radius_graph: n_zero=%d", n_zero);
    TimedSection ts(KID_GRAPH, s);
    if (small_graph_supported(M, N)) {
        hipLaunchKernelGGL(radius_graph_small_kernel, dim3(1), dim3(1024), 0, s,
                           SmallGraphArgs{frames, frame, t_dev, N, R, cutoff, edge_cap, row_ptr, src, dst, num_edges,
                                          status, zero_words, n_zero});
        return check_launch("radius_graph");
    }
    if (N >= kCellMinAtoms && scratch && scratch_bytes >= radius_graph_scratch_bytes(M, N)) {
        // cell list: bounding box + grid, populations, scan, scatter, then the two passes over 27 cells per row
        const CellWs c = carve_cells(scratch, M, N);
        const unsigned rb = (unsigned)((R + 255) / 256);
        hipLaunchKernelGGL(cell_box_kernel, dim3(M), dim3(1024), 0, s, frames, frame, t_dev, N, R, cutoff, c.box, c.start, c.cursor);
        hipLaunchKernelGGL(cell_count_kernel, dim3(rb), dim3(256), 0, s, frames, frame, t_dev, N, R, (const float*)c.box,
                           c.cell_of, c.start);
        hipLaunchKernelGGL(cell_scan_kernel, dim3(M), dim3(1024), 0, s, (const float*)c.box, c.start);
        hipLaunchKernelGGL(cell_scatter_kernel, dim3(rb), dim3(256), 0, s, N, R, (const int*)c.cell_of, (const int*)c.start,
                           c.cursor, c.sorted);
        hipLaunchKernelGGL(radius_cell_kernel<false>, dim3(blocks), dim3(256), 0, s, frames, frame, t_dev, N, R, cutoff,
                           (const float*)c.box, (const int*)c.start, (const int*)c.sorted, src, (const int*)nullptr, edge_cap,
                           (int*)nullptr, (int*)nullptr);
        hipLaunchKernelGGL(scan_rows_kernel, dim3(1), dim3(1024), 0, s, (const int*)src, R, edge_cap, row_ptr,
                           num_edges, status, zero_words, n_zero);
        hipLaunchKernelGGL(radius_cell_kernel<true>, dim3(blocks), dim3(256), 0, s, frames, frame, t_dev, N, R, cutoff,
                           (const float*)c.box, (const int*)c.start, (const int*)c.sorted, (int*)nullptr, (const int*)row_ptr,
                           edge_cap, src, dst);
        return check_launch("radius_graph (cell list)");
    }
    hipLaunchKernelGGL(radius_count_kernel, dim3(blocks), dim3(256), 0, s, frames, frame, t_dev, N, R, cutoff, src);
    hipLaunchKernelGGL(scan_rows_kernel, dim3(1), dim3(1024), 0, s, (const int*)src, R, edge_cap, row_ptr,
                       num_edges, status, zero_words, n_zero);
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

extern "C" size_t mdno_radius_graph_workspace_bytes(int M, int N) { return radius_graph_scratch_bytes(M, N); }

extern "C" int mdno_radius_graph_csr_ws(const float* pos, int M, int N, double cutoff, int32_t* row_ptr, int32_t* src,
                                        int32_t* dst, int64_t edge_cap, int32_t* num_edges, int32_t* status,
                                        void* workspace, size_t workspace_bytes, void* stream) {
    return radius_graph(pos, 0, nullptr, M, N, cutoff, row_ptr, src, dst, (long long)edge_cap, num_edges, status,
                        static_cast<hipStream_t>(stream), nullptr, 0, workspace, workspace_bytes);
}

extern "C" size_t mdno_coo_to_csr_workspace_bytes(int64_t E, int num_nodes) {
    if (E <= 0 || num_nodes <= 0) return 256;
    return carve_coo(nullptr, E, num_nodes).total;
}

static int coo_sort(const EdgeView ei, long long E, int num_nodes, int32_t* row_ptr, int32_t* src, int32_t* dst,
                    int32_t* perm, int32_t* num_edges, int32_t* status, void* workspace, size_t workspace_bytes,
                    hipStream_t s, const char* who) {
    MDNO_REQUIRE(row_ptr && num_nodes > 0 && E >= 0, MDNO_EINVAL, "%s: bad arguments", who);
    MDNO_REQUIRE(E < (1ll << 31) - 1, MDNO_EUNSUPPORTED, "%s: E exceeds int32 indexing", who);
    if (E == 0) {
        MDNO_HIP(hipMemsetAsync(row_ptr, 0, sizeof(int) * (size_t)(num_nodes + 1), s));
        if (num_edges) MDNO_HIP(hipMemsetAsync(num_edges, 0, sizeof(int), s));
        return MDNO_OK;
    }
    MDNO_REQUIRE((ei.ei || (ei.s32 && ei.t32)) && src && perm && workspace, MDNO_EINVAL, "%s: null pointer", who);
    CooWs c = carve_coo(workspace, E, num_nodes);
    MDNO_REQUIRE(workspace_bytes >= c.total, MDNO_EWORKSPACE, "%s: workspace %zu < %zu", who, workspace_bytes, c.total);
    const unsigned nb = (unsigned)((E + 255) / 256);
    MDNO_HIP(hipMemsetAsync(c.deg, 0, sizeof(int) * 2 * (size_t)num_nodes, s));
    hipLaunchKernelGGL(coo_count_kernel, dim3(nb), dim3(256), 0, s, ei, E, num_nodes, c.deg, status);
    // (the scan also leaves the edge count on the device for the caller: no fill launch on the host side)
    hipLaunchKernelGGL(scan_rows_kernel, dim3(1), dim3(1024), 0, s, (const int*)c.deg, num_nodes, E, row_ptr,
                       num_edges ? num_edges : c.scratch, (int*)nullptr, (int*)nullptr, 0);
    hipLaunchKernelGGL(coo_slot_kernel, dim3(nb), dim3(256), 0, s, ei, E, num_nodes, (const int*)row_ptr, c.cursor, c.ids);
    hipLaunchKernelGGL(coo_row_sort_kernel, dim3((num_nodes + 3) / 4), dim3(256), 0, s, ei, num_nodes,
                       (const int*)row_ptr, (const int*)c.ids, perm, src, dst);
    // rows above kBigRow entries exist only if E does
    if (E > kBigRow)
        hipLaunchKernelGGL(coo_big_row_sort_kernel, dim3(num_nodes), dim3(1024), 0, s, ei, num_nodes,
                           (const int*)row_ptr, (const int*)c.ids, perm, src, dst);
    return check_launch(who);
}

extern "C" int mdno_coo_to_csr(const int64_t* edge_index, int64_t E, int num_nodes, int32_t* row_ptr,
                               int32_t* src, int32_t* dst, int32_t* perm, int32_t* num_edges, int32_t* status,
                               void* workspace, size_t workspace_bytes, void* stream) {
    const EdgeView ev{(const long long*)edge_index, nullptr, nullptr, (long long)E};
    return coo_sort(ev, (long long)E, num_nodes, row_ptr, src, dst, perm, num_edges, status, workspace, workspace_bytes,
                    static_cast<hipStream_t>(stream), "mdno_coo_to_csr");
}

extern "C" int mdno_csr_by_source(const int32_t* csr_src, const int32_t* csr_dst, int64_t E, int num_nodes,
                                  int32_t* row_ptr, int32_t* nbr, int32_t* rowid, int32_t* perm, int32_t* status,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    // the same sort with the two arrays' roles swapped: rows = SOURCES, row entries = the targets they send to
    const EdgeView ev{nullptr, csr_dst, csr_src, (long long)E};
    return coo_sort(ev, (long long)E, num_nodes, row_ptr, nbr, rowid, perm, nullptr, status, workspace, workspace_bytes,
                    static_cast<hipStream_t>(stream), "mdno_csr_by_source");
}
