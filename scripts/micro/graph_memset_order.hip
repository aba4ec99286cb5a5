// Do hipMemsetAsync nodes keep their place inside a captured stream?  (round 2 saw run-to-run
// different results with memset nodes between kernels of a captured rollout step and replaced them by kernels.)
//
// The captured sequence repeats, STEPS times:   K_set(flag |= 1 + i)  ->  K_read (flag up: cached in every L2)  ->
// memset(flag, 0)  ->  K_read(out[i] |= flag).
// In stream order every out[i] must stay 0 (a reader that sees the flag up ORs it into out[i]).  The program
// prints the captured graph (node types and dependency edges), replays it, and counts violations — once with the
// memsets on 4-byte words in the middle of an allocation (what the library had) and once on a 16-byte-aligned block of
// its own; plain launches on the same stream are the control.
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/graph_memset_order.hip -o /tmp/graph_memset_order && /tmp/graph_memset_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// the library's pattern: many workgroups (all 8 XCDs) raise the flag with atomicOr; many workgroups of the next
// kernel read it with a non-temporal load and act on it (here: record what they saw)
__global__ void k_set(int* flag, int v) { if (threadIdx.x == 0) atomicOr(flag, v | 1); }
__global__ void k_read(const int* flag, int* out, int i) {
    const int f = __builtin_nontemporal_load(flag);
    if (threadIdx.x == 0 && f != 0) atomicOr(out + i, f);
}
__global__ void k_busy(float* buf, int n) {          // something long enough for an unordered memset to overtake
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += buf[i];
    if (s == 12345.f) buf[0] = s;
}

static const char* type_name(hipGraphNodeType t) {
    switch (t) {
        case hipGraphNodeTypeKernel: return "kernel";
        case hipGraphNodeTypeMemset: return "memset";
        case hipGraphNodeTypeMemcpy: return "memcpy";
        case hipGraphNodeTypeEmpty: return "empty";
        default: return "other";
    }
}

static int enqueue(hipStream_t s, int* flag, int* out, float* buf, int steps) {
    for (int i = 0; i < steps; ++i) {
        hipLaunchKernelGGL(k_busy, dim3(64), dim3(256), 0, s, buf, 1 << 16);
        hipLaunchKernelGGL(k_set, dim3(512), dim3(64), 0, s, flag, 1 + i);
        // every XCD reads the flag WHILE IT IS UP (so that a copy of the line with the old value sits in each L2),
        // as the library's consumers do on a fallback forward; scratch[] takes what they saw
        hipLaunchKernelGGL(k_read, dim3(512), dim3(64), 0, s, (const int*)flag, out + 64, i);
        CK(hipMemsetAsync(flag, 0, sizeof(int), s));
        hipLaunchKernelGGL(k_read, dim3(512), dim3(64), 0, s, (const int*)flag, out, i);
    }
    return 0;
}

static int run_case(const char* what, int* flag, int* out, float* buf, int steps, bool dump) {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    // control: plain launches
    CK(hipMemset(out, 0, steps * sizeof(int)));
    if (enqueue(s, flag, out, buf, steps)) return 1;
    CK(hipStreamSynchronize(s));
    std::vector<int> h(steps);
    CK(hipMemcpy(h.data(), out, steps * sizeof(int), hipMemcpyDeviceToHost));
    int bad_plain = 0;
    for (int v : h) bad_plain += v != 0;
    // captured
    hipGraph_t g;
    hipGraphExec_t ex;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    if (enqueue(s, flag, out, buf, steps)) return 1;
    CK(hipStreamEndCapture(s, &g));
    size_t n = 0;
    CK(hipGraphGetNodes(g, nullptr, &n));
    std::vector<hipGraphNode_t> nodes(n);
    CK(hipGraphGetNodes(g, nodes.data(), &n));
    size_t memsets = 0, roots = 0, chain_ok = 0;
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType t;
        CK(hipGraphNodeGetType(nodes[i], &t));
        size_t nd = 0;
        CK(hipGraphNodeGetDependencies(nodes[i], nullptr, &nd));
        std::vector<hipGraphNode_t> deps(nd);
        if (nd) CK(hipGraphNodeGetDependencies(nodes[i], deps.data(), &nd));
        memsets += t == hipGraphNodeTypeMemset;
        roots += nd == 0;
        // in a single captured stream every node but the first must depend on exactly its predecessor
        if (i > 0 && nd == 1 && deps[0] == nodes[i - 1]) ++chain_ok;
        if (dump && i < 9) {
            printf("    node %zu: %-6s deps %zu", i, type_name(t), nd);
            for (size_t d = 0; d < nd; ++d)
                for (size_t j = 0; j < n; ++j)
                    if (nodes[j] == deps[d]) printf(" <- node %zu", j);
            if (t == hipGraphNodeTypeMemset) {
                hipMemsetParams mp;
                CK(hipGraphMemsetNodeGetParams(nodes[i], &mp));
                printf("   [dst %p value %u elementSize %u width %zu height %zu]", mp.dst, mp.value, mp.elementSize, mp.width, mp.height);
            }
            printf("\n");
        }
    }
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    int bad_graph = 0, replays = 200;
    for (int r = 0; r < replays; ++r) {
        CK(hipMemsetAsync(out, 0, steps * sizeof(int), s));
        CK(hipGraphLaunch(ex, s));
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), out, steps * sizeof(int), hipMemcpyDeviceToHost));
        for (int v : h) bad_graph += v != 0;
    }
    printf("%s: %zu nodes (%zu memset), %zu without dependency, %zu of %zu form the stream's chain;"
           " violations: plain launches %d / %d reads, graph replays %d / %d reads\n",
           what, n, memsets, roots, chain_ok, n - 1, bad_plain, steps, bad_graph, replays * steps);
    CK(hipGraphExecDestroy(ex));
    CK(hipGraphDestroy(g));
    CK(hipStreamDestroy(s));
    return 0;
}

int main() {
    const int steps = 24;
    int *block, *out;
    float* buf;
    CK(hipMalloc(&block, 4096));
    CK(hipMalloc(&out, (steps + 64 + steps) * sizeof(int)));      // out[0..steps): after the memset; out[64..): before it
    CK(hipMalloc(&buf, (1 << 16) * sizeof(float)));
    CK(hipMemset(buf, 0, (1 << 16) * sizeof(float)));
    int ver = 0;
    CK(hipRuntimeGetVersion(&ver));
    printf("HIP runtime %d\n", ver);
    // (a) a 4-byte word 4 bytes past a 256-B boundary inside a larger allocation (a counter among counters)
    if (run_case("4-byte word at +260 of its allocation", block + 65, out, buf, steps, true)) return 1;
    // (b) the first word of the allocation
    if (run_case("4-byte word at +0 of its allocation  ", block, out, buf, steps, false)) return 1;
    return 0;
}
