// Adam as train() runs it (graph_kernel.py:467 `optimizer.step()` on torch.optim.Adam(lr, weight_decay), :541-543), for ALL
// parameter tensors of the model in one launch: torch's fused implementation walks its tensor lists in chunks through
// `multi_tensor_apply` (52 us per step for the 27 tensors / 5.26 M parameters of the CLI model); this one takes the tensors'
// addresses as a kernel argument (no device-side lists, nothing uploaded), gives every workgroup one 4,096-element chunk
// of one tensor and streams p, g, m, v once: 7 x 4 B per parameter.
//   g' = g + weight_decay * p                       (L2 form, as torch.optim.Adam — not AdamW)
//   m  = m + (g' - m) * (1 - beta1)                 (torch's lerp)
//   v  = beta2 * v + (1 - beta2) * g' * g'
//   p  = p - (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// in this order of operations (torch/optim/adam.py, _single_tensor_adam / the fused kernel), fp32 throughout.
#include <cmath>

#include "kernels.h"

namespace mdno {
namespace {

constexpr int AD_MAX_TENSORS = 48;      // per launch (a model has 27); 48 x 4 pointers + the chunk table fit a kernel argument
constexpr int AD_CHUNK = 4096;          // elements per workgroup: 256 threads x 4 float4

struct AdamArgs {
    float* p[AD_MAX_TENSORS];
    const float* g[AD_MAX_TENSORS];
    float* m[AD_MAX_TENSORS];
    float* v[AD_MAX_TENSORS];
    long long n[AD_MAX_TENSORS];
    int chunk0[AD_MAX_TENSORS + 1];     // first workgroup of each tensor
    int count;
    // lr_c = lr / (1 - beta1^t); inv_sqrt_bc2 = 1 / sqrt(1 - beta2^t); om_beta = 1 - beta: all formed in double on the host and
    // rounded once (1.f - 0.999f is off by 1.3e-5 of its value: six steps later the parameters are off by 4e-5)
    float lr_c, beta2, om_beta1, om_beta2, eps, wd, inv_sqrt_bc2;
};

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamArgs& a) {
    g = fmaf(a.wd, p, g);
    m = m + (g - m) * a.om_beta1;
    v = a.beta2 * v + a.om_beta2 * g * g;
    const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
    p = p - a.lr_c * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(const AdamArgs a) {
    // which tensor: the last t with chunk0[t] <= blockIdx.x (a handful of tensors: linear scan, uniform per workgroup)
    int t = 0;
    while (t + 1 < a.count && a.chunk0[t + 1] <= (int)blockIdx.x) ++t;
    const long long base = (long long)((int)blockIdx.x - a.chunk0[t]) * AD_CHUNK;
    const long long n = a.n[t];
    float* __restrict__ p = a.p[t];
    const float* __restrict__ g = a.g[t];
    float* __restrict__ m = a.m[t];
    float* __restrict__ v = a.v[t];
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v)) & 15) == 0;
#pragma unroll
    for (int u = 0; u < AD_CHUNK / 1024; ++u) {
        const long long i = base + u * 1024 + threadIdx.x * 4;
        if (i >= n) return;
        if (vec && i + 4 <= n) {
            float4 pp = *reinterpret_cast<const float4*>(p + i), gg = *reinterpret_cast<const float4*>(g + i);
            float4 mm = *reinterpret_cast<const float4*>(m + i), vv = *reinterpret_cast<const float4*>(v + i);
            adam1(pp.x, gg.x, mm.x, vv.x, a); adam1(pp.y, gg.y, mm.y, vv.y, a);
            adam1(pp.z, gg.z, mm.z, vv.z, a); adam1(pp.w, gg.w, mm.w, vv.w, a);
            *reinterpret_cast<float4*>(p + i) = pp;
            *reinterpret_cast<float4*>(m + i) = mm;
            *reinterpret_cast<float4*>(v + i) = vv;
        } else {
            for (long long j = i; j < n && j < i + 4; ++j) {
                float pp = p[j], mm = m[j], vv = v[j];
                adam1(pp, g[j], mm, vv, a);
                p[j] = pp; m[j] = mm; v[j] = vv;
            }
        }
    }
}

}  // namespace
}  // namespace mdno

using namespace mdno;

extern "C" int mdno_adam_step(int count, const mdno_adam_tensor* tensors, double lr, double beta1, double beta2, double eps,
                              double weight_decay, int64_t step, void* stream) {
    MDNO_REQUIRE(count >= 0 && (count == 0 || tensors != nullptr), MDNO_EINVAL, "mdno_adam_step: bad arguments");
    MDNO_REQUIRE(step >= 1 && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0, MDNO_EINVAL,
                 "mdno_adam_step: step=%lld beta1=%g beta2=%g", (long long)step, beta1, beta2);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the bias corrections in double on the host, as torch computes them for a host-side step count
    const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2 = 1.0 - std::pow(beta2, (double)step);
    for (int first = 0; first < count; first += AD_MAX_TENSORS) {
        AdamArgs a{};
        int chunks = 0;
        for (int i = first; i < count && i < first + AD_MAX_TENSORS; ++i) {
            const mdno_adam_tensor& t = tensors[i];
            if (t.numel <= 0) continue;
            MDNO_REQUIRE(t.param && t.grad && t.exp_avg && t.exp_avg_sq, MDNO_EINVAL, "mdno_adam_step: null pointer in tensor %d", i);
            const int k = a.count++;
            a.p[k] = t.param; a.g[k] = t.grad; a.m[k] = t.exp_avg; a.v[k] = t.exp_avg_sq; a.n[k] = t.numel;
            a.chunk0[k] = chunks;
            chunks += (int)((t.numel + AD_CHUNK - 1) / AD_CHUNK);
        }
        if (a.count == 0) continue;
        a.chunk0[a.count] = chunks;
        a.lr_c = (float)(lr / bc1);
        a.beta2 = (float)beta2; a.om_beta1 = (float)(1.0 - beta1); a.om_beta2 = (float)(1.0 - beta2);
        a.eps = (float)eps; a.wd = (float)weight_decay;
        a.inv_sqrt_bc2 = (float)(1.0 / std::sqrt(bc2));
        hipLaunchKernelGGL(adam_kernel, dim3(chunks), dim3(256), 0, s, a);
    }
    return check_launch("adam_kernel");
}
