// Internal helpers shared by the libmdno translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/mdno.h"

namespace mdno {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return MDNO_ELAUNCH;
    }
    return MDNO_OK;
}

#define MDNO_REQUIRE(cond, code, ...)          \
    do {                                       \
        if (!(cond)) {                         \
            ::mdno::set_error(__VA_ARGS__);    \
            return (code);                     \
        }                                      \
    } while (0)

#define MDNO_HIP(call)                                                           \
    do {                                                                         \
        hipError_t e_ = (call);                                                  \
        if (e_ != hipSuccess) {                                                  \
            ::mdno::set_error("%s: %s", #call, hipGetErrorString(e_));           \
            return MDNO_ELAUNCH;                                                 \
        }                                                                        \
    } while (0)

#define MDNO_TRY(call)                 \
    do {                               \
        int rc_ = (call);              \
        if (rc_ != MDNO_OK) return rc_; \
    } while (0)

constexpr int kWave = 64;  // CDNA4 wavefront

// ReLU as torch.relu has it: a NaN stays a NaN (fmaxf(NaN, 0) — v_max_f32 — returns 0: a non-finite weight or activation
// would be turned into a finite zero at the next ReLU and never reach the output, where the reference shows it), -0 -> +0
__device__ __forceinline__ float relu_f(float v) { return v <= 0.f ? 0.f : v; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: raise it once per
// (call site, device).  `done` is that call site's device bitmask — a cache of "already raised", not
// state a caller can observe; two threads racing on a first use both set the (idempotent) attribute.
inline int raise_dynamic_lds(const void* kernel, int bytes, std::atomic<unsigned long long>& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        set_error("hipGetDevice: %s", hipGetErrorString(e));
        return MDNO_ELAUNCH;
    }
    const bool cached = dev >= 0 && dev < 64;
    const unsigned long long bit = cached ? 1ull << dev : 0ull;
    if (cached && (done.load(std::memory_order_acquire) & bit)) return MDNO_OK;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d): %s", bytes, hipGetErrorString(e));
        return MDNO_ELAUNCH;
    }
    if (cached) done.fetch_or(bit, std::memory_order_release);
    return MDNO_OK;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Bump allocator over the caller's workspace (256-B aligned carves).
struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* p) : base(static_cast<char*>(p)) {}
    template <class T>
    T* take(size_t count) {
        off = align_up(off, 256);
        T* r = reinterpret_cast<T*>(base ? base + off : nullptr);
        off += count * sizeof(T);
        return r;
    }
    size_t used() const { return align_up(off, 256); }
};

}  // namespace mdno
