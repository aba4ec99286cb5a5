// Dev microbenchmark (round 5): can PART of H live in the 256 MiB Infinity Cache across the 12 conv applications of a
// step?  One "application" here = K1's and K2's traffic as plain streams: read H (248 MB; the first C bytes with
// default-policy loads, the rest non-temporal), write S (132 MB), read S back (132 MB), read W3R (17 MB, default).
// Swept: C, and the policy of the S stores / S loads.  If the default-policy part of H stays resident while everything
// else streams past it non-temporally, an application gets shorter as C grows (until the cache overflows).
// hipcc --offload-arch=gfx950 -O3 scripts/micro/mall_partial_residency.hip -o scripts/micro/mall_partial_residency
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// every wave reads per_wave16 16-B words; waves whose range starts below cached16 use default-policy loads
__global__ __launch_bounds__(256) void read_kernel(const f32x4* __restrict__ p, size_t n16, size_t cached16, float* out, size_t per_wave16) {
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const size_t base = wave * per_wave16;
    if (base >= n16) return;
    f32x4 acc = {0, 0, 0, 0};
    if (base < cached16) {
        for (size_t i = 0; i < per_wave16; i += 64 * 8) {
            f32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = p[base + i + j * 64 + lane];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j];
        }
    } else {
        for (size_t i = 0; i < per_wave16; i += 64 * 8) {
            f32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(p + base + i + j * 64 + lane);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <bool NT>
__global__ __launch_bounds__(256) void write_kernel(f32x4* __restrict__ p, size_t n16, size_t per_wave16, float val) {
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const size_t base = wave * per_wave16;
    if (base >= n16) return;
    const f32x4 v = {val, val, val, val};
    for (size_t i = 0; i < per_wave16; i += 64) {
        if (NT) __builtin_nontemporal_store(v, p + base + i + lane);
        else p[base + i + lane] = v;
    }
}

int main() {
    const size_t HB = 248ull * 1000 * 1000 / (256 * 1024) * (256 * 1024), SB = 132ull * 1000 * 1000 / (256 * 1024) * (256 * 1024),
                 WB = 16ull << 20;
    f32x4 *h, *s, *w; float* out;
    hipMalloc(&h, HB); hipMalloc(&s, SB); hipMalloc(&w, WB); hipMalloc(&out, 64);
    hipMemset(h, 0, HB); hipMemset(s, 0, SB); hipMemset(w, 0, WB);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const size_t pw16 = 256 * 1024 / 16;
    auto blocks = [&](size_t bytes) { return (int)((bytes / 16 / pw16 + 3) / 4); };
    const int reps = 36;      // three steps' worth of applications
    printf("H %.1f MB, S %.1f MB, W3R %.1f MB; us per application (H read | S write | S read | W read | all four)\n", HB / 1e6, SB / 1e6, WB / 1e6);
    for (int pol = 0; pol < 4; ++pol) {
        const bool s_store_nt = pol & 1, s_load_nt = pol & 2;
        for (size_t c_mib : {0ul, 64ul, 96ul, 128ul, 160ul, 192ul, 224ul, 240ul}) {
            const size_t cached16 = (c_mib << 20) / 16;
            auto app = [&](int phase_mask) {
                if (phase_mask & 1) read_kernel<<<blocks(HB), 256, 0, st>>>(h, HB / 16, cached16, out, pw16);
                if (phase_mask & 2) {
                    if (s_store_nt) write_kernel<true><<<blocks(SB), 256, 0, st>>>(s, SB / 16, pw16, 1.f);
                    else write_kernel<false><<<blocks(SB), 256, 0, st>>>(s, SB / 16, pw16, 1.f);
                }
                if (phase_mask & 4) read_kernel<<<blocks(SB), 256, 0, st>>>(s, SB / 16, s_load_nt ? 0 : SB / 16, out, pw16);
                if (phase_mask & 8) read_kernel<<<blocks(WB), 256, 0, st>>>(w, WB / 16, WB / 16, out, pw16);
            };
            for (int i = 0; i < 6; ++i) app(15);
            hipStreamSynchronize(st);
            // per-phase times inside the full sequence: events around each launch would add their own cost, so time
            // the whole application and, separately, the application with the H read timed by event pairs
            float ms, ms_h = 0.f;
            hipEventRecord(a, st);
            for (int i = 0; i < reps; ++i) app(15);
            hipEventRecord(b, st); hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
            for (int i = 0; i < reps; ++i) {
                hipEventRecord(a, st); app(1); hipEventRecord(b, st); app(14);
                hipEventSynchronize(b);
                float t; hipEventElapsedTime(&t, a, b); ms_h += t;
            }
            hipStreamSynchronize(st);
            printf("S stores %s, S loads %s, H cached %3zu MiB: application %.1f us, of which H read %.1f us (%.2f TB/s)\n",
                   s_store_nt ? "nt " : "def", s_load_nt ? "nt " : "def", c_mib, ms / reps * 1e3, ms_h / reps * 1e3, HB / (ms_h / reps) / 1e9);
        }
    }
    return 0;
}
