// two_wave_hazard.hip — do plain VALU instruction sequences stay correct when a partner wave on the SAME SIMD streams
// MFMAs?  Background: brdf_compact_kernel with two waves per SIMD gives wrong rows in groups of 16 lanes (profiles/HISTORY.md
// section 2c): r03 traced one cause to v_permlane32_swap (a software wait-state hazard), r04's soak shows the per-row
// geometry form still failing — 10x less with IEEE division compiled out.
// Set-up: one workgroup of 8 waves per CU (launch_bounds(512, 2): waves w and w + 4 share a SIMD).  Waves 4-7 evaluate
// f(x) for a per-lane sequence and fold the result bits into a checksum; waves 0-3 either spin on MFMAs (partner = 1)
// or exit at once (partner = 0).  The checksums of the two runs must be equal lane for lane: any difference is a VALU
// result that depended on what the neighbour wave was doing.
//   f: 0 IEEE division (v_div_scale / v_div_fmas / v_div_fixup)   1 sqrtf   2 v_sin_f32 + v_cos_f32   3 acosf + atan2f
//      4 v_rsq_f32 normalisation (what the closed-form geometry uses)   5 1 / sqrtf (IEEE sqrt + IEEE division)
// Build + run: hipcc --offload-arch=gfx950 -O3 two_wave_hazard.hip -o two_wave_hazard.bin && ./two_wave_hazard.bin
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int F>
__device__ __forceinline__ float fn(float x, float y) {
    if (F == 0) return x / y;
    if (F == 1) return sqrtf(x);
    if (F == 2) return __builtin_amdgcn_sinf(x * 0.15915494f) + __builtin_amdgcn_cosf(y * 0.15915494f);
    if (F == 3) return acosf(fminf(fmaxf(x * 0.01f - 0.5f, -1.f), 1.f)) + atan2f(y, x - 37.f);
    if (F == 4) return x * __builtin_amdgcn_rsqf(fmaxf(x * x + y * y, 1e-6f));
    return 1.0f / sqrtf(x + y);
}

template <int F>
__global__ __launch_bounds__(512, 2) void k(unsigned* out, int iters, int partner, float* sink) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (wave < 4) {
        if (!partner) return;
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (lane + j)); b[j] = (__bf16)(0.002f * (lane - j)); }
        f32x16 c0, c1;
        for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
        for (int i = 0; i < iters * 6; ++i) {      // long enough to cover the test waves' whole loop
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
        }
        if (c0[0] + c1[3] == 12345.678f) sink[tid] = c0[1];
        return;
    }
    unsigned chk = 0;
    float x = 1.0f + 0.37f * lane + 0.011f * blockIdx.x, y = 2.0f + 0.53f * (63 - lane);
    for (int i = 0; i < iters; ++i) {
        const float r = fn<F>(x, y);
        chk = (chk * 1664525u + 1013904223u) ^ __float_as_uint(r);
        x = x * 1.0009765625f + 0.125f;            // exact-ish, data dependent sequence
        y = y * 0.9990234375f + 0.0625f;
        if (x > 90.f) x -= 88.f;
        if (y < 1.f) y += 70.f;
    }
    out[blockIdx.x * 256 + (wave - 4) * 64 + lane] = chk;
}

template <int F>
static void run(const char* name, int blocks, int iters, int reps) {
    unsigned *d0, *d1;
    float* sink;
    const size_t n = (size_t)blocks * 256;
    hipMalloc(&d0, n * 4); hipMalloc(&d1, n * 4); hipMalloc(&sink, 512 * 4);
    std::vector<unsigned> h0(n), h1(n);
    hipMemset(d0, 0, n * 4);
    hipLaunchKernelGGL(k<F>, dim3(blocks), dim3(512), 0, 0, d0, iters, 0, sink);
    hipMemcpy(h0.data(), d0, n * 4, hipMemcpyDeviceToHost);
    long long bad_lanes = 0, bad_groups16 = 0, bad_runs = 0;
    for (int r = 0; r < reps; ++r) {
        hipMemset(d1, 0, n * 4);
        hipLaunchKernelGGL(k<F>, dim3(blocks), dim3(512), 0, 0, d1, iters, 1, sink);
        hipMemcpy(h1.data(), d1, n * 4, hipMemcpyDeviceToHost);
        long long b = 0;
        for (size_t i = 0; i < n; i += 16) {
            int g = 0;
            for (int j = 0; j < 16; ++j) g += h0[i + j] != h1[i + j];
            b += g;
            bad_groups16 += g > 0;
        }
        bad_lanes += b;
        bad_runs += b > 0;
    }
    // and the idle-partner run against itself (must always be 0)
    hipMemset(d1, 0, n * 4);
    hipLaunchKernelGGL(k<F>, dim3(blocks), dim3(512), 0, 0, d1, iters, 0, sink);
    hipMemcpy(h1.data(), d1, n * 4, hipMemcpyDeviceToHost);
    long long self = 0;
    for (size_t i = 0; i < n; ++i) self += h0[i] != h1[i];
    printf("{\"f\": \"%s\", \"test_waves\": %d, \"evaluations_per_lane\": %d, \"launches_with_mfma_partner\": %d, "
           "\"launches_with_a_wrong_lane\": %lld, \"wrong_lane_checksums\": %lld, \"wrong_16_lane_groups\": %lld, "
           "\"idle_partner_rerun_wrong_lanes\": %lld}\n", name, blocks * 4, iters, reps, bad_runs, bad_lanes, bad_groups16, self);
    hipFree(d0); hipFree(d1); hipFree(sink);
}

int main() {
    const int blocks = 256, iters = 20000, reps = 10;
    run<0>("IEEE division (v_div_scale, v_div_fmas, v_div_fixup)", blocks, iters, reps);
    run<1>("sqrtf (IEEE)", blocks, iters, reps);
    run<2>("v_sin_f32 + v_cos_f32", blocks, iters, reps);
    run<3>("acosf + atan2f (libm)", blocks, iters, reps);
    run<4>("v_rsq_f32 normalisation", blocks, iters, reps);
    run<5>("1 / sqrtf (IEEE sqrt, IEEE division)", blocks, iters, reps);
    return 0;
}
