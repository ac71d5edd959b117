// two_wave_valu_pair.hip — round 5 follow-up of two_wave_hazard.hip (profiles/HISTORY.md section 2c).
//
// brdf_compact_kernel<2, 0, 8> (per-row reference geometry, two waves per SIMD) returns wrong rows in whole groups of 16
// lanes; compiling the IEEE division sequences out removes 90 % of them.  Round 4's probe ran the suspect VALU sequences on
// ONE wave of a SIMD while its partner streamed MFMAs, and found nothing.  What the kernel does and that probe did not:
// BOTH waves of a SIMD execute the VALU sequences — v_div_scale and v_cmp write VCC / SGPRs from the VALU, v_div_fmas
// reads VCC behind a software wait-state hazard — at the same time, out of phase, each alternating with MFMA bursts whose
// B operands are the freshly converted VALU results.
//
// Set-up: one workgroup of 8 waves per CU (launch_bounds(512, 2): waves w and w + 4 share a SIMD).  Each half (waves 0-3,
// waves 4-7) runs a mode: 0 exit | 1 MFMA stream only | 2 VALU function loop | 3 VALU function + an MFMA burst on the
// converted results every iteration.  The lane checksums of a half running WITH a busy partner half must equal those of the
// same half running alone.  Half `lo` is given a phase offset (a few hundred idle cycles) so the two do not march in step.
//   f: 0 IEEE division   1 sqrtf   2 acosf + atan2f   3 IEEE division + compare / select chain (VCC written by v_cmp between
//      v_div_scale and v_div_fmas of neighbouring divisions)   4 normalisation by IEEE sqrt + division (the GEO = 0 row code)
// Build + run: hipcc --offload-arch=gfx950 -O3 two_wave_valu_pair.hip -o two_wave_valu_pair.bin && ./two_wave_valu_pair.bin
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
    if (F == 2) return acosf(fminf(fmaxf(x * 0.01f - 0.5f, -1.f), 1.f)) + atan2f(y, x - 37.f);
    if (F == 3) {
        const float a = x / y, b = y / (x + 3.f);
        const float c = a > b ? a / (b + 1.f) : b / (a + 1.f);
        return c > 1.f ? c / x : c / y;
    }
    const float n2 = x * x + y * y + 1.f;          // three components normalised the reference's way
    const float inv = 1.0f / sqrtf(fmaxf(n2, 1e-6f));
    return x * inv + y * inv * 0.5f + inv;
}

template <int F>
__global__ __launch_bounds__(512, 2) void k(unsigned* out, int iters, int mode_lo, int mode_hi, float* sink) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int half = wave >> 2, mode = half ? mode_hi : mode_lo;
    if (mode == 0) return;
    if (half == 0) __builtin_amdgcn_s_sleep(40);   // phase offset between the two waves of a SIMD
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (lane + j)); b[j] = (__bf16)(0.002f * (lane - j)); }
    f32x16 c0;
    for (int r = 0; r < 16; ++r) c0[r] = 0.f;
    if (mode == 1) {
        for (int i = 0; i < iters * 6; ++i) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        if (c0[0] == 12345.678f) sink[tid] = c0[1];
        return;
    }
    unsigned chk = 0;
    float x = 1.0f + 0.37f * lane + 0.011f * blockIdx.x + 0.003f * wave, y = 2.0f + 0.53f * (63 - lane);
    for (int i = 0; i < iters; ++i) {
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = fn<F>(x + 0.25f * j, y + 0.125f * j);
#pragma unroll
        for (int j = 0; j < 8; ++j) chk = (chk * 1664525u + 1013904223u) ^ __float_as_uint(r[j]);
        if (mode == 3) {                           // the results, converted, are the B operand of the next MFMAs
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (__bf16)r[j];
#pragma unroll
            for (int m = 0; m < 4; ++m) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, v, c0, 0, 0, 0);
        }
        x = x * 1.0009765625f + 0.125f;
        y = y * 0.9990234375f + 0.0625f;
        if (x > 90.f) x -= 88.f;
        if (y < 1.f) y += 70.f;
    }
    if (mode == 3) {                               // the MFMA results are a function of the (checked) VALU results only
#pragma unroll
        for (int r = 0; r < 16; ++r) chk = (chk * 1664525u + 1013904223u) ^ __float_as_uint(c0[r]);
    }
    out[(size_t)blockIdx.x * 512 + tid] = chk;
}

template <int F>
static void launch(unsigned* d, std::vector<unsigned>& h, int blocks, int iters, int lo, int hi, float* sink) {
    hipMemset(d, 0, h.size() * 4);
    hipLaunchKernelGGL(k<F>, dim3(blocks), dim3(512), 0, 0, d, iters, lo, hi, sink);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
}

template <int F>
static void run(const char* name, int mode, int blocks, int iters, int reps) {
    unsigned* d;
    float* sink;
    const size_t n = (size_t)blocks * 512;
    hipMalloc(&d, n * 4); hipMalloc(&sink, 512 * 4);
    std::vector<unsigned> lo_alone(n), hi_alone(n), both(n), again(n);
    launch<F>(d, lo_alone, blocks, iters, mode, 0, sink);
    launch<F>(d, hi_alone, blocks, iters, 0, mode, sink);
    long long bad_lanes = 0, bad_groups16 = 0, bad_runs = 0;
    for (int r = 0; r < reps; ++r) {
        launch<F>(d, both, blocks, iters, mode, mode, sink);
        long long bsum = 0;
        for (size_t i = 0; i < n; i += 16) {
            int g = 0;
            for (int j = 0; j < 16; ++j) {
                const size_t e = i + j;
                const bool lo_half = (e % 512) < 256;
                g += both[e] != (lo_half ? lo_alone[e] : hi_alone[e]);
            }
            bsum += g;
            bad_groups16 += g > 0;
        }
        bad_lanes += bsum;
        bad_runs += bsum > 0;
    }
    launch<F>(d, again, blocks, iters, mode, 0, sink);       // a lone half against itself: must always be 0
    long long self = 0;
    for (size_t i = 0; i < n; ++i) self += again[i] != lo_alone[i];
    printf("{\"f\": \"%s\", \"mode\": \"%s\", \"waves_per_simd\": 2, \"evaluations_per_lane\": %d, \"launches\": %d, "
           "\"launches_with_a_wrong_lane\": %lld, \"wrong_lane_checksums\": %lld, \"wrong_16_lane_groups\": %lld, "
           "\"lone_half_rerun_wrong_lanes\": %lld}\n", name, mode == 2 ? "VALU on both waves of every SIMD" :
           "VALU + MFMA bursts on both waves of every SIMD", iters * 8, reps, bad_runs, bad_lanes, bad_groups16, self);
    hipFree(d); hipFree(sink);
}

int main() {
    const int blocks = 256, iters = 4000, reps = 10;
    for (int mode = 2; mode <= 3; ++mode) {
        run<0>("IEEE division (v_div_scale, v_div_fmas, v_div_fixup)", mode, blocks, iters, reps);
        run<1>("sqrtf (IEEE)", mode, blocks, iters, reps);
        run<2>("acosf + atan2f (libm)", mode, blocks, iters, reps);
        run<3>("IEEE divisions with compare / select chains between them", mode, blocks, iters, reps);
        run<4>("normalisation by IEEE sqrt + division", mode, blocks, iters, reps);
    }
    return 0;
}
