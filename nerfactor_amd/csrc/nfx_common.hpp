// nfx_common.hpp — shared device helpers for libnfx (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NFX_WAVE 64

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace nfx {

// ---------------------------------------------------------------------------------------
// sin / cos with a two-constant Cody–Waite reduction (valid for |y| up to ~1e5; the
// positional encoder reaches 2^9 * 6 ~ 3e3) and the Cephes single-precision minimax
// polynomials on [-pi/4, pi/4].  <= ~1.5 ulp; __sinf / v_sin_f32 would lose every digit at
// these arguments.  `shift` = 0 -> sin(y), 1 -> cos(y) (cos(y) = sin(y + pi/2): the
// quadrant index is shifted, not the argument, so no precision is lost).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float sin_shifted(float y, int shift) {
    const float n = rintf(y * 0.6366197466850281f);           // y * 2/pi
    float r = fmaf(n, -1.5707963705062866f, y);               // pi/2 hi
    r = fmaf(n, 4.371138828673793e-08f, r);                   // pi/2 lo
    const int q = (int)n + shift;
    const float r2 = r * r;
    float ps = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(ps, r2, -1.6666654611e-1f);
    ps = fmaf(ps * r2, r, r);                                  // sin(r)
    float pc = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(pc, r2, 4.166664568298827e-2f);
    pc = fmaf(pc * r2, r2, fmaf(r2, -0.5f, 1.0f));             // cos(r)
    float v = (q & 1) ? pc : ps;
    return (q & 2) ? -v : v;
}

// Both sin(y) and cos(y) from one Cody-Waite reduction (same polynomials as sin_shifted).
__device__ __forceinline__ void sincos_cw(float y, float& sn, float& cs) {
    const float n = rintf(y * 0.6366197466850281f);
    float r = fmaf(n, -1.5707963705062866f, y);
    r = fmaf(n, 4.371138828673793e-08f, r);
    const int q = (int)n;
    const float r2 = r * r;
    float ps = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(ps, r2, -1.6666654611e-1f);
    ps = fmaf(ps * r2, r, r);
    float pc = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(pc, r2, 4.166664568298827e-2f);
    pc = fmaf(pc * r2, r2, fmaf(r2, -0.5f, 1.0f));
    const float s0 = (q & 1) ? pc : ps, c0 = (q & 1) ? ps : pc;
    sn = (q & 2) ? -s0 : s0;
    cs = ((q + 1) & 2) ? -c0 : c0;
}

// Hardware sine/cosine (v_sin_f32 / v_cos_f32 take revolutions) for SMALL arguments: the 4-band encoders of unit
// vectors (|y| <= 8) and the 2-band encoder of Rusinkiewicz angles.  3 instructions instead of ~25.
__device__ __forceinline__ float sin_shifted_small(float y, int shift) {
    const float r = y * 0.15915494309189535f;   // 1 / (2 pi)
    return shift ? __builtin_amdgcn_cosf(r) : __builtin_amdgcn_sinf(r);
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float softplusf(float x) {
    // tf.nn.softplus = log(exp(x) + 1), evaluated stably
    return fmaxf(x, 0.0f) + log1pf(expf(-fabsf(x)));
}

// wave64 inclusive scan (product) and reductions via DPP-free shuffles.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Zero-fill of a 64-bit-word buffer as a KERNEL (a template so that the header can be included by several files).
// The fixed-point gradient accumulators (shade_bwd's light gradient, brdf_spec_bwd's d z / d normal) are cleared in
// front of the kernel that adds into them with integer atomics.  Round 2 cleared them with hipMemsetAsync: correct in
// stream order, but captured into a hipGraph it becomes a MEMSET NODE, and with it the replayed NeRFactor training
// step (the only steps with these accumulators) left the eager step at replay 94 of 200 — low bits of the loss at
// first, 0.22 against 0.065 at the end; NaN / 2e5 in round 2's benchmark — while the shape model, which has no such
// node, stayed bit-identical.  With this kernel in its place the eager and the replayed step agree bit for bit over 200
// steps for all models (tests/test_gpu_train.py::test_graphed_train_step_equals_the_eager_one,
// profiles/r03/graph_divergence/).  Kernel nodes are ordered like kernels.
template <int UNUSED = 0>
__global__ void zero_words_kernel(unsigned long long* p, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0ull;
}
inline void launch_zero_words(void* p, long long n_words, hipStream_t st) {
    if (n_words <= 0) return;
    hipLaunchKernelGGL(zero_words_kernel<0>, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, st,
                       static_cast<unsigned long long*>(p), n_words);
}

// bf16 pack helpers --------------------------------------------------------------------
__device__ __forceinline__ bf16x8 pack8(float a0, float a1, float a2, float a3, float a4, float a5,
                                        float a6, float a7) {
    bf16x8 r;
    r[0] = (__bf16)a0; r[1] = (__bf16)a1; r[2] = (__bf16)a2; r[3] = (__bf16)a3;
    r[4] = (__bf16)a4; r[5] = (__bf16)a5; r[6] = (__bf16)a6; r[7] = (__bf16)a7;
    return r;
}

}  // namespace nfx
