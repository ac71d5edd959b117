// selftest.hip — tiny diagnostic kernels that pin hardware conventions the fused kernels rely on.
#include "nfx_common.hpp"

namespace nfx {
// D[32][32] = A[32][16] * B[16][32] with the documented lane maps of v_mfma_f32_32x32x16_bf16:
//   A: lane l holds A[l&31][8*(l>>5) + j], j = 0..7
//   B: lane l holds B[8*(l>>5) + j][l&31]
//   D: lane l, reg r holds D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31]
__global__ void selftest_mfma_kernel(const float* A, const float* B, float* D) {
    const int l = threadIdx.x, h = l >> 5, n = l & 31;
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)A[n * 16 + 8 * h + j];
        b[j] = (__bf16)B[(8 * h + j) * 32 + n];
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + n] = acc[r];
}
__global__ void selftest_sincos_kernel(const float* in, long long n, int which, float* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = which < 2 ? sin_shifted(in[i], which) : sin_shifted_small(in[i], which - 2);   // 2, 3: v_sin / v_cos
}
}  // namespace nfx

extern "C" {
int nfx_launch_selftest_mfma(const float* a, const float* b, float* d, hipStream_t st) {
    hipLaunchKernelGGL(nfx::selftest_mfma_kernel, dim3(1), dim3(64), 0, st, a, b, d);
    return (int)hipGetLastError();
}
int nfx_launch_selftest_sincos(const float* in, long long n, int which, float* out, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(nfx::selftest_sincos_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       in, n, which, out);
    return (int)hipGetLastError();
}
}
