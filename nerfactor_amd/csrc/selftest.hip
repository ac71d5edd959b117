// selftest.hip — tiny diagnostic kernels that pin hardware conventions the fused kernels rely on.
#include "nfx_common.hpp"
#include "tr16.hpp"

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
// D[32][32] = H^T Z for two ROW-MAJOR [16 rows][32 slots] tiles: rows -> LDS as mlp128_bwd_fused.hip stores them
// (pitch 288 bytes), operands through tr_frag (ds_read_b64_tr_b16), one MFMA whose K axis is the row axis.
// mode 1: raw dump instead — every lane reads 4 bf16 at LDS byte address 8 * lane of a tile holding its own element
// index, D[lane * 4 + j] = what came back (documents the instruction's lane map).
__global__ void selftest_tr16_kernel(const float* H, const float* Z, float* D, int mode) {
    __shared__ __attribute__((aligned(16))) char lds[2 * 16 * 288];
    constexpr int P = 288;
    const int l = threadIdx.x;
    if (mode == 1) {
        short* e = reinterpret_cast<short*>(lds);
        for (int i = l; i < 1024; i += 64) e[i] = (short)i;
        __syncthreads();
        const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + 8 * l));
        for (int j = 0; j < 4; ++j) D[l * 4 + j] = (float)v[j];
        return;
    }
    for (int i = l; i < 16 * 32; i += 64) {
        const int row = i >> 5, slot = i & 31;
        *reinterpret_cast<__bf16*>(lds + row * P + slot * 2) = (__bf16)H[i];
        *reinterpret_cast<__bf16*>(lds + 16 * P + row * P + slot * 2) = (__bf16)Z[i];
    }
    __syncthreads();
    const int off = tr_lane_off(l, P);
    const bf16x8 a = tr_frag<P>(lds + off), b = tr_frag<P>(lds + 16 * P + off);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
}  // namespace nfx

extern "C" {
int nfx_launch_selftest_tr16(const float* h, const float* z, float* d, int mode, hipStream_t st) {
    hipLaunchKernelGGL(nfx::selftest_tr16_kernel, dim3(1), dim3(64), 0, st, h, z, d, mode);
    return (int)hipGetLastError();
}
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
