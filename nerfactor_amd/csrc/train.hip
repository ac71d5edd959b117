// train.hip — training-side kernels that are not part of an MLP chain:
//   amsgrad_step   tf.keras.optimizers.Adam(amsgrad=True) dense update          trainvali.py:110-127
//   wgrad_bf16     dW[K,N] += X[rows,K]^T dZ[rows,N] from feature-major bf16    (tape.gradient, trainvali.py:284)
//                  (+ db[N] += sum_rows dZ[rows,N], folded into the same pass)
// Activations / pre-activation gradients arrive FEATURE-MAJOR ([feature][row], bf16) from the fused
// backward kernels, so an MFMA operand fragment (one feature x 8 consecutive rows) is one 16-byte
// global load — no LDS transposition anywhere.
#include "nfx_common.hpp"

namespace nfx {

// Keras OptimizerV2 Adam._resource_apply_dense with amsgrad (TF 2.2):
//   lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t);  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//   vhat = max(vhat, v);  p -= lr_t * m / (sqrt(vhat) + eps)            (eps = 1e-7)
__global__ void amsgrad_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                               float* __restrict__ v, float* __restrict__ vhat, long long n, float lr_t,
                               float b1, float b2, float eps) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = m[i] * b1 + gi * (1.0f - b1);
    const float vi = v[i] * b2 + (gi * gi) * (1.0f - b2);
    const float vh = fmaxf(vhat[i], vi);
    m[i] = mi;
    v[i] = vi;
    vhat[i] = vh;
    p[i] = p[i] - lr_t * mi / (sqrtf(vh) + eps);
}

// One wave owns a [<=128 x <=128] block of dW (grid y, z) and a slab of rows (grid x).
//   xt: [k_in][ld] bf16 (feature-major), zt: [n_out][ld] bf16; rows in [row0, row1), multiple of 16.
//   dW: [k_in][n_out] fp32 (Keras layout), accumulated with atomics.
constexpr int kWgTiles = 4;  // 4 x 4 tiles of 32 x 32
__global__ __launch_bounds__(64, 1) void wgrad_kernel(const __bf16* __restrict__ xt,
                                                      const __bf16* __restrict__ zt, long long ld,
                                                      int k_in, int n_out, long long rows,
                                                      long long slab, float* __restrict__ dw,
                                                      float* __restrict__ db) {
    const int lane = threadIdx.x, h = lane >> 5, q = lane & 31;
    const int kb = blockIdx.y * (32 * kWgTiles);  // first input feature of this block
    const int nb = blockIdx.z * (32 * kWgTiles);  // first output feature of this block
    const long long r0 = (long long)blockIdx.x * slab;
    long long r1 = r0 + slab;
    if (r1 > rows) r1 = rows;
    f32x16 acc[kWgTiles][kWgTiles];
#pragma unroll
    for (int i = 0; i < kWgTiles; ++i)
#pragma unroll
        for (int j = 0; j < kWgTiles; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool do_bias = db != nullptr && blockIdx.y == 0;  // db[f] += sum_rows dZ[f][row], once per slab
    float bsum[kWgTiles] = {0.f, 0.f, 0.f, 0.f};
    for (long long k0 = r0; k0 < r1; k0 += 16) {
        bf16x8 a[kWgTiles], b[kWgTiles];
#pragma unroll
        for (int i = 0; i < kWgTiles; ++i) {
            const int f = kb + 32 * i + q;
            a[i] = f < k_in ? *reinterpret_cast<const bf16x8*>(xt + (long long)f * ld + k0 + 8 * h) : zero;
        }
#pragma unroll
        for (int j = 0; j < kWgTiles; ++j) {
            const int f = nb + 32 * j + q;
            b[j] = f < n_out ? *reinterpret_cast<const bf16x8*>(zt + (long long)f * ld + k0 + 8 * h) : zero;
        }
        if (do_bias) {
#pragma unroll
            for (int j = 0; j < kWgTiles; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum[j] += (float)b[j][e];
        }
#pragma unroll
        for (int i = 0; i < kWgTiles; ++i)
#pragma unroll
            for (int j = 0; j < kWgTiles; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (do_bias) {
#pragma unroll
        for (int j = 0; j < kWgTiles; ++j) {
            const float s = bsum[j] + __shfl_xor(bsum[j], 32, 64);  // the two row halves of a k-step
            const int col = nb + 32 * j + q;
            if (h == 0 && col < n_out) atomicAdd(db + col, s);
        }
    }
#pragma unroll
    for (int i = 0; i < kWgTiles; ++i)
#pragma unroll
        for (int j = 0; j < kWgTiles; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = kb + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int col = nb + 32 * j + q;
                if (row < k_in && col < n_out) atomicAdd(dw + (long long)row * n_out + col, acc[i][j][r]);
            }
}

}  // namespace nfx

extern "C" {
int nfx_launch_amsgrad(float* p, const float* g, float* m, float* v, float* vhat, long long n, float lr_t,
                       float b1, float b2, float eps, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(nfx::amsgrad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, g, m, v,
                       vhat, n, lr_t, b1, b2, eps);
    return (int)hipGetLastError();
}
int nfx_launch_wgrad(const void* xt, const void* zt, long long ld, int k_in, int n_out, long long rows,
                     float* dw, float* db, hipStream_t st) {
    if (rows <= 0) return 0;
    long long slab = 1024;
    const unsigned gx = (unsigned)((rows + slab - 1) / slab);
    const unsigned gy = (unsigned)((k_in + 127) / 128);
    const unsigned gz = (unsigned)((n_out + 127) / 128);
    hipLaunchKernelGGL(nfx::wgrad_kernel, dim3(gx, gy, gz), dim3(64), 0, st, (const __bf16*)xt, (const __bf16*)zt, ld,
                       k_in, n_out, rows, slab, dw, db);
    return (int)hipGetLastError();
}
}
