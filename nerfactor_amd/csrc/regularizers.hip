// regularizers.hip — the small differentiable pieces of a NeRFactor training step that torch evaluated as chains of
// elementwise launches (round 5: 163 launches per 1024-ray microfacet step, ~120 of them 2-5 us kernels on a few KB —
// a quarter of the step's GPU time once it replays as one hipGraph).  One launch forward, one backward each:
//   l2_normalize_rows   tf.linalg.l2_normalize(x, axis=1, epsilon) = x * rsqrt(max(sum x^2, eps))   (util/math.py:63-64 of the
//                       reference; nerfactor.py:205-206 on the predicted normals, :266-270 on the BRDF codes) and its pull-back
//                       dx = inv dy - x inv^3 (x . dy) where sum x^2 >= eps (the max passes the gradient), dx = inv dy elsewhere;
//   light_smoothness    tv_w sum((L - roll(L, 1, 1))^2 + (L - roll(L, 1, 0))^2) + achro_w sum((L - roll(L, 1, 2))^2) over the
//                       [H, W, 3] light probe (nerfactor.py:526-539) together with its gradient
//                       2 w ((L - L[prev]) - (L[next] - L)) per axis — one block, fixed-order tree reduction (deterministic).
#include <hip/hip_runtime.h>

namespace nfx {

constexpr int kMaxNormCols = 16;

template <bool BWD>
__global__ __launch_bounds__(256) void l2_normalize_rows_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                float* __restrict__ out, long long n, int d, float eps) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v[kMaxNormCols], sq = 0.f;
#pragma unroll
    for (int k = 0; k < kMaxNormCols; ++k) {
        v[k] = k < d ? x[i * d + k] : 0.f;
        sq += v[k] * v[k];
    }
    const float inv = 1.0f / sqrtf(fmaxf(sq, eps));
    if constexpr (!BWD) {
#pragma unroll
        for (int k = 0; k < kMaxNormCols; ++k)
            if (k < d) out[i * d + k] = v[k] * inv;
    } else {
        float g[kMaxNormCols], dot = 0.f;
#pragma unroll
        for (int k = 0; k < kMaxNormCols; ++k) {
            g[k] = k < d ? dy[i * d + k] : 0.f;
            dot += v[k] * g[k];
        }
        const float c = sq >= eps ? dot * inv * inv * inv : 0.f;
#pragma unroll
        for (int k = 0; k < kMaxNormCols; ++k)
            if (k < d) out[i * d + k] = g[k] * inv - v[k] * c;
    }
}

// light[h][w][c]; one block of 256 threads strides over the H W 3 elements
__global__ __launch_bounds__(256) void light_smoothness_kernel(const float* __restrict__ light, int H, int W, float tv_w, float achro_w,
                                                               float* __restrict__ loss, float* __restrict__ grad) {
    __shared__ float part[256];
    const int n = H * W * 3;
    float s = 0.f;
    for (int e = threadIdx.x; e < n; e += 256) {
        const int c = e % 3, w = (e / 3) % W, h = e / (3 * W);
        const float v = light[e];
        const float pw = light[(h * W + (w + W - 1) % W) * 3 + c], nw = light[(h * W + (w + 1) % W) * 3 + c];
        const float ph = light[(((h + H - 1) % H) * W + w) * 3 + c], nh = light[(((h + 1) % H) * W + w) * 3 + c];
        const float pc = light[(h * W + w) * 3 + (c + 2) % 3], nc = light[(h * W + w) * 3 + (c + 1) % 3];
        const float dx = v - pw, dy = v - ph, dc = v - pc;
        s += tv_w * (dx * dx + dy * dy) + achro_w * (dc * dc);
        grad[e] = 2.f * (tv_w * ((dx - (nw - v)) + (dy - (nh - v))) + achro_w * (dc - (nc - v)));
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) part[threadIdx.x] += part[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = part[0];
}

}  // namespace nfx

extern "C" {
int nfx_launch_l2_normalize_rows(int bwd, const float* x, const float* dy, float* out, long long n, int d, float eps, hipStream_t st) {
    if (n <= 0) return 0;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (bwd) hipLaunchKernelGGL(nfx::l2_normalize_rows_kernel<true>, grid, dim3(256), 0, st, x, dy, out, n, d, eps);
    else hipLaunchKernelGGL(nfx::l2_normalize_rows_kernel<false>, grid, dim3(256), 0, st, x, dy, out, n, d, eps);
    return (int)hipGetLastError();
}
int nfx_launch_light_smoothness(const float* light, int H, int W, float tv_w, float achro_w, float* loss, float* grad, hipStream_t st) {
    hipLaunchKernelGGL(nfx::light_smoothness_kernel, dim3(1), dim3(256), 0, st, light, H, W, tv_w, achro_w, loss, grad);
    return (int)hipGetLastError();
}
}
