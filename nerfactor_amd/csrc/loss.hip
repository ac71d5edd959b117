// loss.hip — the per-ray training losses of the surface models as ONE forward and ONE backward launch
// (reference: nerfactor/models/nerfactor.py:463-541, shape.py:239-277 compute_loss; util/img.py:alpha_blend;
// keras.losses.MSE / MAE = mean over the last axis).  A loss is a weighted sum of up to 8 terms
//     w_t * mean_d f( A_t[ray, d] - B_t[ray, d] ),   f = square | abs,
// where A / B are optionally alpha-blended onto the background first (x alpha + bg (1 - alpha), the reference's op
// order).  In torch these are ~70 elementwise launches forward and ~80 backward for a 1024-ray step whose GPU time is
// 3.6 ms in total; here one wave per ray walks the terms (D = 3 ... 512), reduces with shuffles and lane 0 writes the
// ray's loss; the backward recomputes the differences and writes (or, for a tensor that appears in two terms, adds
// to) d loss / d A and d loss / d B.  Deterministic: no atomics, fixed reduction order.
#include <hip/hip_runtime.h>

#include "../../include/nfx.h"

namespace nfx {

constexpr int kLossWaves = 4;

struct LossArgs {
    nfx_loss_term t[NFX_LOSS_MAX_TERMS];
    int n_terms;
    const float* alpha;   // [n] or null (= 1)
    float bg;
    long long n;
    float* loss;          // fwd: [n]
    const float* dloss;   // bwd: [n]
};

__device__ __forceinline__ float blend(float x, float al, float bg) { return x * al + bg * (1.0f - al); }

template <bool BWD>
__global__ __launch_bounds__(kLossWaves * 64) void pair_loss_kernel(LossArgs A) {
    const int lane = threadIdx.x & 63;
    const long long ray = (long long)blockIdx.x * kLossWaves + (threadIdx.x >> 6);
    if (ray >= A.n) return;
    const float al = A.alpha ? A.alpha[ray] : 1.0f;
    float total = 0.0f;
    const float up = BWD ? A.dloss[ray] : 0.0f;
    for (int ti = 0; ti < A.n_terms; ++ti) {
        const nfx_loss_term& t = A.t[ti];
        const float* a = t.a + ray * t.d;
        const float* b = t.b + ray * t.d;
        const bool ba = t.flags & NFX_LOSS_BLEND_A, bb = t.flags & NFX_LOSS_BLEND_B;
        if constexpr (!BWD) {
            float s = 0.0f;
            for (int d = lane; d < t.d; d += 64) {
                const float x = ba ? blend(a[d], al, A.bg) : a[d], y = bb ? blend(b[d], al, A.bg) : b[d];
                const float diff = x - y;
                s += t.kind == NFX_LOSS_MAE ? fabsf(diff) : diff * diff;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            total += t.w * (s / (float)t.d);
        } else {
            const float c = up * t.w / (float)t.d;
            float* ga = t.ga ? t.ga + ray * t.d : nullptr;
            float* gb = t.gb ? t.gb + ray * t.d : nullptr;
            for (int d = lane; d < t.d; d += 64) {
                const float x = ba ? blend(a[d], al, A.bg) : a[d], y = bb ? blend(b[d], al, A.bg) : b[d];
                const float diff = x - y;
                const float g = c * (t.kind == NFX_LOSS_MAE ? (float)((diff > 0.0f) - (diff < 0.0f)) : 2.0f * diff);
                if (ga) {
                    const float v = ba ? g * al : g;
                    ga[d] = (t.flags & NFX_LOSS_ACCUM_A) ? ga[d] + v : v;
                }
                if (gb) {
                    const float v = bb ? -g * al : -g;
                    gb[d] = (t.flags & NFX_LOSS_ACCUM_B) ? gb[d] + v : v;
                }
            }
        }
    }
    if constexpr (!BWD) {
        if (lane == 0) A.loss[ray] = total;
    }
}

}  // namespace nfx

extern "C" int nfx_launch_pair_loss(int bwd, const nfx_loss_term* terms, int n_terms, const float* alpha, float bg,
                                    long long n, float* loss, const float* dloss, hipStream_t st) {
    using namespace nfx;
    if (n <= 0) return 0;
    LossArgs A;
    for (int i = 0; i < n_terms; ++i) A.t[i] = terms[i];
    A.n_terms = n_terms;
    A.alpha = alpha;
    A.bg = bg;
    A.n = n;
    A.loss = loss;
    A.dloss = dloss;
    const int grid = (int)((n + kLossWaves - 1) / kLossWaves);
    if (bwd) hipLaunchKernelGGL(pair_loss_kernel<true>, dim3(grid), dim3(kLossWaves * 64), 0, st, A);
    else hipLaunchKernelGGL(pair_loss_kernel<false>, dim3(grid), dim3(kLossWaves * 64), 0, st, A);
    return (int)hipGetLastError();
}
