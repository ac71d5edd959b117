// nerf_mlp_x3.hip — NFX_PREC_FP32 for the fused NeRF MLP: fp32-class accuracy on the bf16 matrix pipe.
// Every operand is carried as a bf16 PAIR (hi = bf16(x), lo = bf16(x - hi): 16 significant bits) and every product is
// three MFMAs, a_hi b_hi + a_hi b_lo + a_lo b_hi, accumulated in fp32 (the dropped a_lo b_lo term is 2^-18 relative):
// weights, positional encoding and every re-quantised activation.  Same register-resident dataflow and fragment
// layout as the bf16 kernels; the blob is [hi fragments | lo fragments | biases] (nfx_nerf_pack_weights, NFX_PREC_FP32).
// 3 MFMAs per product at the bf16 rate is still ~5x the v_mfma_f32_32x32x2_f32 peak (157 TF).  Stated tolerance against
// the fp32 oracle: max |d rgb| <= 2e-4 (tests/test_gpu_nerf.py), the bound SURVEY.md §8d sets for an fp32 MFMA path.
#include "mlp_x3.hpp"
#include "nerf_layout.hpp"

namespace nfx {
namespace x3 {

constexpr int kLds = 2 * kSlot + nerf::kBiasFloats * 4;

__global__ __launch_bounds__(kNW * 64, 1) void nerf_mlp_x3_kernel(
    const float* __restrict__ rayo, const float* __restrict__ rayd, const float* __restrict__ zbuf, long long n_pts,
    int n_samples, const char* __restrict__ blob, float4* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using namespace nerf;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
    constexpr int kTilePts = kNW * 32;
    float* bias_lds = reinterpret_cast<float*>(smem + 2 * kSlot);
    {
        const float* bsrc = reinterpret_cast<const float*>(blob + 2 * (size_t)kWeightBytes);
        for (int i = tid; i < kBiasFloats; i += kNW * 64) bias_lds[i] = bsrc[i];
    }
    Stream st;
    st.base = reinterpret_cast<const u32x4*>(blob);
    st.end = reinterpret_cast<const u32x4*>(blob + kWeightBytes);
    st.ghi = st.base;
    st.lo_off = kWeightBytes / 16;
    st.ring = smem;
    prologue<kNL0>(st, tid);
    const long long n_tiles = (n_pts + kTilePts - 1) / kTilePts;
    for (long long tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
        const long long m = tl * kTilePts + wave * 32 + p;
        Pair pe[4], pv[2];
        {
            const long long mm = m < n_pts ? m : n_pts - 1;
            const long long ray = mm / n_samples;
            const float zz = zbuf[mm];
            float x[3], d[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                d[k] = rayd[ray * 3 + k];
                x[k] = rayo[ray * 3 + k] + d[k] * zz;
            }
            posenc_pair<10>(x, h, pe);
            posenc_pair<4>(d, h, pv);
        }
        Pair ha[16], hb[16], r0[8];
        layer<4, 0, 8, kNL0, kNLH, true>(st, tid, bias_lds + kBiasL0, pe, pe, ha);
        layer<16, 0, 8, kNLH, kNLH, true>(st, tid, bias_lds + kBiasL0 + 256 * 1, ha, pe, hb);
        layer<16, 0, 8, kNLH, kNLH, true>(st, tid, bias_lds + kBiasL0 + 256 * 2, hb, pe, ha);
        layer<16, 0, 8, kNLH, kNLH, true>(st, tid, bias_lds + kBiasL0 + 256 * 3, ha, pe, hb);
        layer<16, 0, 8, kNLH, kNL5, true>(st, tid, bias_lds + kBiasL0 + 256 * 4, hb, pe, ha);
        layer<16, 4, 8, kNL5, kNLH, true>(st, tid, bias_lds + kBiasL0 + 256 * 5, ha, pe, hb);
        layer<16, 0, 8, kNLH, kNLH, true>(st, tid, bias_lds + kBiasL0 + 256 * 6, hb, pe, ha);
        layer<16, 0, 8, kNLH, kNLH, true>(st, tid, bias_lds + kBiasL0 + 256 * 7, ha, pe, hb);
        layer<16, 0, 8, kNLH, kNLH, false>(st, tid, bias_lds + kBiasBott, hb, pe, ha);  // bottleneck
        float sigma;
        {
            f32x16 acc;
            tile<16, 0, kNLR0>(st, tid, bias_lds + kBiasBott + 256, hb, pe, acc);
            sigma = acc[0];
        }
        layer<16, 2, 4, kNLR0, kNLR1, true>(st, tid, bias_lds + kBiasRgb0, ha, pv, r0);
        {
            f32x16 acc;
            tile<8, 0, kNL0>(st, tid, bias_lds + kBiasRgb1, r0, pe, acc);
            if (h == 0 && m < n_pts) out[m] = make_float4(acc[0], acc[1], acc[2], sigma);
        }
    }
}

}  // namespace x3
}  // namespace nfx

extern "C" int nfx_launch_nerf_mlp_x3(const float* rayo, const float* rayd, const float* z, long long n_pts,
                                      int n_samples, const void* blob, float* out, int max_blocks, hipStream_t stream) {
    using namespace nfx;
    if (n_pts <= 0) return 0;
    const long long n_tiles = (n_pts + x3::kNW * 32 - 1) / (x3::kNW * 32);
    const int grid = (int)(n_tiles < max_blocks ? n_tiles : max_blocks);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(x3::nerf_mlp_x3_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, x3::kLds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(x3::nerf_mlp_x3_kernel, dim3(grid), dim3(x3::kNW * 64), x3::kLds, stream, rayo, rayd, z, n_pts,
                       n_samples, (const char*)blob, (float4*)out);
    return (int)hipGetLastError();
}
