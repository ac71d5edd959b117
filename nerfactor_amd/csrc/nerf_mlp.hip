// nerf_mlp.hip — fused point generation + positional encoding + NeRF MLP (bf16 MFMA).
// Replaces Model._eval_nerf_at (nerfactor/models/nerf.py:256-290) together with the point
// generation of nerf.py:162-164 / 175-177 and Embedder.__call__ (networks/embedder.py:46-47).
#include "mlp_engine.hpp"
#include "nerf_layout.hpp"

namespace nfx {

// LDS: [2 weight slots][biases]
constexpr int kNerfLdsBytes = 2 * kSlotBytes + nerf::kBiasFloats * 4;

template <int CT, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void nerf_mlp_bf16_kernel(
    const float* __restrict__ rayo, const float* __restrict__ rayd, const float* __restrict__ zbuf,
    long long n_pts, int n_samples, const char* __restrict__ blob, float4* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, p = lane & 31;
    constexpr int kTilePts = NW * 32 * CT;
    constexpr int kWgThreads = NW * 64;

    float* bias_lds = reinterpret_cast<float*>(smem + 2 * kSlotBytes);
    {
        const float* bsrc = reinterpret_cast<const float*>(blob + nerf::kWeightBytes);
        for (int i = tid; i < nerf::kBiasFloats; i += kWgThreads) bias_lds[i] = bsrc[i];
    }
    WStream ws;
    ws.gbase = reinterpret_cast<const u32x4*>(blob);
    ws.gend = reinterpret_cast<const u32x4*>(blob + nerf::kWeightBytes);
    ws.gnext = ws.gbase;
    ws.ring = smem;
    stream_prologue<nerf::kNL0, NW>(ws, tid);  // also orders the bias copy before first use

    const long long n_tiles = (n_pts + kTilePts - 1) / kTilePts;
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        // ---- inputs: pts = rayo + rayd * z (nerf.py:162-163), views = rayd (nerf.py:164)
        bf16x8 pe[4][CT], pv[2][CT];
        long long m[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            m[c] = tile * kTilePts + wave * (32 * CT) + c * 32 + p;
            const long long mm = m[c] < n_pts ? m[c] : n_pts - 1;
            const long long ray = mm / n_samples;
            const float zz = zbuf[mm];
            float x[3], d[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                d[k] = rayd[ray * 3 + k];
                x[k] = rayo[ray * 3 + k] + d[k] * zz;
            }
            posenc<10, CT>(x, h, c, pe);
            posenc<4, CT>(d, h, c, pv);
        }

        bf16x8 ha[16][CT], hb[16][CT];
        using namespace nerf;
        // enc layers 0..7 (nerf.py:57-59): relu, skip-concat (y, x) feeds layer 5
        layer<4, 0, 8, kNL0, kNLH, true, NW>(ws, tid, bias_lds + kBiasL0, pe, pe, ha);
        layer<16, 0, 8, kNLH, kNLH, true, NW>(ws, tid, bias_lds + kBiasL0 + 256 * 1, ha, pe, hb);
        layer<16, 0, 8, kNLH, kNLH, true, NW>(ws, tid, bias_lds + kBiasL0 + 256 * 2, hb, pe, ha);
        layer<16, 0, 8, kNLH, kNLH, true, NW>(ws, tid, bias_lds + kBiasL0 + 256 * 3, ha, pe, hb);
        layer<16, 0, 8, kNLH, kNL5, true, NW>(ws, tid, bias_lds + kBiasL0 + 256 * 4, hb, pe, ha);
        layer<16, 4, 8, kNL5, kNLH, true, NW>(ws, tid, bias_lds + kBiasL0 + 256 * 5, ha, pe, hb);
        layer<16, 0, 8, kNLH, kNLH, true, NW>(ws, tid, bias_lds + kBiasL0 + 256 * 6, hb, pe, ha);
        layer<16, 0, 8, kNLH, kNLH, true, NW>(ws, tid, bias_lds + kBiasL0 + 256 * 7, ha, pe, hb);
        // bottleneck (nerf.py:68, no activation) — rows 0..255 of the fused [bottleneck | sigma_out]
        layer<16, 0, 8, kNLH, kNLH, false, NW>(ws, tid, bias_lds + kBiasBott, hb, pe, ha);
        // sigma_out (nerf.py:67) — row 256 of the fused matrix = row 0 of tile 8
        float sigma[CT];
        {
            f32x16 acc[CT];
            tile_raw<16, 0, kNLR0, NW>(ws, tid, bias_lds + kBiasBott + 256, hb, pe, acc);
#pragma unroll
            for (int c = 0; c < CT; ++c) sigma[c] = acc[c][0];
        }
        // rgb_out[0]: Dense(128, relu) on concat(bottleneck, posenc(view)) (nerf.py:280-281)
        bf16x8 r0[8][CT];
        layer<16, 2, 4, kNLR0, kNLR1, true, NW>(ws, tid, bias_lds + kBiasRgb0, ha, pv, r0);
        // rgb_out[1]: Dense(3)
        {
            f32x16 acc[CT];
            tile_raw<8, 0, kNL0, NW>(ws, tid, bias_lds + kBiasRgb1, r0, pe, acc);
            if (h == 0) {
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    if (m[c] < n_pts)
                        out[m[c]] = make_float4(acc[c][0], acc[c][1], acc[c][2], sigma[c]);
            }
        }
    }
}


// Variant 4: variant 1 (8 waves x 32 points, register-staged weights) with the six identical
// 256->256 ReLU layers (enc[1..4], enc[6..7]) executed by ONE rolled two-layer loop body, to test
// whether the ~55-110 KiB of straight-line code of the other variants (vs a 64 KiB instruction
// cache shared by two CUs) is what holds them near 48 % of MFMA peak.  enc[5] sits inside the loop
// (third trip) followed by a register copy hb -> ha so the body's buffer roles repeat.
__global__ __launch_bounds__(512, 2) void nerf_mlp_bf16_rolled_kernel(
    const float* __restrict__ rayo, const float* __restrict__ rayd, const float* __restrict__ zbuf,
    long long n_pts, int n_samples, const char* __restrict__ blob, float4* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = 8, CT = 1;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, p = lane & 31;
    constexpr int kTilePts = NW * 32;
    float* bias_lds = reinterpret_cast<float*>(smem + 2 * kSlotBytes);
    {
        const float* bsrc = reinterpret_cast<const float*>(blob + nerf::kWeightBytes);
        for (int i = tid; i < nerf::kBiasFloats; i += NW * 64) bias_lds[i] = bsrc[i];
    }
    WStream ws;
    ws.gbase = reinterpret_cast<const u32x4*>(blob);
    ws.gend = reinterpret_cast<const u32x4*>(blob + nerf::kWeightBytes);
    ws.gnext = ws.gbase;
    ws.ring = smem;
    stream_prologue<nerf::kNL0, NW>(ws, tid);
    const long long n_tiles = (n_pts + kTilePts - 1) / kTilePts;
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        bf16x8 pe[4][CT], pv[2][CT];
        const long long m = tile * kTilePts + wave * 32 + p;
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
            posenc<10, CT>(x, h, 0, pe);
            posenc<4, CT>(d, h, 0, pv);
        }
        bf16x8 ha[16][CT], hb[16][CT];
        using namespace nerf;
        layer<4, 0, 8, kNL0, kNLH, true, NW>(ws, tid, bias_lds + kBiasL0, pe, pe, ha);
        const float* bias = bias_lds + kBiasL0 + 256;
#pragma unroll 1
        for (int it = 0; it < 3; ++it) {
            if (it == 2) {  // enc[5]: [y, posenc(x)] -> hb, then hb -> ha so the roles below repeat
                layer<16, 4, 8, kNL5, kNLH, true, NW>(ws, tid, bias, ha, pe, hb);
                bias += 256;
#pragma unroll
                for (int s = 0; s < 16; ++s) ha[s][0] = hb[s][0];
            }
            // (it, layer): (0: enc1, enc2) (1: enc3, enc4 -> next chunk is enc5's) (2: enc6, enc7)
            layer_rt<16, 0, 8, kNL5, true, NW>(ws, tid, bias, kNLH, kNLH, ha, pe, hb);
            layer_rt<16, 0, 8, kNL5, true, NW>(ws, tid, bias + 256, kNLH, it == 1 ? kNL5 : kNLH, hb, pe, ha);
            bias += 512;
        }
        // after the loop the encoder output is in ha
        layer<16, 0, 8, kNLH, kNLH, false, NW>(ws, tid, bias_lds + kBiasBott, ha, pe, hb);  // bottleneck -> hb
        float sigma;
        {
            f32x16 acc[CT];
            tile_raw<16, 0, kNLR0, NW>(ws, tid, bias_lds + kBiasBott + 256, ha, pe, acc);
            sigma = acc[0][0];
        }
        bf16x8 r0[8][CT];
        layer<16, 2, 4, kNLR0, kNLR1, true, NW>(ws, tid, bias_lds + kBiasRgb0, hb, pv, r0);
        {
            f32x16 acc[CT];
            tile_raw<8, 0, kNL0, NW>(ws, tid, bias_lds + kBiasRgb1, r0, pe, acc);
            if (h == 0 && m < n_pts) out[m] = make_float4(acc[0][0], acc[0][1], acc[0][2], sigma);
        }
    }
}

}  // namespace nfx

template <int CT, int NW>
static int launch_variant(const float* rayo, const float* rayd, const float* z, long long n_pts,
                          int n_samples, const void* blob, float* out, int max_blocks,
                          hipStream_t stream) {
    using namespace nfx;
    const int tile_pts = NW * 32 * CT;
    const long long n_tiles = (n_pts + tile_pts - 1) / tile_pts;
    const int grid = (int)(n_tiles < max_blocks ? n_tiles : max_blocks);
    auto kern = nerf_mlp_bf16_kernel<CT, NW>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kNerfLdsBytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), kNerfLdsBytes, stream, rayo, rayd, z,
                       n_pts, n_samples, (const char*)blob, (float4*)out);
    return (int)hipGetLastError();
}

// variant: 0 = 4 waves x 64 points (CT=2, one wave per SIMD), 1 = 8 waves x 32 points (CT=1,
// two waves per SIMD).  Returns a hipError_t.
extern "C" int nfx_launch_nerf_mlp_bf16(const float* rayo, const float* rayd, const float* z,
                                        long long n_pts, int n_samples, const void* blob,
                                        float* out, int variant, int max_blocks,
                                        hipStream_t stream) {
    if (n_pts <= 0) return 0;
    if (variant == 4) {
        using namespace nfx;
        const long long n_tiles = (n_pts + 255) / 256;
        const int grid = (int)(n_tiles < max_blocks ? n_tiles : max_blocks);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(nerf_mlp_bf16_rolled_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kNerfLdsBytes);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(nerf_mlp_bf16_rolled_kernel, dim3(grid), dim3(512), kNerfLdsBytes, stream, rayo,
                           rayd, z, n_pts, n_samples, (const char*)blob, (float4*)out);
        return (int)hipGetLastError();
    }
    if (variant == 0)
        return launch_variant<2, 4>(rayo, rayd, z, n_pts, n_samples, blob, out, max_blocks, stream);
    return launch_variant<1, 8>(rayo, rayd, z, n_pts, n_samples, blob, out, max_blocks, stream);
}
