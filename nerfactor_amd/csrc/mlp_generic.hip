// mlp_generic.hip — the network shapes the tuned kernels do NOT cover (round 4): any mlp.Network the reference can build
// (nerfactor/networks/mlp.py:24-50: widths, activations, skip_at anywhere; nerfactor/models/nerf.py:53-90: mlp_width,
// enc_depth, use_views = False, pos_enc = False) evaluated by ONE runtime-shaped fused kernel, forward only.
//
// The tuned kernels (nerf_mlp_v6.hip, mlp128.hip, lvis_v2.hip) are compile-time specialisations of one architecture
// each — register-resident activations, packed weight streams.  This one trades their speed for generality and keeps
// their arithmetic class: bf16 operands, fp32 accumulation and bias, one v_mfma_f32_32x32x16_bf16 per
// (32 outputs x 32 rows x 16 inputs), transposed formulation H^T = W^T X^T (mlp_engine.hpp):
//   * one wave = 32 rows, no workgroup-level synchronisation at all (waves are independent);
//   * activations live in the wave's own LDS area as ROW-MAJOR bf16 — the network input (kept for the skip
//     concatenations) and two ping-pong hidden buffers — so the B operand of any layer is a plain ds_read_b128 of the
//     lane's row, whatever the width, and a layer's output tile goes back with four 8-byte stores per lane;
//   * weights are packed on the host in LOGICAL feature order (no permutation is needed: the operand comes from LDS,
//     not from the previous tile's accumulators) as 1-KiB A fragments, read straight from global memory / L2;
//   * the layer table (input widths, tiles, activation, fragment and bias offsets) is a kernel argument.
// Limits: network input <= 320 features, hidden widths <= 256, <= 16 layers, output <= 256.  No backward: a model
// with a non-shipped shape renders (test.py, nerf_test.py) but trainvali raises NotImplementedError for it.
// nfx_embed is the Embedder (embedder.py:23-47) as its own kernel, with the point generation o + d z folded in.
#include "mlp_engine.hpp"
#include "mlp_generic.hpp"

namespace nfx {
namespace generic {

constexpr int kWavesPerBlock = 2;
constexpr int kXPitch = kMaxIn * 2 + 16, kHPitch = kMaxHidden * 2 + 16;   // bytes per row (+16: rows 4 banks apart)
constexpr int kWaveLds = 32 * (kXPitch + 2 * kHPitch);
constexpr int kLds = kWavesPerBlock * kWaveLds;
static_assert(kLds <= 160 * 1024, "LDS");

__device__ __forceinline__ float activate(float v, int act) {
    switch (act) {
        case 1: return fmaxf(v, 0.f);
        case 2: return sigmoidf(v);
        case 3: return softplusf(v);
        default: return v;
    }
}

__global__ __launch_bounds__(kWavesPerBlock * 64) void mlp_generic_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, p = lane & 31;
    char* xb = smem + wave * kWaveLds;                 // [32][kXPitch]  network input
    char* hb[2] = {xb + 32 * kXPitch, xb + 32 * kXPitch + 32 * kHPitch};
    const long long n_tiles_rows = (a.n + 31) / 32;
    for (long long rt = (long long)blockIdx.x * kWavesPerBlock + wave; rt < n_tiles_rows; rt += (long long)gridDim.x * kWavesPerBlock) {
        const long long row0 = rt * 32;
        // ---- network input -> bf16 rows (zero padded to a multiple of 16 features); lane half g takes the odd / even 16-byte groups
        const int ks_in = (a.d_in + 15) / 16;
        {
            const long long r = row0 + p < a.n ? row0 + p : a.n - 1;
            const float* src = a.x + r * a.ld_x;
            for (int c0 = 8 * g; c0 < ks_in * 16; c0 += 16) {
                bf16x8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (__bf16)(c0 + j < a.d_in ? src[c0 + j] : 0.f);
                *reinterpret_cast<bf16x8*>(xb + p * kXPitch + c0 * 2) = v;
            }
        }
        int cur = 0;
        for (int l = 0; l < a.n_layers; ++l) {
            const Layer L = a.layer[l];
            const bool last = l == a.n_layers - 1;
            const char* hsrc = hb[cur] + p * kHPitch + g * 16;      // this lane's row, its 8 of every 16 features
            const char* xsrc = xb + p * kXPitch + g * 16;
            char* hdst = hb[cur ^ 1] + p * kHPitch;
            for (int t = 0; t < L.n_tiles; ++t) {
                f32x16 acc;
                {
                    const float* bt = a.biases + L.b_off + 32 * t + 4 * g;   // D row of register r: (r&3) + 8 (r>>2) + 4 g
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(bt + 8 * q);
                        acc[4 * q] = v[0]; acc[4 * q + 1] = v[1]; acc[4 * q + 2] = v[2]; acc[4 * q + 3] = v[3];
                    }
                }
                const char* w = a.weights + ((size_t)L.w_off + (size_t)t * (L.ks_h + L.ks_x)) * kFragBytes + lane * 16;
#pragma unroll 4
                for (int s = 0; s < L.ks_h; ++s) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(w + (size_t)s * kFragBytes);
                    const bf16x8 bf = *reinterpret_cast<const bf16x8*>(hsrc + s * 32);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc, 0, 0, 0);
                }
                w += (size_t)L.ks_h * kFragBytes;
#pragma unroll 4
                for (int s = 0; s < L.ks_x; ++s) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(w + (size_t)s * kFragBytes);
                    const bf16x8 bf = *reinterpret_cast<const bf16x8*>(xsrc + s * 32);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc, 0, 0, 0);
                }
                // D: lane = row p (+ half g), register r = output feature 32 t + (r&3) + 8 (r>>2) + 4 g
                if (last) {
                    if (row0 + p < a.n) {
                        float* dst = a.y + (row0 + p) * a.ld_y + a.col0;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int f = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * g;
                            if (f < L.n_out) dst[f] = activate(acc[r], L.act);
                        }
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {   // registers 4 q .. 4 q + 3 = four consecutive features: one 8-byte store
                        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                        bf16x4 v;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int f = 32 * t + j + 8 * q + 4 * g;
                            v[j] = (__bf16)(f < L.n_out ? activate(acc[4 * q + j], L.act) : 0.f);   // pad features: exact zeros
                        }
                        *reinterpret_cast<bf16x4*>(hdst + (32 * t + 8 * q + 4 * g) * 2) = v;
                    }
                }
            }
            cur ^= 1;
        }
    }
}

__global__ __launch_bounds__(256) void embed_kernel(EmbedArgs a) {
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= a.n) return;
    const long long src = row / a.per_ray;
    float v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (a.mode == 0) v[k] = a.x[src * 3 + k];
        else if (a.mode == 1) v[k] = a.x[src * 3 + k] + a.dir[src * 3 + k] * a.z[row];
        else if (a.mode == 2) v[k] = a.dir[src * 3 + k];
        else v[k] = a.dir[(row - src * a.per_ray) * 3 + k] - a.x[src * 3 + k];   // light (row % per_ray) - point
    }
    if (a.mode == 3) {   // shape.py:128-131: safe_l2_normalize(lxyz - x), eps 1e-6
        const float inv = 1.0f / sqrtf(fmaxf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2], 1e-6f));
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] *= inv;
    }
    float* o = a.out + row * a.ld_out + a.col0;
    int c = 0;
    if (a.incl_input) {
#pragma unroll
        for (int k = 0; k < 3; ++k) o[c++] = v[k];
    }
    for (int f = 0; f < a.n_freqs; ++f) {
        const float s = (float)(1 << f);     // 2^k x is exact in fp32, as in TensorFlow
        float sn[3], cs[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) sincos_cw(v[k] * s, sn[k], cs[k]);
#pragma unroll
        for (int k = 0; k < 3; ++k) o[c + k] = sn[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) o[c + 3 + k] = cs[k];
        c += 6;
    }
}

}  // namespace generic
}  // namespace nfx

extern "C" {
int nfx_launch_mlp_generic(const nfx::generic::Args* args, int max_blocks, hipStream_t st) {
    using namespace nfx::generic;
    if (args->n <= 0) return 0;
    const long long tiles = (args->n + 31) / 32, want = (tiles + kWavesPerBlock - 1) / kWavesPerBlock;
    const int grid = (int)(want < max_blocks ? want : max_blocks);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_generic_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(mlp_generic_kernel, dim3(grid), dim3(kWavesPerBlock * 64), kLds, st, *args);
    return (int)hipGetLastError();
}
int nfx_launch_embed(const nfx::generic::EmbedArgs* a, hipStream_t st) {
    if (a->n <= 0) return 0;
    hipLaunchKernelGGL(nfx::generic::embed_kernel, dim3((unsigned)((a->n + 255) / 256)), dim3(256), 0, st, *a);
    return (int)hipGetLastError();
}
}
