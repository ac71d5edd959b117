// mlp128.hip — the width-128 surface MLPs of NeRFactor as fused bf16-MFMA kernels:
//   mlp128_xyz   _pred_normal_at / _pred_albedo_at / _pred_brdf_at   shape.py:196-211,
//                                                                     nerfactor.py:377-411
//   lvis_pre + lvis   _pred_lvis_at over the light sphere             shape.py:213-237 (+128-135)
//   brdf_spec    learned-BRDF specular term of _eval_brdf_at          nerfactor.py:413-458,
//                                                                     util/geom.py:119-192
#include "geom.hpp"
#include "mlp128_layout.hpp"
#include "mlp_engine.hpp"

namespace nfx {

constexpr int kNW = 8;                     // 8 waves x 32 rows, two waves per SIMD
constexpr int kRowsPerTile = kNW * 32;
constexpr int kM128Lds = 2 * kSlotBytes + m128::kMainBiasFloats * 4;

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case 1: return fmaxf(v, 0.0f);
        case 2: return sigmoidf(v);
        case 3: return softplusf(v);
        default: return v;
    }
}

// Layers L1, L2 (shared by all three nets): ha -> hb -> ha
template <int NLAFTER, typename HA, typename HB, typename PE>
__device__ __forceinline__ void mid_layers(WStream& ws, int tid, const float* bias_lds, HA& ha,
                                           HB& hb, PE& dummy) {
    using namespace m128;
    layer<8, 0, 4, kNLH, kNLH, true, kNW>(ws, tid, bias_lds + 128, ha, dummy, hb);
    layer<8, 0, 4, kNLH, NLAFTER, true, kNW>(ws, tid, bias_lds + 256, hb, dummy, ha);
}

// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kNW * 64, 2) void mlp128_xyz_kernel(
    const float* __restrict__ xyz, long long n, float xyz_scale, const char* __restrict__ blob,
    int out_dim, int out_act, float post_scale, float post_bias, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using namespace m128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
    float* bias_lds = reinterpret_cast<float*>(smem + 2 * kSlotBytes);
    {
        const float* bsrc = reinterpret_cast<const float*>(blob + kMainWeightBytes);
        for (int i = tid; i < kMainBiasFloats; i += kNW * 64) bias_lds[i] = bsrc[i];
    }
    WStream ws;
    ws.gbase = reinterpret_cast<const u32x4*>(blob);
    ws.gend = reinterpret_cast<const u32x4*>(blob + kMainWeightBytes);
    ws.gnext = ws.gbase;
    ws.ring = smem;
    stream_prologue<kNL0, kNW>(ws, tid);
    const long long n_tiles = (n + kRowsPerTile - 1) / kRowsPerTile;
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long m = tile * kRowsPerTile + wave * 32 + p;
        const long long mm = m < n ? m : n - 1;
        float x[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) x[k] = xyz_scale * xyz[mm * 3 + k];  // shape.py:199
        bf16x8 pe[4][1], ha[8][1], hb[8][1];
        posenc<10, 1>(x, h, 0, pe);
        layer<4, 0, 4, kNL0, kNLH, true, kNW>(ws, tid, bias_lds, pe, pe, ha);
        mid_layers<kNL3>(ws, tid, bias_lds, ha, hb, pe);
        layer<8, 4, 4, kNL3, kNLOut, true, kNW>(ws, tid, bias_lds + 384, ha, pe, hb);
        f32x16 acc[1];
        tile_raw<8, 0, kNL0, kNW>(ws, tid, bias_lds + 512, hb, pe, acc);
        if (m < n) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = r + 4 * h;
                if (row < out_dim)
                    out[m * out_dim + row] = post_scale * apply_act(acc[0][r], out_act) + post_bias;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// lvis_pre: pre[n][0:128] = W0[xyz rows]^T pe(x_n) + b0, pre[n][128:256] = W3[xyz rows]^T pe(x_n) + b3
__global__ __launch_bounds__(kNW * 64, 2) void lvis_pre_kernel(
    const float* __restrict__ xyz, long long n, float xyz_scale, const char* __restrict__ blob,
    float* __restrict__ pre) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using namespace m128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
    float* bias_lds = reinterpret_cast<float*>(smem + 2 * kSlotBytes);
    {
        const float* bsrc = reinterpret_cast<const float*>(blob + kPreWeightBytes);
        for (int i = tid; i < kPreBiasFloats; i += kNW * 64) bias_lds[i] = bsrc[i];
    }
    WStream ws;
    ws.gbase = reinterpret_cast<const u32x4*>(blob);
    ws.gend = reinterpret_cast<const u32x4*>(blob + kPreWeightBytes);
    ws.gnext = ws.gbase;
    ws.ring = smem;
    stream_prologue<kNL0, kNW>(ws, tid);
    const long long n_tiles = (n + kRowsPerTile - 1) / kRowsPerTile;
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long m = tile * kRowsPerTile + wave * 32 + p;
        const long long mm = m < n ? m : n - 1;
        float x[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) x[k] = xyz_scale * xyz[mm * 3 + k];
        bf16x8 pe[4][1];
        posenc<10, 1>(x, h, 0, pe);
        static_for<0, 8>([&](auto T) {
            constexpr int t = decltype(T)::value;
            f32x16 acc[1];
            tile_raw<4, 0, kNL0, kNW>(ws, tid, bias_lds + 32 * t, pe, pe, acc);
            if (m < n) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {acc[0][4 * g], acc[0][4 * g + 1], acc[0][4 * g + 2], acc[0][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(pre + m * 256 + 32 * t + 8 * g + 4 * h) = v;
                }
            }
        });
    }
}

// lvis: rows (n, l); every wave owns 32 consecutive lights of ONE point (n_lights % 32 == 0).
__global__ __launch_bounds__(kNW * 64, 2) void lvis_kernel(
    const float* __restrict__ xyz, long long n, const float* __restrict__ lxyz, int n_lights,
    const float* __restrict__ pre, const char* __restrict__ blob, float* __restrict__ lvis) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using namespace m128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
    float* bias_lds = reinterpret_cast<float*>(smem + 2 * kSlotBytes);
    {
        const float* bsrc = reinterpret_cast<const float*>(blob + kMainWeightBytes);
        for (int i = tid; i < kMainBiasFloats; i += kNW * 64) bias_lds[i] = bsrc[i];
    }
    WStream ws;
    ws.gbase = reinterpret_cast<const u32x4*>(blob);
    ws.gend = reinterpret_cast<const u32x4*>(blob + kMainWeightBytes);
    ws.gnext = ws.gbase;
    ws.ring = smem;
    stream_prologue<kNL0, kNW>(ws, tid);
    const long long n_rows = n * n_lights;
    const long long n_tiles = (n_rows + kRowsPerTile - 1) / kRowsPerTile;
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long m0 = tile * kRowsPerTile + wave * 32;  // wave-uniform
        const bool valid = m0 < n_rows;
        const long long mc = valid ? m0 : 0;
        const long long pt = mc / n_lights;
        const int l = (int)(mc % n_lights) + p;
        float x[3], d[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) x[k] = xyz[pt * 3 + k];
        // _calc_ldir (shape.py:128-131): normalize(lxyz[l] - x), eps 1e-6, from the UNscaled point
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            d[k] = lxyz[l * 3 + k] - x[k];
            sq += d[k] * d[k];
        }
        const float inv = 1.0f / sqrtf(fmaxf(sq, 1e-6f));
#pragma unroll
        for (int k = 0; k < 3; ++k) d[k] *= inv;
        bf16x8 pl[2][1], ha[8][1], hb[8][1];
        posenc<4, 1>(d, h, 0, pl);
        const float* pre_pt = pre + pt * 256;
        layer_pre<2, 0, 4, kNL0, kNLH, true, kNW>(ws, tid, pre_pt, pl, pl, ha);
        mid_layers<kNL3>(ws, tid, bias_lds, ha, hb, pl);
        layer_pre<8, 2, 4, kNL3, kNLOut, true, kNW>(ws, tid, pre_pt + 128, ha, pl, hb);
        f32x16 acc[1];
        tile_raw<8, 0, kNL0, kNW>(ws, tid, bias_lds + 512, hb, pl, acc);
        if (valid && h == 0) lvis[m0 + p] = sigmoidf(acc[0][0]);  // shape.py:93 sigmoid out
    }
}

// ---------------------------------------------------------------------------------------
// brdf_spec: rows (n, l).  Input slots (k-step s, half h, element j), see brdf_input_slots():
//   s=0: h=0: sin(r0) sin(r1) sin(r2) sin(2r0) sin(2r1) sin(2r2) r0 r1
//        h=1: cos(r0) ...                         cos(2r2)       r2 z0
//   s=1: z_i (i >= 1) at h = (i-1)&1, j = (i-1)>>1; zero elsewhere
__global__ __launch_bounds__(kNW * 64, 2) void brdf_spec_kernel(
    const float* __restrict__ xyz, const float* __restrict__ cam, const float* __restrict__ normal,
    const float* __restrict__ z, int z_dim, const float* __restrict__ lxyz, int n_lights,
    const char* __restrict__ blob, long long n, float* __restrict__ spec) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using namespace m128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
    float* bias_lds = reinterpret_cast<float*>(smem + 2 * kSlotBytes);
    {
        const float* bsrc = reinterpret_cast<const float*>(blob + kMainWeightBytes);
        for (int i = tid; i < kMainBiasFloats; i += kNW * 64) bias_lds[i] = bsrc[i];
    }
    WStream ws;
    ws.gbase = reinterpret_cast<const u32x4*>(blob);
    ws.gend = reinterpret_cast<const u32x4*>(blob + kMainWeightBytes);
    ws.gnext = ws.gbase;
    ws.ring = smem;
    stream_prologue<kNL0, kNW>(ws, tid);
    const long long n_rows = n * n_lights;
    const long long n_tiles = (n_rows + kRowsPerTile - 1) / kRowsPerTile;
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long m0 = tile * kRowsPerTile + wave * 32;
        const bool valid = m0 < n_rows;
        const long long mc = valid ? m0 : 0;
        const long long pt = mc / n_lights;
        const int l = (int)(mc % n_lights) + p;
        float x[3], c[3], nr[3], lp[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            x[k] = xyz[pt * 3 + k];
            c[k] = cam[pt * 3 + k];
            nr[k] = normal[pt * 3 + k];
            lp[k] = lxyz[l * 3 + k];
        }
        float ldir[3], vdir[3], rot[9], ll[3], vl[3], rus[3];
        dir_to(lp, x, ldir);          // shape.py:128-131
        dir_to(c, x, vdir);           // shape.py:137-140
        world2local(nr, rot);         // util/geom.py:119-149
        mat3_apply(rot, ldir, ll);    // nerfactor.py:418-419
        mat3_apply(rot, vdir, vl);
        dir2rusink(ll, vl, rus);      // util/geom.py:152-192 with a = light, b = view
        const bool front = ll[2] > 0.0f;  // nerfactor.py:429-432
        // B operand
        float v[16];
#pragma unroll
        for (int q = 0; q < 6; ++q) v[q] = sin_shifted_small(rus[q % 3] * (float)(1 << (q / 3)), h);   // angles <= pi, 2 bands
        v[6] = h ? rus[2] : rus[0];
        v[7] = h ? z[pt * z_dim] : rus[1];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = 1 + 2 * j + h;  // z index held by (half h, element j) of k-step 1
            v[8 + j] = i < z_dim ? z[pt * z_dim + i] : 0.0f;
        }
        bf16x8 bin[2][1], ha[8][1], hb[8][1];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bin[s][0][j] = (__bf16)v[8 * s + j];
            mfma_operand_fence(bin[s][0]);
        }
        layer<2, 0, 4, kNL0, kNLH, true, kNW>(ws, tid, bias_lds, bin, bin, ha);
        mid_layers<kNL3>(ws, tid, bias_lds, ha, hb, bin);
        layer<8, 2, 4, kNL3, kNLOut, true, kNW>(ws, tid, bias_lds + 384, ha, bin, hb);
        f32x16 acc[1];
        tile_raw<8, 0, kNL0, kNW>(ws, tid, bias_lds + 512, hb, bin, acc);
        if (valid && h == 0) spec[m0 + p] = front ? softplusf(acc[0][0]) : 0.0f;  // brdf.py:65
    }
}

}  // namespace nfx

extern "C" {
static int grid_for(long long rows, int max_blocks) {
    const long long tiles = (rows + nfx::kRowsPerTile - 1) / nfx::kRowsPerTile;
    return (int)(tiles < max_blocks ? tiles : max_blocks);
}
#define NFX_SET_LDS(kern, bytes)                                                               \
    do {                                                                                       \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),               \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, bytes); \
        if (e_ != hipSuccess) return (int)e_;                                                  \
    } while (0)

int nfx_launch_mlp128_xyz(const float* xyz, long long n, float xyz_scale, const void* blob, int out_dim,
                          int out_act, float post_scale, float post_bias, float* out, int max_blocks,
                          hipStream_t st) {
    if (n <= 0) return 0;
    NFX_SET_LDS(nfx::mlp128_xyz_kernel, nfx::kM128Lds);
    hipLaunchKernelGGL(nfx::mlp128_xyz_kernel, dim3(grid_for(n, max_blocks)), dim3(nfx::kNW * 64),
                       nfx::kM128Lds, st, xyz, n, xyz_scale, (const char*)blob, out_dim, out_act,
                       post_scale, post_bias, out);
    return (int)hipGetLastError();
}
int nfx_launch_lvis_pre(const float* xyz, long long n, float xyz_scale, const void* blob_pre, float* pre,
                        int max_blocks, hipStream_t st) {
    if (n <= 0) return 0;
    NFX_SET_LDS(nfx::lvis_pre_kernel, nfx::kM128Lds);
    hipLaunchKernelGGL(nfx::lvis_pre_kernel, dim3(grid_for(n, max_blocks)), dim3(nfx::kNW * 64),
                       nfx::kM128Lds, st, xyz, n, xyz_scale, (const char*)blob_pre, pre);
    return (int)hipGetLastError();
}
int nfx_launch_lvis(const float* xyz, long long n, const float* lxyz, int n_lights, const float* pre,
                    const void* blob_main, float* lvis, int max_blocks, hipStream_t st) {
    if (n <= 0) return 0;
    NFX_SET_LDS(nfx::lvis_kernel, nfx::kM128Lds);
    hipLaunchKernelGGL(nfx::lvis_kernel, dim3(grid_for(n * n_lights, max_blocks)), dim3(nfx::kNW * 64),
                       nfx::kM128Lds, st, xyz, n, lxyz, n_lights, pre, (const char*)blob_main, lvis);
    return (int)hipGetLastError();
}
int nfx_launch_brdf_spec(const float* xyz, const float* cam, const float* normal, const float* z, int z_dim,
                         const float* lxyz, int n_lights, const void* blob, long long n, float* spec,
                         int max_blocks, hipStream_t st) {
    if (n <= 0) return 0;
    NFX_SET_LDS(nfx::brdf_spec_kernel, nfx::kM128Lds);
    hipLaunchKernelGGL(nfx::brdf_spec_kernel, dim3(grid_for(n * n_lights, max_blocks)),
                       dim3(nfx::kNW * 64), nfx::kM128Lds, st, xyz, cam, normal, z, z_dim, lxyz, n_lights,
                       (const char*)blob, n, spec);
    return (int)hipGetLastError();
}
}
