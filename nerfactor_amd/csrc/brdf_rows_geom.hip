// brdf_rows_geom.hip — the learned BRDF's per-(point, light) geometry as EXPLICIT rows, fp32, forward and pull-back
// (round 5; nerfactor/models/nerfactor.py:413-461, util/geom.py:119-192, networks/embedder.py:23-47).
//
// `precision = fp32` training evaluates the frozen BRDF prior on the runtime-shaped fp32-class kernels (mlp_generic.hip),
// which take explicit input rows.  Round 4 assembled those rows with ~25 torch launches per call (local frames, two
// einsums, Rusinkiewicz angles through autograd, nonzero compaction, index_put scatter) and a host round trip for the
// front-lit count.  Here:
//   brdf_rows_geom_kernel      one thread per (point, light): world -> local frame of the normal, Rusinkiewicz angles
//                              (geom_ad.hpp:rusink_dual — the reference's op sequence), the Embedder, the row
//                              [z | rusink | sin, cos bands] and the front-lit flag [l_local.z > 0] (nerfactor.py:429-434);
//                              EVERY row is written (back-lit ones are evaluated by the MLP and multiplied by 0 — no
//                              data-dependent shape, so the step stays capturable in a hipGraph);
//   brdf_rows_geom_bwd_kernel  one wave per point: pulls dLoss/d row back through the Embedder and d rusink / d normal
//                              (forward-mode duals with the reference's custom gradients of safe_acos / safe_atan2) and sums
//                              over the point's lights — d normal[n], d z[n] — with a fixed-order wave reduction: no atomics,
//                              bit-reproducible.
#include "geom.hpp"
#include "geom_ad.hpp"
#include "nfx_common.hpp"
#include "brdf_rows_geom.hpp"

namespace nfx {
namespace rowsgeom {

__device__ __forceinline__ void row_geometry(const Args& a, long long pt, int l, Dual3 (&rus)[3], float& lz) {
    float x[3], c[3], nr[3], lp[3], ldir[3], vdir[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        x[k] = a.xyz[pt * 3 + k];
        c[k] = a.cam[pt * 3 + k];
        nr[k] = a.normal[pt * 3 + k];
        lp[k] = a.lxyz[l * 3 + k];
    }
    dir_to(lp, x, ldir);      // shape.py:128-131: safe_l2_normalize(lxyz - x)
    dir_to(c, x, vdir);       // shape.py:139-144
    rusink_dual(nr, ldir, vdir, rus, lz);
}

__global__ __launch_bounds__(256) void brdf_rows_geom_kernel(Args a) {
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= a.n * a.L) return;
    const long long pt = row / a.L;
    const int l = (int)(row - pt * a.L);
    Dual3 rus[3];
    float lz;
    row_geometry(a, pt, l, rus, lz);
    float* o = a.rows + row * a.ld;
    for (int i = 0; i < a.z_dim; ++i) o[i] = a.z[pt * a.z_dim + i];
    o += a.z_dim;
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = rus[k].v;
    for (int f = 0; f < a.n_freqs; ++f) {
        const float s = (float)(1 << f);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float sn, cs;
            sincos_cw(rus[k].v * s, sn, cs);
            o[3 + 6 * f + k] = sn;
            o[6 + 6 * f + k] = cs;
        }
    }
    a.front[row] = lz > 0.0f ? 1.0f : 0.0f;
}

__global__ __launch_bounds__(256) void brdf_rows_geom_bwd_kernel(Args a) {
    const int lane = threadIdx.x & 63;
    const long long pt = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pt >= a.n) return;      // (whole waves leave: no cross-wave synchronisation below)
    float dn[3] = {0.f, 0.f, 0.f}, dz[kMaxZ];
#pragma unroll
    for (int i = 0; i < kMaxZ; ++i) dz[i] = 0.f;
    for (int l = lane; l < a.L; l += 64) {
        Dual3 rus[3];
        float lz;
        row_geometry(a, pt, l, rus, lz);
        if (!(lz > 0.0f)) continue;                       // back-lit rows carry no gradient (their spec is the constant 0)
        const float* d = a.d_rows + (pt * a.L + l) * a.ld;
#pragma unroll
        for (int i = 0; i < kMaxZ; ++i)
            if (i < a.z_dim) dz[i] += d[i];
        d += a.z_dim;
        float g[3] = {d[0], d[1], d[2]};                  // d / d rusink: identity block + the bands
        for (int f = 0; f < a.n_freqs; ++f) {
            const float s = (float)(1 << f);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float sn, cs;
                sincos_cw(rus[k].v * s, sn, cs);
                g[k] += s * (cs * d[3 + 6 * f + k] - sn * d[6 + 6 * f + k]);
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) dn[j] += g[0] * rus[0].d[j] + g[1] * rus[1].d[j] + g[2] * rus[2].d[j];
    }
    // fixed-order butterfly over the 64 lanes (same tree every launch)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
        for (int j = 0; j < 3; ++j) dn[j] += __shfl_xor(dn[j], off, 64);
#pragma unroll
        for (int i = 0; i < kMaxZ; ++i)
            if (i < a.z_dim) dz[i] += __shfl_xor(dz[i], off, 64);
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 3; ++j) a.d_normal[pt * 3 + j] = dn[j];
#pragma unroll
        for (int i = 0; i < kMaxZ; ++i)
            if (i < a.z_dim) a.d_z[pt * a.z_dim + i] = dz[i];
    }
}

}  // namespace rowsgeom
}  // namespace nfx

extern "C" {
int nfx_launch_brdf_rows_geom(const nfx::rowsgeom::Args* a, int bwd, hipStream_t st) {
    if (a->n <= 0 || a->L <= 0) return 0;
    if (bwd) hipLaunchKernelGGL(nfx::rowsgeom::brdf_rows_geom_bwd_kernel, dim3((unsigned)((a->n + 3) / 4)), dim3(256), 0, st, *a);
    else hipLaunchKernelGGL(nfx::rowsgeom::brdf_rows_geom_kernel, dim3((unsigned)((a->n * a->L + 255) / 256)), dim3(256), 0, st, *a);
    return (int)hipGetLastError();
}
}
