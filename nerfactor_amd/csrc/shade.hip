// shade.hip — the rendering integral of NeRFactor, fused per surface point (HBM-bound, fp32):
//   directions (shape.py:128-144) -> BRDF (microfacet.py:30-111, or albedo/pi + learned spec,
//   nerfactor.py:459-461) -> cos / front-lit mask / visibility / solid angle / light products
//   -> sum over the light sphere -> clip -> linear2srgb       (nerfactor.py:315-365)
// One wave per surface point, the lights of the sphere on the lanes (l = lane, lane+64, ...);
// light positions, solid angles and all probes are staged once per workgroup in LDS.  Nothing of
// size N x L x 3 is materialised; lvis[n, :] is read once (one coalesced row per point).
#include "geom.hpp"

namespace nfx {

struct ShadeArgs {
    const float *xyz, *cam, *normal, *albedo, *rough, *spec, *lvis, *lxyz, *lareas, *lights;
    float spec_scale, f0;
    long long n;
    int n_lights, n_probes, to_srgb;
    float olat_inten, ambient;
    float* out;
    const int* lvis_row;   // optional (round 6): row of point i in `lvis` — the visibilities live in the caller's FULL [n_all, L] buffer
    const int* out_row;    // optional (round 6, OLAT kernel): row of point i in `out` — the caller's FULL [n_all, L, 3] buffer
    int* nan_flag;         // optional (round 6, OLAT kernel): 1 is OR-ed into it when a radiance is NaN BEFORE the clip to [0, 1]
                           // (tf.clip_by_value keeps a NaN, fminf / fmaxf do not: the check_numerics of nerfactor.py:363 has to look here)
};

struct PointCtx {
    float x[3], nrm[3], alb_pi[3];
    MicrofacetPoint mp;
};

__device__ __forceinline__ void load_point(const ShadeArgs& a, long long pt, PointCtx& pc) {
    float cam[3], v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        pc.x[k] = a.xyz[3 * pt + k];
        cam[k] = a.cam[3 * pt + k];
        pc.nrm[k] = a.normal[3 * pt + k];
        pc.alb_pi[k] = a.albedo[3 * pt + k] / 3.14159265358979323846f;  // microfacet.py:64
    }
    dir_to(cam, pc.x, v);  // shape.py:137-140
    if (a.spec == nullptr) microfacet_point(v, pc.nrm, a.rough[pt], pc.mp);
}

// T[c] = brdf[n,l,c] * lvis[n,l]*[cos>0] * cos[n,l] * area[l]     (nerfactor.py:325-336)
// lvis_l / spec_l: the point's visibility and (learned BRDF) specular value for light l, already in registers.
__device__ __forceinline__ void light_transport_v(const ShadeArgs& a, const PointCtx& pc, int l, float lvis_l,
                                                  float spec_l, const float* lxyz_s, const float* area_s,
                                                  float (&T)[3]) {
    const float lp[3] = {lxyz_s[3 * l], lxyz_s[3 * l + 1], lxyz_s[3 * l + 2]};
    float ldir[3];
    dir_to(lp, pc.x, ldir);  // shape.py:128-131
    const float cosv = dot3(ldir, pc.nrm);
    const float lv = cosv > 0.0f ? lvis_l : 0.0f;
    const float s = a.spec ? spec_l * a.spec_scale : microfacet_spec(pc.mp, ldir, a.f0);
    const float k = lv * cosv * area_s[l];
#pragma unroll
    for (int c = 0; c < 3; ++c) T[c] = (s + pc.alb_pi[c]) * k;
}
__device__ __forceinline__ void light_transport(const ShadeArgs& a, const PointCtx& pc, long long pt, int l,
                                                const float* lxyz_s, const float* area_s, float (&T)[3]) {
    light_transport_v(a, pc, l, a.lvis[(a.lvis_row ? (long long)a.lvis_row[pt] : pt) * a.n_lights + l], a.spec ? a.spec[pt * a.n_lights + l] : 0.0f, lxyz_s,
                      area_s, T);
}

__device__ __forceinline__ float tonemap(float v, int to_srgb) {
    v = fminf(fmaxf(v, 0.0f), 1.0f);  // nerfactor.py:338
    return to_srgb ? linear2srgb1(v) : v;
}

// 8 waves per workgroup share one LDS copy of the lights (63 KiB at 9 probes): two workgroups = 4 waves per SIMD; with
// 4 waves per workgroup the kernel ran 2 waves per SIMD and stalled on every visibility row (r02: 2.5 -> see profiles)
constexpr int kShadeWaves = 8;
constexpr int kLightsPerPass = 8;  // per lane -> 512 lights per pass

// LDS: lxyz[L*3] | area[L] | lights[P*L*3] | part[kShadeWaves][P*3]
__global__ __launch_bounds__(kShadeWaves * 64) void shade_kernel(ShadeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int L = a.n_lights, P = a.n_probes;
    float* lxyz_s = sm;
    float* area_s = lxyz_s + 3 * L;
    float* light_s = area_s + L;
    float* part_s = light_s + (size_t)P * L * 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 3 * L; i += blockDim.x) lxyz_s[i] = a.lxyz[i];
    for (int i = tid; i < L; i += blockDim.x) area_s[i] = a.lareas[i];
    for (int i = tid; i < P * L * 3; i += blockDim.x) light_s[i] = a.lights[i];
    __syncthreads();
    float* part = part_s + wave * P * 3;
    const long long stride = (long long)gridDim.x * kShadeWaves;
    // the visibility / specular rows of the NEXT point are fetched while this one is shaded (first pass of lights)
    float lvn[kLightsPerPass], spn[kLightsPerPass];
    auto fetch = [&](long long pt) {
#pragma unroll
        for (int k = 0; k < kLightsPerPass; ++k) {
            const int l = k * 64 + lane;
            const bool ok = pt < a.n && l < L;
            lvn[k] = ok ? a.lvis[(a.lvis_row ? (long long)a.lvis_row[pt] : pt) * L + l] : 0.0f;
            spn[k] = ok && a.spec ? a.spec[pt * L + l] : 0.0f;
        }
    };
    long long pt = (long long)blockIdx.x * kShadeWaves + wave;
    fetch(pt);
    for (; pt < a.n; pt += stride) {
        PointCtx pc;
        load_point(a, pt, pc);
        float lvc[kLightsPerPass], spc[kLightsPerPass];
#pragma unroll
        for (int k = 0; k < kLightsPerPass; ++k) {
            lvc[k] = lvn[k];
            spc[k] = spn[k];
        }
        fetch(pt + stride);
        for (int i = lane; i < P * 3; i += 64) part[i] = 0.0f;
        for (int l0 = 0; l0 < L; l0 += 64 * kLightsPerPass) {
            float T[kLightsPerPass][3];
#pragma unroll
            for (int k = 0; k < kLightsPerPass; ++k) {
                const int l = l0 + k * 64 + lane;
                if (l < L) {
                    if (l0 == 0) light_transport_v(a, pc, l, lvc[k], spc[k], lxyz_s, area_s, T[k]);
                    else light_transport(a, pc, pt, l, lxyz_s, area_s, T[k]);
                } else {
                    T[k][0] = T[k][1] = T[k][2] = 0.0f;
                }
            }
            for (int p = 0; p < P; ++p) {
                float s[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < kLightsPerPass; ++k) {
                    const int l = l0 + k * 64 + lane;
                    if (l < L) {
                        const float* lg = light_s + ((size_t)p * L + l) * 3;
                        s[0] += T[k][0] * lg[0];
                        s[1] += T[k][1] * lg[1];
                        s[2] += T[k][2] * lg[2];
                    }
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) s[c] = wave_sum(s[c]);
                if (lane == 0) {
                    part[3 * p] += s[0];
                    part[3 * p + 1] += s[1];
                    part[3 * p + 2] += s[2];
                }
            }
        }
        // part[] was written by lane 0 only; same-wave LDS ops are ordered
        for (int i = lane; i < P * 3; i += 64) a.out[pt * P * 3 + i] = tonemap(part[i], a.to_srgb);
    }
}

// OLAT: rgb[n, l, c] = tonemap(inten * T[l][c] + ambient * sum_l' T[l'][c])   (nerfactor.py:79-84,348-354)
// LDS: lxyz[L*3] | area[L].  HBM-bound on what it writes (12 B x L per point; read: the point's visibility / specular
// rows).  One wave per point; a lane keeps the transport of its 8 lights (l = lane + 64 k) in registers, the wave sums
// them for the ambient term and every lane stores its lights' three channels as one 12-byte piece (768 contiguous bytes
// per store instruction) — round 2's form staged T through LDS, read it back channel by channel (i % 3) and ran a
// libm powf per output float: 3.1 ms per 384 030 points = 1.0 TB/s.  The visibility / specular rows of the NEXT point
// are fetched while this one is shaded, as in shade_kernel.
struct f32x3_t {
    float x, y, z;
};
// linear2srgb with v_log_f32 / v_exp_f32 (1 ulp each): |error| <= 1e-6 on [0, 1] against powf — 1536 tonemaps per
// point make the libm pow the whole cost of the kernel otherwise
__device__ __forceinline__ float tonemap_fast(float v, int to_srgb) {
    v = fminf(fmaxf(v, 0.0f), 1.0f);
    if (!to_srgb) return v;
    const float p = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(v) * (1.0f / 2.4f));
    return v <= 0.0031308f ? v * 12.92f : 1.055f * p - 0.055f;
}

__global__ __launch_bounds__(kShadeWaves * 64) void shade_olat_kernel(ShadeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int L = a.n_lights;
    float* lxyz_s = sm;
    float* area_s = lxyz_s + 3 * L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 3 * L; i += blockDim.x) lxyz_s[i] = a.lxyz[i];
    for (int i = tid; i < L; i += blockDim.x) area_s[i] = a.lareas[i];
    __syncthreads();
    const long long stride = (long long)gridDim.x * kShadeWaves;
    float lvn[kLightsPerPass], spn[kLightsPerPass];
    auto fetch = [&](long long pt) {
#pragma unroll
        for (int k = 0; k < kLightsPerPass; ++k) {
            const int l = k * 64 + lane;
            const bool ok = pt < a.n && l < L;
            lvn[k] = ok ? a.lvis[(a.lvis_row ? (long long)a.lvis_row[pt] : pt) * L + l] : 0.0f;
            spn[k] = ok && a.spec ? a.spec[pt * L + l] : 0.0f;
        }
    };
    long long pt = (long long)blockIdx.x * kShadeWaves + wave;
    fetch(pt);
    bool bad = false;
    for (; pt < a.n; pt += stride) {
        PointCtx pc;
        load_point(a, pt, pc);
        float lvc[kLightsPerPass], spc[kLightsPerPass];
#pragma unroll
        for (int k = 0; k < kLightsPerPass; ++k) {
            lvc[k] = lvn[k];
            spc[k] = spn[k];
        }
        fetch(pt + stride);
        float T0[kLightsPerPass][3];          // the first 512 lights stay in registers
        float tot[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < kLightsPerPass; ++k) {
            const int l = k * 64 + lane;
            if (l < L) light_transport_v(a, pc, l, lvc[k], spc[k], lxyz_s, area_s, T0[k]);
            else T0[k][0] = T0[k][1] = T0[k][2] = 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) tot[c] += T0[k][c];
        }
        if (a.ambient != 0.0f || L > 64 * kLightsPerPass) {
            for (int l = 64 * kLightsPerPass + lane; l < L; l += 64) {   // (more than 512 lights: summed here, re-evaluated below)
                float T[3];
                light_transport(a, pc, pt, l, lxyz_s, area_s, T);
#pragma unroll
                for (int c = 0; c < 3; ++c) tot[c] += T[c];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) tot[c] = wave_sum(tot[c]) * a.ambient;
        } else {
            tot[0] = tot[1] = tot[2] = 0.0f;
        }
        float* o = a.out + (a.out_row ? (long long)a.out_row[pt] : pt) * L * 3;
#pragma unroll
        for (int k = 0; k < kLightsPerPass; ++k) {
            const int l = k * 64 + lane;
            if (l < L) {
                const float r0 = a.olat_inten * T0[k][0] + tot[0], r1 = a.olat_inten * T0[k][1] + tot[1],
                            r2 = a.olat_inten * T0[k][2] + tot[2];
                bad = bad || !(r0 + r1 + r2 == r0 + r1 + r2);      // (a NaN in any channel; Inf - Inf cannot arise: all terms >= 0)
                f32x3_t v;
                v.x = tonemap_fast(r0, a.to_srgb);
                v.y = tonemap_fast(r1, a.to_srgb);
                v.z = tonemap_fast(r2, a.to_srgb);
                *reinterpret_cast<f32x3_t*>(o + 3 * l) = v;
            }
        }
        for (int l = 64 * kLightsPerPass + lane; l < L; l += 64) {
            float T[3];
            light_transport(a, pc, pt, l, lxyz_s, area_s, T);
            const float r0 = a.olat_inten * T[0] + tot[0], r1 = a.olat_inten * T[1] + tot[1], r2 = a.olat_inten * T[2] + tot[2];
            bad = bad || !(r0 + r1 + r2 == r0 + r1 + r2);
            f32x3_t v;
            v.x = tonemap_fast(r0, a.to_srgb);
            v.y = tonemap_fast(r1, a.to_srgb);
            v.z = tonemap_fast(r2, a.to_srgb);
            *reinterpret_cast<f32x3_t*>(o + 3 * l) = v;
        }
    }
    if (a.nan_flag != nullptr && __ballot(bad) != 0ull && lane == 0) atomicOr(a.nan_flag, 1);
}

__global__ void dir2rusink_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n,
                                  float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float av[3] = {a[3 * i], a[3 * i + 1], a[3 * i + 2]};
    const float bv[3] = {b[3 * i], b[3 * i + 1], b[3 * i + 2]};
    float r[3];
    dir2rusink(av, bv, r);
    out[3 * i] = r[0];
    out[3 * i + 1] = r[1];
    out[3 * i + 2] = r[2];
}

}  // namespace nfx

extern "C" {
static nfx::ShadeArgs make_args(const float* xyz, const float* cam, const float* normal, const float* albedo,
                                const float* rough, const float* spec, float spec_scale, float f0,
                                const float* lvis, const float* lxyz, const float* lareas,
                                const float* lights, long long n, int n_lights, int n_probes, int to_srgb,
                                float olat_inten, float ambient, float* out, const int* lvis_row = nullptr,
                                const int* out_row = nullptr, int* nan_flag = nullptr) {
    nfx::ShadeArgs a;
    a.lvis_row = lvis_row;
    a.out_row = out_row;
    a.nan_flag = nan_flag;
    a.xyz = xyz; a.cam = cam; a.normal = normal; a.albedo = albedo; a.rough = rough; a.spec = spec;
    a.lvis = lvis; a.lxyz = lxyz; a.lareas = lareas; a.lights = lights;
    a.spec_scale = spec_scale; a.f0 = f0; a.n = n; a.n_lights = n_lights; a.n_probes = n_probes;
    a.to_srgb = to_srgb; a.olat_inten = olat_inten; a.ambient = ambient; a.out = out;
    return a;
}
__attribute__((visibility("default"))) size_t nfx_shade_lds_bytes(int n_lights, int n_probes) {   // public: include/nfx.h
    return sizeof(float) * ((size_t)4 * n_lights + (size_t)n_probes * n_lights * 3 +
                            (size_t)nfx::kShadeWaves * n_probes * 3);
}
size_t nfx_shade_olat_lds_bytes(int n_lights) { return sizeof(float) * (size_t)4 * n_lights; }
int nfx_launch_shade(const float* xyz, const float* cam, const float* normal, const float* albedo,
                     const float* rough, const float* spec, float spec_scale, float f0, const float* lvis,
                     const float* lxyz, const float* lareas, const float* lights, long long n, int n_lights,
                     int n_probes, int to_srgb, float* out, hipStream_t st, const int* lvis_row) {
    if (n <= 0) return 0;
    const size_t lds = nfx_shade_lds_bytes(n_lights, n_probes);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(nfx::shade_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    long long blocks = (n + nfx::kShadeWaves - 1) / nfx::kShadeWaves;
    if (blocks > 512) blocks = 512;   // two resident workgroups per CU, each stages the lights once and loops over points
    hipLaunchKernelGGL(nfx::shade_kernel, dim3((unsigned)blocks), dim3(nfx::kShadeWaves * 64), lds, st,
                       make_args(xyz, cam, normal, albedo, rough, spec, spec_scale, f0, lvis, lxyz, lareas,
                                 lights, n, n_lights, n_probes, to_srgb, 0.f, 0.f, out, lvis_row));
    return (int)hipGetLastError();
}
int nfx_launch_shade_olat(const float* xyz, const float* cam, const float* normal, const float* albedo,
                          const float* rough, const float* spec, float spec_scale, float f0,
                          const float* lvis, const float* lxyz, const float* lareas, float olat_inten,
                          float ambient, long long n, int n_lights, int to_srgb, float* out, hipStream_t st,
                          const int* lvis_row, const int* out_row, int* nan_flag) {
    if (n <= 0) return 0;
    const size_t lds = nfx_shade_olat_lds_bytes(n_lights);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(nfx::shade_olat_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    long long blocks = (n + nfx::kShadeWaves - 1) / nfx::kShadeWaves;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(nfx::shade_olat_kernel, dim3((unsigned)blocks), dim3(nfx::kShadeWaves * 64), lds, st,
                       make_args(xyz, cam, normal, albedo, rough, spec, spec_scale, f0, lvis, lxyz, lareas,
                                 nullptr, n, n_lights, 0, to_srgb, olat_inten, ambient, out, lvis_row, out_row, nan_flag));
    return (int)hipGetLastError();
}
int nfx_launch_dir2rusink(const float* a, const float* b, long long n, float* out, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(nfx::dir2rusink_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, b, n,
                       out);
    return (int)hipGetLastError();
}
}

// =========================================================================================
// Backward of the rendering integral for ONE light (the trained one), microfacet or given specular:
// d(loss)/d(rgb) -> d albedo, d rough | d spec, d normal, d lvis, d light   (tape.gradient through
// nerfactor.py:315-342 and microfacet.py:30-111).  One wave per point, two passes over the sphere.
// =========================================================================================
namespace nfx {

struct ShadeBwdArgs {
    ShadeArgs f;          // forward inputs; f.lights = the trained light [L,3]; f.out unused
    const float* drgb;    // [n,3]
    float *d_albedo, *d_rough, *d_spec, *d_normal, *d_lvis;
    long long* d_light_fx;   // [L,3] fixed-point (2^40) sums of the light gradient: integer atomics are order-independent
};

__device__ __forceinline__ float tonemap_grad(float s, int to_srgb) {
    if (s < 0.0f || s > 1.0f) return 0.0f;                 // clip_by_value passes gradient inside [0,1]
    if (!to_srgb) return 1.0f;
    return s <= 0.0031308f ? 12.92f : (1.055f / 2.4f) * powf(s, 1.0f / 2.4f - 1.0f);
}

// spec and its partial derivatives w.r.t. the (re-normalised) normal n^ and a2 = rough^4.
__device__ __forceinline__ float microfacet_spec_grad(const MicrofacetPoint& mp, const float (&l_in)[3], float f0,
                                                      float (&dn)[3], float& da2) {
    float l[3] = {l_in[0], l_in[1], l_in[2]};
    normalize3(l, 1e-6f);
    float hv[3] = {l[0] + mp.v[0], l[1] + mp.v[1], l[2] + mp.v[2]};
    normalize3(hv, 1e-6f);
    const float om = 1.0f - dot3(l, hv), om2 = om * om;
    const float F = f0 + (1.0f - f0) * (om2 * om2 * om);
    const float a2 = mp.alpha * mp.alpha;
    const float cm = dot3(hv, mp.n), q = cm * cm;
    const float chi = cm > 0.0f ? 1.0f : 0.0f;
    const float tm = div_no_nan(1.0f - q, q), E = a2 + tm;
    const float pi = 3.14159265358979323846f;
    const float D = div_no_nan(a2 * chi, pi * (q * q) * (E * E));
    // Derivatives in overflow-free closed form (the literal chain rule through q^2 E^2 and tan^2 produces inf - inf
    // for grazing half vectors, |h.n| < ~1e-6, which 512 lights x 1024 rays hit every few steps):
    //   q^2 E^2 = W^2 with W = q E = 1 + (a2 - 1) q  in [min(1,a2), max(1,a2)]   =>  D = a2 chi / (pi W^2)
    // Every quotient is a divide_no_nan, as in the forward (microfacet.py:92-104): where the reference's denominator
    // pi q^2 E^2 = pi W^2 is 0 its gradient is 0.  r03: with the roughness driven to 0 (sigmoid output 7e-17 after 115
    // steps on noise targets: a2 = rough^4 underflows to 0) and a half vector exactly along the normal (q = 1), W = 0
    // and the plain quotients gave 0 / 0 -> NaN in d rough, i.e. NaN in every parameter one optimizer step later.
    float dD_dq = 0.f, dD_da2 = 0.f;
    if (chi > 0.f && q > 0.f) {
        const float W = 1.0f + (a2 - 1.0f) * q;
        const float inv_w2 = div_no_nan(1.0f, pi * (W * W)), inv_w = div_no_nan(1.0f, W);
        dD_dq = -2.0f * a2 * (a2 - 1.0f) * inv_w2 * inv_w;
        dD_da2 = inv_w2 * (1.0f - 2.0f * a2 * q * inv_w);
    }
    const float cv = mp.cos_v, ct = dot3(hv, mp.v);
    const float chig = div_no_nan(ct, cv) > 0.0f ? 1.0f : 0.0f;
    const float cv2 = cv * cv, p = fminf(fmaxf(cv2, 0.0f), 1.0f);
    const float tv_raw = div_no_nan(1.0f - p, p), tv = fmaxf(tv_raw, 0.0f);
    const float s = sqrtf(1.0f + a2 * tv);
    const float G = div_no_nan(chig * 2.0f, 1.0f + s);
    //   with c = |n.v|, r = sqrt(a2 + (1 - a2) c^2) (= c s):  G = 2 chi_g c / (c + r)
    //   dG/dc = 2 a2 / (r (c + r)^2),   dG/da2 = -(1 - c^2) c / (r (c + r)^2)
    float dG_dcv = 0.f, dG_da2 = 0.f;
    if (chig > 0.f && cv2 > 0.0f && cv2 < 1.0f) {
        const float c = fabsf(cv), r = sqrtf(a2 + (1.0f - a2) * p), cr2 = (c + r) * (c + r);
        const float inv = div_no_nan(1.0f, r * cr2);
        dG_dcv = (cv > 0.f ? 1.f : -1.f) * 2.0f * a2 * inv;
        dG_da2 = -(1.0f - p) * c * inv;
    }
    const float cl = dot3(l, mp.n);
    const float den = 4.0f * fabsf(cl) * fabsf(cv);
    if (den == 0.0f) {
        dn[0] = dn[1] = dn[2] = 0.f;
        da2 = 0.f;
        return 0.f;
    }
    const float spec = F * G * D / den;
    const float sgl = cl > 0.f ? 1.f : (cl < 0.f ? -1.f : 0.f), sgv = cv > 0.f ? 1.f : (cv < 0.f ? -1.f : 0.f);
    const float kq = F / den * G * dD_dq * 2.0f * cm;   // along h
    const float kv = F / den * D * dG_dcv;               // along v
    const float kd = spec / den * 4.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        dn[k] = kq * hv[k] + kv * mp.v[k] - kd * (sgl * fabsf(cv) * l[k] + fabsf(cl) * sgv * mp.v[k]);
    da2 = F / den * (G * dD_da2 + D * dG_da2);
    return spec;
}

// The light's gradient d_light[l, c] = sum over points of dS_c b_c k is accumulated as 64-bit fixed point (2^40 per
// unit: resolution 9e-13, range +-8e6): integer addition is associative, so the result does not depend on the order the
// atomics land in — two runs of a training step give the same bits (float atomics did not).
constexpr double kLightFxScale = 1099511627776.0;
__global__ void light_fx_finish_kernel(const long long* __restrict__ fx, float* __restrict__ d_light, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d_light[i] += (float)((double)fx[i] / kLightFxScale);
}

__global__ __launch_bounds__(kShadeWaves * 64) void shade_bwd_kernel(ShadeBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const ShadeArgs& f = a.f;
    const int L = f.n_lights;
    float* lxyz_s = sm;
    float* area_s = lxyz_s + 3 * L;
    float* light_s = area_s + L;
    // d loss / d light of THIS workgroup's rays, fixed point as in the global buffer (round 5).  Round 4 sent every
    // (ray, light, channel) term to HBM as its own 64-bit device-scope atomic — 1.6 M of them on 1536 addresses per
    // 1024-ray step, which the memory side serialises: 130 us of a 1.3 ms training step for a 14-us forward.  The sums are
    // integers: adding them here first (ds_add_u64) and once per workgroup to the global buffer gives the same bits.
    unsigned long long* dl_s = reinterpret_cast<unsigned long long*>(sm + (7 * L + 1) / 2 * 2);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 3 * L; i += blockDim.x) lxyz_s[i] = f.lxyz[i];
    for (int i = tid; i < L; i += blockDim.x) area_s[i] = f.lareas[i];
    for (int i = tid; i < 3 * L; i += blockDim.x) light_s[i] = f.lights[i];
    if (a.d_light_fx)
        for (int i = tid; i < 3 * L; i += blockDim.x) dl_s[i] = 0ull;
    __syncthreads();
    const float pi = 3.14159265358979323846f;
    for (long long pt = (long long)blockIdx.x * kShadeWaves + wave; pt < f.n;
         pt += (long long)gridDim.x * kShadeWaves) {
        PointCtx pc;
        load_point(f, pt, pc);
        // ---- pass 1: the pre-tonemap sums S[c]
        float S[3] = {0.f, 0.f, 0.f};
        for (int l = lane; l < L; l += 64) {
            float T[3];
            light_transport(f, pc, pt, l, lxyz_s, area_s, T);
#pragma unroll
            for (int c = 0; c < 3; ++c) S[c] += T[c] * light_s[3 * l + c];
        }
        float dS[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            S[c] = wave_sum(S[c]);
            dS[c] = a.drgb[3 * pt + c] * tonemap_grad(S[c], f.to_srgb);
        }
        // ---- pass 2: gradients
        float d_alb[3] = {0.f, 0.f, 0.f};
        float d_nhat[3] = {0.f, 0.f, 0.f};  // w.r.t. the normal re-normalised inside the microfacet BRDF
        float d_dir[3] = {0.f, 0.f, 0.f};   // w.r.t. the normal as used in cos = l . n
        float d_a2 = 0.f;
        for (int l = lane; l < L; l += 64) {
            const float lp[3] = {lxyz_s[3 * l], lxyz_s[3 * l + 1], lxyz_s[3 * l + 2]};
            float ldir[3];
            dir_to(lp, pc.x, ldir);
            const float cosv = dot3(ldir, pc.nrm);
            const bool front = cosv > 0.0f;
            const float lvis = f.lvis[pt * L + l];
            const float area = area_s[l];
            const float k = front ? lvis * cosv * area : 0.0f;
            float dsn[3] = {0.f, 0.f, 0.f}, dsa2 = 0.f, s;
            if (f.spec) s = f.spec[pt * L + l] * f.spec_scale;
            else s = microfacet_spec_grad(pc.mp, ldir, f.f0, dsn, dsa2);
            float T = 0.f, U = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float lg = light_s[3 * l + c];
                const float b = s + pc.alb_pi[c];
                T += dS[c] * b * lg;
                U += dS[c] * lg;
                d_alb[c] += dS[c] * k * lg / pi;
                if (a.d_light_fx)
                    atomicAdd(dl_s + 3 * l + c, (unsigned long long)__double2ll_rn((double)(dS[c] * b * k) * kLightFxScale));
            }
            if (a.d_lvis) a.d_lvis[pt * L + l] = front ? cosv * area * T : 0.0f;
            const float d_cos = front ? lvis * area * T : 0.0f;
            const float d_s = k * U;
            if (f.spec) {
                if (a.d_spec) a.d_spec[pt * L + l] = d_s * f.spec_scale;
            } else {
                d_a2 += d_s * dsa2;
#pragma unroll
                for (int c = 0; c < 3; ++c) d_nhat[c] += d_s * dsn[c];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) d_dir[c] += d_cos * ldir[c];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            d_alb[c] = wave_sum(d_alb[c]);
            d_nhat[c] = wave_sum(d_nhat[c]);
            d_dir[c] = wave_sum(d_dir[c]);
        }
        d_a2 = wave_sum(d_a2);
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) a.d_albedo[3 * pt + c] = d_alb[c];
            // n^ = n / max(|n|, ..): d n = (d n^ - n^ (n^ . d n^)) / |n|
            float nn = sqrtf(fmaxf(dot3(pc.nrm, pc.nrm), 1e-6f));
            float nh[3] = {pc.nrm[0] / nn, pc.nrm[1] / nn, pc.nrm[2] / nn};
            const float proj = dot3(nh, d_nhat);
#pragma unroll
            for (int c = 0; c < 3; ++c) a.d_normal[3 * pt + c] = d_dir[c] + (d_nhat[c] - nh[c] * proj) / nn;
            if (a.d_rough && !f.spec) {
                const float r = f.rough[pt];
                const float da2_dr = 4.0f * r * r * r;      // a2 = rough^4
                a.d_rough[pt] = da2_dr == 0.0f ? 0.0f : d_a2 * da2_dr;   // (an underflowed roughness gets no gradient, never inf x 0)
            }
        }
    }
    if (a.d_light_fx) {
        __syncthreads();
        for (int i = tid; i < 3 * L; i += blockDim.x) {
            const unsigned long long v = dl_s[i];
            if (v != 0ull) atomicAdd(reinterpret_cast<unsigned long long*>(a.d_light_fx + i), v);
        }
    }
}

}  // namespace nfx

extern "C" int nfx_launch_shade_bwd(const float* xyz, const float* cam, const float* normal, const float* albedo,
                                    const float* rough, const float* spec, float spec_scale, float f0,
                                    const float* lvis, const float* lxyz, const float* lareas, const float* light,
                                    long long n, int n_lights, int to_srgb, const float* drgb, float* d_albedo,
                                    float* d_rough, float* d_spec, float* d_normal, float* d_lvis, float* d_light,
                                    void* workspace, hipStream_t st) {
    if (n <= 0) return 0;
    nfx::ShadeBwdArgs a;
    a.f = make_args(xyz, cam, normal, albedo, rough, spec, spec_scale, f0, lvis, lxyz, lareas, light, n, n_lights, 1,
                    to_srgb, 0.f, 0.f, nullptr);
    a.drgb = drgb; a.d_albedo = d_albedo; a.d_rough = d_rough; a.d_spec = d_spec; a.d_normal = d_normal;
    a.d_lvis = d_lvis;
    a.d_light_fx = d_light ? static_cast<long long*>(workspace) : nullptr;
    if (a.d_light_fx) nfx::launch_zero_words(workspace, 3ll * n_lights, st);   // (a kernel, not hipMemsetAsync: nfx_common.hpp)
    const size_t lds = sizeof(float) * (size_t)((7 * n_lights + 1) / 2 * 2) + (d_light ? sizeof(long long) * (size_t)3 * n_lights : 0);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(nfx::shade_bwd_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    long long blocks = (n + nfx::kShadeWaves - 1) / nfx::kShadeWaves;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(nfx::shade_bwd_kernel, dim3((unsigned)blocks), dim3(nfx::kShadeWaves * 64), lds, st, a);
    if (a.d_light_fx)
        hipLaunchKernelGGL(nfx::light_fx_finish_kernel, dim3((3 * n_lights + 255) / 256), dim3(256), 0, st,
                           a.d_light_fx, d_light, 3 * n_lights);
    return (int)hipGetLastError();
}
