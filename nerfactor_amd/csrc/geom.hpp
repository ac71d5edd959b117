// geom.hpp — per-(point, light) geometry and BRDF math evaluated in registers (fp32).
//   dir_to          safe_l2_normalize(a - b), eps 1e-6          shape.py:128-144, util/math.py:63-64
//   world2local     gen_world2local                             util/geom.py:119-149
//   dir2rusink      Rusinkiewicz (phi_d, theta_h, theta_d)      util/geom.py:152-192
//   microfacet_spec GGX D * G(view) * Schlick F / (4|n.l||n.v|) brdf/microfacet/microfacet.py:30-111
//   linear2srgb                                                  util/img.py:140-163
#pragma once
#include "nfx_common.hpp"

namespace nfx {

__device__ __forceinline__ float dot3(const float (&a)[3], const float (&b)[3]) {
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
// tf.linalg.l2_normalize(x, epsilon=eps): x * rsqrt(max(sum x^2, eps))
__device__ __forceinline__ void normalize3(float (&v)[3], float eps) {
    const float inv = 1.0f / sqrtf(fmaxf(dot3(v, v), eps));
    v[0] *= inv;
    v[1] *= inv;
    v[2] *= inv;
}
__device__ __forceinline__ void dir_to(const float (&a)[3], const float (&b)[3], float (&out)[3]) {
    out[0] = a[0] - b[0];
    out[1] = a[1] - b[1];
    out[2] = a[2] - b[2];
    normalize3(out, 1e-6f);
}
__device__ __forceinline__ void cross3(const float (&a)[3], const float (&b)[3], float (&o)[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
// rows of rot = (tangent, binormal, normal)
__device__ __forceinline__ void world2local(const float (&normal_in)[3], float (&rot)[9]) {
    float n[3] = {normal_in[0], normal_in[1], normal_in[2]};
    normalize3(n, 1e-6f);
    const float z[3] = {0.0f + 1e-6f, 0.0f + 1e-6f, 1.0f + 1e-6f};  // geom.py:128
    float t[3], b[3];
    cross3(n, z, t);
    normalize3(t, 1e-6f);
    cross3(n, t, b);
    normalize3(b, 1e-6f);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        rot[k] = t[k];
        rot[3 + k] = b[k];
        rot[6 + k] = n[k];
    }
}
__device__ __forceinline__ void mat3_apply(const float (&m)[9], const float (&v)[3], float (&o)[3]) {
#pragma unroll
    for (int r = 0; r < 3; ++r) o[r] = m[3 * r] * v[0] + m[3 * r + 1] * v[1] + m[3 * r + 2] * v[2];
}
__device__ __forceinline__ float safe_acos(float x) { return acosf(fminf(fmaxf(x, -1.0f), 1.0f)); }

// rot_vec(vector, axis, angle) of geom.py:168-180 for the two fixed axes it is used with.
__device__ __forceinline__ void rot_about_z(const float (&v)[3], float ang, float (&o)[3]) {
    const float c = cosf(ang), s = sinf(ang);
    // v*c + z*(v.z)*(1-c) + cross(z, v)*s,  cross((0,0,1), v) = (-v1, v0, 0)
    o[0] = v[0] * c + 0.0f * v[2] * (1.0f - c) + (-v[1]) * s;
    o[1] = v[1] * c + 0.0f * v[2] * (1.0f - c) + v[0] * s;
    o[2] = v[2] * c + 1.0f * v[2] * (1.0f - c) + 0.0f * s;
}
__device__ __forceinline__ void rot_about_y(const float (&v)[3], float ang, float (&o)[3]) {
    const float c = cosf(ang), s = sinf(ang);
    // cross((0,1,0), v) = (v2, 0, -v0)
    o[0] = v[0] * c + 0.0f * v[1] * (1.0f - c) + v[2] * s;
    o[1] = v[1] * c + 1.0f * v[1] * (1.0f - c) + 0.0f * s;
    o[2] = v[2] * c + 0.0f * v[1] * (1.0f - c) + (-v[0]) * s;
}
// a = light direction, b = view direction (both in the local frame); out = (phi_d, theta_h, theta_d)
__device__ __forceinline__ void dir2rusink(const float (&a_in)[3], const float (&b_in)[3],
                                           float (&out)[3]) {
    float a[3] = {a_in[0], a_in[1], a_in[2]}, b[3] = {b_in[0], b_in[1], b_in[2]};
    normalize3(a, 1e-6f);
    normalize3(b, 1e-6f);
    float hv[3] = {(a[0] + b[0]) / 2.0f, (a[1] + b[1]) / 2.0f, (a[2] + b[2]) / 2.0f};
    normalize3(hv, 1e-6f);
    const float theta_h = safe_acos(hv[2]);
    const float phi_h = atan2f(hv[1], hv[0]);
    float tmp[3], diff[3];
    rot_about_z(b, -phi_h, tmp);
    rot_about_y(tmp, -theta_h, diff);
    const float theta_d = safe_acos(diff[2]);
    float phi_d = atan2f(diff[1], diff[0]);
    const float pi = 3.14159265358979323846f;
    phi_d = phi_d - floorf(phi_d / pi) * pi;  // tf.math.floormod(x, pi)
    out[0] = phi_d;
    out[1] = theta_h;
    out[2] = theta_d;
}

__device__ __forceinline__ float div_no_nan(float a, float b) { return b == 0.0f ? 0.0f : a / b; }

// Per-point terms of the microfacet BRDF that do not depend on the light (microfacet.py:75-90).
struct MicrofacetPoint {
    float v[3], n[3];
    float alpha;      // rough^2
    float cos_v;      // n.v
    float g_denom;    // 1 + sqrt(1 + alpha^2 tan^2(theta_v))
};
__device__ __forceinline__ void microfacet_point(const float (&pts2c)[3], const float (&normal)[3],
                                                 float rough, MicrofacetPoint& mp) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        mp.v[k] = pts2c[k];
        mp.n[k] = normal[k];
    }
    normalize3(mp.v, 1e-6f);  // microfacet.py:48-50 re-normalises its inputs
    normalize3(mp.n, 1e-6f);
    mp.alpha = rough * rough;
    mp.cos_v = dot3(mp.n, mp.v);
    float cv2 = fminf(fmaxf(mp.cos_v * mp.cos_v, 0.0f), 1.0f);
    float tan2 = fmaxf(div_no_nan(1.0f - cv2, cv2), 0.0f);
    mp.g_denom = 1.0f + sqrtf(1.0f + mp.alpha * mp.alpha * tan2);
}
// Achromatic glossy term for one light direction l (already unit length).
__device__ __forceinline__ float microfacet_spec(const MicrofacetPoint& mp, const float (&l_in)[3],
                                                 float f0) {
    float l[3] = {l_in[0], l_in[1], l_in[2]};
    normalize3(l, 1e-6f);
    float hv[3] = {l[0] + mp.v[0], l[1] + mp.v[1], l[2] + mp.v[2]};
    normalize3(hv, 1e-6f);
    const float ldh = dot3(l, hv);
    const float om = 1.0f - ldh;
    const float om2 = om * om;
    const float f = f0 + (1.0f - f0) * (om2 * om2 * om);                        // :106-111
    const float cos_m = dot3(hv, mp.n);
    const float chi = cos_m > 0.0f ? 1.0f : 0.0f;
    const float cm2 = cos_m * cos_m;
    const float tan_m2 = div_no_nan(1.0f - cm2, cm2);
    const float a2 = mp.alpha * mp.alpha;
    const float dden = 3.14159265358979323846f * (cm2 * cm2) * ((a2 + tan_m2) * (a2 + tan_m2));
    const float d = div_no_nan(a2 * chi, dden);                                 // :92-104
    const float cos_t = dot3(hv, mp.v);
    const float chi_g = div_no_nan(cos_t, mp.cos_v) > 0.0f ? 1.0f : 0.0f;
    const float g = div_no_nan(chi_g * 2.0f, mp.g_denom);                       // :74-90
    const float ldn = dot3(l, mp.n);
    const float denom = 4.0f * fabsf(ldn) * fabsf(mp.cos_v);
    return div_no_nan(f * g * d, denom);                                        // :58-61
}

__device__ __forceinline__ float linear2srgb1(float x) {
    x = fminf(fmaxf(x, 0.0f), 1.0f);
    return x <= 0.0031308f ? x * 12.92f : 1.055f * powf(x, 1.0f / 2.4f) - 0.055f;
}

}  // namespace nfx
