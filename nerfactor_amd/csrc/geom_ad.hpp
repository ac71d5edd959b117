// geom_ad.hpp — forward-mode derivatives of the learned-BRDF geometry w.r.t. the surface normal:
// world2local (util/geom.py:119-149) and dir2rusink (util/geom.py:152-192) evaluated on dual numbers
// (value + 3 tangents = d/d normal), so J = d(rusink)/d(normal) comes out of the same arithmetic as the
// forward, with the reference's CUSTOM gradients where it defines them (util/math.py:24-60):
//   safe_acos'(x)      = -1 / (sqrt(1 - clip(x)^2 + 1e-6) + 1e-6)
//   safe_atan2(x, y)'  = ( y / (x^2 + y^2 + 1e-6), -x / (x^2 + y^2 + 1e-6) )
// tf.linalg.l2_normalize differentiates x * rsqrt(max(sum x^2, eps)) (no gradient through the max when
// the floor is active); floormod passes the gradient through.
#pragma once
#include "nfx_common.hpp"

namespace nfx {

struct Dual3 {
    float v;
    float d[3];
};
__device__ __forceinline__ Dual3 dconst(float v) { return {v, {0.f, 0.f, 0.f}}; }
__device__ __forceinline__ Dual3 operator+(const Dual3& a, const Dual3& b) {
    return {a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2]}};
}
__device__ __forceinline__ Dual3 operator-(const Dual3& a, const Dual3& b) {
    return {a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2]}};
}
__device__ __forceinline__ Dual3 operator-(const Dual3& a) { return {-a.v, {-a.d[0], -a.d[1], -a.d[2]}}; }
__device__ __forceinline__ Dual3 operator*(const Dual3& a, const Dual3& b) {
    return {a.v * b.v, {a.d[0] * b.v + a.v * b.d[0], a.d[1] * b.v + a.v * b.d[1], a.d[2] * b.v + a.v * b.d[2]}};
}
__device__ __forceinline__ Dual3 operator*(const Dual3& a, float s) {
    return {a.v * s, {a.d[0] * s, a.d[1] * s, a.d[2] * s}};
}
__device__ __forceinline__ Dual3 operator/(const Dual3& a, float s) { return a * (1.0f / s); }
__device__ __forceinline__ Dual3 chain(const Dual3& a, float fv, float fprime) {  // f(a)
    return {fv, {fprime * a.d[0], fprime * a.d[1], fprime * a.d[2]}};
}
__device__ __forceinline__ Dual3 dcos(const Dual3& a) { return chain(a, cosf(a.v), -sinf(a.v)); }
__device__ __forceinline__ Dual3 dsin(const Dual3& a) { return chain(a, sinf(a.v), cosf(a.v)); }
__device__ __forceinline__ Dual3 dsafe_acos(const Dual3& a) {
    const float xc = fminf(fmaxf(a.v, -1.0f), 1.0f);
    return chain(a, acosf(xc), -1.0f / (sqrtf(1.0f - xc * xc + 1e-6f) + 1e-6f));
}
__device__ __forceinline__ Dual3 dsafe_atan2(const Dual3& x, const Dual3& y) {  // atan2(x, y), TF argument order
    const float den = x.v * x.v + y.v * y.v + 1e-6f;
    const float gx = y.v / den, gy = -x.v / den;
    return {atan2f(x.v, y.v), {gx * x.d[0] + gy * y.d[0], gx * x.d[1] + gy * y.d[1], gx * x.d[2] + gy * y.d[2]}};
}
__device__ __forceinline__ void dnormalize3(Dual3 (&v)[3], float eps) {
    const Dual3 sq = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    Dual3 inv;
    if (sq.v > eps) inv = chain(sq, 1.0f / sqrtf(sq.v), -0.5f / (sq.v * sqrtf(sq.v)));
    else inv = dconst(1.0f / sqrtf(eps));
    v[0] = v[0] * inv;
    v[1] = v[1] * inv;
    v[2] = v[2] * inv;
}
__device__ __forceinline__ void dcross3(const Dual3 (&a)[3], const Dual3 (&b)[3], Dual3 (&o)[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// rusink(normal) for fixed unit directions ldir, vdir (world frame); also returns local l.z (front-lit test)
__device__ __forceinline__ void rusink_dual(const float (&normal)[3], const float (&ldir)[3], const float (&vdir)[3],
                                            Dual3 (&rus)[3], float& l_local_z) {
    Dual3 n[3] = {{normal[0], {1.f, 0.f, 0.f}}, {normal[1], {0.f, 1.f, 0.f}}, {normal[2], {0.f, 0.f, 1.f}}};
    dnormalize3(n, 1e-6f);
    const Dual3 z[3] = {dconst(1e-6f), dconst(1e-6f), dconst(1.0f + 1e-6f)};
    Dual3 t[3], b[3];
    dcross3(n, z, t);
    dnormalize3(t, 1e-6f);
    dcross3(n, t, b);
    dnormalize3(b, 1e-6f);
    auto dotf = [](const Dual3 (&r)[3], const float (&w)[3]) { return r[0] * w[0] + r[1] * w[1] + r[2] * w[2]; };
    Dual3 a[3] = {dotf(t, ldir), dotf(b, ldir), dotf(n, ldir)};   // light, local frame
    Dual3 c[3] = {dotf(t, vdir), dotf(b, vdir), dotf(n, vdir)};   // view, local frame
    l_local_z = a[2].v;
    dnormalize3(a, 1e-6f);
    dnormalize3(c, 1e-6f);
    Dual3 h[3] = {(a[0] + c[0]) / 2.0f, (a[1] + c[1]) / 2.0f, (a[2] + c[2]) / 2.0f};
    dnormalize3(h, 1e-6f);
    const Dual3 theta_h = dsafe_acos(h[2]);
    const Dual3 phi_h = dsafe_atan2(h[1], h[0]);
    // rot_vec(c, (0,0,1), -phi_h): v*cos + axis*(v.axis)*(1-cos) + cross(axis, v)*sin
    const Dual3 mph = -phi_h, cz = dcos(mph), sz = dsin(mph);
    const Dual3 one = dconst(1.0f);
    Dual3 tmp[3] = {c[0] * cz - c[1] * sz, c[1] * cz + c[0] * sz, c[2] * cz + c[2] * (one - cz)};
    // rot_vec(tmp, (0,1,0), -theta_h): cross((0,1,0), v) = (v2, 0, -v0)
    const Dual3 mth = -theta_h, cy = dcos(mth), sy = dsin(mth);
    Dual3 diff[3] = {tmp[0] * cy + tmp[2] * sy, tmp[1] * cy + tmp[1] * (one - cy), tmp[2] * cy - tmp[0] * sy};
    const Dual3 theta_d = dsafe_acos(diff[2]);
    Dual3 phi_d = dsafe_atan2(diff[1], diff[0]);
    const float pi = 3.14159265358979323846f;
    phi_d.v = phi_d.v - floorf(phi_d.v / pi) * pi;
    rus[0] = phi_d;
    rus[1] = theta_h;
    rus[2] = theta_d;
}

}  // namespace nfx
