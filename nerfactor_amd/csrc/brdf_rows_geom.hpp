// brdf_rows_geom.hpp — arguments of brdf_rows_geom.hip, shared with the C-ABI (capi_nerfactor.cpp)
#pragma once
namespace nfx {
namespace rowsgeom {

constexpr int kMaxZ = 8, kMaxFreqs = 8;

struct Args {
    const float* xyz;      // [n, 3]
    const float* cam;      // [n, 3]
    const float* normal;   // [n, 3]
    const float* z;        // [n, z_dim]
    const float* lxyz;     // [L, 3]
    long long n;
    int L, z_dim, n_freqs;
    float* rows;           // fwd: [n L, ld]: z_dim + 3 + 6 n_freqs columns written
    float* front;          // fwd: [n L] 1.0 / 0.0
    const float* d_rows;   // bwd: [n L, ld]
    int ld;
    float* d_normal;       // bwd: [n, 3]
    float* d_z;            // bwd: [n, z_dim]
};

}  // namespace rowsgeom
}  // namespace nfx
