// mlp128_x3.hip — NFX_PREC_FP32 for the width-128 surface MLPs (normal / albedo / BRDF-code heads, light visibility,
// learned-BRDF specular term): fp32-class accuracy on the bf16 matrix pipe (mlp_x3.hpp: hi / lo operand pairs, three
// MFMAs per product).  One template for the three input kinds; the light-visibility network takes its plain 90-dim
// input here (the per-point fold of the bf16 kernels is an optimisation of their layer 0, not needed for a precision
// option), so the blob is [hi fragments | lo fragments | biases] with the UN-folded chunk geometry:
//   kind          L0 chunk   L1, L2    L3 chunk          out
//   NFX_IN_XYZ       4 frags    8 frags   12 (8 + 4 posenc)  8      136 fragments per half
//   NFX_IN_XYZ_LDIR  8 (6 used) 8         16 (8 + 6 used)    8      168
//   NFX_IN_Z_RUSINK  4 (2 used) 8         12 (8 + 2 used)    8      136
// (reference: shape.py:184-237, nerfactor.py:377-461 — the reference computes all of this in fp32).
#include "geom.hpp"
#include "mlp128_layout.hpp"
#include "mlp_x3.hpp"

namespace nfx {
namespace x3m {

using x3::Pair;
constexpr int kNW = 4;
constexpr int kRows = kNW * 32;

template <int KIND>
struct Geo {
    static constexpr int kKSX = KIND == 0 ? 4 : KIND == 1 ? 6 : 2;
    static constexpr int kP0 = KIND == 1 ? 8 : 4, kP3 = KIND == 1 ? 16 : 12;
    static constexpr int kNL0 = kP0 / 4, kNLH = 2, kNL3 = kP3 / 4, kNLO = 2;
    static constexpr int kFrags = 4 * kP0 + 32 + 32 + 4 * kP3 + 8;
    static constexpr int kWeightBytes = kFrags * 1024;
};
constexpr int kLds = 2 * x3::kSlot + m128::kMainBiasFloats * 4;

__device__ __forceinline__ float act(float v, int a) {
    switch (a) {
        case 1: return fmaxf(v, 0.0f);
        case 2: return sigmoidf(v);
        case 3: return softplusf(v);
        default: return v;
    }
}
__device__ __forceinline__ void to_pairs(const float (&v)[16], Pair (&out)[2]) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        __bf16 a, b;
        x3::split(v[q], a, b);
        out[q >> 3].hi[q & 7] = a;
        out[q >> 3].lo[q & 7] = b;
    }
}

struct Args {
    const float* xyz;      // [n,3]
    const float* xyz_dir;  // KIND 1: points the light directions are taken from
    const float* lxyz;     // KIND 1, 2: [L,3]
    const float* cam;      // KIND 2
    const float* normal;   // KIND 2
    const float* z;        // KIND 2: [n, z_dim]
    int z_dim;
    long long n;
    int n_lights;
    float xyz_scale;
    const char* blob;
    int out_dim, out_act;
    float post_scale, post_bias;
    float* out;            // KIND 0: [n, out_dim]; KIND 1, 2: [n, L]
};

template <int KIND>
__global__ __launch_bounds__(kNW * 64, 1) void mlp128_x3_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = Geo<KIND>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
    float* bias_lds = reinterpret_cast<float*>(smem + 2 * x3::kSlot);
    {
        const float* bsrc = reinterpret_cast<const float*>(a.blob + 2 * (size_t)G::kWeightBytes);
        for (int i = tid; i < m128::kMainBiasFloats; i += kNW * 64) bias_lds[i] = bsrc[i];
    }
    x3::Stream st;
    st.base = reinterpret_cast<const u32x4*>(a.blob);
    st.end = reinterpret_cast<const u32x4*>(a.blob + G::kWeightBytes);
    st.ghi = st.base;
    st.lo_off = G::kWeightBytes / 16;
    st.ring = smem;
    x3::prologue<G::kNL0>(st, tid);
    const long long n_rows = KIND == 0 ? a.n : a.n * a.n_lights;
    const long long n_tiles = (n_rows + kRows - 1) / kRows;
    for (long long tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
        const long long m = tl * kRows + wave * 32 + p;
        const long long mm = m < n_rows ? m : n_rows - 1;
        const long long pt = KIND == 0 ? mm : mm / a.n_lights;
        const int l = KIND == 0 ? 0 : (int)(mm % a.n_lights);
        Pair xin[G::kKSX];
        bool front = true;
        if constexpr (KIND == 0 || KIND == 1) {
            float x[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) x[k] = a.xyz_scale * a.xyz[pt * 3 + k];   // shape.py:199
            Pair pe[4];
            x3::posenc_pair<10>(x, h, pe);
#pragma unroll
            for (int s = 0; s < 4; ++s) xin[s] = pe[s];
            if constexpr (KIND == 1) {
                float xd[3], lp[3], d[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    xd[k] = a.xyz_dir[pt * 3 + k];
                    lp[k] = a.lxyz[l * 3 + k];
                }
                dir_to(lp, xd, d);                                                  // shape.py:128-131
                Pair pl[2];
                x3::posenc_pair<4>(d, h, pl);
                xin[4] = pl[0];
                xin[5] = pl[1];
            }
        } else {
            float x[3], lp[3], cm[3], nr[3], ldir[3], vdir[3], rot[9], ll[3], vl[3], rus[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                x[k] = a.xyz[pt * 3 + k];
                lp[k] = a.lxyz[l * 3 + k];
                cm[k] = a.cam[pt * 3 + k];
                nr[k] = a.normal[pt * 3 + k];
            }
            dir_to(lp, x, ldir);          // shape.py:128-131
            dir_to(cm, x, vdir);          // shape.py:137-140
            world2local(nr, rot);         // util/geom.py:119-149
            mat3_apply(rot, ldir, ll);    // nerfactor.py:418-419
            mat3_apply(rot, vdir, vl);
            dir2rusink(ll, vl, rus);      // util/geom.py:152-192
            front = ll[2] > 0.0f;         // nerfactor.py:429-432
            const float* zp = a.z + pt * a.z_dim;
            float v[16];
#pragma unroll
            for (int q = 0; q < 6; ++q) v[q] = sin_shifted(rus[q % 3] * (float)(1 << (q / 3)), h);
            v[6] = h ? rus[2] : rus[0];
            v[7] = h ? zp[0] : rus[1];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = 1 + 2 * j + h;
                v[8 + j] = i < a.z_dim ? zp[i] : 0.0f;
            }
            Pair pr[2];
            to_pairs(v, pr);
            xin[0] = pr[0];
            xin[1] = pr[1];
        }
        Pair ha[8], hb[8];
        x3::layer<G::kKSX, 0, 4, G::kNL0, G::kNLH, true>(st, tid, bias_lds, xin, xin, ha);
        x3::layer<8, 0, 4, G::kNLH, G::kNLH, true>(st, tid, bias_lds + 128, ha, xin, hb);
        x3::layer<8, 0, 4, G::kNLH, G::kNL3, true>(st, tid, bias_lds + 256, hb, xin, ha);
        x3::layer<8, G::kKSX, 4, G::kNL3, G::kNLO, true>(st, tid, bias_lds + 384, ha, xin, hb);
        f32x16 acc;
        x3::tile<8, 0, G::kNL0>(st, tid, bias_lds + 512, hb, xin, acc);
        if (m < n_rows) {
            if constexpr (KIND == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = r + 4 * h;
                    if (row < a.out_dim) a.out[m * a.out_dim + row] = a.post_scale * act(acc[r], a.out_act) + a.post_bias;
                }
            } else if constexpr (KIND == 1) {
                if (h == 0) a.out[m] = sigmoidf(acc[0]);                              // shape.py:236
            } else {
                if (h == 0) a.out[m] = front ? softplusf(acc[0]) : 0.0f;              // brdf.py:65, scatter_nd's zeros
            }
        }
    }
}

template <int KIND>
static int launch(const Args& a, int max_blocks, hipStream_t st) {
    const long long rows = KIND == 0 ? a.n : a.n * a.n_lights;
    if (rows <= 0) return 0;
    const long long tiles = (rows + kRows - 1) / kRows;
    const int grid = (int)(tiles < max_blocks ? tiles : max_blocks);
    auto k = mlp128_x3_kernel<KIND>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(kNW * 64), kLds, st, a);
    return (int)hipGetLastError();
}

}  // namespace x3m
}  // namespace nfx

extern "C" {
int nfx_mlp128_x3_weight_bytes(int in_kind) {
    using namespace nfx::x3m;
    return in_kind == 0 ? Geo<0>::kWeightBytes : in_kind == 1 ? Geo<1>::kWeightBytes : Geo<2>::kWeightBytes;
}
int nfx_launch_mlp128_x3(int in_kind, const float* xyz, const float* xyz_dir, const float* lxyz, const float* cam,
                         const float* normal, const float* z, int z_dim, long long n, int n_lights, float xyz_scale,
                         const void* blob, int out_dim, int out_act, float post_scale, float post_bias, float* out,
                         int max_blocks, hipStream_t st) {
    nfx::x3m::Args a{xyz, xyz_dir, lxyz, cam, normal, z, z_dim, n, n_lights, xyz_scale, (const char*)blob,
                     out_dim, out_act, post_scale, post_bias, out};
    if (in_kind == 0) return nfx::x3m::launch<0>(a, max_blocks, st);
    if (in_kind == 1) return nfx::x3m::launch<1>(a, max_blocks, st);
    return nfx::x3m::launch<2>(a, max_blocks, st);
}
}
