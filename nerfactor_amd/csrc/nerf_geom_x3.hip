// nerf_geom_x3.hip — NFX_PREC_FP32 of nerf_geom.hip: density and its spatial gradient at sample points with fp32-class
// arithmetic on the bf16 matrix pipe (mlp_x3.hpp: hi / lo operand pairs, three MFMAs per product) — forward
// activations, the back-propagated dZ and both weight orientations.  Same schedule as the bf16 kernel (forward through
// the 8 x 256 encoder keeping 1-bit ReLU masks, reverse sweep with the transposed weights, input-gradient tiles landing
// in the positional-encoding slot layout, analytic posenc Jacobian); the blob is nerf_geom_layout.hpp's chunk sequence
// twice, [hi fragments | lo fragments | floats], the sigma_out kernel among the floats in full fp32
// (geometry_from_nerf.py:280-297, 322-350: the reference differentiates in fp32).  Register budget: two 256-feature
// activations as pairs are 256 VGPRs, with masks, staging and accumulators the gradient kernel spills 82 dwords per
// lane to scratch — accepted for a precision option (the bf16 kernel is the fast path).
#include "feat_store.hpp"
#include "mlp_x3.hpp"
#include "nerf_geom_layout.hpp"

namespace nfx {
namespace geo3 {

using x3::Pair;
constexpr int kNW = x3::kNW;
constexpr int kRows = kNW * 32;
constexpr int kZeroTile = nerf::kGeoFloats;                    // 32 zero floats: the "bias" of the reverse-sweep tiles
constexpr int kLds = 2 * x3::kSlot + (nerf::kGeoFloats + 32) * 4;
constexpr int kNLD = 4;                                        // every backward chunk: 16 fragments

template <int KS1, int KS2, int NL_SELF, int NL_NEXT, int KS1A, int KS2A>
__device__ __forceinline__ void fwd_layer(x3::Stream& st, int tid, const float* bias, const Pair (&b1)[KS1A],
                                          const Pair (&b2)[KS2A], Pair (&bout)[16], unsigned (&m)[4]) {
    static_for<0, 8>([&](auto T) {
        constexpr int t = decltype(T)::value;
        f32x16 acc;
        x3::tile<KS1, KS2, (t == 7 ? NL_NEXT : NL_SELF)>(st, tid, bias + 32 * t, b1, b2, acc);
        const unsigned bits = bwd::relu_bits16(acc);
        if constexpr (t & 1) m[t >> 1] |= bits << 16;
        else m[t >> 1] = bits;
        x3::acc_to_pair<true>(acc, bout[2 * t], bout[2 * t + 1]);
    });
}

__device__ __forceinline__ void put(Pair (&dst)[16], int t, int r, float v) {
    __bf16 a, b;
    x3::split(v, a, b);
    dst[2 * t + (r >> 3)].hi[r & 7] = a;
    dst[2 * t + (r >> 3)].lo[r & 7] = b;
}

__device__ __forceinline__ void dgrad(x3::Stream& st, int tid, const float* zero, const Pair (&dz)[16],
                                      const unsigned (&m)[4], Pair (&dout)[16]) {
    static_for<0, 8>([&](auto T) {
        constexpr int t = decltype(T)::value;
        f32x16 acc;
        x3::tile<16, 0, kNLD>(st, tid, zero, dz, dz, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) put(dout, t, r, bwd::mask_bit(m, t, r) ? acc[r] : 0.f);
    });
}

// d posenc-slot accumulators += (input rows of a layer)^T dz: 2 tiles = 32 slots per lane half
template <int NL_LAST>
__device__ __forceinline__ void input_grad(x3::Stream& st, int tid, const float* zero, const Pair (&dz)[16],
                                           f32x16 (&dpe)[2]) {
    static_for<0, 2>([&](auto T) {
        constexpr int t = decltype(T)::value;
        f32x16 acc;
        x3::tile<16, 0, (t == 1 ? NL_LAST : kNLD)>(st, tid, zero, dz, dz, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) dpe[t][r] += acc[r];
    });
}

template <bool GRAD>
__global__ __launch_bounds__(kNW * 64, 1) void nerf_sigma_x3_kernel(
    const float* __restrict__ rayo, const float* __restrict__ rayd, const float* __restrict__ zbuf, long long n_pts,
    int n_samples, const char* __restrict__ blob, float* __restrict__ out, const int* __restrict__ list,
    const int* __restrict__ count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using namespace nerf;
    // LIST mode (round 6): the points are the flat sample indices list[0 .. *count) (both in device memory: the number of
    // selected samples never visits the host).  Density only (the selective coarse refinement of the bf16 render): the
    // density of point i goes to out[4 i + 3] — the sigma channel of rgbs[N, S, 4]; GRAD (nfx_nerf_sigma_grad_rows: the
    // samples with a positive density): (normal, sigma) of point i to row i of out[n][4].  A workgroup with no tile leaves at once.
    if (list != nullptr) {
        n_pts = *count;
        if ((long long)blockIdx.x * kRows >= n_pts) return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
    float* fl = reinterpret_cast<float*>(smem + 2 * x3::kSlot);
    {
        const float* src = reinterpret_cast<const float*>(blob + 2 * (size_t)kGeoWeightBytes);
        for (int i = tid; i < kGeoFloats; i += kNW * 64) fl[i] = src[i];
        if (tid < 32) fl[kZeroTile + tid] = 0.f;
    }
    const float* zero = fl + kZeroTile;
    x3::Stream st;
    st.base = reinterpret_cast<const u32x4*>(blob);
    // the density-only kernel wraps right after the sigma tile: none of the reverse-sweep chunks is fetched
    st.end = reinterpret_cast<const u32x4*>(blob + (GRAD ? (size_t)kGeoWeightBytes : (size_t)kGeoFwdFrags * kFragBytes));
    st.ghi = st.base;
    st.lo_off = kGeoWeightBytes / 16;
    st.ring = smem;
    x3::prologue<kNL0>(st, tid);
    const long long n_tiles = (n_pts + kRows - 1) / kRows;
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long row = tile * kRows + wave * 32 + p;
        const bool valid = row < n_pts;
        long long mm = valid ? row : n_pts - 1;
        if (list != nullptr) mm = list[mm];
        float x[3];
        {
            const long long ray = mm / n_samples;
            const float zz = zbuf[mm];
#pragma unroll
            for (int k = 0; k < 3; ++k) x[k] = rayo[ray * 3 + k] + rayd[ray * 3 + k] * zz;
        }
        Pair pe[4];
        x3::posenc_pair<10>(x, h, pe);
        // ------------------------------------------------------------------ forward, masks kept
        unsigned mk[8][4];
        Pair ha[16], hb[16];
        fwd_layer<4, 0, kNL0, kNLH>(st, tid, fl + 256 * 0, pe, pe, ha, mk[0]);
        fwd_layer<16, 0, kNLH, kNLH>(st, tid, fl + 256 * 1, ha, pe, hb, mk[1]);
        fwd_layer<16, 0, kNLH, kNLH>(st, tid, fl + 256 * 2, hb, pe, ha, mk[2]);
        fwd_layer<16, 0, kNLH, kNLH>(st, tid, fl + 256 * 3, ha, pe, hb, mk[3]);
        fwd_layer<16, 0, kNLH, kNL5>(st, tid, fl + 256 * 4, hb, pe, ha, mk[4]);
        fwd_layer<16, 4, kNL5, kNLH>(st, tid, fl + 256 * 5, ha, pe, hb, mk[5]);
        fwd_layer<16, 0, kNLH, kNLH>(st, tid, fl + 256 * 6, hb, pe, ha, mk[6]);
        fwd_layer<16, 0, kNLH, kNLH>(st, tid, fl + 256 * 7, ha, pe, hb, mk[7]);
        float sigma;
        {
            f32x16 acc;
            x3::tile<16, 0, (GRAD ? kNLD : kNL0)>(st, tid, fl + kGeoBiasSig, hb, pe, acc);
            sigma = acc[0];  // row 0 of the tile, on the h = 0 lanes
        }
        if constexpr (!GRAD) {
            if (valid && h == 0) {
                if (list != nullptr) out[4 * mm + 3] = sigma;
                else out[row] = sigma;
            }
        } else {
            // -------------------------------------------------------------- reverse sweep
            // dZ7 = mask7 . W_sigma (the same vector for every point: no MFMA)
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float wv = fl[kGeoWSig + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
                    put(ha, t, r, bwd::mask_bit(mk[7], t, r) ? wv : 0.f);
                }
            f32x16 dpe[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) dpe[t][r] = 0.f;
            dgrad(st, tid, zero, ha, mk[6], hb);          // enc[7]^T -> dZ6
            dgrad(st, tid, zero, hb, mk[5], ha);          // enc[6]^T -> dZ5
            input_grad<kNLD>(st, tid, zero, ha, dpe);     // enc[5][256:]^T dZ5 -> posenc slots
            dgrad(st, tid, zero, ha, mk[4], hb);          // enc[5][:256]^T -> dZ4
            dgrad(st, tid, zero, hb, mk[3], ha);          // enc[4]^T -> dZ3
            dgrad(st, tid, zero, ha, mk[2], hb);          // enc[3]^T -> dZ2
            dgrad(st, tid, zero, hb, mk[1], ha);          // enc[2]^T -> dZ1
            dgrad(st, tid, zero, ha, mk[0], hb);          // enc[1]^T -> dZ0
            input_grad<kNL0>(st, tid, zero, hb, dpe);     // enc[0]^T dZ0 -> posenc slots; next chunk = L0 again
            // -------------------------------------------------------------- posenc Jacobian (slot q = 8 s + j)
            float g[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const float dq = dpe[q >> 4][q & 15];
                if (q < 30) {
                    const float freq = (float)(1 << (q / 3));
                    // half 0 holds sin(f x): d/dx = f cos(f x); half 1 holds cos(f x): d/dx = -f sin(f x)
                    const float other = sin_shifted(x[q % 3] * freq, h ^ 1);
                    g[q % 3] += dq * freq * (h ? -other : other);
                } else if (q == 30) {
                    if (h) g[2] += dq; else g[0] += dq;
                } else {
                    if (!h) g[1] += dq;
                }
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) g[k] += __shfl_xor(g[k], 32, 64);
            if (valid && h == 0) {
                const float on = sigma > 0.f ? 1.f : 0.f;           // gradient of relu(sigma_raw)
                const float gx = g[0] * on, gy = g[1] * on, gz = g[2] * on;
                const float inv = -1.0f / sqrtf(fmaxf(gx * gx + gy * gy + gz * gz, 1e-12f));  // -l2_normalize(., 1e-12)
                reinterpret_cast<float4*>(out)[list != nullptr ? mm : row] = make_float4(gx * inv, gy * inv, gz * inv, sigma);
            }
        }
    }
}

template <bool GRAD>
static int launch(const float* rayo, const float* rayd, const float* z, long long n_pts, int n_samples,
                  const void* blob, float* out, int max_blocks, hipStream_t st, const int* list = nullptr,
                  const int* count = nullptr) {
    if (n_pts <= 0) return 0;
    const long long tiles = (n_pts + kRows - 1) / kRows;
    const int grid = (int)(tiles < max_blocks ? tiles : max_blocks);
    auto k = nerf_sigma_x3_kernel<GRAD>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(kNW * 64), kLds, st, rayo, rayd, z, n_pts, n_samples, (const char*)blob, out,
                       list, count);
    return (int)hipGetLastError();
}

}  // namespace geo3
}  // namespace nfx

extern "C" int nfx_launch_nerf_sigma_grad_x3(const float* rayo, const float* rayd, const float* z, long long n_pts,
                                             int n_samples, const void* blob, float* out, int max_blocks,
                                             hipStream_t st) {
    return nfx::geo3::launch<true>(rayo, rayd, z, n_pts, n_samples, blob, out, max_blocks, st);
}
extern "C" int nfx_launch_nerf_sigma_x3(const float* rayo, const float* rayd, const float* z, long long n_pts,
                                        int n_samples, const void* blob, float* out, int max_blocks, hipStream_t st) {
    return nfx::geo3::launch<false>(rayo, rayd, z, n_pts, n_samples, blob, out, max_blocks, st);
}
// density gradient at the listed samples, each written to its own row of out[n_pts][4] (nfx_nerf_sigma_grad_rows)
extern "C" int nfx_launch_nerf_sigma_grad_x3_list(const float* rayo, const float* rayd, const float* z, long long n_pts,
                                                  int n_samples, const void* blob, float* out, const int* list,
                                                  const int* count, int max_blocks, hipStream_t st) {
    return nfx::geo3::launch<true>(rayo, rayd, z, n_pts, n_samples, blob, out, max_blocks, st, list, count);
}
// density at the listed samples (n_pts = the list's capacity: sizes the grid; the kernel reads the real count on the device)
extern "C" int nfx_launch_nerf_sigma_x3_list(const float* rayo, const float* rayd, const float* z, long long n_pts,
                                             int n_samples, const void* blob, float* rgbs, const int* list,
                                             const int* count, int max_blocks, hipStream_t st) {
    return nfx::geo3::launch<false>(rayo, rayd, z, n_pts, n_samples, blob, rgbs, max_blocks, st, list, count);
}
