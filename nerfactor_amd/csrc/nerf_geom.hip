// nerf_geom.hip — density and its spatial gradient at sample points: sigma_raw(x) = sigma_out(enc(posenc(x))) and
// n(x) = -normalize(d relu(sigma_raw) / dx), the per-sample "normal" of geometry_from_nerf.py:280-297 (there a
// GradientTape.batch_jacobian through embedder + fine_enc + fine_sigma_out).  One fused kernel: forward through the
// 8x256 encoder keeping 1-bit ReLU masks, reverse sweep with the transposed weights in the same register-resident
// MFMA dataflow, input-gradient tiles landing in the positional-encoding SLOT layout, analytic posenc Jacobian.
#include "feat_store.hpp"
#include "rowsel.hpp"
#include "nerf_geom_layout.hpp"

namespace nfx {
namespace geo {

constexpr int kNW = 4;
constexpr int kRows = kNW * 32;
constexpr int kLds = 2 * kSlotBytes + nerf::kGeoFloats * 4;
constexpr int kNLD = 4;  // every backward chunk: 16 fragments

template <int KS1, int KS2, int NL_SELF, int NL_NEXT, int KS1A, int KS2A>
__device__ __forceinline__ void fwd_layer(WStream& ws, int tid, const float* bias, const bf16x8 (&b1)[KS1A][1],
                                          const bf16x8 (&b2)[KS2A][1], bf16x8 (&bout)[16][1], unsigned (&m)[4]) {
    static_for<0, 8>([&](auto T) {
        constexpr int t = decltype(T)::value;
        f32x16 acc[1];
        tile_raw<KS1, KS2, (t == 7 ? NL_NEXT : NL_SELF), kNW>(ws, tid, bias + 32 * t, b1, b2, acc);
        const unsigned bits = bwd::relu_bits16(acc[0]);
        if constexpr (t & 1) m[t >> 1] |= bits << 16;
        else m[t >> 1] = bits;
        acc_to_b<true, 1>(acc, bout[2 * t], bout[2 * t + 1]);
    });
}

__device__ __forceinline__ void dgrad(WStream& ws, int tid, const bf16x8 (&dz)[16][1], const unsigned (&m)[4],
                                      bf16x8 (&dout)[16][1]) {
    static_for<0, 8>([&](auto T) {
        constexpr int t = decltype(T)::value;
        f32x16 acc[1];
        tile_init<16, 0, kNLD, kNW>(ws, tid, [&](f32x16(&a)[1]) { bwd::zero_init<1>(a); }, dz, dz, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            dout[2 * t + (r >> 3)][0][r & 7] = (__bf16)(bwd::mask_bit(m, t, r) ? acc[0][r] : 0.f);
        mfma_operand_fence(dout[2 * t][0]);
        mfma_operand_fence(dout[2 * t + 1][0]);
    });
}

// d posenc-slot accumulators += (input rows of a layer)^T dz: 2 tiles = 32 slots per lane half
template <int NL_LAST>
__device__ __forceinline__ void input_grad(WStream& ws, int tid, const bf16x8 (&dz)[16][1], f32x16 (&dpe)[2]) {
    static_for<0, 2>([&](auto T) {
        constexpr int t = decltype(T)::value;
        f32x16 acc[1];
        tile_init<16, 0, (t == 1 ? NL_LAST : kNLD), kNW>(
            ws, tid, [&](f32x16(&a)[1]) { a[0] = dpe[t]; }, dz, dz, acc);
        dpe[t] = acc[0];
    });
}

// LIST (round 6): the points are the flat sample indices list[0 .. *count) — the samples whose density is positive
// (nfx_nerf_sigma_grad_rows: d relu(sigma) / dx of every other sample is zero, its output row is written by the selecting pass)
// — and point list[c]'s result goes to out[list[c]].  A workgroup with no tile leaves before it touches the weight stream.
template <bool LIST>
__global__ __launch_bounds__(kNW * 64, 1) void nerf_sigma_grad_kernel(
    const float* __restrict__ rayo, const float* __restrict__ rayd, const float* __restrict__ zbuf, long long n_pts,
    int n_samples, const char* __restrict__ blob, float4* __restrict__ out, const int* __restrict__ list,
    const int* __restrict__ count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using namespace nerf;
    if constexpr (LIST) {
        n_pts = *count;
        if ((long long)blockIdx.x * kRows >= n_pts) return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
    float* fl = reinterpret_cast<float*>(smem + 2 * kSlotBytes);
    {
        const float* src = reinterpret_cast<const float*>(blob + kGeoWeightBytes);
        for (int i = tid; i < kGeoFloats; i += kNW * 64) fl[i] = src[i];
    }
    WStream ws;
    ws.gbase = reinterpret_cast<const u32x4*>(blob);
    ws.gend = reinterpret_cast<const u32x4*>(blob + kGeoWeightBytes);
    ws.gnext = ws.gbase;
    ws.ring = smem;
    stream_prologue<kNL0, kNW>(ws, tid);
    const long long n_tiles = (n_pts + kRows - 1) / kRows;
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long row = tile * kRows + wave * 32 + p;
        const bool valid = row < n_pts;
        long long mm = valid ? row : n_pts - 1;
        if constexpr (LIST) mm = list[mm];
        float x[3];
        {
            const long long ray = mm / n_samples;
            const float zz = zbuf[mm];
#pragma unroll
            for (int k = 0; k < 3; ++k) x[k] = rayo[ray * 3 + k] + rayd[ray * 3 + k] * zz;
        }
        bf16x8 pe[4][1];
        posenc<10, 1>(x, h, 0, pe);
        // ------------------------------------------------------------------ forward, masks only
        unsigned mk[8][4];
        bf16x8 ha[16][1], hb[16][1];
        fwd_layer<4, 0, kNL0, kNLH>(ws, tid, fl + 256 * 0, pe, pe, ha, mk[0]);
        fwd_layer<16, 0, kNLH, kNLH>(ws, tid, fl + 256 * 1, ha, pe, hb, mk[1]);
        fwd_layer<16, 0, kNLH, kNLH>(ws, tid, fl + 256 * 2, hb, pe, ha, mk[2]);
        fwd_layer<16, 0, kNLH, kNLH>(ws, tid, fl + 256 * 3, ha, pe, hb, mk[3]);
        fwd_layer<16, 0, kNLH, kNL5>(ws, tid, fl + 256 * 4, hb, pe, ha, mk[4]);
        fwd_layer<16, 4, kNL5, kNLH>(ws, tid, fl + 256 * 5, ha, pe, hb, mk[5]);
        fwd_layer<16, 0, kNLH, kNLH>(ws, tid, fl + 256 * 6, hb, pe, ha, mk[6]);
        fwd_layer<16, 0, kNLH, kNLH>(ws, tid, fl + 256 * 7, ha, pe, hb, mk[7]);
        float sigma;
        {
            f32x16 acc[1];
            tile_raw<16, 0, kNLD, kNW>(ws, tid, fl + kGeoBiasSig, hb, pe, acc);
            sigma = __shfl(acc[0][0], p, 64);  // row 0 of the tile lives on the h = 0 lanes
        }
        // ------------------------------------------------------------------ reverse sweep
        // dZ7 = mask7 . W_sigma (the same vector for every point: no MFMA)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float wv = fl[kGeoWSig + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
                ha[2 * t + (r >> 3)][0][r & 7] = (__bf16)(bwd::mask_bit(mk[7], t, r) ? wv : 0.f);
            }
            mfma_operand_fence(ha[2 * t][0]);
            mfma_operand_fence(ha[2 * t + 1][0]);
        }
        f32x16 dpe[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) dpe[t][r] = 0.f;
        dgrad(ws, tid, ha, mk[6], hb);          // enc[7]^T -> dZ6
        dgrad(ws, tid, hb, mk[5], ha);          // enc[6]^T -> dZ5
        input_grad<kNLD>(ws, tid, ha, dpe);     // enc[5][256:]^T dZ5 -> posenc slots
        dgrad(ws, tid, ha, mk[4], hb);          // enc[5][:256]^T -> dZ4
        dgrad(ws, tid, hb, mk[3], ha);          // enc[4]^T -> dZ3
        dgrad(ws, tid, ha, mk[2], hb);          // enc[3]^T -> dZ2
        dgrad(ws, tid, hb, mk[1], ha);          // enc[2]^T -> dZ1
        dgrad(ws, tid, ha, mk[0], hb);          // enc[1]^T -> dZ0
        input_grad<kNL0>(ws, tid, hb, dpe);     // enc[0]^T dZ0 -> posenc slots; next chunk = L0 of the next tile
        // ------------------------------------------------------------------ posenc Jacobian (slot q = 8 s + j)
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
            const float inv = -1.0f / sqrtf(fmaxf(gx * gx + gy * gy + gz * gz, 1e-12f));  // -l2_normalize(., eps 1e-12)
            out[LIST ? mm : row] = make_float4(gx * inv, gy * inv, gz * inv, sigma);
        }
    }
}

// Density only, for the shadow-ray and coarse camera marches (eval_sigma_mlp, geometry_from_nerf.py:322-350, before its
// relu): the encoder + sigma tile are the first 65 chunks of the GEOM blob, so the weight stream wraps right after the
// sigma tile — none of the 13 bottleneck / rgb chunks of the inference blob is fetched, staged or synchronised on.
// 8 waves x 32 points like nerf_mlp_bf16_kernel<1, 8>; same arithmetic: bit-identical sigma.
__global__ __launch_bounds__(512, 2) void nerf_sigma_geo_kernel(
    const float* __restrict__ rayo, const float* __restrict__ rayd, const float* __restrict__ zbuf, long long n_pts,
    int n_samples, const char* __restrict__ blob, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using namespace nerf;
    constexpr int NW = 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
    constexpr int kTilePts = NW * 32;
    float* fl = reinterpret_cast<float*>(smem + 2 * kSlotBytes);
    {
        const float* src = reinterpret_cast<const float*>(blob + kGeoWeightBytes);
        for (int i = tid; i < kGeoWSig; i += NW * 64) fl[i] = src[i];   // encoder biases + the sigma bias tile
    }
    WStream ws;
    ws.gbase = reinterpret_cast<const u32x4*>(blob);
    ws.gend = reinterpret_cast<const u32x4*>(blob + (size_t)kGeoFwdFrags * kFragBytes);
    ws.gnext = ws.gbase;
    ws.ring = smem;
    stream_prologue<kNL0, NW>(ws, tid);
    const long long n_tiles = (n_pts + kTilePts - 1) / kTilePts;
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        bf16x8 pe[4][1];
        const long long m = tile * kTilePts + wave * 32 + p;
        {
            const long long mm = m < n_pts ? m : n_pts - 1;
            const long long ray = mm / n_samples;
            const float zz = zbuf[mm];
            float x[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) x[k] = rayo[ray * 3 + k] + rayd[ray * 3 + k] * zz;
            posenc<10, 1>(x, h, 0, pe);
        }
        bf16x8 ha[16][1], hb[16][1];
        layer<4, 0, 8, kNL0, kNLH, true, NW>(ws, tid, fl + 256 * 0, pe, pe, ha);
        layer<16, 0, 8, kNLH, kNLH, true, NW>(ws, tid, fl + 256 * 1, ha, pe, hb);
        layer<16, 0, 8, kNLH, kNLH, true, NW>(ws, tid, fl + 256 * 2, hb, pe, ha);
        layer<16, 0, 8, kNLH, kNLH, true, NW>(ws, tid, fl + 256 * 3, ha, pe, hb);
        layer<16, 0, 8, kNLH, kNL5, true, NW>(ws, tid, fl + 256 * 4, hb, pe, ha);
        layer<16, 4, 8, kNL5, kNLH, true, NW>(ws, tid, fl + 256 * 5, ha, pe, hb);
        layer<16, 0, 8, kNLH, kNLH, true, NW>(ws, tid, fl + 256 * 6, hb, pe, ha);
        layer<16, 0, 8, kNLH, kNLH, true, NW>(ws, tid, fl + 256 * 7, ha, pe, hb);
        f32x16 acc[1];
        tile_raw<16, 0, kNL0, NW>(ws, tid, fl + kGeoBiasSig, hb, pe, acc);   // next chunk: enc[0] tile 0 again
        if (h == 0 && m < n_pts) out[m] = acc[0][0];
    }
}

}  // namespace geo
}  // namespace nfx

// list / count null: every point; else n_pts = the list's capacity (sizes the grid), the kernel reads the count on the device
extern "C" int nfx_launch_nerf_sigma_grad_list(const float* rayo, const float* rayd, const float* z, long long n_pts,
                                               int n_samples, const void* blob, float* out, const int* list,
                                               const int* count, int max_blocks, hipStream_t st) {
    using namespace nfx;
    if (n_pts <= 0) return 0;
    const long long tiles = (n_pts + geo::kRows - 1) / geo::kRows;
    const int grid = (int)(tiles < max_blocks ? tiles : max_blocks);
    auto k = list ? geo::nerf_sigma_grad_kernel<true> : geo::nerf_sigma_grad_kernel<false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       geo::kLds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(geo::kNW * 64), geo::kLds, st, rayo, rayd, z, n_pts, n_samples,
                       (const char*)blob, (float4*)out, list, count);
    return (int)hipGetLastError();
}
extern "C" int nfx_launch_nerf_sigma_grad(const float* rayo, const float* rayd, const float* z, long long n_pts,
                                          int n_samples, const void* blob, float* out, int max_blocks, hipStream_t st) {
    return nfx_launch_nerf_sigma_grad_list(rayo, rayd, z, n_pts, n_samples, blob, out, nullptr, nullptr, max_blocks, st);
}

// The samples with a positive density, ascending (rowsel.hpp), for the LIST form of the gradient kernels; the same pass
// writes the output row of every sample without density: relu has no slope there, the gradient is zero and
// -l2_normalize(0) = 0 * (-1 / sqrt(1e-12)) = (-0, -0, -0) (the every-sample kernel forms g * 0 and so keeps g's signs on
// its zeros: equal as numbers); the density channel keeps the raw value.
// (A NaN density is listed: the gradient kernel then writes what it always wrote for it.)
namespace nfx {
namespace geo {
struct HasDensity {
    const float* sigma;
    float4* out;
    __device__ bool operator()(long long r) const { return !(sigma[r] <= 0.f); }
    __device__ void visit(long long r, bool on) const {
        if (!on) out[r] = make_float4(-0.f, -0.f, -0.f, sigma[r]);
    }
};
}  // namespace geo
}  // namespace nfx
extern "C" int nfx_launch_select_density(const float* sigma, long long n_pts, float* out, void* list_ws, hipStream_t st) {
    return nfx::rowsel::build(nfx::geo::HasDensity{sigma, (float4*)out}, n_pts, list_ws, st);
}

extern "C" int nfx_launch_nerf_sigma_geo(const float* rayo, const float* rayd, const float* z, long long n_pts,
                                         int n_samples, const void* blob, float* out, int max_blocks, hipStream_t st) {
    using namespace nfx;
    if (n_pts <= 0) return 0;
    const long long tiles = (n_pts + 255) / 256;
    const int grid = (int)(tiles < max_blocks ? tiles : max_blocks);
    constexpr int lds = 2 * kSlotBytes + nerf::kGeoWSig * 4;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(geo::nerf_sigma_geo_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(geo::nerf_sigma_geo_kernel, dim3(grid), dim3(512), lds, st, rayo, rayd, z, n_pts, n_samples,
                       (const char*)blob, out);
    return (int)hipGetLastError();
}
