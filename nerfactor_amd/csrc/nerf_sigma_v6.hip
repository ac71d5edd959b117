// nerf_sigma_v6.hip — density only, in the dataflow of the headline render kernel (round 6).
//   sigma_raw(x) = sigma_out(enc(posenc(x)))   (eval_sigma_mlp, geometry_from_nerf.py:322-350, before its relu)
// nerf_sigma_geo_kernel (nerf_geom.hip: two workgroups of eight waves per CU, register-staged weight stream) runs the 65
// encoder + sigma tiles at 1.11 PFLOP/s; the render kernel (nerf_mlp_v6.hip, DMA = 1: one wave per SIMD, 64 points per
// wave, LDS-DMA weight ring with counted vmcnt, the epilogue of tile i-1 under the MFMAs of tile i) sustains 1.45 on the
// same layers.  This translation unit is that kernel's tile / layer machinery (included, NFX_V6_SIGMA) over the GEOM blob:
// chunks 0..63 are the render blob's encoder chunks byte for byte, chunk 64 is the sigma tile, and chunk 65 — the first
// reverse-sweep chunk of the blob — is fetched into the ring and not multiplied, so that the 66-chunk sequence wraps on
// the 6-slot ring.  Same MFMAs on the same operands as every other density kernel: bit-identical sigma
// (tests/test_gpu_nerf.py::test_sigma_only_kernel_is_the_full_kernel_s_density).
#define NFX_V6_SIGMA 1
#include "nerf_mlp_v6.hip"
#include "nerf_geom_layout.hpp"

namespace nfx {
namespace v6s {

static_assert(nerf::chunk_frags(64) == 16 && nerf::chunk_frags(65) == 16 && used_frags(64) == 16 && used_frags(65) == 16,
              "the sigma tile and the idle chunk are 16-fragment chunks");
static_assert(nerf::kGeoFwdFrags == nerf::chunk_frag_offset(65) && nerf::kGeoFrags >= nerf::chunk_frag_offset(66),
              "GEOM blob: sigma tile at chunk 64, a whole chunk behind it");
static_assert(nerf::kGeoFloats <= nerf::kBiasFloats, "the float section fits the render kernel's bias area");

constexpr int kDma = 1;

__global__ __launch_bounds__(kNW * 64, 1) void nerf_sigma_v6_kernel(
    const float* __restrict__ rayo, const float* __restrict__ rayd, const float* __restrict__ zbuf, long long n_pts,
    int n_samples, const char* __restrict__ blob, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using namespace nerf;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
    constexpr int kTilePts = kNW * 32 * kCT;
    float* fl = reinterpret_cast<float*>(smem + ring_of<kDma> * kSlotBytes);
    {
        const float* src = reinterpret_cast<const float*>(blob + kGeoWeightBytes);
        for (int i = tid; i < kGeoFloats; i += kNW * 64) fl[i] = src[i];
    }
    typedef __attribute__((address_space(3))) char lds_char;
    Ctx cx{smem, blob, tid, (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_char*)smem),
           __builtin_amdgcn_readfirstlane(tid >> 6)};
    Acc accs[2];
    Pre pre;
    Regs rg;
    {   // chunks 0, 1, 2 -> slots 0, 1, 2 (fetch distance 3)
        Stage<chunk_frags(0) / 4, kNW> s0;
        Stage<chunk_frags(1) / 4, kNW> s1;
        Stage<chunk_frags(2) / 4, kNW> s2;
        s0.load(reinterpret_cast<const u32x4*>(blob), tid);
        s1.load(reinterpret_cast<const u32x4*>(blob + (size_t)chunk_frag_offset(1) * kFragBytes), tid);
        s2.load(reinterpret_cast<const u32x4*>(blob + (size_t)chunk_frag_offset(2) * kFragBytes), tid);
        s0.store(reinterpret_cast<u32x4*>(smem), tid);
        s1.store(reinterpret_cast<u32x4*>(smem + kSlotBytes), tid);
        s2.store(reinterpret_cast<u32x4*>(smem + 2 * kSlotBytes), tid);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kPreA; ++i) pre.a[i] = *reinterpret_cast<const bf16x8*>(smem + lane * 16 + i * kFragBytes);
        bias_to_acc(fl, lane, accs[0]);
    }
    const long long n_tiles = (n_pts + kTilePts - 1) / kTilePts;
    for (long long tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
        bf16x8 pe[4][kCT];
        long long m[kCT];
#pragma unroll
        for (int c = 0; c < kCT; ++c) {
            m[c] = tl * kTilePts + wave * (32 * kCT) + c * 32 + p;
            const long long mm = m[c] < n_pts ? m[c] : n_pts - 1;
            const long long ray = mm / n_samples;
            const float zz = zbuf[mm];
            float x[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) x[k] = rayo[ray * 3 + k] + rayd[ray * 3 + k] * zz;
            posenc<10, kCT>(x, h, c, pe);
        }
        bf16x8 ha[16][kCT], hb[16][kCT];
        float sigma[kCT];
        auto pend = [&](const Acc& a, bf16x8(&lo)[kCT], bf16x8(&hi)[kCT]) { return EpiB<true>{a, lo, hi}; };
        // chunk index K: L0 0-7, L1-4 8-39, L5 40-47, L6-7 48-63, sigma 64, idle 65; tile K accumulates in accs[K & 1]
        layer<0, 4, 0, 8, true, 0, kDma>(cx, rg, fl, fl + 256 * 1, pe, pe, ha, accs, pre, EpiNone{});
        layer<8, 16, 0, 8, true, 0, kDma>(cx, rg, fl + 256 * 1, fl + 256 * 2, ha, pe, hb, accs, pre, pend(accs[1], ha[14], ha[15]));
        layer<16, 16, 0, 8, true, 0, kDma>(cx, rg, fl + 256 * 2, fl + 256 * 3, hb, pe, ha, accs, pre, pend(accs[1], hb[14], hb[15]));
        layer<24, 16, 0, 8, true, 0, kDma>(cx, rg, fl + 256 * 3, fl + 256 * 4, ha, pe, hb, accs, pre, pend(accs[1], ha[14], ha[15]));
        layer<32, 16, 0, 8, true, 0, kDma>(cx, rg, fl + 256 * 4, fl + 256 * 5, hb, pe, ha, accs, pre, pend(accs[1], hb[14], hb[15]));
        layer<40, 16, 4, 8, true, 0, kDma>(cx, rg, fl + 256 * 5, fl + 256 * 6, ha, pe, hb, accs, pre, pend(accs[1], ha[14], ha[15]));
        layer<48, 16, 0, 8, true, 0, kDma>(cx, rg, fl + 256 * 6, fl + 256 * 7, hb, pe, ha, accs, pre, pend(accs[1], hb[14], hb[15]));
        layer<56, 16, 0, 8, true, 0, kDma>(cx, rg, fl + 256 * 7, fl + kGeoBiasSig, ha, pe, hb, accs, pre, pend(accs[1], ha[14], ha[15]));
        // the sigma tile (K = 64 -> accs[0]); pending: the last tile of enc[7] (accs[1] -> hb[14], hb[15]); the bias handed to
        // the idle tile's accumulators is never used
        tile<64, 16, 0, 0, kDma>(cx, rg, fl + kGeoBiasSig, hb, pe, accs[0], accs[1], pre, pend(accs[1], hb[14], hb[15]));
        // the idle tile (K = 65 -> accs[1], ablation mask 4: A fragments read, no MFMA): chunk 2 of the next pass is fetched,
        // sigma leaves accs[0], accs[0] gets the bias of L0's first tile
        {
            EpiSigma es{accs[0], sigma};
            tile<65, 16, 0, 4, kDma>(cx, rg, fl, hb, pe, accs[1], accs[0], pre, es);
        }
        if (h == 0) {
#pragma unroll
            for (int c = 0; c < kCT; ++c)
                if (m[c] < n_pts) out[m[c]] = sigma[c];
        }
    }
}

}  // namespace v6s
}  // namespace nfx

extern "C" int nfx_launch_nerf_sigma_v6(const float* rayo, const float* rayd, const float* z, long long n_pts, int n_samples,
                                        const void* blob, float* out, int max_blocks, hipStream_t stream) {
    using namespace nfx;
    if (n_pts <= 0) return 0;
    const int tile_pts = v6s::kNW * 32 * v6s::kCT;
    const long long n_tiles = (n_pts + tile_pts - 1) / tile_pts;
    const int grid = (int)(n_tiles < max_blocks ? n_tiles : max_blocks);
    constexpr int lds = v6s::lds_of<v6s::kDma>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(v6s::nerf_sigma_v6_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(v6s::nerf_sigma_v6_kernel, dim3(grid), dim3(v6s::kNW * 64), lds, stream, rayo, rayd, z, n_pts,
                       n_samples, (const char*)blob, out);
    return (int)hipGetLastError();
}
