// nerf_mlp_x3.hip — NFX_PREC_FP32 for the fused NeRF MLP: fp32-class accuracy on the bf16 matrix pipe.
// Every operand is carried as a bf16 PAIR (hi = bf16(x), lo = bf16(x - hi): 16 significant bits) and every product is
// three MFMAs, a_hi b_hi + a_hi b_lo + a_lo b_hi, accumulated in fp32 (the dropped a_lo b_lo term is 2^-18 relative):
// weights, positional encoding and every re-quantised activation.  Same register-resident dataflow and fragment
// layout as the bf16 kernels; the blob is [hi fragments | lo fragments | biases] (nfx_nerf_pack_weights, NFX_PREC_FP32).
// 3 MFMAs per product at the bf16 rate is still ~5x the v_mfma_f32_32x32x2_f32 peak (157 TF).  Stated tolerance against
// the fp32 oracle: max |d rgb| <= 2e-4 (tests/test_gpu_nerf.py), the bound SURVEY.md §8d sets for an fp32 MFMA path.
#include "mlp_engine.hpp"
#include "nerf_layout.hpp"

namespace nfx {
namespace x3 {

constexpr int kNW = 4;
constexpr int kSlot = 2 * kSlotBytes;                       // [hi chunk | lo chunk]
constexpr int kLds = 2 * kSlot + nerf::kBiasFloats * 4;

struct Pair {
    bf16x8 hi, lo;
};

__device__ __forceinline__ void split(float v, __bf16& hi, __bf16& lo) {
    hi = (__bf16)v;
    lo = (__bf16)(v - (float)hi);
}

struct Stream {
    const u32x4* ghi;   // next chunk, hi half
    const u32x4* base;
    const u32x4* end;   // end of the hi half
    long long lo_off;   // u32x4 elements from a hi fragment to its lo twin
    char* ring;
    int cur;
};

template <int NL>
__device__ __forceinline__ void load2(const Stream& st, int tid, Stage<NL, kNW>& sh, Stage<NL, kNW>& sl) {
    sh.load(st.ghi, tid);
    sl.load(st.ghi + st.lo_off, tid);
}

template <int NL0>
__device__ __forceinline__ void prologue(Stream& st, int tid) {
    Stage<NL0, kNW> sh, sl;
    load2<NL0>(st, tid, sh, sl);
    sh.store(reinterpret_cast<u32x4*>(st.ring), tid);
    sl.store(reinterpret_cast<u32x4*>(st.ring + kSlotBytes), tid);
    st.ghi += NL0 * kPieceThreads;
    st.cur = 0;
    __syncthreads();
}

// One 32-row output tile from inputs [b1 ; b2] (pairs); consumes one chunk (hi + lo).
template <int KS1, int KS2, int NL_NEXT, int KS1A, int KS2A>
__device__ __forceinline__ void tile(Stream& st, int tid, const float* bias_tile, const Pair (&b1)[KS1A],
                                     const Pair (&b2)[KS2A], f32x16& acc) {
    const int lane = tid & 63, h = lane >> 5;
    {
        f32x16 a1[1];
        bias_init<1>(bias_tile, h, a1);
        acc = a1[0];
    }
    Stage<NL_NEXT, kNW> sh, sl;
    load2<NL_NEXT>(st, tid, sh, sl);
    const char* fh = st.ring + st.cur * kSlot + lane * 16;
    const char* fl = fh + kSlotBytes;
    static_for<0, KS1 + KS2>([&](auto S) {
        constexpr int s = decltype(S)::value;
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(fh + s * kFragBytes);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(fl + s * kFragBytes);
        const Pair& b = s < KS1 ? b1[s < KS1 ? s : 0] : b2[s >= KS1 ? s - KS1 : 0];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b.hi, acc, 0, 0, 0);   // small terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b.lo, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b.hi, acc, 0, 0, 0);
    });
    char* dst = st.ring + (st.cur ^ 1) * kSlot;
    sh.store(reinterpret_cast<u32x4*>(dst), tid);
    sl.store(reinterpret_cast<u32x4*>(dst + kSlotBytes), tid);
    st.ghi += NL_NEXT * kPieceThreads;
    if (st.ghi == st.end) st.ghi = st.base;
    st.cur ^= 1;
    __syncthreads();
}

template <bool RELU>
__device__ __forceinline__ void acc_to_pair(const f32x16& acc, Pair& lo8, Pair& hi8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v0 = acc[j], v1 = acc[8 + j];
        if (RELU) {
            v0 = __builtin_amdgcn_fmed3f(v0, 0.0f, __builtin_inff());
            v1 = __builtin_amdgcn_fmed3f(v1, 0.0f, __builtin_inff());
        }
        __bf16 a, b;
        split(v0, a, b);
        lo8.hi[j] = a;
        lo8.lo[j] = b;
        split(v1, a, b);
        hi8.hi[j] = a;
        hi8.lo[j] = b;
    }
}

template <int KS1, int KS2, int NT, int NL_SELF, int NL_NEXT, bool RELU, int KS1A, int KS2A, int NTA>
__device__ __forceinline__ void layer(Stream& st, int tid, const float* bias, const Pair (&b1)[KS1A],
                                      const Pair (&b2)[KS2A], Pair (&bout)[NTA]) {
    static_for<0, NT>([&](auto T) {
        constexpr int t = decltype(T)::value;
        f32x16 acc;
        tile<KS1, KS2, (t == NT - 1 ? NL_NEXT : NL_SELF)>(st, tid, bias + 32 * t, b1, b2, acc);
        acc_to_pair<RELU>(acc, bout[2 * t], bout[2 * t + 1]);
    });
}

// positional encoding in the slot layout of mlp_engine.hpp:posenc, as hi/lo pairs
template <int L>
__device__ __forceinline__ void posenc_pair(const float (&x)[3], int h, Pair (&out)[PeSlots<L>::kKS]) {
    constexpr int NQ = PeSlots<L>::kKS * 8;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        float v;
        if (q < 3 * L) v = sin_shifted(x[q % 3] * (float)(1 << (q / 3)), h);
        else if (q == 3 * L) v = h ? x[2] : x[0];
        else if (q == 3 * L + 1) v = h ? 0.0f : x[1];
        else v = 0.0f;
        __bf16 a, b;
        split(v, a, b);
        out[q >> 3].hi[q & 7] = a;
        out[q >> 3].lo[q & 7] = b;
    }
}

__global__ __launch_bounds__(kNW * 64, 1) void nerf_mlp_x3_kernel(
    const float* __restrict__ rayo, const float* __restrict__ rayd, const float* __restrict__ zbuf, long long n_pts,
    int n_samples, const char* __restrict__ blob, float4* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using namespace nerf;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
    constexpr int kTilePts = kNW * 32;
    float* bias_lds = reinterpret_cast<float*>(smem + 2 * kSlot);
    {
        const float* bsrc = reinterpret_cast<const float*>(blob + 2 * (size_t)kWeightBytes);
        for (int i = tid; i < kBiasFloats; i += kNW * 64) bias_lds[i] = bsrc[i];
    }
    Stream st;
    st.base = reinterpret_cast<const u32x4*>(blob);
    st.end = reinterpret_cast<const u32x4*>(blob + kWeightBytes);
    st.ghi = st.base;
    st.lo_off = kWeightBytes / 16;
    st.ring = smem;
    prologue<kNL0>(st, tid);
    const long long n_tiles = (n_pts + kTilePts - 1) / kTilePts;
    for (long long tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
        const long long m = tl * kTilePts + wave * 32 + p;
        Pair pe[4], pv[2];
        {
            const long long mm = m < n_pts ? m : n_pts - 1;
            const long long ray = mm / n_samples;
            const float zz = zbuf[mm];
            float x[3], d[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                d[k] = rayd[ray * 3 + k];
                x[k] = rayo[ray * 3 + k] + d[k] * zz;
            }
            posenc_pair<10>(x, h, pe);
            posenc_pair<4>(d, h, pv);
        }
        Pair ha[16], hb[16], r0[8];
        layer<4, 0, 8, kNL0, kNLH, true>(st, tid, bias_lds + kBiasL0, pe, pe, ha);
        layer<16, 0, 8, kNLH, kNLH, true>(st, tid, bias_lds + kBiasL0 + 256 * 1, ha, pe, hb);
        layer<16, 0, 8, kNLH, kNLH, true>(st, tid, bias_lds + kBiasL0 + 256 * 2, hb, pe, ha);
        layer<16, 0, 8, kNLH, kNLH, true>(st, tid, bias_lds + kBiasL0 + 256 * 3, ha, pe, hb);
        layer<16, 0, 8, kNLH, kNL5, true>(st, tid, bias_lds + kBiasL0 + 256 * 4, hb, pe, ha);
        layer<16, 4, 8, kNL5, kNLH, true>(st, tid, bias_lds + kBiasL0 + 256 * 5, ha, pe, hb);
        layer<16, 0, 8, kNLH, kNLH, true>(st, tid, bias_lds + kBiasL0 + 256 * 6, hb, pe, ha);
        layer<16, 0, 8, kNLH, kNLH, true>(st, tid, bias_lds + kBiasL0 + 256 * 7, ha, pe, hb);
        layer<16, 0, 8, kNLH, kNLH, false>(st, tid, bias_lds + kBiasBott, hb, pe, ha);  // bottleneck
        float sigma;
        {
            f32x16 acc;
            tile<16, 0, kNLR0>(st, tid, bias_lds + kBiasBott + 256, hb, pe, acc);
            sigma = acc[0];
        }
        layer<16, 2, 4, kNLR0, kNLR1, true>(st, tid, bias_lds + kBiasRgb0, ha, pv, r0);
        {
            f32x16 acc;
            tile<8, 0, kNL0>(st, tid, bias_lds + kBiasRgb1, r0, pe, acc);
            if (h == 0 && m < n_pts) out[m] = make_float4(acc[0], acc[1], acc[2], sigma);
        }
    }
}

}  // namespace x3
}  // namespace nfx

extern "C" int nfx_launch_nerf_mlp_x3(const float* rayo, const float* rayd, const float* z, long long n_pts,
                                      int n_samples, const void* blob, float* out, int max_blocks, hipStream_t stream) {
    using namespace nfx;
    if (n_pts <= 0) return 0;
    const long long n_tiles = (n_pts + x3::kNW * 32 - 1) / (x3::kNW * 32);
    const int grid = (int)(n_tiles < max_blocks ? n_tiles : max_blocks);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(x3::nerf_mlp_x3_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, x3::kLds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(x3::nerf_mlp_x3_kernel, dim3(grid), dim3(x3::kNW * 64), x3::kLds, stream, rayo, rayd, z, n_pts,
                       n_samples, (const char*)blob, (float4*)out);
    return (int)hipGetLastError();
}
