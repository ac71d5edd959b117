// mlp_x3.hpp — fp32-class arithmetic on the bf16 matrix pipe, shared by nerf_mlp_x3.hip and mlp128_x3.hip
// (NFX_PREC_FP32): every operand is a bf16 PAIR (hi = bf16(x), lo = bf16(x - hi): 16 significant bits), every product
// three MFMAs (a_lo b_hi + a_hi b_lo + a_hi b_hi, fp32 accumulate; the dropped a_lo b_lo term is 2^-18 relative).
// Same register-resident transposed dataflow and fragment layout as mlp_engine.hpp; a chunk is streamed as its hi
// fragments followed (lo_off further on in the blob) by its lo fragments.
#pragma once
#include "mlp_engine.hpp"

namespace nfx {
namespace x3 {

constexpr int kNW = 4;
constexpr int kSlot = 2 * kSlotBytes;                       // [hi chunk | lo chunk]

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
    mfma_operand_fence(lo8.hi);   // (mlp_engine.hpp: operands written by packed conversions)
    mfma_operand_fence(lo8.lo);
    mfma_operand_fence(hi8.hi);
    mfma_operand_fence(hi8.lo);
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
#pragma unroll
    for (int s = 0; s < PeSlots<L>::kKS; ++s) {
        mfma_operand_fence(out[s].hi);
        mfma_operand_fence(out[s].lo);
    }
}

}  // namespace x3
}  // namespace nfx
