// feat_store.hpp — feature-major bf16 stores of register-resident activations / gradients, shared by the fused
// backward kernels (mlp128_bwd.hip, nerf_bwd.hip).  The weight-gradient GEMM (train.hip) consumes [feature][row].
#pragma once
#include "mlp_engine.hpp"

namespace nfx {
namespace bwd {

// Feature-major store: wave-uniform base (SGPR pair, global_store saddr form) + 32-bit lane offset.
// `ld2` = bytes per feature row; it is laundered through an empty asm once per tile so the ~1100
// per-feature bases are computed next to their stores instead of being hoisted and spilled.
struct FeatStore {
    char* base;
    unsigned long long ld2;
    unsigned roff;  // row * 2
};
// Re-materialises the uniform base / stride behind an opaque asm so the per-feature base addresses of the NEXT
// group of stores are computed next to those stores (not hoisted to the top of the tile and spilled).
__device__ __forceinline__ FeatStore relaunder(const FeatStore& fs) {
    unsigned long long ld2 = fs.ld2, b = reinterpret_cast<unsigned long long>(fs.base);
    asm volatile("" : "+s"(ld2), "+s"(b));
    FeatStore o;
    o.base = reinterpret_cast<char*>(b);
    o.ld2 = ld2;
    o.roff = fs.roff;
    return o;
}
__device__ __forceinline__ void st16(const FeatStore& fs, int feat, __bf16 v) {
    // explicit global address space: the laundered integer base would otherwise make this a flat store with a
    // 64-bit VGPR address; this form is global_store_short v_off, v_data, s[base:base+1]
    typedef __attribute__((address_space(1))) __bf16* gbf16_ptr;
    const unsigned long long a = reinterpret_cast<unsigned long long>(fs.base) + (unsigned long long)feat * fs.ld2;
    gbf16_ptr rowp = (gbf16_ptr)a;
    *(gbf16_ptr)((__attribute__((address_space(1))) char*)rowp + fs.roff) = v;
}
#ifdef NFX_XP_ST32   // TIMING EXPERIMENT ONLY (wrong layout): one dword store per adjacent feature pair instead of two 16-bit stores
__device__ __forceinline__ void st32(const FeatStore& fs, int feat, unsigned v) {
    typedef __attribute__((address_space(1))) unsigned* gu32_ptr;
    const unsigned long long a = reinterpret_cast<unsigned long long>(fs.base) + (unsigned long long)feat * fs.ld2;
    *(gu32_ptr)((__attribute__((address_space(1))) char*)a + 2u * fs.roff) = v;   // row * 4: 256 distinct bytes per wave store
}
#endif
// B-operand registers of a hidden activation (k-step s, element j <-> feature F(s,h,j)) -> feature-major
template <int KS>
__device__ __forceinline__ void store_hidden(const FeatStore& fs0, int feat0, int h, const bf16x8 (&b)[KS][1]) {
    // the lane half selects between two uniform bases instead of entering the per-store address
    const FeatStore fs = relaunder(fs0);
    FeatStore f2 = fs;
    f2.roff = fs.roff + (h ? (unsigned)(4 * fs.ld2) : 0u);  // + 4 features for half 1 (needs 4*ld2 < 4 GiB)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#ifdef NFX_XP_ST32
        const u32x4 w = __builtin_bit_cast(u32x4, b[s][0]);
#pragma unroll
        for (int j = 0; j < 8; j += 2)
            st32(f2, feat0 + 32 * (s >> 1) + 16 * (s & 1) + (j & 3) + 8 * (j >> 2), w[j >> 1]);
#else
#pragma unroll
        for (int j = 0; j < 8; ++j)
            st16(f2, feat0 + 32 * (s >> 1) + 16 * (s & 1) + (j & 3) + 8 * (j >> 2), b[s][0][j]);
#endif
    }
}
// posenc slots (mlp_engine.hpp:posenc) -> logical Embedder order [x, sin f0, cos f0, ...] starting at e0
template <int L, int KS>
__device__ __forceinline__ void store_posenc(const FeatStore& fs0, int e0, int h, const bf16x8 (&b)[KS][1]) {
    const FeatStore fs = relaunder(fs0);
    FeatStore f3 = fs, f2 = fs;
    f3.roff = fs.roff + (h ? (unsigned)(3 * fs.ld2) : 0u);  // cosines sit 3 features after the sines
    f2.roff = fs.roff + (h ? (unsigned)(2 * fs.ld2) : 0u);  // x[2] sits 2 features after x[0]
#pragma unroll
    for (int q = 0; q < KS * 8; ++q) {
        if (q < 3 * L) st16(f3, e0 + 3 + 6 * (q / 3) + (q % 3), b[q >> 3][0][q & 7]);
        else if (q == 3 * L) st16(f2, e0, b[q >> 3][0][q & 7]);
        else if (q == 3 * L + 1) { if (h == 0) st16(fs, e0 + 1, b[q >> 3][0][q & 7]); }
    }
}

template <int CT>
__device__ __forceinline__ void zero_init(f32x16 (&acc)[CT]) {
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
}

// One 32-feature output tile (two k-steps lo/hi of the next B operand) -> feature-major rows feat0 .. feat0+31.
__device__ __forceinline__ void store_tile(const FeatStore& fs0, int feat0, int h, const bf16x8& lo, const bf16x8& hi) {
    const FeatStore fs = relaunder(fs0);
    FeatStore f2 = fs;
    f2.roff = fs.roff + (h ? (unsigned)(4 * fs.ld2) : 0u);
#ifdef NFX_XP_ST32
    const u32x4 wl = __builtin_bit_cast(u32x4, lo), wh = __builtin_bit_cast(u32x4, hi);
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        st32(f2, feat0 + (j & 3) + 8 * (j >> 2), wl[j >> 1]);
        st32(f2, feat0 + 16 + (j & 3) + 8 * (j >> 2), wh[j >> 1]);
    }
#else
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        st16(f2, feat0 + (j & 3) + 8 * (j >> 2), lo[j]);
        st16(f2, feat0 + 16 + (j & 3) + 8 * (j >> 2), hi[j]);
    }
#endif
}
// ReLU mask bits of a pre-activation tile: accumulator register r of tile t <-> bit 16*(t&1) + r of word t>>1
// (= B slot (2t + (r>>3), r&7) of the activation).
__device__ __forceinline__ unsigned relu_bits16(const f32x16& acc) {
    unsigned bits = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) bits |= (acc[r] > 0.f ? 1u : 0u) << r;
    asm volatile("" : "+v"(bits));  // opaque: the backward must read the BIT, not keep 2k compare results alive
    return bits;
}
template <int NW_>
__device__ __forceinline__ bool mask_bit(const unsigned (&m)[NW_], int t, int r) {
    return (m[t >> 1] >> ((t & 1) * 16 + r)) & 1u;
}

}  // namespace bwd
}  // namespace nfx
