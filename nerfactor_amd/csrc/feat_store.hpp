// feat_store.hpp — bf16 stores of register-resident activations / gradients for the weight-gradient GEMMs (train.hip),
// shared by the fused backward kernels (mlp128_bwd.hip, nerf_bwd.hip, brdf_bwd.hip).
//
// Layout (round 3): FEATURE-PAIR-major, [pair = feature >> 1][row][feature & 1] bf16 — a dword holds two adjacent
// features of one row.  That is what a B-operand register already is (v_cvt_pk_bf16_f32 packs features f, f + 1 of the
// lane's point), so a hidden activation leaves the wave as ONE dword store per feature pair: 256 contiguous bytes per
// wave-instruction and half as many of them.  Rounds 1-2 stored [feature][row] with one 16-bit store per value: the
// backward kernels were bound by the issue of those sub-dword stores (564 per 32-row tile; a timing experiment with
// dword stores took the NeRF training step from 3.26 to 2.71 ms before any kernel read the new layout).  The weight-
// gradient kernels split the pairs again while they stage their operands (train.hip); the bytes moved are the same.
// A feature offset handed to the GEMMs must be even (all of them are: mlp128_bwd.hip Geo, nerf_train_layout.hpp).
#pragma once
#include "mlp_engine.hpp"
// Non-temporal stores (r03): the backward kernels write 1-2 GB of activations per call THROUGH the L2 their weight
// stream lives in; with plain stores the weight lines are evicted all the time (nerf_bwd ring kernel: TCP->TCC read
// latency 789 cycles, 379 with `nt`; kernel 558 -> 515 us).  -DNFX_FEAT_NT=0 for the A/B.
#ifndef NFX_FEAT_NT
#define NFX_FEAT_NT 1
#endif

namespace nfx {
namespace bwd {

// wave-uniform base (SGPR pair, global_store saddr form) + 32-bit lane offset.
// `ld2` = 2 * ld = bytes per FEATURE (a pair row is 2 * ld2 bytes); it is laundered through an empty asm once per tile
// so the ~600 per-pair bases are computed next to their stores instead of being hoisted and spilled.
struct FeatStore {
    char* base;
    unsigned long long ld2;
    unsigned roff;  // row * 4: the row's dword inside a pair row (callers add 4 * ld2 per 4 features for lane half 1)
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
// explicit global address space below: the laundered integer base would otherwise make these flat stores with a
// 64-bit VGPR address; this form is global_store_{short,dword} v_off, v_data, s[base:base+1]
__device__ __forceinline__ unsigned long long pair_base(const FeatStore& fs, int feat) {
    return reinterpret_cast<unsigned long long>(fs.base) + (unsigned long long)(feat >> 1) * (2 * fs.ld2);
}
// one value: feature `feat` of this lane's row (lane offset `loff` relative to the pair row, default the plain row)
__device__ __forceinline__ void st16_at(const FeatStore& fs, int feat, unsigned loff, __bf16 v) {
    typedef __attribute__((address_space(1))) __bf16* gbf16_ptr;
#if NFX_FEAT_NT
    __builtin_nontemporal_store(v, (gbf16_ptr)((__attribute__((address_space(1))) char*)pair_base(fs, feat) + loff));
#else
    *(gbf16_ptr)((__attribute__((address_space(1))) char*)pair_base(fs, feat) + loff) = v;
#endif
}
__device__ __forceinline__ void st16(const FeatStore& fs, int feat, __bf16 v) {
    st16_at(fs, feat, fs.roff + 2u * (unsigned)(feat & 1), v);
}
// feature `fa` from the lanes of half 0, feature `fb` from half 1 (any two features; 2 extra VALU per store).
// The lane offset is unsigned (saddr form): both are taken relative to the pair row of the smaller feature.
__device__ __forceinline__ void st16_ab(const FeatStore& fs, int fa, int fb, int h, __bf16 v) {
    const int f0 = fa < fb ? fa : fb;
    const unsigned offa = (unsigned)((unsigned long long)((fa >> 1) - (f0 >> 1)) * (2 * fs.ld2)) + 2u * (unsigned)(fa & 1);
    const unsigned offb = (unsigned)((unsigned long long)((fb >> 1) - (f0 >> 1)) * (2 * fs.ld2)) + 2u * (unsigned)(fb & 1);
    st16_at(fs, f0, fs.roff + (h ? offb : offa), v);
}
// the dword of an EVEN feature and its successor
__device__ __forceinline__ void st32(const FeatStore& fs, int feat_even, unsigned v) {
    typedef __attribute__((address_space(1))) unsigned* gu32_ptr;
#if NFX_FEAT_NT
    __builtin_nontemporal_store(v, (gu32_ptr)((__attribute__((address_space(1))) char*)pair_base(fs, feat_even) + fs.roff));
#else
    *(gu32_ptr)((__attribute__((address_space(1))) char*)pair_base(fs, feat_even) + fs.roff) = v;
#endif
}
// B-operand registers of a hidden activation (k-step s, element j <-> feature F(s,h,j)): dword j / 2 = features
// (F, F + 1) with F even.  `feat0` even.
template <int KS>
__device__ __forceinline__ void store_hidden(const FeatStore& fs0, int feat0, int h, const bf16x8 (&b)[KS][1]) {
    // the lane half selects between two uniform bases instead of entering the per-store address
    const FeatStore fs = relaunder(fs0);
    FeatStore f2 = fs;
    f2.roff = fs.roff + (h ? (unsigned)(4 * fs.ld2) : 0u);  // + 4 features = + 2 pair rows for half 1 (needs 4*ld2 < 4 GiB)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const u32x4 w = __builtin_bit_cast(u32x4, b[s][0]);
#pragma unroll
        for (int j = 0; j < 8; j += 2)
            st32(f2, feat0 + 32 * (s >> 1) + 16 * (s & 1) + (j & 3) + 8 * (j >> 2), w[j >> 1]);
    }
}
// posenc slots (mlp_engine.hpp:posenc) -> logical Embedder order [x, sin f0, cos f0, ...] starting at e0 (any parity):
// half 0 holds the sines (feature fa), half 1 the cosines (fa + 3); x[0] / x[2] sit two features apart.  The lane
// offsets of the two parities of fa are formed once: relative to fa's pair row,
//   fa even: half 0 -> +0,  half 1 (fa + 3, odd,  next pair row)      -> 2 ld2 + 2
//   fa odd : half 0 -> +2,  half 1 (fa + 3, even, two pair rows on)   -> 4 ld2
template <int L, int KS>
__device__ __forceinline__ void store_posenc(const FeatStore& fs0, int e0, int h, const bf16x8 (&b)[KS][1]) {
    const FeatStore fs = relaunder(fs0);
    const unsigned off_e = fs.roff + (h ? (unsigned)(2 * fs.ld2) + 2u : 0u);
    const unsigned off_o = fs.roff + (h ? (unsigned)(4 * fs.ld2) : 2u);
    const unsigned off_x = fs.roff + 2u * (unsigned)(e0 & 1) + (h ? (unsigned)(2 * fs.ld2) : 0u);   // x[2] = e0 + 2: same parity
#pragma unroll
    for (int q = 0; q < KS * 8; ++q) {
        if (q < 3 * L) {
            const int fa = e0 + 3 + 6 * (q / 3) + (q % 3);
            st16_at(fs, fa, (fa & 1) ? off_o : off_e, b[q >> 3][0][q & 7]);
        } else if (q == 3 * L) {
            st16_at(fs, e0, off_x, b[q >> 3][0][q & 7]);
        } else if (q == 3 * L + 1) {
            if (h == 0) st16(fs, e0 + 1, b[q >> 3][0][q & 7]);
        }
    }
}

template <int CT>
__device__ __forceinline__ void zero_init(f32x16 (&acc)[CT]) {
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
}

// One 32-feature output tile (two k-steps lo/hi of the next B operand) -> features feat0 .. feat0+31 (feat0 even).
__device__ __forceinline__ void store_tile(const FeatStore& fs0, int feat0, int h, const bf16x8& lo, const bf16x8& hi) {
    const FeatStore fs = relaunder(fs0);
    FeatStore f2 = fs;
    f2.roff = fs.roff + (h ? (unsigned)(4 * fs.ld2) : 0u);
    const u32x4 wl = __builtin_bit_cast(u32x4, lo), wh = __builtin_bit_cast(u32x4, hi);
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        st32(f2, feat0 + (j & 3) + 8 * (j >> 2), wl[j >> 1]);
        st32(f2, feat0 + 16 + (j & 3) + 8 * (j >> 2), wh[j >> 1]);
    }
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
