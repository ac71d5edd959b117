// mlp_engine.hpp — register-resident fused-MLP building blocks for gfx950 (CDNA4).
//
// Formulation: every Dense layer is computed TRANSPOSED,  H^T[out, pts] = W^T[out, in] · X^T[in, pts],
// with one v_mfma_f32_32x32x16_bf16 per (32 outputs) x (32 points) x (16 inputs):
//   A operand = a 32x16 block of W^T (streamed global -> LDS -> VGPR, pre-packed on the host in
//               exactly the lane order the instruction wants: one 1 KiB "fragment" per MFMA),
//   B operand = 16 input features of 32 points, held in VGPRs,
//   C/D       = 32 output features of the same 32 points, in VGPRs/AGPRs.
// The C/D lane map of the instruction (col = lane&31 = point, row = (r&3)+8(r>>2)+4(lane>>5)) is,
// after bias+activation+bf16 conversion, *already* a valid B operand for the next layer as long
// as the next layer's weight rows are permuted accordingly — the host packer (pack.cpp) does
// that permutation, so activations never leave the register file between layers:
//   B k-step s (16 features), lane-half h, element j  <->  feature F(s,h,j) =
//       32*(s>>1) + 16*(s&1) + (j&3) + 8*(j>>2) + 4*h                        (hidden inputs)
// Points live on lanes (lane&31) for the whole network; the two lane halves hold disjoint
// feature subsets of the same 32 points.  CT column tiles (32 points each) per wave share every
// A fragment read from LDS.
//
// Weight stream: the network's fragments are stored in consumption order as "chunks" (one chunk =
// all K-steps of one 32-row output tile).  A workgroup (4 waves) walks the chunks in lock-step;
// chunk c+1 is fetched into registers while chunk c is consumed from LDS, then written to the
// other LDS slot (register-staged double buffer, one barrier per chunk).
#pragma once
#include <type_traits>
#include "nfx_common.hpp"

namespace nfx {

constexpr int kFragBytes = 1024;                  // 64 lanes x 16 B
constexpr int kSlotFrags = 24;                    // largest chunk: K = 320 -> 20 k-steps, padded to 24
constexpr int kSlotBytes = kSlotFrags * kFragBytes;
constexpr int kPieceThreads = 256;                // a "piece" = 256 lanes x 16 B = 4 KiB

// --------------------------------------------------------------------------------------
// MFMA operands written by packed 16-bit VALU instructions — a round-3 finding that is NOT a lever (profiles/HISTORY.md §2d,
// profiles/r03/nerf_first_tile/).  In the cycle-stamp build of the NeRF kernel an MFMA that reads an A / B operand
// register whose LAST WRITER was v_cvt_pk_bf16_f32 or v_pk_max_i16 — the two instructions every epilogue here ends
// with — runs at about half rate the first time it reads that register: the "first tile of every layer takes two tile
// times" signature rounds 1 and 2 chased (first tile 2600-3400 cycles against 1450-1650; 1300-1960 with the epilogue
// converting into scratch registers; 1400-1900 with every converted register re-written by a plain `v_mov_b32 v, v`,
// whole pass 131.0 k -> 123.3 k cycles; elapsed time since the write does not matter).  But the stamps' branches cut the
// scheduling regions at every tile: that build is a different, 27 % slower kernel (131 k cycles per pass against 103 k).
// In the PRODUCT schedule the same "operands never rewritten" experiment is worth 3.5 % (1518 against 1466 TFLOP/s),
// and the moves that heal it cost more than they recover: -1.6 % (v_mov_b32 per pair), -1.0 % (eight v_mov_b64 per
// tile); the light-visibility kernel -2.7 %, the density-gradient kernel -19 %.  The helpers stay as an opt-in
// (-DNFX_OPERAND_FENCE) at every place an MFMA operand is converted, for whoever re-measures on another toolchain.
// --------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned mfma_operand_dword(unsigned x) {
#ifdef NFX_OPERAND_FENCE
    asm("v_mov_b32 %0, %0" : "+v"(x));   // (not volatile: free to move between the instructions around it)
#endif
    return x;
}
__device__ __forceinline__ void mfma_operand_fence(bf16x8& v) {
#ifdef NFX_OPERAND_FENCE
    u32x4 w = __builtin_bit_cast(u32x4, v);
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = mfma_operand_dword(w[q]);
    v = __builtin_bit_cast(bf16x8, w);
#endif
}

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

struct WStream {
    const u32x4* gnext;  // next chunk to fetch
    const u32x4* gbase;  // first chunk of the network
    const u32x4* gend;   // one past the last chunk
    char* ring;          // LDS: 2 slots of kSlotBytes
    int cur;             // slot holding the chunk being consumed
};

// A chunk of NL pieces is moved by NW waves: piece i belongs to the 256-thread group (i % G),
// G = NW / 4; every thread therefore stages at most ceil(NL / G) pieces.
template <int NL, int NW>
struct Stage {
    static constexpr int G = NW / 4;
    static constexpr int N = (NL + G - 1) / G;
    u32x4 r[N];
    __device__ __forceinline__ void load(const u32x4* g, int tid) {
        const int grp = tid >> 8, t = tid & 255;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int i = k * G + grp;
            if ((k + 1) * G <= NL || i < NL) r[k] = g[i * kPieceThreads + t];
        }
    }
    __device__ __forceinline__ void store(u32x4* dst, int tid) const {
        const int grp = tid >> 8, t = tid & 255;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int i = k * G + grp;
            if ((k + 1) * G <= NL || i < NL) dst[i * kPieceThreads + t] = r[k];
        }
    }
};

// Bring the first chunk (NL x 4 KiB) into slot 0.
template <int NL, int NW>
__device__ __forceinline__ void stream_prologue(WStream& ws, int tid) {
    Stage<NL, NW> st;
    st.load(ws.gnext, tid);
    st.store(reinterpret_cast<u32x4*>(ws.ring), tid);
    ws.gnext += NL * kPieceThreads;
    ws.cur = 0;
    __syncthreads();
}

// Consume the current chunk with `compute(lds_chunk_base)` while the next chunk (NL_NEXT x 4 KiB)
// is in flight; publish it to the other slot and flip.
template <int NL_NEXT, int NW, typename F>
__device__ __forceinline__ void with_chunk(WStream& ws, int tid, F&& compute) {
    Stage<NL_NEXT, NW> st;
    st.load(ws.gnext, tid);
    compute(ws.ring + ws.cur * kSlotBytes);
    st.store(reinterpret_cast<u32x4*>(ws.ring + (ws.cur ^ 1) * kSlotBytes), tid);
    ws.gnext += NL_NEXT * kPieceThreads;
    if (ws.gnext == ws.gend) ws.gnext = ws.gbase;
    ws.cur ^= 1;
    __syncthreads();
}

// Same as with_chunk, but the size of the NEXT chunk is a run-time (wave-uniform) piece count
// n_next <= NLMAX: lets identical layers share one rolled loop body (smaller instruction footprint).
template <int NLMAX, int NW, typename F>
__device__ __forceinline__ void with_chunk_rt(WStream& ws, int tid, int n_next, F&& compute) {
    constexpr int G = NW / 4;
    constexpr int N = (NLMAX + G - 1) / G;
    u32x4 r[N];
    const int grp = tid >> 8, t = tid & 255;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int i = k * G + grp;
        if (i < n_next) r[k] = ws.gnext[i * kPieceThreads + t];
    }
    compute(ws.ring + ws.cur * kSlotBytes);
    u32x4* dst = reinterpret_cast<u32x4*>(ws.ring + (ws.cur ^ 1) * kSlotBytes);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int i = k * G + grp;
        if (i < n_next) dst[i * kPieceThreads + t] = r[k];
    }
    ws.gnext += n_next * kPieceThreads;
    if (ws.gnext == ws.gend) ws.gnext = ws.gbase;
    ws.cur ^= 1;
    __syncthreads();
}

// acc[c][r] <- bias of output row (r&3) + 8(r>>2) + 4h of this tile (bias_tile: 32 floats in LDS).
template <int CT>
__device__ __forceinline__ void bias_init(const float* bias_tile, int h, f32x16 (&acc)[CT]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(bias_tile + 8 * g + 4 * h);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            acc[c][4 * g + 0] = v[0];
            acc[c][4 * g + 1] = v[1];
            acc[c][4 * g + 2] = v[2];
            acc[c][4 * g + 3] = v[3];
        }
    }
}

// acc += A(frag 0 .. KS-1 from lane_frag0) x b[s].
// (A depth-4 software pipeline of the ds_reads pinned with sched_group_barrier was measured on r01:
//  no throughput change on any variant and 10-100x longer compiles — the LDS round trip is not
//  what limits these kernels; left to the compiler's own schedule.)
template <int KS, int KSA, int CT>
__device__ __forceinline__ void mma_k(const char* lane_frag0, const bf16x8 (&b)[KSA][CT],
                                      f32x16 (&acc)[CT]) {
    static_assert(KS <= KSA, "operand array too small");
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(lane_frag0 + s * kFragBytes);
#pragma unroll
        for (int c = 0; c < CT; ++c)
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[s][c], acc[c], 0, 0, 0);
    }
}

// One 32-row output tile: acc = init + W_tile^T [b1 ; b2].  Consumes one chunk.
// `init(acc)` fills the accumulators (bias from LDS, or a per-point pre-activation vector).
template <int KS1, int KS2, int NL_NEXT, int NW, int KS1A, int KS2A, int CT, typename Init>
__device__ __forceinline__ void tile_init(WStream& ws, int tid, Init&& init,
                                          const bf16x8 (&b1)[KS1A][CT],
                                          const bf16x8 (&b2)[KS2A][CT], f32x16 (&acc)[CT]) {
    const int lane = tid & 63;
    init(acc);
    with_chunk<NL_NEXT, NW>(ws, tid, [&](const char* chunk) {
        const char* f0 = chunk + lane * 16;
        mma_k<KS1>(f0, b1, acc);
        if constexpr (KS2 > 0) mma_k<KS2>(f0 + KS1 * kFragBytes, b2, acc);
    });
}

template <int KS1, int KS2, int NL_NEXT, int NW, int KS1A, int KS2A, int CT>
__device__ __forceinline__ void tile_raw(WStream& ws, int tid, const float* bias_tile,
                                         const bf16x8 (&b1)[KS1A][CT],
                                         const bf16x8 (&b2)[KS2A][CT], f32x16 (&acc)[CT]) {
    const int h = (tid & 63) >> 5;
    tile_init<KS1, KS2, NL_NEXT, NW>(
        ws, tid, [&](f32x16(&a)[CT]) { bias_init<CT>(bias_tile, h, a); }, b1, b2, acc);
}

// acc[c][r] <- vec[(r&3) + 8(r>>2) + 4h] from a GLOBAL fp32 vector of 32 values per tile (all lanes
// of a half read the same 16 B: one broadcast transaction each).
template <int CT>
__device__ __forceinline__ void vec_init(const float* __restrict__ vec_tile, int h,
                                         f32x16 (&acc)[CT]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(vec_tile + 8 * g + 4 * h);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            acc[c][4 * g + 0] = v[0];
            acc[c][4 * g + 1] = v[1];
            acc[c][4 * g + 2] = v[2];
            acc[c][4 * g + 3] = v[3];
        }
    }
}

template <bool RELU, int CT>
__device__ __forceinline__ void acc_to_b(const f32x16 (&acc)[CT], bf16x8 (&lo)[CT],
                                         bf16x8 (&hi)[CT]) {
#pragma unroll
    for (int c = 0; c < CT; ++c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v0 = acc[c][j], v1 = acc[c][8 + j];
            if (RELU) {  // one v_med3_f32 each (fmaxf would add a canonicalising v_max)
                v0 = __builtin_amdgcn_fmed3f(v0, 0.0f, __builtin_inff());
                v1 = __builtin_amdgcn_fmed3f(v1, 0.0f, __builtin_inff());
            }
            lo[c][j] = (__bf16)v0;
            hi[c][j] = (__bf16)v1;
        }
        mfma_operand_fence(lo[c]);
        mfma_operand_fence(hi[c]);
    }
}

// A whole Dense layer: NT output tiles (32 rows each) from inputs [b1 ; b2]; the result is the
// next layer's B operand (2 k-steps per tile).  NL_SELF / NL_NEXT: 4-KiB pieces per thread of
// this layer's chunks / of the chunk that follows the layer's last one.
template <int KS1, int KS2, int NT, int NL_SELF, int NL_NEXT, bool RELU, int NW, int KS1A,
          int KS2A, int NTA, int CT>
__device__ __forceinline__ void layer(WStream& ws, int tid, const float* bias,
                                      const bf16x8 (&b1)[KS1A][CT],
                                      const bf16x8 (&b2)[KS2A][CT], bf16x8 (&bout)[NTA][CT]) {
    static_assert(2 * NT <= NTA, "output array too small");
    static_for<0, NT>([&](auto T) {
        constexpr int t = decltype(T)::value;
        f32x16 acc[CT];
        tile_raw<KS1, KS2, (t == NT - 1 ? NL_NEXT : NL_SELF), NW>(ws, tid, bias + 32 * t, b1, b2,
                                                                 acc);
        acc_to_b<RELU, CT>(acc, bout[2 * t], bout[2 * t + 1]);
    });
}

// layer() with run-time chunk sizes: nl_self pieces per chunk of this layer, nl_next for the chunk
// after the layer's last tile (both wave-uniform, <= NLMAX).
template <int KS1, int KS2, int NT, int NLMAX, bool RELU, int NW, int KS1A, int KS2A, int NTA, int CT>
__device__ __forceinline__ void layer_rt(WStream& ws, int tid, const float* bias, int nl_self, int nl_next,
                                         const bf16x8 (&b1)[KS1A][CT], const bf16x8 (&b2)[KS2A][CT],
                                         bf16x8 (&bout)[NTA][CT]) {
    static_assert(2 * NT <= NTA, "output array too small");
    const int lane = tid & 63;
    static_for<0, NT>([&](auto T) {
        constexpr int t = decltype(T)::value;
        f32x16 acc[CT];
        bias_init<CT>(bias + 32 * t, lane >> 5, acc);
        with_chunk_rt<NLMAX, NW>(ws, tid, t == NT - 1 ? nl_next : nl_self, [&](const char* chunk) {
            const char* f0 = chunk + lane * 16;
            mma_k<KS1>(f0, b1, acc);
            if constexpr (KS2 > 0) mma_k<KS2>(f0 + KS1 * kFragBytes, b2, acc);
        });
        acc_to_b<RELU, CT>(acc, bout[2 * t], bout[2 * t + 1]);
    });
}

// Same, but the accumulators start from a per-point pre-activation vector in global memory
// (`pre` points at this layer's first value for this wave's point) instead of the bias.
template <int KS1, int KS2, int NT, int NL_SELF, int NL_NEXT, bool RELU, int NW, int KS1A,
          int KS2A, int NTA, int CT>
__device__ __forceinline__ void layer_pre(WStream& ws, int tid, const float* __restrict__ pre,
                                          const bf16x8 (&b1)[KS1A][CT],
                                          const bf16x8 (&b2)[KS2A][CT], bf16x8 (&bout)[NTA][CT]) {
    static_assert(2 * NT <= NTA, "output array too small");
    const int h = (tid & 63) >> 5;
    static_for<0, NT>([&](auto T) {
        constexpr int t = decltype(T)::value;
        f32x16 acc[CT];
        tile_init<KS1, KS2, (t == NT - 1 ? NL_NEXT : NL_SELF), NW>(
            ws, tid, [&](f32x16(&a)[CT]) { vec_init<CT>(pre + 32 * t, h, a); }, b1, b2, acc);
        acc_to_b<RELU, CT>(acc, bout[2 * t], bout[2 * t + 1]);
    });
}

// --------------------------------------------------------------------------------------
// Positional encoding straight into B-operand registers.  Slot map (shared with pack.cpp):
// for an L-band encoder of a 3-vector, each lane half owns NQ = ceil((3L+2)/8)*8 slots
// q = 8*s + j (k-step s, element j):
//     q < 3L      : half 0 -> sin(2^(q/3) x[q%3]),  half 1 -> cos(2^(q/3) x[q%3])
//     q = 3L      : half 0 -> x[0],                 half 1 -> x[2]
//     q = 3L + 1  : half 0 -> x[1],                 half 1 -> 0
//     else        : 0
// so the two halves split the 6L+3 features with no duplicated transcendental work.
// --------------------------------------------------------------------------------------
template <int L>
struct PeSlots {
    static constexpr int kKS = (3 * L + 2 + 7) / 8;  // k-steps: L=10 -> 4, L=4 -> 2, L=2 -> 1
};

template <int L, int CT>
__device__ __forceinline__ void posenc(const float (&x)[3], int h, int c,
                                       bf16x8 (&out)[PeSlots<L>::kKS][CT]) {
    constexpr int NQ = PeSlots<L>::kKS * 8;
    float v[NQ];
#pragma unroll
    for (int q = 3 * L; q < NQ; ++q) v[q] = q == 3 * L ? (h ? x[2] : x[0]) : (q == 3 * L + 1 && !h) ? x[1] : 0.0f;
    if constexpr (L <= 4) {
        // unit vectors (|arg| <= 8): v_sin_f32 / v_cos_f32, 7e-7 abs error measured on [-8.5, 8.5]
#pragma unroll
        for (int q = 0; q < 3 * L; ++q) v[q] = sin_shifted_small(x[q % 3] * (float)(1 << (q / 3)), h);
    } else {
        // positions (|arg| up to ~3000): a Cody-Waite sin/cos pair every 5th band, the four bands after it by angle
        // doubling (sin 2a = 2 sin a cos a, cos 2a = 1 - 2 sin^2 a; the error doubles per step: <= 16 x 1e-7).
        // A third of the instructions of one reduction per band.
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int k0 = 0; k0 < L; k0 += 5) {
                float sn, cs;
                sincos_cw(x[d] * (float)(1 << k0), sn, cs);
#pragma unroll
                for (int k = k0; k < k0 + 5 && k < L; ++k) {
                    v[3 * k + d] = h ? cs : sn;
                    const float s2 = 2.0f * sn * cs;
                    cs = fmaf(-2.0f * sn, sn, 1.0f);
                    sn = s2;
                }
            }
    }
#pragma unroll
    for (int s = 0; s < PeSlots<L>::kKS; ++s) {
#pragma unroll
        for (int j = 0; j < 8; ++j) out[s][c][j] = (__bf16)v[8 * s + j];
        mfma_operand_fence(out[s][c]);
    }
}

}  // namespace nfx
