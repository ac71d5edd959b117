// nerf_mlp_v6.hip — variant 6 of the fused NeRF MLP: variant 5 (one wave per SIMD, 64 points per wave, epilogue of
// tile i-1 in the shadow of tile i's MFMAs) with the serialisation at the tile boundary removed.  The r01 ablation of
// variant 5 (profiles/r01/ablation_variant5.log) showed its non-MFMA stream alone takes longer than the MFMAs:
// ds_write -> s_waitcnt lgkmcnt(0) -> s_barrier -> bias/A ds_read -> first MFMA is ~300 dead cycles per tile.  Here:
//   * 3-slot LDS ring, chunk k lives in slot k % 3 (78 = 3 x 26 chunks): chunk k+2 is fetched global -> VGPR at the
//     start of tile k and written to its slot at the END of the tile; nobody waits for those writes (they are a full
//     tile old when first read),
//   * the first three A fragments and the bias of tile k+1 are read BEFORE the end-of-tile barrier (chunk k+1 has been
//     in LDS for a whole tile), so their latency overlaps the tail MFMAs and the barrier,
//   * the barrier is a bare s_barrier (no lgkmcnt(0) drain).
// Same blob, bit-identical results to variants 0-5.
//
// Three weight-stream modes of the same kernel (template parameter DMA), all built in MFMA VGPR form (build.py):
//   DMA = 0  variant 6   register-staged as described above;
//   DMA = 1  variant 7   THE DEFAULT: LDS-DMA (global_load_lds_dwordx4) into a 6-slot ring, chunk k+3 issued at the
//                        start of tile k, counted s_waitcnt vmcnt before the barrier — no staging VGPRs, no ds_write;
//   DMA = 2  variant 8   register-staged with two staging sets (chunk k+3 fetched during tile k, stored a tile later).
// The r01 experiment switches (second bias read group, A-prefetch depth, epilogue start offset, DMA fetch distance,
// DMA pieces spread over the k-steps) were all measured and left at the values now hard-wired here — profiles/HISTORY.md
// section 2 has the numbers.  The cycle-stamp build (NFX_V6_TIMING), its experiment masks (NFX_V6_XP, NFX_V6_FENCE)
// and the ablation instantiations (NFX_ABLATION_BUILD) that produced the r01-r03 measurements of profiles/HISTORY.md sections
// 2 / 2d left this file in r04 (git history: fdbd16f holds them): the shipped translation unit is the product kernel.
#include "mlp_engine.hpp"
#include "lds_dma.hpp"
#include "nerf_layout.hpp"


// NFX_V6_SIGMA (nerf_sigma_v6.hip includes this file with it defined): the DENSITY-ONLY form of the same dataflow over the
// GEOM blob (nerf_geom_layout.hpp: chunks 0..63 = the encoder exactly as here, chunk 64 = the sigma tile, chunk 65 = the
// first reverse-sweep chunk, fetched and not multiplied: the weight sequence must be a multiple of the 6-slot ring) — the
// tile / layer machinery below is shared, the kernel and the chunk count differ.
#ifdef NFX_V6_SIGMA
#define NFX_V6_NS v6s
#else
#define NFX_V6_NS v6
#endif

namespace nfx {
namespace NFX_V6_NS {

constexpr int kNW = 4, kCT = 2;
// ring size / fetch distance: register-staged 3 slots, chunk K+2 fetched during tile K; LDS-DMA 6 slots (78 = 6 x 13),
// chunk K+3 issued during tile K and awaited at the end of tile K: when tile K+1 prefetches the head of chunk K+2
// before ITS barrier, every wave's pieces of that chunk have been behind a barrier already
// DMA = 2 (variant 8): register-staged like 0, but chunk K+3 is fetched during tile K into one of TWO register sets and
// written to its slot at the end of tile K+1: the global loads get two tile times to land instead of one, and the
// compiler's counted vmcnt lets the newer set stay in flight across the store of the older one.
template <int DMA> constexpr int ring_of = DMA == 1 ? 6 : 3;
constexpr int kDmaDist = 3;   // LDS-DMA fetch distance in tiles (4 measured the same; the 6-slot ring holds either)
template <int DMA> constexpr int dist_of = DMA == 1 ? kDmaDist : DMA ? 3 : 2;
template <int DMA> constexpr int lds_of = ring_of<DMA> * kSlotBytes + nerf::kBiasFloats * 4;
#ifdef NFX_V6_SIGMA
constexpr int kNChunks = 66;              // 64 encoder chunks + the sigma tile + one idle chunk = 6 x 11
#else
constexpr int kNChunks = nerf::kNChunks;  // 78
#endif
static_assert(kNChunks % 6 == 0, "the chunk sequence wraps on the 6-slot (and the 3-slot) ring");

constexpr int kPreA = 3;   // A fragments in flight ahead of their MFMAs (2 / 3 / 4 measured: 1373 / 1372 / 1363 TFLOP/s)

struct Acc {
    f32x16 v[kCT];
};
struct Pre {
    bf16x8 a[kPreA];  // first A fragments of the next tile
};

template <bool RELU>
__device__ __forceinline__ void cvt_pair(float v0, float v1, bf16x8& dst, int j) {
    typedef short s2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 vv = {v0, v1};
    b2 pr = __builtin_convertvector(vv, b2);  // one v_cvt_pk_bf16_f32
    if (RELU) {                               // a negative bf16 is a negative int16: ReLU = one v_pk_max_i16
        s2 w = __builtin_bit_cast(s2, pr);
        const s2 z = {0, 0};
        w = __builtin_elementwise_max(w, z);
        pr = __builtin_bit_cast(b2, w);
    }
    dst[j] = pr[0];
    dst[j + 1] = pr[1];
}

template <bool RELU>
struct EpiB {
    const Acc& acc;
    bf16x8 (&lo)[kCT];
    bf16x8 (&hi)[kCT];
    template <int R0, int R1>
    __device__ __forceinline__ void run() {
#pragma unroll
        for (int r = R0; r < R1; r += 2)
#pragma unroll
            for (int c = 0; c < kCT; ++c) {
                if (r < 8) cvt_pair<RELU>(acc.v[c][r], acc.v[c][r + 1], lo[c], r);
                else cvt_pair<RELU>(acc.v[c][r], acc.v[c][r + 1], hi[c], r - 8);
            }
    }
    __device__ __forceinline__ void finish() {
    }
};
struct EpiNone {
    template <int R0, int R1>
    __device__ __forceinline__ void run() {}
    __device__ __forceinline__ void finish() {}
};
struct EpiSigma {
    const Acc& acc;
    float (&sigma)[kCT];
    template <int R0, int R1>
    __device__ __forceinline__ void run() {
        if constexpr (R0 == 0) {
#pragma unroll
            for (int c = 0; c < kCT; ++c) sigma[c] = acc.v[c][0];
        }
    }
    __device__ __forceinline__ void finish() {}
};

__device__ __forceinline__ void bias_to_acc(const float* bias_tile, int lane, Acc& acc) {
    // one broadcast read group, the second column tile's accumulators copied from the first: +1.8 % on r01 once the
    // accumulators live in ArchVGPRs (MFMA VGPR form); a second read group (NFX_V6_BIAS_READ2) costs a full LDS pass
    const float* bt = bias_tile + 4 * (lane >> 5);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(bt + 8 * g);
#pragma unroll
        for (int c = 0; c < kCT; ++c) {
            acc.v[c][4 * g + 0] = v[0];
            acc.v[c][4 * g + 1] = v[1];
            acc.v[c][4 * g + 2] = v[2];
            acc.v[c][4 * g + 3] = v[3];
        }
    }
}

typedef __attribute__((address_space(1))) u32x4 gu32x4;   // explicit global address space: global_load, not flat_load

struct Regs {
    u32x4 r[2][6];   // two staging sets of up to six 4-KiB pieces (variant 8)
};

struct Ctx {
    char* smem;
    const char* blob;
    int tid;
    unsigned smem_lds;   // LDS byte address of smem (for M0)
    int wave;            // wave-uniform
};

// DMA = 1: the weight stream goes global -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave-instruction, no VGPR
// staging, no ds_write, no lgkmcnt drain); completion is tracked with counted s_waitcnt vmcnt (the pieces of chunk
// K+2 must have landed before the barrier that ends tile K+1, those of chunk K+3 may still be in flight).
// r03: all pieces of a chunk in ONE statement — M0 is saved, set and restored once, and the pieces are addressed
// through the instruction's immediate offset, which the hardware adds to the global AND to the LDS address (the pieces of a
// wave are 1 KiB apart in both).  The r02 form issued every piece as its own statement: 8 instructions per piece (two
// 64-bit address adds, one LDS address add, M0 save / set / s_nop / restore) = 32-48 scalar instructions in the first two
// MFMA gaps of every tile, where a lone wave can hide five (MI355X_MICROARCH.md).  Only the fragments a tile really
// multiplies are fetched (the chunks of layer 0, layer 5 and rgb_out[0] are padded to 8 / 24 fragments in the blob):
// 298 pieces per pass instead of 318.
constexpr int used_frags(int k) { return k < 8 ? 4 : k < 40 ? 16 : k < 48 ? 20 : k < 73 ? 16 : k < 77 ? 18 : 8; }
constexpr int dma_pieces(int k) { return (used_frags(k) + kNW - 1) / kNW; }   // 1-KiB pieces per wave: 1 | 4 | 5 | 2
template <int K>
__device__ __forceinline__ void dma_chunk(const Ctx& cx) {
    constexpr int n = dma_pieces(K);
    unsigned long long base = reinterpret_cast<unsigned long long>(cx.blob);
    unsigned lds = cx.smem_lds;
    asm volatile("" : "+s"(base), "+s"(lds));       // per tile: keeps the piece addresses out of the loop preheader
    const int piece0 = cx.wave * n;
    const char* g = reinterpret_cast<const char*>(base) + (size_t)nerf::chunk_frag_offset(K) * kFragBytes + piece0 * 1024;
    const unsigned l = lds + (K % 6) * kSlotBytes + piece0 * 1024;
    lds_dma_pieces<n>((cx.tid & 63) * 16, g, l);
}
// Tile K (global chunk index).  On entry `acc` holds the tile's bias and `pre` its first three A fragments; on exit
// `acc_next` / `pre` hold the same for tile K+1 (bias from `next_bias`).  AB: timing-only ablation mask
// (1 no weight staging, 2 no barrier, 4 no MFMA, 8 no A reads, 64 no bias reads).
template <int K, int KS1, int KS2, int AB, int DMA, int KS1A, int KS2A, typename Epi>
__device__ __forceinline__ void tile(const Ctx& cx, Regs& rg, const float* next_bias, const bf16x8 (&b1)[KS1A][kCT],
                                     const bf16x8 (&b2)[KS2A][kCT], Acc& acc, Acc& acc_next, Pre& pre, Epi&& prev) {
    constexpr int KS = KS1 + KS2;
    constexpr int PIECES = KS >= 16 ? 8 : 4;
    // (starting the previous tile's epilogue two k-steps into the tile, behind its first MFMAs, measured no gain on r01)
    constexpr int EOFF = 0;
    constexpr int SP = (PIECES < KS ? PIECES : KS - 1) + EOFF;  // k-step after which the previous tile's epilogue is complete
    constexpr int R = ring_of<DMA>;
    constexpr int K1 = (K + 1) % kNChunks, K2 = (K + dist_of<DMA>) % kNChunks;   // K2: the chunk fetched during this tile
    constexpr int NL2 = nerf::chunk_frags(K2) / 4;
    const int lane = cx.tid & 63;
    const char* f0 = cx.smem + (K % R) * kSlotBytes + lane * 16;
    Stage<DMA ? 1 : NL2, kNW> st;
    if constexpr (DMA == 1 && !(AB & 1)) {
        dma_chunk<K2>(cx);
    } else if constexpr (DMA == 2 && !(AB & 1)) {
        unsigned long long gb = reinterpret_cast<unsigned long long>(cx.blob);
        asm volatile("" : "+s"(gb));   // (an integer: a laundered generic pointer would turn the loads into flat_load)
        const gu32x4* g = reinterpret_cast<const gu32x4*>(gb + (size_t)nerf::chunk_frag_offset(K2) * kFragBytes);
#pragma unroll
        for (int k = 0; k < NL2; ++k) rg.r[K & 1][k] = g[k * kPieceThreads + cx.tid];
    } else if constexpr (!(AB & 1)) {
        // opaque per tile: otherwise the ~300 loop-invariant chunk addresses are hoisted out of the point-tile loop
        // and spilled (same cure as variant 2)
        unsigned long long gb = reinterpret_cast<unsigned long long>(cx.blob);
        asm volatile("" : "+s"(gb));
        const gu32x4* g = reinterpret_cast<const gu32x4*>(gb + (size_t)nerf::chunk_frag_offset(K2) * kFragBytes);
#pragma unroll
        for (int k = 0; k < NL2; ++k) st.r[k] = g[k * kPieceThreads + cx.tid];   // (kNW = 4: one piece group)
    }
    bf16x8 abuf[kPreA + 1];
#pragma unroll
    for (int i = 0; i < kPreA; ++i) abuf[i] = pre.a[i];
    static_for<0, KS>([&](auto S) {
        constexpr int s = decltype(S)::value;
        if constexpr (s + kPreA < KS && !(AB & 8))
            abuf[(s + kPreA) % (kPreA + 1)] = *reinterpret_cast<const bf16x8*>(f0 + (s + kPreA) * kFragBytes);
        const bf16x8 a = abuf[s % (kPreA + 1)];
#pragma unroll
        for (int c = 0; c < kCT; ++c) {
            const bf16x8 b = s < KS1 ? b1[s < KS1 ? s : 0][c] : b2[s >= KS1 ? s - KS1 : 0][c];
            if constexpr (AB & 4) {
                asm volatile("" ::"v"(a), "v"(b));
            } else {
                acc.v[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc.v[c], 0, 0, 0);
            }
        }
        if constexpr (EOFF > 0 && s == EOFF)
            // neither VALU nor MFMA may cross this point (SALU, VMEM, DS may): without it the scheduler puts the
            // epilogue's first reads directly behind the previous tile's last MFMAs again
            __builtin_amdgcn_sched_barrier(0x4 | 0x10 | 0x80);
        if constexpr (s >= EOFF && s - EOFF < PIECES)
            prev.template run<16 * (s - EOFF) / PIECES, 16 * (s - EOFF + 1) / PIECES>();
        if constexpr (s == SP) {
            // the other accumulator set is free now: tile K+1's bias goes to its accumulators
            if constexpr (!(AB & 64)) bias_to_acc(next_bias, lane, acc_next);
            prev.finish();
        }
    });
    // chunk K+2 to its slot as late as possible (its global loads had the whole tile to land; measured: storing at
    // mid-tile stalls on vmcnt, L2 latency under this load exceeds half a tile)
    if constexpr (DMA == 1 && !(AB & 1)) {
        // the chunk issued one tile ago must be complete before the barrier; this tile's pieces may stay in flight
        // (fetch distance 4: the chunk issued during the previous tile may stay in flight too)
        constexpr int kInFlight = dma_pieces(K2) + (kDmaDist == 4 ? dma_pieces((K + 3) % kNChunks) : 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kInFlight) : "memory");
    } else if constexpr (DMA == 2 && !(AB & 1)) {
        // chunk K+2, fetched during tile K-1 into the other register set, to slot (K+2) % 3 = the slot tile K-1 read
        constexpr int KP = (K + 2) % kNChunks, NLP = nerf::chunk_frags(KP) / 4;
        u32x4* dst = reinterpret_cast<u32x4*>(cx.smem + (KP % R) * kSlotBytes);
#pragma unroll
        for (int k = 0; k < NLP; ++k) dst[k * kPieceThreads + cx.tid] = rg.r[(K + 1) & 1][k];
    } else if constexpr (!(AB & 1)) {
        st.store(reinterpret_cast<u32x4*>(cx.smem + (K2 % R) * kSlotBytes), cx.tid);
    }
    if constexpr (!(AB & 8)) {
        const char* f1 = cx.smem + (K1 % R) * kSlotBytes + lane * 16;
#pragma unroll
        for (int i = 0; i < kPreA; ++i) pre.a[i] = *reinterpret_cast<const bf16x8*>(f1 + i * kFragBytes);
    }
    // (a __builtin_amdgcn_sched_barrier(0) here costs 4 %: it stops the scheduler from draining the tail MFMAs of this
    //  tile behind the barrier; the per-tile laundering of the blob base above is what keeps the weight addresses
    //  from being hoisted)
    if constexpr (!(AB & 2)) asm volatile("s_barrier" ::: "memory");
}

// A Dense layer of NT tiles starting at chunk K0, outputs to bout.  `prev0` = pending epilogue of tile K0-1;
// `next_bias` = bias of the tile after this layer's last one.  On return the last tile's epilogue is pending.
template <int K0, int KS1, int KS2, int NT, bool RELU, int AB, int DMA, int KS1A, int KS2A, int NTA, typename Epi0>
__device__ __forceinline__ void layer(const Ctx& cx, Regs& rg, const float* bias, const float* next_bias,
                                      const bf16x8 (&b1)[KS1A][kCT], const bf16x8 (&b2)[KS2A][kCT],
                                      bf16x8 (&bout)[NTA][kCT], Acc (&accs)[2], Pre& pre, Epi0&& prev0) {
    static_for<0, NT>([&](auto T) {
        constexpr int t = decltype(T)::value;
        constexpr int K = K0 + t;
        const float* nb = t == NT - 1 ? next_bias : bias + 32 * (t + 1);
        if constexpr (t == 0) {
            tile<K, KS1, KS2, AB, DMA>(cx, rg, nb, b1, b2, accs[K & 1], accs[(K + 1) & 1], pre, prev0);
        } else {
            EpiB<RELU> e{accs[(K - 1) & 1], bout[2 * (t - 1)], bout[2 * (t - 1) + 1]};
            tile<K, KS1, KS2, AB, DMA>(cx, rg, nb, b1, b2, accs[K & 1], accs[(K + 1) & 1], pre, e);
        }
    });
}

#ifndef NFX_V6_SIGMA
template <int AB, int DMA>
__global__ __launch_bounds__(kNW * 64, 1) void nerf_mlp_bf16_v6_kernel(
    const float* __restrict__ rayo, const float* __restrict__ rayd, const float* __restrict__ zbuf, long long n_pts,
    int n_samples, const char* __restrict__ blob, float4* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using namespace nerf;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
    constexpr int kTilePts = kNW * 32 * kCT;
    float* bias_lds = reinterpret_cast<float*>(smem + ring_of<DMA> * kSlotBytes);
    {
        const float* bsrc = reinterpret_cast<const float*>(blob + kWeightBytes);
        for (int i = tid; i < kBiasFloats; i += kNW * 64) bias_lds[i] = bsrc[i];
    }
    typedef __attribute__((address_space(3))) char lds_char;
    Ctx cx{smem, blob, tid, (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_char*)smem),
           __builtin_amdgcn_readfirstlane(tid >> 6)};
    Acc accs[2];
    Pre pre;
    Regs rg;
    if constexpr (DMA == 2) {   // chunk 2 plays "fetched during tile -1": register set 1, stored at the end of tile 0
        const u32x4* g = reinterpret_cast<const u32x4*>(blob + (size_t)chunk_frag_offset(2) * kFragBytes);
#pragma unroll
        for (int k = 0; k < chunk_frags(2) / 4; ++k) rg.r[1][k] = g[k * kPieceThreads + tid];
    }
    {   // chunks 0 and 1 -> slots 0 and 1
        Stage<chunk_frags(0) / 4, kNW> s0;
        Stage<chunk_frags(1) / 4, kNW> s1;
        s0.load(reinterpret_cast<const u32x4*>(blob), tid);
        s1.load(reinterpret_cast<const u32x4*>(blob + (size_t)chunk_frag_offset(1) * kFragBytes), tid);
        s0.store(reinterpret_cast<u32x4*>(smem), tid);
        s1.store(reinterpret_cast<u32x4*>(smem + kSlotBytes), tid);
        if constexpr (DMA == 1) {   // fetch distance 3: chunk 2 must be resident before the first tile as well
            Stage<chunk_frags(2) / 4, kNW> s2;
            s2.load(reinterpret_cast<const u32x4*>(blob + (size_t)chunk_frag_offset(2) * kFragBytes), tid);
            s2.store(reinterpret_cast<u32x4*>(smem + 2 * kSlotBytes), tid);
            if constexpr (kDmaDist == 4) {
                Stage<chunk_frags(3) / 4, kNW> s3;
                s3.load(reinterpret_cast<const u32x4*>(blob + (size_t)chunk_frag_offset(3) * kFragBytes), tid);
                s3.store(reinterpret_cast<u32x4*>(smem + 3 * kSlotBytes), tid);
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kPreA; ++i) pre.a[i] = *reinterpret_cast<const bf16x8*>(smem + lane * 16 + i * kFragBytes);
        bias_to_acc(bias_lds + kBiasL0, lane, accs[0]);
    }
    const long long n_tiles = (n_pts + kTilePts - 1) / kTilePts;
    for (long long tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
        bf16x8 pe[4][kCT], pv[2][kCT];
        long long m[kCT];
#pragma unroll
        for (int c = 0; c < kCT; ++c) {
            m[c] = tl * kTilePts + wave * (32 * kCT) + c * 32 + p;
            const long long mm = m[c] < n_pts ? m[c] : n_pts - 1;
            const long long ray = mm / n_samples;
            const float zz = zbuf[mm];
            float x[3], d[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                d[k] = rayd[ray * 3 + k];
                x[k] = rayo[ray * 3 + k] + d[k] * zz;
            }
            posenc<10, kCT>(x, h, c, pe);
            posenc<4, kCT>(d, h, c, pv);
        }
        bf16x8 ha[16][kCT], hb[16][kCT], r0[8][kCT];
        float sigma[kCT];
        const float* bl = bias_lds + kBiasL0;
        auto pend = [&](auto relu_tag, const Acc& a, bf16x8(&lo)[kCT], bf16x8(&hi)[kCT]) {
            return EpiB<decltype(relu_tag)::value>{a, lo, hi};
        };
        using T = std::true_type;
        using F = std::false_type;
        // chunk index K: L0 0-7, L1-4 8-39, L5 40-47, L6-7 48-63, bottleneck 64-71, sigma 72, rgb0 73-76, rgb1 77;
        // tile K accumulates in accs[K & 1]
        layer<0, 4, 0, 8, true, AB, DMA>(cx, rg, bl, bl + 256 * 1, pe, pe, ha, accs, pre, EpiNone{});
        layer<8, 16, 0, 8, true, AB, DMA>(cx, rg, bl + 256 * 1, bl + 256 * 2, ha, pe, hb, accs, pre, pend(T{}, accs[1], ha[14], ha[15]));
        layer<16, 16, 0, 8, true, AB, DMA>(cx, rg, bl + 256 * 2, bl + 256 * 3, hb, pe, ha, accs, pre, pend(T{}, accs[1], hb[14], hb[15]));
        layer<24, 16, 0, 8, true, AB, DMA>(cx, rg, bl + 256 * 3, bl + 256 * 4, ha, pe, hb, accs, pre, pend(T{}, accs[1], ha[14], ha[15]));
        layer<32, 16, 0, 8, true, AB, DMA>(cx, rg, bl + 256 * 4, bl + 256 * 5, hb, pe, ha, accs, pre, pend(T{}, accs[1], hb[14], hb[15]));
        layer<40, 16, 4, 8, true, AB, DMA>(cx, rg, bl + 256 * 5, bl + 256 * 6, ha, pe, hb, accs, pre, pend(T{}, accs[1], ha[14], ha[15]));
        layer<48, 16, 0, 8, true, AB, DMA>(cx, rg, bl + 256 * 6, bl + 256 * 7, hb, pe, ha, accs, pre, pend(T{}, accs[1], hb[14], hb[15]));
        layer<56, 16, 0, 8, true, AB, DMA>(cx, rg, bl + 256 * 7, bias_lds + kBiasBott, ha, pe, hb, accs, pre, pend(T{}, accs[1], ha[14], ha[15]));
        // bottleneck (no activation) hb -> ha; next tile = sigma (bias row 256 of the fused matrix)
        layer<64, 16, 0, 8, false, AB, DMA>(cx, rg, bias_lds + kBiasBott, bias_lds + kBiasBott + 256, hb, pe, ha, accs, pre,
                                       pend(T{}, accs[1], hb[14], hb[15]));
        // sigma tile (K = 72 -> accs[0]); pending: last bottleneck tile (accs[1]); next: rgb_out[0] tile 0
        tile<72, 16, 0, AB, DMA>(cx, rg, bias_lds + kBiasRgb0, hb, pe, accs[0], accs[1], pre, pend(F{}, accs[1], ha[14], ha[15]));
        {
            EpiSigma es{accs[0], sigma};
            layer<73, 16, 2, 4, true, AB, DMA>(cx, rg, bias_lds + kBiasRgb0, bias_lds + kBiasRgb1, ha, pv, r0, accs, pre, es);
        }
        // rgb_out[1] (K = 77 -> accs[1]); pending: last rgb_out[0] tile (K = 76 -> accs[0]); next: L0 tile 0
        tile<77, 8, 0, AB, DMA>(cx, rg, bl, r0, pe, accs[1], accs[0], pre, pend(T{}, accs[0], r0[6], r0[7]));
        if (h == 0) {
#pragma unroll
            for (int c = 0; c < kCT; ++c)
                if (m[c] < n_pts) out[m[c]] = make_float4(accs[1].v[c][0], accs[1].v[c][1], accs[1].v[c][2], sigma[c]);
        }
    }
}

#endif   // !NFX_V6_SIGMA
}  // namespace NFX_V6_NS
}  // namespace nfx

#ifndef NFX_V6_SIGMA
template <int AB, int DMA>
static int launch_v6(const float* rayo, const float* rayd, const float* z, long long n_pts, int n_samples,
                     const void* blob, float* out, int max_blocks, hipStream_t stream) {
    using namespace nfx;
    const int tile_pts = v6::kNW * 32 * v6::kCT;
    const long long n_tiles = (n_pts + tile_pts - 1) / tile_pts;
    const int grid = (int)(n_tiles < max_blocks ? n_tiles : max_blocks);
    auto kern = v6::nerf_mlp_bf16_v6_kernel<AB, DMA>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       v6::lds_of<DMA>);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(v6::kNW * 64), v6::lds_of<DMA>, stream, rayo, rayd, z, n_pts, n_samples,
                       (const char*)blob, (float4*)out);
    return (int)hipGetLastError();
}

extern "C" int nfx_launch_nerf_mlp_bf16_v6(const float* rayo, const float* rayd, const float* z, long long n_pts,
                                           int n_samples, const void* blob, float* out, int max_blocks, int dma_mode,
                                           hipStream_t stream) {
    if (n_pts <= 0) return 0;
    if (dma_mode == 1) return launch_v6<0, 1>(rayo, rayd, z, n_pts, n_samples, blob, out, max_blocks, stream);   // variant 7
    if (dma_mode == 2) return launch_v6<0, 2>(rayo, rayd, z, n_pts, n_samples, blob, out, max_blocks, stream);   // variant 8
    return launch_v6<0, 0>(rayo, rayd, z, n_pts, n_samples, blob, out, max_blocks, stream);                      // variant 6
}
#endif   // !NFX_V6_SIGMA
