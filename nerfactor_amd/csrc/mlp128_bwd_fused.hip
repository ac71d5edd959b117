// mlp128_bwd_fused.hip — backward of the width-128 surface MLPs with the WEIGHT GRADIENTS ACCUMULATED ON CHIP
// (round 4; tape.gradient of trainvali.py:284 through shape.py:196-237 / nerfactor.py:377-411).
//
// Round 3's backward (mlp128_bwd.hip) stores every layer's input and pre-activation gradient — 1130 bf16 features per
// row, 2.37 GB for the 1 048 576 light-visibility rows of a 1024-ray step — and a separate GEMM launch reads them back:
// 4.7 GB of HBM traffic for a few MB of algorithmic input / output, both kernels at the 64-requests-per-CU ceiling
// (profiles/HISTORY.md section 3b).  Here nothing but the weight gradients leaves the CU:
//
//   * one workgroup = 4 waves x 32 rows, persistent over its 128-row tiles; per tile it re-computes the forward and
//     runs the dgrad chain exactly like mlp128_bwd_ring_kernel (same MFMA order, same operand registers);
//   * when dZ_l of a tile exists, each wave writes its 32 rows of dZ_l and of the layer's input H_{l-1} ROW-MAJOR into
//     two LDS buffers ([128 rows][128 slots] bf16, one ds_write_b128 per B-operand register: a lane holds 8 slots of
//     its row), and after one barrier reads them back with the TRANSPOSING LDS load of gfx950
//     (ds_read_b64_tr_b16: a 16-lane group turns a [4 rows][16 slots] block into "4 rows of one slot per lane") as
//     MFMA operands whose K dimension is the ROW axis:   dW[i, j] += sum_rows H[row, i] dZ[row, j];
//   * wave w owns the 32 dZ slots [32 w, 32 w + 32) of every layer and ALL input slots: its dW blocks (32 x 32 fp32
//     = 16 VGPRs each) stay in registers for the whole launch and are written once, at the end, to a per-workgroup
//     slice of `partial`; mlp128_wgrad_reduce_kernel adds the slices IN WORKGROUP ORDER into dkernels / dbiases
//     (deterministic: no atomics) and undoes the slot permutations (B-operand slot <-> logical feature);
//   * bias gradients ride along: the positional-encoding operand has a zero pad slot, set to 1.0 here (its forward
//     weights are zero), so the dW row of that slot IS db of layers 0 and 3; for layers 1, 2 and `out` every lane adds
//     the rows of the dZ operand it holds anyway (v_dot2c_f32_bf16 with a (1, 1) operand: 4 VALU per k-step, one
//     register per layer — an all-ones MFMA block would cost 16);
//   * 19 blocks = 304 accumulator registers do not fit beside the chain's ~270: the layers are split over TWO launches.
//     PART 0 takes layer 3 and `out` (8 blocks) and STOPS behind dZ3: forward + one dgrad step, 23 weight sub-chunks per
//     tile.  PART 1 takes layers 2, 1, 0 (11 blocks) and runs the whole chain (35 sub-chunks), keeping h2 / h3 only as
//     ReLU mask bits.  Per 32 rows: 156 + 72 and 252 + 104 matrix instructions = 584 against 428 for one pass, against
//     2.3 KB of HBM traffic per row removed.  (The first split — layer 3, out, layer 0 | layers 2, 1, both running the
//     whole chain, bias by all-ones MFMA blocks — measured 571 + 564 us per 1 048 576 rows: profiles/r04/call_c.)
//   * ALL weights (forward + dgrad, 248-304 KiB per tile from L2) stream through the 5-slot LDS-DMA ring of
//     mlp128_bwd.hip, extended by the 14 dgrad sub-chunks: with no store in the tile loop the counted vmcnt waits see
//     DMA pieces (and the tile's few input loads, which only make a wait conservative).
// LDS: ring 40 KiB | biases | X rows | H rows | dZ rows = 141 568 bytes.  The three row buffers are [128 rows][64 dwords]
// with NO padding and an XOR swizzle of the dword index by the row (swz_bytes below, round 5).  Round 4 padded the rows to
// 72 dwords: under the LDS's banking rules (MI355X_MICROARCH.md: ds_write_b128 = 8 lanes x 4 dwords over 32 banks,
// ds_read_b64_tr_b16 = 32 lanes x 2 dwords over 64 banks, ds_read_b128 = 16 lanes x 4 dwords over 64 banks) that pitch makes
// every row store AND every transposing read a 2-way conflict — 816 of the 2640 LDS cycles of a PART-1 tile by that model,
// 29 % by the counters (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, profiles/r04/pmc_train_digest.json).  No pitch serves
// both (stores want the row stride = 4 mod 8 dwords, the transposing reads = 16 mod 64); the swizzle does.
#include <type_traits>

#include "mlp_engine.hpp"
#include "lds_dma.hpp"
#include "mlp128_train_layout.hpp"
#include "tr16.hpp"

namespace nfx {
namespace bwd {
namespace fused {

constexpr int kNW = 4, kRows = kNW * 32;
#ifndef NFX_FUSED_RING
#define NFX_FUSED_RING 5
#endif
constexpr int kR = NFX_FUSED_RING, kD = kR - 1, kSlot = 8192;
constexpr int kFwdSub = 21;   // sub-chunks per tile: forward 0-20, dgrad through `out` 21-22, W3 23-26, W2 27-30, W1 31-34
// PART 0 (layers 3 and out) stops behind dZ3: 23 sub-chunks per tile; PART 1 (layers 2, 1, 0) runs the whole chain: 35
constexpr int sub_n(int part) { return part == 0 ? 23 : 35; }
constexpr int kHPitch = 128 * 2;          // bytes per row of the X / H / dZ buffers (64 dwords, swizzled)
// Physical byte offset inside a row = logical offset ^ swz_bytes(row) (bits 4-7 = dword bits 2-5):
//   dword bits 4-5 ^= row bits 0-1   -> the 4 consecutive rows of a transposing read (16 contiguous dwords each) fall into
//                                       the four 16-dword quarters of the 64 banks;
//   dword bits 2-3 ^= (row bits 1, 2) ^ row bit 3 -> with bit 4 the 8 consecutive rows of a ds_write_b128 lane group get 8
//                                       distinct 4-dword spans of the 32 store banks, and the 16 rows of a ds_read_b128 lane
//                                       group ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}) 16 distinct spans of the 64.
__device__ __forceinline__ int swz_bytes(int row) {
    const int b0 = row & 1, b1 = (row >> 1) & 1, b2 = (row >> 2) & 1, b3 = (row >> 3) & 1;
    return ((b1 ^ b3) << 4) | ((b2 ^ b3) << 5) | (b0 << 6) | (b1 << 7);
}
// this lane's byte offset (from the buffer's start) of a transposing read: rows 8 (q >> 1) + 4 hi + (i >> 2) (+ 16 kk: not
// in the swizzle), slots [32 tile + 16 (q & 1) + 4 (i & 3), + 4) — tr16.hpp:tr_lane_off with the swizzle applied
__device__ __forceinline__ int tr_swz_off(int lane, int tile, int hi) {
    const int i = lane & 15, q = lane >> 4;
    const int row = 8 * (q >> 1) + 4 * hi + (i >> 2);
    return row * kHPitch + ((tile * 64 + 32 * (q & 1) + 8 * (i & 3)) ^ swz_bytes(row));
}
struct TrOff {      // per lane: offsets of the lo / hi half of an operand for column tiles 0..3 and for this wave's own tile
    int lo[4], hi[4], mylo[1], myhi[1];
};
__device__ __forceinline__ bf16x8 tr_frag2(const char* lo_p, const char* hi_p) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lo_p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(hi_p));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <int KSX, int NS>
struct Sub {
    using G = Geo<KSX>;
    static constexpr int frags(int k) {
        return k < 4 ? G::kP0 : k < 12 ? 8 : k < 20 ? ((k - 12) % 2 == 0 ? 8 : G::kP3 - 8) : 8;
    }
    static constexpr int off(int k) {   // first fragment in the train blob
        return k < 4 ? k * G::kP0 : k < 8 ? 4 * G::kP0 + (k - 4) * 8 : k < 12 ? 4 * G::kP0 + 32 + (k - 8) * 8
             : k < 20 ? 4 * G::kP0 + 64 + ((k - 12) / 2) * G::kP3 + ((k - 12) % 2) * 8
             : k == 20 ? 4 * G::kP0 + 64 + 4 * G::kP3
             : G::kFwdFrags + (k - kFwdSub) * 8;   // the 112 dgrad fragments are contiguous: 14 sub-chunks of 8
    }
    static constexpr int pieces(int k) { return frags(k) / kNW; }   // 1-KiB pieces per wave: 1 | 2
    // pieces that may still be in flight at the barrier that ends sub-chunk k: those of k + 3 ... k + kD.  k + 1 AND k + 2 have
    // landed for every wave behind it (round 5: the first fragments of sub-chunk k + 2 are read while k + 1 multiplies — see
    // sub_mma; until then the barrier guaranteed k + 1 only and every sub-chunk began with "read, wait ~130 cycles, MFMA")
    static constexpr int allow(int k) {
        int n = 0;
        for (int j = 3; j <= kD; ++j) n += pieces((k + j) % NS);
        return n;
    }
};
static_assert(Sub<6, 35>::off(kFwdSub) == Geo<6>::kFwdFrags && Sub<6, 35>::off(34) + 8 == Geo<6>::kFwdFrags + Geo<6>::kBwdFrags, "blob");
static_assert(Sub<4, 35>::off(kFwdSub) == Geo<4>::kFwdFrags && Sub<4, 35>::off(34) + 8 == Geo<4>::kFwdFrags + Geo<4>::kBwdFrags, "blob");

template <int KSX>
struct Lds {
    static constexpr int kBias = kR * kSlot;
    static constexpr int kX = (kBias + m128::kMainBiasFloats * 4 + 255) / 256 * 256;   // PART 0: the input rows (KSX k-steps of the 8 a row holds); PART 1: h0
    static constexpr int kH = kX + kRows * kHPitch;
    static constexpr int kDZ = kH + kRows * kHPitch;
    static constexpr int kTotal = kDZ + kRows * kHPitch;
    static_assert(kX % 256 == 0 && kH % 256 == 0 && kDZ % 256 == 0, "rows start on bank 0");
    static_assert(kTotal <= 160 * 1024, "LDS");
};

// blocks of 32 x 32 weight-gradient accumulators per wave and launch
template <int KSX, int PART>
struct Blocks {
    static constexpr int NX = KSX / 2;                      // 32-slot tiles of the network input
    // PART 0: [0, 4) W3 rows of h2 | [4, 4 + NX) W3 rows of the input | out || bias rows
    // PART 1: [0, 4) W2 | [4, 8) W1 | [8, 8 + NX) W0 || bias rows
    // NACC accumulator blocks (16 registers each) + one block-sized slot of the workgroup's slice whose first rows hold
    // the per-lane column sums of dZ (row 0: out | layer 2, row 1: layer 1): N slots in all
    static constexpr int kW3h = 0, kW3x = 4, kOut = 4 + NX;
    static constexpr int kW2 = 0, kW1 = 4, kW0 = 8;
    static constexpr int NACC = PART == 0 ? 5 + NX : 8 + NX, kBias = NACC, N = NACC + 1;
};

struct Ctx {
    char* smem;
    unsigned smem_lds;
    const char* blob;
    int lane, wave;   // wave: wave-uniform
    int cur;          // ring slot of the sub-chunk being consumed (wave-uniform)
};
template <int KSX, int NS, int K>
__device__ __forceinline__ void begin(const Ctx& cx) {
    constexpr int F = (K + kD) % NS, n = Sub<KSX, NS>::pieces(F);
    unsigned long long base = reinterpret_cast<unsigned long long>(cx.blob);
    unsigned lds = cx.smem_lds;
    asm volatile("" : "+s"(base), "+s"(lds));   // per sub-chunk: keeps the addresses out of the loop preheader
    int slot = cx.cur + kD;
    slot = slot >= kR ? slot - kR : slot;
    const int piece0 = cx.wave * n;
    lds_dma_pieces<n>((unsigned)cx.lane * 16u, reinterpret_cast<const char*>(base) + (size_t)Sub<KSX, NS>::off(F) * 1024 + piece0 * 1024,
                      lds + (unsigned)slot * kSlot + (unsigned)piece0 * 1024u);
}
// LGKM: the LDS reads that may still be in flight across the barrier — the NEXT sub-chunk's carry fragments, issued last.
// Everything older (this sub-chunk's own fragment reads: the compiler is free to sink their MFMAs below the barrier) has
// returned before the wave arrives, so no wave's DMA into this slot (begin<K + 1>) can overtake a read of it.
template <int KSX, int NS, int K, int LGKM>
__device__ __forceinline__ void end(Ctx& cx) {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(%1)\n\ts_barrier" ::"n"(Sub<KSX, NS>::allow(K)), "n"(LGKM) : "memory");
    cx.cur = cx.cur + 1 == kR ? 0 : cx.cur + 1;
}
// Fragment reads and MFMAs of one sub-chunk, software-pipelined ACROSS sub-chunks (round 5).  With one wave per SIMD nothing
// hides the ~130 cycles between a ds_read_b128 and its first use but the wave's own MFMAs, and the ring's barrier sits in
// front of every sub-chunk: "barrier, 8 reads, wait, MFMAs" exposed that latency 35 times per 128-row tile (the ISA of r04 /
// early r05: every sub-chunk "rrrrrrrr s_waitcnt lgkmcnt(0) MFMA").  Now the first kCarry fragments of a sub-chunk are read
// into the CARRY registers while the previous sub-chunk's second half multiplies — its barrier has guaranteed them, see
// Sub::allow — and the remaining ones at its own start, under its first-half MFMAs.  Fragment registers alive: 16 + 16, as
// before.  The carry is dropped where a weight-gradient section sits between two sub-chunks (HAVE = false behind it).
constexpr int kCarry = 4;
struct Carry {
    bf16x8 f[kCarry];
};
template <int N>
__device__ __forceinline__ void read_frags(const Ctx& cx, int slot, int f0, bf16x8* dst) {
    const char* p = cx.smem + slot * kSlot + f0 * kFragBytes + cx.lane * 16;
#pragma unroll
    for (int s = 0; s < N; ++s) dst[s] = *reinterpret_cast<const bf16x8*>(p + s * kFragBytes);
}
constexpr int kSchedAluOnly = 0x2 | 0x4;   // sched_barrier mask: VALU / SALU may cross, MFMA and LDS instructions may not
// acc += A(sub-chunk K: KS fragments) x b[0 .. KS); `pre`: VALU work that runs under the reads (the previous tile's epilogue).
// HAVE: the carry holds this sub-chunk's first fragments; NEXT: how many of the FOLLOWING sub-chunk's to read into it.
template <int KSX, int NS, int K, int KS, bool HAVE, int NEXT, int KSA, class Pre>
__device__ __forceinline__ void sub_mma(Ctx& cx, Carry& c, const bf16x8 (&b)[KSA][1], f32x16& acc, Pre pre) {
    constexpr int H = KS < kCarry ? KS : kCarry, R = KS - H;
    static_assert(NEXT <= kCarry && KS <= KSA, "fragments");
    bf16x8 rest[R > 0 ? R : 1];
    if constexpr (!HAVE) read_frags<H>(cx, cx.cur, 0, c.f);
    if constexpr (R > 0) read_frags<R>(cx, cx.cur, H, rest);
    __builtin_amdgcn_sched_barrier(0);      // (the reads stay in front of what follows)
    pre();
#pragma unroll
    for (int s = 0; s < H; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c.f[s], b[s][0], acc, 0, 0, 0);
    if constexpr (NEXT > 0) {
        __builtin_amdgcn_sched_barrier(kSchedAluOnly);
        read_frags<NEXT>(cx, cx.cur + 1 == kR ? 0 : cx.cur + 1, 0, c.f);
        __builtin_amdgcn_sched_barrier(kSchedAluOnly);
    }
#pragma unroll
    for (int s = 0; s < R; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rest[s], b[H + s][0], acc, 0, 0, 0);
    end<KSX, NS, K, NEXT>(cx);
}
// ---- packed 16-bit epilogues (the kernels are VALU-bound beside their MFMAs: r04 call C counted ~3300 VALU instructions
// per 128-row tile and wave against 340 matrix instructions; the scalar forms — v_med3 + v_cvt per value, bf16 -> f32 +
// compare + select per masked value, compare + select + shift-or per mask bit — were a third of them)
// Everything below works on DWORDS holding two bf16 (a B-operand register = 4 of them).  The two flag helpers are inline
// asm on purpose: written as vector min / subtract, hipcc 7.2 turns "is this bf16 non-zero" back into shift + float
// compare + select + permute per element (7 VALU per pair instead of 1).
__device__ __forceinline__ unsigned cvt2(float a, float b) {   // v_cvt_pk_bf16_f32: low half = a
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));
}
__device__ __forceinline__ unsigned relu2(unsigned v) {        // v_pk_max_i16: a negative bf16 is a negative int16
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), z));
}
__device__ __forceinline__ unsigned nonzero2(unsigned h) {     // 1 per half whose (non-negative) bf16 is not zero
    unsigned o;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(o) : "v"(h), "s"(0x00010001u));
    return o;
}
__device__ __forceinline__ unsigned keep2(unsigned v, unsigned flag01) {   // v where the half's flag is 1, +0 where it is 0
    unsigned m;
    asm("v_pk_sub_u16 %0, 0, %1" : "=v"(m) : "v"(flag01));                 // 0 - 1 = 0xffff
    return v & m;
}
// ReLU + bf16 of one accumulator tile (bias already in it) -> the next layer's two B-operand k-steps; BITS: also the
// tile's ReLU mask, one word per tile: pair j (accumulator registers 2 j, 2 j + 1) at bit 7 - j of the low / high half
template <bool BITS>
__device__ __forceinline__ void relu_tile(const f32x16& acc, bf16x8& lo, bf16x8& hi, unsigned& bits) {
    u32x4 wl, wh;
    unsigned w = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned pr = relu2(cvt2(acc[2 * j], acc[2 * j + 1]));
        if (j < 4) wl[j] = pr;
        else wh[j - 4] = pr;
        if constexpr (BITS) w = (w << 1) | nonzero2(pr);
    }
    if constexpr (BITS) bits = w;
    lo = __builtin_bit_cast(bf16x8, wl);
    hi = __builtin_bit_cast(bf16x8, wh);
    mfma_operand_fence(lo);
    mfma_operand_fence(hi);
}
// one forward layer of 4 tiles whose chunk is ONE sub-chunk each (K0 = its first sub-chunk)
// `park` (optional): the tile's two output k-steps go to this lane's parked row in LDS as soon as they exist
template <int KSX, int NS, int K0, int KS, bool BITS, bool HAVE, int NEXT, int KSA>
__device__ __forceinline__ void layer(Ctx& cx, Carry& c, const float* bias, const bf16x8 (&b)[KSA][1], bf16x8 (&out)[8][1],
                                      unsigned (&mk)[4], char* park = nullptr, int lx = 0) {
    f32x16 prev;      // tile t - 1's accumulators: its epilogue runs under tile t's fragment reads
    auto epilogue = [&](auto T) {
        constexpr int t = decltype(T)::value;
        relu_tile<BITS>(prev, out[2 * t][0], out[2 * t + 1][0], mk[t]);
        if (park != nullptr) {      // (the lane's swizzled row: k-step s at park + ((32 s) ^ lx), see store_rows)
            *reinterpret_cast<bf16x8*>(park + (((2 * t) * 32) ^ lx)) = out[2 * t][0];
            *reinterpret_cast<bf16x8*>(park + (((2 * t + 1) * 32) ^ lx)) = out[2 * t + 1][0];
        }
    };
    static_for<0, 4>([&](auto T) {
        constexpr int t = decltype(T)::value;
        begin<KSX, NS, K0 + t>(cx);
        f32x16 acc[1];
        bias_init<1>(bias + 32 * t, cx.lane >> 5, acc);      // (broadcast reads: in flight under the epilogue as well)
        sub_mma<KSX, NS, K0 + t, KS, (t > 0 || HAVE), (t < 3 ? (KS < kCarry ? KS : kCarry) : NEXT)>(cx, c, b, acc[0], [&]() {
            if constexpr (t > 0) epilogue(std::integral_constant<int, t - 1>{});
        });
        prev = acc[0];
    });
    epilogue(std::integral_constant<int, 3>{});
}
__device__ __forceinline__ void zero_acc(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}
// ReLU mask of tile t of a layer.  MODE 0: from the re-computed activation in registers; 1: from its mask bits; 2: from the
// activation parked in LDS (`park` = this lane's row, `lx` its swizzled half: k-step s at park + ((32 s) ^ lx) — exactly
// what store_rows wrote).
template <int MODE>
__device__ __forceinline__ void relu_mask(const f32x16& acc, const bf16x8 (&hact)[8][1], const unsigned (&mk)[4],
                                          const char* park, int lx, int t, bf16x8& olo, bf16x8& ohi) {
    u32x4 plo, phi;
    if constexpr (MODE == 2) {
        plo = *reinterpret_cast<const u32x4*>(park + (((2 * t) * 32) ^ lx));
        phi = *reinterpret_cast<const u32x4*>(park + (((2 * t + 1) * 32) ^ lx));
    } else if constexpr (MODE == 0) {
        plo = __builtin_bit_cast(u32x4, hact[2 * t][0]);
        phi = __builtin_bit_cast(u32x4, hact[2 * t + 1][0]);
    }
    u32x4 wl, wh;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        unsigned on;
        if constexpr (MODE == 1) on = (mk[t] >> (7 - j)) & 0x00010001u;
        else on = nonzero2(j < 4 ? plo[j] : phi[j - 4]);
        const unsigned v = keep2(cvt2(acc[2 * j], acc[2 * j + 1]), on);
        if (j < 4) wl[j] = v;
        else wh[j - 4] = v;
    }
    olo = __builtin_bit_cast(bf16x8, wl);
    ohi = __builtin_bit_cast(bf16x8, wh);
    mfma_operand_fence(olo);
    mfma_operand_fence(ohi);
}
// dgrad layer, one sub-chunk (8 fragments) per 32-feature tile: dH^T = W dZ^T, ReLU-masked by the activation
template <int KSX, int NS, int K0, int MODE, bool HAVE>
__device__ __forceinline__ void dgrad(Ctx& cx, Carry& c, const bf16x8 (&dz)[8][1], const bf16x8 (&hact)[8][1], const unsigned (&mk)[4],
                                      const char* park, int lx, bf16x8 (&dout)[8][1]) {
    f32x16 prev;
    static_for<0, 4>([&](auto T) {
        constexpr int t = decltype(T)::value;
        begin<KSX, NS, K0 + t>(cx);
        f32x16 acc[1];
        zero_acc(acc[0]);
        sub_mma<KSX, NS, K0 + t, 8, (t > 0 || HAVE), (t < 3 ? kCarry : 0)>(cx, c, dz, acc[0], [&]() {   // (a weight-gradient section follows tile 3)
            if constexpr (t > 0) relu_mask<MODE>(prev, hact, mk, park, lx, t - 1, dout[2 * t - 2][0], dout[2 * t - 1][0]);
        });
        prev = acc[0];
    });
    relu_mask<MODE>(prev, hact, mk, park, lx, 3, dout[6][0], dout[7][0]);
}

// ---- rows -> LDS, LDS -> row-contracting MFMA operands
// lane (row p, half h) stores the 8 slots of k-step s (slot index = 16 s + 8 h + j) at logical byte s * 32 + h * 16 of its
// row, physical (s * 32) ^ lx with lx = (h * 16) ^ swz_bytes(row): one v_xor per store instead of an immediate offset
template <int KS, int KSA>
__device__ __forceinline__ void store_rows(char* row, int lx, const bf16x8 (&v)[KSA][1]) {
#pragma unroll
    for (int s = 0; s < KS; ++s) *reinterpret_cast<bf16x8*>(row + ((s * 32) ^ lx)) = v[s][0];
}
// acc[i] += A(tile i of the buffer at `a`, pitch APITCH)^T-contracted-over-rows with B = this wave's dZ tile at `b`.
// bsum (optional): + the sum of the B operand's 8 rows — lane (slot n, k-group g) adds its rows of every k-step, so the
// column sum of slot n (the bias gradient) is bsum of lane n + bsum of lane n + 32.  v_dot2c_f32_bf16 with a (1, 1)
// operand adds two rows per instruction: 4 VALU per k-step instead of a 16-register all-ones MFMA block.
// `a`, `b`: the buffers' starts; A tiles i < NI at the lane's offsets alo[i] / ahi[i] (MINE: the wave's own tile), B at
// blo / bhi; the 16 rows of k-step kk are an immediate offset (kk * 4 KiB: outside the swizzle).
// Software-pipelined (round 5): the transposing reads of MFMA s + kWgradAhead are issued before MFMA s.  Round 4 read an
// operand and used it at once — "ds_read_tr x2, s_waitcnt lgkmcnt(0), MFMA" 104 times a PART-1 tile, the LDS latency
// exposed every time (142 full lgkmcnt drains per tile in the ISA, a quarter of the wave's time waiting on a counter).
#ifndef NFX_FUSED_WGRAD_AHEAD
#define NFX_FUSED_WGRAD_AHEAD 2
#endif
constexpr int kWgradAhead = NFX_FUSED_WGRAD_AHEAD;
template <int NI, int A0, int NA, int NO>
__device__ __forceinline__ void wgrad(const char* a, const int (&alo)[NO], const int (&ahi)[NO], const char* b, int blo, int bhi,
                                      f32x16 (&acc)[NA], float* bsum) {
    static_assert(NI <= NO, "offsets");
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const b2 ones = {(__bf16)1.f, (__bf16)1.f};
    constexpr int D = kWgradAhead, NSTEP = (kRows / 16) * NI;     // step s: k-step s / NI, A tile s % NI
    constexpr int NB = (D + NI - 1) / NI + 1;                     // B operands alive at once
    bf16x8 af[D + 1], bf[NB];
    auto load_a = [&](int s) { return tr_frag2(a + alo[s % NI] + (s / NI) * 16 * kHPitch, a + ahi[s % NI] + (s / NI) * 16 * kHPitch); };
    auto load_b = [&](int kk) { return tr_frag2(b + blo + kk * 16 * kHPitch, b + bhi + kk * 16 * kHPitch); };
#pragma unroll
    for (int s = 0; s < D && s < NSTEP; ++s) {
        if (s % NI == 0) bf[(s / NI) % NB] = load_b(s / NI);
        af[s % (D + 1)] = load_a(s);
    }
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        const int kk = s / NI, i = s % NI;
        if (s + D < NSTEP) {
            if ((s + D) % NI == 0) bf[((s + D) / NI) % NB] = load_b((s + D) / NI);
            af[(s + D) % (D + 1)] = load_a(s + D);
        }
        acc[A0 + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s % (D + 1)], bf[kk % NB], acc[A0 + i], 0, 0, 0);
        if (bsum != nullptr && i == NI - 1) {
            // (hipcc 7.2: four dot products of the dwords of a bit_cast u32x4 all read the FIRST dword — r04 call D, every
            //  dot2-summed bias gradient wrong; element pairs spelled out read the right registers)
            float t = *bsum;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const b2 pr = {bf[kk % NB][2 * q], bf[kk % NB][2 * q + 1]};
                t = __builtin_amdgcn_fdot2_f32_bf16(pr, ones, t, false);
            }
            *bsum = t;
        }
        __builtin_amdgcn_sched_barrier(0);   // one MFMA's operands at a time: the scheduler otherwise hoists every read of a
                                             // k-step (80 fragment registers) above the first MFMA and the accumulators spill
    }
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Up to kMaxHeads networks over the SAME input rows in one launch (blockIdx.y = head; round 5): the three xyz heads of a
// NeRFactor step — normal, albedo, BRDF code, 2048 rows each — are 16 workgroups per launch on their own, three launches per
// head back to back; as heads of one launch they fill 48 CUs for the time of one.
constexpr int kMaxHeads = 4;
struct HeadArgs {
    const char* blob[kMaxHeads];
    const float* dout[kMaxHeads];
    float* partial[kMaxHeads];
    int out_dim[kMaxHeads], out_act[kMaxHeads];
    float post_scale[kMaxHeads];
};
// IN_KIND 0: posenc10(xyz_scale*xyz) ; 1: [posenc10(xyz_scale*xyz), posenc4(normalize(lxyz_l - xyz_dir))]
template <int IN_KIND, int PART>
__global__ __launch_bounds__(kNW * 64, 1) void mlp128_bwd_fused_kernel(
    const float* __restrict__ xyz, const float* __restrict__ xyz_dir, long long n, float xyz_scale,
    const float* __restrict__ lxyz, int n_lights, HeadArgs heads) {
    const int hd = blockIdx.y;
    const char* __restrict__ blob = heads.blob[hd];
    const float* __restrict__ dout = heads.dout[hd];
    float* __restrict__ partial = heads.partial[hd];
    const int out_dim = heads.out_dim[hd], out_act = heads.out_act[hd];
    const float post_scale = heads.post_scale[hd];
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KSX = IN_KIND == 0 ? 4 : 6;
    using G = Geo<KSX>;
    constexpr int NS = sub_n(PART);
    using S = Sub<KSX, NS>;
    using L = Lds<KSX>;
    using B = Blocks<KSX, PART>;
    constexpr int NX = B::NX;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* bias_lds = reinterpret_cast<float*>(smem + L::kBias);
    {
        const float* bsrc = reinterpret_cast<const float*>(blob + G::kWeightBytes);
        for (int i = tid; i < G::kBiasFloats; i += kNW * 64) bias_lds[i] = bsrc[i];
#pragma unroll
        for (int k = 0; k < kD; ++k) {   // sub-chunks 0 .. kD-1 of the first tile -> slots 0 .. kD-1
            const u32x4* src = reinterpret_cast<const u32x4*>(blob + (size_t)S::off(k) * 1024);
            u32x4* dst = reinterpret_cast<u32x4*>(smem + k * kSlot);
            for (int i = tid; i < S::frags(k) * 64; i += kNW * 64) dst[i] = src[i];
        }
        __syncthreads();
    }
    typedef __attribute__((address_space(3))) char lds_char;
    Ctx cx{smem, (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_char*)smem), blob, lane, wave, 0};
    // this lane's row of the three row-major buffers (stores) and its gather origin (transposing reads)
    const int row_local = wave * 32 + p;
    const int lx = (h * 16) ^ swz_bytes(row_local);                  // this lane's half of its (swizzled) row
    char* xrow = smem + L::kX + row_local * kHPitch;                 // PART 0: input rows; PART 1: h0 parked in the same region
    char* hrow = smem + L::kH + row_local * kHPitch;
    char* zrow = smem + L::kDZ + row_local * kHPitch;
    char* const h0row = xrow;
    const char* const xa = smem + L::kX;
    const char* const ha = smem + L::kH;
    const char* const za = smem + L::kDZ;
    TrOff tr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        tr.lo[i] = tr_swz_off(lane, i, 0);
        tr.hi[i] = tr_swz_off(lane, i, 1);
    }
    tr.mylo[0] = tr_swz_off(lane, wave, 0);                             // the wave's own tile: dZ slots [32 wave, 32 wave + 32)
    tr.myhi[0] = tr_swz_off(lane, wave, 1);
    f32x16 acc[B::NACC];
#pragma unroll
    for (int i = 0; i < B::NACC; ++i) zero_acc(acc[i]);
    float bsum0 = 0.f, bsum1 = 0.f;   // column sums of dZ_out (PART 0) | dZ2, dZ1 (PART 1): the bias gradients

    const long long n_rows = IN_KIND == 0 ? n : n * n_lights;
    const long long n_tiles = (n_rows + kRows - 1) / kRows;
    // the rows' inputs are fetched ONE TILE AHEAD: a lone wave per SIMD has nothing to put under the first touch of a
    // point (an HBM miss) at the top of a tile; the loads ride in the DMA window, where they only make a counted wait
    // conservative
    float nx[3], ndv[4], nxd[3];   // (the light position comes from a 6-KiB table that stays in L1 / L2: loaded at the tile's top)
    auto fetch = [&](long long tl) {
        const long long row = tl * kRows + row_local;
        const bool valid = row < n_rows;
        const long long rc = valid ? row : n_rows - 1;
        const long long pt = IN_KIND == 0 ? rc : rc / n_lights;
#pragma unroll
        for (int k = 0; k < 3; ++k) nx[k] = xyz[pt * 3 + k];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 4 * h + r;
            ndv[r] = (valid && f < out_dim) ? dout[row * out_dim + f] : 0.f;
        }
        if constexpr (IN_KIND == 1) {
#pragma unroll
            for (int k = 0; k < 3; ++k) nxd[k] = xyz_dir[pt * 3 + k];
        }
    };
    fetch(blockIdx.x);
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long row = tile * kRows + row_local;
        const bool valid = row < n_rows;
        float x[3], dv[4], xd[3], lp[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            x[k] = xyz_scale * nx[k];
            xd[k] = nxd[k];
        }
        if constexpr (IN_KIND == 1) {
            const int l = (int)((valid ? row : n_rows - 1) % n_lights);
#pragma unroll
            for (int k = 0; k < 3; ++k) lp[k] = lxyz[l * 3 + k];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) dv[r] = ndv[r];
        fetch(tile + gridDim.x < n_tiles ? tile + gridDim.x : tile);
        bf16x8 xin[KSX][1];
        {
            bf16x8 pe[4][1];
            posenc<10, 1>(x, h, 0, pe);
#pragma unroll
            for (int s = 0; s < 4; ++s) xin[s][0] = pe[s][0];
            // the zero pad slot of the encoding (q = 31 of lane half 1; its forward weights are zero, pack.cpp
            // seg_row) carries 1.0: the dW row of that slot is the bias gradient of layers 0 and 3
            if (h == 1) xin[3][0][7] = (__bf16)1.0f;
        }
        if constexpr (IN_KIND == 1) {
            float d[3];
            dir_to(lp, xd, d);
            bf16x8 pl[2][1];
            posenc<4, 1>(d, h, 0, pl);
            xin[4][0] = pl[0][0];
            xin[5][0] = pl[1][0];
        }
        if constexpr (PART == 0) store_rows<KSX>(xrow, lx, xin);   // the input rows: read by the W3 step of this tile
        // ------------------------------------------------------------------ forward (re-computed)
        bf16x8 h0[8][1], h1[8][1], h2[8][1], h3[8][1];
        // PART 1 multiplies h1, h0 and the input.  h0 and h1 are PARKED in LDS the moment they exist (h0 in the region
        // PART 0 uses for the input rows, h1 in the H buffer): the weight-gradient steps read them there, the dgrad masks
        // re-read the lane's own row, and their 64 registers are free for the whole backward (kept live they spilled: 35
        // registers, 730 us per 1 048 576 rows — r04 call D).  h2 and h3 it needs only as masks: 2 registers of bits.
        // PART 0 stops behind dZ3 and needs no mask below h3.
        unsigned m2[4], m3[4];
        constexpr bool kBits = PART == 1;
        Carry cr;     // the next sub-chunk's first fragments (sub_mma)
        constexpr int kXc = KSX < kCarry ? KSX : kCarry;
        layer<KSX, NS, 0, KSX, false, false, kCarry>(cx, cr, bias_lds, xin, h0, m2, PART == 1 ? h0row : nullptr, lx);
        layer<KSX, NS, 4, 8, false, true, kCarry>(cx, cr, bias_lds + 128, h0, h1, m2, PART == 1 ? hrow : nullptr, lx);
        layer<KSX, NS, 8, 8, kBits, true, kCarry>(cx, cr, bias_lds + 256, h1, h2, m2);
        {
            f32x16 prev3;
            static_for<0, 4>([&](auto T) {   // layer 3: [h2 ; input], two sub-chunks per tile
                constexpr int t = decltype(T)::value;
                begin<KSX, NS, 12 + 2 * t>(cx);
                f32x16 a3[1];
                bias_init<1>(bias_lds + 384 + 32 * t, h, a3);
                sub_mma<KSX, NS, 12 + 2 * t, 8, true, kXc>(cx, cr, h2, a3[0], [&]() {
                    if constexpr (t > 0) relu_tile<kBits>(prev3, h3[2 * t - 2][0], h3[2 * t - 1][0], m3[t - 1]);   // (under the reads)
                });
                begin<KSX, NS, 13 + 2 * t>(cx);
                sub_mma<KSX, NS, 13 + 2 * t, KSX, true, kCarry>(cx, cr, xin, a3[0], []() {});   // next: tile t + 1's h2 part, or `out`
                prev3 = a3[0];
            });
            relu_tile<kBits>(prev3, h3[6][0], h3[7][0], m3[3]);
        }
        f32x16 logit[1];
        begin<KSX, NS, 20>(cx);
        bias_init<1>(bias_lds + 512, h, logit);
        // (PART 0: the out layer's weight-gradient section sits between sub-chunks 20 and 21 — no carry across it)
        sub_mma<KSX, NS, 20, 8, true, (PART == 1 ? 1 : 0)>(cx, cr, h3, logit[0], []() {});
        // ------------------------------------------------------------------ dZ_out
        bf16x8 dzo[1][1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float g = dv[r] * post_scale * act_grad(logit[0][r], out_act);
            dzo[0][0][r] = (__bf16)((valid && 4 * h + r < out_dim) ? g : 0.f);
            dzo[0][0][4 + r] = (__bf16)0.f;
        }
        if constexpr (PART == 0) {   // out layer: dWo[i, f] = sum h3[row, i] dZo[row, f]; wave w takes h3 slots [32 w, +32)
            store_rows<8>(hrow, lx, h3);
            store_rows<1>(zrow, lx, dzo);
            lds_barrier();
            wgrad<1, B::kOut>(ha, tr.mylo, tr.myhi, za, tr.lo[0], tr.hi[0], acc, &bsum0);
        }
        // ------------------------------------------------------------------ dgrad chain + weight gradients
        bf16x8 dz3[8][1];
        {   // through the out layer: one k-step (16 padded output slots); tiles 0, 1 in sub-chunk 21, tiles 2, 3 in 22
            static_for<0, 2>([&](auto U) {      // fragments 0 and 4 of the sub-chunk: 0 rides in the carry
                constexpr int u = decltype(U)::value;
                begin<KSX, NS, 21 + u>(cx);
                f32x16 a0[1], a1[1];
                zero_acc(a0[0]);
                zero_acc(a1[0]);
                bf16x8 f4[1];
                if constexpr (u == 0 && PART == 0) read_frags<1>(cx, cx.cur, 0, cr.f);
                read_frags<1>(cx, cx.cur, 4, f4);
                __builtin_amdgcn_sched_barrier(0);
                a0[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cr.f[0], dzo[0][0], a0[0], 0, 0, 0);
                if constexpr (u == 0 || PART == 1) {      // next: sub-chunk 22 (fragment 0), or PART 1's dgrad through W3
                    __builtin_amdgcn_sched_barrier(kSchedAluOnly);
                    read_frags<(u == 0 ? 1 : kCarry)>(cx, cx.cur + 1 == kR ? 0 : cx.cur + 1, 0, cr.f);
                    __builtin_amdgcn_sched_barrier(kSchedAluOnly);
                }
                a1[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f4[0], dzo[0][0], a1[0], 0, 0, 0);
                end<KSX, NS, 21 + u, (u == 0 ? 1 : PART == 1 ? kCarry : 0)>(cx);
                relu_mask<kBits ? 1 : 0>(a0[0], h3, m3, nullptr, 0, 2 * u, dz3[4 * u][0], dz3[4 * u + 1][0]);
                relu_mask<kBits ? 1 : 0>(a1[0], h3, m3, nullptr, 0, 2 * u + 1, dz3[4 * u + 2][0], dz3[4 * u + 3][0]);
            });
        }
        if constexpr (PART == 0) {   // W3 = [h2 ; input]^T dZ3 (the two ring barriers above separate it from the out step)
            store_rows<8>(hrow, lx, h2);
            store_rows<8>(zrow, lx, dz3);
            lds_barrier();
            wgrad<4, B::kW3h>(ha, tr.lo, tr.hi, za, tr.mylo[0], tr.myhi[0], acc, nullptr);
            wgrad<NX, B::kW3x>(xa, tr.lo, tr.hi, za, tr.mylo[0], tr.myhi[0], acc, nullptr);
            lds_barrier();   // the next tile's first statement rewrites the X rows
        } else {
            bf16x8 dz2[8][1], dz1[8][1], dz0[8][1];
            dgrad<KSX, NS, 23, 1, true>(cx, cr, dz3, h2, m2, nullptr, 0, dz2);   // W3[:128, :]
            store_rows<8>(zrow, lx, dz2);
            lds_barrier();
            wgrad<4, B::kW2>(ha, tr.lo, tr.hi, za, tr.mylo[0], tr.myhi[0], acc, &bsum0);   // A = h1, parked in the H buffer by the forward
            dgrad<KSX, NS, 27, 2, false>(cx, cr, dz2, h1, m2, hrow, lx, dz1);     // W2; mask = this lane's parked h1 row
            store_rows<8>(zrow, lx, dz1);
            lds_barrier();
            wgrad<4, B::kW1>(xa, tr.lo, tr.hi, za, tr.mylo[0], tr.myhi[0], acc, &bsum1);   // A = h0, parked in the X region
            dgrad<KSX, NS, 31, 2, false>(cx, cr, dz1, h0, m2, h0row, lx, dz0);    // W1; mask = the parked h0 row
            store_rows<KSX>(hrow, lx, xin);                            // the input rows take the H buffer (h1 is done with)
            store_rows<8>(zrow, lx, dz0);
            lds_barrier();
            wgrad<NX, B::kW0>(ha, tr.lo, tr.hi, za, tr.mylo[0], tr.myhi[0], acc, nullptr);
            lds_barrier();   // the next tile's forward parks h1 in the H buffer again
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the sub-chunks fetched ahead for a tile that does not exist
    // ------------------------------------------------------------------ accumulators -> this workgroup's slice
    float* mine = partial + ((size_t)blockIdx.x * kNW + wave) * (size_t)(B::N * 1024) + lane;
#pragma unroll
    for (int b = 0; b < B::NACC; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[(b * 16 + r) * 64] = acc[b][r];
    mine[(B::kBias * 16 + 0) * 64] = bsum0;
    mine[(B::kBias * 16 + 1) * 64] = bsum1;
}

// ---- slot <-> logical feature (pack.cpp seg_row): slot c = 16 s + 8 h + j of a B operand
__device__ __forceinline__ int hidden_feature(int c) {
    const int s = c >> 4, hh = (c >> 3) & 1, j = c & 7;
    return 32 * (s >> 1) + 16 * (s & 1) + (j & 3) + 8 * (j >> 2) + 4 * hh;
}
constexpr int kSlotPad = -1, kSlotOnes = -2;
// network input: posenc10(xyz) in k-steps 0-3 (Embedder order, embedder.py:38-47), posenc4(ldir) in 4-5 (features 63 ..)
__device__ __forceinline__ int input_feature(int c) {
    const int s = c >> 4, hh = (c >> 3) & 1, j = c & 7;
    if (s < 4) {
        const int q = 8 * s + j;
        if (q < 30) return 3 + 6 * (q / 3) + (q % 3) + 3 * hh;
        if (q == 30) return hh ? 2 : 0;
        return hh ? kSlotOnes : 1;
    }
    const int q = 8 * (s - 4) + j;
    if (q < 12) return 63 + 3 + 6 * (q / 3) + (q % 3) + 3 * hh;
    if (q == 12) return 63 + (hh ? 2 : 0);
    if (q == 13) return hh ? kSlotPad : 64;
    return kSlotPad;
}

struct ReduceArgs {
    const float* part[2];
    int n_wg, out_dim, in_dims;
    float* dk[5];
    float* db[5];
};
struct ReduceHeads {
    ReduceArgs h[kMaxHeads];     // blockIdx.y = head
};
// One WAVE-QUARTET per 64 accumulator elements (part, wave, block, register, lanes 0-63): wave q of a workgroup sums
// the workgroups [q n/4, (q+1) n/4) of its 64 elements, the four partial sums are added in wave order through LDS
// (fixed order: deterministic), and the total is added to the gradient element it stands for.
template <int KSX>
__global__ __launch_bounds__(256) void mlp128_wgrad_reduce_kernel(ReduceHeads heads) {
    const ReduceArgs& a = heads.h[blockIdx.y];
    constexpr int N0 = Blocks<KSX, 0>::N, N1 = Blocks<KSX, 1>::N, NX = KSX / 2;
    using B0 = Blocks<KSX, 0>;
    using B1 = Blocks<KSX, 1>;
    __shared__ float quarter[4][64];
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), seg = threadIdx.x >> 6;
    const int lane = e & 63, r = (e >> 6) & 15;
    int blk = (e >> 10) % (N0 + N1);
    const int wave = (e >> 10) / (N0 + N1);
    const int part = blk >= N0 ? 1 : 0;
    if (part) blk -= N0;
    const int nb = part ? N1 : N0;
    const int n = lane & 31, m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);   // D element (row m, column n)
    float* dst = nullptr;
    const int jl = hidden_feature(32 * wave + n);   // the dZ slot of this column, as an output feature of its layer
    bool bias_row = false;   // a row of the bias slot: the column sum of slot n = the values of lanes n and n + 32
    if (part == 0) {
        if (blk < B0::kW3x) dst = a.dk[3] + hidden_feature(32 * blk + m) * 128 + jl;
        else if (blk < B0::kOut) {
            const int xf = input_feature(32 * (blk - B0::kW3x) + m);
            if (xf >= 0 && xf < a.in_dims) dst = a.dk[3] + (128 + xf) * 128 + jl;
            else if (xf == kSlotOnes) dst = a.db[3] + jl;
        } else {   // out layer: column n = dZo slot (half n >> 3, element n & 7) <-> output 4 (n >> 3) + (n & 7)
            const int f = 4 * (n >> 3) + (n & 7);
            if (n < 16 && (n & 7) < 4 && f < a.out_dim) {
                if (blk == B0::kOut) dst = a.dk[4] + hidden_feature(32 * wave + m) * a.out_dim + f;
                else if (wave == 0 && r == 0 && lane < 32) { dst = a.db[4] + f; bias_row = true; }
            }
        }
    } else {
        if (blk < B1::kW1) dst = a.dk[2] + hidden_feature(32 * blk + m) * 128 + jl;
        else if (blk < B1::kW0) dst = a.dk[1] + hidden_feature(32 * (blk - B1::kW1) + m) * 128 + jl;
        else if (blk < B1::kBias) {
            const int xf = input_feature(32 * (blk - B1::kW0) + m);
            if (xf >= 0 && xf < a.in_dims) dst = a.dk[0] + xf * 128 + jl;
            else if (xf == kSlotOnes) dst = a.db[0] + jl;
        } else if (r < 2 && lane < 32) { dst = a.db[r == 0 ? 2 : 1] + jl; bias_row = true; }
    }
    (void)NX;
    const float* src = a.part[part] + ((size_t)wave * nb + blk) * 1024 + r * 64 + lane;
    const size_t stride = (size_t)kNW * nb * 1024;
    const int g0 = (int)((long long)a.n_wg * seg / 4), g1 = (int)((long long)a.n_wg * (seg + 1) / 4);
    float s = 0.f;
    if (dst != nullptr) {   // (uniform per 64-element group except for the bias rows' upper lanes)
        if (bias_row) for (int g = g0; g < g1; ++g) s += src[g * stride] + src[g * stride + 32];
        else for (int g = g0; g < g1; ++g) s += src[g * stride];
    }
    quarter[seg][threadIdx.x & 63] = s;
    __syncthreads();
    if (seg == 0 && dst != nullptr) {
        const int i = threadIdx.x;
        *dst += (quarter[0][i] + quarter[1][i]) + (quarter[2][i] + quarter[3][i]);
    }
}

}  // namespace fused
}  // namespace bwd
}  // namespace nfx

template <int IN_KIND>
static int launch_fused(const float* xyz, const float* xyz_dir, long long n, float xyz_scale, const float* lxyz, int n_lights,
                        int n_heads, const void* const* blobs, const int* out_dims, const int* out_acts, const float* post_scales,
                        const float* const* douts, float* partial, int grid, float* const* dk, float* const* db, hipStream_t st) {
    using namespace nfx::bwd::fused;
    constexpr int KSX = IN_KIND == 0 ? 4 : 6;
    constexpr int lds = Lds<KSX>::kTotal;
    const size_t floats0 = (size_t)grid * kNW * Blocks<KSX, 0>::N * 1024, floats1 = (size_t)grid * kNW * Blocks<KSX, 1>::N * 1024;
    auto k0 = mlp128_bwd_fused_kernel<IN_KIND, 0>;
    auto k1 = mlp128_bwd_fused_kernel<IN_KIND, 1>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    HeadArgs h0 = {}, h1 = {};
    ReduceHeads rh = {};
    for (int i = 0; i < n_heads; ++i) {       // head i's slice of the workspace: [PART 0 partials | PART 1 partials]
        float* p0 = partial + (size_t)i * (floats0 + floats1);
        h0.blob[i] = h1.blob[i] = static_cast<const char*>(blobs[i]);
        h0.dout[i] = h1.dout[i] = douts[i];
        h0.out_dim[i] = h1.out_dim[i] = out_dims[i];
        h0.out_act[i] = h1.out_act[i] = out_acts[i];
        h0.post_scale[i] = h1.post_scale[i] = post_scales[i];
        h0.partial[i] = p0;
        h1.partial[i] = p0 + floats0;
        ReduceArgs& ra = rh.h[i];
        ra.part[0] = p0;
        ra.part[1] = p0 + floats0;
        ra.n_wg = grid;
        ra.out_dim = out_dims[i];
        ra.in_dims = IN_KIND == 0 ? 63 : 90;
        for (int j = 0; j < 5; ++j) {
            ra.dk[j] = dk[5 * i + j];
            ra.db[j] = db[5 * i + j];
        }
    }
    hipLaunchKernelGGL(k0, dim3(grid, n_heads), dim3(kNW * 64), lds, st, xyz, xyz_dir, n, xyz_scale, lxyz, n_lights, h0);
    hipLaunchKernelGGL(k1, dim3(grid, n_heads), dim3(kNW * 64), lds, st, xyz, xyz_dir, n, xyz_scale, lxyz, n_lights, h1);
    const int elems = kNW * (Blocks<KSX, 0>::N + Blocks<KSX, 1>::N) * 1024;
    hipLaunchKernelGGL(mlp128_wgrad_reduce_kernel<KSX>, dim3(elems / 64, n_heads), dim3(256), 0, st, rh);
    return (int)hipGetLastError();
}

extern "C" {
// floats of `partial` one launch pair needs for `grid` workgroups
size_t nfx_mlp128_fused_partial_floats(int in_kind, int grid) {
    using namespace nfx::bwd::fused;
    const int nb = in_kind == 0 ? Blocks<4, 0>::N + Blocks<4, 1>::N : Blocks<6, 0>::N + Blocks<6, 1>::N;
    return (size_t)grid * kNW * nb * 1024;
}
int nfx_mlp128_fused_grid(int in_kind, long long n, int n_lights, int max_blocks) {
    const long long rows = in_kind == 0 ? n : n * n_lights;
    const long long tiles = (rows + nfx::bwd::fused::kRows - 1) / nfx::bwd::fused::kRows;
    return (int)(tiles < max_blocks ? tiles : max_blocks);
}

// Fused backward + weight gradients of n_heads (<= kMaxHeads) width-128 networks over the same input rows: two kernel
// launches (the layers split over them) and one ordered reduction into dkernels / dbiases (accumulated into, like
// nfx_launch_wgrad_batch).  dk / db: 5 pointers per head; `partial`: n_heads x nfx_mlp128_fused_partial_floats floats.
int nfx_launch_mlp128_bwd_fused(int in_kind, const float* xyz, const float* xyz_dir, long long n, float xyz_scale,
                                const float* lxyz, int n_lights, int n_heads, const void* const* blobs, const int* out_dims,
                                const int* out_acts, const float* post_scales, const float* const* douts, float* partial,
                                int grid, float* const* dk, float* const* db, hipStream_t st) {
    if (n <= 0 || n_heads <= 0) return 0;
    if (n_heads > nfx::bwd::fused::kMaxHeads) return (int)hipErrorInvalidValue;
    return in_kind == 0 ? launch_fused<0>(xyz, xyz_dir, n, xyz_scale, lxyz, n_lights, n_heads, blobs, out_dims, out_acts,
                                          post_scales, douts, partial, grid, dk, db, st)
                        : launch_fused<1>(xyz, xyz_dir, n, xyz_scale, lxyz, n_lights, n_heads, blobs, out_dims, out_acts,
                                          post_scales, douts, partial, grid, dk, db, st);
}
}
