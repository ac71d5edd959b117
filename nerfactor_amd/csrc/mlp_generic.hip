// mlp_generic.hip — the network shapes the tuned kernels do NOT cover (round 4): any mlp.Network the reference can build
// (nerfactor/networks/mlp.py:24-50: widths, activations, skip_at anywhere; nerfactor/models/nerf.py:53-90: mlp_width,
// enc_depth, use_views = False, pos_enc = False) evaluated by ONE runtime-shaped fused kernel, forward only.
//
// The tuned kernels (nerf_mlp_v6.hip, mlp128.hip, lvis_v2.hip) are compile-time specialisations of one architecture
// each — register-resident activations, packed weight streams.  This one trades their speed for generality and keeps
// their arithmetic class: bf16 operands, fp32 accumulation and bias, one v_mfma_f32_32x32x16_bf16 per
// (32 outputs x 32 rows x 16 inputs), transposed formulation H^T = W^T X^T (mlp_engine.hpp):
//   * one wave = 32 rows, no workgroup-level synchronisation at all (waves are independent);
//   * activations live in the wave's own LDS area (sized from the network's real widths) as ROW-MAJOR bf16 — the network
//     input (kept for the skip concatenations) and ONE hidden buffer: a layer accumulates ALL its output tiles in
//     registers (k-group outer, tile inner: the B operand of a group — a plain ds_read_b128 of the lane's row — is read
//     once for all tiles) and only then overwrites its input with its output, four 8-byte stores per lane and tile;
//   * weights are packed on the host in LOGICAL feature order (no permutation is needed: the operand comes from LDS,
//     not from the previous tile's accumulators) as 1-KiB A fragments, ONE stream in consumption order that each wave
//     pulls through its own LDS ring with the DMA path, three groups of four fragments ahead of its MFMAs;
//   * the layer table (input widths, tiles, activation, fragment and bias offsets) is a kernel argument.
// Limits: network input <= 576 features, hidden widths <= 512, <= 16 layers, output <= 512 (and the wave's two activation rows
// beside the weight ring within the 160 KiB of LDS: checked per launch).
// nfx_embed is the Embedder (embedder.py:23-47) as its own kernel, with the point generation o + d z folded in.
//
// Backward (mlp_generic_bwd_kernel + mlp_generic_wgrad_kernel + one ordered reduction; mlp_generic.hpp has the
// workspace layout): the same wave re-computes its 32 rows' forward, turns dLoss/dy into the output-layer gradient and
// walks the layers back with the TRANSPOSED weight fragments — dH^T = W dZ^T is the forward's loop with another weight
// stream — multiplying by the activation's derivative taken from the stored bf16 outputs.  Every layer's input and
// gradient go to the workspace through the transposing LDS read (tr16.hpp), 1 KiB contiguous per 16 features; the
// weight-gradient kernel then needs no LDS at all: one wave per (64 x 64 block of dW, row split), all MFMA operands
// 16-byte loads.  Deterministic: fixed split count per problem shape, ordered reductions, no atomics.
//
// NFX_PREC_FP32_NATIVE: the same kernels instantiated with fp32 activations in LDS, fp32 fragments (2 KiB per
// k-step) and the native fp32 matrix instruction v_mfma_f32_32x32x2_f32 — eight per k-step, k = 8 g + i on both
// operands — i.e. the reference's own arithmetic (trainvali.py:273-285 differentiates in fp32), forward AND backward,
// for any shape including the shipped ones.  157 TFLOP/s is that instruction's peak; at one wave per SIMD its 64-cycle
// issue hides the loop's scalar work, so this path is MFMA-bound where the bf16 one is issue-bound.
//
// NFX_PREC_FP32 (round 5): fp32-class arithmetic on the bf16 matrix pipe, the hi / lo operand pairs of mlp_x3.hpp in
// the runtime-shaped kernels.  Activations, gradients and the workspace stay fp32 exactly as in the native mode; a
// weight fragment is the same 2 KiB, pre-split by the packer into [hi plane | lo plane] (hi = bf16(w), lo = bf16(w - hi),
// lane order of the bf16 fragment: the 8 consecutive k of a lane are the same in both instructions), the B operand — 8
// consecutive fp32 features of the lane's row — is split in registers once per k-group for all output tiles, and a
// k-step is three v_mfma_f32_32x32x16_bf16 (a_lo b_hi + a_hi b_lo + a_hi b_hi) = 96 matrix-pipe cycles instead of the
// 512 of eight fp32 instructions.  Operands carry 16 significant bits: gradients within 1e-3 of the reference's
// (tests/test_gpu_reference_grads.py), at a fifth of the native mode's matrix time.
#include "mlp_engine.hpp"
#include "lds_dma.hpp"
#include "mlp_generic.hpp"
#include "tr16.hpp"

// This file is compiled three times (build time: one translation unit per operand mode, in parallel): as itself
// (NFX_GENERIC_TU = 0: the bf16 instantiations, the mode-independent kernels and the C launch entry points) and through
// mlp_generic_x3.hip / mlp_generic_native.hip (NFX_GENERIC_TU = 1 | 2: the fp32-class / native-fp32 instantiations).
#ifndef NFX_GENERIC_TU
#define NFX_GENERIC_TU 0
#endif

namespace nfx {
namespace generic {

// One wave per workgroup (waves are independent: no barrier anywhere).  LDS of a wave: the weight ring, then 32 rows x
// (x_pitch + h_pitch) bytes — the network input and ONE hidden buffer that every layer updates in place — the pitches
// sized by the HOST from the network's real widths (Args::x_pitch / h_pitch: features x element size + 16, rows 4 banks
// apart) — bf16: a 256-wide network keeps 4 waves per CU resident (33.5 KiB each), fp32: 2 (66 KiB).
// LDS is addressed through address_space(3) pointers THROUGHOUT: a generic pointer that the compiler cannot trace back to
// the shared array (a select between two buffers is enough) becomes a flat load, which waits on vmcnt — i.e. on the
// weight ring's look-ahead — before every MFMA.
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) bf16x8 lds_bf16x8;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;

// What differs between the operand modes M: kBf16: bf16 operands, v_mfma_f32_32x32x16_bf16; kNative: fp32 operands,
// 8 x v_mfma_f32_32x32x2_f32 per k-step; kX3: the fp32 DATA layout (LDS, workspace, fragment size) of kNative with
// hi / lo bf16 pairs as MFMA operands, 3 x v_mfma_f32_32x32x16_bf16 per k-step.
enum : int { kBf16 = 0, kX3 = 1, kNative = 2 };        // = Args::f32 = the C-ABI's NFX_PREC_BF16 / _FP32 / _FP32_NATIVE
template <int M>
struct P {
    static constexpr bool kF32 = M != kBf16;                  // fp32 activations / workspace / 2-KiB fragments
    static constexpr int kElem = kF32 ? 4 : 2;                // bytes per activation / weight
    static constexpr int kFrag = 32 * 16 * kElem;             // one k-step's A fragment (32 outputs x 16 inputs): 1 or 2 KiB
    static constexpr int kGroupBytes = kGroup * kFrag;
    static constexpr int kPieces = kGroupBytes / 1024;        // DMA pieces per group: 4 or 8
    // ring depth in groups: 3 slots of 4 KiB (bf16) or 8 KiB (fp32 modes: a group is 12 (pairs) or 32 (native) MFMAs per
    // output tile there, so two groups ahead are > 700 matrix-pipe cycles of cover for an L2 fetch)
#ifndef NFX_GENERIC_RING_F32
#define NFX_GENERIC_RING_F32 3
#endif
#ifndef NFX_GENERIC_RING_BF16
#define NFX_GENERIC_RING_BF16 3
#endif
    static constexpr int kRingGroups = kF32 ? NFX_GENERIC_RING_F32 : NFX_GENERIC_RING_BF16;
    static constexpr int kRingBytes = kRingGroups * kGroupBytes;
    static constexpr int kStep = 16 * kElem;                  // bytes of one k-step in a row: 32 or 64
    static constexpr int kTile = 32 * kElem;                  // bytes of 32 features in a row
    static constexpr int kWsFeat = 32 * kElem;                // workspace bytes per (feature, row tile)
};
// (every shape the reference builds with widths <= 512 fits one wave's area beside the ring — a 512-wide NeRF's colour head reads
//  539 features into 256 units; the host refuses the launch with the numbers when a layer table does not: nfx_launch_mlp_generic*)
static_assert(P<kBf16>::kRingBytes + 32 * ((kMaxIn * 2 + 16) + (kMaxHidden * 2 + 16)) <= 160 * 1024, "LDS");
static_assert(P<kNative>::kRingBytes + 32 * ((kMaxIn * 4 + 16) + (kMaxHidden / 2 * 4 + 16)) <= 160 * 1024, "LDS");
static_assert(P<kNative>::kRingBytes + 32 * ((kMaxIn / 2 * 4 + 16) + (kMaxHidden * 4 + 16)) <= 160 * 1024, "LDS");
static_assert(kGroup == 4, "a group = 4 fragments = 4 (bf16) or 8 (fp32 modes) 1-KiB DMA pieces");

// The weight stream: the whole network's fragments in consumption order (mlp_generic.hpp: every tile padded to whole
// groups of kGroup), copied global -> LDS by the DMA path (lds_dma.hpp) ahead of the MFMAs — a register-staged prefetch
// cannot rotate its buffers without waiting for the loads it just issued; an LDS slot is only an address.  The stream is
// circular: behind its last group the ring already fetches the next row tile's first.
//
// Round 5, fp32 modes: ONE ring per workgroup of NW waves (NW = 4, 2 or 1: what fits the LDS beside NW activation areas;
// bf16 keeps NW = 1, see generic_waves).  Round 4 gave
// every wave its own ring: each wave then issues every 1-KiB DMA piece of the network itself, 4 (bf16) or 8 (fp32 modes)
// global_load_lds per 4 (12, 32) MFMAs, and at 60-180 cycles of issue per piece (MI355X_MICROARCH.md) the DMA issue, not
// the matrix pipe, set the pace (bf16: 13 % matrix-pipe busy at 384 TFLOP/s).  The NW waves of a workgroup run the same
// network on NW different 32-row tiles in lock step — identical control flow, since every trip count comes from the layer
// table — so a group is fetched ONCE, each wave issuing pieces / NW of it, and consumed by all:
//   acquire(g): s_waitcnt vmcnt(my pieces of the groups behind g still allowed in flight) ; s_barrier   -> every wave's
//               pieces of group g have landed AND every wave has finished reading group g - 1 (release below) -> the
//               slot of g - 1 is refilled right here, with group g - 1 + kRingGroups;
//   ... A fragments of group g from its slot, MFMAs ...
//   release():  s_waitcnt lgkmcnt(0)   (this wave's reads of the slot have returned before it can reach the next barrier).
// vmcnt is in order, so "at most k pieces outstanding" = the older ones have landed; the wave's other VMEM operations
// can only make that wait stricter.  Every instantiation computes the same sums in the same order: bit-identical to r04.
template <int M, int NW>
struct Ring {
    static constexpr int kMine = P<M>::kPieces / NW;     // DMA pieces of every group this wave issues: 8, 4, 2 or 1
    static constexpr int kR = P<M>::kRingGroups;
    static_assert(P<M>::kPieces % NW == 0 && kMine >= 1, "pieces per wave");
    const char* next;    // next group to fetch (wave-uniform)
    const char* begin;
    const char* end;
    const lds_char* lds_ptr; // the ring, as a pointer (reads) ...
    unsigned lds;            // ... and as the LDS address M0 takes
    unsigned lane_off;
    unsigned mine;       // byte offset of this wave's pieces inside a group
    int slot;            // the group the MFMAs read next
    int fill;            // the slot the next fetched group goes to
    __device__ __forceinline__ void fetch() {
        const unsigned dst = lds + (unsigned)fill * P<M>::kGroupBytes + mine;
        if constexpr (kMine == 8) {
            lds_dma_pieces<4>(lane_off, next + mine, dst);
            lds_dma_pieces<4>(lane_off, next + mine + 4096, dst + 4096u);
        } else {
            lds_dma_pieces<kMine>(lane_off, next + mine, dst);
        }
        next += P<M>::kGroupBytes;
        if (next == end) next = begin;
        fill = fill == kR - 1 ? 0 : fill + 1;
    }
    __device__ __forceinline__ void start(lds_char* ring, const char* stream, int n_frags, int lane, int wave) {
        lds_ptr = ring;
        lds = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
        lane_off = (unsigned)lane * 16u;
        mine = (unsigned)wave * (kMine * 1024u);
        begin = next = stream;
        end = stream + (size_t)n_frags * P<M>::kFrag;
        slot = fill = 0;
#pragma unroll
        for (int i = 0; i < (NW > 1 ? kR - 1 : kR); ++i) fetch();
    }
    // group `slot` is ready.  NW > 1: for every wave of the workgroup, and one more group is fetched into the slot read
    // before it (kR - 1 groups ahead).  NW = 1 (a private ring): the slot is refilled by release() as soon as this wave's own
    // reads have returned (kR groups ahead, no barrier) — round 4's protocol.
    __device__ __forceinline__ void acquire() {
        if constexpr (NW > 1) {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((kR - 2) * kMine) : "memory");
            fetch();
        } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kR - 1) * kMine) : "memory");
        }
    }
    __device__ __forceinline__ const lds_char* group() const { return lds_ptr + slot * P<M>::kGroupBytes + lane_off; }
    __device__ __forceinline__ void release() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (NW == 1) fetch();      // (fill == slot: the ring is full)
        slot = slot == kR - 1 ? 0 : slot + 1;
    }
};
// kX3: eight consecutive fp32 features of a row -> the bf16 pair (mlp_x3.hpp: hi = bf16(v), lo = bf16(v - hi); v - hi is exact)
struct HiLo {
    bf16x8 hi, lo;
};
__device__ __forceinline__ HiLo split8(f32x4 a, f32x4 b) {
    HiLo r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r.hi[j] = (__bf16)a[j];
        r.lo[j] = (__bf16)(a[j] - (float)r.hi[j]);
        r.hi[4 + j] = (__bf16)b[j];
        r.lo[4 + j] = (__bf16)(b[j] - (float)r.hi[4 + j]);
    }
    return r;
}
// acc += a b with both operands as pairs: small terms first, the a_lo b_lo term (2^-18 relative) dropped
__device__ __forceinline__ f32x16 mma_x3(const bf16x8& ah, const bf16x8& al, const HiLo& b, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b.lo, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b.hi, acc, 0, 0, 0);
}
// One 32 x 32 output tile: acc += W_tile^T [h | x], kg_h groups of four k-steps over the previous layer's output, then
// kg_x over the network input.  A: the ring; B: 64 consecutive features of the lane's own row in LDS.  Nothing in the
// loop depends on the k-step but immediate offsets: pad steps multiply zero fragments with whatever finite number the
// row holds there (the activation area is zeroed once, then only ever holds activations).
// fp32: a fragment is [half][lane][4 floats] — lane (m, g) holds W[16 s + 8 g + 4 half + r][m] — and MFMA i of a k-step
// contracts k = 8 g + i on both operands.
template <int M, int NW>
__device__ __forceinline__ f32x16 tile_mma(f32x16 acc, Ring<M, NW>& w, int kg_h, int kg_x, const lds_char* hsrc, const lds_char* xsrc) {
    for (int gi = 0; gi < kg_h + kg_x; ++gi) {
        const lds_char* bsrc = gi < kg_h ? hsrc + gi * (kGroup * P<M>::kStep) : xsrc + (gi - kg_h) * (kGroup * P<M>::kStep);
        w.acquire();
        const lds_char* grp = w.group();
        if constexpr (M == kBf16) {
            bf16x8 af[kGroup], bf[kGroup];
#pragma unroll
            for (int j = 0; j < kGroup; ++j) {
                af[j] = *reinterpret_cast<const lds_bf16x8*>(grp + j * 1024);
                bf[j] = *reinterpret_cast<const lds_bf16x8*>(bsrc + j * 32);
            }
#pragma unroll
            for (int j = 0; j < kGroup; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[j], bf[j], acc, 0, 0, 0);
        } else if constexpr (M == kX3) {      // fragment = [hi plane | lo plane], 1 KiB each, bf16 lane order
            bf16x8 ah[kGroup], al[kGroup];
            HiLo b[kGroup];
#pragma unroll
            for (int j = 0; j < kGroup; ++j) {
                ah[j] = *reinterpret_cast<const lds_bf16x8*>(grp + j * 2048);
                al[j] = *reinterpret_cast<const lds_bf16x8*>(grp + j * 2048 + 1024);
                {
                    const f32x4 f0 = *reinterpret_cast<const lds_f32x4*>(bsrc + j * 64), f1 = *reinterpret_cast<const lds_f32x4*>(bsrc + j * 64 + 16);
                    b[j] = split8(f0, f1);
                }
            }
#pragma unroll
            for (int j = 0; j < kGroup; ++j) acc = mma_x3(ah[j], al[j], b[j], acc);
        } else {
#pragma unroll
            for (int j = 0; j < kGroup; ++j) {
                const f32x4 a0 = *reinterpret_cast<const lds_f32x4*>(grp + j * 2048), a1 = *reinterpret_cast<const lds_f32x4*>(grp + j * 2048 + 1024);
                const f32x4 b0 = *reinterpret_cast<const lds_f32x4*>(bsrc + j * 64), b1 = *reinterpret_cast<const lds_f32x4*>(bsrc + j * 64 + 16);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[i], acc, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[i], acc, 0, 0, 0);
            }
        }
        // the slot is refilled only after every read of it has RETURNED (the compiler may sink MFMAs below this point,
        // not memory operations above it); the wait overlaps the first MFMAs of the group
        w.release();
    }
    return acc;
}
// A whole layer, k-group OUTER and output tile INNER: the four B operands of a group are read from LDS once and swept
// over the NT tiles' accumulators (the stream holds a layer's fragments in exactly this order), so a hidden layer costs
// 1 + 1/NT LDS reads per MFMA instead of 2 — and because every output tile is complete before any is written, the layer's
// output can overwrite its input IN PLACE: one hidden buffer per wave instead of two.
// `tile_done(t, acc)` is called once per output tile after the last k-group (the accumulators never leave this function:
// handed out by reference they end up in scratch memory).
template <int M, int NW, int NT, bool ROLLED, class TileDone>
__device__ __forceinline__ void layer_mma(Ring<M, NW>& w, int kg_h, int kg_x, const lds_char* hsrc, const lds_char* xsrc, TileDone tile_done) {
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
    for (int gi = 0; gi < kg_h + kg_x; ++gi) {
        const lds_char* bsrc = gi < kg_h ? hsrc + gi * (kGroup * P<M>::kStep) : xsrc + (gi - kg_h) * (kGroup * P<M>::kStep);
        if constexpr (M == kBf16) {
            bf16x8 bf[kGroup];
#pragma unroll
            for (int j = 0; j < kGroup; ++j) bf[j] = *reinterpret_cast<const lds_bf16x8*>(bsrc + j * 32);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                w.acquire();
                const lds_char* grp = w.group();
                bf16x8 af[kGroup];
#pragma unroll
                for (int j = 0; j < kGroup; ++j) af[j] = *reinterpret_cast<const lds_bf16x8*>(grp + j * 1024);
#pragma unroll
                for (int j = 0; j < kGroup; ++j) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[j], bf[j], acc[t], 0, 0, 0);
                w.release();     // (every read of the slot has returned)
            }
        } else if constexpr (M == kX3) {
            // the group's B operand is split ONCE (24 VALU per k-step) for all NT tiles; the A pairs come split from the blob
            HiLo b[kGroup];
#pragma unroll
            for (int j = 0; j < kGroup; ++j) {
                const f32x4 f0 = *reinterpret_cast<const lds_f32x4*>(bsrc + j * 64), f1 = *reinterpret_cast<const lds_f32x4*>(bsrc + j * 64 + 16);
                b[j] = split8(f0, f1);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                w.acquire();
                const lds_char* grp = w.group();
                bf16x8 ah[kGroup], al[kGroup];
#pragma unroll
                for (int j = 0; j < kGroup; ++j) {
                    ah[j] = *reinterpret_cast<const lds_bf16x8*>(grp + j * 2048);
                    al[j] = *reinterpret_cast<const lds_bf16x8*>(grp + j * 2048 + 1024);
                }
#pragma unroll
                for (int j = 0; j < kGroup; ++j) acc[t] = mma_x3(ah[j], al[j], b[j], acc[t]);
                w.release();
            }
        } else {
            f32x4 b0[kGroup], b1[kGroup];
#pragma unroll
            for (int j = 0; j < kGroup; ++j) {
                b0[j] = *reinterpret_cast<const lds_f32x4*>(bsrc + j * 64);
                b1[j] = *reinterpret_cast<const lds_f32x4*>(bsrc + j * 64 + 16);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                w.acquire();
                const lds_char* grp = w.group();
#pragma unroll
                for (int j = 0; j < kGroup; ++j) {
                    const f32x4 a0 = *reinterpret_cast<const lds_f32x4*>(grp + j * 2048), a1 = *reinterpret_cast<const lds_f32x4*>(grp + j * 2048 + 1024);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[j][i], acc[t], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[j][i], acc[t], 0, 0, 0);
                }
                w.release();
            }
        }
    }
    if constexpr (!ROLLED) {      // forward kernel: the epilogues unrolled (measured faster there: 378 vs 365 TFLOP/s on 256 x 8)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            asm volatile("" ::: "memory");     // (one tile's epilogue at a time: hoisted together, eight tiles' bias loads spill)
            tile_done(t, acc[t]);
        }
    } else {
        // backward kernel: ONE copy of the epilogue code, looped over the tiles (unrolled it is ~25 K instructions per row
        // tile of a 256 x 8 backward, and the instruction cache, not the matrix pipe, sets the pace: 4.5 ms against
        // 2.8); the tile's accumulator is picked by a wave-uniform branch, 16 register moves
#pragma nounroll
        for (int t = 0; t < NT; ++t) {
            f32x16 c;
#define NFX_PICK(K)                        \
    if constexpr (NT > K) {                \
        if (t == K) {                      \
            c = acc[K];                    \
            asm volatile("" : "+v"(c));    \
        }                                  \
    }
            NFX_PICK(0) NFX_PICK(1) NFX_PICK(2) NFX_PICK(3) NFX_PICK(4) NFX_PICK(5) NFX_PICK(6) NFX_PICK(7)
            NFX_PICK(8) NFX_PICK(9) NFX_PICK(10) NFX_PICK(11) NFX_PICK(12) NFX_PICK(13) NFX_PICK(14) NFX_PICK(15)
#undef NFX_PICK
            tile_done(t, c);
        }
    }
}
// (1 <= n <= kMaxHidden / 32 output tiles: one instantiation each, chosen by a wave-uniform branch.  9 .. 16 tiles — widths
//  288 .. 512, round 5: all 16 accumulator tiles = 256 registers of a one-wave-per-SIMD kernel — exist in the WIDE
//  instantiations only (one wave per workgroup; nfx_launch_mlp_generic* picks them for a network with such a layer): the
//  kernels of every narrower network keep their register counts — the bf16 backward 240, two waves per SIMD)
#define NFX_WIDE_TILES(k, CALL) \
    case k:                      \
        if constexpr (WIDE) {    \
            CALL(k);             \
        }                        \
        break;
#define NFX_FOR_TILE_COUNT(n, CALL) \
    switch (n) {                     \
        case 1: CALL(1); break;      \
        case 2: CALL(2); break;      \
        case 3: CALL(3); break;      \
        case 4: CALL(4); break;      \
        case 5: CALL(5); break;      \
        case 6: CALL(6); break;      \
        case 7: CALL(7); break;      \
        case 8: CALL(8); break;      \
        NFX_WIDE_TILES(9, CALL) NFX_WIDE_TILES(10, CALL) NFX_WIDE_TILES(11, CALL) NFX_WIDE_TILES(12, CALL) \
        NFX_WIDE_TILES(13, CALL) NFX_WIDE_TILES(14, CALL) NFX_WIDE_TILES(15, CALL) NFX_WIDE_TILES(16, CALL) \
        default: break;              \
    }
static_assert(kMaxHidden == 512, "NFX_FOR_TILE_COUNT covers 1 .. 16 tiles");

// the wave's activation area starts as zeros: every feature a pad k-step can touch is a finite number
__device__ __forceinline__ void zero_lds(lds_char* p, int bytes, int lane) {
    for (int o = lane * 16; o < bytes; o += 64 * 16) *reinterpret_cast<lds_f32x4*>(p + o) = f32x4{0.f, 0.f, 0.f, 0.f};
}
// the lane's row of the network input -> LDS, zero padded to `feats`; lane half g takes every other 8 features
template <int M>
__device__ __forceinline__ void load_x(const float* __restrict__ src, int d_in, int feats, lds_char* dst, int g) {
    for (int c0 = 8 * g; c0 < feats; c0 += 16) {
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = src[c0 + j < d_in ? c0 + j : d_in - 1];     // (unconditional: eight loads in flight)
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = c0 + j < d_in ? f[j] : 0.f;
        if constexpr (P<M>::kF32) {
            *reinterpret_cast<lds_f32x4*>(dst + c0 * 4) = f32x4{f[0], f[1], f[2], f[3]};
            *reinterpret_cast<lds_f32x4*>(dst + c0 * 4 + 16) = f32x4{f[4], f[5], f[6], f[7]};
        } else {
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (__bf16)f[j];
            *reinterpret_cast<lds_bf16x8*>(dst + c0 * 2) = v;
        }
    }
}
// bias of the lane's 16 outputs of a tile (D row of register q: (q&3) + 8 (q>>2) + 4 g) through wave-uniform (scalar)
// loads: a vector load here would put a compiler-placed vmcnt(0) — a drain of the weight ring — into every tile
__device__ __forceinline__ void load_bias(const float* __restrict__ bt, int g, float* v) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int c = (q & 3) + 8 * (q >> 2);
        const float lo = bt[c], hi = bt[c + 4];
        v[q] = g ? hi : lo;
    }
}

// The activation code is wave-uniform: ONE branch per tile around sixteen straight-line evaluations — a switch inside
// the per-element loop is if-converted into "evaluate relu, sigmoid and softplus, select", ~50 VALU per element.
__device__ __forceinline__ void activate16(float* v, int act) {
    if (act == 1) {
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = fmaxf(v[q], 0.f);
    } else if (act == 2) {
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = sigmoidf(v[q]);
    } else if (act == 3) {
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = softplusf(v[q]);
    }
}
// features beyond the layer's width (last tile only) are exact zeros
__device__ __forceinline__ void zero_pad16(float* v, int t, int g, int n_out) {
    if (32 * t + 32 > n_out) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (32 * t + (q & 3) + 8 * (q >> 2) + 4 * g >= n_out) v[q] = 0.f;
    }
}
// 16 floats (register q = feature (q&3) + 8 (q>>2) + 4 g of the tile) -> the lane's row in LDS: four 8- or 16-byte stores
template <int M>
__device__ __forceinline__ void store_row16(lds_char* row_tile, int g, const float* v) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if constexpr (P<M>::kF32) {
            *reinterpret_cast<lds_f32x4*>(row_tile + (8 * q + 4 * g) * 4) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
        } else {
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (__bf16)v[4 * q + j];
            *reinterpret_cast<lds_bf16x4*>(row_tile + (8 * q + 4 * g) * 2) = o;
        }
    }
}

// A layer with an odd tile count leaves features [32 NT, 32 NT + 32) of its last k-group to the next layer's pad k-steps
// (zero fragments): they must be finite — a stale Inf / NaN of a wider layer or an earlier row tile times zero is NaN.
template <int M, int NT>
__device__ __forceinline__ void zero_pad_tile(lds_char* hrow, int g) {
    if constexpr (NT & 1) {
        const float z[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        store_row16<M>(hrow + P<M>::kTile * NT, g, z);
    }
}

// forward of one layer with NT output tiles; `hrow` = the lane's row of THE hidden buffer: read (k-steps over the previous
// layer's output) and, once all tiles are accumulated, overwritten with this layer's output
template <int M, int NW, int NT>
__device__ __forceinline__ void forward_layer(const Args& a, const Layer& L, bool last, Ring<M, NW>& w, lds_char* hrow, const lds_char* xrow,
                                              int g, float* yrow) {
    // (more than 8 tiles: the looped epilogue — sixteen unrolled ones beside 256 accumulator registers spill)
    layer_mma<M, NW, NT, (NT > 8)>(w, L.ks_h / kGroup, L.ks_x / kGroup, hrow + g * (P<M>::kStep / 2), xrow + g * (P<M>::kStep / 2),
                              [&](int t, const f32x16& acc) {
        // D: lane = row p (+ half g), register q = output feature 32 t + (q&3) + 8 (q>>2) + 4 g
        float bias[16], v[16];
        load_bias(a.biases + L.b_off + 32 * t, g, bias);
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = acc[q] + bias[q];
        activate16(v, L.act);
        if (last) {
            if (yrow) {
                float* dst = yrow + 32 * t + 4 * g;
                if (32 * t + 32 <= L.n_out) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) dst[(q & 3) + 8 * (q >> 2)] = v[q];
                } else {
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        if (32 * t + (q & 3) + 8 * (q >> 2) + 4 * g < L.n_out) dst[(q & 3) + 8 * (q >> 2)] = v[q];
                }
            }
        } else {
            zero_pad16(v, t, g, L.n_out);
            store_row16<M>(hrow + P<M>::kTile * t, g, v);
        }
    });
    if (!last) zero_pad_tile<M, NT>(hrow, g);
}

// One workgroup = NW waves, each with its own 32-row tile and its own activation area in LDS, ONE weight ring (Ring above).
// Every wave runs the same number of row-tile iterations (the ring's barriers are workgroup barriers): a wave whose tile
// lies past the end re-computes the last tile and stores nothing.
template <int M, int NW, bool WIDE>
__global__ __launch_bounds__(64 * NW) void mlp_generic_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, g = lane >> 5, p = lane & 31;
    const int wave = NW == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // (NW = 1: every LDS address keeps its compile-time base)
    const int x_pitch = a.x_pitch, h_pitch = a.h_pitch;
    lds_char* ring = (lds_char*)smem;
    lds_char* xb = ring + P<M>::kRingBytes + wave * 32 * (x_pitch + h_pitch);   // [32][x_pitch]  network input
    lds_char* hb = xb + 32 * x_pitch;                   // [32][h_pitch]  THE hidden buffer (updated in place)
    const long long n_tiles_rows = (a.n + 31) / 32;
    Ring<M, NW> w;
    w.start(ring, a.weights, a.n_frags, lane, wave);
    zero_lds(xb, 32 * (x_pitch + h_pitch), lane);
    for (long long base = (long long)blockIdx.x * NW; base < n_tiles_rows; base += (long long)gridDim.x * NW) {
        const bool mine = base + wave < n_tiles_rows;
        const long long row0 = (mine ? base + wave : n_tiles_rows - 1) * 32;
        {
            const long long r = row0 + p < a.n ? row0 + p : a.n - 1;
            load_x<M>(a.x + r * a.ld_x, a.d_in, (a.d_in + 15) / 16 * 16, xb + p * x_pitch, g);
        }
        float* yrow = (mine && row0 + p < a.n) ? a.y + (row0 + p) * a.ld_y + a.col0 : nullptr;
        for (int l = 0; l < a.n_layers; ++l) {
            const Layer L = a.layer[l];
            const bool last = l == a.n_layers - 1;
#define NFX_FWD(NT) forward_layer<M, NW, NT>(a, L, last, w, hb + p * h_pitch, xb + p * x_pitch, g, yrow)
            NFX_FOR_TILE_COUNT(L.n_tiles, NFX_FWD)
#undef NFX_FWD
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's look-ahead must not outlive the workgroup's LDS
}

// ------------------------------------------------------------------------------------------------ backward
// d[q] *= d act / d logit at logit z[q]  (the output layer) ...
__device__ __forceinline__ void scale_by_act_grad_logit16(float* d, const float* z, int act) {
    if (act == 1) {
#pragma unroll
        for (int q = 0; q < 16; ++q) d[q] = z[q] > 0.f ? d[q] : 0.f;
    } else if (act == 2) {
#pragma unroll
        for (int q = 0; q < 16; ++q) { const float s = sigmoidf(z[q]); d[q] *= s * (1.f - s); }
    } else if (act == 3) {
#pragma unroll
        for (int q = 0; q < 16; ++q) d[q] *= sigmoidf(z[q]);
    }
}
// ... and the same from the activated output y[q] (hidden layers: what the workspace holds)
__device__ __forceinline__ void scale_by_act_grad_output16(float* d, const float* y, int act) {
    if (act == 1) {
#pragma unroll
        for (int q = 0; q < 16; ++q) d[q] = y[q] > 0.f ? d[q] : 0.f;
    } else if (act == 2) {
#pragma unroll
        for (int q = 0; q < 16; ++q) d[q] *= y[q] * (1.f - y[q]);
    } else if (act == 3) {
#pragma unroll
        for (int q = 0; q < 16; ++q) d[q] *= 1.f - expf(-y[q]);
    }
}
// [32 rows][F features] row-major in the wave's LDS -> feature rows [frow, frow + F) of the wave's workspace tile.
// bf16: 16-lane group q takes rows 4 q .. 4 q + 3 (then + 16): lane i supplies row 4 q + (i >> 2), features f0 + 4 (i & 3) ..,
// receives feature f0 + i of the four rows = one 8-byte store.
// fp32 (round 5; round 4 stored the 16 accumulators of a tile as 16 dwords straight from the registers — 1260 vector-memory
// instructions per 32 rows of the light-visibility network, and at 60-180 cycles of issue each THEY set the kernel's pace,
// not the matrix pipe: profiles/r05/README.md): lane takes feature f0 + (lane >> 3), rows 4 (lane & 7) .. + 3 — four
// ds_read_b32 a pitch apart, ONE 16-byte store; a wave instruction = 8 whole feature rows, 1 KiB contiguous.
template <int M>
__device__ __forceinline__ void store_blocked(const lds_char* lds, int pitch, int F, char* wst, int frow, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (P<M>::kF32) {
        typedef __attribute__((address_space(3))) float lds_float;
        const int fi = lane >> 3, r0 = 4 * (lane & 7);
        const lds_char* s0 = lds + r0 * pitch + fi * 4;
        const lds_char* s1 = s0 + pitch;
        const lds_char* s2 = s1 + pitch;
        const lds_char* s3 = s2 + pitch;
        float* dst = reinterpret_cast<float*>(wst) + (size_t)(frow + fi) * 32 + r0;
        for (int f0 = 0; f0 < F; f0 += 8) {
            const f32x4 v = {*(const lds_float*)(s0 + f0 * 4), *(const lds_float*)(s1 + f0 * 4), *(const lds_float*)(s2 + f0 * 4),
                             *(const lds_float*)(s3 + f0 * 4)};
            *reinterpret_cast<f32x4*>(dst + (size_t)f0 * 32) = v;
        }
    } else {
        const int i = lane & 15, q = lane >> 4;
        const lds_char* src = lds + (4 * q + (i >> 2)) * pitch + 8 * (i & 3);
        char* dst = wst + ((size_t)(frow + i) * 32 + 4 * q) * 2;
        for (int f0 = 0; f0 < F; f0 += 16) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(src + f0 * 2));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(src + 16 * pitch + f0 * 2));
            *reinterpret_cast<s16x4*>(dst + (size_t)f0 * 64) = lo;
            *reinterpret_cast<s16x4*>(dst + (size_t)f0 * 64 + 32) = hi;
        }
    }
}

// the backward kernel's forward of one layer: as forward_layer, plus the workspace copies (fp32: straight from the
// registers) and, for the output layer, dZ = dy * act'(logit) in place of the activation
template <int M, int NW, int NT>
__device__ __forceinline__ void recompute_layer(const BwdArgs& ba, int l, Ring<M, NW>& w, lds_char* hrow, const lds_char* xrow, int g, int p,
                                                bool live, const float* dyr, char* wst) {
    const Args& a = ba.f;
    const Layer L = a.layer[l];
    const bool last = l == a.n_layers - 1;
    layer_mma<M, NW, NT, true>(w, L.ks_h / kGroup, L.ks_x / kGroup, hrow + g * (P<M>::kStep / 2), xrow + g * (P<M>::kStep / 2),
                             [&](int t, const f32x16& acc) {
        float bias[16], v[16];
        load_bias(a.biases + L.b_off + 32 * t, g, bias);
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = acc[q] + bias[q];
        if (last) {      // dZ = dy * act'(logit); rows past n and features past the width contribute nothing
            float d[16];                     // (every index clamped INTO the row: the last row's pad columns lie past the buffer)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int f = 32 * t + 4 * g + (q & 3) + 8 * (q >> 2);
                d[q] = dyr[f < L.n_out ? f : 0];
            }
            scale_by_act_grad_logit16(d, v, L.act);
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = live ? d[q] : 0.f;
        } else {
            activate16(v, L.act);
        }
        zero_pad16(v, t, g, L.n_out);
        store_row16<M>(hrow + P<M>::kTile * t, g, v);
    });
    zero_pad_tile<M, NT>(hrow, g);
}
// dZ_{l-1} = (dZ_l W_l^T over the previous layer's NT output tiles) * act'(H_{l-1}), written over dZ_l in place
template <int M, int NW, int NT>
__device__ __forceinline__ void dgrad_layer(const BwdArgs& ba, int l, Ring<M, NW>& w, lds_char* hrow, int g, int p, char* wst) {
    const Args& a = ba.f;
    const int kg_o = pad_group(2 * a.layer[l].n_tiles) / kGroup;
    layer_mma<M, NW, NT, true>(w, kg_o, 0, hrow + g * (P<M>::kStep / 2), hrow, [&](int mt, const f32x16& acc) {
        // the previous layer's outputs at the lane's 16 features, back from the workspace (same wave, own cache lines)
        float d[16], y[16];
        if constexpr (P<M>::kF32) {
            const float* hy = reinterpret_cast<const float*>(wst) + (size_t)(ba.b[l - 1].h_row + 32 * mt + 4 * g) * 32 + p;
#pragma unroll
            for (int q = 0; q < 16; ++q) y[q] = hy[((q & 3) + 8 * (q >> 2)) * 32];
        } else {
            const __bf16* hy = reinterpret_cast<const __bf16*>(wst) + (size_t)(ba.b[l - 1].h_row + 32 * mt + 4 * g) * 32 + p;
#pragma unroll
            for (int q = 0; q < 16; ++q) y[q] = (float)hy[((q & 3) + 8 * (q >> 2)) * 32];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) d[q] = acc[q];
        scale_by_act_grad_output16(d, y, a.layer[l - 1].act);
        store_row16<M>(hrow + P<M>::kTile * mt, g, d);
    });
    zero_pad_tile<M, NT>(hrow, g);
}

template <int M, int NW, bool WIDE>
__global__ __launch_bounds__(64 * NW) void mlp_generic_bwd_kernel(BwdArgs ba) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Args& a = ba.f;
    const int lane = threadIdx.x & 63, g = lane >> 5, p = lane & 31;
    const int wave = NW == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // (NW = 1: every LDS address keeps its compile-time base)
    const int x_pitch = a.x_pitch, h_pitch = a.h_pitch;
    lds_char* ring = (lds_char*)smem;
    lds_char* xb = ring + P<M>::kRingBytes + wave * 32 * (x_pitch + h_pitch);
    lds_char* hb = xb + 32 * x_pitch;                   // THE hidden buffer: activations forward, gradients backward
    lds_char* hrow = hb + p * h_pitch;
    const int fx = (a.d_in + 31) / 32 * 32, mx = fx / 32;
    Ring<M, NW> w;
    w.start(ring, a.weights, ba.stream_frags, lane, wave);   // forward fragments, then the transposed ones, as one stream
    zero_lds(xb, 32 * (x_pitch + h_pitch), lane);
    for (long long base = (long long)blockIdx.x * NW; base < ba.tiles; base += (long long)gridDim.x * NW) {
        // a wave past the last tile keeps step with its workgroup (the ring's barriers) on the last tile's rows; everything it
        // would store goes nowhere (live = false) or to the workspace's spare tile (index ba.tiles), which nothing reads
        const bool mine = base + wave < ba.tiles;
        const long long rt = mine ? base + wave : ba.tiles;
        const long long row0 = (mine ? rt : ba.tiles - 1) * 32;
        const bool live = mine && row0 + p < a.n;
        const long long r = row0 + p < a.n ? row0 + p : a.n - 1;
        char* wst = ba.ws + (size_t)rt * ba.feat_rows * P<M>::kWsFeat;
        load_x<M>(a.x + r * a.ld_x, a.d_in, fx, xb + p * x_pitch, g);
        store_blocked<M>(xb, x_pitch, fx, wst, 0, lane);
        // ---- forward; the last layer turns dy into its own gradient
        const float* dyr = ba.dy + r * ba.ld_dy + ba.col0_dy;
        for (int l = 0; l < a.n_layers; ++l) {
#define NFX_RECOMPUTE(NT) recompute_layer<M, NW, NT>(ba, l, w, hrow, xb + p * x_pitch, g, p, live, dyr, wst)
            NFX_FOR_TILE_COUNT(a.layer[l].n_tiles, NFX_RECOMPUTE)
#undef NFX_RECOMPUTE
            if (l + 1 < a.n_layers) store_blocked<M>(hb, h_pitch, a.layer[l].n_tiles * 32, wst, ba.b[l].h_row, lane);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stored activations are read back below (same wave, own lines)
        // ---- backward: the hidden buffer holds dZ of layer l.  The transposed fragments follow the forward ones in the
        // stream, last layer first, per layer the input-gradient tiles (tile-major) then the hidden tiles (k-group outer);
        // the input-gradient tiles are always computed (the stream does not skip) and stored only when dx is wanted —
        // except layer 0's, which end the stream and are cut off it by the host.
        bool dx_written = false;
        for (int l = a.n_layers - 1; l >= 0; --l) {
            const Layer L = a.layer[l];
            store_blocked<M>(hb, h_pitch, L.n_tiles * 32, wst, ba.b[l].dz_row, lane);
            const int kg_o = pad_group(2 * L.n_tiles) / kGroup;
            const lds_char* zsrc = hrow + g * (P<M>::kStep / 2);
            if (L.ks_x > 0 && (ba.dx || l > 0)) {
                for (int mt = 0; mt < mx; ++mt) {
                    f32x16 acc;
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
                    acc = tile_mma<M, NW>(acc, w, kg_o, 0, zsrc, zsrc);
                    if (live && ba.dx) {
                        float* dst = ba.dx + (row0 + p) * ba.ld_dx;
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            const int f = 32 * mt + (q & 3) + 8 * (q >> 2) + 4 * g;
                            if (f < a.d_in) dst[f] = dx_written ? dst[f] + acc[q] : acc[q];
                        }
                    }
                }
                dx_written = true;
            }
            if (l > 0) {
#define NFX_DGRAD(NT) dgrad_layer<M, NW, NT>(ba, l, w, hrow, g, p, wst)
                NFX_FOR_TILE_COUNT(a.layer[l - 1].n_tiles, NFX_DGRAD)
#undef NFX_DGRAD
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// one wave per (64 x 64 block of one layer's dW = 2 x 2 MFMA tiles, row split): dW[i, o] = sum over rows of IN[row, i] dZ[row, o].
// Four 16-byte loads feed four MFMAs (1 KiB of operands per MFMA and wave instead of 2); a block's second input / output
// tile may not exist (odd tile counts) — it then aliases the first and is not stored.  fp32: the same with 32-byte
// operands (8 rows) and eight v_mfma_f32_32x32x2_f32 per tile pair and 16 rows.
struct InTile { int frow, n_valid, i_base; };
__device__ __forceinline__ InTile in_tile(const WgradArgs& a, int l, int it) {
    const int mh = l > 0 ? a.layer[l - 1].n_tiles : 0, prev = l > 0 ? a.layer[l - 1].n_out : 0;
    if (it < mh) return {a.b[l - (l > 0)].h_row + 32 * it, prev - 32 * it, 32 * it};
    return {32 * (it - mh), a.d_in - 32 * (it - mh), prev + 32 * (it - mh)};
}
template <int M> struct Frag;
template <> struct Frag<kBf16> {
    bf16x8 v;
    __device__ __forceinline__ void load(const char* p) { v = *reinterpret_cast<const bf16x8*>(p); }
    __device__ __forceinline__ void ones() {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (__bf16)1.0f;
    }
    static __device__ __forceinline__ f32x16 mma(const Frag& a, const Frag& b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x16 mma_ones(const Frag& one, const Frag& b, f32x16 c) { return mma(one, b, c); }
};
template <> struct Frag<kX3> {      // 8 consecutive fp32 rows of one feature, split in registers (both operands are activations)
    HiLo v;
    __device__ __forceinline__ void load(const char* p) { v = split8(*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 16)); }
    __device__ __forceinline__ void ones() {
#pragma unroll
        for (int j = 0; j < 8; ++j) { v.hi[j] = (__bf16)1.0f; v.lo[j] = (__bf16)0.0f; }
    }
    static __device__ __forceinline__ f32x16 mma(const Frag& a, const Frag& b, f32x16 c) { return mma_x3(a.v.hi, a.v.lo, b.v, c); }
    static __device__ __forceinline__ f32x16 mma_ones(const Frag& one, const Frag& b, f32x16 c) {      // (1 has no lo half)
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(one.v.hi, b.v.lo, c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(one.v.hi, b.v.hi, c, 0, 0, 0);
    }
};
template <> struct Frag<kNative> {
    f32x4 lo, hi;
    __device__ __forceinline__ void load(const char* p) { lo = *reinterpret_cast<const f32x4*>(p); hi = *reinterpret_cast<const f32x4*>(p + 16); }
    __device__ __forceinline__ void ones() { lo = hi = f32x4{1.f, 1.f, 1.f, 1.f}; }
    static __device__ __forceinline__ f32x16 mma(const Frag& a, const Frag& b, f32x16 c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo[i], b.lo[i], c, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi[i], b.hi[i], c, 0, 0, 0);
        return c;
    }
    static __device__ __forceinline__ f32x16 mma_ones(const Frag& one, const Frag& b, f32x16 c) { return mma(one, b, c); }
};
template <int M>
__global__ __launch_bounds__(256) void mlp_generic_wgrad_kernel(WgradArgs a) {
    constexpr int E = P<M>::kElem;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, m = lane & 31, g = lane >> 5;
    // Which (64 x 64 block of dW = job, row split) this wave takes.  The jobs of one layer re-read each other's operands
    // (a 128 x 128 layer: 4 jobs, every workspace tile of the layer read by 2 of them; 2.6 reads per stored byte over the
    // light-visibility network), so the order decides how much of that is HBM traffic:
    //   map 0 (round 4): consecutive waves = consecutive SPLITS of one job — the re-readers of a tile sit `splits` waves
    //     apart, in other workgroups;
    //   map 1: the 4 waves of a workgroup = 4 consecutive JOBS of one split (same rows, in lock step), and the workgroups
    //     of a split are those the dispatcher hands to ONE XCD (workgroup b -> XCD b mod 8): every re-read is an L2 hit.
    // Same (job, split) sums either way: bit-identical results.
    int job, sp;
    if (a.map == 0) {
        const long long wid = (long long)blockIdx.x * 4 + wave;
        if (wid >= (long long)a.n_jobs * a.splits) return;
        job = (int)(wid / a.splits), sp = (int)(wid % a.splits);
    } else {
        const int per_split = (a.n_jobs + 3) / 4, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        sp = (j / per_split) * 8 + xcd, job = (j % per_split) * 4 + wave;
        if (sp >= a.splits || job >= a.n_jobs) return;
    }
    int l = 0;
    while (l + 1 < a.n_layers && a.b[l + 1].job0 <= job) ++l;
    const Layer L = a.layer[l];
    const int m_in = (l > 0 ? a.layer[l - 1].n_tiles : 0) + (L.ks_x ? (a.d_in + 31) / 32 : 0);
    const int o_pairs = (L.n_tiles + 1) / 2, local = job - a.b[l].job0, ip = local / o_pairs, op = local - ip * o_pairs;
    const bool two_i = 2 * ip + 1 < m_in, two_o = 2 * op + 1 < L.n_tiles;
    const InTile ti[2] = {in_tile(a, l, 2 * ip), in_tile(a, l, two_i ? 2 * ip + 1 : 2 * ip)};
    const int fb[2] = {a.b[l].dz_row + 64 * op, a.b[l].dz_row + 64 * op + (two_o ? 32 : 0)};
    const long long t0 = a.tiles * sp / a.splits, t1 = a.tiles * (sp + 1) / a.splits;
    const char* pa[2] = {a.ws + ((size_t)(ti[0].frow + m) * 32 + 8 * g) * E, a.ws + ((size_t)(ti[1].frow + m) * 32 + 8 * g) * E};
    const char* pb[2] = {a.ws + ((size_t)(fb[0] + m) * 32 + 8 * g) * E, a.ws + ((size_t)(fb[1] + m) * 32 + 8 * g) * E};
    const size_t tile_bytes = (size_t)a.feat_rows * 32 * E;
    f32x16 acc[2][2], accb[2];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[0][0][q] = acc[0][1][q] = acc[1][0][q] = acc[1][1][q] = accb[0][q] = accb[1][q] = 0.f;
    // the jobs of a layer's first input pair also sum their dZ columns: db[o] = sum over rows of 1 x dZ[row, o]
    const bool with_bias = ip == 0;
    Frag<M> ones;
    ones.ones();
#pragma unroll 2
    for (long long t = t0; t < t1; ++t) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const size_t off = t * tile_bytes + 16 * E * kk;
            Frag<M> a0, a1, b0, b1;
            a0.load(pa[0] + off); a1.load(pa[1] + off); b0.load(pb[0] + off); b1.load(pb[1] + off);
            acc[0][0] = Frag<M>::mma(a0, b0, acc[0][0]);
            acc[0][1] = Frag<M>::mma(a0, b1, acc[0][1]);
            acc[1][0] = Frag<M>::mma(a1, b0, acc[1][0]);
            acc[1][1] = Frag<M>::mma(a1, b1, acc[1][1]);
            if (with_bias) {
                accb[0] = Frag<M>::mma_ones(ones, b0, accb[0]);
                accb[1] = Frag<M>::mma_ones(ones, b1, accb[1]);
            }
        }
    }
    // D: column (lane & 31) = o, row (q&3) + 8 (q>>2) + 4 g = i
    if (with_bias && g == 0) {      // every row of accb holds the column sums: row 0 = register 0 of lanes 0..31
        float* db = a.partial + (size_t)sp * a.slice + a.b[l].db_off;
        if (64 * op + m < L.n_out) db[64 * op + m] = accb[0][0];
        if (two_o && 64 * op + 32 + m < L.n_out) db[64 * op + 32 + m] = accb[1][0];
    }
    float* dst = a.partial + (size_t)sp * a.slice + a.b[l].dw_off;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        if (x && !two_i) break;
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            if (y && !two_o) break;
            const int o = 64 * op + 32 * y + m;
            if (o >= L.n_out) continue;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int i = (q & 3) + 8 * (q >> 2) + 4 * g;
                if (i < ti[x].n_valid) dst[(size_t)(ti[x].i_base + i) * L.n_out + o] = acc[x][y][q];
            }
        }
    }
}

#if NFX_GENERIC_TU == 0      // the shape- and mode-independent kernels live in the first translation unit only
__global__ __launch_bounds__(256) void mlp_generic_wgrad_reduce_kernel(WgradArgs a) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.slice) return;
    float s = 0.f;
    for (int sp = 0; sp < a.splits; ++sp) s += a.partial[(size_t)sp * a.slice + idx];     // fixed order
    int l = 0;
    if (idx < a.dw_total) {
        while (l + 1 < a.n_layers && a.b[l + 1].dw_off <= idx) ++l;
        a.dw[l][idx - a.b[l].dw_off] += s;
    } else {
        while (l + 1 < a.n_layers && a.b[l + 1].db_off <= idx) ++l;
        a.db[l][idx - a.b[l].db_off] += s;
    }
}

// NFX_PREC_FP32_NATIVE fragments ([half][lane][4 floats]) -> NFX_PREC_FP32 fragments ([hi plane | lo plane] of [lane][8 bf16]),
// IN PLACE: thread (fragment, lane) reads exactly the 32 bytes it writes.  The device re-pack of an fp32-class blob is the
// native blob's gather (ops.DevicePacker) followed by this.
__global__ __launch_bounds__(256) void split_hilo_kernel(char* frags, long long n_frags) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if ((t >> 6) >= n_frags) return;
    char* f = frags + (t >> 6) * 2048 + (t & 63) * 16;
    const f32x4 a = *reinterpret_cast<const f32x4*>(f), b = *reinterpret_cast<const f32x4*>(f + 1024);
    const HiLo r = split8(a, b);
    *reinterpret_cast<bf16x8*>(f) = r.hi;
    *reinterpret_cast<bf16x8*>(f + 1024) = r.lo;
}

// One thread per OUTPUT ELEMENT, columns fastest: a wave writes 64 consecutive floats of a row (round 4 gave every thread a
// whole row — 27 or 63 stores a lane, each wave store scattered over 64 rows: 0.4 TB/s).  An element re-computes its
// vector (and, mode 3, the normalisation) and takes one half of sincos_cw: the same values bit for bit.
__global__ __launch_bounds__(256) void embed_kernel(EmbedArgs a) {
    const int d = (a.incl_input ? 3 : 0) + 6 * a.n_freqs;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= a.n * d) return;
    const long long row = e / d;
    const int c = (int)(e - row * d);
    const long long src = row / a.per_ray;
    float v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (a.mode == 0) v[k] = a.x[src * 3 + k];
        else if (a.mode == 1) v[k] = a.x[src * 3 + k] + a.dir[src * 3 + k] * a.z[row];
        else if (a.mode == 2) v[k] = a.dir[src * 3 + k];
        else v[k] = a.dir[(row - src * a.per_ray) * 3 + k] - a.x[src * 3 + k];   // light (row % per_ray) - point
    }
    if (a.mode == 3) {   // shape.py:128-131: safe_l2_normalize(lxyz - x), eps 1e-6
        const float inv = 1.0f / sqrtf(fmaxf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2], 1e-6f));
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] *= inv;
    }
    const int cb = c - (a.incl_input ? 3 : 0);        // column inside the bands: [sin x3 | cos x3] per band
    const int k = cb < 0 ? c : cb % 3;
    const float vk = k == 0 ? v[0] : k == 1 ? v[1] : v[2];
    float r = vk;
    if (cb >= 0) {
        const int f = cb / 6;
        float sn, cs;
        sincos_cw(vk * (float)(1 << f), sn, cs);      // 2^k x is exact in fp32, as in TensorFlow
        r = (cb - 6 * f) < 3 ? sn : cs;
    }
    a.out[row * a.ld_out + a.col0 + c] = r;
}

// d (embedding) / d v pulled back: dv = d[identity block] + sum over bands of 2^k (cos(2^k v) d[sin block] - sin(2^k v) d[cos block])
__global__ __launch_bounds__(256) void embed_bwd_kernel(EmbedArgs a, const float* d_out, float* dv) {
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= a.n) return;
    const float* d = d_out + row * a.ld_out + a.col0;
    float v[3], g[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = a.x[row * 3 + k];
    int c = 0;
    if (a.incl_input) {
#pragma unroll
        for (int k = 0; k < 3; ++k) g[k] = d[k];
        c = 3;
    }
    for (int f = 0; f < a.n_freqs; ++f) {
        const float s = (float)(1 << f);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float sn, cs;
            sincos_cw(v[k] * s, sn, cs);
            g[k] += s * (cs * d[c + k] - sn * d[c + 3 + k]);
        }
        c += 6;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) dv[row * 3 + k] = g[k];
}

#endif

}  // namespace generic
}  // namespace nfx

// ---- launch: one pair of functions per translation unit = per operand mode (NFX_GENERIC_TU), dispatched by the C entry
// points of translation unit 0
#define NFX_CAT_(a, b) a##b
#define NFX_CAT(a, b) NFX_CAT_(a, b)
extern "C" {
int nfx_generic_fwd_m0(const nfx::generic::Args*, int, int, int, hipStream_t);
int nfx_generic_fwd_m1(const nfx::generic::Args*, int, int, int, hipStream_t);
int nfx_generic_fwd_m2(const nfx::generic::Args*, int, int, int, hipStream_t);
int nfx_generic_bwd_m0(const nfx::generic::BwdArgs*, const nfx::generic::WgradArgs*, int, int, int, hipStream_t);
int nfx_generic_bwd_m1(const nfx::generic::BwdArgs*, const nfx::generic::WgradArgs*, int, int, int, hipStream_t);
int nfx_generic_bwd_m2(const nfx::generic::BwdArgs*, const nfx::generic::WgradArgs*, int, int, int, hipStream_t);
}
namespace {
template <int M, int NW, bool WIDE = false>
int launch_fwd(const nfx::generic::Args* args, int grid, int lds, hipStream_t st) {
    using namespace nfx::generic;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_generic_kernel<M, NW, WIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((mlp_generic_kernel<M, NW, WIDE>), dim3(grid), dim3(64 * NW), lds, st, *args);
    return 0;
}
template <int M, int NW, bool WIDE = false>
int launch_bwd(const nfx::generic::BwdArgs* ba, int grid, int lds, hipStream_t st) {
    using namespace nfx::generic;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_generic_bwd_kernel<M, NW, WIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((mlp_generic_bwd_kernel<M, NW, WIDE>), dim3(grid), dim3(64 * NW), lds, st, *ba);
    return 0;
}
bool wide_layers(const nfx::generic::Args& a) {      // a layer of more than 8 output tiles (width > 256)
    for (int l = 0; l < a.n_layers; ++l)
        if (a.layer[l].n_tiles > 8) return true;
    return false;
}
}  // namespace
extern "C" {
int NFX_CAT(nfx_generic_fwd_m, NFX_GENERIC_TU)(const nfx::generic::Args* args, int nw, int grid, int lds, hipStream_t st) {
    constexpr int M = NFX_GENERIC_TU;
#if NFX_GENERIC_TU != 0      // (bf16 keeps a private ring per wave: generic_waves)
    if (nw == 4) return launch_fwd<M, 4>(args, grid, lds, st);
    if (nw == 2) return launch_fwd<M, 2>(args, grid, lds, st);
#endif
    (void)nw;
    if (wide_layers(*args)) return launch_fwd<M, 1, true>(args, grid, lds, st);
    return launch_fwd<M, 1>(args, grid, lds, st);
}
int NFX_CAT(nfx_generic_bwd_m, NFX_GENERIC_TU)(const nfx::generic::BwdArgs* ba, const nfx::generic::WgradArgs* wa, int nw, int grid, int lds, hipStream_t st) {
    using namespace nfx::generic;
    constexpr int M = NFX_GENERIC_TU;
    int rc;
#if NFX_GENERIC_TU != 0
    if (nw == 4) rc = launch_bwd<M, 4>(ba, grid, lds, st);
    else if (nw == 2) rc = launch_bwd<M, 2>(ba, grid, lds, st);
    else
#endif
        if (wide_layers(ba->f)) rc = launch_bwd<M, 1, true>(ba, grid, lds, st);
    else rc = launch_bwd<M, 1>(ba, grid, lds, st);
    (void)nw;
    if (rc) return rc;
    if (wa->dw[0]) {     // (no gradient buffers: the caller wants dLoss/dx only)
        const long long waves = (long long)wa->n_jobs * wa->splits;
        const unsigned wgs = wa->map == 0 ? (unsigned)((waves + 3) / 4)
                                          : (unsigned)((wa->splits + 7) / 8 * ((wa->n_jobs + 3) / 4) * 8);
        hipLaunchKernelGGL(mlp_generic_wgrad_kernel<M>, dim3(wgs), dim3(256), 0, st, *wa);
    }
    return 0;
}
}

#if NFX_GENERIC_TU == 0
static_assert(nfx::generic::kBf16 == 0 && nfx::generic::kX3 == 1 && nfx::generic::kNative == 2, "NFX_GENERIC_TU = the operand mode");
// waves per workgroup: as many (4, 2, 1) as fit the 160 KiB of LDS beside the shared ring
static int generic_waves(const nfx::generic::Args& a, long long tiles) {
    using namespace nfx::generic;
    const int ring = a.f32 ? P<kNative>::kRingBytes : P<kBf16>::kRingBytes, act = 32 * (a.x_pitch + a.h_pitch);
    // bf16 keeps a private ring per wave: a group is only 4 MFMAs there, a workgroup barrier per group costs more than the
    // DMA pieces it saves, and the backward's workspace stores sit in the same in-order vmcnt queue the shared ring has to
    // drain to 2 (measured, r05 call K: 256 x 8 forward 380 against 384 TFLOP/s, 128 x 4 backward 126 against 173)
    if (!a.f32) return 1;
    for (int l = 0; l < a.n_layers; ++l)
        if (a.layer[l].n_tiles > 8) return 1;      // (more than 8 output tiles: instantiated for one wave per workgroup only)
    int nw = 4;
    while (nw > 1 && (ring + nw * act > 160 * 1024 || tiles < nw)) nw /= 2;
    return nw;
}
extern "C" {
int nfx_launch_mlp_generic(const nfx::generic::Args* args, int max_blocks, hipStream_t st) {
    using namespace nfx::generic;
    if (args->n <= 0) return 0;
    const long long tiles = (args->n + 31) / 32;
    const int nw = generic_waves(*args, tiles);
    const long long wgs = (tiles + nw - 1) / nw, cap = max_blocks / nw > 0 ? max_blocks / nw : 1;
    const int grid = (int)(wgs < cap ? wgs : cap);
    const int lds = (args->f32 ? P<kNative>::kRingBytes : P<kBf16>::kRingBytes) + nw * 32 * (args->x_pitch + args->h_pitch);
    if (lds > 160 * 1024) return -lds;      // (the C entry point turns this into NFX_ENOSUP with the numbers)
    const int rc = args->f32 == kBf16 ? nfx_generic_fwd_m0(args, nw, grid, lds, st)
                 : args->f32 == kX3 ? nfx_generic_fwd_m1(args, nw, grid, lds, st) : nfx_generic_fwd_m2(args, nw, grid, lds, st);
    return rc ? rc : (int)hipGetLastError();
}
int nfx_launch_mlp_generic_bwd(const nfx::generic::BwdArgs* ba, const nfx::generic::WgradArgs* wa, int max_blocks, hipStream_t st) {
    using namespace nfx::generic;
    if (ba->f.n <= 0) return 0;
    const int nw = generic_waves(ba->f, ba->tiles);
    const long long wgs = (ba->tiles + nw - 1) / nw, cap = max_blocks / nw > 0 ? max_blocks / nw : 1;
    const int grid = (int)(wgs < cap ? wgs : cap);
    const int lds = (ba->f.f32 ? P<kNative>::kRingBytes : P<kBf16>::kRingBytes) + nw * 32 * (ba->f.x_pitch + ba->f.h_pitch);
    if (lds > 160 * 1024) return -lds;
    const int rc = ba->f.f32 == kBf16 ? nfx_generic_bwd_m0(ba, wa, nw, grid, lds, st)
                 : ba->f.f32 == kX3 ? nfx_generic_bwd_m1(ba, wa, nw, grid, lds, st) : nfx_generic_bwd_m2(ba, wa, nw, grid, lds, st);
    if (rc) return rc;
    if (wa->dw[0]) hipLaunchKernelGGL(mlp_generic_wgrad_reduce_kernel, dim3((unsigned)((wa->slice + 255) / 256)), dim3(256), 0, st, *wa);
    return (int)hipGetLastError();
}
int nfx_launch_split_hilo(void* frags, long long n_frags, hipStream_t st) {
    if (n_frags <= 0) return 0;
    hipLaunchKernelGGL(nfx::generic::split_hilo_kernel, dim3((unsigned)((n_frags * 64 + 255) / 256)), dim3(256), 0, st, static_cast<char*>(frags), n_frags);
    return (int)hipGetLastError();
}
int nfx_launch_embed_bwd(const nfx::generic::EmbedArgs* a, const float* d_out, float* dv, hipStream_t st) {
    if (a->n <= 0) return 0;
    hipLaunchKernelGGL(nfx::generic::embed_bwd_kernel, dim3((unsigned)((a->n + 255) / 256)), dim3(256), 0, st, *a, d_out, dv);
    return (int)hipGetLastError();
}
int nfx_launch_embed(const nfx::generic::EmbedArgs* a, hipStream_t st) {
    if (a->n <= 0) return 0;
    const long long elems = a->n * ((a->incl_input ? 3 : 0) + 6 * a->n_freqs);
    hipLaunchKernelGGL(nfx::generic::embed_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, st, *a);
    return (int)hipGetLastError();
}
}
#endif  // NFX_GENERIC_TU == 0
