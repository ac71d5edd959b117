// nerf_bwd.hip — backward of the NeRF MLP (tape.gradient of trainvali.py:284 through Model._eval_nerf_at,
// nerfactor/models/nerf.py:256-290), ONE fused kernel per network call:
//   1. re-run the forward of the 128-point tile (point generation + posenc + 8x256 encoder + bottleneck +
//      rgb_out[0]); every activation is written FEATURE-MAJOR (bf16 [feature][point]) for the weight-gradient
//      GEMMs and only its 1-bit ReLU mask stays in registers (8 x 256 activations do not fit otherwise),
//   2. dZ(rgb_out[1]) = d rgbs[..., :3], dZ(sigma_out) = d rgbs[..., 3] (the compositing backward already applied
//      the sigmoid / relu derivatives),
//   3. dgrad chain dH_{l-1}^T = W_l dZ_l^T with the register-resident MFMA dataflow of the forward
//      (nerf_train_layout.hpp lists the transposed "layers"), masked by the ReLU bits, every dZ written feature-major.
// Sample positions carry no gradient (z_fine is under stop_gradient, nerf.py:145; rays are data).
#include <cstdlib>
#include "feat_store.hpp"
#include "rowsel.hpp"
#include "lds_dma.hpp"
#include "nerf_train_layout.hpp"

namespace nfx {
namespace bwd {

constexpr int kNerfNW = 4;  // one wave per SIMD (launch_bounds(256, 1): up to 512 registers)
constexpr int kNerfRows = kNerfNW * 32;
constexpr int kNerfBwdLds = 2 * kSlotBytes + nerf::kBiasFloats * 4;

// Forward layer of the re-computation: NT tiles; each tile's activation goes to the next B operand, to the
// feature-major workspace (rows feat0 + 32 t ..) and, for ReLU layers, to the mask words m[t>>1].
template <int KS1, int KS2, int NT, int NL_SELF, int NL_NEXT, bool RELU, int KS1A, int KS2A, int NTA>
__device__ __forceinline__ void fwd_layer(WStream& ws, int tid, const float* bias, const bf16x8 (&b1)[KS1A][1],
                                          const bf16x8 (&b2)[KS2A][1], bf16x8 (&bout)[NTA][1], const FeatStore& fs,
                                          int feat0, unsigned (&m)[NT / 2]) {
    const int h = (tid & 63) >> 5;
    static_for<0, NT>([&](auto T) {
        constexpr int t = decltype(T)::value;
        f32x16 acc[1];
        tile_raw<KS1, KS2, (t == NT - 1 ? NL_NEXT : NL_SELF), kNerfNW>(ws, tid, bias + 32 * t, b1, b2, acc);
        if constexpr (RELU) {
            const unsigned bits = relu_bits16(acc[0]);
            if constexpr (t & 1) m[t >> 1] |= bits << 16;
            else m[t >> 1] = bits;
        }
        acc_to_b<RELU, 1>(acc, bout[2 * t], bout[2 * t + 1]);
        store_tile(fs, feat0 + 32 * t, h, bout[2 * t][0], bout[2 * t + 1][0]);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// dgrad layer -> NT tiles of 32 input features, masked by `m` (MASK) and written feature-major at feat0;
// KS2 extra k-steps from b2 (the sigma-gradient slot).
template <int KS1, int KS2, int NT, int NL_SELF, int NL_NEXT, bool MASK, int KS1A, int KS2A, int NTA>
__device__ __forceinline__ void dgrad_layer(WStream& ws, int tid, const bf16x8 (&dz)[KS1A][1],
                                            const bf16x8 (&b2)[KS2A][1], const unsigned (&m)[NT / 2],
                                            bf16x8 (&dout)[NTA][1], const FeatStore& fs, int feat0) {
    const int h = (tid & 63) >> 5;
    static_for<0, NT>([&](auto T) {
        constexpr int t = decltype(T)::value;
        f32x16 acc[1];
        tile_init<KS1, KS2, (t == NT - 1 ? NL_NEXT : NL_SELF), kNerfNW>(
            ws, tid, [&](f32x16(&a)[1]) { zero_init<1>(a); }, dz, b2, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool on = MASK ? mask_bit(m, t, r) : true;
            dout[2 * t + (r >> 3)][0][r & 7] = (__bf16)(on ? acc[0][r] : 0.f);
        }
        mfma_operand_fence(dout[2 * t][0]);
        mfma_operand_fence(dout[2 * t + 1][0]);
        store_tile(fs, feat0 + 32 * t, h, dout[2 * t][0], dout[2 * t + 1][0]);
        __builtin_amdgcn_sched_barrier(0);
    });
}

__global__ __launch_bounds__(kNerfNW * 64, 1) void nerf_bwd_kernel(
    const float* __restrict__ rayo, const float* __restrict__ rayd, const float* __restrict__ zbuf, long long n_pts,
    int n_samples, const char* __restrict__ blob, const float4* __restrict__ d_rgbs, __bf16* __restrict__ wsp,
    long long ld, const int* /*list*/, const int* /*count*/) {   // (the ring kernels' signature; every point)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using namespace nerf;
    constexpr int NW = kNerfNW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
    float* bias_lds = reinterpret_cast<float*>(smem + 2 * kSlotBytes);
    {
        const float* bsrc = reinterpret_cast<const float*>(blob + kTrainWeightBytes);
        for (int i = tid; i < kBiasFloats; i += NW * 64) bias_lds[i] = bsrc[i];
    }
    WStream ws;
    ws.gbase = reinterpret_cast<const u32x4*>(blob);
    ws.gend = reinterpret_cast<const u32x4*>(blob + kTrainWeightBytes);
    ws.gnext = ws.gbase;
    ws.ring = smem;
    stream_prologue<kNL0, NW>(ws, tid);
    const long long n_tiles = (n_pts + kNerfRows - 1) / kNerfRows;
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long row = tile * kNerfRows + wave * 32 + p;  // < ld (ld is a multiple of kNerfRows)
        FeatStore fs;
        {
            unsigned long long ld2 = (unsigned long long)ld * 2, b = reinterpret_cast<unsigned long long>(wsp);
            asm volatile("" : "+s"(ld2), "+s"(b));
            fs.base = reinterpret_cast<char*>(b);
            fs.ld2 = ld2;
            fs.roff = (unsigned)(row * 4);   // pair layout: one dword per row and feature pair (feat_store.hpp)
        }
        const bool valid = row < n_pts;
        const long long mm = valid ? row : n_pts - 1;
        bf16x8 pe[4][1], pv[2][1];
        {
            const long long ray = mm / n_samples;
            const float zz = zbuf[mm];
            float x[3], d[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                d[k] = rayd[ray * 3 + k];
                x[k] = rayo[ray * 3 + k] + d[k] * zz;
            }
            posenc<10, 1>(x, h, 0, pe);
            posenc<4, 1>(d, h, 0, pv);
            store_posenc<10, 4>(fs, kOffPe, h, pe);
            store_posenc<4, 2>(fs, kOffPv, h, pv);
        }
        // ------------------------------------------------------------------ forward (re-computed)
        unsigned mk[8][4], mr[2], mnone[4];
        bf16x8 ha[16][1], hb[16][1];
        fwd_layer<4, 0, 8, kNL0, kNLH, true>(ws, tid, bias_lds + kBiasL0, pe, pe, ha, fs, kOffA + 0 * 256, mk[0]);
        fwd_layer<16, 0, 8, kNLH, kNLH, true>(ws, tid, bias_lds + kBiasL0 + 256 * 1, ha, pe, hb, fs, kOffA + 1 * 256, mk[1]);
        fwd_layer<16, 0, 8, kNLH, kNLH, true>(ws, tid, bias_lds + kBiasL0 + 256 * 2, hb, pe, ha, fs, kOffA + 2 * 256, mk[2]);
        fwd_layer<16, 0, 8, kNLH, kNLH, true>(ws, tid, bias_lds + kBiasL0 + 256 * 3, ha, pe, hb, fs, kOffA + 3 * 256, mk[3]);
        fwd_layer<16, 0, 8, kNLH, kNL5, true>(ws, tid, bias_lds + kBiasL0 + 256 * 4, hb, pe, ha, fs, kOffA + 4 * 256, mk[4]);
        fwd_layer<16, 4, 8, kNL5, kNLH, true>(ws, tid, bias_lds + kBiasL0 + 256 * 5, ha, pe, hb, fs, kOffA + 5 * 256, mk[5]);
        fwd_layer<16, 0, 8, kNLH, kNLH, true>(ws, tid, bias_lds + kBiasL0 + 256 * 6, hb, pe, ha, fs, kOffA + 6 * 256, mk[6]);
        fwd_layer<16, 0, 8, kNLH, kNLH, true>(ws, tid, bias_lds + kBiasL0 + 256 * 7, ha, pe, hb, fs, kOffA + 7 * 256, mk[7]);
        fwd_layer<16, 0, 8, kNLH, kNLH, false>(ws, tid, bias_lds + kBiasBott, hb, pe, ha, fs, kOffBott, mnone);  // bottleneck
        with_chunk<kNLR0, NW>(ws, tid, [](const char*) {});  // sigma_out tile: its value is not needed here
        bf16x8 r0[8][1];
        fwd_layer<16, 2, 4, kNLR0, kNLR1, true>(ws, tid, bias_lds + kBiasRgb0, ha, pv, r0, fs, kOffR0, mr);
        with_chunk<kNLD1, NW>(ws, tid, [](const char*) {});  // rgb_out[1] tile: skipped likewise
        // ------------------------------------------------------------------ output gradients
        bf16x8 dzo[1][1], dsg[1][1];
        {
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid && h == 0) g = d_rgbs[row];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                dzo[0][0][j] = (__bf16)0.f;
                dsg[0][0][j] = (__bf16)0.f;
            }
            dzo[0][0][0] = (__bf16)g.x;
            dzo[0][0][1] = (__bf16)g.y;
            dzo[0][0][2] = (__bf16)g.z;
            dsg[0][0][0] = (__bf16)g.w;
            if (h == 0) {
                st16(fs, kOffDRgb + 0, dzo[0][0][0]);
                st16(fs, kOffDRgb + 1, dzo[0][0][1]);
                st16(fs, kOffDRgb + 2, dzo[0][0][2]);
                st16(fs, kOffDSig, dsg[0][0][0]);
            }
        }
        // ------------------------------------------------------------------ dgrad chain
        bf16x8 dr0[8][1];
        dgrad_layer<1, 0, 4, kNLD1, kNLD2, true>(ws, tid, dzo, dzo, mr, dr0, fs, kOffDR0);            // D1: rgb_out[1]^T
        dgrad_layer<8, 0, 8, kNLD2, kNLD3, false>(ws, tid, dr0, dzo, mnone, ha, fs, kOffDBott);       // D2: into the bottleneck
        dgrad_layer<16, 1, 8, kNLD3, kNLDH, true>(ws, tid, ha, dsg, mk[7], hb, fs, kOffDZ + 7 * 256);  // D3: [bott | sigma]^T
        dgrad_layer<16, 0, 8, kNLDH, kNLDH, true>(ws, tid, hb, dsg, mk[6], ha, fs, kOffDZ + 6 * 256);  // enc[7]^T
        dgrad_layer<16, 0, 8, kNLDH, kNLDH, true>(ws, tid, ha, dsg, mk[5], hb, fs, kOffDZ + 5 * 256);  // enc[6]^T
        dgrad_layer<16, 0, 8, kNLDH, kNLDH, true>(ws, tid, hb, dsg, mk[4], ha, fs, kOffDZ + 4 * 256);  // enc[5][:256]^T
        dgrad_layer<16, 0, 8, kNLDH, kNLDH, true>(ws, tid, ha, dsg, mk[3], hb, fs, kOffDZ + 3 * 256);  // enc[4]^T
        dgrad_layer<16, 0, 8, kNLDH, kNLDH, true>(ws, tid, hb, dsg, mk[2], ha, fs, kOffDZ + 2 * 256);  // enc[3]^T
        dgrad_layer<16, 0, 8, kNLDH, kNLDH, true>(ws, tid, ha, dsg, mk[1], hb, fs, kOffDZ + 1 * 256);  // enc[2]^T
        dgrad_layer<16, 0, 8, kNLDH, kNL0, true>(ws, tid, hb, dsg, mk[0], ha, fs, kOffDZ + 0 * 256);   // enc[1]^T; next = L0
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Round 3: the same arithmetic with the weight stream on an LDS-DMA ring.
// The kernel above gives a chunk ONE chunk time to arrive (register-staged double buffer) and, because gfx9 counts loads
// and stores in one in-order `vmcnt`, every wait for a staged chunk is also a wait for the 8 activation stores issued
// before it.  Here chunk i + 5 is fetched by `global_load_lds_dwordx4` (lds_dma.hpp) while chunk i is consumed — 6 slots
// of 20 KiB (4 waves) or 24 KiB (8 waves) — and the wait behind chunk i names exactly what may still be in flight: the
// pieces of chunks i + 2 .. i + 5 and the 8 stores of each of the last four epilogues (<= 52 / 44 of the counter's 63;
// tests/test_cpu_ring_protocol.py replays the stream).  Chunks the backward does not use (sigma_out, rgb_out[1] forward
// tiles) are not fetched at all.  Same MFMA order and operands as the kernel above: bit-identical workspace and
// gradients (scripts/grad_identity.py, tests/test_gpu_train.py::test_ring_backward_kernels_equal_the_register_staged_ones).
// What the measurements of the round say bounds it (profiles/HISTORY.md section 3b): not the fetch distance but the 64 requests a CU's
// L1 keeps in flight towards L2 — hence non-temporal activation stores (feat_store.hpp: the weights stay in L2) and
// 8 waves per weight fetch.
namespace nring {
#ifndef NFX_NRING_D
#define NFX_NRING_D 5   // fetch distance in chunks (experiments: 2 .. 5; 6 would exceed the 6-bit counter)
#endif
constexpr int kSeq = 152, kD = NFX_NRING_D, kR = kD + 1, kEpiStores = 8;
// chunk i of a tile's sequence: first fragment in the train blob
constexpr int off(int i) {
    if (i < 72) return nerf::chunk_frag_offset(i);                 // enc[0..7], bottleneck
    if (i < 76) return nerf::chunk_frag_offset(i + 1);             // rgb_out[0] (the sigma chunk is skipped)
    const int j = i - 76;
    return nerf::kFrags + (j < 4 ? j * 4 : j < 12 ? 16 + (j - 4) * 8 : j < 20 ? 80 + (j - 12) * 20 : 240 + (j - 20) * 16);
}
static_assert(off(kSeq - 1) + 16 == nerf::kTrainFrags, "sequence covers the train blob");
// NW waves per workgroup = NW x 32 rows per tile on one weight stream.  NW = 4: one wave per SIMD.  NW = 8: two per
// SIMD (<= 256 registers each), half the L2 weight traffic per row, and one wave's epilogue under its partner's MFMAs.
template <int NW>
struct Cfg {
    // 1-KiB pieces per wave (every wave the same number: the wait counts are immediates).  NW = 8 rounds a chunk up to
    // a multiple of 8 fragments — the blob's own padding, or the first fragments of the next chunk (never past the
    // blob: the last chunk has 16)
    static constexpr int pieces(int i) {
        if (i >= kSeq || i < 0) return 0;
        const int j = i - 76;
        const int used = i < 8 ? 4 : i < 40 ? 16 : i < 48 ? 20 : i < 72 ? 16 : i < 76 ? 18
                       : j < 4 ? 1 : j < 12 ? 8 : j < 20 ? 17 : 16;
        return (used + NW - 1) / NW;
    }
    static constexpr int kSlot = (NW == 4 ? 20 : 24) * 1024;
    static constexpr int kLds = kR * kSlot + nerf::kBiasFloats * 4;
    // what may still be in flight when chunk i + 1 must have landed: the wait sits behind the MFMAs of chunk i and
    // before its epilogue, so the stores of epilogues max(0, i + 1 - kD) .. i - 1 are younger than the fetch of chunk
    // i + 1 (i = -1: the wait behind the priming fetches)
    static constexpr int allow(int i) {
        int n = kEpiStores * (i < 0 ? 0 : i < kD - 1 ? i : kD - 1);
        for (int j = i + 2; j <= i + kD; ++j) n += pieces(j);
        return n;
    }
    static constexpr int max_allow() {
        int m = 0;
        for (int i = 0; i < kSeq; ++i) m = allow(i) > m ? allow(i) : m;
        return m;
    }
    static constexpr int max_pieces() {
        int m = 0;
        for (int i = 0; i < kSeq; ++i) m = pieces(i) > m ? pieces(i) : m;
        return m;
    }
    static_assert(kLds <= 160 * 1024, "LDS");
    static_assert(max_allow() <= 63, "vmcnt is a 6-bit counter");
    static_assert(max_pieces() * NW * 1024 <= kSlot, "slot");
};

struct Ctx {
    char* smem;
    unsigned smem_lds;
    const char* blob;
    int lane, wave;   // wave: wave-uniform
    unsigned bias_addr;   // LDS byte address of bias[4 h]: opaque, so every tile's bias read is this register + an immediate
};
// mlp_engine.hpp:bias_init with the tile's offset as an immediate (the compiler otherwise forms the 76 per-tile lane
// addresses ahead of the loop and spills them: one scratch reload per chunk, each draining the DMA window)
__device__ __forceinline__ void bias_acc(const Ctx& cx, int off_floats, f32x16& acc) {
    typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;
    typedef __attribute__((address_space(3))) const char lds_cchar;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<lds_f32x4*>((lds_cchar*)(uintptr_t)cx.bias_addr + (off_floats + 8 * g) * 4);
        acc[4 * g + 0] = v[0];
        acc[4 * g + 1] = v[1];
        acc[4 * g + 2] = v[2];
        acc[4 * g + 3] = v[3];
    }
}

// fetch chunk F of the sequence into its slot (F % kR; last read by chunk F - kR, whose closing barrier every wave
// has passed: F is fetched during chunk F - kD = F - kR + 1, or before chunk 0 for the first kD)
template <int NW, int F>
__device__ __forceinline__ void fetch(const Ctx& cx) {
    if constexpr (F < kSeq) {
        constexpr int n = Cfg<NW>::pieces(F);
        unsigned long long base = reinterpret_cast<unsigned long long>(cx.blob);
        unsigned lds = cx.smem_lds;
        asm volatile("" : "+s"(base), "+s"(lds));   // per chunk: keeps the 152 address pairs out of the loop preheader
        const int piece0 = cx.wave * n;
        lds_dma_pieces<n>((unsigned)cx.lane * 16u, reinterpret_cast<const char*>(base) + (size_t)off(F) * 1024 + piece0 * 1024,
               lds + (unsigned)(F % kR) * Cfg<NW>::kSlot + (unsigned)piece0 * 1024u);
    }
}
// chunk I: acc (initialised by the caller) += W_tile [b1 ; b2]; returns with chunk I + 1 landed and published
template <int NW, int I, int KS1, int KS2, int KS1A, int KS2A>
__device__ __forceinline__ void chunk(const Ctx& cx, const bf16x8 (&b1)[KS1A][1], const bf16x8 (&b2)[KS2A][1],
                                      f32x16 (&acc)[1]) {
    fetch<NW, I + kD>(cx);
    const char* f0 = cx.smem + (I % kR) * Cfg<NW>::kSlot + cx.lane * 16;
    mma_k<KS1>(f0, b1, acc);
    if constexpr (KS2 > 0) mma_k<KS2>(f0 + KS1 * kFragBytes, b2, acc);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(Cfg<NW>::allow(I)) : "memory");
}

template <int NW, int I0, int KS1, int KS2, int NT, bool RELU, int KS1A, int KS2A, int NTA>
__device__ __forceinline__ void fwd_layer(const Ctx& cx, int bias_off, const bf16x8 (&b1)[KS1A][1],
                                          const bf16x8 (&b2)[KS2A][1], bf16x8 (&bout)[NTA][1], const FeatStore& fs,
                                          int feat0, unsigned (&m)[NT / 2]) {
    const int h = cx.lane >> 5;
    static_for<0, NT>([&](auto T) {
        constexpr int t = decltype(T)::value;
        f32x16 acc[1];
        bias_acc(cx, bias_off + 32 * t, acc[0]);
        chunk<NW, I0 + t, KS1, KS2>(cx, b1, b2, acc);
        if constexpr (RELU) {
            const unsigned bits = relu_bits16(acc[0]);
            if constexpr (t & 1) m[t >> 1] |= bits << 16;
            else m[t >> 1] = bits;
        }
        acc_to_b<RELU, 1>(acc, bout[2 * t], bout[2 * t + 1]);
        store_tile(fs, feat0 + 32 * t, h, bout[2 * t][0], bout[2 * t + 1][0]);   // kEpiStores dword stores
        __builtin_amdgcn_sched_barrier(0);
    });
}
template <int NW, int I0, int KS1, int KS2, int NT, bool MASK, int KS1A, int KS2A, int NTA>
__device__ __forceinline__ void dgrad_layer(const Ctx& cx, const bf16x8 (&dz)[KS1A][1], const bf16x8 (&b2)[KS2A][1],
                                            const unsigned (&m)[NT / 2], bf16x8 (&dout)[NTA][1], const FeatStore& fs,
                                            int feat0) {
    const int h = cx.lane >> 5;
    static_for<0, NT>([&](auto T) {
        constexpr int t = decltype(T)::value;
        f32x16 acc[1];
        zero_init<1>(acc);
        chunk<NW, I0 + t, KS1, KS2>(cx, dz, b2, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool on = MASK ? mask_bit(m, t, r) : true;
            dout[2 * t + (r >> 3)][0][r & 7] = (__bf16)(on ? acc[0][r] : 0.f);
        }
        mfma_operand_fence(dout[2 * t][0]);
        mfma_operand_fence(dout[2 * t + 1][0]);
        store_tile(fs, feat0 + 32 * t, h, dout[2 * t][0], dout[2 * t + 1][0]);   // kEpiStores dword stores
        __builtin_amdgcn_sched_barrier(0);
    });
}
}  // namespace nring

// LIST: the kernel's rows are the points list[0 .. *count) (ascending; rowsel below: the points whose upstream gradient
// is not all zeros), row c of the feature workspace is point list[c].  A point with a zero upstream gradient has zeros
// in every dZ, so it adds nothing to any weight gradient: the sums over the listed rows are the sums over all rows.
template <int NW, bool LIST>
__global__ __launch_bounds__(NW * 64, 1) void nerf_bwd_ring_kernel(
    const float* __restrict__ rayo, const float* __restrict__ rayd, const float* __restrict__ zbuf, long long n_pts,
    int n_samples, const char* __restrict__ blob, const float4* __restrict__ d_rgbs, __bf16* __restrict__ wsp,
    long long ld, const int* __restrict__ list, const int* __restrict__ count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using namespace nerf;
    constexpr int kRows = NW * 32;
    using RC = nring::Cfg<NW>;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* bias_lds = reinterpret_cast<float*>(smem + nring::kR * RC::kSlot);
    {
        const float* bsrc = reinterpret_cast<const float*>(blob + kTrainWeightBytes);
        for (int i = tid; i < kBiasFloats; i += NW * 64) bias_lds[i] = bsrc[i];
        __syncthreads();
    }
    typedef __attribute__((address_space(3))) char lds_char;
    nring::Ctx cx{smem, (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_char*)smem), blob, lane, wave, 0u};
    cx.bias_addr = (unsigned)(uintptr_t)(lds_char*)smem + (unsigned)(nring::kR * RC::kSlot) + 16u * (unsigned)h;
    long long n_act = n_pts;
    if constexpr (LIST) n_act = *count;
    const long long n_tiles = (n_act + kRows - 1) / kRows;
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long row = tile * kRows + wave * 32 + p;  // < ld (ld is a multiple of 256)
        FeatStore fs;
        {
            unsigned long long ld2 = (unsigned long long)ld * 2, b = reinterpret_cast<unsigned long long>(wsp);
            asm volatile("" : "+s"(ld2), "+s"(b));
            fs.base = reinterpret_cast<char*>(b);
            fs.ld2 = ld2;
            fs.roff = (unsigned)(row * 4);
        }
        const bool valid = row < n_act;
        long long mm = valid ? row : n_pts - 1;
        if constexpr (LIST) {
            if (valid) mm = list[row];
        }
        bf16x8 pe[4][1], pv[2][1];
        bf16x8 dzo[1][1], dsg[1][1];
        // every load of the tile and every store that is not a chunk epilogue happens HERE, before the ring is primed:
        // the counted waits below know of nothing else in the vmcnt window
        {
            const long long ray = mm / n_samples;
            const float zz = zbuf[mm];
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid && h == 0) g = d_rgbs[LIST ? mm : row];
            float x[3], d[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                d[k] = rayd[ray * 3 + k];
                x[k] = rayo[ray * 3 + k] + d[k] * zz;
            }
            posenc<10, 1>(x, h, 0, pe);
            posenc<4, 1>(d, h, 0, pv);
            store_posenc<10, 4>(fs, kOffPe, h, pe);
            store_posenc<4, 2>(fs, kOffPv, h, pv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                dzo[0][0][j] = (__bf16)0.f;
                dsg[0][0][j] = (__bf16)0.f;
            }
            dzo[0][0][0] = (__bf16)g.x;
            dzo[0][0][1] = (__bf16)g.y;
            dzo[0][0][2] = (__bf16)g.z;
            dsg[0][0][0] = (__bf16)g.w;
            if (h == 0) {
                st16(fs, kOffDRgb + 0, dzo[0][0][0]);
                st16(fs, kOffDRgb + 1, dzo[0][0][1]);
                st16(fs, kOffDRgb + 2, dzo[0][0][2]);
                st16(fs, kOffDSig, dsg[0][0][0]);
            }
            // the operands are final before the first DMA leaves: no compiler-placed vmcnt wait inside the window
            u32x4 w0 = __builtin_bit_cast(u32x4, dzo[0][0]), w1 = __builtin_bit_cast(u32x4, dsg[0][0]);
            asm volatile("" : "+v"(w0), "+v"(w1));
            dzo[0][0] = __builtin_bit_cast(bf16x8, w0);
            dsg[0][0] = __builtin_bit_cast(bf16x8, w1);
        }
        asm volatile("" : "+v"(cx.bias_addr));
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, nring::kD>([&](auto F) { nring::fetch<NW, decltype(F)::value>(cx); });
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(RC::allow(-1)) : "memory");   // chunk 0 landed
        // ------------------------------------------------------------------ forward (re-computed)
        unsigned mk[8][4], mr[2], mnone[4];
        bf16x8 ha[16][1], hb[16][1];
        nring::fwd_layer<NW, 0, 4, 0, 8, true>(cx, kBiasL0, pe, pe, ha, fs, kOffA + 0 * 256, mk[0]);
        nring::fwd_layer<NW, 8, 16, 0, 8, true>(cx, kBiasL0 + 256 * 1, ha, pe, hb, fs, kOffA + 1 * 256, mk[1]);
        nring::fwd_layer<NW, 16, 16, 0, 8, true>(cx, kBiasL0 + 256 * 2, hb, pe, ha, fs, kOffA + 2 * 256, mk[2]);
        nring::fwd_layer<NW, 24, 16, 0, 8, true>(cx, kBiasL0 + 256 * 3, ha, pe, hb, fs, kOffA + 3 * 256, mk[3]);
        nring::fwd_layer<NW, 32, 16, 0, 8, true>(cx, kBiasL0 + 256 * 4, hb, pe, ha, fs, kOffA + 4 * 256, mk[4]);
        nring::fwd_layer<NW, 40, 16, 4, 8, true>(cx, kBiasL0 + 256 * 5, ha, pe, hb, fs, kOffA + 5 * 256, mk[5]);
        nring::fwd_layer<NW, 48, 16, 0, 8, true>(cx, kBiasL0 + 256 * 6, hb, pe, ha, fs, kOffA + 6 * 256, mk[6]);
        nring::fwd_layer<NW, 56, 16, 0, 8, true>(cx, kBiasL0 + 256 * 7, ha, pe, hb, fs, kOffA + 7 * 256, mk[7]);
        nring::fwd_layer<NW, 64, 16, 0, 8, false>(cx, kBiasBott, hb, pe, ha, fs, kOffBott, mnone);  // bottleneck
        bf16x8 r0[8][1];
        nring::fwd_layer<NW, 72, 16, 2, 4, true>(cx, kBiasRgb0, ha, pv, r0, fs, kOffR0, mr);
        // ------------------------------------------------------------------ dgrad chain
        bf16x8 dr0[8][1];
        nring::dgrad_layer<NW, 76, 1, 0, 4, true>(cx, dzo, dzo, mr, dr0, fs, kOffDR0);               // D1: rgb_out[1]^T
        nring::dgrad_layer<NW, 80, 8, 0, 8, false>(cx, dr0, dzo, mnone, ha, fs, kOffDBott);          // D2: into the bottleneck
        nring::dgrad_layer<NW, 88, 16, 1, 8, true>(cx, ha, dsg, mk[7], hb, fs, kOffDZ + 7 * 256);    // D3: [bott | sigma]^T
        nring::dgrad_layer<NW, 96, 16, 0, 8, true>(cx, hb, dsg, mk[6], ha, fs, kOffDZ + 6 * 256);    // enc[7]^T
        nring::dgrad_layer<NW, 104, 16, 0, 8, true>(cx, ha, dsg, mk[5], hb, fs, kOffDZ + 5 * 256);   // enc[6]^T
        nring::dgrad_layer<NW, 112, 16, 0, 8, true>(cx, hb, dsg, mk[4], ha, fs, kOffDZ + 4 * 256);   // enc[5][:256]^T
        nring::dgrad_layer<NW, 120, 16, 0, 8, true>(cx, ha, dsg, mk[3], hb, fs, kOffDZ + 3 * 256);   // enc[4]^T
        nring::dgrad_layer<NW, 128, 16, 0, 8, true>(cx, hb, dsg, mk[2], ha, fs, kOffDZ + 2 * 256);   // enc[3]^T
        nring::dgrad_layer<NW, 136, 16, 0, 8, true>(cx, ha, dsg, mk[1], hb, fs, kOffDZ + 1 * 256);   // enc[2]^T
        nring::dgrad_layer<NW, 144, 16, 0, 8, true>(cx, hb, dsg, mk[0], ha, fs, kOffDZ + 0 * 256);   // enc[1]^T
    }
}

// ------------------------------------------------------------ the rows with a gradient, in ascending order (rowsel.hpp)
// d_rgbs of a point the composite gave no weight (alpha = 0: raw density <= 0) is four exact zeros — half the samples of a
// freshly initialised network, most of a fitted scene's.
struct HasGradient {
    const float4* g;
    __device__ bool operator()(long long r) const {
        const uint4 v = *reinterpret_cast<const uint4*>(g + r);
        return ((v.x | v.y | v.z | v.w) & 0x7fffffffu) != 0u;   // -0 is zero; a NaN is a gradient (and reaches the weights)
    }
    __device__ void visit(long long, bool) const {}
};
}  // namespace bwd

namespace rowsel {
__global__ __launch_bounds__(1024) void scan_kernel(int* __restrict__ block_count, int n_blocks, int* __restrict__ count) {
    __shared__ int s[1024];
    const int tid = threadIdx.x;
    int carry = 0;
    for (int base = 0; base < n_blocks; base += 1024) {
        const int i = base + tid;
        const int v = i < n_blocks ? block_count[i] : 0;
        s[tid] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const int t = tid >= d ? s[tid - d] : 0;
            __syncthreads();
            s[tid] += t;
            __syncthreads();
        }
        if (i < n_blocks) block_count[i] = carry + s[tid] - v;
        carry += s[1023];
        __syncthreads();
    }
    if (tid == 0) *count = carry;
}
int launch_scan(int* block_count, int n_blocks, int* count, hipStream_t st) {
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, block_count, n_blocks, count);
    return (int)hipGetLastError();
}
}  // namespace rowsel
}  // namespace nfx

extern "C" int nfx_option_int(const char* name, int dflt);   // capi.cpp

// the row-list workspace of nfx_launch_nerf_bwd (rowsel.hpp: count, block counts, n_pts indices)
extern "C" size_t nfx_nerf_bwd_list_bytes(long long n_pts) { return nfx::rowsel::workspace_bytes(n_pts); }

extern "C" int nfx_launch_nerf_bwd(const float* rayo, const float* rayd, const float* z, long long n_pts, int n_samples,
                                   const void* blob, const float* d_rgbs, void* wsp, long long ld, int max_blocks,
                                   hipStream_t st, void* list_ws) {
    using namespace nfx;
    if (n_pts <= 0) return 0;
    const long long tiles = (n_pts + bwd::kNerfRows - 1) / bwd::kNerfRows;
    const int grid = (int)(tiles < max_blocks ? tiles : max_blocks);
    // r03 default: weights through the LDS-DMA ring; NFX_NERF_BWD=0 keeps the register-staged kernel (identity reference)
    const bool use_ring = nfx_option_int("nerf_bwd", 1) != 0;            // (per call, like every knob: INTEGRATION.md)
    const int ring_nw = nfx_option_int("nerf_bwd_nw", 8) == 4 ? 4 : 8;
    // list_ws not null: only the points with a gradient (in list_ws: nfx_nerf_bwd_list_bytes), count = (int*)list_ws
    if (list_ws && !use_ring) return (int)hipErrorInvalidValue;
    int *count = static_cast<int*>(list_ws), *list = nullptr;
    if (list_ws) {
        list = rowsel::list_of(list_ws, n_pts);
        const int rc = rowsel::build(bwd::HasGradient{(const float4*)d_rgbs}, n_pts, list_ws, st);
        if (rc) return rc;
    }
    auto k = !use_ring ? bwd::nerf_bwd_kernel
             : ring_nw == 8 ? (list ? bwd::nerf_bwd_ring_kernel<8, true> : bwd::nerf_bwd_ring_kernel<8, false>)
                            : (list ? bwd::nerf_bwd_ring_kernel<4, true> : bwd::nerf_bwd_ring_kernel<4, false>);
    const int nw = use_ring ? ring_nw : bwd::kNerfNW;
    const int lds = !use_ring ? bwd::kNerfBwdLds : ring_nw == 8 ? bwd::nring::Cfg<8>::kLds : bwd::nring::Cfg<4>::kLds;
    const long long tiles_nw = (n_pts + nw * 32 - 1) / (nw * 32);
    const int grid_nw = (int)(tiles_nw < max_blocks ? tiles_nw : max_blocks);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k, dim3(grid_nw), dim3(nw * 64), lds, st, rayo, rayd, z, n_pts, n_samples,
                       (const char*)blob, (const float4*)d_rgbs, (__bf16*)wsp, ld, (const int*)list, (const int*)count);
    return (int)hipGetLastError();
}
