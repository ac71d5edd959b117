// mlp128_bwd.hip — backward of the width-128 surface MLPs (tape.gradient of trainvali.py:284 through
// shape.py:196-237 / nerfactor.py:377-411), as ONE fused kernel per network call:
//   1. re-run the forward for the tile (activations h0..h3 stay in registers: 4 x 32 VGPRs),
//   2. dZ_out = dOut * post_scale * act'(logit),
//   3. dgrad chain  dH_{l-1}^T = W_l dZ_l^T  with the SAME register-resident MFMA dataflow as the
//      forward (the dgrad "weights" are W_l packed as if it were a layer with in' = out, out' = in),
//      ReLU masks taken from the re-computed activations,
//   4. store the layer inputs X, h0..h3 and the pre-activation gradients dZ0..dZ3, dZ_out FEATURE-MAJOR
//      (bf16 [feature][row]) for the weight-gradient GEMMs of train.hip (nfx_wgrad_bf16).
// No gradient w.r.t. the inputs (points are data).  Light visibility uses the plain 90-dim input
// here (the per-point fold of the inference kernel is a forward-only optimisation).
#include <stdlib.h>

#include "geom.hpp"
#include "lds_dma.hpp"
#include "mlp128_layout.hpp"
#include "mlp128_train_layout.hpp"
#include "mlp_engine.hpp"
#include "feat_store.hpp"

namespace nfx {
namespace bwd {

constexpr int kNW = 4;  // one wave per SIMD: the re-computed activations + gradients need > 256 registers
constexpr int kRows = kNW * 32;

// dgrad layer: dH^T = W dZ^T (NT tiles of 32 input features), ReLU-masked by the activation `hact`
// the features belong to; result = next dZ (bf16, B layout).
template <int KS, int NL_SELF, int NL_NEXT>
__device__ __forceinline__ void dgrad_layer(WStream& ws, int tid, const bf16x8 (&dz)[8][1],
                                            const bf16x8 (&hact)[8][1], bf16x8 (&dout)[8][1]) {
    static_for<0, 4>([&](auto T) {
        constexpr int t = decltype(T)::value;
        f32x16 acc[1];
        tile_init<KS, 0, (t == 3 ? NL_NEXT : NL_SELF), kNW>(
            ws, tid, [&](f32x16(&a)[1]) { zero_init<1>(a); }, dz, dz, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float hv = (float)hact[2 * t + (r >> 3)][0][r & 7];
            dout[2 * t + (r >> 3)][0][r & 7] = (__bf16)(hv > 0.f ? acc[0][r] : 0.f);
        }
        mfma_operand_fence(dout[2 * t][0]);
        mfma_operand_fence(dout[2 * t + 1][0]);
    });
}

// IN_KIND 0: posenc10(xyz_scale*xyz) ; 1: [posenc10(xyz_scale*xyz), posenc4(normalize(lxyz_l - xyz_dir))]
template <int IN_KIND>
__global__ __launch_bounds__(kNW * 64, 1) void mlp128_bwd_kernel(
    const float* __restrict__ xyz, const float* __restrict__ xyz_dir, long long n, float xyz_scale,
    const float* __restrict__ lxyz, int n_lights, const char* __restrict__ blob, int out_dim, int out_act,
    float post_scale, const float* __restrict__ dout, __bf16* __restrict__ wsp, long long ld) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KSX = IN_KIND == 0 ? 4 : 6;
    using G = Geo<KSX>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
    float* bias_lds = reinterpret_cast<float*>(smem + 2 * kSlotBytes);
    {
        const float* bsrc = reinterpret_cast<const float*>(blob + G::kWeightBytes);
        for (int i = tid; i < G::kBiasFloats; i += kNW * 64) bias_lds[i] = bsrc[i];
    }
    WStream ws;
    ws.gbase = reinterpret_cast<const u32x4*>(blob);
    ws.gend = reinterpret_cast<const u32x4*>(blob + G::kWeightBytes);
    ws.gnext = ws.gbase;
    ws.ring = smem;
    stream_prologue<G::kNL0, kNW>(ws, tid);
    const long long n_rows = IN_KIND == 0 ? n : n * n_lights;
    const long long n_tiles = (n_rows + kRows - 1) / kRows;
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long row = tile * kRows + wave * 32 + p;  // always < ld (ld is a multiple of kRows)
        FeatStore fs;
        {
            unsigned long long ld2 = (unsigned long long)ld * 2, b = reinterpret_cast<unsigned long long>(wsp);
            asm volatile("" : "+s"(ld2), "+s"(b));
            fs.base = reinterpret_cast<char*>(b);
            fs.ld2 = ld2;
            fs.roff = (unsigned)(row * 4);   // pair layout: one dword per row and feature pair (feat_store.hpp)
        }
        const bool valid = row < n_rows;
        const long long rc = valid ? row : n_rows - 1;
        const long long pt = IN_KIND == 0 ? rc : rc / n_lights;
        float x[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) x[k] = xyz_scale * xyz[pt * 3 + k];
        bf16x8 xin[KSX][1];
        {
            bf16x8 pe[4][1];
            posenc<10, 1>(x, h, 0, pe);
#pragma unroll
            for (int s = 0; s < 4; ++s) xin[s][0] = pe[s][0];
            store_posenc<10, 4>(fs, 0, h, pe);
            // pad feature 63 — only where nothing else owns the slot: with IN_KIND == 1 feature 63 is ldir.x, written by
            // store_posenc<4, 2>(fs, 63) below from the OTHER lane half; two stores of different lanes to one address
            // have no defined order
            if constexpr (IN_KIND == 0) { if (h == 1) st16(fs, 63, (__bf16)0.f); }
        }
        if constexpr (IN_KIND == 1) {
            const int l = (int)(rc % n_lights);
            float d[3], xd[3], lp[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                xd[k] = xyz_dir[pt * 3 + k];
                lp[k] = lxyz[l * 3 + k];
            }
            dir_to(lp, xd, d);
            bf16x8 pl[2][1];
            posenc<4, 1>(d, h, 0, pl);
            xin[4][0] = pl[0][0];
            xin[5][0] = pl[1][0];
            store_posenc<4, 2>(fs, 63, h, pl);
            if (h == 1) {  // pad features 90..95
#pragma unroll
                for (int e = 90; e < 96; ++e) st16(fs, e, (__bf16)0.f);
            }
        }
        // ------------------------------------------------------------------ forward (re-computed)
        bf16x8 h0[8][1], h1[8][1], h2[8][1], h3[8][1];
        layer<KSX, 0, 4, G::kNL0, G::kNLH, true, kNW>(ws, tid, bias_lds, xin, xin, h0);
        layer<8, 0, 4, G::kNLH, G::kNLH, true, kNW>(ws, tid, bias_lds + 128, h0, xin, h1);
        layer<8, 0, 4, G::kNLH, G::kNL3, true, kNW>(ws, tid, bias_lds + 256, h1, xin, h2);
        layer<8, KSX, 4, G::kNL3, G::kNLO, true, kNW>(ws, tid, bias_lds + 384, h2, xin, h3);
        store_hidden<8>(fs, G::kOffH + 0, h, h0);
        store_hidden<8>(fs, G::kOffH + 128, h, h1);
        store_hidden<8>(fs, G::kOffH + 256, h, h2);
        store_hidden<8>(fs, G::kOffH + 384, h, h3);
        f32x16 logit[1];
        tile_raw<8, 0, G::kNLD, kNW>(ws, tid, bias_lds + 512, h3, xin, logit);
        // ------------------------------------------------------------------ dZ_out
        bf16x8 dzo[1][1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 4 * h + r;
            float g = 0.f;
            if (valid && f < out_dim) g = dout[row * out_dim + f] * post_scale * act_grad(logit[0][r], out_act);
            dzo[0][0][r] = (__bf16)g;
            dzo[0][0][4 + r] = (__bf16)0.f;
            FeatStore f4 = fs;
            f4.roff = fs.roff + (h ? (unsigned)(4 * fs.ld2) : 0u);
            st16(f4, G::kOffDZo + r, (__bf16)g);
        }
        // ------------------------------------------------------------------ dgrad chain
        bf16x8 dz3[8][1], dz2[8][1], dz1[8][1], dz0[8][1];
        static_for<0, 4>([&](auto T) {  // through the out layer: K = 16 padded output slots
            constexpr int t = decltype(T)::value;
            f32x16 acc[1];
            tile_init<1, 0, (t == 3 ? G::kNLH : G::kNLD), kNW>(
                ws, tid, [&](f32x16(&a)[1]) { zero_init<1>(a); }, dzo, dzo, acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float hv = (float)h3[2 * t + (r >> 3)][0][r & 7];
                dz3[2 * t + (r >> 3)][0][r & 7] = (__bf16)(hv > 0.f ? acc[0][r] : 0.f);
            }
            mfma_operand_fence(dz3[2 * t][0]);
            mfma_operand_fence(dz3[2 * t + 1][0]);
        });
        store_hidden<8>(fs, G::kOffDZ + 384, h, dz3);
        dgrad_layer<8, G::kNLH, G::kNLH>(ws, tid, dz3, h2, dz2);   // W3[:128, :]
        store_hidden<8>(fs, G::kOffDZ + 256, h, dz2);
        dgrad_layer<8, G::kNLH, G::kNLH>(ws, tid, dz2, h1, dz1);   // W2
        store_hidden<8>(fs, G::kOffDZ + 128, h, dz1);
        dgrad_layer<8, G::kNLH, G::kNL0>(ws, tid, dz1, h0, dz0);   // W1; next chunk = L0 of the next tile
        store_hidden<8>(fs, G::kOffDZ + 0, h, dz0);
        // (r03: all eight store_hidden groups issued here, behind the tile's last weight chunk, measured 2.98-4.4 ms per
        //  step against 2.75: the stores are better off spread between the chunks)
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Round 3: the same backward with its weights where the matrix pipe can reach them.  PMC of the kernel above on the
// 1 048 576-row light-visibility call: matrix pipe 13 % busy, half of the wave cycles parked in s_waitcnt / s_barrier —
// its 33 weight chunks per 128-row tile carry 4-14 MFMAs of work per wave each and the register-staged double buffer
// gives a chunk one chunk time (~300 cycles) to arrive from L2: ~1700 cycles per chunk.  Here
//   * the DGRAD half of the train blob (112 fragments) is RESIDENT in LDS: the dgrad chain runs with no weight traffic,
//     no wait and no barrier at all, its stores are fire-and-forget;
//   * the FORWARD half streams through a 5-slot ring of 8-KiB sub-chunks by LDS-DMA (global_load_lds_dwordx4, no VGPR
//     staging), fetched FOUR sub-chunks ahead with counted s_waitcnt vmcnt; a layer-3 tile (8 + KSX k-steps) is two
//     sub-chunks.  21 sub-chunks per tile, one barrier each;
//   * gfx950 counts loads AND stores on vmcnt, in order: the forward phase therefore issues no store at all (the
//     network input and h0..h3 are stored behind the logit tile, the row's dOut is loaded at the top of the tile), so the
//     counted waits see DMA pieces only.
// LDS: [ring 5 x 8 KiB | dgrad fragments 112 KiB | biases] = 157 824 bytes.  Same arithmetic in the same order as the
// kernel above: bit-identical gradients (scripts/grad_identity.py).
// ------------------------------------------------------------------------------------------------------------------
namespace ring {
constexpr int kSub = 21, kR = 5, kD = 4, kSlot = 8192, kResFrags = 4 * 4 + 3 * 32;
constexpr int kLds = kR * kSlot + kResFrags * 1024 + m128::kMainBiasFloats * 4;
static_assert(kLds <= 160 * 1024, "LDS");
// sub-chunk k of a tile: fragments and first fragment in the blob (P0 / P3 = fragments per layer-0 / layer-3 chunk)
template <int KSX>
struct Sub {
    using G = Geo<KSX>;
    static constexpr int frags(int k) {
        return k < 4 ? G::kP0 : k < 12 ? 8 : k < 20 ? ((k - 12) % 2 == 0 ? 8 : G::kP3 - 8) : 8;
    }
    static constexpr int off(int k) {
        return k < 4 ? k * G::kP0 : k < 8 ? 4 * G::kP0 + (k - 4) * 8 : k < 12 ? 4 * G::kP0 + 32 + (k - 8) * 8
             : k < 20 ? 4 * G::kP0 + 64 + ((k - 12) / 2) * G::kP3 + ((k - 12) % 2) * 8 : 4 * G::kP0 + 64 + 4 * G::kP3;
    }
    static constexpr int pieces(int k) { return frags(k) / kNW; }   // 1-KiB pieces per wave: 1 | 2
    static constexpr int allow(int k) {   // pieces that may still be in flight when sub-chunk k + 1 must have landed
        int n = 0;
        for (int j = 2; j <= kD; ++j) n += pieces((k + j) % kSub);
        return n;
    }
};
struct Ctx {
    char* smem;
    unsigned smem_lds;
    const char* blob;
    int lane, wave;   // wave: wave-uniform
    int cur;          // ring slot of the sub-chunk being consumed (wave-uniform)
};
// start of sub-chunk K: fetch sub-chunk K + kD into the slot kD ahead (last read kR - kD = 1 sub-chunk ago: every wave
// has passed the barrier that ended it)
template <int KSX, int K>
__device__ __forceinline__ void begin(const Ctx& cx) {
    constexpr int F = (K + kD) % kSub, n = Sub<KSX>::pieces(F);
    unsigned long long base = reinterpret_cast<unsigned long long>(cx.blob);
    unsigned lds = cx.smem_lds;
    asm volatile("" : "+s"(base), "+s"(lds));   // per sub-chunk: keeps the addresses out of the loop preheader
    int slot = cx.cur + kD;
    slot = slot >= kR ? slot - kR : slot;
    const int piece0 = cx.wave * n;
    lds_dma_pieces<n>((unsigned)cx.lane * 16u, reinterpret_cast<const char*>(base) + (size_t)Sub<KSX>::off(F) * 1024 + piece0 * 1024,
           lds + (unsigned)slot * kSlot + (unsigned)piece0 * 1024u);
}
template <int KSX, int K>
__device__ __forceinline__ void end(Ctx& cx) {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(Sub<KSX>::allow(K)) : "memory");
    cx.cur = cx.cur + 1 == kR ? 0 : cx.cur + 1;
}
// acc += A(current slot, KS fragments) x b[b0 ...]
template <int KS, int KSA>
__device__ __forceinline__ void mma(const Ctx& cx, const bf16x8 (&b)[KSA][1], int b0, f32x16& acc) {
    const char* f0 = cx.smem + cx.cur * kSlot + cx.lane * 16;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(f0 + s * kFragBytes);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[b0 + s][0], acc, 0, 0, 0);
    }
}
// one forward layer of 4 tiles whose chunk is ONE sub-chunk each (K0 = its first sub-chunk)
template <int KSX, int K0, int KS, int KSA>
__device__ __forceinline__ void layer(Ctx& cx, const float* bias, const bf16x8 (&b)[KSA][1], bf16x8 (&out)[8][1]) {
    static_for<0, 4>([&](auto T) {
        constexpr int t = decltype(T)::value;
        begin<KSX, K0 + t>(cx);
        f32x16 acc[1];
        bias_init<1>(bias + 32 * t, cx.lane >> 5, acc);
        mma<KS>(cx, b, 0, acc[0]);
        end<KSX, K0 + t>(cx);
        acc_to_b<true, 1>(acc, out[2 * t], out[2 * t + 1]);
    });
}
// dgrad tile from the resident fragments: dH^T = W dZ^T, ReLU-masked by the activation the features belong to
template <int KS>
__device__ __forceinline__ void dgrad(const char* res_frag0, int lane, const bf16x8 (&dz)[8][1],
                                      const bf16x8 (&hact)[8][1], bf16x8 (&dout)[8][1]) {
    static_for<0, 4>([&](auto T) {
        constexpr int t = decltype(T)::value;
        const char* f0 = res_frag0 + (t * 8) * kFragBytes + lane * 16;
        f32x16 acc[1];
        zero_init<1>(acc);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(f0 + s * kFragBytes);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, dz[s][0], acc[0], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float hv = (float)hact[2 * t + (r >> 3)][0][r & 7];
            dout[2 * t + (r >> 3)][0][r & 7] = (__bf16)(hv > 0.f ? acc[0][r] : 0.f);
        }
        mfma_operand_fence(dout[2 * t][0]);
        mfma_operand_fence(dout[2 * t + 1][0]);
    });
}
}  // namespace ring

template <int IN_KIND>
__global__ __launch_bounds__(kNW * 64, 1) void mlp128_bwd_ring_kernel(
    const float* __restrict__ xyz, const float* __restrict__ xyz_dir, long long n, float xyz_scale,
    const float* __restrict__ lxyz, int n_lights, const char* __restrict__ blob, int out_dim, int out_act,
    float post_scale, const float* __restrict__ dout, __bf16* __restrict__ wsp, long long ld) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KSX = IN_KIND == 0 ? 4 : 6;
    using G = Geo<KSX>;
    using S = ring::Sub<KSX>;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* res = smem + ring::kR * ring::kSlot;                                   // dgrad fragments, resident
    float* bias_lds = reinterpret_cast<float*>(res + ring::kResFrags * 1024);
    {
        const float* bsrc = reinterpret_cast<const float*>(blob + G::kWeightBytes);
        for (int i = tid; i < G::kBiasFloats; i += kNW * 64) bias_lds[i] = bsrc[i];
        const u32x4* rsrc = reinterpret_cast<const u32x4*>(blob + (size_t)G::kFwdFrags * 1024);
        u32x4* rdst = reinterpret_cast<u32x4*>(res);
        for (int i = tid; i < ring::kResFrags * 64; i += kNW * 64) rdst[i] = rsrc[i];
        // sub-chunks 0 .. kD-1 of the first tile -> slots 0 .. kD-1
#pragma unroll
        for (int k = 0; k < ring::kD; ++k) {
            const u32x4* src = reinterpret_cast<const u32x4*>(blob + (size_t)S::off(k) * 1024);
            u32x4* dst = reinterpret_cast<u32x4*>(smem + k * ring::kSlot);
            for (int i = tid; i < S::frags(k) * 64; i += kNW * 64) dst[i] = src[i];
        }
        __syncthreads();
    }
    typedef __attribute__((address_space(3))) char lds_char;
    ring::Ctx cx{smem, (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_char*)smem), blob, lane, wave, 0};
    const long long n_rows = IN_KIND == 0 ? n : n * n_lights;
    const long long n_tiles = (n_rows + kRows - 1) / kRows;
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long row = tile * kRows + wave * 32 + p;  // always < ld (ld is a multiple of kRows)
        FeatStore fs;
        {
            unsigned long long ld2 = (unsigned long long)ld * 2, b = reinterpret_cast<unsigned long long>(wsp);
            asm volatile("" : "+s"(ld2), "+s"(b));
            fs.base = reinterpret_cast<char*>(b);
            fs.ld2 = ld2;
            fs.roff = (unsigned)(row * 4);   // pair layout: one dword per row and feature pair (feat_store.hpp)
        }
        const bool valid = row < n_rows;
        const long long rc = valid ? row : n_rows - 1;
        const long long pt = IN_KIND == 0 ? rc : rc / n_lights;
        float x[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) x[k] = xyz_scale * xyz[pt * 3 + k];
        float dv[4];   // this lane's four dOut values: loaded here, not behind the stores of the tile
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 4 * h + r;
            dv[r] = (valid && f < out_dim) ? dout[row * out_dim + f] : 0.f;
        }
        bf16x8 xin[KSX][1];
        bf16x8 pe[4][1], pl[2][1];
        posenc<10, 1>(x, h, 0, pe);
#pragma unroll
        for (int s = 0; s < 4; ++s) xin[s][0] = pe[s][0];
        if constexpr (IN_KIND == 1) {
            const int l = (int)(rc % n_lights);
            float d[3], xd[3], lp[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                xd[k] = xyz_dir[pt * 3 + k];
                lp[k] = lxyz[l * 3 + k];
            }
            dir_to(lp, xd, d);
            posenc<4, 1>(d, h, 0, pl);
            xin[4][0] = pl[0][0];
            xin[5][0] = pl[1][0];
        }
        // ------------------------------------------------------------------ forward (re-computed), no store in here
        bf16x8 h0[8][1], h1[8][1], h2[8][1], h3[8][1];
        ring::layer<KSX, 0, KSX>(cx, bias_lds, xin, h0);
        ring::layer<KSX, 4, 8>(cx, bias_lds + 128, h0, h1);
        ring::layer<KSX, 8, 8>(cx, bias_lds + 256, h1, h2);
        static_for<0, 4>([&](auto T) {   // layer 3: [h2 ; input], two sub-chunks per tile
            constexpr int t = decltype(T)::value;
            ring::begin<KSX, 12 + 2 * t>(cx);
            f32x16 acc[1];
            bias_init<1>(bias_lds + 384 + 32 * t, h, acc);
            ring::mma<8>(cx, h2, 0, acc[0]);
            ring::end<KSX, 12 + 2 * t>(cx);
            ring::begin<KSX, 13 + 2 * t>(cx);
            ring::mma<KSX>(cx, xin, 0, acc[0]);
            ring::end<KSX, 13 + 2 * t>(cx);
            acc_to_b<true, 1>(acc, h3[2 * t], h3[2 * t + 1]);
        });
        f32x16 logit[1];
        ring::begin<KSX, 20>(cx);
        bias_init<1>(bias_lds + 512, h, logit);
        ring::mma<8>(cx, h3, 0, logit[0]);
        ring::end<KSX, 20>(cx);
        // ------------------------------------------------------------------ stores of the forward half
        store_posenc<10, 4>(fs, 0, h, pe);
        if constexpr (IN_KIND == 0) { if (h == 1) st16(fs, 63, (__bf16)0.f); }   // (IN_KIND == 1: the slot is ldir.x, see the streamed kernel)
        if constexpr (IN_KIND == 1) {
            store_posenc<4, 2>(fs, 63, h, pl);
            if (h == 1) {  // pad features 90..95
#pragma unroll
                for (int e = 90; e < 96; ++e) st16(fs, e, (__bf16)0.f);
            }
        }
        store_hidden<8>(fs, G::kOffH + 0, h, h0);
        store_hidden<8>(fs, G::kOffH + 128, h, h1);
        store_hidden<8>(fs, G::kOffH + 256, h, h2);
        store_hidden<8>(fs, G::kOffH + 384, h, h3);
        // ------------------------------------------------------------------ dZ_out
        bf16x8 dzo[1][1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float g = dv[r] * post_scale * act_grad(logit[0][r], out_act);
            const float gg = (valid && 4 * h + r < out_dim) ? g : 0.f;
            dzo[0][0][r] = (__bf16)gg;
            dzo[0][0][4 + r] = (__bf16)0.f;
            FeatStore f4 = fs;
            f4.roff = fs.roff + (h ? (unsigned)(4 * fs.ld2) : 0u);
            st16(f4, G::kOffDZo + r, (__bf16)gg);
        }
        // ------------------------------------------------------------------ dgrad chain, weights resident
        bf16x8 dz3[8][1], dz2[8][1], dz1[8][1], dz0[8][1];
        static_for<0, 4>([&](auto T) {  // through the out layer: K = 16 padded output slots, one k-step
            constexpr int t = decltype(T)::value;
            const char* f0 = res + (t * 4) * kFragBytes + lane * 16;
            f32x16 acc[1];
            zero_init<1>(acc);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(f0), dzo[0][0], acc[0], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float hv = (float)h3[2 * t + (r >> 3)][0][r & 7];
                dz3[2 * t + (r >> 3)][0][r & 7] = (__bf16)(hv > 0.f ? acc[0][r] : 0.f);
            }
            mfma_operand_fence(dz3[2 * t][0]);
            mfma_operand_fence(dz3[2 * t + 1][0]);
        });
        store_hidden<8>(fs, G::kOffDZ + 384, h, dz3);
        ring::dgrad<8>(res + 16 * kFragBytes, lane, dz3, h2, dz2);   // W3[:128, :]
        store_hidden<8>(fs, G::kOffDZ + 256, h, dz2);
        ring::dgrad<8>(res + 48 * kFragBytes, lane, dz2, h1, dz1);   // W2
        store_hidden<8>(fs, G::kOffDZ + 128, h, dz1);
        ring::dgrad<8>(res + 80 * kFragBytes, lane, dz1, h0, dz0);   // W1
        store_hidden<8>(fs, G::kOffDZ + 0, h, dz0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the sub-chunks fetched ahead for a tile that does not exist
}

}  // namespace bwd
}  // namespace nfx

extern "C" int nfx_option_int(const char* name, int dflt);   // capi.cpp

extern "C" {
int nfx_launch_mlp128_bwd(int in_kind, const float* xyz, const float* xyz_dir, long long n, float xyz_scale,
                          const float* lxyz, int n_lights, const void* blob, int out_dim, int out_act,
                          float post_scale, const float* dout, void* wsp, long long ld, int max_blocks,
                          hipStream_t st) {
    using namespace nfx;
    if (n <= 0) return 0;
    const long long rows = in_kind == 0 ? n : n * n_lights;
    const long long tiles = (rows + bwd::kRows - 1) / bwd::kRows;
    const int grid = (int)(tiles < max_blocks ? tiles : max_blocks);
    const bool use_ring = nfx_option_int("m128_bwd", 1) != 0;   // (per call, like every knob: INTEGRATION.md)
    if (use_ring) {   // r03 default: dgrad weights resident in LDS, forward weights through a DMA ring
        const int rl = bwd::ring::kLds;
        auto launch = [&](auto k) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, rl);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(k, dim3(grid), dim3(bwd::kNW * 64), rl, st, xyz, xyz_dir, n, xyz_scale, lxyz, n_lights,
                               (const char*)blob, out_dim, out_act, post_scale, dout, (__bf16*)wsp, ld);
            return (int)hipGetLastError();
        };
        return in_kind == 0 ? launch(bwd::mlp128_bwd_ring_kernel<0>) : launch(bwd::mlp128_bwd_ring_kernel<1>);
    }
    const int lds = 2 * kSlotBytes + m128::kMainBiasFloats * 4;
    if (in_kind == 0) {
        auto k = bwd::mlp128_bwd_kernel<0>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(bwd::kNW * 64), lds, st, xyz, xyz_dir, n, xyz_scale, lxyz, n_lights,
                           (const char*)blob, out_dim, out_act, post_scale, dout, (__bf16*)wsp, ld);
    } else {
        auto k = bwd::mlp128_bwd_kernel<1>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(bwd::kNW * 64), lds, st, xyz, xyz_dir, n, xyz_scale, lxyz, n_lights,
                           (const char*)blob, out_dim, out_act, post_scale, dout, (__bf16*)wsp, ld);
    }
    return (int)hipGetLastError();
}
int nfx_mlp128_train_feats(int in_kind) { return in_kind == 0 ? nfx::bwd::Geo<4>::kFeats : nfx::bwd::Geo<6>::kFeats; }
int nfx_mlp128_train_blob_bytes(int in_kind) {
    return in_kind == 0 ? nfx::bwd::Geo<4>::kBlobBytes : nfx::bwd::Geo<6>::kBlobBytes;
}
}
