// nerf_mlp_v5.hip — variant 5 of the fused NeRF MLP: ONE wave per SIMD, 64 points per wave (two column tiles share
// every A fragment: half the LDS read traffic of the 32-point variants, which at 1 KiB per MFMA need more LDS
// bandwidth than a CU has once the MFMA pipe is > ~80 % busy), and a SOFTWARE-PIPELINED epilogue: the
// bias + ReLU + bf16-convert of output tile i-1 is issued in the shadow of the MFMAs of tile i (independent
// accumulators, same basic block between two barriers), so a single wave keeps the matrix pipe busy without a
// second wave to hide behind.  One wave per SIMD hides at most ~5 single-issue instructions per MFMA
// (MI355X_MICROARCH.md), so the epilogue is kept to the minimum: accumulators start from the bias (two broadcast
// ds_read_b128 groups straight into the accumulator registers), ReLU is ONE v_pk_max_i16 per converted bf16 PAIR
// (a negative bf16 is a negative int16), i.e. per output element 1 accvgpr_read + 1/2 cvt_pk + 1/2 pk_max.
// Same blob and bit-identical results to variants 0-4.
#include "mlp_engine.hpp"
#include "nerf_layout.hpp"

#ifdef NFX_V5_TIMING
__device__ unsigned long long nfx_v5_times[128];   // cycle stamp before each tile of one wave (diagnostic build)
__device__ int nfx_v5_idx = -1;
#endif

namespace nfx {
namespace v5 {

constexpr int kLds = 2 * kSlotBytes + nerf::kBiasFloats * 4;

// Timing-only ablation mask (diagnostic builds, NFX_ABLATE): 1 no weight staging (global loads + LDS writes),
// 2 no workgroup barrier, 4 no MFMA, 8 no A-fragment ds_reads, 16 no epilogue, 64 no bias init.  Results are garbage.
template <int AB, int NL_NEXT, int NW, typename F>
__device__ __forceinline__ void with_chunk_ab(WStream& ws, int tid, F&& compute) {
    if constexpr (AB == 0) {
        with_chunk<NL_NEXT, NW>(ws, tid, compute);
    } else {
        Stage<NL_NEXT, NW> st;
        if constexpr (!(AB & 1)) st.load(ws.gnext, tid);
        compute(ws.ring + ws.cur * kSlotBytes);
        if constexpr (!(AB & 1)) st.store(reinterpret_cast<u32x4*>(ws.ring + (ws.cur ^ 1) * kSlotBytes), tid);
        ws.gnext += NL_NEXT * kPieceThreads;
        if (ws.gnext == ws.gend) ws.gnext = ws.gbase;
        ws.cur ^= 1;
        if constexpr (!(AB & 2)) __syncthreads();
    }
}

template <int CT>
struct Acc {
    f32x16 v[CT];
};

// bf16(v0), bf16(v1) -> elements j, j+1 of dst, ReLU as a packed signed-16-bit max with 0
template <bool RELU>
__device__ __forceinline__ void cvt_pair(float v0, float v1, bf16x8& dst, int j) {
    typedef short s2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 vv = {v0, v1};
    b2 pr = __builtin_convertvector(vv, b2);   // one v_cvt_pk_bf16_f32
    if (RELU) {
        s2 w = __builtin_bit_cast(s2, pr);
        const s2 z = {0, 0};
        w = __builtin_elementwise_max(w, z);
        pr = __builtin_bit_cast(b2, w);
    }
    dst[j] = pr[0];
    dst[j + 1] = pr[1];
}

// Epilogue "to the next layer's B operand", executable in pieces: run<R0, R1>() handles accumulator registers
// [R0, R1) (even bounds).  The bias is already in the accumulators.
template <bool RELU, int CT>
struct EpiB {
    const Acc<CT>& acc;
    bf16x8 (&lo)[CT];
    bf16x8 (&hi)[CT];
    template <int R0, int R1>
    __device__ __forceinline__ void run() {
        static_assert((R0 & 1) == 0 && (R1 & 1) == 0, "pairs");
#pragma unroll
        for (int r = R0; r < R1; r += 2)
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                if (r < 8) cvt_pair<RELU>(acc.v[c][r], acc.v[c][r + 1], lo[c], r);
                else cvt_pair<RELU>(acc.v[c][r], acc.v[c][r + 1], hi[c], r - 8);
            }
    }
};
struct EpiNone {
    template <int R0, int R1>
    __device__ __forceinline__ void run() {}
};
// sigma_out tile: only row 0 (register 0 of the h = 0 lanes) is a real output
template <int CT>
struct EpiSigma {
    const Acc<CT>& acc;
    float (&sigma)[CT];
    template <int R0, int R1>
    __device__ __forceinline__ void run() {
        if constexpr (R0 == 0) {
#pragma unroll
            for (int c = 0; c < CT; ++c) sigma[c] = acc.v[c][0];
        }
    }
};

// One output tile: acc = W_tile^T [b1 ; b2] (accumulation from an inline zero), with the previous tile's epilogue
// `prev` spread over the first PIECES k-steps.  Consumes one chunk.
template <int KS1, int KS2, int NL_NEXT, int NW, int AB, int KS1A, int KS2A, int CT, typename Epi>
__device__ __forceinline__ void tile_pipe(WStream& ws, int tid, const float* bias_tile, const bf16x8 (&b1)[KS1A][CT],
                                          const bf16x8 (&b2)[KS2A][CT], Acc<CT>& acc, Epi&& prev) {
    constexpr int KS = KS1 + KS2;
    // the pending outputs may be this tile's LAST input k-steps (layer boundary): finish them well before those
    // k-steps are multiplied -> 8 pieces for 16+ k-steps, 4 pieces otherwise
    constexpr int PIECES = KS >= 16 ? 8 : 4;
    const int lane = tid & 63;
#ifdef NFX_V5_TIMING
    if (blockIdx.x == 7 && tid == 0 && nfx_v5_idx >= 0 && nfx_v5_idx < 128) nfx_v5_times[nfx_v5_idx++] = __builtin_readcyclecounter();
#endif
    with_chunk_ab<AB, NL_NEXT, NW>(ws, tid, [&](const char* chunk) {
        const char* f0 = chunk + lane * 16;
        // accumulators start from the bias.  NFX_V5_BIAS_COPY: one broadcast read group + register copies for the
        // second column tile instead of a second read group (a broadcast ds_read_b128 costs a full LDS pass)
#pragma unroll
        for (int c = 0; c < ((AB & 64) ? 0 : CT); ++c) {
            int hoff = 4 * (lane >> 5);
#ifndef NFX_V5_BIAS_COPY
            asm volatile("" : "+v"(hoff));  // (an integer: laundering the pointer itself would lose its LDS address space)
#endif
            const float* bt = bias_tile + hoff;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(bt + 8 * g);
                acc.v[c][4 * g + 0] = v[0];
                acc.v[c][4 * g + 1] = v[1];
                acc.v[c][4 * g + 2] = v[2];
                acc.v[c][4 * g + 3] = v[3];
            }
        }
        // A fragments kADepth k-steps ahead of their MFMAs (one wave per SIMD: nobody else hides the LDS latency; a
        // k-step is only 64 MFMA cycles here, the LDS round trip under load several times that)
#ifndef NFX_V5_ADEPTH
#define NFX_V5_ADEPTH 2
#endif
        constexpr int kADepth = NFX_V5_ADEPTH, kABuf = kADepth + 1;
        bf16x8 abuf[kABuf];
        if constexpr (AB & 8) {
#pragma unroll
            for (int i = 0; i < kABuf; ++i) abuf[i] = b1[0][0];
        } else {
            static_for<0, kADepth>([&](auto I) {
                constexpr int i = decltype(I)::value;
                if constexpr (i < KS) abuf[i] = *reinterpret_cast<const bf16x8*>(f0 + i * kFragBytes);
            });
        }
        static_for<0, KS>([&](auto S) {
            constexpr int s = decltype(S)::value;
            if constexpr (s + kADepth < KS && !(AB & 8))
                abuf[(s + kADepth) % kABuf] = *reinterpret_cast<const bf16x8*>(f0 + (s + kADepth) * kFragBytes);
            const bf16x8 a = abuf[s % kABuf];
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const bf16x8 b = s < KS1 ? b1[s < KS1 ? s : 0][c] : b2[s >= KS1 ? s - KS1 : 0][c];
                if constexpr (AB & 4) {  // keep the operands alive without the matrix instruction
                    asm volatile("" ::"v"(a), "v"(b));
                } else {
                    acc.v[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc.v[c], 0, 0, 0);
                }
            }
            if constexpr (s < PIECES && !(AB & 16)) prev.template run<16 * s / PIECES, 16 * (s + 1) / PIECES>();
        });
    });
}

// A Dense layer of NT tiles whose outputs feed the next layer (bout); `prev0` = the pending epilogue of the tile
// before this layer's first one.  On return the LAST tile's epilogue is still pending: its accumulators are in
// accs[(BASE + NT - 1) & 1] and the caller passes make_epi(...) for it to whatever tile comes next.
template <int KS1, int KS2, int NT, int NL_SELF, int NL_NEXT, bool RELU, int BASE, int NW, int AB, int KS1A, int KS2A,
          int NTA, int CT, typename Epi0>
__device__ __forceinline__ void layer_pipe(WStream& ws, int tid, const float* bias, const bf16x8 (&b1)[KS1A][CT],
                                           const bf16x8 (&b2)[KS2A][CT], bf16x8 (&bout)[NTA][CT], Acc<CT> (&accs)[2],
                                           Epi0&& prev0) {
    static_for<0, NT>([&](auto T) {
        constexpr int t = decltype(T)::value;
        constexpr int NLN = t == NT - 1 ? NL_NEXT : NL_SELF;
        if constexpr (t == 0) {
            tile_pipe<KS1, KS2, NLN, NW, AB>(ws, tid, bias, b1, b2, accs[BASE & 1], prev0);
        } else {
            EpiB<RELU, CT> e{accs[(BASE + t - 1) & 1], bout[2 * (t - 1)], bout[2 * (t - 1) + 1]};
            tile_pipe<KS1, KS2, NLN, NW, AB>(ws, tid, bias + 32 * t, b1, b2, accs[(BASE + t) & 1], e);
        }
    });
}
template <int CT, int NW, int AB>
__global__ __launch_bounds__(NW * 64, 1) void nerf_mlp_bf16_v5_kernel(
    const float* __restrict__ rayo, const float* __restrict__ rayd, const float* __restrict__ zbuf, long long n_pts,
    int n_samples, const char* __restrict__ blob, float4* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using namespace nerf;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
    constexpr int kTilePts = NW * 32 * CT;
    float* bias_lds = reinterpret_cast<float*>(smem + 2 * kSlotBytes);
    {
        const float* bsrc = reinterpret_cast<const float*>(blob + kWeightBytes);
        for (int i = tid; i < kBiasFloats; i += NW * 64) bias_lds[i] = bsrc[i];
    }
    WStream ws;
    ws.gbase = reinterpret_cast<const u32x4*>(blob);
    ws.gend = reinterpret_cast<const u32x4*>(blob + kWeightBytes);
    ws.gnext = ws.gbase;
    ws.ring = smem;
    stream_prologue<kNL0, NW>(ws, tid);
    const long long n_tiles = (n_pts + kTilePts - 1) / kTilePts;
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        bf16x8 pe[4][CT], pv[2][CT];
        long long m[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            m[c] = tile * kTilePts + wave * (32 * CT) + c * 32 + p;
            const long long mm = m[c] < n_pts ? m[c] : n_pts - 1;
            const long long ray = mm / n_samples;
            const float zz = zbuf[mm];
            float x[3], d[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                d[k] = rayd[ray * 3 + k];
                x[k] = rayo[ray * 3 + k] + d[k] * zz;
            }
            posenc<10, CT>(x, h, c, pe);
            posenc<4, CT>(d, h, c, pv);
        }
#ifdef NFX_V5_TIMING
        if (blockIdx.x == 7 && tid == 0) nfx_v5_idx = (tile == blockIdx.x + 4 * (long long)gridDim.x) ? 0 : -1;
#endif
        bf16x8 ha[16][CT], hb[16][CT], r0[8][CT];
        Acc<CT> accs[2];
        float sigma[CT];
        // Tiles are numbered consecutively through the network so that tile i accumulates in accs[i & 1] while the
        // epilogue of tile i-1 reads accs[(i-1) & 1]: every layer has an even number of tiles except the sigma tile
        // and rgb_out[1], handled explicitly below.
        auto pend = [&](auto relu_tag, const Acc<CT>& a, bf16x8(&lo)[CT], bf16x8(&hi)[CT]) {
            return EpiB<decltype(relu_tag)::value, CT>{a, lo, hi};
        };
        using T = std::true_type;
        using F = std::false_type;
        const float* bl = bias_lds + kBiasL0;
        layer_pipe<4, 0, 8, kNL0, kNLH, true, 0, NW, AB>(ws, tid, bl, pe, pe, ha, accs, EpiNone{});
        layer_pipe<16, 0, 8, kNLH, kNLH, true, 0, NW, AB>(ws, tid, bl + 256 * 1, ha, pe, hb, accs, pend(T{}, accs[1], ha[14], ha[15]));
        layer_pipe<16, 0, 8, kNLH, kNLH, true, 0, NW, AB>(ws, tid, bl + 256 * 2, hb, pe, ha, accs, pend(T{}, accs[1], hb[14], hb[15]));
        layer_pipe<16, 0, 8, kNLH, kNLH, true, 0, NW, AB>(ws, tid, bl + 256 * 3, ha, pe, hb, accs, pend(T{}, accs[1], ha[14], ha[15]));
        layer_pipe<16, 0, 8, kNLH, kNL5, true, 0, NW, AB>(ws, tid, bl + 256 * 4, hb, pe, ha, accs, pend(T{}, accs[1], hb[14], hb[15]));
        layer_pipe<16, 4, 8, kNL5, kNLH, true, 0, NW, AB>(ws, tid, bl + 256 * 5, ha, pe, hb, accs, pend(T{}, accs[1], ha[14], ha[15]));
        layer_pipe<16, 0, 8, kNLH, kNLH, true, 0, NW, AB>(ws, tid, bl + 256 * 6, hb, pe, ha, accs, pend(T{}, accs[1], hb[14], hb[15]));
        layer_pipe<16, 0, 8, kNLH, kNLH, true, 0, NW, AB>(ws, tid, bl + 256 * 7, ha, pe, hb, accs, pend(T{}, accs[1], ha[14], ha[15]));
        // bottleneck (no activation): hb -> ha
        layer_pipe<16, 0, 8, kNLH, kNLH, false, 0, NW, AB>(ws, tid, bias_lds + kBiasBott, hb, pe, ha, accs,
                                                       pend(T{}, accs[1], hb[14], hb[15]));
        // sigma tile (tile index even -> accs[0]); pending: last bottleneck tile (accs[1])
        tile_pipe<16, 0, kNLR0, NW, AB>(ws, tid, bias_lds + kBiasBott + 256, hb, pe, accs[0], pend(F{}, accs[1], ha[14], ha[15]));
        // rgb_out[0]: 4 tiles starting at an odd index (BASE = 1); pending: the sigma tile (accs[0])
        {
            EpiSigma<CT> es{accs[0], sigma};
            layer_pipe<16, 2, 4, kNLR0, kNLR1, true, 1, NW, AB>(ws, tid, bias_lds + kBiasRgb0, ha, pv, r0, accs, es);
        }
        // rgb_out[1] (tile index 1 + 4 = odd -> accs[1]); pending: last rgb_out[0] tile (accs[(1 + 3) & 1] = accs[0])
        tile_pipe<8, 0, kNL0, NW, AB>(ws, tid, bias_lds + kBiasRgb1, r0, pe, accs[1], pend(T{}, accs[0], r0[6], r0[7]));
        if (h == 0) {
#pragma unroll
            for (int c = 0; c < CT; ++c)
                if (m[c] < n_pts) out[m[c]] = make_float4(accs[1].v[c][0], accs[1].v[c][1], accs[1].v[c][2], sigma[c]);
        }
    }
}

}  // namespace v5
}  // namespace nfx

template <int CT, int NW, int AB>
static int launch_v5(const float* rayo, const float* rayd, const float* z, long long n_pts, int n_samples,
                     const void* blob, float* out, int max_blocks, hipStream_t stream) {
    using namespace nfx;
    const int tile_pts = NW * 32 * CT;
    const long long n_tiles = (n_pts + tile_pts - 1) / tile_pts;
    const int grid = (int)(n_tiles < max_blocks ? n_tiles : max_blocks);
    auto kern = v5::nerf_mlp_bf16_v5_kernel<CT, NW, AB>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       v5::kLds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), v5::kLds, stream, rayo, rayd, z, n_pts, n_samples,
                       (const char*)blob, (float4*)out);
    return (int)hipGetLastError();
}

extern "C" int nfx_launch_nerf_mlp_bf16_v5(const float* rayo, const float* rayd, const float* z, long long n_pts,
                                           int n_samples, const void* blob, float* out, int max_blocks, int ablate,
                                           hipStream_t stream) {
    if (n_pts <= 0) return 0;
#ifdef NFX_ABLATION_BUILD
    switch (ablate) {
#define NFX_V5_CASE(m) case m: return launch_v5<2, 4, m>(rayo, rayd, z, n_pts, n_samples, blob, out, max_blocks, stream);
        NFX_V5_CASE(1) NFX_V5_CASE(2) NFX_V5_CASE(3) NFX_V5_CASE(4) NFX_V5_CASE(8) NFX_V5_CASE(16) NFX_V5_CASE(64)
        NFX_V5_CASE(80) NFX_V5_CASE(83) NFX_V5_CASE(7) NFX_V5_CASE(12)
#undef NFX_V5_CASE
        default: break;
    }
#endif
    (void)ablate;
    return launch_v5<2, 4, 0>(rayo, rayd, z, n_pts, n_samples, blob, out, max_blocks, stream);
}

#ifdef NFX_V5_TIMING
extern "C" int nfx_debug_v5_times(unsigned long long* host128) {
    return (int)hipMemcpyFromSymbol(host128, HIP_SYMBOL(nfx_v5_times), sizeof(unsigned long long) * 128);
}
#endif
