// nerf_mlp_v2.hip — NeRF MLP, variant 2: same register-resident dataflow as nerf_mlp.hip, different
// weight pipeline and wave schedule:
//   * weights go global -> LDS by LDS-DMA (global_load_lds_dwordx4, no staging VGPRs, no ds_write),
//     into a 3-slot ring; completion is tracked with COUNTED s_waitcnt vmcnt(N), never a drain;
//   * the two waves of every SIMD run HALF A TILE out of phase: group A (waves 0-3) starts chunk k
//     at workgroup barrier 2k, group B (waves 4-7) at barrier 2k+1, so one wave's bias / ReLU /
//     bf16-convert epilogue and LDS latency sit under the other wave's MFMAs instead of every wave
//     stalling at the same barrier.
// Protocol (R = 3 slots, chunk j lives in slot j % 3; 78 chunks per pass, 78 % 3 == 0):
//   - after every odd barrier 2k+3 every wave issues its share of chunk k+3 (its slot was last
//     read by group B, which finished chunk k at that barrier);
//   - before every even barrier 2j every wave has waited for its own share of chunk j
//     (one younger chunk may still be in flight: vmcnt(pieces of chunk j+1)).
//   Group A therefore: wait(chunk k) ; BARRIER ; first half ; BARRIER ; issue(chunk k+2) ; second half
//   Group B:                           BARRIER ; issue(chunk k+2) ; first half ; wait(chunk k+1) ; BARRIER ; second half
//   plus one extra barrier for B before its first tile and for A after its last.
#include "mlp_engine.hpp"
#include "nerf_layout.hpp"

namespace nfx {
namespace v2 {

constexpr int kSlot = 24 * 1024;
constexpr int kRing = 3;
constexpr int kBiasBytes = 10240;  // biases FIRST: their ds_read offsets must fit the 16-bit immediate
constexpr int kLds = kBiasBytes + kRing * kSlot;
static_assert(nerf::kBiasFloats * 4 <= kBiasBytes, "bias region");
typedef __attribute__((address_space(3))) char lds_char;

// Ablation mask (diagnostic, NFX_ABLATE): 1 no weight DMA / vmcnt waits, 2 no workgroup barriers, 4 no MFMA,
// 8 no A-fragment ds_reads, 16 no epilogue (ReLU/convert), 32 no positional encoding.  Results are wrong by
// construction for any non-zero mask; only the timing is meaningful.
struct Ctx {
    const char* blob;
    char* ring;         // generic pointer to the ring (for ds_read)
    unsigned ring_lds;  // LDS byte address of the ring (for M0)
    int lane, wave;
    unsigned lane_off;  // lane * 16
    bool grp_b;
};

template <int N, int AB = 0>
__device__ __forceinline__ void wait_vm() {
    if constexpr (AB & 1) return;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int AB = 0>
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (!(AB & 2)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// One 1-KiB LDS-DMA piece: 64 lanes x 16 B from gbase + lane_off (lane_off = lane * 16) to
// lds_dst + lane * 16.  The global base is a wave-uniform SGPR pair (saddr form): no per-chunk
// address VGPRs for the compiler to hoist out of the tile loop and spill.
__device__ __forceinline__ void dma_piece(unsigned lane_off, const char* gbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane_off), "s"(gbase), "s"(lds_dst)
        : "memory");
}

// This wave's share of chunk J: pieces [wave*n, wave*n + n), n = frags(J) / 8.
template <int J, int AB = 0>
__device__ __forceinline__ void dma_chunk(const Ctx& cx) {
    if constexpr (AB & 1) return;
    constexpr int n = nerf::chunk_frags(J) / 8;
    constexpr int goff = nerf::chunk_frag_offset(J) * 1024;
    // Opaque copies: without them LICM hoists the ~160 loop-invariant piece addresses of a pass out
    // of the tile loop and spills them (measured: 349 SGPR + 58 VGPR spills, scratch traffic whose
    // vmcnt(0) waits would drain the DMA pipeline).
    unsigned long long base = reinterpret_cast<unsigned long long>(cx.blob);
    unsigned ring = cx.ring_lds;
    asm volatile("" : "+s"(base), "+s"(ring));
    const int piece0 = cx.wave * n;
    const char* g = reinterpret_cast<const char*>(base) + goff + piece0 * 1024;
    const unsigned l = ring + (J % kRing) * kSlot + piece0 * 1024;
#pragma unroll
    for (int i = 0; i < n; ++i) dma_piece(cx.lane_off, g + i * 1024, l + i * 1024);
}

// acc += sum over k-steps [S0, S1) with the A fragments software-pipelined kDepth deep: measured on
// variant 1 (rocprofv3 PMC, profiles/r01): waves spent 57 % of their cycles in s_waitcnt/barrier and
// the MFMA pipe was 53 % busy because only two ds_read_b128 were in flight per wave — every second
// MFMA waited a full LDS round trip.
constexpr int kDepth = 4;
constexpr bool kPinSchedule = false;  // sched_group_barrier pinning: no gain measured, slow compiles
template <int S0, int S1, int KS1, int PRE, int AB, int KS1A, int KS2A, int CT>
__device__ __forceinline__ void mma_range(const char* lane_frag0, const bf16x8 (&b1)[KS1A][CT],
                                          const bf16x8 (&b2)[KS2A][CT], f32x16 (&acc)[CT]) {
    constexpr int N = S1 - S0;
    bf16x8 a[N];
    if constexpr (AB & 8) {
        static_for<0, N>([&](auto I) { a[decltype(I)::value] = b1[0][0]; });
    }
    static_for<0, (kDepth < N ? kDepth : N)>([&](auto I) {
        constexpr int i = decltype(I)::value;
        if constexpr (!(AB & 8)) a[i] = *reinterpret_cast<const bf16x8*>(lane_frag0 + (S0 + i) * kFragBytes);
    });
    static_for<0, N>([&](auto I) {
        constexpr int i = decltype(I)::value;
        constexpr int s = S0 + i;
        if constexpr (i + kDepth < N && !(AB & 8))
            a[i + kDepth] = *reinterpret_cast<const bf16x8*>(lane_frag0 + (s + kDepth) * kFragBytes);
        if constexpr (AB & 4) {
            asm volatile("" ::"v"(a[i]));
            return;
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            if constexpr (s < KS1)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b1[s][c], acc[c], 0, 0, 0);
            else
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b2[s - KS1][c], acc[c], 0, 0, 0);
        }
    });
    // Pin the schedule (the machine scheduler otherwise re-sinks the reads to 2 in flight):
    // [PRE + kDepth ds_reads] then (1 MFMA, 1 ds_read) ... then the last kDepth MFMAs.
    if constexpr ((AB & (4 | 8)) || !kPinSchedule) return;
    constexpr int D = kDepth < N ? kDepth : N;
    __builtin_amdgcn_sched_group_barrier(0x100, PRE + D, 0);
    static_for<0, N - D>([&](auto) {
        __builtin_amdgcn_sched_group_barrier(0x008, CT, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    });
    __builtin_amdgcn_sched_group_barrier(0x008, D * CT, 0);
}

// One 32-row output tile = chunk K of the pass.
template <int K, int KS1, int KS2, int AB, int KS1A, int KS2A>
__device__ __forceinline__ void tile(const Ctx& cx, const float* bias_tile,
                                     const bf16x8 (&b1)[KS1A][1], const bf16x8 (&b2)[KS2A][1],
                                     f32x16 (&acc)[1]) {
    constexpr int kN = nerf::kNChunks;
    constexpr int N1 = nerf::chunk_frags((K + 1) % kN) / 8;
    constexpr int N2 = nerf::chunk_frags((K + 2) % kN) / 8;
    constexpr int KS = KS1 + KS2, H1 = KS / 2;
    static_assert(KS <= nerf::chunk_frags(K), "chunk too small");
    const char* f0 = cx.ring + (K % kRing) * kSlot + cx.lane * 16;
    if (!cx.grp_b) wait_vm<N1, AB>();      // A: my share of chunk K has landed
    wg_barrier<AB>();                      // even barrier for A (chunk K readable), odd for B
    if (cx.grp_b) dma_chunk<(K + 2) % kN, AB>(cx);
    bias_init<1>(bias_tile, cx.lane >> 5, acc);   // 4 ds_read_b128, scheduled with the first fragments
    mma_range<0, H1, KS1, 4, AB>(f0, b1, b2, acc);
    if (cx.grp_b) wait_vm<N2, AB>();       // B: my share of chunk K+1 has landed
    wg_barrier<AB>();                      // odd barrier for A, even for B
    if (!cx.grp_b) dma_chunk<(K + 2) % kN, AB>(cx);
    mma_range<H1, KS, KS1, 0, AB>(f0, b1, b2, acc);
}

template <int K0, int KS1, int KS2, int NT, bool RELU, int AB, int KS1A, int KS2A, int NTA>
__device__ __forceinline__ void layer(const Ctx& cx, const float* bias, const bf16x8 (&b1)[KS1A][1],
                                      const bf16x8 (&b2)[KS2A][1], bf16x8 (&bout)[NTA][1]) {
    static_assert(2 * NT <= NTA, "output array too small");
    static_for<0, NT>([&](auto T) {
        constexpr int t = decltype(T)::value;
        f32x16 acc[1];
        tile<K0 + t, KS1, KS2, AB>(cx, bias + 32 * t, b1, b2, acc);
        if constexpr (AB & 16) {
            asm volatile("" ::"v"(acc[0]));
            bout[2 * t][0] = b1[0][0];
            bout[2 * t + 1][0] = b1[0][0];
        } else {
            acc_to_b<RELU, 1>(acc, bout[2 * t], bout[2 * t + 1]);
        }
    });
}

template <int AB>
__global__ __launch_bounds__(512, 2) void nerf_mlp_bf16_v2_kernel(
    const float* __restrict__ rayo, const float* __restrict__ rayd, const float* __restrict__ zbuf,
    long long n_pts, int n_samples, const char* __restrict__ blob, float4* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    Ctx cx;
    cx.blob = blob;
    cx.ring = smem + kBiasBytes;
    cx.ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_char*)smem) + kBiasBytes;
    cx.lane = tid & 63;
    cx.lane_off = (tid & 63) * 16;
    cx.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    cx.grp_b = cx.wave >= 4;
    const int h = cx.lane >> 5, p = cx.lane & 31;
    constexpr int kTilePts = 8 * 32;

    float* bias_lds = reinterpret_cast<float*>(smem);
    {
        const float* bsrc = reinterpret_cast<const float*>(blob + nerf::kWeightBytes);
        for (int i = tid; i < nerf::kBiasFloats; i += 512) bias_lds[i] = bsrc[i];
    }
    dma_chunk<0, AB>(cx);
    dma_chunk<1, AB>(cx);
    __syncthreads();  // biases visible to every wave
    if (cx.grp_b) {   // half-a-tile phase offset: B's first tile starts at barrier 1
        wait_vm<nerf::chunk_frags(1) / 8, AB>();
        wg_barrier<AB>();
    }

    const long long n_tiles = (n_pts + kTilePts - 1) / kTilePts;
    for (long long t_ = blockIdx.x; t_ < n_tiles; t_ += gridDim.x) {
        bf16x8 pe[4][1], pv[2][1];
        const long long m = t_ * kTilePts + cx.wave * 32 + p;
        {
            const long long mm = m < n_pts ? m : n_pts - 1;
            const long long ray = mm / n_samples;
            const float zz = zbuf[mm];
            float x[3], d[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                d[k] = rayd[ray * 3 + k];
                x[k] = rayo[ray * 3 + k] + d[k] * zz;  // nerf.py:162-163
            }
            if constexpr (AB & 32) {
#pragma unroll
                for (int q = 0; q < 4; ++q) pe[q][0] = pack8(x[0], x[1], x[2], d[0], d[1], d[2], zz, zz);
                pv[0][0] = pe[0][0];
                pv[1][0] = pe[1][0];
            } else {
                posenc<10, 1>(x, h, 0, pe);
                posenc<4, 1>(d, h, 0, pv);
            }
        }
        bf16x8 ha[16][1], hb[16][1];
        using namespace nerf;
        layer<0, 4, 0, 8, true, AB>(cx, bias_lds + kBiasL0, pe, pe, ha);
        layer<8, 16, 0, 8, true, AB>(cx, bias_lds + kBiasL0 + 256 * 1, ha, pe, hb);
        layer<16, 16, 0, 8, true, AB>(cx, bias_lds + kBiasL0 + 256 * 2, hb, pe, ha);
        layer<24, 16, 0, 8, true, AB>(cx, bias_lds + kBiasL0 + 256 * 3, ha, pe, hb);
        layer<32, 16, 0, 8, true, AB>(cx, bias_lds + kBiasL0 + 256 * 4, hb, pe, ha);
        layer<40, 16, 4, 8, true, AB>(cx, bias_lds + kBiasL0 + 256 * 5, ha, pe, hb);
        layer<48, 16, 0, 8, true, AB>(cx, bias_lds + kBiasL0 + 256 * 6, hb, pe, ha);
        layer<56, 16, 0, 8, true, AB>(cx, bias_lds + kBiasL0 + 256 * 7, ha, pe, hb);
        layer<64, 16, 0, 8, false, AB>(cx, bias_lds + kBiasBott, hb, pe, ha);  // bottleneck
        float sigma;
        {
            f32x16 acc[1];
            tile<72, 16, 0, AB>(cx, bias_lds + kBiasBott + 256, hb, pe, acc);   // sigma_out row
            sigma = acc[0][0];
        }
        bf16x8 r0[8][1];
        layer<73, 16, 2, 4, true, AB>(cx, bias_lds + kBiasRgb0, ha, pv, r0);
        {
            f32x16 acc[1];
            tile<77, 8, 0, AB>(cx, bias_lds + kBiasRgb1, r0, pe, acc);
            if (h == 0 && m < n_pts) out[m] = make_float4(acc[0][0], acc[0][1], acc[0][2], sigma);
        }
    }
    if (!cx.grp_b) wg_barrier<AB>();  // matches B's extra initial barrier
    wait_vm<0>();                 // the two chunks prefetched for a pass that never comes
}

}  // namespace v2
}  // namespace nfx

template <int AB>
static int launch_v2(const float* rayo, const float* rayd, const float* z, long long n_pts, int n_samples,
                     const void* blob, float* out, int max_blocks, hipStream_t stream) {
    using namespace nfx;
    const long long n_tiles = (n_pts + 255) / 256;
    const int grid = (int)(n_tiles < max_blocks ? n_tiles : max_blocks);
    auto kern = v2::nerf_mlp_bf16_v2_kernel<AB>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, v2::kLds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), v2::kLds, stream, rayo, rayd, z, n_pts, n_samples,
                       (const char*)blob, (float4*)out);
    return (int)hipGetLastError();
}

extern "C" int nfx_launch_nerf_mlp_bf16_v2(const float* rayo, const float* rayd, const float* z,
                                           long long n_pts, int n_samples, const void* blob, float* out,
                                           int max_blocks, int ablate, hipStream_t stream) {
    if (n_pts <= 0) return 0;
#define NFX_AB(m) case m: return launch_v2<m>(rayo, rayd, z, n_pts, n_samples, blob, out, max_blocks, stream)
    switch (ablate) {
        NFX_AB(0);
#ifdef NFX_ABLATION_BUILD
        NFX_AB(1); NFX_AB(2); NFX_AB(3); NFX_AB(4); NFX_AB(8); NFX_AB(12); NFX_AB(16); NFX_AB(32); NFX_AB(28); NFX_AB(31);
#endif
        default: return (int)hipErrorInvalidValue;
    }
#undef NFX_AB
}
