// nerf_mlp_v3.hip — NeRF MLP, variant 3.  Same register-resident dataflow and blob as variants 1/2;
// the wave schedule follows what the variant-2 ablation (profiles/r01/ablation_variant2.log) showed:
// with A-fragment reads, waits and MFMAs finely interleaved in every wave, MFMA time (68 ms) and
// everything else (48 ms) simply ADD UP (120 ms) — the two waves of a SIMD stall on LDS at the same
// moments and want the matrix pipe at the same moments.  Here every wave works in SEGMENTS:
//     LOAD  = 8 ds_read_b128 (one half tile of A fragments, 32 VGPRs), no MFMA
//     COMP  = 8 back-to-back MFMAs on operands that are already in registers
// and the two waves of a SIMD run them in anti-phase inside every barrier interval:
//     waves 0-3 ("early"):  COMP(half i)  then LOAD(half i+1) [+ epilogue]
//     waves 4-7 ("late"):   LOAD(half i)  then COMP(half i)   [+ epilogue]
// so one wave's pure-MFMA segment covers the other's LDS round trip, and vice versa.
// Weights: LDS-DMA into a 6-slot ring (6 x 24 KiB), counted vmcnt; chunk j is complete before
// barrier 2j-1 and its slot is re-filled after barrier 2j+2 (lead: 9 half-tile intervals).
#include "mlp_engine.hpp"
#include "nerf_layout.hpp"

namespace nfx {
namespace v3 {

constexpr int kSlot = 24 * 1024;
constexpr int kRing = 6;
constexpr int kBiasBytes = 10240;
constexpr int kLds = kBiasBytes + kRing * kSlot;  // 157 696 B of the 163 840 B LDS
constexpr int kN = nerf::kNChunks;
static_assert(kN % kRing == 0, "ring");
typedef __attribute__((address_space(3))) char lds_char;

constexpr int pieces(int k) { return nerf::chunk_frags(((k % kN) + kN) % kN) / 8; }
// DMA pieces of chunks [k+lo, k+hi] still in flight when chunk k is waited for
constexpr int younger(int k, int lo, int hi) {
    int s = 0;
    for (int i = lo; i <= hi; ++i) s += pieces(k + i);
    return s;
}

struct Ctx {
    const char* blob;
    char* ring;
    unsigned ring_lds;
    int lane, wave;
    unsigned lane_off;
    bool late;  // waves 4-7
};

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
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
template <int J>
__device__ __forceinline__ void dma_chunk(const Ctx& cx) {
    constexpr int n = pieces(J);
    constexpr int goff = nerf::chunk_frag_offset(J % kN) * 1024;
    unsigned long long base = reinterpret_cast<unsigned long long>(cx.blob);
    unsigned ring = cx.ring_lds;
    asm volatile("" : "+s"(base), "+s"(ring));  // keep the ~160 piece addresses out of LICM's hands
    const int piece0 = cx.wave * n;
    const char* g = reinterpret_cast<const char*>(base) + goff + piece0 * 1024;
    const unsigned l = ring + ((J % kN) % kRing) * kSlot + piece0 * 1024;
#pragma unroll
    for (int i = 0; i < n; ++i) dma_piece(cx.lane_off, g + i * 1024, l + i * 1024);
}

// k-steps per half tile of a layer with KS k-steps
constexpr int half_lo(int ks, int half) { return half == 0 ? 0 : ks / 2; }
constexpr int half_hi(int ks, int half) { return half == 0 ? ks / 2 : ks; }
constexpr int kABuf = 12;  // largest half: 24/2

struct ABuf {
    bf16x8 a[kABuf];
};

// LOAD segment: A fragments [S0, S1) of chunk K into registers.
template <int K, int S0, int S1>
__device__ __forceinline__ void load_seg(const Ctx& cx, ABuf& ab) {
    const char* f0 = cx.ring + ((K % kN) % kRing) * kSlot + cx.lane * 16;
    static_for<S0, S1>([&](auto S) {
        constexpr int s = decltype(S)::value;
        ab.a[s - S0] = *reinterpret_cast<const bf16x8*>(f0 + s * kFragBytes);
    });
    __builtin_amdgcn_sched_barrier(0);
}
// COMP segment: acc += A[s] x B[s] for s in [S0, S1).
template <int S0, int S1, int KS1, int KS1A, int KS2A>
__device__ __forceinline__ void comp_seg(const ABuf& ab, const bf16x8 (&b1)[KS1A][1],
                                         const bf16x8 (&b2)[KS2A][1], f32x16& acc) {
    static_for<S0, S1>([&](auto S) {
        constexpr int s = decltype(S)::value;
        if constexpr (s < KS1)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab.a[s - S0], b1[s][0], acc, 0, 0, 0);
        else
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab.a[s - S0], b2[s - KS1][0], acc, 0, 0, 0);
    });
    __builtin_amdgcn_sched_barrier(0);
}

// One tile (chunk K, KS = KS1 + KS2 k-steps); KSN = k-steps of the NEXT tile's chunk (for the early
// waves' look-ahead load).  `fin(acc)` consumes the finished accumulator (epilogue).
template <int K, int KS1, int KS2, int KSN, int KS1A, int KS2A, typename Fin>
__device__ __forceinline__ void tile(const Ctx& cx, ABuf& ab, const float* bias_tile,
                                     const bf16x8 (&b1)[KS1A][1], const bf16x8 (&b2)[KS2A][1],
                                     Fin&& fin) {
    constexpr int KS = KS1 + KS2;
    static_assert(KS <= nerf::chunk_frags(K % kN), "chunk too small");
    f32x16 acc[1];
    // ---- interval 2K (first half)
    wg_barrier();
    dma_chunk<K + kRing - 1>(cx);
    bias_init<1>(bias_tile, cx.lane >> 5, acc);
    if (cx.late) load_seg<K, half_lo(KS, 0), half_hi(KS, 0)>(cx, ab);
    comp_seg<half_lo(KS, 0), half_hi(KS, 0), KS1>(ab, b1, b2, acc[0]);
    if (!cx.late) load_seg<K, half_lo(KS, 1), half_hi(KS, 1)>(cx, ab);
    wait_vm<younger(K + 1, 1, kRing - 2)>();  // my share of chunk K+1 has landed
    // ---- interval 2K+1 (second half)
    wg_barrier();
    if (cx.late) load_seg<K, half_lo(KS, 1), half_hi(KS, 1)>(cx, ab);
    comp_seg<half_lo(KS, 1), half_hi(KS, 1), KS1>(ab, b1, b2, acc[0]);
    if (!cx.late) load_seg<K + 1, half_lo(KSN, 0), half_hi(KSN, 0)>(cx, ab);
    fin(acc);
}

template <int K0, int KS1, int KS2, int NT, int KSNEXT, bool RELU, int KS1A, int KS2A, int NTA>
__device__ __forceinline__ void layer(const Ctx& cx, ABuf& ab, const float* bias,
                                      const bf16x8 (&b1)[KS1A][1], const bf16x8 (&b2)[KS2A][1],
                                      bf16x8 (&bout)[NTA][1]) {
    static_assert(2 * NT <= NTA, "output array too small");
    static_for<0, NT>([&](auto T) {
        constexpr int t = decltype(T)::value;
        tile<K0 + t, KS1, KS2, (t == NT - 1 ? KSNEXT : KS1 + KS2)>(
            cx, ab, bias + 32 * t, b1, b2,
            [&](f32x16(&acc)[1]) { acc_to_b<RELU, 1>(acc, bout[2 * t], bout[2 * t + 1]); });
    });
}

__global__ __launch_bounds__(512, 2) void nerf_mlp_bf16_v3_kernel(
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
    cx.late = cx.wave >= 4;
    const int h = cx.lane >> 5, p = cx.lane & 31;
    constexpr int kTilePts = 8 * 32;

    float* bias_lds = reinterpret_cast<float*>(smem);
    {
        const float* bsrc = reinterpret_cast<const float*>(blob + nerf::kWeightBytes);
        for (int i = tid; i < nerf::kBiasFloats; i += 512) bias_lds[i] = bsrc[i];
    }
    static_for<0, kRing - 1>([&](auto J) { dma_chunk<decltype(J)::value>(cx); });
    wait_vm<younger(0, 1, kRing - 2)>();  // chunk 0
    __syncthreads();                      // biases + chunk 0 visible to every wave
    ABuf ab;
    if (!cx.late) load_seg<0, half_lo(4, 0), half_hi(4, 0)>(cx, ab);  // early waves enter every interval loaded

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
            posenc<10, 1>(x, h, 0, pe);
            posenc<4, 1>(d, h, 0, pv);
        }
        bf16x8 ha[16][1], hb[16][1];
        using namespace nerf;
        layer<0, 4, 0, 8, 16, true>(cx, ab, bias_lds + kBiasL0, pe, pe, ha);
        layer<8, 16, 0, 8, 16, true>(cx, ab, bias_lds + kBiasL0 + 256 * 1, ha, pe, hb);
        layer<16, 16, 0, 8, 16, true>(cx, ab, bias_lds + kBiasL0 + 256 * 2, hb, pe, ha);
        layer<24, 16, 0, 8, 16, true>(cx, ab, bias_lds + kBiasL0 + 256 * 3, ha, pe, hb);
        layer<32, 16, 0, 8, 20, true>(cx, ab, bias_lds + kBiasL0 + 256 * 4, hb, pe, ha);
        layer<40, 16, 4, 8, 16, true>(cx, ab, bias_lds + kBiasL0 + 256 * 5, ha, pe, hb);
        layer<48, 16, 0, 8, 16, true>(cx, ab, bias_lds + kBiasL0 + 256 * 6, hb, pe, ha);
        layer<56, 16, 0, 8, 16, true>(cx, ab, bias_lds + kBiasL0 + 256 * 7, ha, pe, hb);
        layer<64, 16, 0, 8, 16, false>(cx, ab, bias_lds + kBiasBott, hb, pe, ha);  // bottleneck
        float sigma;
        tile<72, 16, 0, 18>(cx, ab, bias_lds + kBiasBott + 256, hb, pe,
                            [&](f32x16(&acc)[1]) { sigma = acc[0][0]; });           // sigma_out row
        bf16x8 r0[8][1];
        layer<73, 16, 2, 4, 8, true>(cx, ab, bias_lds + kBiasRgb0, ha, pv, r0);
        tile<77, 8, 0, 4>(cx, ab, bias_lds + kBiasRgb1, r0, pe, [&](f32x16(&acc)[1]) {
            if (h == 0 && m < n_pts) out[m] = make_float4(acc[0][0], acc[0][1], acc[0][2], sigma);
        });
    }
    wait_vm<0>();
}

}  // namespace v3
}  // namespace nfx

extern "C" int nfx_launch_nerf_mlp_bf16_v3(const float* rayo, const float* rayd, const float* z,
                                           long long n_pts, int n_samples, const void* blob, float* out,
                                           int max_blocks, hipStream_t stream) {
    using namespace nfx;
    if (n_pts <= 0) return 0;
    const long long n_tiles = (n_pts + 255) / 256;
    const int grid = (int)(n_tiles < max_blocks ? n_tiles : max_blocks);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(v3::nerf_mlp_bf16_v3_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, v3::kLds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(v3::nerf_mlp_bf16_v3_kernel, dim3(grid), dim3(512), v3::kLds, stream, rayo, rayd, z,
                       n_pts, n_samples, (const char*)blob, (float4*)out);
    return (int)hipGetLastError();
}
