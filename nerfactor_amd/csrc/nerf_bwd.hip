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
#include "feat_store.hpp"
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
    long long ld) {
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

}  // namespace bwd
}  // namespace nfx

extern "C" int nfx_launch_nerf_bwd(const float* rayo, const float* rayd, const float* z, long long n_pts, int n_samples,
                                   const void* blob, const float* d_rgbs, void* wsp, long long ld, int max_blocks,
                                   hipStream_t st) {
    using namespace nfx;
    if (n_pts <= 0) return 0;
    const long long tiles = (n_pts + bwd::kNerfRows - 1) / bwd::kNerfRows;
    const int grid = (int)(tiles < max_blocks ? tiles : max_blocks);
    auto k = bwd::nerf_bwd_kernel;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       bwd::kNerfBwdLds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(bwd::kNerfNW * 64), bwd::kNerfBwdLds, st, rayo, rayd, z, n_pts, n_samples,
                       (const char*)blob, (const float4*)d_rgbs, (__bf16*)wsp, ld);
    return (int)hipGetLastError();
}
