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
#include "geom.hpp"
#include "mlp128_layout.hpp"
#include "mlp_engine.hpp"
#include "feat_store.hpp"

namespace nfx {
namespace bwd {

constexpr int kNW = 4;  // one wave per SIMD: the re-computed activations + gradients need > 256 registers
constexpr int kRows = kNW * 32;

// Train-blob chunk geometry, KSX = k-steps of the network input (4: posenc10(xyz); 6: + posenc4(ldir)).
template <int KSX>
struct Geo {
    static constexpr int kP0 = KSX <= 4 ? 4 : 8;           // frags per L0 chunk
    static constexpr int kP3 = KSX <= 4 ? 12 : 16;         // frags per L3 chunk (8 + KSX used)
    static constexpr int kNL0 = kP0 / 4, kNLH = 2, kNL3 = kP3 / 4, kNLO = 2, kNLD = 1;
    static constexpr int kFwdFrags = 4 * kP0 + 32 + 32 + 4 * kP3 + 8;
    static constexpr int kBwdFrags = 4 * 4 + 3 * 32;
    static constexpr int kWeightBytes = (kFwdFrags + kBwdFrags) * 1024;
    static constexpr int kBiasFloats = m128::kMainBiasFloats;  // 544
    static constexpr int kBlobBytes = kWeightBytes + kBiasFloats * 4;
    static constexpr int kXFeats = KSX * 16;               // 64 or 96 stored input features
    // feature-major workspace rows
    static constexpr int kOffH = kXFeats;                   // h0..h3: 4 x 128
    static constexpr int kOffDZ = kXFeats + 512;            // dZ0..dZ3: 4 x 128
    static constexpr int kOffDZo = kXFeats + 1024;          // dZ_out: 8
    static constexpr int kFeats = kXFeats + 1032;
};

__device__ __forceinline__ float act_grad(float logit, int act) {
    switch (act) {
        case 1: return logit > 0.f ? 1.f : 0.f;
        case 2: { const float s = sigmoidf(logit); return s * (1.f - s); }
        case 3: return sigmoidf(logit);  // d softplus
        default: return 1.f;
    }
}

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
            if (h == 1) st16(fs, 63, (__bf16)0.f);
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

}  // namespace bwd
}  // namespace nfx

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
