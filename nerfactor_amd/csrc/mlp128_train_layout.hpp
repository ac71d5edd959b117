// mlp128_train_layout.hpp — train-blob geometry of the width-128 surface MLPs (nfx_mlp128_pack_train_weights,
// capi_train.cpp) and the output-activation derivative, shared by mlp128_bwd.hip (backward kernels that store the
// activations for separate weight-gradient GEMMs) and mlp128_bwd_fused.hip (weight gradients accumulated on chip).
#pragma once
#include "geom.hpp"
#include "mlp128_layout.hpp"

namespace nfx {
namespace bwd {

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
    static_assert(kXFeats % 2 == 0, "feature-pair-major storage (feat_store.hpp): every group starts on an even feature");
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

}  // namespace bwd
}  // namespace nfx
