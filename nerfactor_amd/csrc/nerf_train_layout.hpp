// nerf_train_layout.hpp — training blob and feature-major workspace of the NeRF MLP backward (nerf_bwd.hip),
// shared with the host packer / wgrad driver in capi_train.cpp.
//
// Train blob = [forward fragments, exactly nerf_layout.hpp's 78 chunks | dgrad fragments | biases].
// dgrad "layers" are Dense layers whose Keras kernel is W^T, consumed in this order:
//   D1  rgb_out[1]^T            16 (3 used)  -> 128    4 chunks x  4 frags (1 used)
//   D2  rgb_out[0][:256]^T     128           -> 256    8 chunks x  8 frags
//   D3  [bottleneck | sigma]^T 256 + 16 (1)  -> 256    8 chunks x 20 frags (17 used)
//   D4  enc[7]^T .. enc[1]^T   256           -> 256    7 x 8 chunks x 16 frags   (enc[5]: its first 256 rows)
//   D5  (input-gradient mode)  enc[0]^T and enc[5][256:]^T: 256 -> 64 posenc slots, 2 x 2 chunks x 16 frags
#pragma once
#include "nerf_layout.hpp"
namespace nfx {
namespace nerf {
constexpr int kNLD1 = 1, kNLD2 = 2, kNLD3 = 5, kNLDH = 4;       // 4-KiB pieces per dgrad chunk
constexpr int kDgradFrags = 4 * 4 + 8 * 8 + 8 * 20 + 7 * 8 * 16;  // 1136
constexpr int kTrainFrags = kFrags + kDgradFrags;
constexpr int kTrainWeightBytes = kTrainFrags * 1024;
constexpr int kTrainBlobBytes = kTrainWeightBytes + kBiasFloats * 4;
// feature-major workspace rows ([feature][ld] bf16)
constexpr int kOffPe = 0;                    // posenc10(x): 63 (+1 pad)
constexpr int kOffPv = 64;                   // posenc4(view): 27 (+5 pad)
constexpr int kOffA = 96;                    // enc activations a0..a7: 8 x 256
constexpr int kOffBott = kOffA + 8 * 256;    // bottleneck output: 256
constexpr int kOffR0 = kOffBott + 256;       // rgb_out[0] activation: 128
constexpr int kOffDZ = kOffR0 + 128;         // dZ of enc[0..7]: 8 x 256
constexpr int kOffDBott = kOffDZ + 8 * 256;  // dZ of the bottleneck: 256
constexpr int kOffDSig = kOffDBott + 256;    // dZ of sigma_out: 1 (+7 pad)
constexpr int kOffDR0 = kOffDSig + 8;        // dZ of rgb_out[0]: 128
constexpr int kOffDRgb = kOffDR0 + 128;      // dZ of rgb_out[1]: 3 (+5 pad)
constexpr int kTrainFeats = kOffDRgb + 8;    // 4976
// feature-PAIR-major storage (feat_store.hpp): every group starts on an even feature
static_assert((kOffPe | kOffPv | kOffA | kOffBott | kOffR0 | kOffDZ | kOffDBott | kOffDSig | kOffDR0 | kOffDRgb) % 2 == 0,
              "feature offsets must be even");
}  // namespace nerf
}  // namespace nfx
