// nerf_layout.hpp — blob layout of the packed NeRF network, shared by pack.cpp (host) and
// nerf_mlp.hip (device).  Network shape: nerfactor/models/nerf.py:53-71 at config/nerf.ini:55-70.
//
// Chunk order (one chunk = all k-steps of one 32-row output tile, 1 KiB per k-step fragment,
// padded to a multiple of 4 fragments so 256 threads move it with whole 16-B pieces):
//   L0  enc[0]   63(->64)  -> 256   8 chunks x  4 frags
//   L1-4 enc[1-4] 256      -> 256  32 chunks x 16 frags
//   L5  enc[5]   256+64    -> 256   8 chunks x 20 frags   (skip-concat: [y, posenc(x)])
//   L6-7 enc[6-7] 256      -> 256  16 chunks x 16 frags
//   L8  [bottleneck | sigma_out] 256 -> 257(->288)  9 chunks x 16 frags
//   L9  rgb_out[0] 256+32  -> 128   4 chunks x 20 frags (18 used)
//   L10 rgb_out[1] 128     -> 3(->32) 1 chunk x 8 frags
#pragma once
namespace nfx {
namespace nerf {
constexpr int kNL0 = 1, kNLH = 4, kNL5 = 5, kNLR0 = 5, kNLR1 = 2;  // 4-KiB pieces per chunk
constexpr int kFrags = 8 * 4 + 32 * 16 + 8 * 20 + 16 * 16 + 9 * 16 + 4 * 20 + 1 * 8;  // 1192
constexpr int kWeightBytes = kFrags * 1024;
constexpr int kBiasL0 = 0;            // 8 x 256
constexpr int kBiasBott = 8 * 256;    // 288 (256 bottleneck + sigma + pad)
constexpr int kBiasRgb0 = kBiasBott + 288;  // 128
constexpr int kBiasRgb1 = kBiasRgb0 + 128;  // 32
constexpr int kBiasFloats = kBiasRgb1 + 32; // 2496
constexpr int kBlobBytes = kWeightBytes + kBiasFloats * 4;
}  // namespace nerf
}  // namespace nfx
