// nerf_layout.hpp — blob layout of the packed NeRF network, shared by capi.cpp (host packer) and
// nerf_mlp*.hip (device).  Network shape: nerfactor/models/nerf.py:53-71 at config/nerf.ini:55-70.
//
// 78 chunks (one chunk = all k-steps of one 32-row output tile; 1 KiB per k-step fragment), each
// padded to a multiple of 8 fragments so 8 waves move it with whole 1-KiB DMA pieces:
//   L0   enc[0]    63(->64)     -> 256    8 chunks x  8 frags ( 4 used)
//   L1-4 enc[1-4]  256          -> 256   32 chunks x 16 frags
//   L5   enc[5]    256+64       -> 256    8 chunks x 24 frags (20 used; skip-concat [y, posenc(x)])
//   L6-7 enc[6-7]  256          -> 256   16 chunks x 16 frags
//   L8   [bottleneck | sigma_out] 256 -> 257(->288)  9 chunks x 16 frags
//   L9   rgb_out[0] 256+32      -> 128    4 chunks x 24 frags (18 used)
//   L10  rgb_out[1] 128         -> 3(->32) 1 chunk x  8 frags
// 78 = 3 x 26: a 3-slot LDS ring maps every chunk to the same slot in every pass.
#pragma once
namespace nfx {
namespace nerf {
constexpr int kNChunks = 78;
constexpr int chunk_frags(int k) {
    return k < 8 ? 8 : k < 40 ? 16 : k < 48 ? 24 : k < 73 ? 16 : k < 77 ? 24 : 8;
}
constexpr int chunk_frag_offset(int k) {
    int off = 0;
    for (int i = 0; i < k; ++i) off += chunk_frags(i);
    return off;
}
constexpr int kFrags = chunk_frag_offset(kNChunks);  // 1272
constexpr int kWeightBytes = kFrags * 1024;
// 4-KiB staging pieces per chunk (register-staged variants)
constexpr int kNL0 = 2, kNLH = 4, kNL5 = 6, kNLR0 = 6, kNLR1 = 2;
constexpr int kBiasL0 = 0;            // 8 x 256
constexpr int kBiasBott = 8 * 256;    // 288 (256 bottleneck + sigma + pad)
constexpr int kBiasRgb0 = kBiasBott + 288;  // 128
constexpr int kBiasRgb1 = kBiasRgb0 + 128;  // 32
constexpr int kBiasFloats = kBiasRgb1 + 32; // 2496
constexpr int kBlobBytes = kWeightBytes + kBiasFloats * 4;
static_assert(kFrags == 1272, "layout");
static_assert(kNChunks % 3 == 0, "ring");
}  // namespace nerf
}  // namespace nfx
