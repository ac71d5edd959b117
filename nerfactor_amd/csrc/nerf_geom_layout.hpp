// nerf_geom_layout.hpp — blob of the density-gradient kernel (nerf_geom.hip), shared with its packer (capi_geom.cpp).
//   forward   enc[0..7] (nerf_layout.hpp chunks 0..63) + the sigma_out tile (chunk 72 of the inference blob)
//   backward  enc[7]^T, enc[6]^T, G5 = enc[5][256:]^T -> posenc slots, enc[5][:256]^T, enc[4]^T .. enc[1]^T,
//             G0 = enc[0]^T -> posenc slots          (all chunks 16 fragments: K = 256)
//   floats    enc biases (8 x 256) | sigma bias tile (32) | bf16-rounded sigma_out kernel (256)
#pragma once
#include "nerf_layout.hpp"
namespace nfx {
namespace nerf {
constexpr int kGeoFwdFrags = chunk_frag_offset(64) + 16;
constexpr int kGeoBwdFrags = (7 * 8 + 2 * 2) * 16;
constexpr int kGeoFrags = kGeoFwdFrags + kGeoBwdFrags;
constexpr int kGeoWeightBytes = kGeoFrags * 1024;
constexpr int kGeoBiasSig = 8 * 256;         // 32 floats (row 0 = sigma_out bias)
constexpr int kGeoWSig = kGeoBiasSig + 32;   // 256 floats
constexpr int kGeoFloats = kGeoWSig + 256;
constexpr int kGeoBlobBytes = kGeoWeightBytes + kGeoFloats * 4;
}  // namespace nerf
}  // namespace nfx
