// mlp128_layout.hpp — blob layouts of the width-128 surface MLPs, shared by capi.cpp (host packer)
// and mlp128.hip (device).  Network shape: mlp.Network([128]*4, relu, skip_at=[2]) + out layer
// (nerfactor/models/shape.py:79-94, nerfactor/models/nerfactor.py:128-143, models/brdf.py:57-66).
//
// Every "main" stream has the same chunk geometry (fragments of 1 KiB, chunks padded to x4):
//   L0   in      -> 128   4 chunks x  4 frags   (xyz: 4 used; ldir / z+rusink: 2 used)
//   L1   128     -> 128   4 chunks x  8 frags
//   L2   128     -> 128   4 chunks x  8 frags
//   L3   128+in  -> 128   4 chunks x 12 frags   (8 + 4 used for xyz, 8 + 2 otherwise)
//   out  128     -> <=8   1 chunk  x  8 frags
// Light visibility (NFX_IN_XYZ_LDIR) additionally has a "pre" stream: the posenc(xyz) rows of L0 and
// L3 (8 chunks x 4 frags, biases b0 | b3 folded in) evaluated once per surface point, so the
// 512 lights of a point never recompute them.
#pragma once
namespace nfx {
namespace m128 {
constexpr int kNL0 = 1, kNLH = 2, kNL3 = 3, kNLOut = 2;
constexpr int kMainFrags = 4 * 4 + 4 * 8 + 4 * 8 + 4 * 12 + 8;  // 136
constexpr int kMainWeightBytes = kMainFrags * 1024;
constexpr int kMainBiasFloats = 4 * 128 + 32;                   // 544
constexpr int kMainBytes = kMainWeightBytes + kMainBiasFloats * 4;
constexpr int kPreFrags = 8 * 4;
constexpr int kPreWeightBytes = kPreFrags * 1024;
constexpr int kPreBiasFloats = 256;
constexpr int kPreBytes = kPreWeightBytes + kPreBiasFloats * 4;
// blob = [main] for NFX_IN_XYZ / NFX_IN_Z_RUSINK, [pre][main] for NFX_IN_XYZ_LDIR
constexpr int kMaxZDim = 9;  // z0 rides in k-step 0, z1.. in k-step 1 (see brdf_input_slots)
}  // namespace m128
}  // namespace nfx
