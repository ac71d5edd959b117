// pack.hpp — host-side weight packer (no GPU needed): Keras Dense kernels [in, out] fp32 ->
// MFMA A-operand fragments in the order the fused-MLP kernels consume them (see mlp_engine.hpp).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <vector>

namespace nfx {
namespace pack {

// One input segment of a layer = a run of B-operand k-steps.
//   kHidden: `n` features produced by a previous layer (n % 32 == 0), k-steps follow the C/D
//            lane map: slot (s,h,j) <-> feature 32(s>>1) + 16(s&1) + (j&3) + 8(j>>2) + 4h.
//   kPosEnc: the 6L+3 features of an L-band positional encoding, slots as in mlp_engine.hpp
//            (half 0 = sines + x0,x1; half 1 = cosines + x2).
//   kRaw:    `n` (<= 8 per half... see pack.cpp) raw features placed by an explicit slot table.
enum SegKind { kHidden = 0, kPosEnc = 1, kRaw = 2 };
struct Seg {
    SegKind kind;
    int n;     // kHidden: feature count; kPosEnc: L; kRaw: number of k-steps
    int row0;  // first row of this segment in the Keras kernel
    const int* raw_slots;  // kRaw: [n][2][8] row offsets relative to row0, -1 = zero
};
// Output columns = concatenation of column blocks taken from (possibly different) kernels that
// share the same input rows (used to fuse [bottleneck | sigma_out]).
struct Src {
    const float* kernel;  // [rows, cols] row-major
    const float* bias;    // [cols]
    int cols;
};

int seg_ksteps(const Seg& s);
// Source row for B slot (k-step s within the segment, lane half h, element j), or -1.
int seg_row(const Seg& seg, int s, int h, int j);

// Packs one layer: n_tiles chunks of `chunk_frags` 1-KiB fragments (zero padded), then returns
// the number of bytes written to wdst.  bias_dst gets n_tiles*32 floats.
size_t pack_layer_bf16(const std::vector<Seg>& segs, const std::vector<Src>& srcs, int n_tiles,
                       int chunk_frags, uint8_t* wdst, float* bias_dst);

uint16_t f32_to_bf16_rne(float f);

}  // namespace pack
}  // namespace nfx
