// tr16.hpp — row-contracting MFMA operands out of ROW-MAJOR bf16 tiles in LDS, through gfx950's transposing LDS load.
// Used by mlp128_bwd_fused.hip (weight gradients: dW[i, j] = sum over rows of H[row, i] dZ[row, j]) and pinned by
// nfx_selftest_tr16 (selftest.hip; tests/test_gpu_train.py::test_transposing_lds_read_contracts_over_rows).
#pragma once
#include "nfx_common.hpp"

namespace nfx {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
// ds_read_b64_tr_b16: lane i of a 16-lane group supplies the address of 4 consecutive bf16 (row i >> 2, slots
// 4 (i & 3) .. + 3 of a [4 rows][16 slots] block) and receives slot i of the block's four rows.  MFMA 32x32x16 operand
// of lane l (m = l & 31, k-group g = l >> 5, elements e = 0..7 <-> k = 8 g + e): group q = l >> 4 reads the block at
// rows R + 8 g (+ 4 for e >= 4), slots C + 16 (q & 1): `p` is that lane's address for e = 0..3 (tr_lane_off below).
template <int PITCH>
__device__ __forceinline__ bf16x8 tr_frag(const char* p) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * PITCH));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ int tr_lane_off(int lane, int pitch) {
    const int i = lane & 15, q = lane >> 4;
    return (8 * (q >> 1) + (i >> 2)) * pitch + (16 * (q & 1) + 4 * (i & 3)) * 2;
}

}  // namespace nfx
