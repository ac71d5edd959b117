// lds_dma.hpp — global -> LDS without a register stop (gfx950 `global_load_lds_dwordx4`), shared by the weight rings of
// nerf_mlp_v6.hip, mlp128_bwd.hip and nerf_bwd.hip.
//
// One wave moves N consecutive 1-KiB pieces (64 lanes x 16 B each): lane offset in a VGPR, the wave-uniform global base
// in an SGPR pair, the LDS destination in M0.  The instruction's immediate offset is added to BOTH addresses (checked on
// hardware in round 3: a statement with `offset:1024` is bit-identical to a second statement with both bases advanced),
// so N pieces are N instructions behind ONE M0 save / set / restore — 3 scalar instructions per chunk instead of 3 N.
// The pieces count in `vmcnt` like any load (and, on gfx9, in the same in-order counter as stores): the caller waits
// with `s_waitcnt vmcnt(k)` for exactly what may stay in flight.
#pragma once

namespace nfx {

template <int N>
__device__ __forceinline__ void lds_dma_pieces(unsigned lane_off, const char* gbase, unsigned lds_dst) {
    unsigned keep;
    static_assert(N >= 1 && N <= 5, "pieces per wave and statement");
    // (N = 5: the 13-bit signed offset ends at 4095, so the statement is centred on the third piece)
    if constexpr (N == 1)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\t"
                     "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(lane_off), "s"(gbase), "s"(lds_dst) : "memory");
    else if constexpr (N == 2)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(lane_off), "s"(gbase), "s"(lds_dst) : "memory");
    else if constexpr (N == 3)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                     "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(lane_off), "s"(gbase), "s"(lds_dst) : "memory");
    else if constexpr (N == 4)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
                     "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(lane_off), "s"(gbase), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:-2048\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:-1024\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                     "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(lane_off), "s"(gbase + 2048), "s"(lds_dst + 2048) : "memory");
}

}  // namespace nfx
