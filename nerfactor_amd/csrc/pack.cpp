// pack.cpp — see pack.hpp.  Pure host C++ (compiled into libnfx.so by hipcc, runs without a GPU).
#include "pack.hpp"

#include <string.h>

namespace nfx {
namespace pack {

uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

int seg_ksteps(const Seg& s) {
    switch (s.kind) {
        case kHidden: return s.n / 16;
        case kPosEnc: return (3 * s.n + 2 + 7) / 8;
        default: return s.n;
    }
}

int seg_row(const Seg& seg, int s, int h, int j) {
    if (seg.kind == kHidden) {
        return seg.row0 + 32 * (s >> 1) + 16 * (s & 1) + (j & 3) + 8 * (j >> 2) + 4 * h;
    }
    if (seg.kind == kPosEnc) {
        const int L = seg.n, q = 8 * s + j;
        int e;
        if (q < 3 * L) e = 3 + 6 * (q / 3) + (q % 3) + (h ? 3 : 0);  // embedder.py:38-47 order
        else if (q == 3 * L) e = h ? 2 : 0;
        else if (q == 3 * L + 1) e = h ? -1 : 1;
        else e = -1;
        return e < 0 ? -1 : seg.row0 + e;
    }
    const int r = seg.raw_slots[(s * 2 + h) * 8 + j];
    return r < 0 ? -1 : seg.row0 + r;
}

size_t pack_layer_bf16(const std::vector<Seg>& segs, const std::vector<Src>& srcs, int n_tiles,
                       int chunk_frags, uint8_t* wdst, float* bias_dst) {
    int n_out = 0;
    for (const Src& s : srcs) n_out += s.cols;
    uint16_t* w = reinterpret_cast<uint16_t*>(wdst);
    const size_t chunk_elems = (size_t)chunk_frags * 512;  // 1 KiB = 512 bf16
    memset(wdst, 0, (size_t)n_tiles * chunk_elems * 2);
    for (int t = 0; t < n_tiles; ++t) {
        uint16_t* chunk = w + (size_t)t * chunk_elems;
        int frag = 0;
        for (const Seg& seg : segs) {
            const int ks = seg_ksteps(seg);
            for (int s = 0; s < ks; ++s, ++frag) {
                for (int lane = 0; lane < 64; ++lane) {
                    const int h = lane >> 5, n = lane & 31;
                    const int col = 32 * t + n;
                    if (col >= n_out) continue;
                    // locate the column block
                    int c = col;
                    const Src* src = nullptr;
                    for (const Src& sb : srcs) {
                        if (c < sb.cols) { src = &sb; break; }
                        c -= sb.cols;
                    }
                    for (int j = 0; j < 8; ++j) {
                        const int row = seg_row(seg, s, h, j);
                        if (row < 0) continue;
                        chunk[(size_t)frag * 512 + lane * 8 + j] =
                            f32_to_bf16_rne(src->kernel[(size_t)row * src->cols + c]);
                    }
                }
            }
        }
        for (int n = 0; n < 32; ++n) {
            const int col = 32 * t + n;
            float b = 0.f;
            if (col < n_out) {
                int c = col;
                for (const Src& sb : srcs) {
                    if (c < sb.cols) { b = sb.bias ? sb.bias[c] : 0.f; break; }
                    c -= sb.cols;
                }
            }
            bias_dst[32 * t + n] = b;
        }
    }
    return (size_t)n_tiles * chunk_elems * 2;
}

}  // namespace pack
}  // namespace nfx
