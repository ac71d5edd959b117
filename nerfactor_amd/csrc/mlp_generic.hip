// mlp_generic.hip — the network shapes the tuned kernels do NOT cover (round 4): any mlp.Network the reference can build
// (nerfactor/networks/mlp.py:24-50: widths, activations, skip_at anywhere; nerfactor/models/nerf.py:53-90: mlp_width,
// enc_depth, use_views = False, pos_enc = False) evaluated by ONE runtime-shaped fused kernel, forward only.
//
// The tuned kernels (nerf_mlp_v6.hip, mlp128.hip, lvis_v2.hip) are compile-time specialisations of one architecture
// each — register-resident activations, packed weight streams.  This one trades their speed for generality and keeps
// their arithmetic class: bf16 operands, fp32 accumulation and bias, one v_mfma_f32_32x32x16_bf16 per
// (32 outputs x 32 rows x 16 inputs), transposed formulation H^T = W^T X^T (mlp_engine.hpp):
//   * one wave = 32 rows, no workgroup-level synchronisation at all (waves are independent);
//   * activations live in the wave's own LDS area as ROW-MAJOR bf16 — the network input (kept for the skip
//     concatenations) and two ping-pong hidden buffers — so the B operand of any layer is a plain ds_read_b128 of the
//     lane's row, whatever the width, and a layer's output tile goes back with four 8-byte stores per lane;
//   * weights are packed on the host in LOGICAL feature order (no permutation is needed: the operand comes from LDS,
//     not from the previous tile's accumulators) as 1-KiB A fragments, read straight from global memory / L2;
//   * the layer table (input widths, tiles, activation, fragment and bias offsets) is a kernel argument.
// Limits: network input <= 320 features, hidden widths <= 256, <= 16 layers, output <= 256.
// nfx_embed is the Embedder (embedder.py:23-47) as its own kernel, with the point generation o + d z folded in.
//
// Backward (mlp_generic_bwd_kernel + mlp_generic_wgrad_kernel + two ordered reductions; mlp_generic.hpp has the
// workspace layout): the same wave re-computes its 32 rows' forward, turns dLoss/dy into the output-layer gradient and
// walks the layers back with the TRANSPOSED weight fragments — dH^T = W dZ^T is the forward's loop with another weight
// stream — multiplying by the activation's derivative taken from the stored bf16 outputs.  Every layer's input and
// gradient go to the workspace through the transposing LDS read (tr16.hpp), 1 KiB contiguous per 16 features; the
// weight-gradient kernel then needs no LDS at all: one wave per (32 x 32 tile of dW, row split), both MFMA operands
// 16-byte loads.  Deterministic: fixed split count per problem shape, ordered reductions, no atomics.
#include "mlp_engine.hpp"
#include "mlp_generic.hpp"
#include "tr16.hpp"

namespace nfx {
namespace generic {

constexpr int kWavesPerBlock = 2;
constexpr int kXPitch = kMaxIn * 2 + 16, kHPitch = kMaxHidden * 2 + 16;   // bytes per row (+16: rows 4 banks apart)
constexpr int kWaveLds = 32 * (kXPitch + 2 * kHPitch);
constexpr int kLds = kWavesPerBlock * kWaveLds;
static_assert(kLds <= 160 * 1024, "LDS");

__device__ __forceinline__ float activate(float v, int act) {
    switch (act) {
        case 1: return fmaxf(v, 0.f);
        case 2: return sigmoidf(v);
        case 3: return softplusf(v);
        default: return v;
    }
}

__global__ __launch_bounds__(kWavesPerBlock * 64) void mlp_generic_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, p = lane & 31;
    char* xb = smem + wave * kWaveLds;                 // [32][kXPitch]  network input
    char* hb[2] = {xb + 32 * kXPitch, xb + 32 * kXPitch + 32 * kHPitch};
    const long long n_tiles_rows = (a.n + 31) / 32;
    for (long long rt = (long long)blockIdx.x * kWavesPerBlock + wave; rt < n_tiles_rows; rt += (long long)gridDim.x * kWavesPerBlock) {
        const long long row0 = rt * 32;
        // ---- network input -> bf16 rows (zero padded to a multiple of 16 features); lane half g takes the odd / even 16-byte groups
        const int ks_in = (a.d_in + 15) / 16;
        {
            const long long r = row0 + p < a.n ? row0 + p : a.n - 1;
            const float* src = a.x + r * a.ld_x;
            for (int c0 = 8 * g; c0 < ks_in * 16; c0 += 16) {
                bf16x8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (__bf16)(c0 + j < a.d_in ? src[c0 + j] : 0.f);
                *reinterpret_cast<bf16x8*>(xb + p * kXPitch + c0 * 2) = v;
            }
        }
        int cur = 0;
        for (int l = 0; l < a.n_layers; ++l) {
            const Layer L = a.layer[l];
            const bool last = l == a.n_layers - 1;
            const char* hsrc = hb[cur] + p * kHPitch + g * 16;      // this lane's row, its 8 of every 16 features
            const char* xsrc = xb + p * kXPitch + g * 16;
            char* hdst = hb[cur ^ 1] + p * kHPitch;
            for (int t = 0; t < L.n_tiles; ++t) {
                f32x16 acc;
                {
                    const float* bt = a.biases + L.b_off + 32 * t + 4 * g;   // D row of register r: (r&3) + 8 (r>>2) + 4 g
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(bt + 8 * q);
                        acc[4 * q] = v[0]; acc[4 * q + 1] = v[1]; acc[4 * q + 2] = v[2]; acc[4 * q + 3] = v[3];
                    }
                }
                const char* w = a.weights + ((size_t)L.w_off + (size_t)t * (L.ks_h + L.ks_x)) * kFragBytes + lane * 16;
#pragma unroll 4
                for (int s = 0; s < L.ks_h; ++s) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(w + (size_t)s * kFragBytes);
                    const bf16x8 bf = *reinterpret_cast<const bf16x8*>(hsrc + s * 32);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc, 0, 0, 0);
                }
                w += (size_t)L.ks_h * kFragBytes;
#pragma unroll 4
                for (int s = 0; s < L.ks_x; ++s) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(w + (size_t)s * kFragBytes);
                    const bf16x8 bf = *reinterpret_cast<const bf16x8*>(xsrc + s * 32);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc, 0, 0, 0);
                }
                // D: lane = row p (+ half g), register r = output feature 32 t + (r&3) + 8 (r>>2) + 4 g
                if (last) {
                    if (row0 + p < a.n) {
                        float* dst = a.y + (row0 + p) * a.ld_y + a.col0;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int f = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * g;
                            if (f < L.n_out) dst[f] = activate(acc[r], L.act);
                        }
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {   // registers 4 q .. 4 q + 3 = four consecutive features: one 8-byte store
                        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                        bf16x4 v;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int f = 32 * t + j + 8 * q + 4 * g;
                            v[j] = (__bf16)(f < L.n_out ? activate(acc[4 * q + j], L.act) : 0.f);   // pad features: exact zeros
                        }
                        *reinterpret_cast<bf16x4*>(hdst + (32 * t + 8 * q + 4 * g) * 2) = v;
                    }
                }
            }
            cur ^= 1;
        }
    }
}


// ------------------------------------------------------------------------------------------------ backward
__device__ __forceinline__ float act_grad_logit(float v, int act) {     // d act / d logit
    switch (act) {
        case 1: return v > 0.f ? 1.f : 0.f;
        case 2: { const float s = sigmoidf(v); return s * (1.f - s); }
        case 3: return sigmoidf(v);
        default: return 1.f;
    }
}
__device__ __forceinline__ float act_grad_output(float y, int act) {    // the same from the activated output
    switch (act) {
        case 1: return y > 0.f ? 1.f : 0.f;
        case 2: return y * (1.f - y);
        case 3: return 1.f - expf(-y);
        default: return 1.f;
    }
}
// [32 rows][F features] row-major bf16 in the wave's LDS -> feature rows [frow, frow + F) of the wave's workspace tile.
// 16-lane group q takes rows 4 q .. 4 q + 3 (then + 16): lane i supplies row 4 q + (i >> 2), features f0 + 4 (i & 3) ..,
// receives feature f0 + i of the four rows = one 8-byte store.
__device__ __forceinline__ void store_blocked(const char* lds, int pitch, int F, char* wst, int frow, int lane) {
    const int i = lane & 15, q = lane >> 4;
    const char* src = lds + (4 * q + (i >> 2)) * pitch + 8 * (i & 3);
    char* dst = wst + ((size_t)(frow + i) * 32 + 4 * q) * 2;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int f0 = 0; f0 < F; f0 += 16) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(src + f0 * 2));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(src + 16 * pitch + f0 * 2));
        *reinterpret_cast<s16x4*>(dst + (size_t)f0 * 64) = lo;
        *reinterpret_cast<s16x4*>(dst + (size_t)f0 * 64 + 32) = hi;
    }
}

__global__ __launch_bounds__(kWavesPerBlock * 64) void mlp_generic_bwd_kernel(BwdArgs ba) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Args& a = ba.f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, p = lane & 31;
    char* xb = smem + wave * kWaveLds;
    char* hb[2] = {xb + 32 * kXPitch, xb + 32 * kXPitch + 32 * kHPitch};
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    const int fx = (a.d_in + 31) / 32 * 32, ks_in = (a.d_in + 15) / 16, mx = fx / 32;
    for (long long rt = (long long)blockIdx.x * kWavesPerBlock + wave; rt < ba.tiles; rt += (long long)gridDim.x * kWavesPerBlock) {
        const long long row0 = rt * 32;
        const bool live = row0 + p < a.n;
        const long long r = live ? row0 + p : a.n - 1;
        char* wst = ba.ws + (size_t)rt * ba.feat_rows * 64;
        {
            const float* src = a.x + r * a.ld_x;
            for (int c0 = 8 * g; c0 < fx; c0 += 16) {
                bf16x8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (__bf16)(c0 + j < a.d_in ? src[c0 + j] : 0.f);
                *reinterpret_cast<bf16x8*>(xb + p * kXPitch + c0 * 2) = v;
            }
            store_blocked(xb, kXPitch, fx, wst, 0, lane);
        }
        // ---- forward; the last layer turns dy into its own gradient
        int cur = 0;
        for (int l = 0; l < a.n_layers; ++l) {
            const Layer L = a.layer[l];
            const bool last = l == a.n_layers - 1;
            const char* hsrc = hb[cur] + p * kHPitch + g * 16;
            const char* xsrc = xb + p * kXPitch + g * 16;
            char* hdst = hb[cur ^ 1] + p * kHPitch;
            for (int t = 0; t < L.n_tiles; ++t) {
                f32x16 acc;
                {
                    const float* bt = a.biases + L.b_off + 32 * t + 4 * g;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(bt + 8 * q);
                        acc[4 * q] = v[0]; acc[4 * q + 1] = v[1]; acc[4 * q + 2] = v[2]; acc[4 * q + 3] = v[3];
                    }
                }
                const char* w = a.weights + ((size_t)L.w_off + (size_t)t * (L.ks_h + L.ks_x)) * kFragBytes + lane * 16;
#pragma unroll 4
                for (int s = 0; s < L.ks_h; ++s)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(w + (size_t)s * kFragBytes),
                                                                  *reinterpret_cast<const bf16x8*>(hsrc + s * 32), acc, 0, 0, 0);
                w += (size_t)L.ks_h * kFragBytes;
#pragma unroll 4
                for (int s = 0; s < L.ks_x; ++s)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(w + (size_t)s * kFragBytes),
                                                                  *reinterpret_cast<const bf16x8*>(xsrc + s * 32), acc, 0, 0, 0);
                const float* dyr = ba.dy + r * ba.ld_dy + ba.col0_dy;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bf16x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int f = 32 * t + j + 8 * q + 4 * g;
                        float o = 0.f;
                        if (f < L.n_out) o = last ? (live ? dyr[f] * act_grad_logit(acc[4 * q + j], L.act) : 0.f) : activate(acc[4 * q + j], L.act);
                        v[j] = (__bf16)o;
                    }
                    *reinterpret_cast<bf16x4*>(hdst + (32 * t + 8 * q + 4 * g) * 2) = v;
                }
            }
            if (!last) store_blocked(hb[cur ^ 1], kHPitch, L.n_tiles * 32, wst, ba.b[l].h_row, lane);
            cur ^= 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stored activations are read back below (same wave, own lines)
        // ---- backward: hb[cur] holds dZ of layer l
        bool dx_written = false;
        for (int l = a.n_layers - 1; l >= 0; --l) {
            const Layer L = a.layer[l];
            store_blocked(hb[cur], kHPitch, L.n_tiles * 32, wst, ba.b[l].dz_row, lane);
            const int ks_o = 2 * L.n_tiles, mh = l > 0 ? a.layer[l - 1].n_tiles : 0;
            const char* zsrc = hb[cur] + p * kHPitch + g * 16;
            const char* w = ba.wt + (size_t)ba.b[l].wt_off * kFragBytes + lane * 16;
            for (int mt = 0; mt < mh; ++mt) {
                f32x16 acc;
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll 4
                for (int s = 0; s < ks_o; ++s)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(w + ((size_t)mt * ks_o + s) * kFragBytes),
                                                                  *reinterpret_cast<const bf16x8*>(zsrc + s * 32), acc, 0, 0, 0);
                const int pact = a.layer[l - 1].act;
                const __bf16* hy = reinterpret_cast<const __bf16*>(wst) + (size_t)(ba.b[l - 1].h_row + 32 * mt + 4 * g) * 32 + p;
                char* zdst = hb[cur ^ 1] + p * kHPitch;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bf16x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (__bf16)(acc[4 * q + j] * act_grad_output((float)hy[(8 * q + j) * 32], pact));
                    *reinterpret_cast<bf16x4*>(zdst + (32 * mt + 8 * q + 4 * g) * 2) = v;
                }
            }
            if (ba.dx && L.ks_x > 0) {
                for (int mt = 0; mt < mx; ++mt) {
                    f32x16 acc;
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll 4
                    for (int s = 0; s < ks_o; ++s)
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(w + ((size_t)(mh + mt) * ks_o + s) * kFragBytes),
                                                                      *reinterpret_cast<const bf16x8*>(zsrc + s * 32), acc, 0, 0, 0);
                    if (live) {
                        float* dst = ba.dx + (row0 + p) * ba.ld_dx;
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            const int f = 32 * mt + (q & 3) + 8 * (q >> 2) + 4 * g;
                            if (f < a.d_in) dst[f] = dx_written ? dst[f] + acc[q] : acc[q];
                        }
                    }
                }
                dx_written = true;
            }
            cur ^= 1;
        }
    }
}

// one wave per (32 x 32 tile of one layer's dW, row split): dW[i, o] = sum over rows of IN[row, i] dZ[row, o]
__global__ __launch_bounds__(256) void mlp_generic_wgrad_kernel(WgradArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, m = lane & 31, g = lane >> 5;
    const long long wid = (long long)blockIdx.x * 4 + wave;
    if (wid >= (long long)a.n_jobs * a.splits) return;
    const int job = (int)(wid / a.splits), sp = (int)(wid % a.splits);
    int l = 0;
    while (l + 1 < a.n_layers && a.b[l + 1].job0 <= job) ++l;
    const Layer L = a.layer[l];
    const int mh = l > 0 ? a.layer[l - 1].n_tiles : 0, prev = l > 0 ? a.layer[l - 1].n_out : 0;
    const int local = job - a.b[l].job0, it = local / L.n_tiles, ot = local - it * L.n_tiles;
    const bool from_x = it >= mh;
    const int fa = from_x ? 32 * (it - mh) : a.b[l - (l > 0)].h_row + 32 * it;
    const int fb = a.b[l].dz_row + 32 * ot;
    const long long t0 = a.tiles * sp / a.splits, t1 = a.tiles * (sp + 1) / a.splits;
    const char* pa = a.ws + ((size_t)(fa + m) * 32 + 8 * g) * 2;
    const char* pb = a.ws + ((size_t)(fb + m) * 32 + 8 * g) * 2;
    const size_t tile_bytes = (size_t)a.feat_rows * 64;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll 2
    for (long long t = t0; t < t1; ++t) {
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(pa + t * tile_bytes), a1 = *reinterpret_cast<const bf16x8*>(pa + t * tile_bytes + 32);
        const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(pb + t * tile_bytes), b1 = *reinterpret_cast<const bf16x8*>(pb + t * tile_bytes + 32);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
    }
    // D: column (lane & 31) = o, row (q&3) + 8 (q>>2) + 4 g = i
    const int n_i = from_x ? a.d_in - 32 * (it - mh) : prev - 32 * it, i_base = from_x ? prev + 32 * (it - mh) : 32 * it;
    const int o = 32 * ot + m;
    float* dst = a.partial + (size_t)sp * a.slice + a.b[l].dw_off;
    if (o < L.n_out) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = (q & 3) + 8 * (q >> 2) + 4 * g;
            if (i < n_i) dst[(size_t)(i_base + i) * L.n_out + o] = acc[q];
        }
    }
}

__global__ __launch_bounds__(256) void mlp_generic_wgrad_reduce_kernel(WgradArgs a) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.slice) return;
    int l = 0;
    while (l + 1 < a.n_layers && a.b[l + 1].dw_off <= idx) ++l;
    float s = 0.f;
    for (int sp = 0; sp < a.splits; ++sp) s += a.partial[(size_t)sp * a.slice + idx];
    a.dw[l][idx - a.b[l].dw_off] += s;
}

// one block per (layer, output): db[o] += sum over rows of dZ[row, o], a fixed-shape tree
__global__ __launch_bounds__(256) void mlp_generic_bias_kernel(WgradArgs a) {
    __shared__ float red[256];
    int l = 0, o = blockIdx.x;
    while (o >= a.layer[l].n_out) o -= a.layer[l++].n_out;
    const char* src = a.ws + (size_t)(a.b[l].dz_row + o) * 64;
    const size_t tile_bytes = (size_t)a.feat_rows * 64;
    float s = 0.f;
    for (long long t = threadIdx.x; t < a.tiles; t += 256) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + t * tile_bytes + 16 * c);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (float)v[j];
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.db[l][o] += red[0];
}

__global__ __launch_bounds__(256) void embed_kernel(EmbedArgs a) {
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= a.n) return;
    const long long src = row / a.per_ray;
    float v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (a.mode == 0) v[k] = a.x[src * 3 + k];
        else if (a.mode == 1) v[k] = a.x[src * 3 + k] + a.dir[src * 3 + k] * a.z[row];
        else if (a.mode == 2) v[k] = a.dir[src * 3 + k];
        else v[k] = a.dir[(row - src * a.per_ray) * 3 + k] - a.x[src * 3 + k];   // light (row % per_ray) - point
    }
    if (a.mode == 3) {   // shape.py:128-131: safe_l2_normalize(lxyz - x), eps 1e-6
        const float inv = 1.0f / sqrtf(fmaxf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2], 1e-6f));
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] *= inv;
    }
    float* o = a.out + row * a.ld_out + a.col0;
    int c = 0;
    if (a.incl_input) {
#pragma unroll
        for (int k = 0; k < 3; ++k) o[c++] = v[k];
    }
    for (int f = 0; f < a.n_freqs; ++f) {
        const float s = (float)(1 << f);     // 2^k x is exact in fp32, as in TensorFlow
        float sn[3], cs[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) sincos_cw(v[k] * s, sn[k], cs[k]);
#pragma unroll
        for (int k = 0; k < 3; ++k) o[c + k] = sn[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) o[c + 3 + k] = cs[k];
        c += 6;
    }
}

}  // namespace generic
}  // namespace nfx

extern "C" {
int nfx_launch_mlp_generic(const nfx::generic::Args* args, int max_blocks, hipStream_t st) {
    using namespace nfx::generic;
    if (args->n <= 0) return 0;
    const long long tiles = (args->n + 31) / 32, want = (tiles + kWavesPerBlock - 1) / kWavesPerBlock;
    const int grid = (int)(want < max_blocks ? want : max_blocks);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_generic_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(mlp_generic_kernel, dim3(grid), dim3(kWavesPerBlock * 64), kLds, st, *args);
    return (int)hipGetLastError();
}
int nfx_launch_mlp_generic_bwd(const nfx::generic::BwdArgs* ba, const nfx::generic::WgradArgs* wa, int max_blocks, hipStream_t st) {
    using namespace nfx::generic;
    if (ba->f.n <= 0) return 0;
    const long long want = (ba->tiles + kWavesPerBlock - 1) / kWavesPerBlock;
    const int grid = (int)(want < max_blocks ? want : max_blocks);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_generic_bwd_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(mlp_generic_bwd_kernel, dim3(grid), dim3(kWavesPerBlock * 64), kLds, st, *ba);
    const long long waves = (long long)wa->n_jobs * wa->splits;
    hipLaunchKernelGGL(mlp_generic_wgrad_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, *wa);
    hipLaunchKernelGGL(mlp_generic_wgrad_reduce_kernel, dim3((unsigned)((wa->slice + 255) / 256)), dim3(256), 0, st, *wa);
    int n_bias = 0;
    for (int l = 0; l < wa->n_layers; ++l) n_bias += wa->layer[l].n_out;
    hipLaunchKernelGGL(mlp_generic_bias_kernel, dim3(n_bias), dim3(256), 0, st, *wa);
    return (int)hipGetLastError();
}
int nfx_launch_embed(const nfx::generic::EmbedArgs* a, hipStream_t st) {
    if (a->n <= 0) return 0;
    hipLaunchKernelGGL(nfx::generic::embed_kernel, dim3((unsigned)((a->n + 255) / 256)), dim3(256), 0, st, *a);
    return (int)hipGetLastError();
}
}
