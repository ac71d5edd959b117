// train.hip — training-side kernels that are not part of an MLP chain:
//   amsgrad_step   tf.keras.optimizers.Adam(amsgrad=True) dense update          trainvali.py:110-127
//   wgrad_bf16     dW[K,N] += X[rows,K]^T dZ[rows,N] from feature-major bf16    (tape.gradient, trainvali.py:284)
//                  (+ db[N] += sum_rows dZ[rows,N], folded into the same pass)
// Activations / pre-activation gradients arrive FEATURE-PAIR-major ([feature >> 1][row][feature & 1], bf16:
// feat_store.hpp) from the fused backward kernels: 16 bytes = 4 consecutive rows of two adjacent features.  An MFMA
// operand fragment (one feature x 8 consecutive rows) is two such pieces with the feature's halves picked out by
// v_perm_b32 — done once per piece while the LDS-staged kernels stage their tiles, per fragment in the direct kernel.
#include <stdlib.h>

#include "nfx_common.hpp"

namespace nfx {

// One launch computes up to kWgMaxCalls weight-gradient GEMMs of one backward pass (same rows / leading dimension).
// Every (row slab, dW block) writes its partial sum to its own slice of `part` with plain stores; wgrad_reduce_kernel
// then adds the slabs IN SLAB ORDER into dW / db: the gradients are bit-reproducible from run to run (no float
// atomics anywhere), and a backward pass costs three launches however many layers it has.
constexpr int kWgMaxCalls = 16;
struct WgCall {
    const __bf16* xt;   // layer inputs, [ceil(k_in / 2)][ld][2] (first feature even)
    const __bf16* zt;   // pre-activation gradients, [ceil(n_out / 2)][ld][2]
    float* dw;          // [k_in][n_out] fp32, accumulated into
    float* db;          // [n_out] or null
    float* part;        // [n_slabs][k_pad][n_pad] partial sums of this call
    float* bpart;       // [n_slabs][n_pad] partial bias sums (db != null)
    int k_in, n_out, k_pad, n_pad;
    int block0;         // index of this call's first dW block in the launch's block list
    int gy, gz;         // dW blocks along k / n
};
struct WgBatch {
    WgCall c[kWgMaxCalls];
    int n_calls, n_slabs;
    int wide_only;      // NFX_WGRAD_NARROW=0: the 256 x 256 block form also for narrow GEMMs (A/B)
    long long ld, rows, slab;
    const int* count;   // not null (wgrad_lds_kernel only): the batch's rows are [0, *count) — a number only the device knows
                        // (nerf_bwd.hip, the rows with a gradient); the n_slabs blocks split THOSE rows evenly
};
// the two features of a 16-byte piece (4 rows x [even, odd]) as 8 bytes each: rows r .. r+3 of one feature
__device__ __forceinline__ void wg_split(const u32x4& v, unsigned (&even)[2], unsigned (&odd)[2]) {
    even[0] = __builtin_amdgcn_perm(v[1], v[0], 0x05040100u);
    even[1] = __builtin_amdgcn_perm(v[3], v[2], 0x05040100u);
    odd[0] = __builtin_amdgcn_perm(v[1], v[0], 0x07060302u);
    odd[1] = __builtin_amdgcn_perm(v[3], v[2], 0x07060302u);
}
// feature f's fragment (8 consecutive rows from `row`) straight from global memory
__device__ __forceinline__ bf16x8 wg_fragment(const __bf16* __restrict__ t, long long ld, int f, long long row) {
    const u32x4* p = reinterpret_cast<const u32x4*>(t + ((long long)(f >> 1) * ld + row) * 2);
    const u32x4 lo = p[0], hi = p[1];
    const unsigned sel = (f & 1) ? 0x07060302u : 0x05040100u;
    u32x4 r;
    r[0] = __builtin_amdgcn_perm(lo[1], lo[0], sel);
    r[1] = __builtin_amdgcn_perm(lo[3], lo[2], sel);
    r[2] = __builtin_amdgcn_perm(hi[1], hi[0], sel);
    r[3] = __builtin_amdgcn_perm(hi[3], hi[2], sel);
    return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ int wg_find_call(const WgBatch& b, int block) {
    int ci = 0;
#pragma unroll 1
    for (int i = 1; i < b.n_calls; ++i)
        if (block >= b.c[i].block0) ci = i;
    return ci;
}

// Keras OptimizerV2 Adam._resource_apply_dense with amsgrad (TF 2.2):
//   lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t);  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//   vhat = max(vhat, v);  p -= lr_t * m / (sqrt(vhat) + eps)            (eps = 1e-7)
__global__ void amsgrad_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                               float* __restrict__ v, float* __restrict__ vhat, long long n, float lr_t,
                               float b1, float b2, float eps) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = m[i] * b1 + gi * (1.0f - b1);
    const float vi = v[i] * b2 + (gi * gi) * (1.0f - b2);
    const float vh = fmaxf(vhat[i], vi);
    m[i] = mi;
    v[i] = vi;
    vhat[i] = vh;
    p[i] = p[i] - lr_t * mi / (sqrtf(vh) + eps);
}
// the same update with the step size read from device memory: a captured hipGraph replays it with the step size of
// the CURRENT step (the host refreshes the word before each replay) instead of the one baked in at capture time
__global__ void amsgrad_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                   float* __restrict__ v, float* __restrict__ vhat, long long n,
                                   const float* __restrict__ lr_t_dev, float b1, float b2, float eps) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float lr_t = lr_t_dev[0];
    const float gi = g[i];
    const float mi = m[i] * b1 + gi * (1.0f - b1);
    const float vi = v[i] * b2 + (gi * gi) * (1.0f - b2);
    const float vh = fmaxf(vhat[i], vi);
    m[i] = mi;
    v[i] = vi;
    vhat[i] = vh;
    p[i] = p[i] - lr_t * mi / (sqrtf(vh) + eps);
}

// One wave owns a [<=128 x <=128] block of dW (grid y, z) and a slab of rows (grid x).
//   xt: [k_in][ld] bf16 (feature-major), zt: [n_out][ld] bf16; rows in [row0, row1), multiple of 16.
//   dW: [k_in][n_out] fp32 (Keras layout), accumulated with atomics.
constexpr int kWgTiles = 4;  // 4 x 4 tiles of 32 x 32
__global__ __launch_bounds__(64, 1) void wgrad_kernel(WgBatch bt) {
    const int lane = threadIdx.x, h = lane >> 5, q = lane & 31;
    const int ci = wg_find_call(bt, blockIdx.y);
    const WgCall& c = bt.c[ci];
    const int lb = blockIdx.y - c.block0, by = lb / c.gz, bz = lb % c.gz;
    const __bf16* __restrict__ xt = c.xt;
    const __bf16* __restrict__ zt = c.zt;
    const long long ld = bt.ld;
    const int k_in = c.k_in, n_out = c.n_out;
    const int kb = by * (32 * kWgTiles);  // first input feature of this block
    const int nb = bz * (32 * kWgTiles);  // first output feature of this block
    const long long r0 = (long long)blockIdx.x * bt.slab;
    long long r1 = r0 + bt.slab;
    if (r1 > bt.rows) r1 = bt.rows;
    f32x16 acc[kWgTiles][kWgTiles];
#pragma unroll
    for (int i = 0; i < kWgTiles; ++i)
#pragma unroll
        for (int j = 0; j < kWgTiles; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool do_bias = c.db != nullptr && by == 0;  // db[f] += sum_rows dZ[f][row], once per slab
    float bsum[kWgTiles] = {0.f, 0.f, 0.f, 0.f};
    auto load = [&](long long k0, bf16x8 (&a)[kWgTiles], bf16x8 (&b)[kWgTiles]) {
#pragma unroll
        for (int i = 0; i < kWgTiles; ++i) {
            const int f = kb + 32 * i + q;
            a[i] = f < k_in ? wg_fragment(xt, ld, f, k0 + 8 * h) : zero;
        }
#pragma unroll
        for (int j = 0; j < kWgTiles; ++j) {
            const int f = nb + 32 * j + q;
            b[j] = f < n_out ? wg_fragment(zt, ld, f, k0 + 8 * h) : zero;
        }
    };
    bf16x8 an[kWgTiles], bn[kWgTiles];
    if (r0 < r1) load(r0, an, bn);
    for (long long k0 = r0; k0 < r1; k0 += 16) {
        bf16x8 a[kWgTiles], b[kWgTiles];
#pragma unroll
        for (int i = 0; i < kWgTiles; ++i) {
            a[i] = an[i];
            b[i] = bn[i];
        }
        if (k0 + 16 < r1) load(k0 + 16, an, bn);   // the next k-step's fragments are in flight under this one's MFMAs
        if (do_bias) {
#pragma unroll
            for (int j = 0; j < kWgTiles; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum[j] += (float)b[j][e];
        }
#pragma unroll
        for (int i = 0; i < kWgTiles; ++i)
#pragma unroll
            for (int j = 0; j < kWgTiles; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (do_bias) {
#pragma unroll
        for (int j = 0; j < kWgTiles; ++j) {
            const float s = bsum[j] + __shfl_xor(bsum[j], 32, 64);  // the two row halves of a k-step
            const int col = nb + 32 * j + q;
            if (h == 0) c.bpart[(long long)blockIdx.x * c.n_pad + col] = s;
        }
    }
    float* part = c.part + (long long)blockIdx.x * c.k_pad * c.n_pad;
#pragma unroll
    for (int i = 0; i < kWgTiles; ++i)
#pragma unroll
        for (int j = 0; j < kWgTiles; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = kb + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int col = nb + 32 * j + q;
                if (row < k_in && col < n_out) part[(long long)row * c.n_pad + col] = acc[i][j][r];
            }
}

// dw[row][col] += sum over slabs (in slab order) of part[slab][row][col]; db likewise.  grid = (elements / 256, calls)
// r04: four waves per 64 elements — wave q sums the slabs [q n / 4, (q + 1) n / 4), the four partial sums are combined in
// wave order through LDS (a fixed order: still bit-reproducible).  One thread per element walking all slabs read the
// 154 MB of a NeRF step's partials at 1.75 TB/s (88 us per call); the split form of the fused width-128 backward's
// reduction went from 73 to 22 us for 79 MB.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgBatch bt) {
    __shared__ float quarter[4][64];
    const WgCall& c = bt.c[blockIdx.y];
    const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const long long e = (long long)blockIdx.x * 64 + lane;
    const long long n_w = (long long)c.k_in * c.n_out;
    const int s0 = (int)((long long)bt.n_slabs * seg / 4), s1 = (int)((long long)bt.n_slabs * (seg + 1) / 4);
    float s = 0.0f;
    float* dst = nullptr;
    if (e < n_w) {
        const int row = (int)(e / c.n_out), col = (int)(e % c.n_out);
        const float* p = c.part + (long long)row * c.n_pad + col;
        const long long stride = (long long)c.k_pad * c.n_pad;
        for (int sl = s0; sl < s1; ++sl) s += p[sl * stride];
        dst = c.dw + e;
    } else if (c.db != nullptr && e < n_w + c.n_out) {
        const int col = (int)(e - n_w);
        for (int sl = s0; sl < s1; ++sl) s += c.bpart[(long long)sl * c.n_pad + col];
        dst = c.db + col;
    }
    quarter[seg][lane] = s;
    __syncthreads();
    if (seg == 0 && dst != nullptr) *dst += (quarter[0][lane] + quarter[1][lane]) + (quarter[2][lane] + quarter[3][lane]);
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-staged variant for large row counts: one workgroup (4 waves, one per SIMD) owns a [<=256 x <=256] block of dW
// and a slab of rows; per 64-row chunk the X^T and Z^T tiles (256 features x 128 B each, COALESCED: 8 lanes per
// feature row) are staged global -> VGPR -> LDS (row pitch 144 B: conflict-free ds_read_b128 across 16 features),
// double buffered; each wave multiplies its 128 x 128 quadrant (4 x 4 MFMA tiles, 256 accumulator registers) straight
// from LDS — the feature-major layout already is the A / B fragment layout (one feature x 8 consecutive rows = 16 B).
// Every operand byte is read from HBM once per 256-wide block instead of twice per 128-wide block through strided
// 16-byte loads.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kWlRows = 64;                  // rows per chunk
constexpr int kWlPitch = 2 * kWlRows + 16;   // bytes per feature row in LDS
constexpr int kWlTile = 256 * kWlPitch;      // one operand tile
constexpr int kWlLds = 4 * kWlTile;          // 2 operands x 2 stages = 147456 B

// Narrow GEMMs (k_in, n_out <= 128: every GEMM of a width-128 network): the 256 x 256 block form would leave three of
// the four waves idle and half of the staging loads empty, i.e. 32 KB in flight per CU — the kernel ran at 2.6 TB/s,
// bound by memory latency.  Here a chunk is 128 rows (256-byte runs per feature row, 64 KB per chunk and workgroup),
// every thread stages 8 useful pieces per operand, and the four waves split the chunk's rows (two 16-row k-steps each)
// accumulating the FULL 128 x 128 block; at the end the four accumulators are added through LDS in a fixed order
// (1 -> 0, 3 -> 2, 2 -> 0), so the workgroup still writes one partial block and the result stays order-independent.
constexpr int kWnRows = 128;
constexpr int kWnPitch = 2 * kWnRows + 16;    // 272 B: conflict-free ds_read_b128 across 16 features
constexpr int kWnTile = 128 * kWnPitch;       // one operand tile
static_assert(4 * kWnTile <= kWlLds && 2 * 65536 + 2048 <= kWlLds, "narrow mode fits the LDS of the wide one");

__device__ __forceinline__ void wgrad_lds_narrow(const WgBatch& bt, const WgCall& c, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, q = lane & 31;
    const __bf16* __restrict__ xt = c.xt;
    const __bf16* __restrict__ zt = c.zt;
    const long long ld = bt.ld;
    const int k_in = c.k_in, n_out = c.n_out;
    const long long r0 = (long long)blockIdx.x * bt.slab;
    long long r1 = r0 + bt.slab;
    if (r1 > bt.rows) r1 = bt.rows;
    const int n_chunks = (int)((r1 - r0 + kWnRows - 1) / kWnRows);
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    u32x4 sx[8], sz[8];   // piece p = i * 256 + tid -> feature pair p >> 5, 16-byte part (4 rows x 2 features) p & 31
    auto load_chunk = [&](long long row0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pp = i * 256 + tid, pr = pp >> 5, part = pp & 31;
            const long long row = row0 + part * 4;
            const bool rok = row < r1;
            sx[i] = (2 * pr < k_in && rok) ? *reinterpret_cast<const u32x4*>(xt + ((long long)pr * ld + row) * 2) : zero4;
            sz[i] = (2 * pr < n_out && rok) ? *reinterpret_cast<const u32x4*>(zt + ((long long)pr * ld + row) * 2) : zero4;
        }
    };
    auto store_chunk = [&](int stage) {   // the LDS tiles stay [feature][row]: the pairs are split here
        char* bx = smem + stage * 2 * kWnTile;
        char* bz = bx + kWnTile;
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pp = i * 256 + tid, pr = pp >> 5, part = pp & 31;
            unsigned e[2], o[2];
            wg_split(sx[i], e, o);
            // odd feature counts (63, 27 inputs; 1 or 3 outputs): the odd half of the last pair is never written by the
            // backward kernels — zero it here instead of feeding stale bits to the MFMAs
            if (2 * pr + 1 >= k_in) o[0] = o[1] = 0u;
            *reinterpret_cast<u32x2*>(bx + (2 * pr) * kWnPitch + part * 8) = u32x2{e[0], e[1]};
            *reinterpret_cast<u32x2*>(bx + (2 * pr + 1) * kWnPitch + part * 8) = u32x2{o[0], o[1]};
            wg_split(sz[i], e, o);
            if (2 * pr + 1 >= n_out) o[0] = o[1] = 0u;
            *reinterpret_cast<u32x2*>(bz + (2 * pr) * kWnPitch + part * 8) = u32x2{e[0], e[1]};
            *reinterpret_cast<u32x2*>(bz + (2 * pr + 1) * kWnPitch + part * 8) = u32x2{o[0], o[1]};
        }
    };
    const bool do_bias = c.db != nullptr;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    load_chunk(r0);
    store_chunk(0);
    __syncthreads();
    for (int ch = 0; ch < n_chunks; ++ch) {
        if (ch + 1 < n_chunks) load_chunk(r0 + (long long)(ch + 1) * kWnRows);
        const char* bx = smem + (ch & 1) * 2 * kWnTile + q * kWnPitch + h * 16;
        const char* bz = bx + kWnTile;
#pragma unroll 1
        for (int ss = 0; ss < 2; ++ss) {   // rolled: 32 fragment registers live next to the 256 accumulators
            const int s = 2 * wave + ss;   // this wave's two 16-row k-steps of the chunk
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8*>(bx + 32 * i * kWnPitch + s * 32);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(bz + 32 * j * kWnPitch + s * 32);
            if (do_bias) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum[j] += (float)b[j][e];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (ch + 1 < n_chunks) store_chunk((ch + 1) & 1);
        __syncthreads();
    }
    // ---- the four row-partial accumulators -> wave 0, in a fixed order; slot e of lane l at (e * 64 + l) * 16 bytes
    f32x4* red = reinterpret_cast<f32x4*>(smem);
    float* bred = reinterpret_cast<float*>(smem + 2 * 65536);   // [4 waves][128 columns]
    auto put = [&](int region) {
        f32x4* dst = red + region * 4096 + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    dst[((i * 4 + j) * 4 + g) * 64] = v;
                }
    };
    auto add = [&](int region) {
        const f32x4* src = red + region * 4096 + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = src[((i * 4 + j) * 4 + g) * 64];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] += v[e];
                }
    };
    if (do_bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sm = bsum[j] + __shfl_xor(bsum[j], 32, 64);
            if (h == 0) bred[wave * 128 + 32 * j + q] = sm;
        }
    }
    if (wave & 1) put(wave >> 1);
    __syncthreads();
    if (!(wave & 1)) add(wave >> 1);
    __syncthreads();
    if (wave == 2) put(0);
    __syncthreads();
    if (wave != 0) return;
    add(0);
    if (do_bias && lane < 32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = 32 * j + lane;
            c.bpart[(long long)blockIdx.x * c.n_pad + col] =
                ((bred[col] + bred[128 + col]) + bred[256 + col]) + bred[384 + col];
        }
    }
    float* part = c.part + (long long)blockIdx.x * c.k_pad * c.n_pad;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int col = 32 * j + q;
                if (row < k_in && col < n_out) part[(long long)row * c.n_pad + col] = acc[i][j][r];
            }
}

__global__ __launch_bounds__(256, 1) void wgrad_lds_narrow_kernel(WgBatch bt) {   // every call: k_in, n_out <= 128
    extern __shared__ __attribute__((aligned(16))) char smem[];
    wgrad_lds_narrow(bt, bt.c[wg_find_call(bt, blockIdx.y)], smem);
}

__global__ __launch_bounds__(256, 1) void wgrad_lds_kernel(WgBatch bt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, q = lane & 31;
    const int ci = wg_find_call(bt, blockIdx.y);
    const WgCall& c = bt.c[ci];
    const int lb = blockIdx.y - c.block0;
    const __bf16* __restrict__ xt = c.xt;
    const __bf16* __restrict__ zt = c.zt;
    const long long ld = bt.ld;
    const int k_in = c.k_in, n_out = c.n_out;
    const int kb = (lb / c.gz) * 256, nb = (lb % c.gz) * 256;
    const int wk = wave >> 1, wn = wave & 1;
    long long rows = bt.rows, slab = bt.slab;
    if (bt.count) {   // a slab past the counted rows: zero chunks, and its block of partial sums is written as zeros
        rows = ((long long)*bt.count + 15) & ~15ll;
        slab = ((rows + gridDim.x - 1) / gridDim.x + kWlRows - 1) / kWlRows * kWlRows;
        if (slab < 256) slab = 256;
    }
    const long long r0 = (long long)blockIdx.x * slab;
    long long r1 = r0 + slab;
    if (r1 > rows) r1 = rows;
    if (r1 < r0) r1 = r0;
    const int n_chunks = (int)((r1 - r0 + kWlRows - 1) / kWlRows);
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    // staging: piece p = i * 256 + tid -> feature pair p >> 4 (of this block's 128), 16-byte part (4 rows x 2 features) p & 15
    u32x4 sx[8], sz[8];
    auto load_chunk = [&](long long row0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pp = i * 256 + tid, pr = pp >> 4, part = pp & 15;
            const long long row = row0 + part * 4;  // rows beyond r1 inside the last chunk: zero (ld is padded, but
            const bool rok = row < r1;              // another slab's rows must not be counted twice)
            sx[i] = (kb + 2 * pr < k_in && rok) ? *reinterpret_cast<const u32x4*>(xt + ((long long)(kb / 2 + pr) * ld + row) * 2) : zero4;
            sz[i] = (nb + 2 * pr < n_out && rok) ? *reinterpret_cast<const u32x4*>(zt + ((long long)(nb / 2 + pr) * ld + row) * 2) : zero4;
        }
    };
    auto store_chunk = [&](int stage) {   // the LDS tiles stay [feature][row]: the pairs are split here
        char* bx = smem + stage * 2 * kWlTile;
        char* bz = bx + kWlTile;
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pp = i * 256 + tid, pr = pp >> 4, part = pp & 15;
            unsigned e[2], o[2];
            wg_split(sx[i], e, o);
            if (kb + 2 * pr + 1 >= k_in) o[0] = o[1] = 0u;   // (odd feature counts: see wgrad_lds_narrow_kernel)
            *reinterpret_cast<u32x2*>(bx + (2 * pr) * kWlPitch + part * 8) = u32x2{e[0], e[1]};
            *reinterpret_cast<u32x2*>(bx + (2 * pr + 1) * kWlPitch + part * 8) = u32x2{o[0], o[1]};
            wg_split(sz[i], e, o);
            if (nb + 2 * pr + 1 >= n_out) o[0] = o[1] = 0u;
            *reinterpret_cast<u32x2*>(bz + (2 * pr) * kWlPitch + part * 8) = u32x2{e[0], e[1]};
            *reinterpret_cast<u32x2*>(bz + (2 * pr + 1) * kWlPitch + part * 8) = u32x2{o[0], o[1]};
        }
    };
    const bool active = kb + 128 * wk < k_in && nb + 128 * wn < n_out;  // wave-uniform: quadrant has real features
    const bool do_bias = c.db != nullptr && kb == 0 && wk == 0;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    load_chunk(r0);
    store_chunk(0);
    __syncthreads();
    for (int c = 0; c < n_chunks; ++c) {
        if (c + 1 < n_chunks) load_chunk(r0 + (long long)(c + 1) * kWlRows);
        const char* bx = smem + (c & 1) * 2 * kWlTile + (128 * wk + q) * kWlPitch + h * 16;
        const char* bz = smem + (c & 1) * 2 * kWlTile + kWlTile + (128 * wn + q) * kWlPitch + h * 16;
        if (active) {
#pragma unroll 1
            for (int s = 0; s < kWlRows / 16; ++s) {   // rolled: 32 fragment registers live, not 128
                bf16x8 a[4], b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8*>(bx + 32 * i * kWlPitch + s * 32);
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(bz + 32 * j * kWlPitch + s * 32);
                if (do_bias) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < 8; ++e) bsum[j] += (float)b[j][e];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        if (c + 1 < n_chunks) store_chunk((c + 1) & 1);
        __syncthreads();
    }
    if (!active) return;
    if (do_bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sm = bsum[j] + __shfl_xor(bsum[j], 32, 64);
            const int col = nb + 128 * wn + 32 * j + q;
            if (h == 0) c.bpart[(long long)blockIdx.x * c.n_pad + col] = sm;
        }
    }
    float* part = c.part + (long long)blockIdx.x * c.k_pad * c.n_pad;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = kb + 128 * wk + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int col = nb + 128 * wn + 32 * j + q;
                if (row < k_in && col < n_out) part[(long long)row * c.n_pad + col] = acc[i][j][r];
            }
}

}  // namespace nfx

extern "C" {
int nfx_launch_amsgrad(float* p, const float* g, float* m, float* v, float* vhat, long long n, float lr_t,
                       float b1, float b2, float eps, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(nfx::amsgrad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, g, m, v,
                       vhat, n, lr_t, b1, b2, eps);
    return (int)hipGetLastError();
}
int nfx_launch_amsgrad_dev(float* p, const float* g, float* m, float* v, float* vhat, long long n,
                           const float* lr_t_dev, float b1, float b2, float eps, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(nfx::amsgrad_dev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, g, m, v,
                       vhat, n, lr_t_dev, b1, b2, eps);
    return (int)hipGetLastError();
}
// Host-side description of one GEMM of a batch (capi_train.cpp fills these).
struct nfx_wgrad_call {
    const void* xt;
    const void* zt;
    int k_in, n_out;
    float* dw;
    float* db;
};

int nfx_option_int(const char* key, int dflt);   // capi.cpp

// blocks of 256 x 256 outputs the batch has when it takes the wide LDS form (wgrad_lds_kernel); 0: every GEMM is narrow
static int wg_wide_blocks(const nfx_wgrad_call* calls, int n_calls) {
    bool all_narrow = nfx_option_int("wgrad_narrow", 1) != 0;
    int blocks = 0;
    for (int i = 0; i < n_calls; ++i) {
        all_narrow = all_narrow && calls[i].k_in <= 128 && calls[i].n_out <= 128;
        blocks += ((calls[i].k_in + 255) / 256) * ((calls[i].n_out + 255) / 256);
    }
    return all_narrow ? 0 : blocks;
}

static void wgrad_plan(long long rows, int wide_blocks, bool* use_lds, long long* slab, int* n_slabs) {
    const int force_lds = nfx_option_int("wgrad_lds", -1);
    *use_lds = force_lds >= 0 ? force_lds != 0 : rows >= 16384;
    const int n_forced = nfx_option_int("wgrad_slabs", 0);
    long long sl;
    if (*use_lds) {
        // fewer, longer slabs = fewer partial blocks to write and sum; at least 64 slabs, at most one per CU
        if (n_forced > 0) {
            sl = rows / n_forced;
        } else if (wide_blocks > 0 && nfx_option_int("wgrad_rounds", 1) > 0) {
            // the wide form is one workgroup per CU and HBM-bound on the stored activations: what the slab count buys is
            // only partial sums to write and read back (3.7 MB per slab for the NeRF batch).  One workgroup per CU in
            // all — slabs x blocks <= 256 — measured best (profiles/r06/wgrad_slabs_sweep.txt: 18 / 36 slabs are the
            // minima for the NeRF batch's 14 blocks, 18 the lower)
            const int per_round = 256 / wide_blocks > 0 ? 256 / wide_blocks : 1;
            sl = (rows + (long long)per_round * nfx_option_int("wgrad_rounds", 1) - 1) / ((long long)per_round * nfx_option_int("wgrad_rounds", 1));
        } else {
            sl = rows / 256 > 2048 ? rows / 256 : 2048;
            const long long cap = rows / 64 > 256 ? rows / 64 : 256;
            if (sl > cap) sl = cap;
        }
        sl = (sl + 63) / 64 * 64;
        if (sl < 256) sl = 256;
    } else {
        // latency-bound regime (one wave per block walks its slab with dependent loads): short slabs in parallel
        sl = n_forced > 0 ? rows / n_forced : 128;
        sl = (sl + 15) / 16 * 16;
        if (sl < 16) sl = 16;
    }
    *slab = sl;
    *n_slabs = (int)((rows + sl - 1) / sl);
}
static long long wg_round(long long v, int m) { return (v + m - 1) / m * m; }

// bytes of partial-sum workspace a batch needs (16-byte aligned pieces)
size_t nfx_wgrad_partial_bytes(const nfx_wgrad_call* calls, int n_calls, long long rows) {
    if (rows <= 0 || n_calls <= 0) return 0;
    bool lds;
    long long slab;
    int n_slabs;
    wgrad_plan(rows, wg_wide_blocks(calls, n_calls), &lds, &slab, &n_slabs);
    const int bs = lds ? 256 : 128;
    size_t total = 0;
    for (int i = 0; i < n_calls; ++i) {
        const long long kp = wg_round(calls[i].k_in, bs), np = wg_round(calls[i].n_out, bs);
        total += (size_t)n_slabs * (kp * np + np) * sizeof(float);
    }
    return total;
}

// `count` not null: rows = the most rows the batch can hold (the plan and the partial-sum workspace are sized for it), the
// rows that count are [0, *count), read by the kernel.  Only the wide LDS form reads it: hipErrorInvalidValue otherwise.
int nfx_launch_wgrad_batch_counted(const nfx_wgrad_call* calls, int n_calls, long long ld, long long rows, void* partial,
                                   const int* count, hipStream_t st) {
    if (rows <= 0 || n_calls <= 0) return 0;
    if (n_calls > nfx::kWgMaxCalls) return (int)hipErrorInvalidValue;
    bool lds;
    nfx::WgBatch bt;
    wgrad_plan(rows, wg_wide_blocks(calls, n_calls), &lds, &bt.slab, &bt.n_slabs);
    bt.count = count;
    bt.n_calls = n_calls;
    bt.ld = ld;
    bt.rows = rows;
    bt.wide_only = nfx_option_int("wgrad_narrow", 1) == 0;
    const int bs = lds ? 256 : 128;
    float* p = static_cast<float*>(partial);
    int blocks = 0;
    long long max_elems = 0;
    for (int i = 0; i < n_calls; ++i) {
        nfx::WgCall& c = bt.c[i];
        c.xt = (const __bf16*)calls[i].xt;
        c.zt = (const __bf16*)calls[i].zt;
        c.dw = calls[i].dw;
        c.db = calls[i].db;
        c.k_in = calls[i].k_in;
        c.n_out = calls[i].n_out;
        c.k_pad = (int)wg_round(c.k_in, bs);
        c.n_pad = (int)wg_round(c.n_out, bs);
        c.gy = c.k_pad / bs;
        c.gz = c.n_pad / bs;
        c.block0 = blocks;
        blocks += c.gy * c.gz;
        c.part = p;
        p += (size_t)bt.n_slabs * c.k_pad * c.n_pad;
        c.bpart = p;
        p += (size_t)bt.n_slabs * c.n_pad;
        const long long el = (long long)c.k_in * c.n_out + c.n_out;
        if (el > max_elems) max_elems = el;
    }
    bool all_narrow = !bt.wide_only;
    for (int i = 0; i < n_calls; ++i) all_narrow = all_narrow && calls[i].k_in <= 128 && calls[i].n_out <= 128;
    if (count && (!lds || all_narrow)) return (int)hipErrorInvalidValue;
    if (lds && all_narrow) {   // width-128 networks: 128-row chunks, the four waves split the rows (wgrad_lds_narrow)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(nfx::wgrad_lds_narrow_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, nfx::kWlLds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(nfx::wgrad_lds_narrow_kernel, dim3((unsigned)bt.n_slabs, (unsigned)blocks), dim3(256),
                           nfx::kWlLds, st, bt);
    } else if (lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(nfx::wgrad_lds_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, nfx::kWlLds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(nfx::wgrad_lds_kernel, dim3((unsigned)bt.n_slabs, (unsigned)blocks), dim3(256), nfx::kWlLds, st, bt);
    } else {
        hipLaunchKernelGGL(nfx::wgrad_kernel, dim3((unsigned)bt.n_slabs, (unsigned)blocks), dim3(64), 0, st, bt);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(nfx::wgrad_reduce_kernel, dim3((unsigned)((max_elems + 63) / 64), (unsigned)n_calls), dim3(256), 0, st, bt);
    return (int)hipGetLastError();
}
int nfx_launch_wgrad_batch(const nfx_wgrad_call* calls, int n_calls, long long ld, long long rows, void* partial,
                           hipStream_t st) {
    return nfx_launch_wgrad_batch_counted(calls, n_calls, ld, rows, partial, nullptr, st);
}
// whether a batch of `rows` rows takes the wide LDS form (the one that can read its row count from the device)
int nfx_wgrad_counted_ok(long long rows) {
    bool lds;
    long long slab;
    int n_slabs;
    wgrad_plan(rows, 0, &lds, &slab, &n_slabs);
    return lds ? 1 : 0;
}
}
