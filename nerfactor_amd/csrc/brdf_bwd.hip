// brdf_bwd.hip — backward of the learned-BRDF specular term (tape.gradient through
// nerfactor.py:413-458 with the FROZEN prior MLP of models/brdf.py:57-66):
//   d spec[n, l]  ->  d z[n, z_dim]  and  d normal[n, 3]
// per (point, light) row:  d logit = d spec * softplus'(logit) for front-lit rows;  dgrad chain through
// the four ReLU layers (weights frozen: no wgrad, nothing stored) INCLUDING the two input-gradient
// products  dx = W0 dZ0 + W3[128:, :] dZ3  (the input re-enters at layer 3);  dx lands, by construction of
// the packed fragments, in the same (k-step, half, element) slots the lane filled in the forward, so
//   d z_i      = dx[z slots]                                                  (summed over the lights)
//   d rusink_k = dx[r_k] + sum_b 2^b ( cos(2^b r_k) dx[sin] - sin(2^b r_k) dx[cos] )
//   d normal   = J^T d rusink,  J = d rusink / d normal by forward-mode duals (geom_ad.hpp)
#include "geom.hpp"
#include "geom_ad.hpp"
#include "mlp128_layout.hpp"
#include "mlp_engine.hpp"
#include "feat_store.hpp"

namespace nfx {
namespace brdfbwd {

constexpr int kNW = 4;
constexpr int kRows = kNW * 32;
// chunks: fwd L0 4x4, L1 4x8, L2 4x8, L3 4x12, out 1x8 | bwd dOut 4x4, dL3x 1x8, dL3 4x8, dL2 4x8, dL1 4x8, dL0x 1x8
constexpr int kFwdFrags = 16 + 32 + 32 + 48 + 8;
constexpr int kBwdFrags = 16 + 8 + 32 + 32 + 32 + 8;
constexpr int kWeightBytes = (kFwdFrags + kBwdFrags) * 1024;
constexpr int kBiasFloats = m128::kMainBiasFloats;
constexpr int kBlobBytes = kWeightBytes + kBiasFloats * 4;

template <int CT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[CT]) {
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
}

template <int KS, int NL_SELF, int NL_NEXT>
__device__ __forceinline__ void dgrad_layer(WStream& ws, int tid, const bf16x8 (&dz)[8][1],
                                            const bf16x8 (&hact)[8][1], bf16x8 (&dout)[8][1]) {
    static_for<0, 4>([&](auto T) {
        constexpr int t = decltype(T)::value;
        f32x16 acc[1];
        tile_init<KS, 0, (t == 3 ? NL_NEXT : NL_SELF), kNW>(
            ws, tid, [&](f32x16(&a)[1]) { zero_acc<1>(a); }, dz, dz, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float hv = (float)hact[2 * t + (r >> 3)][0][r & 7];
            dout[2 * t + (r >> 3)][0][r & 7] = (__bf16)(hv > 0.f ? acc[0][r] : 0.f);
        }
        mfma_operand_fence(dout[2 * t][0]);
        mfma_operand_fence(dout[2 * t + 1][0]);
    });
}

constexpr double kFxScale = 1099511627776.0;   // 2^40 per unit: resolution 9e-13, range +-8e6
__device__ __forceinline__ unsigned long long to_fx(float v) {
    return (unsigned long long)__double2ll_rn((double)v * kFxScale);
}
// fx[n, z_dim + 3] -> d_z[n, z_dim] += , d_normal[n, 3] +=
__global__ void brdf_fx_finish_kernel(const long long* __restrict__ fx, long long n, int z_dim, float* __restrict__ d_z,
                                      float* __restrict__ d_normal) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int w = z_dim + 3;
    if (i >= n * w) return;
    const long long pt = i / w;
    const int c = (int)(i % w);
    const float v = (float)((double)fx[i] / kFxScale);
    if (c < z_dim) d_z[pt * z_dim + c] += v;
    else d_normal[pt * 3 + (c - z_dim)] += v;
}

// LIST (round 6): the rows are the flat (point, light) indices list[0 .. *count) — the rows whose upstream gradient is not zero
// (select_nonzero_kernel; a row with d spec = 0 contributes exactly nothing: its d logit is 0 and every product behind it).  The
// shading backward zeroes d spec of every back-facing light, so about half of the rows — the back-lit half the FORWARD kernel
// never evaluates either (nerfactor.py:429-434) — are not re-computed and not differentiated: 313 -> ~170 us per 1024-ray step.
// Rows of a wave may then belong to different points: the per-point sums are segmented (below).
template <bool LIST>
__global__ __launch_bounds__(kNW * 64, 1) void brdf_spec_bwd_kernel(
    const float* __restrict__ xyz, const float* __restrict__ cam, const float* __restrict__ normal,
    const float* __restrict__ z, int z_dim, const float* __restrict__ lxyz, int n_lights,
    const char* __restrict__ blob, long long n, const float* __restrict__ dspec, long long* __restrict__ fx,
    const int* __restrict__ list, const int* __restrict__ count) {
    // fx: [n, z_dim + 3] fixed-point (2^40) sums of d z and d normal over the point's lights — integer atomics, so the
    // waves that share a point may arrive in any order (float atomics made the step's gradients run-dependent)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using namespace m128;
    long long n_rows = n * n_lights;
    if constexpr (LIST) {
        n_rows = *count;
        if ((long long)blockIdx.x * kRows >= n_rows) return;      // (before the first barrier: the whole workgroup leaves)
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
    float* bias_lds = reinterpret_cast<float*>(smem + 2 * kSlotBytes);
    {
        const float* bsrc = reinterpret_cast<const float*>(blob + kWeightBytes);
        for (int i = tid; i < kBiasFloats; i += kNW * 64) bias_lds[i] = bsrc[i];
    }
    WStream ws;
    ws.gbase = reinterpret_cast<const u32x4*>(blob);
    ws.gend = reinterpret_cast<const u32x4*>(blob + kWeightBytes);
    ws.gnext = ws.gbase;
    ws.ring = smem;
    stream_prologue<1, kNW>(ws, tid);
    const long long n_tiles = (n_rows + kRows - 1) / kRows;
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long m0 = tile * kRows + wave * 32;  // dense: wave-uniform, 32 lights of one point
        bool valid;
        long long pt, mrow;
        int l;
        if constexpr (LIST) {
            const long long r = m0 + p;
            valid = r < n_rows;
            const unsigned mu = (unsigned)list[valid ? r : n_rows - 1];      // n * n_lights < 2^31 (checked by the launcher)
            const unsigned pu = mu / (unsigned)n_lights;
            pt = pu;
            l = (int)(mu - pu * (unsigned)n_lights);
            mrow = mu;
        } else {
            valid = m0 < n_rows;
            const long long mc = valid ? m0 : 0;
            pt = mc / n_lights;
            l = (int)(mc % n_lights) + p;
            mrow = m0 + p;
        }
        float x[3], c[3], nr[3], lp[3], ldir[3], vdir[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            x[k] = xyz[pt * 3 + k];
            c[k] = cam[pt * 3 + k];
            nr[k] = normal[pt * 3 + k];
            lp[k] = lxyz[l * 3 + k];
        }
        dir_to(lp, x, ldir);
        dir_to(c, x, vdir);
        Dual3 rus[3];
        float lz;
        rusink_dual(nr, ldir, vdir, rus, lz);
        const bool front = lz > 0.0f;
        // ---- forward input (same slots as brdf_spec_kernel)
        float v[16];
#pragma unroll
        for (int q = 0; q < 6; ++q) v[q] = sin_shifted_small(rus[q % 3].v * (float)(1 << (q / 3)), h);
        v[6] = h ? rus[2].v : rus[0].v;
        v[7] = h ? z[pt * z_dim] : rus[1].v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = 1 + 2 * j + h;
            v[8 + j] = i < z_dim ? z[pt * z_dim + i] : 0.0f;
        }
        bf16x8 bin[2][1];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bin[s][0][j] = (__bf16)v[8 * s + j];
            mfma_operand_fence(bin[s][0]);
        }
        // ---- forward (re-computed)
        bf16x8 h0[8][1], h1[8][1], h2[8][1], h3[8][1];
        layer<2, 0, 4, 1, 2, true, kNW>(ws, tid, bias_lds, bin, bin, h0);
        layer<8, 0, 4, 2, 2, true, kNW>(ws, tid, bias_lds + 128, h0, bin, h1);
        layer<8, 0, 4, 2, 3, true, kNW>(ws, tid, bias_lds + 256, h1, bin, h2);
        layer<8, 2, 4, 3, 2, true, kNW>(ws, tid, bias_lds + 384, h2, bin, h3);
        f32x16 logit[1];
        tile_raw<8, 0, 1, kNW>(ws, tid, bias_lds + 512, h3, bin, logit);
        // ---- d logit (row 0 of the out tile lives in reg 0 of the half-0 lanes)
        float g = 0.f;
        if (valid && front && h == 0) g = dspec[mrow] * sigmoidf(logit[0][0]);  // softplus' = sigmoid
        bf16x8 dzo[1][1];
#pragma unroll
        for (int j = 0; j < 8; ++j) dzo[0][0][j] = (__bf16)0.f;
        dzo[0][0][0] = (__bf16)g;
        // ---- dgrad chain
        bf16x8 dz3[8][1], dz2[8][1], dz1[8][1], dz0[8][1];
        static_for<0, 4>([&](auto T) {
            constexpr int t = decltype(T)::value;
            f32x16 acc[1];
            tile_init<1, 0, (t == 3 ? 2 : 1), kNW>(
                ws, tid, [&](f32x16(&a)[1]) { zero_acc<1>(a); }, dzo, dzo, acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float hv = (float)h3[2 * t + (r >> 3)][0][r & 7];
                dz3[2 * t + (r >> 3)][0][r & 7] = (__bf16)(hv > 0.f ? acc[0][r] : 0.f);
            }
            mfma_operand_fence(dz3[2 * t][0]);
            mfma_operand_fence(dz3[2 * t + 1][0]);
        });
        f32x16 dx[1];  // gradient w.r.t. the 32 input slots of this lane's half: reg r <-> (s = r>>3, j = r&7)
        tile_init<8, 0, 2, kNW>(ws, tid, [&](f32x16(&a)[1]) { zero_acc<1>(a); }, dz3, dz3, dx);   // W3[128:, :] dZ3
        dgrad_layer<8, 2, 2>(ws, tid, dz3, h2, dz2);
        dgrad_layer<8, 2, 2>(ws, tid, dz2, h1, dz1);
        dgrad_layer<8, 2, 2>(ws, tid, dz1, h0, dz0);
        tile_init<8, 0, 1, kNW>(ws, tid, [&](f32x16(&a)[1]) {}, dz0, dz0, dx);                     // += W0 dZ0
        // ---- input slots -> d rusink, d z
        float dr[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int k = q % 3;
            const float f = (float)(1 << (q / 3));
            const float arg = rus[k].v * f;
            // half 0 holds sin(f r): d/dr = f cos(f r); half 1 holds cos(f r): d/dr = -f sin(f r)
            dr[k] += dx[0][q] * f * (h ? -sin_shifted(arg, 0) : sin_shifted(arg, 1));
        }
        if (h == 0) {
            dr[0] += dx[0][6];
            dr[1] += dx[0][7];
        } else {
            dr[2] += dx[0][6];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) dr[k] += __shfl_xor(dr[k], 32, 64);  // the two halves of the same row
        // d normal = J^T d rusink, summed over this wave's 32 lights
        float dn[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) dn[a] = rus[0].d[a] * dr[0] + rus[1].d[a] * dr[1] + rus[2].d[a] * dr[2];
        float dzv[m128::kMaxZDim];
#pragma unroll
        for (int i = 0; i < m128::kMaxZDim; ++i) dzv[i] = 0.f;
        if (h == 1) dzv[0] = dx[0][7];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // k-step 1: element j of half hh holds z_i, i = 1 + 2j + hh  (compile-time split by half)
            if (1 + 2 * j < m128::kMaxZDim && h == 0) dzv[1 + 2 * j] = dx[0][8 + j];
            if (2 + 2 * j < m128::kMaxZDim && h == 1) dzv[2 + 2 * j] = dx[0][8 + j];
        }
        // Sum over the rows of the wave that belong to ONE point (dense: all 32 of the half; LIST: usually 1-2 points per wave),
        // in FIXED POINT from the first addition on (round 6): integer sums do not depend on how rows are grouped into waves, so
        // the dense and the list form — and any order of the list — give the same bits.  Halves hold disjoint z indices /
        // identical dn; an invalid row contributes 0.
        long long sn[3], sz[m128::kMaxZDim];
#pragma unroll
        for (int a = 0; a < 3; ++a) sn[a] = valid ? (long long)to_fx(dn[a]) : 0ll;
#pragma unroll
        for (int i = 0; i < m128::kMaxZDim; ++i) sz[i] = valid ? (long long)to_fx(dzv[i]) : 0ll;
        auto shfl_xor_ll = [](long long v, int o) {
            const int lo = __shfl_xor((int)(unsigned)v, o, 64), hi = __shfl_xor((int)((unsigned long long)v >> 32), o, 64);
            return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
        };
        unsigned long long todo = __ballot(valid);
        while (todo != 0ull) {                                             // wave-uniform loop over the distinct points
            const int leader = __builtin_ctzll(todo);
            const long long pt_u = (long long)__shfl((int)pt, leader, 64); // (pt < 2^31 in both forms)
            const bool mine = valid && pt == pt_u;
            long long tn[3], tz[m128::kMaxZDim];
#pragma unroll
            for (int a = 0; a < 3; ++a) tn[a] = mine ? sn[a] : 0ll;
#pragma unroll
            for (int i = 0; i < m128::kMaxZDim; ++i) tz[i] = mine ? sz[i] : 0ll;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {
#pragma unroll
                for (int a = 0; a < 3; ++a) tn[a] += shfl_xor_ll(tn[a], o);
#pragma unroll
                for (int i = 0; i < m128::kMaxZDim; ++i) tz[i] += shfl_xor_ll(tz[i], o);
            }
            if (p == 0) {
                unsigned long long* row = reinterpret_cast<unsigned long long*>(fx + pt_u * (z_dim + 3));
                if (h == 0) {
#pragma unroll
                    for (int a = 0; a < 3; ++a) atomicAdd(row + z_dim + a, (unsigned long long)tn[a]);
                }
#pragma unroll
                for (int i = 0; i < m128::kMaxZDim; ++i)
                    if (i < z_dim && ((i == 0) ? h == 1 : ((i - 1) & 1) == h)) atomicAdd(row + i, (unsigned long long)tz[i]);
            }
            todo &= ~__ballot(mine);
        }
    }
}

// list[] <- the indices i < n_elems with x[i] != 0, *count <- their number (zeroed by the launcher).  One atomic per workgroup of
// 1024 elements; the order of the list is whatever the workgroups' arrival makes it (the consumer's sums are integers).
__global__ __launch_bounds__(256) void select_nonzero_kernel(const float* __restrict__ x, unsigned n_elems, int* __restrict__ list,
                                                             int* __restrict__ count) {
    __shared__ int s_cnt[4][4], s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned i0 = blockIdx.x * 1024u + wave * 256u;
    unsigned long long masks[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned i = i0 + k * 64u + lane;
        masks[k] = __ballot(i < n_elems && x[i] != 0.0f);
        if (lane == 0) s_cnt[wave][k] = __popcll(masks[k]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int sum = 0;
        for (int w = 0; w < 4; ++w)
            for (int k = 0; k < 4; ++k) {
                const int c = s_cnt[w][k];
                s_cnt[w][k] = sum;
                sum += c;
            }
        s_base = sum ? atomicAdd(count, sum) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if ((masks[k] >> lane) & 1ull)
            list[s_base + s_cnt[wave][k] + __popcll(masks[k] & ((1ull << lane) - 1ull))] = (int)(i0 + k * 64u + lane);
}


// ---------------------------------------------------------------------------------------------------------------
// The BRDF prior on EXPLICIT rows (models/brdf.py:57-66, 87-136): row r < n evaluates (z[r], rusink[r]); with
// `rows` = 2 n the rows r >= n repeat the inputs with phi_d + pi (the reciprocal Rusinkiewicz coordinates that share
// the ground truth, brdf.py:103-106).  BWD = false: out[rows] = softplus(logit) — the stream wraps after the forward
// chunks of the train blob.  BWD = true: the forward is re-computed, d logit = dout * sigmoid(logit), the dgrad chain
// runs as in brdf_spec_bwd_kernel, d_z[rows, z_dim] is stored per row, and the layer inputs X (LOGICAL Embedder order
// [z | rusink | sin, cos band 0 | sin, cos band 1], so dW comes out in the Keras layout), h0..h3 and the
// pre-activation gradients dZ0..dZ3, dZ_out go feature-major to `wsp` for the weight-gradient GEMMs (train.hip).
constexpr int kRowsX = 32;                       // feature rows reserved for the network input
constexpr int kOffH = kRowsX, kOffDZ = kRowsX + 512, kOffDZo = kRowsX + 1024, kRowFeats = kRowsX + 1032;
static_assert(kRowsX % 2 == 0, "feature-pair-major storage (feat_store.hpp): every group starts on an even feature");

template <bool BWD>
__global__ __launch_bounds__(kNW * 64, 1) void brdf_rows_kernel(
    const float* __restrict__ z, int z_dim, const float* __restrict__ rusink, long long n, long long rows,
    const char* __restrict__ blob, const float* __restrict__ dout, float* __restrict__ out_or_dz,
    __bf16* __restrict__ wsp, long long ld) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using namespace m128;
    using namespace nfx::bwd;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
    float* bias_lds = reinterpret_cast<float*>(smem + 2 * kSlotBytes);
    {
        const float* bsrc = reinterpret_cast<const float*>(blob + kWeightBytes);
        for (int i = tid; i < kBiasFloats; i += kNW * 64) bias_lds[i] = bsrc[i];
    }
    WStream ws;
    ws.gbase = reinterpret_cast<const u32x4*>(blob);
    ws.gend = reinterpret_cast<const u32x4*>(blob + (BWD ? kWeightBytes : kFwdFrags * 1024));
    ws.gnext = ws.gbase;
    ws.ring = smem;
    stream_prologue<1, kNW>(ws, tid);
    const long long n_tiles = (rows + kRows - 1) / kRows;
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long row = tile * kRows + wave * 32 + p;
        const bool valid = row < rows;
        const long long rc = valid ? row : rows - 1;
        const long long src = rc >= n ? rc - n : rc;
        float rus[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) rus[k] = rusink[src * 3 + k];
        if (rc >= n) rus[0] = rus[0] + 3.14159265358979323846f;   // brdf.py:103
        const float* zp = z + src * z_dim;
        float v[16];
#pragma unroll
        for (int q = 0; q < 6; ++q) v[q] = sin_shifted_small(rus[q % 3] * (float)(1 << (q / 3)), h);
        v[6] = h ? rus[2] : rus[0];
        v[7] = h ? zp[0] : rus[1];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = 1 + 2 * j + h;
            v[8 + j] = i < z_dim ? zp[i] : 0.0f;
        }
        bf16x8 bin[2][1];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bin[s][0][j] = (__bf16)v[8 * s + j];
            mfma_operand_fence(bin[s][0]);
        }
        bf16x8 h0[8][1], h1[8][1], h2[8][1], h3[8][1];
        layer<2, 0, 4, 1, 2, true, kNW>(ws, tid, bias_lds, bin, bin, h0);
        layer<8, 0, 4, 2, 2, true, kNW>(ws, tid, bias_lds + 128, h0, bin, h1);
        layer<8, 0, 4, 2, 3, true, kNW>(ws, tid, bias_lds + 256, h1, bin, h2);
        layer<8, 2, 4, 3, 2, true, kNW>(ws, tid, bias_lds + 384, h2, bin, h3);
        f32x16 logit[1];
        tile_raw<8, 0, 1, kNW>(ws, tid, bias_lds + 512, h3, bin, logit);   // next chunk: dOut (BWD) or L0 (4 frags each)
        if constexpr (!BWD) {
            if (valid && h == 0) out_or_dz[row] = softplusf(logit[0][0]);   // brdf.py:65
        } else {
            FeatStore fs;
            {
                unsigned long long ld2 = (unsigned long long)ld * 2, b = reinterpret_cast<unsigned long long>(wsp);
                asm volatile("" : "+s"(ld2), "+s"(b));
                fs.base = reinterpret_cast<char*>(b);
                fs.ld2 = ld2;
                fs.roff = (unsigned)(row * 4);   // pair layout: one dword per row and feature pair (feat_store.hpp)
            }
            {   // network input, logical order [z | rusink, sin, cos bands]: half 0 / half 1 hold different features
#pragma unroll
                for (int q = 0; q < 6; ++q) {   // cosines sit 3 features after the sines
                    const int fa = z_dim + 3 + 6 * (q / 3) + (q % 3);
                    st16_ab(fs, fa, fa + 3, h, bin[0][0][q]);
                }
                st16_ab(fs, z_dim, z_dim + 2, h, bin[0][0][6]);          // theta_d sits 2 features after phi_d
                st16_ab(fs, z_dim + 1, 0, h, bin[0][0][7]);              // half 0: theta_h, half 1: z_0
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (1 + 2 * j + h < z_dim) st16_ab(fs, 1 + 2 * j, 2 + 2 * j, h, bin[1][0][j]);   // z_{i+1} after z_i
            }
            store_hidden<8>(fs, kOffH + 0, h, h0);
            store_hidden<8>(fs, kOffH + 128, h, h1);
            store_hidden<8>(fs, kOffH + 256, h, h2);
            store_hidden<8>(fs, kOffH + 384, h, h3);
            // ---- d logit (row 0 of the out tile lives in reg 0 of the half-0 lanes)
            float g = 0.f;
            if (valid && h == 0) g = dout[row] * sigmoidf(logit[0][0]);      // softplus' = sigmoid
            bf16x8 dzo[1][1];
#pragma unroll
            for (int j = 0; j < 8; ++j) dzo[0][0][j] = (__bf16)0.f;
            dzo[0][0][0] = (__bf16)g;
            {
                FeatStore f4 = relaunder(fs);
                f4.roff = fs.roff + (h ? (unsigned)(4 * fs.ld2) : 0u);
#pragma unroll
                for (int r = 0; r < 4; ++r) st16(f4, kOffDZo + r, dzo[0][0][r]);
            }
            // ---- dgrad chain
            bf16x8 dz3[8][1], dz2[8][1], dz1[8][1], dz0[8][1];
            static_for<0, 4>([&](auto T) {
                constexpr int t = decltype(T)::value;
                f32x16 acc[1];
                tile_init<1, 0, (t == 3 ? 2 : 1), kNW>(
                    ws, tid, [&](f32x16(&a)[1]) { zero_acc<1>(a); }, dzo, dzo, acc);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float hv = (float)h3[2 * t + (r >> 3)][0][r & 7];
                    dz3[2 * t + (r >> 3)][0][r & 7] = (__bf16)(hv > 0.f ? acc[0][r] : 0.f);
                }
                mfma_operand_fence(dz3[2 * t][0]);
                mfma_operand_fence(dz3[2 * t + 1][0]);
            });
            store_hidden<8>(fs, kOffDZ + 384, h, dz3);
            f32x16 dx[1];  // gradient w.r.t. the 32 input slots of this lane's half
            tile_init<8, 0, 2, kNW>(ws, tid, [&](f32x16(&a)[1]) { zero_acc<1>(a); }, dz3, dz3, dx);   // W3[128:, :] dZ3
            dgrad_layer<8, 2, 2>(ws, tid, dz3, h2, dz2);
            store_hidden<8>(fs, kOffDZ + 256, h, dz2);
            dgrad_layer<8, 2, 2>(ws, tid, dz2, h1, dz1);
            store_hidden<8>(fs, kOffDZ + 128, h, dz1);
            dgrad_layer<8, 2, 2>(ws, tid, dz1, h0, dz0);
            store_hidden<8>(fs, kOffDZ + 0, h, dz0);
            tile_init<8, 0, 1, kNW>(ws, tid, [&](f32x16(&a)[1]) {}, dz0, dz0, dx);                     // += W0 dZ0
            if (valid) {   // d z: slot (k-step 0, elem 7) of half 1 holds z_0, k-step 1 elem j of half hh holds z_{1+2j+hh}
                float* dzr = out_or_dz + row * z_dim;
                if (h == 1) dzr[0] = dx[0][7];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (1 + 2 * j + h < z_dim) dzr[1 + 2 * j + h] = dx[0][8 + j];
            }
        }
    }
}

}  // namespace brdfbwd
}  // namespace nfx

extern "C" {
int nfx_brdf_train_blob_bytes(void) { return nfx::brdfbwd::kBlobBytes; }
// list_ws: NULL = the dense form (every row); else 4 (n n_lights + 4) bytes: [count | pad | list] — the rows with d spec != 0 only
int nfx_launch_brdf_spec_bwd(const float* xyz, const float* cam, const float* normal, const float* z, int z_dim,
                             const float* lxyz, int n_lights, const void* blob, long long n, const float* dspec,
                             float* d_z, float* d_normal, void* workspace, int max_blocks, hipStream_t st, void* list_ws) {
    using namespace nfx;
    if (n <= 0) return 0;
    const long long tiles = (n * n_lights + brdfbwd::kRows - 1) / brdfbwd::kRows;
    const int grid = (int)(tiles < max_blocks ? tiles : max_blocks);
    const int lds = 2 * kSlotBytes + m128::kMainBiasFloats * 4;
    const long long words = n * (z_dim + 3);
    launch_zero_words(workspace, words, st);   // (a kernel, not hipMemsetAsync: nfx_common.hpp)
    if (list_ws != nullptr && n * n_lights < (1ll << 31)) {
        int* count = static_cast<int*>(list_ws);
        int* list = count + 4;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(brdfbwd::brdf_spec_bwd_kernel<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        launch_zero_words(count, 2, st);        // (a kernel node, not a memset node: nfx_common.hpp — the step is captured in a hipGraph)
        const unsigned n_elems = (unsigned)(n * n_lights);
        hipLaunchKernelGGL(brdfbwd::select_nonzero_kernel, dim3((n_elems + 1023u) / 1024u), dim3(256), 0, st, dspec, n_elems, list, count);
        hipLaunchKernelGGL(brdfbwd::brdf_spec_bwd_kernel<true>, dim3(grid), dim3(brdfbwd::kNW * 64), lds, st, xyz, cam, normal,
                           z, z_dim, lxyz, n_lights, (const char*)blob, n, dspec, static_cast<long long*>(workspace), list, count);
    } else {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(brdfbwd::brdf_spec_bwd_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(brdfbwd::brdf_spec_bwd_kernel<false>, dim3(grid), dim3(brdfbwd::kNW * 64), lds, st, xyz, cam, normal,
                           z, z_dim, lxyz, n_lights, (const char*)blob, n, dspec, static_cast<long long*>(workspace), nullptr, nullptr);
    }
    hipLaunchKernelGGL(brdfbwd::brdf_fx_finish_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st,
                       static_cast<const long long*>(workspace), n, z_dim, d_z, d_normal);
    return (int)hipGetLastError();
}
int nfx_brdf_rows_feats(void) { return nfx::brdfbwd::kRowFeats; }
// bwd = 0: out_or_dz = out[rows]; bwd = 1: out_or_dz = d_z[rows, z_dim], wsp = feature-major workspace [kRowFeats][ld]
int nfx_launch_brdf_rows(int bwd, const float* z, int z_dim, const float* rusink, long long n, long long rows,
                         const void* blob, const float* dout, float* out_or_dz, void* wsp, long long ld,
                         int max_blocks, hipStream_t st) {
    using namespace nfx;
    if (rows <= 0) return 0;
    const long long tiles = (rows + brdfbwd::kRows - 1) / brdfbwd::kRows;
    const int grid = (int)(tiles < max_blocks ? tiles : max_blocks);
    const int lds = 2 * kSlotBytes + m128::kMainBiasFloats * 4;
    auto launch = [&](auto k) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(brdfbwd::kNW * 64), lds, st, z, z_dim, rusink, n, rows,
                           (const char*)blob, dout, out_or_dz, (__bf16*)wsp, ld);
        return (int)hipGetLastError();
    };
    return bwd ? launch(brdfbwd::brdf_rows_kernel<true>) : launch(brdfbwd::brdf_rows_kernel<false>);
}
}
