// lvis_v2.hip — the two (point, light)-row MLPs of NeRFactor, light visibility (shape.py:213-237) and the learned-BRDF
// specular term (nerfactor.py:413-458), with the whole network RESIDENT in LDS.
// The width-128 net is 136 fragments = 136 KiB (+ 2 KiB of biases): it fits the 160 KiB of a CU, so unlike the NeRF
// kernels there is no weight stream, no staging and — after the one-time load — no barrier at all.  One wave per
// SIMD, CT column tiles of 32 (point, light) rows per wave: one A fragment read from LDS feeds CT MFMAs (CT = 4:
// a quarter of the LDS traffic of the 8 x 32 kernel in mlp128.hip), the epilogue of tile i-1 (accvgpr_read +
// v_cvt_pk_bf16_f32 + v_pk_max_i16 per pair) is issued in the shadow of tile i's MFMAs, the per-point
// pre-activations of layers 0 and 3 (lvis_pre_kernel) are fetched into the free accumulator set one tile ahead.
// MODE 0 = light visibility, MODE 1 = learned BRDF (same chunk geometry: 2 input k-steps, skip into layer 3; biases
// instead of per-point pre-activations, softplus on front-lit rows).  Same blobs and arithmetic as lvis_kernel /
// brdf_spec_kernel of mlp128.hip: bit-identical outputs.
#include "geom.hpp"
#include "mlp128_layout.hpp"
#include "mlp_engine.hpp"


#ifdef NFX_XP_TRANS_LOAD
#ifdef NFX_XP_LOAD_PLAIN_VALU     // control: full-rate VALU instructions of the same total issue time instead of transcendental ones
#define NFX_XP_LOAD_INSN "v_mul_f32 %0, 1.0, %0\n\tv_mul_f32 %0, 1.0, %0\n\tv_mul_f32 %0, 1.0, %0\n\tv_mul_f32 %0, 1.0, %0"
#else
#define NFX_XP_LOAD_INSN "v_sin_f32 %0, %0"
#endif
#endif
namespace nfx {
namespace lv2 {

constexpr int kNW = 4;
constexpr int kLdsNet = m128::kMainWeightBytes + m128::kMainBiasFloats * 4;
// MODE 0 additionally parks the per-point pre-activation rows of the wave's column tiles (CT x 1 KiB per wave), copied
// one point tile ahead: the layer-0 / layer-3 accumulators then start from LDS like a bias, not from a global load
constexpr int kLds = kLdsNet + kNW * 4 * 1024;
// fragment offset of chunk K: L0 0-3 (4 frags), L1 4-7, L2 8-11 (8 frags), L3 12-15 (12 frags), out 16
constexpr int frag_off(int k) { return k < 4 ? 4 * k : k < 12 ? 16 + 8 * (k - 4) : k < 16 ? 80 + 12 * (k - 12) : 128; }

template <int CT>
struct Acc {
    f32x16 v[CT];
};
struct Pre {
    bf16x8 a[2];  // first two A fragments of the next tile (layer 0 has only two k-steps)
};

template <bool RELU>
__device__ __forceinline__ void cvt_pair(float v0, float v1, bf16x8& dst, int j) {
    typedef short s2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 vv = {v0, v1};
    b2 pr = __builtin_convertvector(vv, b2);
    if (RELU) {
        s2 w = __builtin_bit_cast(s2, pr);
        const s2 z = {0, 0};
        w = __builtin_elementwise_max(w, z);
        pr = __builtin_bit_cast(b2, w);
    }
    // (the converted pair is an MFMA operand of the next layer: mlp_engine.hpp, "MFMA operands written by packed ...")
    pr = __builtin_bit_cast(b2, mfma_operand_dword(__builtin_bit_cast(unsigned, pr)));
    dst[j] = pr[0];
    dst[j + 1] = pr[1];
}

template <int CT>
struct EpiB {   // ReLU layer output -> next B operand (k-steps lo / hi), in register-pair pieces
    const Acc<CT>& acc;
    bf16x8 (&lo)[CT];
    bf16x8 (&hi)[CT];
    template <int R0, int R1>
    __device__ __forceinline__ void run() {
#pragma unroll
        for (int r = R0; r < R1; r += 2)
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                if (r < 8) cvt_pair<true>(acc.v[c][r], acc.v[c][r + 1], lo[c], r);
                else cvt_pair<true>(acc.v[c][r], acc.v[c][r + 1], hi[c], r - 8);
            }
    }
};
struct EpiNone {
    template <int R0, int R1>
    __device__ __forceinline__ void run() {}
};

// accumulator initialisers for the NEXT tile
struct InitBias {   // broadcast LDS reads of the bias tile
    const float* bias_tile;
    template <int CT>
    __device__ __forceinline__ void operator()(int lane, Acc<CT>& acc) const {
        // one read group, the other column tiles' accumulators by register copy (accumulators are ArchVGPRs here):
        // light-visibility kernel 17.57 -> 17.1 ms on r01; NFX_LV2_BIAS_READS restores one read group per column tile
        const float* bt = bias_tile + 4 * (lane >> 5);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(bt + 8 * g);
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                acc.v[c][4 * g + 0] = v[0];
                acc.v[c][4 * g + 1] = v[1];
                acc.v[c][4 * g + 2] = v[2];
                acc.v[c][4 * g + 3] = v[3];
            }
        }
    }
};
template <int CT>
struct InitPre {    // per-point pre-activation rows parked in the wave's LDS area (one point per column tile)
    const float* rows;   // [CT][256] floats
    int off;             // 32 t (+128 for layer 3)
    __device__ __forceinline__ void operator()(int lane, Acc<CT>& acc) const {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const float* bt = rows + c * 256 + off + 4 * (lane >> 5);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(bt + 8 * g);
                acc.v[c][4 * g + 0] = v[0];
                acc.v[c][4 * g + 1] = v[1];
                acc.v[c][4 * g + 2] = v[2];
                acc.v[c][4 * g + 3] = v[3];
            }
        }
    }
};

// Tile K: acc (already initialised) += W_K^T [b1 ; b2]; the previous tile's epilogue `prev` is spread over the
// k-steps; once it is complete `init_next` fills acc_next for tile K+1; the first A fragments of chunk K1 are read
// at the end.  No barrier: the weights never change.
template <int K, int K1, int KS1, int KS2, int CT, int KS1A, int KS2A, typename Epi, typename Init>
__device__ __forceinline__ void tile(const char* wlds, int lane, const bf16x8 (&b1)[KS1A][CT],
                                     const bf16x8 (&b2)[KS2A][CT], Acc<CT>& acc, Acc<CT>& acc_next, Pre& pre,
                                     Epi&& prev, Init&& init_next) {
    constexpr int KS = KS1 + KS2;
    // 16 accumulator registers of the previous tile in PIECES groups; at a layer boundary its outputs are this tile's
    // LAST input k-steps, so the epilogue must be complete well before them: 4 pieces (done after k-step 3) for 8-10
    // k-steps
    constexpr int PIECES = KS >= 4 ? 4 : KS;
    static_assert(16 % PIECES == 0 && (16 / PIECES) % 2 == 0, "pairs");
    const char* f0 = wlds + frag_off(K) * kFragBytes + lane * 16;
    bf16x8 abuf[4];
    abuf[0] = pre.a[0];
    abuf[1] = pre.a[1];
    if constexpr (KS > 2) abuf[2] = *reinterpret_cast<const bf16x8*>(f0 + 2 * kFragBytes);
    static_for<0, KS>([&](auto S) {
        constexpr int s = decltype(S)::value;
        if constexpr (s + 3 < KS) abuf[(s + 3) % 4] = *reinterpret_cast<const bf16x8*>(f0 + (s + 3) * kFragBytes);
        const bf16x8 a = abuf[s % 4];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const bf16x8 b = s < KS1 ? b1[s < KS1 ? s : 0][c] : b2[s >= KS1 ? s - KS1 : 0][c];
            acc.v[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc.v[c], 0, 0, 0);
        }
        if constexpr (s < PIECES) prev.template run<16 * s / PIECES, 16 * (s + 1) / PIECES>();
        if constexpr (s == PIECES - 1) {
            // keep the next tile's initial loads BEHIND the epilogue that frees their destination registers (hoisted
            // above it they need 16 CT temporaries and spill at CT = 4)
            __builtin_amdgcn_sched_barrier(0);
            init_next(lane, acc_next);
        }
    });
    const char* f1 = wlds + frag_off(K1) * kFragBytes + lane * 16;
    pre.a[0] = *reinterpret_cast<const bf16x8*>(f1);
    pre.a[1] = *reinterpret_cast<const bf16x8*>(f1 + kFragBytes);
}

struct Args {
    const float* xyz;      // [n,3] points (lvis: the points the directions are taken from)
    const float* lxyz;     // [L,3]
    const float* pre;      // MODE 0: [n,256] per-point pre-activations
    const float* cam;      // MODE 1: [n,3]
    const float* normal;   // MODE 1: [n,3]
    const float* z;        // MODE 1: [n,z_dim]
    int z_dim;
    long long n;
    int n_lights;
    const char* blob;
    float* out;            // [n,L]
    // MODE 0, round 6 (both optional): row of point i in the caller's FULL [n_all, L] buffer (the tf.scatter_nd of
    // shape.py:171-176 done by the store itself: the compact [n, L] tensor is never written and never re-read), and the
    // tf.debugging.check_numerics flag: 1 is OR-ed into it when a visibility is NaN
    const int* out_row;
    int* nan_flag;
};

// NW = 4: one wave per SIMD with up to 512 registers (CT = 3 | 4).  NW = 8 (CT = 2): two waves per SIMD with 256
// registers each share the LDS copy of the network — a wave's stalls (epilogue bursts, the slow first tile of a
// layer, encoder VALU at the start of a point tile) are covered by its partner's MFMAs; an A fragment feeds 2 MFMAs,
// 64 B/clk of LDS reads per CU at full MFMA rate, a quarter of the ds_read_b128 peak.
// ROWS (MODE 0, round 6): the stores go to the rows Args::out_row names and NaNs raise Args::nan_flag.  A separate instantiation:
// the plain kernel keeps the exact code that 5e11 rows of soak have seen (255 registers, no scratch).
template <int CT, int MODE, int NW, bool ROWS = false>
__global__ __launch_bounds__(NW * 64, NW / 4) void resident128_kernel(Args a) {
    constexpr int kNW = NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using namespace m128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
    {   // the whole network, once
        const u32x4* src = reinterpret_cast<const u32x4*>(a.blob);
        u32x4* dst = reinterpret_cast<u32x4*>(smem);
        for (int i = tid; i < kLdsNet / 16; i += kNW * 64) dst[i] = src[i];
        __syncthreads();
    }
    const char* wlds = smem;
    const float* bias_lds = reinterpret_cast<const float*>(smem + kMainWeightBytes);
    constexpr int kTileRows = kNW * CT * 32;
    const int n_lights = a.n_lights;
    const long long n_rows = a.n * n_lights;
    const long long n_tiles = (n_rows + kTileRows - 1) / kTileRows;
    float* pre_rows = reinterpret_cast<float*>(smem + kLdsNet) + wave * CT * 256;   // this wave's [CT][256] floats
    // point of column tile c in point tile t (clamped), and the copy of its pre row: one 16-byte piece per lane
    auto div_l = [&](long long v) -> long long { return v / n_lights; };
    auto mod_l = [&](long long v) -> int { return (int)(v % n_lights); };
    auto point_of = [&](long long t, int c) {
        const long long m = t * kTileRows + (wave * CT + c) * 32;
        return div_l(m < n_rows ? m : 0);
    };
    u32x4 stage[CT];
    auto fetch_rows = [&](long long t) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int c = 0; c < CT; ++c)
                stage[c] = *reinterpret_cast<const u32x4*>(a.pre + point_of(t, c) * 256 + lane * 4);
        }
    };
    auto park_rows = [&]() {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int c = 0; c < CT; ++c) *reinterpret_cast<u32x4*>(pre_rows + c * 256 + lane * 4) = stage[c];
        }
    };
    fetch_rows(blockIdx.x);
    park_rows();
    // point position and light position of this lane's row in column tile c of point tile t, fetched one point tile
    // ahead: the first touch of a point's xyz is an HBM miss (~2-3 k cycles) that the first MFMA of the tile waited for
    float xq[CT][3], lq[CT][3];
    // ROWS: where column tile c of the NEXT point tile stores — out_row[point] * L + first light — as two wave-uniform words
    // (readfirstlane: SGPRs; the point index is computed here anyway, a division at the store would cost the registers the
    // kernel does not have)
    unsigned obq_lo[CT], obq_hi[CT];
    auto fetch_inputs = [&](long long t) {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const long long m = t * kTileRows + (wave * CT + c) * 32;
            const long long mc = m < n_rows ? m : 0;
            const long long pt = div_l(mc);
            const int l0 = mod_l(mc);
            const int l = l0 + p;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                xq[c][k] = a.xyz[pt * 3 + k];
                lq[c][k] = a.lxyz[l * 3 + k];
            }
            if constexpr (ROWS) {
                // everything scalar: the point index is wave-uniform, so its row comes by s_load (no vector-memory wait in
                // front of the prefetch — a readfirstlane of a vector LOAD made every tile wait for the loads it had just issued)
                const unsigned long long ptu = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)pt >> 32)) << 32) |
                                               __builtin_amdgcn_readfirstlane((unsigned)pt);
                const long long ob = (long long)a.out_row[ptu] * n_lights + __builtin_amdgcn_readfirstlane(l0);
                obq_lo[c] = (unsigned)ob;
                obq_hi[c] = (unsigned)((unsigned long long)ob >> 32);
            }
        }
    };
    fetch_inputs(blockIdx.x);
    unsigned long long bad = 0ull;      // lanes that saw a NaN: a ballot, wave-uniform — SGPRs, not one more VGPR (the kernel sits at 255)
    for (long long tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
        bf16x8 pl[2][CT];
        const float* pre_pt[CT];
        const long long tnext = tl + gridDim.x < n_tiles ? tl + gridDim.x : tl;
        fetch_rows(tnext);   // next point tile's rows: in flight during this whole tile, parked after the last layer-3 initialiser
        long long m0[CT];
        bool front[CT];
        float xc[CT][3], lc[CT][3];
        unsigned ob_lo[CT], ob_hi[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                xc[c][k] = xq[c][k];
                lc[c][k] = lq[c][k];
            }
            if constexpr (ROWS) {
                ob_lo[c] = obq_lo[c];
                ob_hi[c] = obq_hi[c];
            }
        }
        fetch_inputs(tnext);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            m0[c] = tl * kTileRows + (wave * CT + c) * 32;   // 32 consecutive lights of one point (n_lights % 32 == 0)
            const long long mc = m0[c] < n_rows ? m0[c] : 0;
            const long long pt = div_l(mc);
            float x[3], lp[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                x[k] = xc[c][k];
                lp[k] = lc[c][k];
            }
            if constexpr (MODE == 0) {
                float d[3], sq = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    d[k] = lp[k] - x[k];   // _calc_ldir (shape.py:128-131): normalize(lxyz[l] - x), eps 1e-6
                    sq += d[k] * d[k];
                }
                const float inv = 1.0f / sqrtf(fmaxf(sq, 1e-6f));
#pragma unroll
                for (int k = 0; k < 3; ++k) d[k] *= inv;
#ifdef NFX_XP_TRANS_LOAD
                {
                    float tl = d[0];
#pragma unroll
                    for (int i = 0; i < NFX_XP_TRANS_LOAD; ++i) asm volatile(NFX_XP_LOAD_INSN : "+v"(tl));
                    asm volatile("" ::"v"(tl));
                }
#endif
                posenc<4, CT>(d, h, c, pl);
                pre_pt[c] = a.pre + pt * 256;
                front[c] = true;
            } else {
                float cm[3], nr[3], ldir[3], vdir[3], rot[9], ll[3], vl[3], rus[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    cm[k] = a.cam[pt * 3 + k];
                    nr[k] = a.normal[pt * 3 + k];
                }
                dir_to(lp, x, ldir);          // shape.py:128-131
                dir_to(cm, x, vdir);          // shape.py:137-140
                world2local(nr, rot);         // util/geom.py:119-149
                mat3_apply(rot, ldir, ll);    // nerfactor.py:418-419
                mat3_apply(rot, vdir, vl);
                dir2rusink(ll, vl, rus);      // util/geom.py:152-192 with a = light, b = view
                front[c] = ll[2] > 0.0f;      // nerfactor.py:429-432
                // B operand slots of brdf_input_slots() (capi_nerfactor.cpp)
                float v[16];
#pragma unroll
                for (int q = 0; q < 6; ++q) v[q] = sin_shifted_small(rus[q % 3] * (float)(1 << (q / 3)), h);   // angles <= pi, 2 bands
                v[6] = h ? rus[2] : rus[0];
                v[7] = h ? a.z[pt * a.z_dim] : rus[1];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = 1 + 2 * j + h;
                    v[8 + j] = i < a.z_dim ? a.z[pt * a.z_dim + i] : 0.0f;
                }
#pragma unroll
                for (int sidx = 0; sidx < 2; ++sidx) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) pl[sidx][c][j] = (__bf16)v[8 * sidx + j];
                    mfma_operand_fence(pl[sidx][c]);
                }
                pre_pt[c] = nullptr;
            }
        }
        bf16x8 ha[8][CT], hb[8][CT];
        Acc<CT> accs[2];
        Pre pre;
        // initial accumulators of the layer-0 / layer-3 tiles: per-point pre-activations (MODE 0) or plain biases
        auto init03 = [&](int off_pre, int off_bias) {
            return [=](int ln, Acc<CT>& ac) {
                if constexpr (MODE == 0) InitPre<CT>{pre_rows, off_pre}(ln, ac);
                else InitBias{bias_lds + off_bias}(ln, ac);
            };
        };
        {   // tile 0's operands
            const char* f0 = wlds + lane * 16;
            pre.a[0] = *reinterpret_cast<const bf16x8*>(f0);
            pre.a[1] = *reinterpret_cast<const bf16x8*>(f0 + kFragBytes);
            init03(0, 0)(lane, accs[0]);
        }
        // chunk K accumulates in accs[K & 1]; layers: L0 K 0-3 (pl -> ha), L1 4-7 (ha -> hb), L2 8-11 (hb -> ha),
        // L3 12-15 ([ha ; pl] -> hb), out 16 (hb -> activation)
#define NFX_LV2_TILE(K, KS1, KS2, B1, B2, PREV, NEXT) \
        tile<K, (K + 1) % 17, KS1, KS2, CT>(wlds, lane, B1, B2, accs[(K) & 1], accs[((K) + 1) & 1], pre, PREV, NEXT)
#define NFX_LV2_EPI(K, OUT, T) EpiB<CT>{accs[(K) & 1], OUT[2 * (T)], OUT[2 * (T) + 1]}
        NFX_LV2_TILE(0, 2, 0, pl, pl, EpiNone{}, init03(32, 32));
        NFX_LV2_TILE(1, 2, 0, pl, pl, NFX_LV2_EPI(0, ha, 0), init03(64, 64));
        NFX_LV2_TILE(2, 2, 0, pl, pl, NFX_LV2_EPI(1, ha, 1), init03(96, 96));
        NFX_LV2_TILE(3, 2, 0, pl, pl, NFX_LV2_EPI(2, ha, 2), (InitBias{bias_lds + 128}));
        NFX_LV2_TILE(4, 8, 0, ha, pl, NFX_LV2_EPI(3, ha, 3), (InitBias{bias_lds + 128 + 32}));
        NFX_LV2_TILE(5, 8, 0, ha, pl, NFX_LV2_EPI(4, hb, 0), (InitBias{bias_lds + 128 + 64}));
        NFX_LV2_TILE(6, 8, 0, ha, pl, NFX_LV2_EPI(5, hb, 1), (InitBias{bias_lds + 128 + 96}));
        NFX_LV2_TILE(7, 8, 0, ha, pl, NFX_LV2_EPI(6, hb, 2), (InitBias{bias_lds + 256}));
        NFX_LV2_TILE(8, 8, 0, hb, pl, NFX_LV2_EPI(7, hb, 3), (InitBias{bias_lds + 256 + 32}));
        NFX_LV2_TILE(9, 8, 0, hb, pl, NFX_LV2_EPI(8, ha, 0), (InitBias{bias_lds + 256 + 64}));
        NFX_LV2_TILE(10, 8, 0, hb, pl, NFX_LV2_EPI(9, ha, 1), (InitBias{bias_lds + 256 + 96}));
        NFX_LV2_TILE(11, 8, 0, hb, pl, NFX_LV2_EPI(10, ha, 2), init03(128, 384));
        NFX_LV2_TILE(12, 8, 2, ha, pl, NFX_LV2_EPI(11, ha, 3), init03(128 + 32, 384 + 32));
        NFX_LV2_TILE(13, 8, 2, ha, pl, NFX_LV2_EPI(12, hb, 0), init03(128 + 64, 384 + 64));
        NFX_LV2_TILE(14, 8, 2, ha, pl, NFX_LV2_EPI(13, hb, 1), init03(128 + 96, 384 + 96));
        NFX_LV2_TILE(15, 8, 2, ha, pl, NFX_LV2_EPI(14, hb, 2), (InitBias{bias_lds + 512}));
        park_rows();   // the rows of THIS point tile were last read by tile 14's initialiser (LDS ops of a wave are in order)
        NFX_LV2_TILE(16, 8, 0, hb, pl, NFX_LV2_EPI(15, hb, 3), [](int, Acc<CT>&) {});
#undef NFX_LV2_TILE
#undef NFX_LV2_EPI
        if (h == 0) {
#pragma unroll
            for (int c = 0; c < CT; ++c)
                if (m0[c] < n_rows) {
                    const float o = accs[0].v[c][0];
                    if constexpr (MODE == 0) {
                        const float v = sigmoidf(o);                                    // shape.py:93 sigmoid out
                        if constexpr (ROWS) {
                            const long long at = (long long)(((unsigned long long)ob_hi[c] << 32) | ob_lo[c]);
                            a.out[at + p] = v;
                            bad |= __ballot(!(v == v));
                        } else {
                            a.out[m0[c] + p] = v;
                        }
                    } else {
                        a.out[m0[c] + p] = front[c] ? softplusf(o) : 0.0f;              // brdf.py:65, back-lit rows 0
                    }
                }
        }
    }
    if constexpr (ROWS) {
        if (a.nan_flag != nullptr && bad != 0ull && lane == 0) atomicOr(a.nan_flag, 1);
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Learned-BRDF specular term with FRONT-LIT COMPACTION (nerfactor.py:429-441: the reference evaluates the BRDF MLP on
// the rows with local l.z > 0 only and scatters them into a zero tensor; about half of the light sphere).
// Every wave works alone (no barrier after the one-time network load): it owns the points gw, gw + NW, gw + 2 NW, ...
// (interleaved, so spatially coherent normals do not unbalance the waves), FILLS a private LDS ring with the
// front-lit (point, light) rows of its next point — ballot + mbcnt prefix, back-lit rows get their 0 right there —
// and whenever the ring holds a full pass (CT x 32 rows) runs the 17-tile network on rows that may straddle points.
// GEO = 0: the per-row geometry of resident128_kernel<CT, 1> (same functions, bit-identical outputs).
// GEO = 1: Rusinkiewicz angles without the two Rodrigues rotations' sin / cos and without the half-vector azimuth
//          (cos phi_h = h_x / s, sin phi_h = h_y / s, sin theta_h = s = |h_xy|), polynomial acos / atan2, and the
//          two-band encoding of the three angles from double-angle identities instead of 6 v_sin / v_cos.
// ring of queued rows per wave: 1024 entries of 16 bits = (point slot << 10) | light, point slot = local point index
// mod 8 (a pass decodes it against the index of the newest filled point; the fill loop never lets the queue span 8
// points).  2 KiB per wave: 8 waves (two per SIMD) fit next to the network.
#ifndef NFX_BRDF_SWAP
#define NFX_BRDF_SWAP 1   // 1 ds_bpermute (default) | 0 __builtin_amdgcn_permlane32_swap | 2 hand-timed asm: brdf_compact_kernel's header
#endif
constexpr int kRing = 1024;
typedef unsigned short ring_t;
// queue geometry per wave: ring entries and point slots.  4 waves: 1024 entries, 8 slots.  8 waves (experiment form):
// 704 entries (>= 64 - 1 + 512 lights), 4 slots — what fits beside the network, the lights and the point tables
template <int NW> struct Queue {
    static constexpr int kCap = NW == 8 ? 704 : kRing, kSlots = NW == 8 ? 4 : 8;
    static constexpr int lds_bytes(int n_lights) {
        return kLdsNet + NW * kCap * (int)sizeof(ring_t) + (n_lights * 12 + 15) / 16 * 16 + NW * kSlots * 32 * 4;
    }
};
__device__ __forceinline__ int ring_wrap(int i, int cap) { return i >= cap ? i - cap : i; }   // i < 2 cap

__device__ __forceinline__ float acos_poly(float x) {   // Abramowitz-Stegun 4.4.46, |err| <= 2e-8 + fp32 rounding
    const float ax = fabsf(x);
    float p = -0.0012624911f;
    p = fmaf(p, ax, 0.0066700901f);
    p = fmaf(p, ax, -0.0170881256f);
    p = fmaf(p, ax, 0.0308918810f);
    p = fmaf(p, ax, -0.0501743046f);
    p = fmaf(p, ax, 0.0889789874f);
    p = fmaf(p, ax, -0.2145988016f);
    p = fmaf(p, ax, 1.5707963050f);
    const float r = sqrtf(1.0f - ax) * p;
    return x < 0.0f ? 3.14159265358979323846f - r : r;
}
__device__ __forceinline__ float atan2_poly(float y, float x) {   // Cephes atanf polynomial, |err| <= 3e-7
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float t = mx > 0.0f ? mn * __builtin_amdgcn_rcpf(mx) : 0.0f;
    const bool big = t > 0.4142135623730950f;
    const float t2 = big ? (t - 1.0f) * __builtin_amdgcn_rcpf(t + 1.0f) : t;
    const float zz = t2 * t2;
    float q = fmaf(8.05374449538e-2f, zz, -1.38776856032e-1f);
    q = fmaf(q, zz, 1.99777106478e-1f);
    q = fmaf(q, zz, -3.33329491539e-1f);
    float r = fmaf(q * zz, t2, t2) + (big ? 0.78539816339744830962f : 0.0f);
    r = ay > ax ? 1.57079632679489661923f - r : r;
    r = x < 0.0f ? 3.14159265358979323846f - r : r;
    return y < 0.0f ? -r : r;
}

// B-operand values (slot order of brdf_input_slots(), capi_nerfactor.cpp) of one (point, light) row, lane half h
template <int GEO>
__device__ __forceinline__ void brdf_row_inputs(const float (&x)[3], const float (&lp)[3], const float (&cm)[3],
                                                const float (&nr)[3], const float* zp, int z_dim, int h,
                                                float (&v)[16]) {
    float ldir[3], vdir[3], rot[9], ll[3], vl[3];
    if constexpr (GEO == 0) {
        dir_to(lp, x, ldir);          // shape.py:128-131
        dir_to(cm, x, vdir);          // shape.py:137-140
        world2local(nr, rot);         // util/geom.py:119-149
    } else {
        // same formulas with v_rsq_f32 (1 ulp) instead of the IEEE sqrt + divide sequences (~22 instructions per
        // normalisation, eight of them per row, executed at dependent-issue latency by a lone wave)
        auto nrm = [](float (&v)[3]) {
            const float inv = __builtin_amdgcn_rsqf(fmaxf(dot3(v, v), 1e-6f));
            v[0] *= inv; v[1] *= inv; v[2] *= inv;
        };
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            ldir[k] = lp[k] - x[k];
            vdir[k] = cm[k] - x[k];
        }
        nrm(ldir);
        nrm(vdir);
        float n[3] = {nr[0], nr[1], nr[2]}, t[3], b[3];
        nrm(n);
        const float zax[3] = {0.0f + 1e-6f, 0.0f + 1e-6f, 1.0f + 1e-6f};   // geom.py:128
        cross3(n, zax, t);
        nrm(t);
        cross3(n, t, b);
        nrm(b);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            rot[k] = t[k];
            rot[3 + k] = b[k];
            rot[6 + k] = n[k];
        }
    }
    mat3_apply(rot, ldir, ll);    // nerfactor.py:418-419
    mat3_apply(rot, vdir, vl);
    if constexpr (GEO == 0) {
        float rus[3];
        dir2rusink(ll, vl, rus);  // util/geom.py:152-192 with a = light, b = view
#pragma unroll
        for (int q = 0; q < 6; ++q) v[q] = sin_shifted_small(rus[q % 3] * (float)(1 << (q / 3)), h);
        v[6] = h ? rus[2] : rus[0];
        v[7] = h ? zp[0] : rus[1];
    } else {
        auto nrm = [](float (&v)[3]) {
            const float inv = __builtin_amdgcn_rsqf(fmaxf(dot3(v, v), 1e-6f));
            v[0] *= inv; v[1] *= inv; v[2] *= inv;
        };
        nrm(ll);                  // dir2rusink re-normalises its inputs (geom.py:158-159)
        nrm(vl);
        float hv[3] = {(ll[0] + vl[0]) * 0.5f, (ll[1] + vl[1]) * 0.5f, (ll[2] + vl[2]) * 0.5f};
        nrm(hv);
        const float cth = fminf(fmaxf(hv[2], -1.0f), 1.0f);
        const float sxy = sqrtf(hv[0] * hv[0] + hv[1] * hv[1]);           // sin(theta_h) >= 0
        const float inv = sxy > 0.0f ? __builtin_amdgcn_rcpf(sxy) : 0.0f;
        const float cph = sxy > 0.0f ? hv[0] * inv : 1.0f, sph = hv[1] * inv;   // atan2(0, 0) = 0
        // diff = R_y(-theta_h) R_z(-phi_h) b with b = view (geom.py:183)
        const float t0 = vl[0] * cph + vl[1] * sph, t1 = vl[1] * cph - vl[0] * sph;
        const float d0 = t0 * cth - vl[2] * sxy, d1 = t1, d2 = vl[2] * cth + t0 * sxy;
        const float ctd = fminf(fmaxf(d2, -1.0f), 1.0f);
        const float rxy = sqrtf(d0 * d0 + d1 * d1);                       // sin(theta_d) >= 0
        const float theta_h = acos_poly(cth), theta_d = acos_poly(ctd);
        const float pi = 3.14159265358979323846f;
        float phi_d = atan2_poly(d1, d0);
        phi_d = phi_d - floorf(phi_d / pi) * pi;                          // tf.math.floormod(x, pi)
        const float rinv = rxy > 0.0f ? __builtin_amdgcn_rcpf(rxy) : 0.0f;
        const bool flip = d1 < 0.0f || (d1 == 0.0f && d0 < 0.0f);        // + pi: both signs change
        const float cpd0 = rxy > 0.0f ? d0 * rinv : 1.0f, spd0 = d1 * rinv;
        const float cpd = flip ? -cpd0 : cpd0, spd = flip ? -spd0 : spd0;
        // band 0: (phi_d, theta_h, theta_d), band 1: the doubled angles
        if (h == 0) {
            v[0] = spd; v[1] = sxy; v[2] = rxy;
            v[3] = 2.0f * spd * cpd; v[4] = 2.0f * sxy * cth; v[5] = 2.0f * rxy * ctd;
        } else {
            v[0] = cpd; v[1] = cth; v[2] = ctd;
            v[3] = cpd * cpd - spd * spd; v[4] = cth * cth - sxy * sxy; v[5] = ctd * ctd - rxy * rxy;
        }
        v[6] = h ? theta_d : phi_d;
        v[7] = h ? zp[0] : theta_h;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = 1 + 2 * j + h;
        v[8 + j] = i < z_dim ? zp[i] : 0.0f;
    }
}

// The closed-form geometry of ONE (point, light) row (brdf_point_frame + brdf_row_angles) with the values of BOTH lane halves: S[q] is what the half-0 lane
// of the row's column puts into input slot q (sines, phi_d, theta_h), C[q] what the half-1 lane does (cosines, theta_d;
// slot 7 of half 1 is z_0, loaded by the caller).  brdf_compact_kernel lets the half-0 lane of column p compute the row
// of column tile 2k and the half-1 lane the row of tile 2k + 1, and swaps halves with v_permlane32_swap: each lane
// runs the geometry of CT / 2 rows instead of CT (the two halves used to compute the same row twice).
__device__ __forceinline__ void nrm_rsq(float (&v)[3]) {
    const float inv = __builtin_amdgcn_rsqf(fmaxf(dot3(v, v), 1e-6f));
    v[0] *= inv; v[1] *= inv; v[2] *= inv;
}
// The per-POINT half of the row geometry: local frame [t; b; n] of the normal and the view direction in it.  The queue
// fill computes it once per point and parks it in LDS (r03); brdf_row_angles is what is left per (point, light) row.
// Same operations in the same order as the per-row form of round 2: the rows come out bit-identical.
__device__ __forceinline__ void brdf_point_frame(const float (&x)[3], const float (&cm)[3], const float (&nr)[3],
                                                 float (&rot)[9], float (&vl)[3]) {
    float vdir[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) vdir[k] = cm[k] - x[k];
    nrm_rsq(vdir);
    float n[3] = {nr[0], nr[1], nr[2]}, t[3], b[3];
    nrm_rsq(n);
    const float zax[3] = {0.0f + 1e-6f, 0.0f + 1e-6f, 1.0f + 1e-6f};   // geom.py:128
    cross3(n, zax, t);
    nrm_rsq(t);
    cross3(n, t, b);
    nrm_rsq(b);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        rot[k] = t[k];
        rot[3 + k] = b[k];
        rot[6 + k] = n[k];
    }
    mat3_apply(rot, vdir, vl);
    nrm_rsq(vl);
}
__device__ __forceinline__ void brdf_row_angles(const float (&x)[3], const float (&lp)[3], const float (&rot)[9],
                                                const float (&vl)[3], float (&S)[8], float (&C)[8]) {
    auto nrm = [](float (&v)[3]) { nrm_rsq(v); };
    float ldir[3], ll[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) ldir[k] = lp[k] - x[k];
    nrm(ldir);
    mat3_apply(rot, ldir, ll);
    nrm(ll);
    float hv[3] = {(ll[0] + vl[0]) * 0.5f, (ll[1] + vl[1]) * 0.5f, (ll[2] + vl[2]) * 0.5f};
    nrm(hv);
    const float cth = fminf(fmaxf(hv[2], -1.0f), 1.0f);
    const float sxy = sqrtf(hv[0] * hv[0] + hv[1] * hv[1]);
    const float inv = sxy > 0.0f ? __builtin_amdgcn_rcpf(sxy) : 0.0f;
    const float cph = sxy > 0.0f ? hv[0] * inv : 1.0f, sph = hv[1] * inv;
    const float t0 = vl[0] * cph + vl[1] * sph, t1 = vl[1] * cph - vl[0] * sph;
    const float d0 = t0 * cth - vl[2] * sxy, d1 = t1, d2 = vl[2] * cth + t0 * sxy;
    const float ctd = fminf(fmaxf(d2, -1.0f), 1.0f);
    const float rxy = sqrtf(d0 * d0 + d1 * d1);
    const float theta_h = acos_poly(cth), theta_d = acos_poly(ctd);
    const float pi = 3.14159265358979323846f;
    float phi_d = atan2_poly(d1, d0);
    phi_d = phi_d - floorf(phi_d / pi) * pi;
    const float rinv = rxy > 0.0f ? __builtin_amdgcn_rcpf(rxy) : 0.0f;
    const bool flip = d1 < 0.0f || (d1 == 0.0f && d0 < 0.0f);
    const float cpd0 = rxy > 0.0f ? d0 * rinv : 1.0f, spd0 = d1 * rinv;
    const float cpd = flip ? -cpd0 : cpd0, spd = flip ? -spd0 : spd0;
    S[0] = spd; S[1] = sxy; S[2] = rxy;
    S[3] = 2.0f * spd * cpd; S[4] = 2.0f * sxy * cth; S[5] = 2.0f * rxy * ctd;
    S[6] = phi_d; S[7] = theta_h;
    C[0] = cpd; C[1] = cth; C[2] = ctd;
    C[3] = cpd * cpd - spd * spd; C[4] = cth * cth - sxy * sxy; C[5] = ctd * ctd - rxy * rxy;
    C[6] = theta_d; C[7] = 0.0f;
}

// NW = 8, CT = 2 (default since r03): two waves per SIMD — one wave's queue fill and Rusinkiewicz VALU run under its
//   partner's MFMAs.  NW = 4 (CT = 2 | 3 | 4): one wave per SIMD, the shipped form of round 2.
// Round 2 measured the 8-wave form faster but NOT deterministic (a few thousand of 10^8 rows wrong, in groups of 16
// lanes of the second column tile, different from run to run) and did not find out why.  Round 3 did: the exchange of
// the two lane halves' geometry by v_permlane32_swap_b32.  With a partner wave on the SIMD the swap reads operands a
// VALU instruction wrote two wait states earlier before they have landed — with the hand-placed `s_nop 1` of rounds
// 1-2 (NFX_BRDF_SWAP=2: 11 000 rows of 10^8 wrong per call) AND with the compiler's own
// __builtin_amdgcn_permlane32_swap, whose hazard recogniser places the same two wait states (NFX_BRDF_SWAP=0: 2 200
// rows).  Through ds_bpermute (NFX_BRDF_SWAP=1, the default: the LDS crossbar, counted by lgkmcnt) the 8-wave kernel is
// bit-identical to the 4-wave one on every call (scripts/brdf_nw8_soak.py).  One wave per SIMD never showed the fault
// (10^10 rows in round 2), but the shipped form does not depend on those wait states any more.
#ifdef NFX_XP_VGPR_CAP      // round-6 experiment: the two-waves-per-SIMD forms may not touch the top of their half of the register file
#define NFX_XP_CAP_ATTR __attribute__((amdgpu_num_vgpr(NFX_XP_VGPR_CAP)))      // (an integer constant: every instantiation; the one-wave reference forms spill, slower but the same sums)
#else
#define NFX_XP_CAP_ATTR
#endif
template <int CT, int GEO, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) NFX_XP_CAP_ATTR void brdf_compact_kernel(Args a) {
    constexpr int kNW = NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using namespace m128;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: point indices and ring base live in SGPRs
#ifdef NFX_XP_FORCE_SCRATCH
    // round-6 experiment: the failing <2, 0, 8> is the ONLY kernel of the library whose register
    // allocation spills to SCRATCH MEMORY (private_segment_fixed_size 20: the lane half h, stored once and re-loaded in every
    // pass, and a constant pair).  This build gives the healthy <2, 1, 8> the same thing: h goes through a private slot.
    volatile int xp_slot[4];
    xp_slot[0] = h;
    xp_slot[2] = 0x3f317218;
#endif
    {   // the whole network, once
        const u32x4* src = reinterpret_cast<const u32x4*>(a.blob);
        u32x4* dst = reinterpret_cast<u32x4*>(smem);
        for (int i = tid; i < kLdsNet / 16; i += kNW * 64) dst[i] = src[i];
        // the light positions too (<= 12 KiB): the queue fill classifies all of them for every point, and a pass
        // gathers one per row — LDS reads instead of global loads with a vmcnt(0) per 64 lights
        float* ldst = reinterpret_cast<float*>(smem + kLdsNet + kNW * Queue<NW>::kCap * (int)sizeof(ring_t));
        for (int i = tid; i < 3 * a.n_lights; i += kNW * 64) ldst[i] = a.lxyz[i];
        __syncthreads();
    }
    const float* lx = reinterpret_cast<const float*>(smem + kLdsNet + kNW * Queue<NW>::kCap * (int)sizeof(ring_t));
    const char* wlds = smem;
    const float* bias_lds = reinterpret_cast<const float*>(smem + kMainWeightBytes);
    constexpr int kCap = Queue<NW>::kCap, kSlots = Queue<NW>::kSlots;
    ring_t* ring = reinterpret_cast<ring_t*>(smem + kLdsNet) + wave * kCap;
    // per-point table of the wave (kPark): 8 slots (the ring's 3-bit point slot) x 32 floats =
    //   [x(3) | view dir in the local frame(3) | local frame t, b, n (9) | z_0 | z_1, z_3, .. (8) | z_2, z_4, .. (8)]
    constexpr bool kPark = GEO == 1 && CT % 2 == 0;
    float* ptab = reinterpret_cast<float*>(smem + kLdsNet + kNW * kCap * (int)sizeof(ring_t) + (a.n_lights * 12 + 15) / 16 * 16)
                  + wave * (kSlots * 32);
    const int L = a.n_lights;
    const long long n = a.n;
    const long long nw = (long long)gridDim.x * kNW, gw = (long long)blockIdx.x * kNW + wave;
    constexpr int kPass = CT * 32;
    long long kfill = 0, k_head = 0;   // next point to fill; point of the oldest queued row
    int head = 0, cnt = 0;
    for (;;) {
        // ---- fill: front-lit rows of the next points until a whole pass is queued (ring: kPass - 1 + L <= kRing)
        while (cnt < kPass) {
            const long long pt = gw + kfill * nw;
            if (pt >= n) break;
            if (cnt > 0 && kfill - k_head >= kSlots - 1) break;   // the queue may span at most kSlots points (point slot)
            // the point's position and normal: wave-uniform scalar loads (a partner wave's MFMAs cover their latency)
            float x[3], nr[3], rot[9];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                x[k] = a.xyz[pt * 3 + k];
                nr[k] = a.normal[pt * 3 + k];
            }
            world2local(nr, rot);
#ifdef NFX_XP_LDS_INPUTS
            // round-6 experiment: the GEO = 0 pass takes its per-point inputs from the wave's LDS table instead of per-lane
            // global loads (x, cam, normal, z: raw values, the per-row op sequence unchanged) — no vector-memory LOAD inside a pass
            if constexpr (!kPark) {
                float cmx[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) cmx[k] = a.cam[pt * 3 + k];
                const float* zp = a.z + pt * a.z_dim;
                float zv[17];
#pragma unroll
                for (int i = 0; i < 17; ++i) zv[i] = i < a.z_dim ? zp[i] : 0.0f;
                if (lane == 0) {
                    f32x4* ps = reinterpret_cast<f32x4*>(ptab + (int)(kfill & (kSlots - 1)) * 32);
                    ps[0] = f32x4{x[0], x[1], x[2], cmx[0]};
                    ps[1] = f32x4{cmx[1], cmx[2], nr[0], nr[1]};
                    ps[2] = f32x4{nr[2], zv[0], zv[1], zv[2]};
                    ps[3] = f32x4{zv[3], zv[4], zv[5], zv[6]};
                    ps[4] = f32x4{zv[7], zv[8], zv[9], zv[10]};
                    ps[5] = f32x4{zv[11], zv[12], zv[13], zv[14]};
                    ps[6] = f32x4{zv[15], zv[16], 0.f, 0.f};
                }
            }
#endif
            if constexpr (kPark) {
                float cm[3], prot[9], pvl[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) cm[k] = a.cam[pt * 3 + k];
                brdf_point_frame(x, cm, nr, prot, pvl);
                const float* zp = a.z + pt * a.z_dim;
                float zv[17];
#pragma unroll
                for (int i = 0; i < 17; ++i) zv[i] = i < a.z_dim ? zp[i] : 0.0f;
                if (lane == 0) {
                    f32x4* ps = reinterpret_cast<f32x4*>(ptab + (int)(kfill & (kSlots - 1)) * 32);
                    ps[0] = f32x4{x[0], x[1], x[2], pvl[0]};
                    ps[1] = f32x4{pvl[1], pvl[2], prot[0], prot[1]};
                    ps[2] = f32x4{prot[2], prot[3], prot[4], prot[5]};
                    ps[3] = f32x4{prot[6], prot[7], prot[8], zv[0]};
                    ps[4] = f32x4{zv[1], zv[3], zv[5], zv[7]};
                    ps[5] = f32x4{zv[9], zv[11], zv[13], zv[15]};
                    ps[6] = f32x4{zv[2], zv[4], zv[6], zv[8]};
                    ps[7] = f32x4{zv[10], zv[12], zv[14], zv[16]};
                }
            }
            // the light of the NEXT 64-group is read while this one is classified (a lone wave has nothing else to
            // put under an LDS round trip)
            float lpn[3];
            {
                const int lc = lane < L ? lane : L - 1;
                lpn[0] = lx[lc * 3]; lpn[1] = lx[lc * 3 + 1]; lpn[2] = lx[lc * 3 + 2];
            }
            for (int l0 = 0; l0 < L; l0 += 64) {
                const int l = l0 + lane;
                const bool valid = l < L;
                const float lp[3] = {lpn[0], lpn[1], lpn[2]};
                if (l0 + 64 < L) {
                    const int ln = l + 64 < L ? l + 64 : L - 1;
                    lpn[0] = lx[ln * 3]; lpn[1] = lx[ln * 3 + 1]; lpn[2] = lx[ln * 3 + 2];
                }
                // (classifying by the sign of n . (light - x) instead — the normalisations are positive scalings — measured
                // 1 % and flips 1 row in 10^8 where the product rounds to 0: not taken)
                float ldir[3], ll[3];
                dir_to(lp, x, ldir);
                mat3_apply(rot, ldir, ll);
                const bool fr = valid && ll[2] > 0.0f;                    // nerfactor.py:429-432
                const unsigned long long mask = __ballot(fr);
                const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                          __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                if (fr) ring[ring_wrap(head + cnt + pos, kCap)] = (ring_t)(((kfill & (kSlots - 1)) << 10) | l);
                else if (valid) a.out[pt * L + l] = 0.0f;                 // scatter_nd's zeros
                cnt += __popcll(mask);
            }
            ++kfill;
        }
        if (cnt == 0) break;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int rows = cnt < kPass ? cnt : kPass;
        // ---- pass: CT column tiles of 32 queued rows
        bf16x8 pl[2][CT];
        long long orow[CT];
        long long rpt[CT];
        int rl[CT], rslot[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int r = c * 32 + p;
            const bool ok = r < rows;
            const unsigned e = ring[ring_wrap(head + (ok ? r : 0), kCap)];
            rslot[c] = (int)(e >> 10);
            const long long kk = (kfill - 1) - (((kfill - 1) - (long long)(e >> 10)) & (kSlots - 1));   // slot -> local point index
            rpt[c] = gw + kk * nw;
            rl[c] = (int)(e & 1023u);
            orow[c] = ok ? rpt[c] * L + rl[c] : -1;
        }
        if constexpr (GEO == 1 && CT % 2 == 0) {
            // lane half h runs the geometry of the row of column tile 2k + h; one v_permlane32_swap per input slot
            // hands each half its own values of both rows (brdf_row_angles; the per-point half comes from the table)
#pragma unroll
            for (int k2 = 0; k2 < CT; k2 += 2) {
                const int l = h ? rl[k2 + 1] : rl[k2];
                const f32x4* ps = reinterpret_cast<const f32x4*>(ptab + (h ? rslot[k2 + 1] : rslot[k2]) * 32);
                const f32x4 q0 = ps[0], q1 = ps[1], q2 = ps[2], q3 = ps[3];
                const float x[3] = {q0[0], q0[1], q0[2]}, vl[3] = {q0[3], q1[0], q1[1]};
                const float rot[9] = {q1[2], q1[3], q2[0], q2[1], q2[2], q2[3], q3[0], q3[1], q3[2]};
                const float lp[3] = {lx[l * 3], lx[l * 3 + 1], lx[l * 3 + 2]};
#ifdef NFX_XP_TRANS_LOAD      // round-6 experiment: the healthy two-wave kernels under a partner that issues MANY quarter-rate
                              // transcendental instructions (what the failing <2, 0, 8> has five times more of): dead work, same outputs
                {
                    float tl = lp[0];
#pragma unroll
                    for (int i = 0; i < NFX_XP_TRANS_LOAD; ++i) asm volatile(NFX_XP_LOAD_INSN : "+v"(tl));
                    asm volatile("" ::"v"(tl));
                }
#endif
                float S[8], C[8];
                brdf_row_angles(x, lp, rot, vl, S, C);
                float v[2][16];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    // S = [half 0: S(row 2k) ; half 1: S(row 2k+1)], C likewise; swap(S.hi, C.lo):
                    //   first  = [S(row 2k) ; C(row 2k)]     = slot q of column tile 2k for both halves
                    //   second = [S(row 2k+1) ; C(row 2k+1)] = slot q of column tile 2k + 1
                    float first = S[q], second = C[q];
#if NFX_BRDF_SWAP == 1      // the exchange through ds_bpermute (LDS crossbar, counted by lgkmcnt)
                    const float recv = __shfl_xor(h ? first : second, 32, 64);
                    first = h ? recv : first;
                    second = h ? second : recv;
#elif NFX_BRDF_SWAP == 2    // rounds 1-2: hand-timed wait states — WRONG with a partner wave on the SIMD (see the kernel's header)
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(first), "+v"(second));
#else                       // the compiler's own v_permlane32_swap (its hazard recogniser places the wait states)
                    {
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(first), __float_as_uint(second), false, false);
                        first = __uint_as_float(sw[0]);
                        second = __uint_as_float(sw[1]);
                    }
#endif
                    v[0][q] = first;
                    v[1][q] = second;
                }
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    // the latent code of the row's point: z_0 (slot 7 of half 1) and this half's z_{1 + 2j + h}
                    const f32x4* pz = reinterpret_cast<const f32x4*>(ptab + rslot[k2 + cc] * 32);
#ifdef NFX_XP_USE_V255
                    // round-6 experiment: the healthy <2, 1, 8> made to keep a live value in v255 (its allocation stops at v246)
                    int hs;
                    asm volatile("v_mov_b32 v255, %1\n\ts_nop 4\n\tv_mov_b32 %0, v255" : "=v"(hs) : "v"(h) : "v255");
#elif defined(NFX_XP_FORCE_SCRATCH)
                    const int hs = xp_slot[0];          // h, re-loaded from scratch memory
                    xp_slot[2] = xp_slot[2] + cc;       // and a slot that is re-stored in the loop, like the spilled constant pair
#else
                    const int hs = h;
#endif
                    const f32x4 za = pz[4 + 2 * hs], zb = pz[5 + 2 * hs];
                    if (hs) v[cc][7] = pz[3][3];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[cc][8 + j] = za[j];
                        v[cc][12 + j] = zb[j];
                    }
#pragma unroll
                    for (int sidx = 0; sidx < 2; ++sidx) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) pl[sidx][k2 + cc][j] = (__bf16)v[cc][8 * sidx + j];
                        mfma_operand_fence(pl[sidx][k2 + cc]);
                    }
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const long long pt = rpt[c];
                const int l = rl[c];
                float x[3], lp[3], cm[3], nr[3];
#ifdef NFX_XP_LDS_INPUTS
                const float* row = ptab + rslot[c] * 32;
                const float* zsrc = row + 9;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    x[k] = row[k];
                    lp[k] = lx[l * 3 + k];
                    cm[k] = row[3 + k];
                    nr[k] = row[6 + k];
                }
#else
                const float* zsrc = a.z + pt * a.z_dim;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    x[k] = a.xyz[pt * 3 + k];
                    lp[k] = lx[l * 3 + k];
                    cm[k] = a.cam[pt * 3 + k];
                    nr[k] = a.normal[pt * 3 + k];
                }
#endif
                float v[16];
#ifdef NFX_XP_NOSCRATCH
                // round-6 experiment: <2, 0, 8> without its scratch spill — the lane half is re-derived where it is used
                // (a volatile asm is neither hoisted nor kept live across the pass), so nothing has to be spilled for it
                int hx;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshrrev_b32 %0, 5, %0" : "=v"(hx));
                brdf_row_inputs<GEO>(x, lp, cm, nr, zsrc, a.z_dim, hx, v);
#else
                brdf_row_inputs<GEO>(x, lp, cm, nr, zsrc, a.z_dim, h, v);
#endif
#pragma unroll
                for (int sidx = 0; sidx < 2; ++sidx) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) pl[sidx][c][j] = (__bf16)v[8 * sidx + j];
                    mfma_operand_fence(pl[sidx][c]);
                }
            }
        }
#ifdef NFX_XP_CLOBBER_TOP     // round-6 experiment: nothing of this wave lives in v248 .. v255 across this point
        asm volatile("" ::: "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");
#endif
        bf16x8 ha[8][CT], hb[8][CT];
        Acc<CT> accs[2];
        Pre pre;
        {
            const char* f0 = wlds + lane * 16;
            pre.a[0] = *reinterpret_cast<const bf16x8*>(f0);
            pre.a[1] = *reinterpret_cast<const bf16x8*>(f0 + kFragBytes);
            InitBias{bias_lds}(lane, accs[0]);
        }
#define NFX_LV3_TILE(K, KS1, KS2, B1, B2, PREV, NEXT) \
        tile<K, (K + 1) % 17, KS1, KS2, CT>(wlds, lane, B1, B2, accs[(K) & 1], accs[((K) + 1) & 1], pre, PREV, NEXT)
#define NFX_LV3_EPI(K, OUT, T) EpiB<CT>{accs[(K) & 1], OUT[2 * (T)], OUT[2 * (T) + 1]}
#define NFX_LV3_BIAS(OFF) (InitBias{bias_lds + (OFF)})
        NFX_LV3_TILE(0, 2, 0, pl, pl, EpiNone{}, NFX_LV3_BIAS(32));
        NFX_LV3_TILE(1, 2, 0, pl, pl, NFX_LV3_EPI(0, ha, 0), NFX_LV3_BIAS(64));
        NFX_LV3_TILE(2, 2, 0, pl, pl, NFX_LV3_EPI(1, ha, 1), NFX_LV3_BIAS(96));
        NFX_LV3_TILE(3, 2, 0, pl, pl, NFX_LV3_EPI(2, ha, 2), NFX_LV3_BIAS(128));
        NFX_LV3_TILE(4, 8, 0, ha, pl, NFX_LV3_EPI(3, ha, 3), NFX_LV3_BIAS(128 + 32));
        NFX_LV3_TILE(5, 8, 0, ha, pl, NFX_LV3_EPI(4, hb, 0), NFX_LV3_BIAS(128 + 64));
        NFX_LV3_TILE(6, 8, 0, ha, pl, NFX_LV3_EPI(5, hb, 1), NFX_LV3_BIAS(128 + 96));
        NFX_LV3_TILE(7, 8, 0, ha, pl, NFX_LV3_EPI(6, hb, 2), NFX_LV3_BIAS(256));
#ifdef NFX_XP_CLOBBER_TOP     // round-6 experiment: nothing of this wave lives in v248 .. v255 across this point
        asm volatile("" ::: "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");
#endif
        NFX_LV3_TILE(8, 8, 0, hb, pl, NFX_LV3_EPI(7, hb, 3), NFX_LV3_BIAS(256 + 32));
        NFX_LV3_TILE(9, 8, 0, hb, pl, NFX_LV3_EPI(8, ha, 0), NFX_LV3_BIAS(256 + 64));
        NFX_LV3_TILE(10, 8, 0, hb, pl, NFX_LV3_EPI(9, ha, 1), NFX_LV3_BIAS(256 + 96));
        NFX_LV3_TILE(11, 8, 0, hb, pl, NFX_LV3_EPI(10, ha, 2), NFX_LV3_BIAS(384));
        NFX_LV3_TILE(12, 8, 2, ha, pl, NFX_LV3_EPI(11, ha, 3), NFX_LV3_BIAS(384 + 32));
        NFX_LV3_TILE(13, 8, 2, ha, pl, NFX_LV3_EPI(12, hb, 0), NFX_LV3_BIAS(384 + 64));
        NFX_LV3_TILE(14, 8, 2, ha, pl, NFX_LV3_EPI(13, hb, 1), NFX_LV3_BIAS(384 + 96));
        NFX_LV3_TILE(15, 8, 2, ha, pl, NFX_LV3_EPI(14, hb, 2), NFX_LV3_BIAS(512));
        NFX_LV3_TILE(16, 8, 0, hb, pl, NFX_LV3_EPI(15, hb, 3), [](int, Acc<CT>&) {});
#undef NFX_LV3_TILE
#undef NFX_LV3_EPI
#undef NFX_LV3_BIAS
        if (h == 0) {
#pragma unroll
            for (int c = 0; c < CT; ++c)
#ifdef NFX_XP_NOSCRATCH      // (the libm log1pf keeps a constant pair live that the allocator spills; hardware exp / log: all CT / NW forms alike)
                if (orow[c] >= 0) a.out[orow[c]] = fmaxf(accs[0].v[c][0], 0.0f) + __logf(1.0f + __expf(-fabsf(accs[0].v[c][0])));
#else
                if (orow[c] >= 0) a.out[orow[c]] = softplusf(accs[0].v[c][0]);   // brdf.py:65
#endif
        }
#ifdef NFX_XP_CLOBBER_TOP     // round-6 experiment: nothing of this wave lives in v248 .. v255 across this point
        asm volatile("" ::: "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");
#endif
        head = ring_wrap(head + rows, kCap);
        cnt -= rows;
        if (cnt > 0) {
            const unsigned e = ring[head];
            k_head = (kfill - 1) - (((kfill - 1) - (long long)(e >> 10)) & (kSlots - 1));
        } else {
            k_head = kfill;
        }
    }
}

}  // namespace lv2
}  // namespace nfx

template <int CT, int MODE, int NW = 4, bool ROWS = false>
static int launch_res(const nfx::lv2::Args& a, int max_blocks, hipStream_t st) {
    using namespace nfx;
    const long long rows = a.n * a.n_lights, tile_rows = NW * CT * 32;
    const long long tiles = (rows + tile_rows - 1) / tile_rows;
    const int grid = (int)(tiles < max_blocks ? tiles : max_blocks);
    constexpr int lds = lv2::kLdsNet + NW * CT * 1024;
    static_assert(lds <= 160 * 1024, "LDS");
    auto k = lv2::resident128_kernel<CT, MODE, NW, ROWS>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds, st, a);
    return (int)hipGetLastError();
}

extern "C" int nfx_launch_lvis_v2(const float* xyz, long long n, const float* lxyz, int n_lights, const float* pre,
                                  const void* blob_main, float* lvis, int ct, int max_blocks, hipStream_t st,
                                  const int* out_row, int* nan_flag) {
    if (n <= 0) return 0;
    nfx::lv2::Args a{xyz, lxyz, pre, nullptr, nullptr, nullptr, 0, n, n_lights, (const char*)blob_main, lvis, out_row, nan_flag};
    if (out_row != nullptr) {     // (a NaN flag alone: the caller passes the identity rows — capi_nerfactor.cpp refuses it)
        if (ct == 8) return launch_res<2, 0, 8, true>(a, max_blocks, st);
        return launch_res<4, 0, 4, true>(a, max_blocks, st);     // variants 2 | 3 | 4: the one-wave-per-SIMD reference form
    }
    if (ct == 8) return launch_res<2, 0, 8>(a, max_blocks, st);   // variant 8: 8 waves x 2 column tiles
    if (ct == 2) return launch_res<2, 0>(a, max_blocks, st);
    if (ct == 3) return launch_res<3, 0>(a, max_blocks, st);
    return launch_res<4, 0>(a, max_blocks, st);
}

extern "C" int nfx_launch_brdf_spec_v2(const float* xyz, const float* cam, const float* normal, const float* z,
                                       int z_dim, const float* lxyz, int n_lights, const void* blob, long long n,
                                       float* spec, int ct, int max_blocks, hipStream_t st) {
    if (n <= 0) return 0;
    nfx::lv2::Args a{xyz, lxyz, nullptr, cam, normal, z, z_dim, n, n_lights, (const char*)blob, spec, nullptr, nullptr};
    if (ct == 2) return launch_res<2, 1>(a, max_blocks, st);
    if (ct == 3) return launch_res<3, 1>(a, max_blocks, st);
    return launch_res<4, 1>(a, max_blocks, st);
}

template <int CT, int GEO, int NW>
static int launch_compact(const nfx::lv2::Args& a, int max_blocks, hipStream_t st) {
    using namespace nfx;
    const long long want = (a.n + NW - 1) / NW;       // at least one point per wave
    const int grid = (int)(want < max_blocks ? want : max_blocks);
    const int lds = lv2::Queue<NW>::lds_bytes(a.n_lights);   // network + row rings + lights + per-point tables
    if (CT * 32 - 1 + a.n_lights > lv2::Queue<NW>::kCap) return -1;
    if (lds > 160 * 1024) return -1;   // (the caller falls back to the dense kernel)
    auto k = lv2::brdf_compact_kernel<CT, GEO, NW>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds, st, a);
    return (int)hipGetLastError();
}

// Front-lit compaction (brdf_compact_kernel).  geo: 0 = reference op sequence per row, 1 = closed-form Rusinkiewicz.
// ct: 8 = 8 waves x 2 column tiles (two waves per SIMD); 2 | 3 | 4 column tiles per wave, one wave per SIMD (anything else: 4).
// Returns -1 when the shape does not fit the row queue (the caller falls back to the dense kernel).
extern "C" int nfx_launch_brdf_spec_v3(const float* xyz, const float* cam, const float* normal, const float* z,
                                       int z_dim, const float* lxyz, int n_lights, const void* blob, long long n,
                                       float* spec, int ct, int geo, int max_blocks, hipStream_t st) {
    if (n <= 0) return 0;
    const int tiles = ct == 8 ? 2 : ct;
    if (n_lights > 1024 || tiles * 32 - 1 + n_lights > nfx::lv2::kRing) return -1;   // (launch_compact checks its own ring)
    nfx::lv2::Args a{xyz, lxyz, nullptr, cam, normal, z, z_dim, n, n_lights, (const char*)blob, spec};
#ifdef NFX_EXPERIMENT_BUILD   // the per-row-geometry form with two waves per SIMD: NOT deterministic on MI355X (profiles/HISTORY.md section 2c), soak builds only
    if (ct == 8 && !geo) return launch_compact<2, 0, 8>(a, max_blocks, st);
#endif
    if (ct == 8) return geo ? launch_compact<2, 1, 8>(a, max_blocks, st) : launch_compact<2, 0, 4>(a, max_blocks, st);
    if (ct == 2) return geo ? launch_compact<2, 1, 4>(a, max_blocks, st) : launch_compact<2, 0, 4>(a, max_blocks, st);
    if (ct == 3) return geo ? launch_compact<3, 1, 4>(a, max_blocks, st) : launch_compact<3, 0, 4>(a, max_blocks, st);
    return geo ? launch_compact<4, 1, 4>(a, max_blocks, st) : launch_compact<4, 0, 4>(a, max_blocks, st);
}

