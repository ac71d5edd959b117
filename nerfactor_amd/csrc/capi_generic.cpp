// capi_generic.cpp — C-ABI of the runtime-shaped MLP (mlp_generic.hip): host packer, forward, Embedder.
#include <hip/hip_runtime.h>
#include <string.h>

#include "../../include/nfx.h"
#include "mlp_generic.hpp"
#include "pack.hpp"

int nfx_fail(int code, const char* fmt, ...);                    // capi.cpp
int nfx_hip_result(int e, const char* what);                     // capi.cpp
extern "C" int nfx_option_int(const char* key, int dflt);        // capi.cpp
#define REQUIRE(cond, ...) \
    do {                   \
        if (!(cond)) return nfx_fail(NFX_EINVAL, __VA_ARGS__); \
    } while (0)

extern "C" {
int nfx_launch_mlp_generic(const nfx::generic::Args* args, int max_blocks, hipStream_t st);
int nfx_launch_embed(const nfx::generic::EmbedArgs* a, hipStream_t st);
int nfx_launch_embed_bwd(const nfx::generic::EmbedArgs* a, const float* d_out, float* dv, hipStream_t st);
int nfx_launch_mlp_generic_bwd(const nfx::generic::BwdArgs* ba, const nfx::generic::WgradArgs* wa, int max_blocks, hipStream_t st);
int nfx_launch_split_hilo(void* frags, long long n_frags, hipStream_t st);
}

// the layer table of a network: mlp.Network(widths, skip_at) semantics (nerfactor/networks/mlp.py:38-50) —
// skip_input[i] != 0 <=> layer i reads concat(output of layer i - 1, network input), i.e. i - 1 is in skip_at
static int layer_table(int d_in, int n_layers, const int* widths, const int* skip_input, const int* acts,
                       nfx::generic::Layer* out, int* n_frags, int* n_bias) {
    using namespace nfx::generic;
    if (d_in < 1 || d_in > kMaxIn) return nfx_fail(NFX_ENOSUP, "generic MLP: network input of %d features (1 .. %d)", d_in, kMaxIn);
    if (n_layers < 1 || n_layers > kMaxLayers) return nfx_fail(NFX_ENOSUP, "generic MLP: %d layers (1 .. %d)", n_layers, kMaxLayers);
    int w = 0, b = 0;
    for (int i = 0; i < n_layers; ++i) {
        if (widths[i] < 1 || widths[i] > kMaxHidden)
            return nfx_fail(NFX_ENOSUP, "generic MLP: layer %d has %d units (1 .. %d)", i, widths[i], kMaxHidden);
        Layer& L = out[i];
        // both operand sources padded to whole groups: a group never mixes the previous layer's output with the input
        L.ks_h = i == 0 ? 0 : nfx::generic::pad_group((widths[i - 1] + 15) / 16);
        L.ks_x = (i == 0 || (skip_input && skip_input[i])) ? nfx::generic::pad_group((d_in + 15) / 16) : 0;
        L.ks_pad = L.ks_h + L.ks_x;
        L.n_tiles = (widths[i] + 31) / 32;
        L.n_out = widths[i];
        L.act = acts ? acts[i] : 0;
        L.w_off = w;
        L.b_off = b;
        w += L.n_tiles * L.ks_pad;
        b += L.n_tiles * 32;
    }
    *n_frags = w;
    *n_bias = b;
    return NFX_OK;
}
static void set_pitches(nfx::generic::Args* a) {
    int widest = 1;
    for (int i = 0; i < a->n_layers; ++i) widest = a->layer[i].n_tiles > widest ? a->layer[i].n_tiles : widest;
    const int elem = a->f32 ? 4 : 2;                         // (f32 = the prec: 0 bf16, else fp32 activations)
    a->x_pitch = (a->d_in + 63) / 64 * 64 * elem + 16;       // whole k-groups (64 features) per row
    a->h_pitch = (widest + 1) / 2 * 64 * elem + 16;
}
static int check_prec(int prec, const char* who) {
    if (prec != NFX_PREC_BF16 && prec != NFX_PREC_FP32 && prec != NFX_PREC_FP32_NATIVE)
        return nfx_fail(NFX_EINVAL, "%s: prec %d (NFX_PREC_BF16 | NFX_PREC_FP32 | NFX_PREC_FP32_NATIVE)", who, prec);
    return NFX_OK;
}
static_assert(NFX_PREC_BF16 == 0 && NFX_PREC_FP32 == 1 && NFX_PREC_FP32_NATIVE == 2, "= nfx::generic::kBf16 / kX3 / kNative (mlp_generic.hip)");
static bool wide(int prec) { return prec != NFX_PREC_BF16; }      // fp32 activations / workspace, 2-KiB fragments
static size_t frag_bytes(int prec) { return wide(prec) ? 2048 : 1024; }
static float bf16_to_f32(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// element (lane, i) of a fragment: bf16: [lane][8]; fp32 native: [half = i / 4][lane][4] (one DMA piece per half);
// fp32-class: [hi plane | lo plane], each [lane][8] bf16 — hi = bf16(v), lo = bf16(v - hi) (csrc/mlp_x3.hpp)
static void put(void* frag, int prec, int lane, int i, float v) {
    if (prec == NFX_PREC_FP32_NATIVE) static_cast<float*>(frag)[(i >> 2) * 256 + lane * 4 + (i & 3)] = v;
    else if (prec == NFX_PREC_FP32) {
        const uint16_t hi = nfx::pack::f32_to_bf16_rne(v);
        static_cast<uint16_t*>(frag)[lane * 8 + i] = hi;
        static_cast<uint16_t*>(frag)[512 + lane * 8 + i] = nfx::pack::f32_to_bf16_rne(v - bf16_to_f32(hi));
    } else static_cast<uint16_t*>(frag)[lane * 8 + i] = nfx::pack::f32_to_bf16_rne(v);
}

size_t nfx_mlp_generic_packed_bytes(int d_in, int n_layers, const int* widths, const int* skip_input, int prec) {
    nfx::generic::Layer t[nfx::generic::kMaxLayers];
    int nf, nb;
    if (!widths || check_prec(prec, "nfx_mlp_generic_packed_bytes") || layer_table(d_in, n_layers, widths, skip_input, nullptr, t, &nf, &nb)) return 0;
    return (size_t)nf * frag_bytes(prec) + (size_t)nb * 4;
}

int nfx_mlp_generic_pack(const float* const* kernels, const float* const* biases, int d_in, int n_layers, const int* widths,
                         const int* skip_input, int prec, void* blob, size_t blob_bytes) {
    REQUIRE(kernels && biases && widths && blob, "nfx_mlp_generic_pack: null argument");
    if (int e = check_prec(prec, "nfx_mlp_generic_pack")) return e;
    const size_t fb = frag_bytes(prec);
    nfx::generic::Layer t[nfx::generic::kMaxLayers];
    int nf, nb;
    int rc = layer_table(d_in, n_layers, widths, skip_input, nullptr, t, &nf, &nb);
    if (rc) return rc;
    const size_t need = (size_t)nf * fb + (size_t)nb * 4;
    REQUIRE(blob_bytes >= need, "nfx_mlp_generic_pack: blob too small (%zu < %zu)", blob_bytes, need);
    memset(blob, 0, need);
    float* b = static_cast<float*>(blob);                                           // [biases | fragments]
    char* w = static_cast<char*>(blob) + (size_t)nb * 4;
    for (int i = 0; i < n_layers; ++i) {
        REQUIRE(kernels[i] && biases[i], "nfx_mlp_generic_pack: layer %d null", i);
        const nfx::generic::Layer& L = t[i];
        const int prev = i == 0 ? 0 : widths[i - 1];          // rows [0, prev): the previous layer's output
        const int n_in = prev + (L.ks_x ? d_in : 0);          // then the network input (mlp.py:48: y first)
        (void)n_in;
        for (int tl = 0; tl < L.n_tiles; ++tl)
            for (int s = 0; s < L.ks_h + L.ks_x; ++s) {
                // stream order of a layer: k-GROUP outer, output tile inner (the kernel reads a group's B operand once
                // and sweeps the tiles' accumulators): group (s / kGroup) of tile tl, step s % kGroup
                const int kg = s / nfx::generic::kGroup, j4 = s % nfx::generic::kGroup;
                char* frag = w + ((size_t)L.w_off + ((size_t)kg * L.n_tiles + tl) * nfx::generic::kGroup + j4) * fb;
                const bool from_x = s >= L.ks_h;
                const int base = from_x ? prev : 0, feat0 = 16 * (from_x ? s - L.ks_h : s), limit = from_x ? d_in : prev;
                for (int lane = 0; lane < 64; ++lane) {
                    const int m = lane & 31, g = lane >> 5, col = 32 * tl + m;
                    if (col >= L.n_out) continue;
                    for (int j = 0; j < 8; ++j) {
                        const int f = feat0 + 8 * g + j;
                        if (f >= limit) continue;
                        put(frag, prec, lane, j, kernels[i][(size_t)(base + f) * L.n_out + col]);
                    }
                }
            }
        for (int c = 0; c < L.n_out; ++c) b[L.b_off + c] = biases[i][c];
    }
    return NFX_OK;
}

int nfx_mlp_generic_fwd(const float* x, int64_t n, int ld_x, int d_in, int n_layers, const int* widths, const int* acts,
                        const int* skip_input, const void* blob, int prec, float* y, int ld_y, int col0, void* stream) {
    REQUIRE(n >= 0, "nfx_mlp_generic_fwd: n < 0");
    if (int e = check_prec(prec, "nfx_mlp_generic_fwd")) return e;
    REQUIRE(widths && acts, "nfx_mlp_generic_fwd: null layer description");
    nfx::generic::Args a;
    int nf, nb;
    int rc = layer_table(d_in, n_layers, widths, skip_input, acts, a.layer, &nf, &nb);
    if (rc) return rc;
    for (int i = 0; i < n_layers; ++i) REQUIRE(acts[i] >= 0 && acts[i] <= 3, "nfx_mlp_generic_fwd: activation %d of layer %d", acts[i], i);
    if (n == 0) return NFX_OK;
    REQUIRE(x && blob && y, "nfx_mlp_generic_fwd: null pointer");
    REQUIRE(ld_x >= d_in && ld_y >= col0 + widths[n_layers - 1] && col0 >= 0, "nfx_mlp_generic_fwd: bad leading dimensions");
    if ((uintptr_t)blob & 15) return nfx_fail(NFX_EALIGN, "nfx_mlp_generic_fwd: blob must be 16-byte aligned");
    a.x = x;
    a.n = n;
    a.ld_x = ld_x;
    a.d_in = d_in;
    a.biases = static_cast<const float*>(blob);
    a.weights = static_cast<const char*>(blob) + (size_t)nb * 4;
    a.y = y;
    a.ld_y = ld_y;
    a.col0 = col0;
    a.n_layers = n_layers;
    a.n_frags = nf;
    a.f32 = prec;
    set_pitches(&a);
    rc = nfx_launch_mlp_generic(&a, 8 * nfx_option_int("nerf_blocks", 256), (hipStream_t)stream);
    if (rc < 0) return nfx_fail(NFX_ENOSUP, "nfx_mlp_generic_fwd: this shape needs %d bytes of LDS per wave (160 KiB per CU): narrower layers or a smaller input", -rc);
    return nfx_hip_result(rc, "mlp_generic_fwd");
}

// ---- backward ------------------------------------------------------------------------------------------------------
// train blob = [biases][forward fragments][transposed fragments]: its head IS the forward blob, the fragments one stream
namespace {
struct BwdPlan {
    nfx::generic::Layer layer[nfx::generic::kMaxLayers];
    nfx::generic::BwdLayer b[nfx::generic::kMaxLayers];
    int n_frags, n_bias, n_tfrags, feat_rows, n_jobs;
    long long slice, dw_total;
};
int bwd_plan(int d_in, int n_layers, const int* widths, const int* skip_input, const int* acts, BwdPlan* p) {
    int rc = layer_table(d_in, n_layers, widths, skip_input, acts, p->layer, &p->n_frags, &p->n_bias);
    if (rc) return rc;
    const int mx = (d_in + 31) / 32;
    int wt = 0, rows = 32 * mx, jobs = 0;
    long long dw = 0;
    for (int i = 0; i + 1 < n_layers; ++i) {
        p->b[i].h_row = rows;
        rows += 32 * p->layer[i].n_tiles;
    }
    p->b[n_layers - 1].h_row = -1;
    for (int i = 0; i < n_layers; ++i) {
        const nfx::generic::Layer& L = p->layer[i];
        const int mh = i ? p->layer[i - 1].n_tiles : 0, m_in = mh + (L.ks_x ? mx : 0);
        p->b[i].dz_row = rows;
        p->b[i].dw_off = (int)dw;
        p->b[i].job0 = jobs;
        rows += 32 * L.n_tiles;
        jobs += ((m_in + 1) / 2) * ((L.n_tiles + 1) / 2);        // 64 x 64 blocks of dW
        dw += (long long)((i ? widths[i - 1] : 0) + (L.ks_x ? d_in : 0)) * L.n_out;
    }
    for (int i = n_layers - 1; i >= 0; --i) {                     // the transposed stream: last layer first, as the backward walks
        const nfx::generic::Layer& L = p->layer[i];
        const int mh = i ? p->layer[i - 1].n_tiles : 0, m_in = mh + (L.ks_x ? mx : 0);
        p->b[i].wt_off = wt;
        wt += m_in * nfx::generic::pad_group(2 * L.n_tiles);
    }
    p->n_tfrags = wt;
    p->feat_rows = rows;
    p->n_jobs = jobs;
    p->dw_total = dw;
    for (int i = 0; i < n_layers; ++i) {
        p->b[i].db_off = (int)dw;
        dw += p->layer[i].n_out;
    }
    p->slice = dw;
    return NFX_OK;
}
// row splits of the weight-gradient contraction: enough waves to fill the chip, a function of the problem shape only
// (cap 256, round 5: the kernel waits on its operand loads at one to two waves per SIMD; 64 -> 256 splits of the
// light-visibility network's 22 jobs: 2.32 -> 2.15 ms per 524 288 rows fp32-class, 5.08 -> 4.76 ms per 2 097 152 rows bf16)
int wgrad_splits(long long tiles, int n_jobs) {
    long long s = (8192 + n_jobs - 1) / n_jobs;
    if (s > tiles / 4) s = tiles / 4;
    const int cap = nfx_option_int("wgrad_splits", 256);
    if (s > cap) s = cap;
    return s < 1 ? 1 : (int)s;
}
size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
}  // namespace

size_t nfx_mlp_generic_train_packed_bytes(int d_in, int n_layers, const int* widths, const int* skip_input, int prec) {
    BwdPlan p;
    if (!widths || check_prec(prec, "nfx_mlp_generic_train_packed_bytes") || bwd_plan(d_in, n_layers, widths, skip_input, nullptr, &p)) return 0;
    return (size_t)(p.n_frags + p.n_tfrags) * frag_bytes(prec) + (size_t)p.n_bias * 4;
}

int nfx_mlp_generic_pack_train(const float* const* kernels, const float* const* biases, int d_in, int n_layers,
                               const int* widths, const int* skip_input, int prec, void* blob, size_t blob_bytes) {
    REQUIRE(kernels && biases && widths && blob, "nfx_mlp_generic_pack_train: null argument");
    if (int e = check_prec(prec, "nfx_mlp_generic_pack_train")) return e;
    const size_t fb = frag_bytes(prec);
    BwdPlan p;
    int rc = bwd_plan(d_in, n_layers, widths, skip_input, nullptr, &p);
    if (rc) return rc;
    const size_t head = (size_t)p.n_frags * fb + (size_t)p.n_bias * 4, need = head + (size_t)p.n_tfrags * fb;
    REQUIRE(blob_bytes >= need, "nfx_mlp_generic_pack_train: blob too small (%zu < %zu)", blob_bytes, need);
    rc = nfx_mlp_generic_pack(kernels, biases, d_in, n_layers, widths, skip_input, prec, blob, head);
    if (rc) return rc;
    char* wt = static_cast<char*>(blob) + head;
    memset(wt, 0, (size_t)p.n_tfrags * fb);
    const int mx = (d_in + 31) / 32;
    for (int i = 0; i < n_layers; ++i) {
        const nfx::generic::Layer& L = p.layer[i];
        const int prev = i ? widths[i - 1] : 0, mh = i ? p.layer[i - 1].n_tiles : 0, m_in = mh + (L.ks_x ? mx : 0), ks_o = 2 * L.n_tiles;
        const int ks_o_pad = nfx::generic::pad_group(ks_o);
        // a layer's transposed fragments: the input-gradient tiles FIRST (tile-major: they read dZ before the hidden
        // tiles overwrite it in place), then the hidden tiles k-group outer / tile inner like the forward
        const int nx = m_in - mh, kgo = ks_o_pad / nfx::generic::kGroup;
        (void)kgo;
        for (int mt = 0; mt < m_in; ++mt)
            for (int s = 0; s < ks_o; ++s) {
                const bool from_x = mt >= mh;
                const int kg = s / nfx::generic::kGroup, j4 = s % nfx::generic::kGroup;
                const size_t idx = from_x ? (size_t)(mt - mh) * ks_o_pad + s
                                          : (size_t)nx * ks_o_pad + ((size_t)kg * mh + mt) * nfx::generic::kGroup + j4;
                char* frag = wt + ((size_t)p.b[i].wt_off + idx) * fb;
                const int base = from_x ? prev : 0, feat0 = 32 * (from_x ? mt - mh : mt), limit = from_x ? d_in : prev;
                for (int lane = 0; lane < 64; ++lane) {
                    const int f = feat0 + (lane & 31), g = lane >> 5;       // A row = input feature, k = output feature
                    if (f >= limit) continue;
                    for (int j = 0; j < 8; ++j) {
                        const int o = 16 * s + 8 * g + j;
                        if (o < L.n_out) put(frag, prec, lane, j, kernels[i][(size_t)(base + f) * L.n_out + o]);
                    }
                }
            }
    }
    return NFX_OK;
}

size_t nfx_mlp_generic_bwd_workspace_bytes(int64_t n, int d_in, int n_layers, const int* widths, const int* skip_input, int prec) {
    BwdPlan p;
    if (n < 0 || !widths || check_prec(prec, "nfx_mlp_generic_bwd_workspace_bytes") || bwd_plan(d_in, n_layers, widths, skip_input, nullptr, &p)) return 0;
    const long long tiles = (n + 31) / 32;
    // (tiles + 1: a spare tile for the waves of the last workgroup that have no row tile of their own, mlp_generic.hip)
    return align256((size_t)(tiles + 1) * p.feat_rows * (wide(prec) ? 128 : 64)) + (size_t)wgrad_splits(tiles, p.n_jobs) * p.slice * 4 + 256;
}

int nfx_mlp_generic_bwd(const float* x, int64_t n, int ld_x, int d_in, int n_layers, const int* widths, const int* acts,
                        const int* skip_input, const void* train_blob, int prec, const float* dy, int ld_dy, int col0_dy,
                        float* dx, int ld_dx, float* const* dkernels, float* const* dbiases, void* workspace,
                        size_t workspace_bytes, void* stream) {
    REQUIRE(n >= 0, "nfx_mlp_generic_bwd: n < 0");
    if (int e = check_prec(prec, "nfx_mlp_generic_bwd")) return e;
    REQUIRE(widths && acts, "nfx_mlp_generic_bwd: null layer description");
    REQUIRE((dkernels && dbiases) || (!dkernels && !dbiases && dx), "nfx_mlp_generic_bwd: gradient buffers for both kernels and "
            "biases, or neither (then dx is the only result and must not be null)");
    BwdPlan p;
    int rc = bwd_plan(d_in, n_layers, widths, skip_input, acts, &p);
    if (rc) return rc;
    for (int i = 0; i < n_layers; ++i) {
        REQUIRE(acts[i] >= 0 && acts[i] <= 3, "nfx_mlp_generic_bwd: activation %d of layer %d", acts[i], i);
        REQUIRE(!dkernels || (dkernels[i] && dbiases[i]), "nfx_mlp_generic_bwd: gradient buffer of layer %d is null", i);
    }
    if (n == 0) return NFX_OK;
    REQUIRE(x && train_blob && dy && workspace, "nfx_mlp_generic_bwd: null pointer");
    REQUIRE(ld_x >= d_in && ld_dy >= col0_dy + widths[n_layers - 1] && col0_dy >= 0 && (!dx || ld_dx >= d_in),
            "nfx_mlp_generic_bwd: bad leading dimensions");
    if (((uintptr_t)train_blob | (uintptr_t)workspace) & 15) return nfx_fail(NFX_EALIGN, "nfx_mlp_generic_bwd: blob / workspace must be 16-byte aligned");
    const size_t need = nfx_mlp_generic_bwd_workspace_bytes(n, d_in, n_layers, widths, skip_input, prec);
    REQUIRE(workspace_bytes >= need, "nfx_mlp_generic_bwd: workspace too small (%zu < %zu)", workspace_bytes, need);
    nfx::generic::BwdArgs ba;
    nfx::generic::WgradArgs wa;
    memset(&ba, 0, sizeof ba);
    memset(&wa, 0, sizeof wa);
    const long long tiles = (n + 31) / 32;
    ba.f.x = x;
    ba.f.n = n;
    ba.f.ld_x = ld_x;
    ba.f.d_in = d_in;
    ba.f.biases = static_cast<const float*>(train_blob);
    ba.f.weights = static_cast<const char*>(train_blob) + (size_t)p.n_bias * 4;
    ba.f.n_layers = n_layers;
    ba.f.n_frags = p.n_frags;
    // layer 0's transposed tiles (all of them input-gradient tiles) end the stream: not walked when dx is not wanted
    ba.stream_frags = p.n_frags + p.n_tfrags - (dx ? 0 : (d_in + 31) / 32 * nfx::generic::pad_group(2 * p.layer[0].n_tiles));
    ba.dy = dy;
    ba.ld_dy = ld_dy;
    ba.col0_dy = col0_dy;
    ba.dx = dx;
    ba.ld_dx = ld_dx;
    ba.ws = static_cast<char*>(workspace);
    ba.tiles = tiles;
    ba.feat_rows = p.feat_rows;
    wa.ws = ba.ws;
    wa.tiles = tiles;
    wa.feat_rows = p.feat_rows;
    wa.n_layers = n_layers;
    wa.d_in = d_in;
    wa.splits = wgrad_splits(tiles, p.n_jobs);
    wa.n_jobs = p.n_jobs;
    wa.map = nfx_option_int("wgrad_map", 1);
    wa.slice = p.slice;
    wa.dw_total = p.dw_total;
    wa.partial = reinterpret_cast<float*>(ba.ws + align256((size_t)(tiles + 1) * p.feat_rows * (wide(prec) ? 128 : 64)));
    for (int i = 0; i < n_layers; ++i) {
        ba.f.layer[i] = wa.layer[i] = p.layer[i];
        ba.b[i] = wa.b[i] = p.b[i];
        wa.dw[i] = dkernels ? dkernels[i] : nullptr;
        wa.db[i] = dkernels ? dbiases[i] : nullptr;
    }
    ba.f.f32 = prec;
    set_pitches(&ba.f);
    rc = nfx_launch_mlp_generic_bwd(&ba, &wa, 8 * nfx_option_int("nerf_blocks", 256), (hipStream_t)stream);
    if (rc < 0) return nfx_fail(NFX_ENOSUP, "nfx_mlp_generic_bwd: this shape needs %d bytes of LDS per wave (160 KiB per CU): narrower layers or a smaller input", -rc);
    return nfx_hip_result(rc, "mlp_generic_bwd");
}

int nfx_mlp_generic_split_hilo(void* blob, int d_in, int n_layers, const int* widths, const int* skip_input, int train, void* stream) {
    REQUIRE(blob && widths, "nfx_mlp_generic_split_hilo: null argument");
    if ((uintptr_t)blob & 15) return nfx_fail(NFX_EALIGN, "nfx_mlp_generic_split_hilo: blob must be 16-byte aligned");
    BwdPlan p;
    int rc = bwd_plan(d_in, n_layers, widths, skip_input, nullptr, &p);
    if (rc) return rc;
    char* frags = static_cast<char*>(blob) + (size_t)p.n_bias * 4;
    return nfx_hip_result(nfx_launch_split_hilo(frags, p.n_frags + (train ? p.n_tfrags : 0), (hipStream_t)stream), "split_hilo");
}

int nfx_embed(const float* x, const float* dir, const float* z, int64_t n, int per_ray, int mode, int n_freqs, int incl_input,
              float* out, int ld_out, int col0, void* stream) {
    REQUIRE(n >= 0 && per_ray >= 1 && mode >= 0 && mode <= 3 && n_freqs >= 0 && n_freqs <= 16, "nfx_embed: bad arguments");
    REQUIRE(incl_input || n_freqs > 0, "nfx_embed: empty encoding");
    if (n == 0) return NFX_OK;
    REQUIRE(out && (mode == 2 || x) && (mode == 0 || dir) && (mode != 1 || z), "nfx_embed: null pointer");
    REQUIRE(ld_out >= col0 + (incl_input ? 3 : 0) + 6 * n_freqs && col0 >= 0, "nfx_embed: output row too short");
    nfx::generic::EmbedArgs a{x, dir, z, n, per_ray, mode, n_freqs, incl_input, out, ld_out, col0};
    return nfx_hip_result(nfx_launch_embed(&a, (hipStream_t)stream), "embed");
}

int nfx_embed_bwd(const float* v, int64_t n, int n_freqs, int incl_input, const float* d_out, int ld_out, int col0, float* dv,
                  void* stream) {
    REQUIRE(n >= 0 && n_freqs >= 0 && n_freqs <= 16 && (incl_input || n_freqs > 0), "nfx_embed_bwd: bad arguments");
    if (n == 0) return NFX_OK;
    REQUIRE(v && d_out && dv, "nfx_embed_bwd: null pointer");
    REQUIRE(ld_out >= col0 + (incl_input ? 3 : 0) + 6 * n_freqs && col0 >= 0, "nfx_embed_bwd: gradient row too short");
    nfx::generic::EmbedArgs a{v, nullptr, nullptr, n, 1, 0, n_freqs, incl_input, nullptr, ld_out, col0};
    return nfx_hip_result(nfx_launch_embed_bwd(&a, d_out, dv, (hipStream_t)stream), "embed_bwd");
}
