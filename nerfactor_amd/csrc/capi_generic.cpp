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
        L.ks_h = i == 0 ? 0 : (widths[i - 1] + 15) / 16;
        L.ks_x = (i == 0 || (skip_input && skip_input[i])) ? (d_in + 15) / 16 : 0;
        L.n_tiles = (widths[i] + 31) / 32;
        L.n_out = widths[i];
        L.act = acts ? acts[i] : 0;
        L.w_off = w;
        L.b_off = b;
        w += L.n_tiles * (L.ks_h + L.ks_x);
        b += L.n_tiles * 32;
    }
    *n_frags = w;
    *n_bias = b;
    return NFX_OK;
}

size_t nfx_mlp_generic_packed_bytes(int d_in, int n_layers, const int* widths, const int* skip_input) {
    nfx::generic::Layer t[nfx::generic::kMaxLayers];
    int nf, nb;
    if (!widths || layer_table(d_in, n_layers, widths, skip_input, nullptr, t, &nf, &nb)) return 0;
    return (size_t)nf * 1024 + (size_t)nb * 4;
}

int nfx_mlp_generic_pack(const float* const* kernels, const float* const* biases, int d_in, int n_layers, const int* widths,
                         const int* skip_input, void* blob, size_t blob_bytes) {
    REQUIRE(kernels && biases && widths && blob, "nfx_mlp_generic_pack: null argument");
    nfx::generic::Layer t[nfx::generic::kMaxLayers];
    int nf, nb;
    int rc = layer_table(d_in, n_layers, widths, skip_input, nullptr, t, &nf, &nb);
    if (rc) return rc;
    const size_t need = (size_t)nf * 1024 + (size_t)nb * 4;
    REQUIRE(blob_bytes >= need, "nfx_mlp_generic_pack: blob too small (%zu < %zu)", blob_bytes, need);
    memset(blob, 0, need);
    uint16_t* w = static_cast<uint16_t*>(blob);
    float* b = reinterpret_cast<float*>(static_cast<char*>(blob) + (size_t)nf * 1024);
    for (int i = 0; i < n_layers; ++i) {
        REQUIRE(kernels[i] && biases[i], "nfx_mlp_generic_pack: layer %d null", i);
        const nfx::generic::Layer& L = t[i];
        const int prev = i == 0 ? 0 : widths[i - 1];          // rows [0, prev): the previous layer's output
        const int n_in = prev + (L.ks_x ? d_in : 0);          // then the network input (mlp.py:48: y first)
        (void)n_in;
        for (int tl = 0; tl < L.n_tiles; ++tl)
            for (int s = 0; s < L.ks_h + L.ks_x; ++s) {
                uint16_t* frag = w + ((size_t)L.w_off + (size_t)tl * (L.ks_h + L.ks_x) + s) * 512;
                const bool from_x = s >= L.ks_h;
                const int base = from_x ? prev : 0, feat0 = 16 * (from_x ? s - L.ks_h : s), limit = from_x ? d_in : prev;
                for (int lane = 0; lane < 64; ++lane) {
                    const int m = lane & 31, g = lane >> 5, col = 32 * tl + m;
                    if (col >= L.n_out) continue;
                    for (int j = 0; j < 8; ++j) {
                        const int f = feat0 + 8 * g + j;
                        if (f >= limit) continue;
                        frag[lane * 8 + j] = nfx::pack::f32_to_bf16_rne(kernels[i][(size_t)(base + f) * L.n_out + col]);
                    }
                }
            }
        for (int c = 0; c < L.n_out; ++c) b[L.b_off + c] = biases[i][c];
    }
    return NFX_OK;
}

int nfx_mlp_generic_fwd(const float* x, int64_t n, int ld_x, int d_in, int n_layers, const int* widths, const int* acts,
                        const int* skip_input, const void* blob, float* y, int ld_y, int col0, void* stream) {
    REQUIRE(n >= 0, "nfx_mlp_generic_fwd: n < 0");
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
    a.weights = static_cast<const char*>(blob);
    a.biases = reinterpret_cast<const float*>(a.weights + (size_t)nf * 1024);
    a.y = y;
    a.ld_y = ld_y;
    a.col0 = col0;
    a.n_layers = n_layers;
    return nfx_hip_result(nfx_launch_mlp_generic(&a, 4 * nfx_option_int("nerf_blocks", 256), (hipStream_t)stream), "mlp_generic_fwd");
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
