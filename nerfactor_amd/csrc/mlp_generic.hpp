// mlp_generic.hpp — argument structures of the runtime-shaped MLP kernels, shared by mlp_generic.hip (device) and
// capi_generic.cpp (host: layer table, packer).
#pragma once
namespace nfx {
namespace generic {

constexpr int kMaxLayers = 16, kMaxIn = 320, kMaxHidden = 256;   // kMaxIn: concat(256 features, embedded view) = 283

struct Layer {
    int ks_h;      // k-steps (16 features) taken from the previous layer's output (0 for the first layer)
    int ks_x;      // k-steps taken from the network input (first layer, and layers behind a skip concatenation)
    int n_tiles;   // 32-wide output tiles
    int n_out;     // true output width
    int act;       // NFX_ACT_*
    int w_off;     // first fragment (1 KiB units) of this layer in the blob
    int b_off;     // first bias float (32 per tile)
};
struct Args {
    const float* x;   // [n, ld_x] network input (fp32)
    long long n;
    int ld_x, d_in;
    const char* weights;   // fragments
    const float* biases;
    float* y;         // [n, ld_y] output, columns [col0, col0 + n_out of the last layer)
    int ld_y, col0;
    int n_layers;
    Layer layer[kMaxLayers];
};
// Embedder (embedder.py:23-47): out[:, col0 ...] = [x, sin(f_0 x), cos(f_0 x), sin(f_1 x), cos(f_1 x), ...], f_k = 2^k,
// every block 3 wide; n_freqs = 0 with incl_input = the identity (pos_enc = False).  The vectors are x[row / per_ray],
// points along rays o[row / per_ray] + d[row / per_ray] z[row] (nerf.py:162-164), or the ray directions d[row / per_ray].
struct EmbedArgs {
    const float* x;      // [*, 3] vectors (mode 0) or ray origins (mode 1)
    const float* dir;    // ray directions (modes 1, 2)
    const float* z;      // [n] depths (mode 1)
    long long n;
    int per_ray;         // rows per source vector / ray
    int mode;            // 0: x   1: o + d z   2: d   3: normalize(dir[row % per_ray] - x[row / per_ray]) (surface -> light)
    int n_freqs, incl_input;
    float* out;
    int ld_out, col0;
};

}  // namespace generic
}  // namespace nfx
