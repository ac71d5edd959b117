// mlp_generic.hpp — argument structures of the runtime-shaped MLP kernels, shared by mlp_generic.hip (device) and
// capi_generic.cpp (host: layer table, packer).
#pragma once
namespace nfx {
namespace generic {

constexpr int kMaxLayers = 16, kMaxIn = 576, kMaxHidden = 512;   // kMaxIn: concat(512 features, embedded view) = 539 (round 5: widths up to 512)
// A wave pulls the network's fragments through its LDS ring kGroup at a time as ONE stream (mlp_generic.hip: Ring): both
// operand sources of every tile are padded to whole groups with zero fragments; within a layer the groups run k-group
// outer, output tile inner (the kernel reads a group's B operand once for all tiles); layers follow each other without
// gaps, the transposed (backward) fragments follow the forward ones, last layer first, per layer the input-gradient tiles
// (tile-major) before the hidden tiles.
constexpr int kGroup = 4;
constexpr int pad_group(int ks) { return (ks + kGroup - 1) / kGroup * kGroup; }

struct Layer {
    int ks_h;      // k-steps (16 features) taken from the previous layer's output (0 for the first layer) ...
    int ks_x;      // ... and from the network input (first layer, and layers behind a skip concatenation): each padded
                   // to whole groups (zero fragments), so that a group reads ONE source
    int ks_pad;    // fragments per output tile in the blob: ks_h + ks_x
    int n_tiles;   // 32-wide output tiles
    int n_out;     // true output width
    int act;       // NFX_ACT_*
    int w_off;     // first fragment of this layer in the blob: fragment ((kg n_tiles + tile) kGroup + j) = k-step 4 kg + j of tile
    int b_off;     // first bias float (32 per tile)
};
struct Args {
    const float* x;   // [n, ld_x] network input (fp32)
    long long n;
    int ld_x, d_in;
    const char* weights;   // fragments
    int n_frags;           // in the forward stream (the prefetch never reads past it)
    const float* biases;
    float* y;         // [n, ld_y] output, columns [col0, col0 + n_out of the last layer)
    int ld_y, col0;
    int n_layers;
    int x_pitch, h_pitch;   // LDS row pitches in bytes: (features padded to 64) x element size + 16 of the input / the widest layer
    int f32;                // operands: 0 bf16 (1-KiB fragments), 1 fp32 (2-KiB fragments, v_mfma_f32_32x32x2_f32)
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

// ---- backward (round 4): one kernel re-computes the forward, runs the dgrad chain and leaves every layer's input
// and output gradient in the workspace as TILE-BLOCKED, FEATURE-MAJOR bf16 — ws[row tile t][feature row f][32 rows],
// every matrix (network input, hidden outputs, gradients) a range of feature rows padded to a multiple of 32 — the
// layout in which a weight-gradient MFMA operand (one feature, 8 consecutive rows) is one 16-byte load, and in which
// everything a wave writes and reads back lies in cache lines no other wave touches.  A second kernel contracts the
// pairs over the rows (the bias gradients ride along as a product with a fragment of ones), a third reduces its row
// splits in a fixed order.
struct BwdLayer {
    int wt_off;    // first TRANSPOSED fragment of this layer: the M tiles over the network input first (tile-major,
                   // pad_group(2 n_tiles) k-steps over this layer's outputs each), then those over the previous layer's
                   // outputs (k-group outer, tile inner)
    int h_row;     // feature row (F units) of this layer's OUTPUT activations in the workspace (hidden layers only)
    int dz_row;    // feature row of this layer's output gradient
    int dw_off;    // float offset of this layer's kernel gradient in a partial slice ...
    int db_off;    // ... and of its bias gradient (behind all kernel gradients)
    int job0;      // first weight-gradient job (64 x 64 block of dW) of this layer
};
struct BwdArgs {
    Args f;                  // the forward's arguments; f.y unused
    int stream_frags;        // forward + transposed fragments the kernel walks per row tile (without layer 0's
                             // input-gradient tiles, the stream's tail, when dx is null)
    const float* dy;         // [n, ld_dy] gradient w.r.t. the activated outputs, columns [col0_dy, ...)
    int ld_dy, col0_dy;
    float* dx;               // [n, ld_dx] gradient w.r.t. the network input, or null
    int ld_dx;
    char* ws;                // workspace: bf16 | fp32 [tiles][feat_rows][32]
    long long tiles;         // row tiles (32 rows)
    int feat_rows;           // sum of all padded feature counts (x at feature row 0, hidden outputs, gradients)
    BwdLayer b[kMaxLayers];
};
struct WgradArgs {
    const char* ws;
    long long tiles;
    int feat_rows, n_layers, d_in, splits, n_jobs;   // (the element type of ws is the kernel's template argument)
    int map;                 // wave -> (job, split): 0 = splits fastest (round 4) | 1 = a split's jobs share workgroups, a split stays on one XCD
    long long slice;         // floats per partial slice: all kernel gradients, then all bias gradients
    long long dw_total;      // floats of kernel gradients in a slice
    float* partial;          // [splits][slice]
    Layer layer[kMaxLayers];
    BwdLayer b[kMaxLayers];
    float* dw[kMaxLayers];   // [n_in, n_out] fp32, ADDED into
    float* db[kMaxLayers];   // [n_out]
};

}  // namespace generic
}  // namespace nfx
