// capi_train.cpp — C-ABI entry points of the training-side kernels (include/nfx.h).
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include <vector>

#include "../../include/nfx.h"
#include "mlp128_layout.hpp"
#include "nerf_train_layout.hpp"
#include "pack.hpp"

int nfx_fail(int code, const char* fmt, ...);
int nfx_hip_result(int e, const char* what);
extern "C" int nfx_option_int(const char* name, int dflt);

#define REQUIRE(cond, ...) \
    do {                   \
        if (!(cond)) return nfx_fail(NFX_EINVAL, __VA_ARGS__); \
    } while (0)
#define ALIGNED(p, a) ((((uintptr_t)(p)) & ((a)-1)) == 0)

extern "C" {
int nfx_launch_mlp128_bwd(int, const float*, const float*, long long, float, const float*, int, const void*, int,
                          int, float, const float*, void*, long long, int, hipStream_t);
int nfx_launch_mlp128_bwd_fused(int, const float*, const float*, long long, float, const float*, int, int, const void* const*,
                                const int*, const int*, const float*, const float* const*, float*, int, float* const*,
                                float* const*, hipStream_t);
size_t nfx_mlp128_fused_partial_floats(int in_kind, int grid);
int nfx_mlp128_fused_grid(int in_kind, long long n, int n_lights, int max_blocks);
int nfx_mlp128_train_feats(int in_kind);
int nfx_mlp128_train_blob_bytes(int in_kind);
struct nfx_wgrad_call {   // one weight-gradient GEMM of a batched launch (train.hip)
    const void* xt;
    const void* zt;
    int k_in, n_out;
    float* dw;
    float* db;
};
size_t nfx_wgrad_partial_bytes(const nfx_wgrad_call* calls, int n_calls, long long rows);
int nfx_launch_wgrad_batch(const nfx_wgrad_call* calls, int n_calls, long long ld, long long rows, void* partial,
                           hipStream_t st);
int nfx_launch_wgrad_batch_counted(const nfx_wgrad_call* calls, int n_calls, long long ld, long long rows, void* partial,
                                   const int* count, hipStream_t st);
int nfx_wgrad_counted_ok(long long rows);
int nfx_launch_amsgrad(float*, const float*, float*, float*, float*, long long, float, float, float, float,
                       hipStream_t);
float nfx_amsgrad_step_size(float lr, float beta1, float beta2, int64_t step);

static bool kind_ok(int k) { return k == NFX_IN_XYZ || k == NFX_IN_XYZ_LDIR; }
static int in_dims(int k) { return k == NFX_IN_XYZ ? 63 : 90; }

size_t nfx_mlp128_train_packed_bytes(int in_kind) {
    return kind_ok(in_kind) ? (size_t)nfx_mlp128_train_blob_bytes(in_kind) : 0;
}

int nfx_mlp128_pack_train_weights(const float* const kernels[5], const float* const biases[5], int in_kind,
                                  int out_dim, int prec, void* blob, size_t blob_bytes) {
    using namespace nfx::pack;
    REQUIRE(kernels && biases && blob, "nfx_mlp128_pack_train_weights: null argument");
    for (int i = 0; i < 5; ++i) REQUIRE(kernels[i] && biases[i], "nfx_mlp128_pack_train_weights: layer %d null", i);
    REQUIRE(kind_ok(in_kind), "nfx_mlp128_pack_train_weights: in_kind %d has no backward", in_kind);
    REQUIRE(out_dim >= 1 && out_dim <= 8, "nfx_mlp128_pack_train_weights: out_dim %d not in [1, 8]", out_dim);
    if (prec != NFX_PREC_BF16) return nfx_fail(NFX_ENOSUP, "nfx_mlp128_pack_train_weights: only bf16 is built");
    const size_t need = nfx_mlp128_train_packed_bytes(in_kind);
    REQUIRE(blob_bytes >= need, "nfx_mlp128_pack_train_weights: blob too small (%zu < %zu)", blob_bytes, need);
    const int ind = in_dims(in_kind);
    const bool lv = in_kind == NFX_IN_XYZ_LDIR;
    const int p0 = lv ? 8 : 4, p3 = lv ? 16 : 12;
    uint8_t* w = static_cast<uint8_t*>(blob);
    float* b = reinterpret_cast<float*>(w + need - nfx::m128::kMainBiasFloats * 4);
    const Seg hid{kHidden, 128, 0, nullptr};
    std::vector<Seg> in0{Seg{kPosEnc, 10, 0, nullptr}}, in3{hid, Seg{kPosEnc, 10, 128, nullptr}};
    if (lv) {
        in0.push_back(Seg{kPosEnc, 4, 63, nullptr});
        in3.push_back(Seg{kPosEnc, 4, 128 + 63, nullptr});
    }
    // ---- forward fragments (same dataflow as the inference kernels, un-folded input)
    w += pack_layer_bf16(in0, {{kernels[0], biases[0], 128}}, 4, p0, w, b);
    w += pack_layer_bf16({hid}, {{kernels[1], biases[1], 128}}, 4, 8, w, b + 128);
    w += pack_layer_bf16({hid}, {{kernels[2], biases[2], 128}}, 4, 8, w, b + 256);
    w += pack_layer_bf16(in3, {{kernels[3], biases[3], 128}}, 4, p3, w, b + 384);
    w += pack_layer_bf16({hid}, {{kernels[4], biases[4], out_dim}}, 1, 8, w, b + 512);
    // ---- dgrad fragments: layer l backward = a Dense whose Keras kernel is W_l^T ([out, in])
    std::vector<float> scratch_bias(128, 0.f), bias_sink(128 * 4);
    auto transposed = [&](const float* k, int rows_used, int cols, int pad_rows) {
        std::vector<float> t((size_t)pad_rows * rows_used, 0.f);  // [cols(pad) , rows_used]
        for (int r = 0; r < rows_used; ++r)
            for (int c = 0; c < cols; ++c) t[(size_t)c * rows_used + r] = k[(size_t)r * cols + c];
        return t;
    };
    {   // through the out layer: dH3[128] = Wo[128, out] dZo[out]; 16 padded gradient slots
        std::vector<float> t = transposed(kernels[4], 128, out_dim, 16);
        w += pack_layer_bf16({Seg{kHidden, 16, 0, nullptr}}, {{t.data(), nullptr, 128}}, 4, 4, w, bias_sink.data());
    }
    {   // through L3: only the h2 rows (first 128) carry a gradient that is needed
        std::vector<float> t = transposed(kernels[3], 128, 128, 128);
        w += pack_layer_bf16({hid}, {{t.data(), nullptr, 128}}, 4, 8, w, bias_sink.data());
    }
    for (int l = 2; l >= 1; --l) {
        std::vector<float> t = transposed(kernels[l], 128, 128, 128);
        w += pack_layer_bf16({hid}, {{t.data(), nullptr, 128}}, 4, 8, w, bias_sink.data());
    }
    (void)ind;
    if (w != reinterpret_cast<uint8_t*>(b)) return nfx_fail(NFX_EINVAL, "nfx_mlp128_pack_train_weights: layout mismatch");
    return NFX_OK;
}

static long long ld_for(int in_kind, int64_t n, int n_lights) {
    const long long rows = in_kind == NFX_IN_XYZ ? n : n * (long long)n_lights;
    return (rows + 127) / 128 * 128;
}

// the six weight-gradient GEMMs of one width-128 backward (pointers filled in by the caller)
static void mlp128_wgrad_calls(int in_kind, int out_dim, nfx_wgrad_call (&calls)[6]) {
    const int ind = in_dims(in_kind);
    const int dims[6][2] = {{ind, 128}, {128, 128}, {128, 128}, {128, 128}, {ind, 128}, {128, out_dim}};
    for (int i = 0; i < 6; ++i) calls[i] = nfx_wgrad_call{nullptr, nullptr, dims[i][0], dims[i][1], nullptr, nullptr};
}
static size_t mlp128_feat_bytes(int in_kind, int64_t n, int n_lights) {
    return ((size_t)nfx_mlp128_train_feats(in_kind) * ld_for(in_kind, n, n_lights) * 2 + 255) / 256 * 256;
}

// option wgrad_fused (default 1): weight gradients accumulated inside the backward kernels (mlp128_bwd_fused.hip); the
// workspace then holds only the per-workgroup partial sums.  0 = the round-3 path (activations stored, GEMM launches).
static bool fused_wgrad() { return nfx_option_int("wgrad_fused", 1) != 0; }

size_t nfx_mlp128_bwd_workspace_bytes(int in_kind, int64_t n, int n_lights) {
    if (!kind_ok(in_kind) || n <= 0) return 0;
    if (fused_wgrad())
        return nfx_mlp128_fused_partial_floats(in_kind, nfx_mlp128_fused_grid(in_kind, n, n_lights,
                                                                            nfx_option_int("m128_blocks", 256))) * sizeof(float);
    nfx_wgrad_call calls[6];
    mlp128_wgrad_calls(in_kind, 8, calls);
    const long long rows = in_kind == NFX_IN_XYZ ? n : n * (long long)n_lights;
    // feature-major activations / gradients, then the per-slab partial sums of the weight gradients
    return mlp128_feat_bytes(in_kind, n, n_lights) + nfx_wgrad_partial_bytes(calls, 6, (rows + 15) / 16 * 16);
}

int nfx_mlp128_bwd(int in_kind, const float* xyz, const float* xyz_dir, int64_t n, float xyz_scale,
                   const float* lxyz, int n_lights, const void* blob, int out_dim, int out_act,
                   float post_scale, const float* dout, void* workspace, size_t workspace_bytes,
                   float* const dkernels[5], float* const dbiases[5], int prec, void* stream) {
    REQUIRE(kind_ok(in_kind), "nfx_mlp128_bwd: in_kind %d has no backward", in_kind);
    REQUIRE(n >= 0, "nfx_mlp128_bwd: n < 0");
    REQUIRE(out_dim >= 1 && out_dim <= 8 && out_act >= 0 && out_act <= 3, "nfx_mlp128_bwd: bad out_dim/act");
    if (prec != NFX_PREC_BF16) return nfx_fail(NFX_ENOSUP, "nfx_mlp128_bwd: only bf16 is built");
    if (in_kind == NFX_IN_XYZ_LDIR)
        REQUIRE(n_lights > 0 && lxyz, "nfx_mlp128_bwd: light positions required for NFX_IN_XYZ_LDIR");
    if (n == 0) return NFX_OK;
    REQUIRE(xyz && blob && dout && workspace && dkernels && dbiases, "nfx_mlp128_bwd: null pointer");
    for (int i = 0; i < 5; ++i) REQUIRE(dkernels[i] && dbiases[i], "nfx_mlp128_bwd: gradient buffer %d null", i);
    REQUIRE(workspace_bytes >= nfx_mlp128_bwd_workspace_bytes(in_kind, n, n_lights),
            "nfx_mlp128_bwd: workspace too small");
    if (!ALIGNED(blob, 16) || !ALIGNED(workspace, 16))
        return nfx_fail(NFX_EALIGN, "nfx_mlp128_bwd: blob and workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (fused_wgrad()) {
        const int grid = nfx_mlp128_fused_grid(in_kind, n, n_lights, nfx_option_int("m128_blocks", 256));
        return nfx_hip_result(nfx_launch_mlp128_bwd_fused(in_kind, xyz, xyz_dir ? xyz_dir : xyz, n, xyz_scale, lxyz, n_lights, 1,
                                                          &blob, &out_dim, &out_act, &post_scale, &dout,
                                                          static_cast<float*>(workspace), grid, dkernels, dbiases, st),
                              "mlp128_bwd(fused)");
    }
    const long long ld = ld_for(in_kind, n, n_lights);
    REQUIRE(12 * ld < (1ll << 32), "nfx_mlp128_bwd: at most %lld rows per call (got %lld; feat_store.hpp's 32-bit lane offsets)",
            (1ll << 32) / 12 - 256, ld);
    const long long rows = in_kind == NFX_IN_XYZ ? n : n * (long long)n_lights;
    const long long rows16 = (rows + 15) / 16 * 16;  // pad rows of the last tile hold exact zeros in dZ
    int rc = nfx_hip_result(nfx_launch_mlp128_bwd(in_kind, xyz, xyz_dir ? xyz_dir : xyz, n, xyz_scale, lxyz, n_lights,
                                                  blob, out_dim, out_act, post_scale, dout, workspace, ld,
                                                  nfx_option_int("m128_blocks", 256), st),
                            "mlp128_bwd");
    if (rc) return rc;
    const int ind = in_dims(in_kind), kx = in_kind == NFX_IN_XYZ ? 64 : 96;
    const char* ws = static_cast<const char*>(workspace);
    auto feat = [&](int f) { return ws + (size_t)f * ld * 2; };
    const int oH = kx, oDZ = kx + 512, oDZo = kx + 1024;
    const nfx_wgrad_call calls[6] = {
        {feat(0), feat(oDZ + 0), ind, 128, dkernels[0], dbiases[0]},
        {feat(oH + 0), feat(oDZ + 128), 128, 128, dkernels[1], dbiases[1]},
        {feat(oH + 128), feat(oDZ + 256), 128, 128, dkernels[2], dbiases[2]},
        {feat(oH + 256), feat(oDZ + 384), 128, 128, dkernels[3], dbiases[3]},
        {feat(0), feat(oDZ + 384), ind, 128, dkernels[3] + 128 * 128, nullptr},
        {feat(oH + 384), feat(oDZo), 128, out_dim, dkernels[4], dbiases[4]},
    };
    void* partial = static_cast<char*>(workspace) + mlp128_feat_bytes(in_kind, n, n_lights);
    return nfx_hip_result(nfx_launch_wgrad_batch(calls, 6, ld, rows16, partial, st), "wgrad");
}

// Up to NFX_MLP128_MAX_HEADS networks over the SAME rows in one launch pair (+ one reduction): the three xyz heads of a
// NeRFactor step.  Head i: blob dev_blobs[i], dout dev_douts[i], gradients dev_dkernels[5 i .. 5 i + 5) / dev_dbiases[...].
int nfx_mlp128_bwd_heads(int in_kind, const float* xyz, const float* xyz_dir, int64_t n, float xyz_scale, const float* lxyz,
                         int n_lights, int n_heads, const void* const* blobs, const int* out_dims, const int* out_acts,
                         const float* post_scales, const float* const* douts, void* workspace, size_t workspace_bytes,
                         float* const* dkernels, float* const* dbiases, int prec, void* stream) {
    REQUIRE(n_heads >= 1 && n_heads <= NFX_MLP128_MAX_HEADS, "nfx_mlp128_bwd_heads: 1 .. %d heads", NFX_MLP128_MAX_HEADS);
    REQUIRE(blobs && out_dims && out_acts && post_scales && douts && dkernels && dbiases, "nfx_mlp128_bwd_heads: null table");
    const size_t per_head = nfx_mlp128_bwd_workspace_bytes(in_kind, n, n_lights);
    if (n_heads == 1 || !fused_wgrad()) {        // (the unfused identity path has no multi-head form: head after head)
        for (int i = 0; i < n_heads; ++i) {
            const int rc = nfx_mlp128_bwd(in_kind, xyz, xyz_dir, n, xyz_scale, lxyz, n_lights, blobs[i], out_dims[i], out_acts[i],
                                          post_scales[i], douts[i], workspace, workspace_bytes, dkernels + 5 * i, dbiases + 5 * i,
                                          prec, stream);
            if (rc) return rc;
        }
        return NFX_OK;
    }
    REQUIRE(kind_ok(in_kind), "nfx_mlp128_bwd_heads: in_kind %d has no backward", in_kind);
    REQUIRE(n >= 0, "nfx_mlp128_bwd_heads: n < 0");
    if (prec != NFX_PREC_BF16) return nfx_fail(NFX_ENOSUP, "nfx_mlp128_bwd_heads: only bf16 is built");
    if (in_kind == NFX_IN_XYZ_LDIR) REQUIRE(n_lights > 0 && lxyz, "nfx_mlp128_bwd_heads: light positions required for NFX_IN_XYZ_LDIR");
    if (n == 0) return NFX_OK;
    REQUIRE(xyz && workspace, "nfx_mlp128_bwd_heads: null pointer");
    for (int i = 0; i < n_heads; ++i) {
        REQUIRE(out_dims[i] >= 1 && out_dims[i] <= 8 && out_acts[i] >= 0 && out_acts[i] <= 3, "nfx_mlp128_bwd_heads: head %d: bad out_dim/act", i);
        REQUIRE(blobs[i] && douts[i], "nfx_mlp128_bwd_heads: head %d: null pointer", i);
        if (!ALIGNED(blobs[i], 16)) return nfx_fail(NFX_EALIGN, "nfx_mlp128_bwd_heads: blob %d must be 16-byte aligned", i);
        for (int j = 0; j < 5; ++j) {
            REQUIRE(dkernels[5 * i + j] && dbiases[5 * i + j], "nfx_mlp128_bwd_heads: head %d: gradient buffer %d null", i, j);
            for (int k = 0; k < i; ++k)          // two heads adding into one buffer would race in the reduction
                REQUIRE(dkernels[5 * k + j] != dkernels[5 * i + j], "nfx_mlp128_bwd_heads: heads %d and %d share a gradient buffer", k, i);
        }
    }
    REQUIRE(workspace_bytes >= per_head * n_heads, "nfx_mlp128_bwd_heads: workspace too small (%zu < %zu)", workspace_bytes, per_head * n_heads);
    if (!ALIGNED(workspace, 16)) return nfx_fail(NFX_EALIGN, "nfx_mlp128_bwd_heads: workspace must be 16-byte aligned");
    const int grid = nfx_mlp128_fused_grid(in_kind, n, n_lights, nfx_option_int("m128_blocks", 256));
    return nfx_hip_result(nfx_launch_mlp128_bwd_fused(in_kind, xyz, xyz_dir ? xyz_dir : xyz, n, xyz_scale, lxyz, n_lights, n_heads,
                                                      blobs, out_dims, out_acts, post_scales, douts, static_cast<float*>(workspace),
                                                      grid, dkernels, dbiases, (hipStream_t)stream),
                          "mlp128_bwd_heads");
}

// ------------------------------------------------------------------------------------ NeRF MLP backward
int nfx_launch_nerf_bwd(const float*, const float*, const float*, long long, int, const void*, const float*, void*,
                        long long, int, hipStream_t, void*);
size_t nfx_nerf_bwd_list_bytes(long long n_pts);
int nfx_launch_composite_bwd(const float*, const float*, const float*, const float*, long long, int, int,
                             const float*, float*, hipStream_t);

size_t nfx_nerf_train_packed_bytes(int prec) { return prec == NFX_PREC_BF16 ? (size_t)nfx::nerf::kTrainBlobBytes : 0; }

int nfx_nerf_pack_train_weights(const float* const kernels[12], const float* const biases[12], int prec, void* blob,
                                size_t blob_bytes) {
    using namespace nfx;
    using namespace nfx::pack;
    REQUIRE(kernels && biases && blob, "nfx_nerf_pack_train_weights: null argument");
    for (int i = 0; i < 12; ++i) REQUIRE(kernels[i] && biases[i], "nfx_nerf_pack_train_weights: layer %d null", i);
    if (prec != NFX_PREC_BF16) return nfx_fail(NFX_ENOSUP, "nfx_nerf_pack_train_weights: only bf16 is built");
    REQUIRE(blob_bytes >= (size_t)nerf::kTrainBlobBytes, "nfx_nerf_pack_train_weights: blob too small (%zu < %d)",
            blob_bytes, nerf::kTrainBlobBytes);
    uint8_t* w0 = static_cast<uint8_t*>(blob);
    // forward fragments + biases: the inference packer's layout, biases moved behind the dgrad fragments
    std::vector<uint8_t> fwd(nfx_nerf_packed_bytes(prec));
    int rc = nfx_nerf_pack_weights(kernels, biases, prec, fwd.data(), fwd.size());
    if (rc) return rc;
    memcpy(w0, fwd.data(), nerf::kWeightBytes);
    memcpy(w0 + nerf::kTrainWeightBytes, fwd.data() + nerf::kWeightBytes, (size_t)nerf::kBiasFloats * 4);
    uint8_t* w = w0 + nerf::kWeightBytes;
    std::vector<float> sink(256);
    // W[r0 : r0 + n_in_used, :cols]^T as a Keras kernel [pad_in' = cols padded, out' = n_rows]
    auto transposed = [](const float* k, int row0, int n_rows, int cols, int pad_cols) {
        std::vector<float> t((size_t)pad_cols * n_rows, 0.f);
        for (int r = 0; r < n_rows; ++r)
            for (int c = 0; c < cols; ++c) t[(size_t)c * n_rows + r] = k[(size_t)(row0 + r) * cols + c];
        return t;
    };
    const Seg hid256{kHidden, 256, 0, nullptr}, hid128{kHidden, 128, 0, nullptr}, hid16{kHidden, 16, 0, nullptr};
    {   // D1: dR0[128] = Wrgb1[128, 3] dZrgb[3]
        std::vector<float> t = transposed(kernels[11], 0, 128, 3, 16);
        w += pack_layer_bf16({hid16}, {{t.data(), nullptr, 128}}, 4, 4, w, sink.data());
    }
    {   // D2: dBott[256] = Wrgb0[:256, 128] dZr0[128]
        std::vector<float> t = transposed(kernels[10], 0, 256, 128, 128);
        w += pack_layer_bf16({hid128}, {{t.data(), nullptr, 256}}, 8, 8, w, sink.data());
    }
    {   // D3: dA7[256] = Wbott[256, 256] dZbott[256] + Wsigma[256, 1] dZsigma  (input rows: 256 + 16 slots)
        std::vector<float> t((size_t)(256 + 16) * 256, 0.f);
        for (int r = 0; r < 256; ++r) {
            for (int c = 0; c < 256; ++c) t[(size_t)c * 256 + r] = kernels[9][(size_t)r * 256 + c];
            t[(size_t)256 * 256 + r] = kernels[8][r];
        }
        const Seg sig{kHidden, 16, 256, nullptr};
        w += pack_layer_bf16({hid256, sig}, {{t.data(), nullptr, 256}}, 8, 20, w, sink.data());
    }
    for (int l = 7; l >= 1; --l) {  // dA_{l-1}[256] = W_l[:256, 256] dZ_l[256]   (enc[5]: the y rows of [y, posenc])
        std::vector<float> t = transposed(kernels[l], 0, 256, 256, 256);
        w += pack_layer_bf16({hid256}, {{t.data(), nullptr, 256}}, 8, 16, w, sink.data());
    }
    if (w != w0 + nerf::kTrainWeightBytes) return nfx_fail(NFX_EINVAL, "nfx_nerf_pack_train_weights: layout mismatch");
    return NFX_OK;
}

static long long nerf_ld(long long n_pts) { return (n_pts + 255) / 256 * 256; }   // whole 256-row tiles (nerf_bwd.hip)

static const int kNerfWgradDims[14][2] = {{63, 256}, {256, 256}, {256, 256}, {256, 256}, {256, 256}, {256, 256},
                                          {256, 256}, {256, 256}, {63, 256}, {256, 1}, {256, 256}, {256, 128},
                                          {27, 128}, {128, 3}};
static size_t nerf_feat_bytes(long long n_pts) {
    return ((size_t)nfx::nerf::kTrainFeats * nerf_ld(n_pts) * 2 + 255) / 256 * 256;
}

size_t nfx_nerf_bwd_workspace_bytes(int64_t n_rays, int n_samples) {
    if (n_rays <= 0 || n_samples <= 0) return 0;
    nfx_wgrad_call calls[14];
    for (int i = 0; i < 14; ++i) calls[i] = nfx_wgrad_call{nullptr, nullptr, kNerfWgradDims[i][0], kNerfWgradDims[i][1], nullptr, nullptr};
    const long long n_pts = (long long)n_rays * n_samples;
    const size_t partial = (nfx_wgrad_partial_bytes(calls, 14, (n_pts + 15) / 16 * 16) + 15) / 16 * 16;
    return nerf_feat_bytes(n_pts) + partial + nfx_nerf_bwd_list_bytes(n_pts);   // (the list: option nerf_bwd_rows)
}
// the list of the points with a gradient sits behind the partial sums
static void* nerf_list_ws(void* workspace, int64_t n_rays, int n_samples) {
    return static_cast<char*>(workspace) + nfx_nerf_bwd_workspace_bytes(n_rays, n_samples) -
           nfx_nerf_bwd_list_bytes((long long)n_rays * n_samples);
}

int nfx_nerf_mlp_bwd(const float* rayo, const float* rayd, const float* z, int64_t n_rays, int n_samples,
                     const void* blob, int prec, const float* d_rgbs, void* workspace, size_t workspace_bytes,
                     float* const dkernels[12], float* const dbiases[12], void* stream) {
    using namespace nfx::nerf;
    REQUIRE(n_rays >= 0 && n_samples >= 1, "nfx_nerf_mlp_bwd: bad shape");
    if (prec != NFX_PREC_BF16) return nfx_fail(NFX_ENOSUP, "nfx_nerf_mlp_bwd: only bf16 is built");
    if (n_rays == 0) return NFX_OK;
    REQUIRE(rayo && rayd && z && blob && d_rgbs && workspace && dkernels && dbiases, "nfx_nerf_mlp_bwd: null pointer");
    for (int i = 0; i < 12; ++i) REQUIRE(dkernels[i] && dbiases[i], "nfx_nerf_mlp_bwd: gradient buffer %d null", i);
    REQUIRE(workspace_bytes >= nfx_nerf_bwd_workspace_bytes(n_rays, n_samples), "nfx_nerf_mlp_bwd: workspace too small");
    if (!ALIGNED(blob, 16) || !ALIGNED(workspace, 16) || !ALIGNED(d_rgbs, 16))
        return nfx_fail(NFX_EALIGN, "nfx_nerf_mlp_bwd: blob, workspace and d_rgbs must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const long long n_pts = (long long)n_rays * n_samples, ld = nerf_ld(n_pts);
    // feat_store.hpp: a lane's 32-bit offset inside a feature-pair row reaches row * 4 + 4 * ld2 = up to 12 * ld
    REQUIRE(12 * ld < (1ll << 32), "nfx_nerf_mlp_bwd: at most %lld points per call (got %lld)", (1ll << 32) / 12 - 256, n_pts);
    const long long rows16 = (n_pts + 15) / 16 * 16;  // pad rows of the last tile hold exact zeros in every dZ
    // default: only the points whose upstream gradient is not all zeros are differentiated (nerf_bwd.hip, rowsel: a
    // sample the composite gave no weight has four exact zeros).  They sit in rows [0, count) of the workspace, count
    // stays on the device, the weight-gradient batch reads it.  Option nerf_bwd_rows = 0: every point.
    const bool listed = nfx_option_int("nerf_bwd_rows", 1) != 0 && nfx_option_int("nerf_bwd", 1) != 0 &&
                        nfx_wgrad_counted_ok(rows16) != 0;
    void* list_ws = listed ? nerf_list_ws(workspace, n_rays, n_samples) : nullptr;
    int rc = nfx_hip_result(nfx_launch_nerf_bwd(rayo, rayd, z, n_pts, n_samples, blob, d_rgbs, workspace, ld,
                                                nfx_option_int("nerf_blocks", 256), st, list_ws),
                            "nerf_bwd");
    if (rc) return rc;
    const char* ws = static_cast<const char*>(workspace);
    auto feat = [&](int f) { return ws + (size_t)f * ld * 2; };
    std::vector<nfx_wgrad_call> calls;
    calls.push_back({feat(kOffPe), feat(kOffDZ), 63, 256, dkernels[0], dbiases[0]});
    for (int l = 1; l <= 7; ++l)
        calls.push_back({feat(kOffA + 256 * (l - 1)), feat(kOffDZ + 256 * l), 256, 256, dkernels[l], dbiases[l]});
    calls.push_back({feat(kOffPe), feat(kOffDZ + 256 * 5), 63, 256, dkernels[5] + 256 * 256, nullptr});  // skip rows
    calls.push_back({feat(kOffA + 256 * 7), feat(kOffDSig), 256, 1, dkernels[8], dbiases[8]});            // sigma_out
    calls.push_back({feat(kOffA + 256 * 7), feat(kOffDBott), 256, 256, dkernels[9], dbiases[9]});         // bottleneck
    calls.push_back({feat(kOffBott), feat(kOffDR0), 256, 128, dkernels[10], dbiases[10]});                // rgb_out[0]
    calls.push_back({feat(kOffPv), feat(kOffDR0), 27, 128, dkernels[10] + 256 * 128, nullptr});
    calls.push_back({feat(kOffR0), feat(kOffDRgb), 128, 3, dkernels[11], dbiases[11]});                   // rgb_out[1]
    for (size_t i = 0; i < calls.size(); ++i)
        if (calls[i].k_in != kNerfWgradDims[i][0] || calls[i].n_out != kNerfWgradDims[i][1])
            return nfx_fail(NFX_EINVAL, "nfx_nerf_mlp_bwd: weight-gradient table out of sync");
    void* partial = static_cast<char*>(workspace) + nerf_feat_bytes(n_pts);
    return nfx_hip_result(nfx_launch_wgrad_batch_counted(calls.data(), (int)calls.size(), ld, rows16, partial,
                                                         static_cast<const int*>(list_ws), st),
                          "wgrad");
}

int nfx_composite_bwd(const float* rgbs, const float* z, const float* rayd, const float* noise, int64_t n_rays,
                      int n_samples, int white_bg, const float* d_rgb, float* d_rgbs, void* stream) {
    REQUIRE(n_rays >= 0 && n_samples >= 1, "nfx_composite_bwd: bad shape");
    REQUIRE(n_samples <= 4096, "nfx_composite_bwd: at most 4096 samples per ray (got %d)", n_samples);
    if (n_rays == 0) return NFX_OK;
    REQUIRE(rgbs && z && rayd && d_rgb && d_rgbs, "nfx_composite_bwd: null pointer");
    if (!ALIGNED(rgbs, 16) || !ALIGNED(d_rgbs, 16))
        return nfx_fail(NFX_EALIGN, "nfx_composite_bwd: rgbs and d_rgbs must be 16-byte aligned");
    return nfx_hip_result(nfx_launch_composite_bwd(rgbs, z, rayd, noise, n_rays, n_samples, white_bg, d_rgb, d_rgbs,
                                                   (hipStream_t)stream),
                          "composite_bwd");
}

// ------------------------------------------------------------------------------------ device-side re-packing
int nfx_launch_pack_gather(const float*, const int*, long long, void*, hipStream_t);

int nfx_pack_gather(const float* src, const int32_t* map, int64_t n_words, void* blob, void* stream) {
    REQUIRE(n_words >= 0, "nfx_pack_gather: negative size");
    if (n_words == 0) return NFX_OK;
    REQUIRE(src && map && blob, "nfx_pack_gather: null pointer");
    if (!ALIGNED(blob, 16) || !ALIGNED(map, 8))
        return nfx_fail(NFX_EALIGN, "nfx_pack_gather: blob must be 16-byte and map 8-byte aligned");
    return nfx_hip_result(nfx_launch_pack_gather(src, map, n_words, blob, (hipStream_t)stream), "pack_gather");
}

int nfx_brdf_train_blob_bytes(void);
int nfx_launch_brdf_spec_bwd(const float*, const float*, const float*, const float*, int, const float*, int,
                             const void*, long long, const float*, float*, float*, void*, int, hipStream_t, void*);

size_t nfx_brdf_train_packed_bytes(void) { return (size_t)nfx_brdf_train_blob_bytes(); }

// same slot table as brdf_input_slots() in capi_nerfactor.cpp / brdf_spec_kernel
static void brdf_slots(int zd, int* slots) {
    for (int i = 0; i < 32; ++i) slots[i] = -1;
    for (int h = 0; h < 2; ++h) {
        int* s0 = slots + h * 8;
        for (int j = 0; j < 6; ++j) s0[j] = zd + 3 + 6 * (j / 3) + (j % 3) + (h ? 3 : 0);
        s0[6] = zd + (h ? 2 : 0);
        s0[7] = h ? 0 : zd + 1;
        int* s1 = slots + 16 + h * 8;
        for (int j = 0; j < 8; ++j) {
            const int i = 1 + 2 * j + h;
            s1[j] = i < zd ? i : -1;
        }
    }
}

int nfx_brdf_pack_train_weights(const float* const kernels[5], const float* const biases[5], int z_dim, int prec,
                                void* blob, size_t blob_bytes) {
    using namespace nfx::pack;
    REQUIRE(kernels && biases && blob, "nfx_brdf_pack_train_weights: null argument");
    for (int i = 0; i < 5; ++i) REQUIRE(kernels[i] && biases[i], "nfx_brdf_pack_train_weights: layer %d null", i);
    REQUIRE(z_dim >= 1 && z_dim <= nfx::m128::kMaxZDim, "nfx_brdf_pack_train_weights: z_dim %d unsupported", z_dim);
    if (prec != NFX_PREC_BF16) return nfx_fail(NFX_ENOSUP, "nfx_brdf_pack_train_weights: only bf16 is built");
    const size_t need = nfx_brdf_train_packed_bytes();
    REQUIRE(blob_bytes >= need, "nfx_brdf_pack_train_weights: blob too small (%zu < %zu)", blob_bytes, need);
    int slots[32];
    brdf_slots(z_dim, slots);
    uint8_t* w = static_cast<uint8_t*>(blob);
    float* b = reinterpret_cast<float*>(w + need - nfx::m128::kMainBiasFloats * 4);
    const Seg hid{kHidden, 128, 0, nullptr};
    const Seg in0{kRaw, 2, 0, slots}, in3{kRaw, 2, 128, slots};
    w += pack_layer_bf16({in0}, {{kernels[0], biases[0], 128}}, 4, 4, w, b);
    w += pack_layer_bf16({hid}, {{kernels[1], biases[1], 128}}, 4, 8, w, b + 128);
    w += pack_layer_bf16({hid}, {{kernels[2], biases[2], 128}}, 4, 8, w, b + 256);
    w += pack_layer_bf16({hid, in3}, {{kernels[3], biases[3], 128}}, 4, 12, w, b + 384);
    w += pack_layer_bf16({hid}, {{kernels[4], biases[4], 1}}, 1, 8, w, b + 512);
    std::vector<float> sink(128 * 4);
    auto hidden_t = [&](const float* k, int cols, int pad_rows) {  // W[:128, :cols]^T as [pad_rows, 128]
        std::vector<float> t((size_t)pad_rows * 128, 0.f);
        for (int r = 0; r < 128; ++r)
            for (int c = 0; c < cols; ++c) t[(size_t)c * 128 + r] = k[(size_t)r * cols + c];
        return t;
    };
    // input-gradient products: kernel'[k][f'] = W[row0 + input_row(slot f')][k], f' = F(s,h,j) of the slot
    auto input_t = [&](const float* k, int row0) {
        std::vector<float> t((size_t)128 * 32, 0.f);
        for (int s = 0; s < 2; ++s)
            for (int h = 0; h < 2; ++h)
                for (int j = 0; j < 8; ++j) {
                    const int src = slots[(s * 2 + h) * 8 + j];
                    if (src < 0) continue;
                    const int fp = 16 * s + (j & 3) + 8 * (j >> 2) + 4 * h;
                    for (int kk = 0; kk < 128; ++kk) t[(size_t)kk * 32 + fp] = k[(size_t)(row0 + src) * 128 + kk];
                }
        return t;
    };
    {
        std::vector<float> t = hidden_t(kernels[4], 1, 16);           // through the out layer
        w += pack_layer_bf16({Seg{kHidden, 16, 0, nullptr}}, {{t.data(), nullptr, 128}}, 4, 4, w, sink.data());
    }
    {
        std::vector<float> t = input_t(kernels[3], 128);              // W3[128:, :] dZ3 -> input slots
        w += pack_layer_bf16({hid}, {{t.data(), nullptr, 32}}, 1, 8, w, sink.data());
    }
    for (int l = 3; l >= 1; --l) {                                    // hidden-to-hidden dgrads
        std::vector<float> t = hidden_t(kernels[l], 128, 128);
        w += pack_layer_bf16({hid}, {{t.data(), nullptr, 128}}, 4, 8, w, sink.data());
    }
    {
        std::vector<float> t = input_t(kernels[0], 0);                // W0 dZ0 -> input slots
        w += pack_layer_bf16({hid}, {{t.data(), nullptr, 32}}, 1, 8, w, sink.data());
    }
    if (w != reinterpret_cast<uint8_t*>(b)) return nfx_fail(NFX_EINVAL, "nfx_brdf_pack_train_weights: layout mismatch");
    return NFX_OK;
}

size_t nfx_brdf_spec_bwd_workspace_bytes(int z_dim, int64_t n) {
    return n > 0 && z_dim >= 1 ? sizeof(long long) * (size_t)n * (z_dim + 3) : 0;
}

size_t nfx_brdf_spec_bwd_list_bytes(int64_t n, int n_lights) {
    return n > 0 && n_lights > 0 && n * (int64_t)n_lights < ((int64_t)1 << 31) ? sizeof(int) * ((size_t)n * n_lights + 4) : 0;
}

int nfx_brdf_spec_bwd(const float* xyz, const float* cam, const float* normal, const float* z, int z_dim,
                      const float* lxyz, int n_lights, const void* blob, int prec, int64_t n, const float* dspec,
                      float* d_z, float* d_normal, void* workspace, size_t workspace_bytes, void* stream) {
    return nfx_brdf_spec_bwd_rows(xyz, cam, normal, z, z_dim, lxyz, n_lights, blob, prec, n, dspec, d_z, d_normal, workspace,
                                  workspace_bytes, nullptr, 0, stream);
}

int nfx_brdf_spec_bwd_rows(const float* xyz, const float* cam, const float* normal, const float* z, int z_dim,
                           const float* lxyz, int n_lights, const void* blob, int prec, int64_t n, const float* dspec,
                           float* d_z, float* d_normal, void* workspace, size_t workspace_bytes, void* list_workspace,
                           size_t list_bytes, void* stream) {
    REQUIRE(n >= 0, "nfx_brdf_spec_bwd: n < 0");
    REQUIRE(z_dim >= 1 && z_dim <= nfx::m128::kMaxZDim, "nfx_brdf_spec_bwd: z_dim %d unsupported", z_dim);
    REQUIRE(n_lights > 0 && n_lights % 32 == 0, "nfx_brdf_spec_bwd: n_lights (%d) must be a multiple of 32", n_lights);
    if (prec != NFX_PREC_BF16) return nfx_fail(NFX_ENOSUP, "nfx_brdf_spec_bwd: only bf16 is built");
    if (n == 0) return NFX_OK;
    REQUIRE(xyz && cam && normal && z && lxyz && blob && dspec && d_z && d_normal, "nfx_brdf_spec_bwd: null pointer");
    REQUIRE(workspace && workspace_bytes >= nfx_brdf_spec_bwd_workspace_bytes(z_dim, n),
            "nfx_brdf_spec_bwd: workspace of nfx_brdf_spec_bwd_workspace_bytes(z_dim, n) bytes required");
    if (!ALIGNED(blob, 16) || !ALIGNED(workspace, 8))
        return nfx_fail(NFX_EALIGN, "nfx_brdf_spec_bwd: blob must be 16-byte, workspace 8-byte aligned");
    if (list_workspace != nullptr) {
        REQUIRE(nfx_brdf_spec_bwd_list_bytes(n, n_lights) > 0 && list_bytes >= nfx_brdf_spec_bwd_list_bytes(n, n_lights),
                "nfx_brdf_spec_bwd_rows: list workspace of nfx_brdf_spec_bwd_list_bytes(n, n_lights) bytes required (n n_lights < 2^31)");
        if (!ALIGNED(list_workspace, 16)) return nfx_fail(NFX_EALIGN, "nfx_brdf_spec_bwd_rows: list workspace must be 16-byte aligned");
    }
    return nfx_hip_result(nfx_launch_brdf_spec_bwd(xyz, cam, normal, z, z_dim, lxyz, n_lights, blob, n, dspec, d_z,
                                                   d_normal, workspace, nfx_option_int("m128_blocks", 256),
                                                   (hipStream_t)stream, list_workspace),
                          "brdf_spec_bwd");
}

// ---------------------------------------------------------------- the BRDF prior on explicit rows (f-4)
int nfx_brdf_rows_feats(void);
int nfx_launch_brdf_rows(int, const float*, int, const float*, long long, long long, const void*, const float*, float*,
                         void*, long long, int, hipStream_t);

static void brdf_rows_wgrad_calls(int z_dim, nfx_wgrad_call (&calls)[6]) {
    const int ind = z_dim + 15;
    const int dims[6][2] = {{ind, 128}, {128, 128}, {128, 128}, {128, 128}, {ind, 128}, {128, 1}};
    for (int i = 0; i < 6; ++i) calls[i] = nfx_wgrad_call{nullptr, nullptr, dims[i][0], dims[i][1], nullptr, nullptr};
}
static long long brdf_rows_ld(int64_t rows) { return (rows + 127) / 128 * 128; }
static size_t brdf_rows_feat_bytes(int64_t rows) {
    return ((size_t)nfx_brdf_rows_feats() * brdf_rows_ld(rows) * 2 + 255) / 256 * 256;
}

int nfx_brdf_rows_fwd(const float* z, int z_dim, const float* rusink, int64_t n, int reci, const void* blob, int prec,
                      float* out, void* stream) {
    REQUIRE(n >= 0, "nfx_brdf_rows_fwd: n < 0");
    REQUIRE(z_dim >= 1 && z_dim <= nfx::m128::kMaxZDim, "nfx_brdf_rows_fwd: z_dim %d unsupported", z_dim);
    if (prec != NFX_PREC_BF16) return nfx_fail(NFX_ENOSUP, "nfx_brdf_rows_fwd: only bf16 is built");
    if (n == 0) return NFX_OK;
    REQUIRE(z && rusink && blob && out, "nfx_brdf_rows_fwd: null pointer");
    if (!ALIGNED(blob, 16)) return nfx_fail(NFX_EALIGN, "nfx_brdf_rows_fwd: blob must be 16-byte aligned");
    return nfx_hip_result(nfx_launch_brdf_rows(0, z, z_dim, rusink, n, reci ? 2 * n : n, blob, nullptr, out, nullptr, 0,
                                               nfx_option_int("m128_blocks", 256), (hipStream_t)stream),
                          "brdf_rows_fwd");
}

size_t nfx_brdf_rows_bwd_workspace_bytes(int z_dim, int64_t n, int reci) {
    if (n <= 0 || z_dim < 1 || z_dim > nfx::m128::kMaxZDim) return 0;
    const int64_t rows = reci ? 2 * n : n;
    nfx_wgrad_call calls[6];
    brdf_rows_wgrad_calls(z_dim, calls);
    return brdf_rows_feat_bytes(rows) + nfx_wgrad_partial_bytes(calls, 6, (rows + 15) / 16 * 16);
}

int nfx_brdf_rows_bwd(const float* z, int z_dim, const float* rusink, int64_t n, int reci, const void* blob, int prec,
                      const float* dout, void* workspace, size_t workspace_bytes, float* d_z,
                      float* const dkernels[5], float* const dbiases[5], void* stream) {
    REQUIRE(n >= 0, "nfx_brdf_rows_bwd: n < 0");
    REQUIRE(z_dim >= 1 && z_dim <= nfx::m128::kMaxZDim, "nfx_brdf_rows_bwd: z_dim %d unsupported", z_dim);
    if (prec != NFX_PREC_BF16) return nfx_fail(NFX_ENOSUP, "nfx_brdf_rows_bwd: only bf16 is built");
    if (n == 0) return NFX_OK;
    REQUIRE(z && rusink && blob && dout && workspace && d_z && dkernels && dbiases, "nfx_brdf_rows_bwd: null pointer");
    for (int i = 0; i < 5; ++i) REQUIRE(dkernels[i] && dbiases[i], "nfx_brdf_rows_bwd: gradient buffer %d null", i);
    REQUIRE(workspace_bytes >= nfx_brdf_rows_bwd_workspace_bytes(z_dim, n, reci), "nfx_brdf_rows_bwd: workspace too small");
    if (!ALIGNED(blob, 16) || !ALIGNED(workspace, 16))
        return nfx_fail(NFX_EALIGN, "nfx_brdf_rows_bwd: blob and workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int64_t rows = reci ? 2 * n : n;
    const long long ld = brdf_rows_ld(rows), rows16 = (rows + 15) / 16 * 16;
    REQUIRE(12 * ld < (1ll << 32), "nfx_brdf_rows_bwd: at most %lld rows per call (got %lld)", (1ll << 32) / 12 - 256, ld);
    int rc = nfx_hip_result(nfx_launch_brdf_rows(1, z, z_dim, rusink, n, rows, blob, dout, d_z, workspace, ld,
                                                 nfx_option_int("m128_blocks", 256), st), "brdf_rows_bwd");
    if (rc) return rc;
    const int ind = z_dim + 15;
    const char* ws = static_cast<const char*>(workspace);
    auto feat = [&](int f) { return ws + (size_t)f * ld * 2; };
    const int oH = 32, oDZ = 32 + 512, oDZo = 32 + 1024;
    const nfx_wgrad_call calls[6] = {
        {feat(0), feat(oDZ + 0), ind, 128, dkernels[0], dbiases[0]},
        {feat(oH + 0), feat(oDZ + 128), 128, 128, dkernels[1], dbiases[1]},
        {feat(oH + 128), feat(oDZ + 256), 128, 128, dkernels[2], dbiases[2]},
        {feat(oH + 256), feat(oDZ + 384), 128, 128, dkernels[3], dbiases[3]},
        {feat(0), feat(oDZ + 384), ind, 128, dkernels[3] + 128 * 128, nullptr},
        {feat(oH + 384), feat(oDZo), 128, 1, dkernels[4], dbiases[4]},
    };
    void* partial = static_cast<char*>(workspace) + brdf_rows_feat_bytes(rows);
    return nfx_hip_result(nfx_launch_wgrad_batch(calls, 6, ld, rows16, partial, st), "wgrad");
}

int nfx_launch_shade_bwd(const float*, const float*, const float*, const float*, const float*, const float*, float,
                         float, const float*, const float*, const float*, const float*, long long, int, int,
                         const float*, float*, float*, float*, float*, float*, float*, void*, hipStream_t);

size_t nfx_shade_bwd_workspace_bytes(int n_lights) { return n_lights > 0 ? sizeof(long long) * 3 * (size_t)n_lights : 0; }

int nfx_shade_bwd(const float* xyz, const float* cam, const float* normal, const float* albedo, const float* rough,
                  const float* spec, float spec_scale, float f0, const float* lvis, const float* lxyz,
                  const float* lareas, const float* light, int64_t n, int n_lights, int linear2srgb,
                  const float* drgb, float* d_albedo, float* d_rough, float* d_spec, float* d_normal, float* d_lvis,
                  float* d_light, void* workspace, size_t workspace_bytes, void* stream) {
    REQUIRE(n >= 0 && n_lights > 0, "nfx_shade_bwd: bad shape");
    REQUIRE((size_t)7 * n_lights * sizeof(float) <= 160 * 1024, "nfx_shade_bwd: too many lights (%d)", n_lights);
    if (n == 0) return NFX_OK;
    REQUIRE(xyz && cam && normal && albedo && lvis && lxyz && lareas && light && drgb, "nfx_shade_bwd: null input");
    REQUIRE(rough || spec, "nfx_shade_bwd: need roughness (microfacet) or a specular term");
    REQUIRE(d_albedo && d_normal, "nfx_shade_bwd: d_albedo and d_normal are required outputs");
    REQUIRE(spec ? d_spec != nullptr : d_rough != nullptr, "nfx_shade_bwd: missing BRDF-parameter gradient output");
    if (d_light) {
        REQUIRE(workspace && workspace_bytes >= nfx_shade_bwd_workspace_bytes(n_lights),
                "nfx_shade_bwd: d_light needs a workspace of nfx_shade_bwd_workspace_bytes(n_lights) bytes");
        if (!ALIGNED(workspace, 8)) return nfx_fail(NFX_EALIGN, "nfx_shade_bwd: workspace must be 8-byte aligned");
    }
    return nfx_hip_result(nfx_launch_shade_bwd(xyz, cam, normal, albedo, rough, spec, spec_scale, f0, lvis, lxyz,
                                               lareas, light, n, n_lights, linear2srgb, drgb, d_albedo, d_rough,
                                               d_spec, d_normal, d_lvis, d_light, workspace, (hipStream_t)stream),
                          "shade_bwd");
}

int nfx_launch_pair_loss(int, const nfx_loss_term*, int, const float*, float, long long, float*, const float*,
                         hipStream_t);
static int pair_loss_check(const char* who, const nfx_loss_term* terms, int n_terms, int64_t n, bool bwd) {
    REQUIRE(n >= 0, "%s: n < 0", who);
    REQUIRE(terms && n_terms >= 1 && n_terms <= NFX_LOSS_MAX_TERMS, "%s: 1..%d terms", who, NFX_LOSS_MAX_TERMS);
    for (int i = 0; i < n_terms; ++i) {
        REQUIRE(terms[i].a && terms[i].b && terms[i].d >= 1, "%s: term %d has a null operand or d < 1", who, i);
        REQUIRE(terms[i].kind == NFX_LOSS_MSE || terms[i].kind == NFX_LOSS_MAE, "%s: term %d: unknown kind", who, i);
        if (!bwd) continue;
        REQUIRE(!(terms[i].flags & NFX_LOSS_ACCUM_A) || terms[i].ga, "%s: term %d accumulates into a null ga", who, i);
        REQUIRE(!(terms[i].flags & NFX_LOSS_ACCUM_B) || terms[i].gb, "%s: term %d accumulates into a null gb", who, i);
    }
    return NFX_OK;
}
int nfx_pair_loss_fwd(const nfx_loss_term* terms, int n_terms, const float* alpha, float bg, int64_t n, float* loss,
                      void* stream) {
    if (int rc = pair_loss_check("nfx_pair_loss_fwd", terms, n_terms, n, false)) return rc;
    if (n == 0) return NFX_OK;
    REQUIRE(loss, "nfx_pair_loss_fwd: null output");
    return nfx_hip_result(nfx_launch_pair_loss(0, terms, n_terms, alpha, bg, n, loss, nullptr, (hipStream_t)stream),
                          "pair_loss_fwd");
}
int nfx_pair_loss_bwd(const nfx_loss_term* terms, int n_terms, const float* alpha, float bg, int64_t n,
                      const float* dloss, void* stream) {
    if (int rc = pair_loss_check("nfx_pair_loss_bwd", terms, n_terms, n, true)) return rc;
    if (n == 0) return NFX_OK;
    REQUIRE(dloss, "nfx_pair_loss_bwd: null dloss");
    return nfx_hip_result(nfx_launch_pair_loss(1, terms, n_terms, alpha, bg, n, nullptr, dloss, (hipStream_t)stream),
                          "pair_loss_bwd");
}

int nfx_launch_l2_normalize_rows(int, const float*, const float*, float*, long long, int, float, hipStream_t);
int nfx_launch_light_smoothness(const float*, int, int, float, float, float*, float*, hipStream_t);
int nfx_l2_normalize_rows(const float* x, float* y, int64_t n, int d, float eps, void* stream) {
    REQUIRE(n >= 0 && d >= 1 && d <= 16, "nfx_l2_normalize_rows: n >= 0 and 1 <= d <= 16");
    REQUIRE(n == 0 || (x && y), "nfx_l2_normalize_rows: null pointer");
    return nfx_hip_result(nfx_launch_l2_normalize_rows(0, x, nullptr, y, n, d, eps, (hipStream_t)stream), "l2_normalize_rows");
}
int nfx_l2_normalize_rows_bwd(const float* x, const float* dy, float* dx, int64_t n, int d, float eps, void* stream) {
    REQUIRE(n >= 0 && d >= 1 && d <= 16, "nfx_l2_normalize_rows_bwd: n >= 0 and 1 <= d <= 16");
    REQUIRE(n == 0 || (x && dy && dx), "nfx_l2_normalize_rows_bwd: null pointer");
    return nfx_hip_result(nfx_launch_l2_normalize_rows(1, x, dy, dx, n, d, eps, (hipStream_t)stream), "l2_normalize_rows_bwd");
}
int nfx_light_smoothness(const float* light, int h, int w, float tv_weight, float achro_weight, float* loss, float* grad,
                         void* stream) {
    REQUIRE(h >= 1 && w >= 1 && (int64_t)h * w <= (1 << 20), "nfx_light_smoothness: bad probe size %d x %d", h, w);
    REQUIRE(light && loss && grad, "nfx_light_smoothness: null pointer");
    return nfx_hip_result(nfx_launch_light_smoothness(light, h, w, tv_weight, achro_weight, loss, grad, (hipStream_t)stream),
                          "light_smoothness");
}

int nfx_amsgrad_step(float* p, const float* g, float* m, float* v, float* vhat, int64_t n, float lr, float beta1,
                     float beta2, float eps, int64_t step, void* stream) {
    REQUIRE(n >= 0 && step >= 1, "nfx_amsgrad_step: bad n/step");
    if (n == 0) return NFX_OK;
    REQUIRE(p && g && m && v && vhat, "nfx_amsgrad_step: null pointer");
    const float lr_t = nfx_amsgrad_step_size(lr, beta1, beta2, step);
    return nfx_hip_result(nfx_launch_amsgrad(p, g, m, v, vhat, n, lr_t, beta1, beta2, eps, (hipStream_t)stream),
                          "amsgrad_step");
}
int nfx_launch_amsgrad_dev(float*, const float*, float*, float*, float*, long long, const float*, float, float, float,
                           hipStream_t);
float nfx_amsgrad_step_size(float lr, float beta1, float beta2, int64_t step) {
    const double b1p = pow((double)beta1, (double)step), b2p = pow((double)beta2, (double)step);
    return (float)((double)lr * sqrt(1.0 - b2p) / (1.0 - b1p));
}
int nfx_amsgrad_step_dev(float* p, const float* g, float* m, float* v, float* vhat, int64_t n, const float* dev_lr_t,
                         float beta1, float beta2, float eps, void* stream) {
    REQUIRE(n >= 0, "nfx_amsgrad_step_dev: bad n");
    if (n == 0) return NFX_OK;
    REQUIRE(p && g && m && v && vhat && dev_lr_t, "nfx_amsgrad_step_dev: null pointer");
    return nfx_hip_result(nfx_launch_amsgrad_dev(p, g, m, v, vhat, n, dev_lr_t, beta1, beta2, eps, (hipStream_t)stream),
                          "amsgrad_step_dev");
}
}  // extern "C"
