// capi.cpp — the C-ABI of libnfx.so (include/nfx.h): argument validation, error reporting and
// dispatch to the kernel launchers.  No device allocation, no synchronisation.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <vector>

#include "../../include/nfx.h"
#include "nerf_layout.hpp"
#include "pack.hpp"

static thread_local char g_err[512] = "";

int nfx_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define fail nfx_fail
int nfx_hip_result(int e, const char* what);
#define hip_result nfx_hip_result
int nfx_hip_result(int e, const char* what) {
    if (e == 0) return NFX_OK;
    return fail(NFX_EHIP, "%s: HIP error %d (%s)", what, e, hipGetErrorString((hipError_t)e));
}
#define REQUIRE(cond, ...) \
    do {                   \
        if (!(cond)) return fail(NFX_EINVAL, __VA_ARGS__); \
    } while (0)
#define ALIGNED(p, a) ((((uintptr_t)(p)) & ((a)-1)) == 0)

extern "C" {
// launchers (defined in the .hip files)
int nfx_launch_nerf_mlp_bf16(const float*, const float*, const float*, long long, int, const void*,
                             float*, int, int, hipStream_t);
int nfx_launch_nerf_mlp_x3(const float*, const float*, const float*, long long, int, const void*, float*, int,
                           hipStream_t);
int nfx_launch_nerf_mlp_bf16_v6(const float*, const float*, const float*, long long, int, const void*, float*, int,
                                int, hipStream_t);
int nfx_launch_l2_normalize3(const float*, float*, long long, float, hipStream_t);
int nfx_launch_nonfinite(const float*, long long, int*, hipStream_t);
int nfx_launch_gen_z(float, float, int, long long, int, const float*, float*, hipStream_t);
int nfx_launch_composite(const float*, const float*, const float*, const float*, long long, int, int,
                         float*, float*, float*, float*, float*, hipStream_t);
int nfx_launch_sample_fine(const float*, const float*, long long, int, int, const float*, float*,
                           hipStream_t);
int nfx_launch_selftest_mfma(const float*, const float*, float*, hipStream_t);
int nfx_launch_selftest_sincos(const float*, long long, int, float*, hipStream_t);

int nfx_version(void) { return 100; }

int nfx_last_error(char* buf, size_t len) {
    if (!buf || len == 0) return NFX_EINVAL;
    strncpy(buf, g_err, len - 1);
    buf[len - 1] = 0;
    return NFX_OK;
}

// Options (nfx.h: nfx_set_option / nfx_get_option): process-wide integers, set by the host before the calls they
// affect.  No environment variable is read by the library; the Python binding forwards NFX_<KEY> variables once, at
// import (nerfactor_amd/_capi.py).  A key that was never set reads as the default of its call site.
namespace {
struct Option {
    const char* key;
    std::atomic<int> value;
    std::atomic<int> is_set;
};
Option g_options[] = {
    {"nerf_variant", {0}, {0}},   // NeRF MLP forward: 7 (default) | 6 | 8 | 1 | 0 — all bit-identical
    {"nerf_blocks", {0}, {0}},    // persistent grid of the NeRF kernels (default 256)
    {"m128_blocks", {0}, {0}},    // persistent grid of the width-128 kernels (default 256)
    {"lvis_variant", {0}, {0}},   // light visibility: 8 (default) | 2 | 3 | 4 | 0 — all bit-identical
    {"lvis_verify", {0}, {0}},    // k > 0: the HOST side (ops.lvis_fwd) re-runs ~1 % of the points of every k-th launch on the one-wave-per-SIMD
                                  //        kernel (variant 4) and raises on any differing bit; 0 / unset = off.  The library only stores it.
    {"lvis_rows", {0}, {0}},      // 1: the HOST side renders through nfx_lvis_fwd_rows / nfx_shade_olat_fwd_rows (visibilities and OLAT renders stored at
                                  //    their final rows, NaN flags from the kernels); 0 / unset = compact tensors + nfx_scatter_rows + nfx_any_nonfinite.
                                  //    Opt-in: measured equal within the noise (profiles/r06/render_rows_ab.txt).
    {"brdf_bwd_rows", {0}, {0}},  // 0: the HOST side differentiates the learned BRDF over every (point, light) row (nfx_brdf_spec_bwd);
                                  //    1 / unset = over the rows with a non-zero upstream gradient only (nfx_brdf_spec_bwd_rows; same bits)
    {"sigma_variant", {0}, {0}},    // bf16 density-only kernel: 1 / unset = the render kernel's dataflow (nerf_sigma_v6.hip), 0 = nerf_sigma_geo_kernel
    {"sigma_grad_rows", {0}, {0}},  // 0: the HOST side calls nfx_nerf_sigma_grad (reverse sweep for every sample); 1 / unset =
                                    //    nfx_nerf_sigma_grad_rows (only the samples with a positive density; same values)
    {"nerf_bwd_rows", {0}, {0}},  // 0: nfx_nerf_mlp_bwd differentiates every point; 1 / unset = only the points whose upstream gradient is not
                                  //    all zeros (device-side ordered row list; same sums, another fp32 summation order)
    {"brdf_variant", {0}, {0}},   // learned BRDF: 6 (default) | 5 | 2 | 3 | 4 | 0
    {"brdf_ct", {0}, {0}},        // column tiles of brdf variants 5 / 6: 4 (default) | 2 | 3; 8 = two waves per SIMD (variant 6 only)
    {"nerf_bwd", {0}, {0}},       // 1 (default) = LDS-DMA ring backward, 0 = register-staged identity reference
    {"nerf_bwd_nw", {0}, {0}},    // waves of the NeRF ring backward: 8 (default) | 4
    {"m128_bwd", {0}, {0}},       // as nerf_bwd for the width-128 networks
    {"wgrad_lds", {0}, {0}},      // force the LDS-staged weight-gradient GEMM on (1) / off (0); default by row count
    {"wgrad_slabs", {0}, {0}},    // number of row slabs of a weight-gradient batch (default by row count)
    {"wgrad_rounds", {0}, {0}},   // wide weight-gradient kernel: slabs x blocks = this many workgroups per CU (default 1); 0 = the
                                  //    row-count rule of rounds 3-5 (64 ... 256 slabs)
    {"wgrad_narrow", {0}, {0}},   // 0 = width-128 networks through the wide kernel as well
    {"wgrad_fused", {0}, {0}},    // 0 = backward kernels store activations, separate weight-gradient GEMMs (identity reference)
    {"wgrad_splits", {0}, {0}},   // runtime-shaped weight gradients: most row splits of the contraction (default 256)
    {"wgrad_map", {0}, {0}},      // runtime-shaped weight gradients: 1 (default) = a row split's jobs on one XCD | 0 = round-4 order
};
Option* find_option(const char* key) {
    if (!key) return nullptr;
    for (Option& o : g_options)
        if (strcmp(o.key, key) == 0) return &o;
    return nullptr;
}
}  // namespace

int nfx_option_int(const char* key, int dflt) {   // internal (hidden): the value of a set option, else dflt
    Option* o = find_option(key);
    return o && o->is_set.load(std::memory_order_acquire) ? o->value.load(std::memory_order_relaxed) : dflt;
}
int nfx_set_option(const char* key, int value) {
    Option* o = find_option(key);
    if (!o) return fail(NFX_EINVAL, "nfx_set_option: unknown option '%s'", key ? key : "(null)");
    o->value.store(value, std::memory_order_relaxed);
    o->is_set.store(1, std::memory_order_release);
    return NFX_OK;
}
int nfx_unset_option(const char* key) {
    Option* o = find_option(key);
    if (!o) return fail(NFX_EINVAL, "nfx_unset_option: unknown option '%s'", key ? key : "(null)");
    o->is_set.store(0, std::memory_order_release);
    return NFX_OK;
}
int nfx_get_option(const char* key, int* value, int* is_set) {
    Option* o = find_option(key);
    if (!o || !value) return fail(NFX_EINVAL, "nfx_get_option: unknown option '%s' or null output", key ? key : "(null)");
    *value = o->value.load(std::memory_order_relaxed);
    if (is_set) *is_set = o->is_set.load(std::memory_order_acquire);
    return NFX_OK;
}
#define env_int nfx_option_int

// --------------------------------------------------------------------------- packing
size_t nfx_nerf_packed_bytes(int prec) {
    if (prec == NFX_PREC_BF16) return nfx::nerf::kBlobBytes;
    if (prec == NFX_PREC_FP32) return (size_t)nfx::nerf::kBlobBytes + nfx::nerf::kWeightBytes;  // hi | lo | biases
    return 0;
}

// bf16 fragments of the 12 layers into w (nerf::kWeightBytes), biases into b (nerf::kBiasFloats)
static int pack_nerf_fragments(const float* const kernels[12], const float* const biases[12], uint8_t* w0, float* b) {
    using namespace nfx;
    using namespace nfx::pack;
    uint8_t* w = w0;
    const Seg pe_xyz{kPosEnc, 10, 0, nullptr};
    const Seg hid256{kHidden, 256, 0, nullptr};
    // enc[0]: posenc(xyz) 63 -> 256
    w += pack_layer_bf16({pe_xyz}, {{kernels[0], biases[0], 256}}, 8, 8, w, b + nerf::kBiasL0);
    for (int l = 1; l <= 7; ++l) {
        float* bl = b + nerf::kBiasL0 + 256 * l;
        if (l == 5) {  // input = concat(y[256], posenc(xyz)[63])   (mlp.py:47-48)
            const Seg pe_skip{kPosEnc, 10, 256, nullptr};
            w += pack_layer_bf16({hid256, pe_skip}, {{kernels[5], biases[5], 256}}, 8, 24, w, bl);
        } else {
            w += pack_layer_bf16({hid256}, {{kernels[l], biases[l], 256}}, 8, 16, w, bl);
        }
    }
    // fused [bottleneck (256 cols) | sigma_out (1 col)] on the encoder output
    w += pack_layer_bf16({hid256}, {{kernels[9], biases[9], 256}, {kernels[8], biases[8], 1}}, 9, 16, w,
                         b + nerf::kBiasBott);
    // rgb_out[0]: concat(bottleneck[256], posenc(view)[27]) -> 128
    const Seg pe_view{kPosEnc, 4, 256, nullptr};
    w += pack_layer_bf16({hid256, pe_view}, {{kernels[10], biases[10], 128}}, 4, 24, w, b + nerf::kBiasRgb0);
    // rgb_out[1]: 128 -> 3
    const Seg hid128{kHidden, 128, 0, nullptr};
    w += pack_layer_bf16({hid128}, {{kernels[11], biases[11], 3}}, 1, 8, w, b + nerf::kBiasRgb1);
    return w == w0 + nerf::kWeightBytes ? NFX_OK : NFX_EINVAL;
}

int nfx_nerf_pack_weights(const float* const kernels[12], const float* const biases[12], int prec,
                          void* blob, size_t blob_bytes) {
    using namespace nfx;
    REQUIRE(kernels && biases && blob, "nfx_nerf_pack_weights: null argument");
    for (int i = 0; i < 12; ++i) REQUIRE(kernels[i] && biases[i], "nfx_nerf_pack_weights: layer %d null", i);
    REQUIRE(prec == NFX_PREC_BF16 || prec == NFX_PREC_FP32, "nfx_nerf_pack_weights: bad prec %d", prec);
    REQUIRE(blob_bytes >= nfx_nerf_packed_bytes(prec), "nfx_nerf_pack_weights: blob too small (%zu < %zu)",
            blob_bytes, nfx_nerf_packed_bytes(prec));
    uint8_t* w = static_cast<uint8_t*>(blob);
    if (prec == NFX_PREC_BF16) {
        if (pack_nerf_fragments(kernels, biases, w, reinterpret_cast<float*>(w + nerf::kWeightBytes)))
            return fail(NFX_EINVAL, "nfx_nerf_pack_weights: internal layout mismatch");
        return NFX_OK;
    }
    // NFX_PREC_FP32 (nerf_mlp_x3.hip): [fragments of hi = bf16(W) | fragments of lo = bf16(W - hi) | fp32 biases]
    static const int rows[12] = {63, 256, 256, 256, 256, 319, 256, 256, 256, 256, 283, 128};
    static const int cols[12] = {256, 256, 256, 256, 256, 256, 256, 256, 1, 256, 128, 3};
    std::vector<std::vector<float>> lo(12);
    const float* lo_ptr[12];
    for (int i = 0; i < 12; ++i) {
        const size_t n = (size_t)rows[i] * cols[i];
        lo[i].resize(n);
        for (size_t k = 0; k < n; ++k) {
            const uint32_t bits = (uint32_t)pack::f32_to_bf16_rne(kernels[i][k]) << 16;
            float hi;
            memcpy(&hi, &bits, 4);
            lo[i][k] = kernels[i][k] - hi;
        }
        lo_ptr[i] = lo[i].data();
    }
    float* b = reinterpret_cast<float*>(w + 2 * (size_t)nerf::kWeightBytes);
    std::vector<float> sink(nerf::kBiasFloats);
    if (pack_nerf_fragments(kernels, biases, w, b) ||
        pack_nerf_fragments(lo_ptr, biases, w + nerf::kWeightBytes, sink.data()))
        return fail(NFX_EINVAL, "nfx_nerf_pack_weights: internal layout mismatch");
    return NFX_OK;
}

// ------------------------------------------------------------------------ NeRF path
int nfx_l2_normalize3(const float* in, float* out, int64_t n, float eps, void* stream) {
    REQUIRE(n >= 0, "nfx_l2_normalize3: n < 0");
    REQUIRE(n == 0 || (in && out), "nfx_l2_normalize3: null pointer");
    return hip_result(nfx_launch_l2_normalize3(in, out, n, eps, (hipStream_t)stream), "l2_normalize3");
}

int nfx_any_nonfinite(const float* x, int64_t n, int* flag, void* stream) {
    REQUIRE(n >= 0, "nfx_any_nonfinite: n < 0");
    REQUIRE(flag, "nfx_any_nonfinite: null flag");
    if (n == 0) return NFX_OK;
    REQUIRE(x, "nfx_any_nonfinite: null tensor");
    if (!ALIGNED(x, 16)) return nfx_fail(NFX_EALIGN, "nfx_any_nonfinite: tensor must be 16-byte aligned");
    return hip_result(nfx_launch_nonfinite(x, n, flag, (hipStream_t)stream), "any_nonfinite");
}

int nfx_launch_scatter_rows(const float*, const int*, long long, int, float*, hipStream_t);
int nfx_scatter_rows(const float* src, const int32_t* row_of, int64_t n_all, int d, float* dst, void* stream) {
    REQUIRE(n_all >= 0 && d >= 1, "nfx_scatter_rows: bad shape (%lld rows of %d)", (long long)n_all, d);
    if (n_all == 0) return NFX_OK;
    REQUIRE(row_of && dst, "nfx_scatter_rows: null pointer");   // (src may be null when no row is selected)
    // at most 2^31 elements (or float4 groups) per launch: the kernel's 32-bit grid-stride index advances by at most
    // 2^21 per step and must not wrap past 2^32; slice the rows
    const long long per_row = d % 4 == 0 ? d / 4 : d, max_rows = (1ll << 31) / per_row > 0 ? (1ll << 31) / per_row : 1;
    for (long long r0 = 0; r0 < n_all; r0 += max_rows) {
        const long long nr = n_all - r0 < max_rows ? n_all - r0 : max_rows;
        const int rc = hip_result(nfx_launch_scatter_rows(src, row_of + r0, nr, d, dst + r0 * d, (hipStream_t)stream),
                                  "scatter_rows");
        if (rc) return rc;
    }
    return NFX_OK;
}

int nfx_gen_z(float near, float far, int n_samples, int64_t n_rays, int lin_in_disp, const float* u,
              float* z, void* stream) {
    REQUIRE(n_samples >= 2, "nfx_gen_z: n_samples must be >= 2 (got %d)", n_samples);
    REQUIRE(n_rays >= 0, "nfx_gen_z: n_rays < 0");
    REQUIRE(n_rays == 0 || z, "nfx_gen_z: null output");
    return hip_result(nfx_launch_gen_z(near, far, n_samples, n_rays, lin_in_disp, u, z, (hipStream_t)stream),
                      "gen_z");
}

int nfx_nerf_mlp_fwd(const float* rayo, const float* rayd, const float* z, int64_t n_rays, int n_samples,
                     const void* blob, int prec, float* rgbs, void* stream) {
    REQUIRE(n_rays >= 0 && n_samples >= 1, "nfx_nerf_mlp_fwd: bad shape (%lld rays, %d samples)",
            (long long)n_rays, n_samples);
    if (n_rays == 0) return NFX_OK;
    REQUIRE(rayo && rayd && z && blob && rgbs, "nfx_nerf_mlp_fwd: null pointer");
    if (!ALIGNED(blob, 16) || !ALIGNED(rgbs, 16))
        return fail(NFX_EALIGN, "nfx_nerf_mlp_fwd: blob and rgbs must be 16-byte aligned");
    const long long n_pts = (long long)n_rays * n_samples;
    const int blocks = env_int("nerf_blocks", 256);
    if (prec == NFX_PREC_BF16) {
        // NFX_NERF_VARIANT: 7 (default) = one wave per SIMD, 64 points per wave, epilogue software-pipelined under the
        // next tile's MFMAs, weight stream by LDS-DMA into a 6-slot ring (nerf_mlp_v6.hip); 6 / 8 = the same kernel
        // with register-staged weights (one / two staging sets); 1 = the 8 waves x 32 points reference geometry with
        // two waves per SIMD, 0 = 4 x 64 plain (nerf_mlp.hip).  All bit-identical.  The intermediate variants 2, 3, 5
        // of r01 live in scripts/experiments/ (not built).
        const int variant = env_int("nerf_variant", 7);
        if (variant == 8)
            return hip_result(nfx_launch_nerf_mlp_bf16_v6(rayo, rayd, z, n_pts, n_samples, blob, rgbs, blocks, 2,
                                                          (hipStream_t)stream),
                              "nerf_mlp_fwd(bf16, v8)");
        if (variant == 7)
            return hip_result(nfx_launch_nerf_mlp_bf16_v6(rayo, rayd, z, n_pts, n_samples, blob, rgbs, blocks,
                                                          1,
                                                          (hipStream_t)stream),
                              "nerf_mlp_fwd(bf16, v7)");
        if (variant == 6)
            return hip_result(nfx_launch_nerf_mlp_bf16_v6(rayo, rayd, z, n_pts, n_samples, blob, rgbs, blocks,
                                                          0, (hipStream_t)stream),
                              "nerf_mlp_fwd(bf16, v6)");
        if (variant != 0 && variant != 1)
            return fail(NFX_EINVAL, "nfx_nerf_mlp_fwd: NFX_NERF_VARIANT %d is not built (0, 1, 6, 7, 8)", variant);
        return hip_result(nfx_launch_nerf_mlp_bf16(rayo, rayd, z, n_pts, n_samples, blob, rgbs, variant,
                                                   blocks, (hipStream_t)stream),
                          "nerf_mlp_fwd(bf16)");
    }
    if (prec == NFX_PREC_FP32)  // split-bf16 operands, 3 MFMAs per product (nerf_mlp_x3.hip)
        return hip_result(nfx_launch_nerf_mlp_x3(rayo, rayd, z, n_pts, n_samples, blob, rgbs, blocks, (hipStream_t)stream),
                          "nerf_mlp_fwd(fp32 via 3 x bf16)");
    return fail(NFX_EINVAL, "nfx_nerf_mlp_fwd: bad prec %d", prec);
}

int nfx_composite_fwd(const float* rgbs, const float* z, const float* rayd, const float* noise,
                      int64_t n_rays, int n_samples, int white_bg, float* rgb, float* occu, float* depth,
                      float* disp, float* weights, void* stream) {
    REQUIRE(n_rays >= 0 && n_samples >= 1, "nfx_composite_fwd: bad shape");
    if (n_rays == 0) return NFX_OK;
    REQUIRE(rgbs && z && rayd, "nfx_composite_fwd: null input");
    if (!ALIGNED(rgbs, 16)) return fail(NFX_EALIGN, "nfx_composite_fwd: rgbs must be 16-byte aligned");
    return hip_result(nfx_launch_composite(rgbs, z, rayd, noise, n_rays, n_samples, white_bg, rgb, occu,
                                           depth, disp, weights, (hipStream_t)stream),
                      "composite_fwd");
}

int nfx_sample_fine(const float* z, const float* weights, int64_t n_rays, int n_coarse, int n_fine,
                    const float* u, float* z_all, void* stream) {
    REQUIRE(n_rays >= 0, "nfx_sample_fine: n_rays < 0");
    REQUIRE(n_coarse <= 258, "nfx_sample_fine: at most 258 coarse samples (256 pdf bins), got %d", n_coarse);
    REQUIRE(n_coarse >= 3 && n_fine >= 2, "nfx_sample_fine: need n_coarse >= 3 and n_fine >= 2 (got %d, %d)",
            n_coarse, n_fine);
    REQUIRE((size_t)4 * (2 * (n_coarse - 1) + n_coarse + n_fine) * 4 <= 64 * 1024,
            "nfx_sample_fine: n_coarse + n_fine too large for one workgroup's LDS");
    if (n_rays == 0) return NFX_OK;
    REQUIRE(z && weights && z_all, "nfx_sample_fine: null pointer");
    return hip_result(
        nfx_launch_sample_fine(z, weights, n_rays, n_coarse, n_fine, u, z_all, (hipStream_t)stream),
        "sample_fine");
}

// --------------------------------------------------------------------- diagnostics
int nfx_selftest_mfma_bf16(const float* a, const float* b, float* d, void* stream) {
    REQUIRE(a && b && d, "nfx_selftest_mfma_bf16: null pointer");
    return hip_result(nfx_launch_selftest_mfma(a, b, d, (hipStream_t)stream), "selftest_mfma");
}
int nfx_launch_selftest_tr16(const float*, const float*, float*, int, hipStream_t);
int nfx_selftest_tr16(const float* h, const float* z, float* d, int mode, void* stream) {
    REQUIRE(d && (mode == 1 || (h && z)), "nfx_selftest_tr16: null pointer");
    return hip_result(nfx_launch_selftest_tr16(h, z, d, mode, (hipStream_t)stream), "selftest_tr16");
}
int nfx_selftest_sincos(const float* in, int64_t n, int which, float* out, void* stream) {
    REQUIRE(n >= 0 && (n == 0 || (in && out)), "nfx_selftest_sincos: bad arguments");
    return hip_result(nfx_launch_selftest_sincos(in, n, which, out, (hipStream_t)stream), "selftest_sincos");
}

}  // extern "C"
