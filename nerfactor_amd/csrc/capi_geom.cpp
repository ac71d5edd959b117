// capi_geom.cpp — C-ABI entry points of the geometry-extraction kernels (include/nfx.h): density only, and density
// with its spatial gradient (geometry_from_nerf.py:249-350).
#include <hip/hip_runtime.h>
#include <string.h>

#include <vector>

#include "../../include/nfx.h"
#include "nerf_geom_layout.hpp"
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
int nfx_launch_nerf_sigma_geo(const float*, const float*, const float*, long long, int, const void*, float*, int,
                              hipStream_t);
int nfx_launch_nerf_sigma_grad(const float*, const float*, const float*, long long, int, const void*, float*, int,
                               hipStream_t);
int nfx_launch_refine_select(const float*, const float*, const float*, long long, int, float, float, float, float, int, int*, int*,
                             hipStream_t);
int nfx_launch_nerf_sigma_v6(const float*, const float*, const float*, long long, int, const void*, float*, int, hipStream_t);
int nfx_launch_nerf_sigma_grad_list(const float*, const float*, const float*, long long, int, const void*, float*, const int*,
                                    const int*, int, hipStream_t);
int nfx_launch_nerf_sigma_grad_x3_list(const float*, const float*, const float*, long long, int, const void*, float*,
                                       const int*, const int*, int, hipStream_t);
int nfx_launch_select_density(const float*, long long, float*, void*, hipStream_t);
size_t nfx_nerf_bwd_list_bytes(long long n_pts);   // nerf_bwd.hip: bytes of a rowsel list over n_pts rows
int nfx_launch_nerf_sigma_x3_list(const float*, const float*, const float*, long long, int, const void*, float*, const int*,
                                  const int*, int, hipStream_t);
int nfx_launch_nerf_sigma_x3(const float*, const float*, const float*, long long, int, const void*, float*, int,
                             hipStream_t);   // nerf_geom_x3.hip
int nfx_launch_nerf_sigma_grad_x3(const float*, const float*, const float*, long long, int, const void*, float*, int,
                                  hipStream_t);

size_t nfx_nerf_geom_packed_bytes(int prec) {
    using namespace nfx::nerf;
    if (prec == NFX_PREC_BF16) return (size_t)kGeoBlobBytes;
    if (prec == NFX_PREC_FP32) return 2 * (size_t)kGeoWeightBytes + kGeoFloats * sizeof(float);   // [hi | lo | floats]
    return 0;
}
}  // extern "C"

// The bf16 fragments of nerf_geom_layout.hpp's chunk sequence at w0 and its kGeoFloats floats at fl.
static int pack_geom_half(const float* const kernels[12], const float* const biases[12], uint8_t* w0, float* fl) {
    using namespace nfx;
    using namespace nfx::pack;
    std::vector<uint8_t> fwd(nfx_nerf_packed_bytes(NFX_PREC_BF16));
    int rc = nfx_nerf_pack_weights(kernels, biases, NFX_PREC_BF16, fwd.data(), fwd.size());
    if (rc) return rc;
    // forward: encoder chunks 0..63 and the sigma tile (chunk 72 = 9th tile of the fused [bottleneck | sigma_out])
    const size_t enc_bytes = (size_t)nerf::chunk_frag_offset(64) * 1024;
    memcpy(w0, fwd.data(), enc_bytes);
    memcpy(w0 + enc_bytes, fwd.data() + (size_t)nerf::chunk_frag_offset(72) * 1024, 16 * 1024);
    uint8_t* w = w0 + enc_bytes + 16 * 1024;
    std::vector<float> sink(256);
    const Seg hid256{kHidden, 256, 0, nullptr};
    auto hidden_t = [](const float* k) {  // W[:256, :256]^T
        std::vector<float> t((size_t)256 * 256);
        for (int r = 0; r < 256; ++r)
            for (int c = 0; c < 256; ++c) t[(size_t)c * 256 + r] = k[(size_t)r * 256 + c];
        return t;
    };
    // input-gradient product: kernel'[k][f'] = W[row0 + posenc_row(slot f')][k], f' = F(s,h,j) of the slot, so the
    // C/D lane map of the output tile IS the posenc slot layout of mlp_engine.hpp:posenc<10>
    auto input_t = [](const float* k, int row0) {
        std::vector<float> t((size_t)256 * 64, 0.f);
        const Seg pe{kPosEnc, 10, 0, nullptr};
        for (int s = 0; s < 4; ++s)
            for (int h = 0; h < 2; ++h)
                for (int j = 0; j < 8; ++j) {
                    const int src = seg_row(pe, s, h, j);
                    if (src < 0) continue;
                    const int fp = 32 * (s >> 1) + 16 * (s & 1) + (j & 3) + 8 * (j >> 2) + 4 * h;
                    for (int kk = 0; kk < 256; ++kk) t[(size_t)kk * 64 + fp] = k[(size_t)(row0 + src) * 256 + kk];
                }
        return t;
    };
    auto dgrad = [&](int l) {
        std::vector<float> t = hidden_t(kernels[l]);
        w += pack_layer_bf16({hid256}, {{t.data(), nullptr, 256}}, 8, 16, w, sink.data());
    };
    auto igrad = [&](int l, int row0) {
        std::vector<float> t = input_t(kernels[l], row0);
        w += pack_layer_bf16({hid256}, {{t.data(), nullptr, 64}}, 2, 16, w, sink.data());
    };
    dgrad(7);
    dgrad(6);
    igrad(5, 256);
    for (int l = 5; l >= 1; --l) dgrad(l);
    igrad(0, 0);
    if (w != w0 + nerf::kGeoWeightBytes) return nfx_fail(NFX_EINVAL, "nfx_nerf_pack_geom_weights: layout mismatch");
    const float* fb = reinterpret_cast<const float*>(fwd.data() + nerf::kWeightBytes);
    memcpy(fl, fb + nerf::kBiasL0, 8 * 256 * 4);
    memcpy(fl + nerf::kGeoBiasSig, fb + nerf::kBiasBott + 256, 32 * 4);
    // sigma_out kernel in fp32: the bf16 kernel rounds it to bf16 when it forms dZ7, as the forward's packed copy is;
    // the fp32-class kernel splits it into a hi / lo pair
    memcpy(fl + nerf::kGeoWSig, kernels[8], 256 * 4);
    return NFX_OK;
}

extern "C" {
int nfx_nerf_pack_geom_weights(const float* const kernels[12], const float* const biases[12], int prec, void* blob,
                               size_t blob_bytes) {
    using namespace nfx;
    REQUIRE(kernels && biases && blob, "nfx_nerf_pack_geom_weights: null argument");
    for (int i = 0; i < 12; ++i) REQUIRE(kernels[i] && biases[i], "nfx_nerf_pack_geom_weights: layer %d null", i);
    REQUIRE(prec == NFX_PREC_BF16 || prec == NFX_PREC_FP32, "nfx_nerf_pack_geom_weights: bad prec %d", prec);
    REQUIRE(blob_bytes >= nfx_nerf_geom_packed_bytes(prec), "nfx_nerf_pack_geom_weights: blob too small (%zu < %zu)",
            blob_bytes, nfx_nerf_geom_packed_bytes(prec));
    uint8_t* w0 = static_cast<uint8_t*>(blob);
    if (prec == NFX_PREC_BF16) return pack_geom_half(kernels, biases, w0, reinterpret_cast<float*>(w0 + nerf::kGeoWeightBytes));
    // NFX_PREC_FP32 (nerf_geom_x3.hip): the chunk sequence of hi = bf16(W), then of lo = bf16(W - hi), then the floats
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
    std::vector<float> sink(nerf::kGeoFloats);
    int rc = pack_geom_half(kernels, biases, w0, reinterpret_cast<float*>(w0 + 2 * (size_t)nerf::kGeoWeightBytes));
    if (rc) return rc;
    return pack_geom_half(lo_ptr, biases, w0 + nerf::kGeoWeightBytes, sink.data());
}

// bf16 density of every sample: the render kernel's dataflow over the GEOM blob (nerf_sigma_v6.hip, round 6; default) or the
// round-2 kernel (option sigma_variant = 0) — the same MFMAs on the same operands, bit-identical
static int launch_sigma_bf16(const float* rayo, const float* rayd, const float* z, long long n_pts, int n_samples,
                             const void* blob, float* sigma, hipStream_t st) {
    const int blocks = nfx_option_int("nerf_blocks", 256);
    if (nfx_option_int("sigma_variant", 1) != 0)
        return nfx_launch_nerf_sigma_v6(rayo, rayd, z, n_pts, n_samples, blob, sigma, blocks, st);
    return nfx_launch_nerf_sigma_geo(rayo, rayd, z, n_pts, n_samples, blob, sigma, blocks, st);
}

int nfx_nerf_sigma_fwd(const float* rayo, const float* rayd, const float* z, int64_t n_rays, int n_samples,
                       const void* blob, int prec, float* sigma, void* stream) {
    REQUIRE(n_rays >= 0 && n_samples >= 1, "nfx_nerf_sigma_fwd: bad shape");
    REQUIRE(prec == NFX_PREC_BF16 || prec == NFX_PREC_FP32, "nfx_nerf_sigma_fwd: bad prec %d", prec);
    if (n_rays == 0) return NFX_OK;
    REQUIRE(rayo && rayd && z && blob && sigma, "nfx_nerf_sigma_fwd: null pointer");
    if (!ALIGNED(blob, 16)) return nfx_fail(NFX_EALIGN, "nfx_nerf_sigma_fwd: blob must be 16-byte aligned");
    if (prec == NFX_PREC_FP32)
        return nfx_hip_result(nfx_launch_nerf_sigma_x3(rayo, rayd, z, (long long)n_rays * n_samples, n_samples, blob,
                                                       sigma, nfx_option_int("nerf_blocks", 256), (hipStream_t)stream),
                              "nerf_sigma_fwd(fp32)");
    return nfx_hip_result(launch_sigma_bf16(rayo, rayd, z, (long long)n_rays * n_samples, n_samples, blob, sigma,
                                            (hipStream_t)stream),
                          "nerf_sigma_fwd");
}

int nfx_nerf_refine_select(const float* rgbs, const float* z, const float* rayd, int64_t n_rays, int n_samples,
                           float t_min, float a_lo, float a_hi, float sigma_margin, int dilate, int* list, int* count,
                           void* stream) {
    REQUIRE(n_rays >= 0 && n_samples >= 1 && n_samples <= 512, "nfx_nerf_refine_select: bad shape (1 <= S <= 512, got %d)", n_samples);
    REQUIRE(n_rays * (int64_t)n_samples < (int64_t)1 << 31, "nfx_nerf_refine_select: %lld samples do not fit int32 indices",
            (long long)(n_rays * n_samples));
    REQUIRE(dilate >= 0 && dilate <= 8, "nfx_nerf_refine_select: dilate = %d (0 .. 8)", dilate);
    REQUIRE(count, "nfx_nerf_refine_select: null count");
    if (n_rays > 0) REQUIRE(rgbs && z && rayd && list, "nfx_nerf_refine_select: null pointer");
    if (n_rays > 0 && !ALIGNED(rgbs, 16)) return nfx_fail(NFX_EALIGN, "nfx_nerf_refine_select: rgbs must be 16-byte aligned");
    return nfx_hip_result(nfx_launch_refine_select(rgbs, z, rayd, (long long)n_rays, n_samples, t_min, a_lo, a_hi, sigma_margin, dilate,
                                                   list, count, (hipStream_t)stream), "nerf_refine_select");
}

int nfx_nerf_sigma_refine(const float* rayo, const float* rayd, const float* z, int64_t n_rays, int n_samples,
                          const void* blob, const int* list, const int* count, float* rgbs, void* stream) {
    REQUIRE(n_rays >= 0 && n_samples >= 1, "nfx_nerf_sigma_refine: bad shape");
    if (n_rays == 0) return NFX_OK;
    REQUIRE(rayo && rayd && z && blob && list && count && rgbs, "nfx_nerf_sigma_refine: null pointer");
    if (!ALIGNED(blob, 16)) return nfx_fail(NFX_EALIGN, "nfx_nerf_sigma_refine: blob must be 16-byte aligned");
    return nfx_hip_result(nfx_launch_nerf_sigma_x3_list(rayo, rayd, z, (long long)n_rays * n_samples, n_samples, blob, rgbs,
                                                        list, count, nfx_option_int("nerf_blocks", 256),
                                                        (hipStream_t)stream), "nerf_sigma_refine");
}

int nfx_nerf_sigma_grad(const float* rayo, const float* rayd, const float* z, int64_t n_rays, int n_samples,
                        const void* geom_blob, int prec, float* normal_sigma, void* stream) {
    REQUIRE(n_rays >= 0 && n_samples >= 1, "nfx_nerf_sigma_grad: bad shape");
    REQUIRE(prec == NFX_PREC_BF16 || prec == NFX_PREC_FP32, "nfx_nerf_sigma_grad: bad prec %d", prec);
    if (n_rays == 0) return NFX_OK;
    REQUIRE(rayo && rayd && z && geom_blob && normal_sigma, "nfx_nerf_sigma_grad: null pointer");
    if (!ALIGNED(geom_blob, 16) || !ALIGNED(normal_sigma, 16))
        return nfx_fail(NFX_EALIGN, "nfx_nerf_sigma_grad: blob and output must be 16-byte aligned");
    if (prec == NFX_PREC_FP32)
        return nfx_hip_result(nfx_launch_nerf_sigma_grad_x3(rayo, rayd, z, (long long)n_rays * n_samples, n_samples,
                                                            geom_blob, normal_sigma,
                                                            nfx_option_int("nerf_blocks", 256), (hipStream_t)stream),
                              "nerf_sigma_grad(fp32)");
    return nfx_hip_result(nfx_launch_nerf_sigma_grad(rayo, rayd, z, (long long)n_rays * n_samples, n_samples,
                                                     geom_blob, normal_sigma, nfx_option_int("nerf_blocks", 256),
                                                     (hipStream_t)stream),
                          "nerf_sigma_grad");
}

// the density of every sample [n_pts floats, padded to 16 bytes], then the row list (rowsel.hpp)
static size_t sigma_grad_sigma_bytes(long long n_pts) { return ((size_t)n_pts * 4 + 15) / 16 * 16; }
size_t nfx_nerf_sigma_grad_workspace_bytes(int64_t n_rays, int n_samples) {
    if (n_rays <= 0 || n_samples <= 0) return 0;
    const long long n_pts = (long long)n_rays * n_samples;
    return sigma_grad_sigma_bytes(n_pts) + nfx_nerf_bwd_list_bytes(n_pts);
}

int nfx_nerf_sigma_grad_rows(const float* rayo, const float* rayd, const float* z, int64_t n_rays, int n_samples,
                             const void* geom_blob, int prec, float* normal_sigma, void* workspace,
                             size_t workspace_bytes, void* stream) {
    REQUIRE(n_rays >= 0 && n_samples >= 1, "nfx_nerf_sigma_grad_rows: bad shape");
    REQUIRE(prec == NFX_PREC_BF16 || prec == NFX_PREC_FP32, "nfx_nerf_sigma_grad_rows: bad prec %d", prec);
    if (n_rays == 0) return NFX_OK;
    REQUIRE(rayo && rayd && z && geom_blob && normal_sigma && workspace, "nfx_nerf_sigma_grad_rows: null pointer");
    const long long n_pts = (long long)n_rays * n_samples;
    REQUIRE(n_pts < (1ll << 31), "nfx_nerf_sigma_grad_rows: %lld samples do not fit int32 indices", n_pts);
    REQUIRE(workspace_bytes >= nfx_nerf_sigma_grad_workspace_bytes(n_rays, n_samples), "nfx_nerf_sigma_grad_rows: workspace too small");
    if (!ALIGNED(geom_blob, 16) || !ALIGNED(normal_sigma, 16) || !ALIGNED(workspace, 16))
        return nfx_fail(NFX_EALIGN, "nfx_nerf_sigma_grad_rows: blob, output and workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int blocks = nfx_option_int("nerf_blocks", 256);
    float* sigma = static_cast<float*>(workspace);
    void* list_ws = static_cast<char*>(workspace) + sigma_grad_sigma_bytes(n_pts);
    const int* count = static_cast<const int*>(list_ws);
    const int* list = reinterpret_cast<const int*>(static_cast<char*>(list_ws) + nfx_nerf_bwd_list_bytes(n_pts)) - (n_pts + 3) / 4 * 4;
    // 1. the density of every sample (the forward-only kernel: bit-identical to the gradient kernel's own density)
    int rc = nfx_hip_result(prec == NFX_PREC_FP32
                                ? nfx_launch_nerf_sigma_x3(rayo, rayd, z, n_pts, n_samples, geom_blob, sigma, blocks, st)
                                : launch_sigma_bf16(rayo, rayd, z, n_pts, n_samples, geom_blob, sigma, st),
                            "nerf_sigma_grad_rows(density)");
    if (rc) return rc;
    // 2. the samples with a density, ascending; every other sample's output row is final after this pass
    rc = nfx_hip_result(nfx_launch_select_density(sigma, n_pts, normal_sigma, list_ws, st), "nerf_sigma_grad_rows(select)");
    if (rc) return rc;
    // 3. forward + reverse sweep over the listed samples only
    return nfx_hip_result(prec == NFX_PREC_FP32
                              ? nfx_launch_nerf_sigma_grad_x3_list(rayo, rayd, z, n_pts, n_samples, geom_blob, normal_sigma,
                                                                   list, count, blocks, st)
                              : nfx_launch_nerf_sigma_grad_list(rayo, rayd, z, n_pts, n_samples, geom_blob, normal_sigma,
                                                                list, count, blocks, st),
                          "nerf_sigma_grad_rows(gradient)");
}
}  // extern "C"
