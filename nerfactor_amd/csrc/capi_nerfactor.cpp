// capi_nerfactor.cpp — C-ABI entry points of the NeRFactor surface-shading stage (include/nfx.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/nfx.h"
#include "brdf_rows_geom.hpp"
#include "mlp128_layout.hpp"
#include "pack.hpp"

int nfx_fail(int code, const char* fmt, ...);       // capi.cpp
int nfx_hip_result(int e, const char* what);        // capi.cpp
extern "C" int nfx_option_int(const char* name, int dflt);  // capi.cpp

#define REQUIRE(cond, ...) \
    do {                   \
        if (!(cond)) return nfx_fail(NFX_EINVAL, __VA_ARGS__); \
    } while (0)
#define ALIGNED(p, a) ((((uintptr_t)(p)) & ((a)-1)) == 0)

extern "C" {
int nfx_launch_brdf_rows_geom(const nfx::rowsgeom::Args*, int, hipStream_t);
int nfx_launch_mlp128_xyz(const float*, long long, float, const void*, int, int, float, float, float*, int,
                          hipStream_t);
int nfx_launch_lvis_pre(const float*, long long, float, const void*, float*, int, hipStream_t);
int nfx_launch_brdf_spec_v2(const float*, const float*, const float*, const float*, int, const float*, int,
                            const void*, long long, float*, int, int, hipStream_t);
int nfx_launch_brdf_spec_v3(const float*, const float*, const float*, const float*, int, const float*, int,
                            const void*, long long, float*, int, int, int, hipStream_t);
int nfx_launch_lvis_v2(const float*, long long, const float*, int, const float*, const void*, float*, int, int,
                       hipStream_t, const int*, int*);
int nfx_launch_zero_rows(float*, const int*, long long, int, hipStream_t);
int nfx_launch_lvis(const float*, long long, const float*, int, const float*, const void*, float*, int,
                    hipStream_t);
int nfx_launch_brdf_spec(const float*, const float*, const float*, const float*, int, const float*, int,
                         const void*, long long, float*, int, hipStream_t);
int nfx_launch_shade(const float*, const float*, const float*, const float*, const float*, const float*, float,
                     float, const float*, const float*, const float*, const float*, long long, int, int, int,
                     float*, hipStream_t, const int*);
int nfx_launch_shade_olat(const float*, const float*, const float*, const float*, const float*, const float*,
                          float, float, const float*, const float*, const float*, float, float, long long, int,
                          int, float*, hipStream_t, const int*, const int*, int*);
int nfx_launch_dir2rusink(const float*, const float*, long long, float*, hipStream_t);
size_t nfx_shade_olat_lds_bytes(int n_lights);
int nfx_mlp128_x3_weight_bytes(int in_kind);   // mlp128_x3.hip
int nfx_launch_mlp128_x3(int, const float*, const float*, const float*, const float*, const float*, const float*, int,
                         long long, int, float, const void*, int, int, float, float, float*, int, hipStream_t);

// ------------------------------------------------------------------------------ packing
static int in_dims_of(int in_kind, int z_dim) {
    switch (in_kind) {
        case NFX_IN_XYZ: return 63;
        case NFX_IN_XYZ_LDIR: return 90;
        case NFX_IN_Z_RUSINK: return z_dim + 15;
        default: return -1;
    }
}

size_t nfx_mlp128_packed_bytes(int in_kind, int z_dim, int out_dim, int prec) {
    using namespace nfx::m128;
    if (out_dim < 1 || out_dim > 8) return 0;
    if (prec == NFX_PREC_FP32) {   // mlp128_x3.hip: [hi fragments | lo fragments | biases]
        if (in_kind != NFX_IN_XYZ && in_kind != NFX_IN_XYZ_LDIR &&
            !(in_kind == NFX_IN_Z_RUSINK && z_dim >= 1 && z_dim <= kMaxZDim))
            return 0;
        return 2 * (size_t)nfx_mlp128_x3_weight_bytes(in_kind) + kMainBiasFloats * sizeof(float);
    }
    if (prec != NFX_PREC_BF16) return 0;
    if (in_kind == NFX_IN_XYZ) return kMainBytes;
    if (in_kind == NFX_IN_XYZ_LDIR) return (size_t)kPreBytes + kMainBytes;
    if (in_kind == NFX_IN_Z_RUSINK && z_dim >= 1 && z_dim <= kMaxZDim) return kMainBytes;
    return 0;
}

// B-operand slot table of the learned-BRDF input [z(z_dim) | posenc2(rusink)(15)], see
// brdf_spec_kernel in mlp128.hip: [k-step][half][element] -> input row, -1 = zero.
static void brdf_input_slots(int zd, int* slots /*[2][2][8]*/) {
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

// One half (hi or lo) of the NFX_PREC_FP32 blob: the plain five layers, the light-visibility input NOT folded.
static int pack_m128_x3_half(const float* const kernels[5], const float* const biases[5], int in_kind, int z_dim,
                             int out_dim, uint8_t* w, float* b) {
    using namespace nfx::pack;
    const uint8_t* w0 = w;
    const Seg hid{kHidden, 128, 0, nullptr};
    int slots[32];
    brdf_input_slots(z_dim, slots);
    std::vector<Seg> in0, in3{hid};
    int p0 = 4, p3 = 12;
    if (in_kind == NFX_IN_Z_RUSINK) {
        in0 = {Seg{kRaw, 2, 0, slots}};
        in3.push_back(Seg{kRaw, 2, 128, slots});
    } else {
        in0 = {Seg{kPosEnc, 10, 0, nullptr}};
        in3.push_back(Seg{kPosEnc, 10, 128, nullptr});
        if (in_kind == NFX_IN_XYZ_LDIR) {
            in0.push_back(Seg{kPosEnc, 4, 63, nullptr});
            in3.push_back(Seg{kPosEnc, 4, 128 + 63, nullptr});
            p0 = 8;
            p3 = 16;
        }
    }
    w += pack_layer_bf16(in0, {{kernels[0], biases[0], 128}}, 4, p0, w, b);
    w += pack_layer_bf16({hid}, {{kernels[1], biases[1], 128}}, 4, 8, w, b + 128);
    w += pack_layer_bf16({hid}, {{kernels[2], biases[2], 128}}, 4, 8, w, b + 256);
    w += pack_layer_bf16(in3, {{kernels[3], biases[3], 128}}, 4, p3, w, b + 384);
    w += pack_layer_bf16({hid}, {{kernels[4], biases[4], out_dim}}, 1, 8, w, b + 512);
    return w - w0 == nfx_mlp128_x3_weight_bytes(in_kind) ? 0 : 1;
}

int nfx_mlp128_pack_weights(const float* const kernels[5], const float* const biases[5], int in_kind,
                            int z_dim, int out_dim, int prec, void* blob, size_t blob_bytes) {
    using namespace nfx::m128;
    using namespace nfx::pack;
    REQUIRE(kernels && biases && blob, "nfx_mlp128_pack_weights: null argument");
    for (int i = 0; i < 5; ++i) REQUIRE(kernels[i] && biases[i], "nfx_mlp128_pack_weights: layer %d null", i);
    REQUIRE(prec == NFX_PREC_BF16 || prec == NFX_PREC_FP32, "nfx_mlp128_pack_weights: bad prec %d", prec);
    const size_t need = nfx_mlp128_packed_bytes(in_kind, z_dim, out_dim, prec);
    REQUIRE(need != 0, "nfx_mlp128_pack_weights: unsupported configuration (in_kind %d, z_dim %d, out_dim %d)",
            in_kind, z_dim, out_dim);
    REQUIRE(blob_bytes >= need, "nfx_mlp128_pack_weights: blob too small (%zu < %zu)", blob_bytes, need);
    const int in_dims = in_dims_of(in_kind, z_dim);
    uint8_t* w = static_cast<uint8_t*>(blob);
    if (prec == NFX_PREC_FP32) {
        static const int out_cols[5] = {128, 128, 128, 128, 0};
        std::vector<std::vector<float>> lo(5);
        const float* lo_ptr[5];
        for (int i = 0; i < 5; ++i) {
            const int rows = i == 0 ? in_dims : i == 3 ? 128 + in_dims : 128;
            const size_t cnt = (size_t)rows * (i == 4 ? out_dim : out_cols[i]);
            lo[i].resize(cnt);
            for (size_t k = 0; k < cnt; ++k) {
                const uint32_t bits = (uint32_t)f32_to_bf16_rne(kernels[i][k]) << 16;
                float hi;
                memcpy(&hi, &bits, 4);
                lo[i][k] = kernels[i][k] - hi;
            }
            lo_ptr[i] = lo[i].data();
        }
        const size_t wb = nfx_mlp128_x3_weight_bytes(in_kind);
        std::vector<float> sink(kMainBiasFloats);
        memset(w, 0, need);
        if (pack_m128_x3_half(kernels, biases, in_kind, z_dim, out_dim, w, reinterpret_cast<float*>(w + 2 * wb)) ||
            pack_m128_x3_half(lo_ptr, biases, in_kind, z_dim, out_dim, w + wb, sink.data()))
            return nfx_fail(NFX_EINVAL, "nfx_mlp128_pack_weights: layout mismatch (fp32)");
        return NFX_OK;
    }
    const Seg hid{kHidden, 128, 0, nullptr};
    int slots[32];
    brdf_input_slots(z_dim, slots);
    if (in_kind == NFX_IN_XYZ_LDIR) {
        float* pb = reinterpret_cast<float*>(w + kPreWeightBytes);
        w += pack_layer_bf16({Seg{kPosEnc, 10, 0, nullptr}}, {{kernels[0], biases[0], 128}}, 4, 4, w, pb);
        w += pack_layer_bf16({Seg{kPosEnc, 10, 128, nullptr}}, {{kernels[3], biases[3], 128}}, 4, 4, w,
                             pb + 128);
        w += kPreBiasFloats * 4;
    }
    uint8_t* main0 = w;
    float* b = reinterpret_cast<float*>(main0 + kMainWeightBytes);
    Seg in0, in3;
    const float *bias0 = biases[0], *bias3 = biases[3];
    if (in_kind == NFX_IN_XYZ) {
        in0 = Seg{kPosEnc, 10, 0, nullptr};
        in3 = Seg{kPosEnc, 10, 128, nullptr};
    } else if (in_kind == NFX_IN_XYZ_LDIR) {
        in0 = Seg{kPosEnc, 4, 63, nullptr};      // rows 63..89 = posenc4(ldir)
        in3 = Seg{kPosEnc, 4, 128 + 63, nullptr};
        bias0 = bias3 = nullptr;                 // folded into the per-point pre-activation
    } else {
        in0 = Seg{kRaw, 2, 0, slots};
        in3 = Seg{kRaw, 2, 128, slots};
    }
    (void)in_dims;
    w += pack_layer_bf16({in0}, {{kernels[0], bias0, 128}}, 4, 4, w, b);
    w += pack_layer_bf16({hid}, {{kernels[1], biases[1], 128}}, 4, 8, w, b + 128);
    w += pack_layer_bf16({hid}, {{kernels[2], biases[2], 128}}, 4, 8, w, b + 256);
    w += pack_layer_bf16({hid, in3}, {{kernels[3], bias3, 128}}, 4, 12, w, b + 384);
    w += pack_layer_bf16({hid}, {{kernels[4], biases[4], out_dim}}, 1, 8, w, b + 512);
    if (w != main0 + kMainWeightBytes) return nfx_fail(NFX_EINVAL, "nfx_mlp128_pack_weights: layout mismatch");
    return NFX_OK;
}

// -------------------------------------------------------------------------------- MLPs
int nfx_mlp128_xyz_fwd(const float* xyz, int64_t n, float xyz_scale, const void* blob, int out_dim,
                       int out_act, float post_scale, float post_bias, int prec, float* out, void* stream) {
    REQUIRE(n >= 0, "nfx_mlp128_xyz_fwd: n < 0");
    REQUIRE(out_dim >= 1 && out_dim <= 8, "nfx_mlp128_xyz_fwd: out_dim %d not in [1, 8]", out_dim);
    REQUIRE(out_act >= 0 && out_act <= 3, "nfx_mlp128_xyz_fwd: bad activation %d", out_act);
    REQUIRE(prec == NFX_PREC_BF16 || prec == NFX_PREC_FP32, "nfx_mlp128_xyz_fwd: bad prec %d", prec);
    if (n == 0) return NFX_OK;
    REQUIRE(xyz && blob && out, "nfx_mlp128_xyz_fwd: null pointer");
    if (!ALIGNED(blob, 16)) return nfx_fail(NFX_EALIGN, "nfx_mlp128_xyz_fwd: blob must be 16-byte aligned");
    if (prec == NFX_PREC_FP32)
        return nfx_hip_result(nfx_launch_mlp128_x3(NFX_IN_XYZ, xyz, nullptr, nullptr, nullptr, nullptr, nullptr, 0, n, 1,
                                                   xyz_scale, blob, out_dim, out_act, post_scale, post_bias, out,
                                                   nfx_option_int("m128_blocks", 256), (hipStream_t)stream),
                              "mlp128_xyz_fwd(fp32)");
    return nfx_hip_result(nfx_launch_mlp128_xyz(xyz, n, xyz_scale, blob, out_dim, out_act, post_scale, post_bias,
                                                out, nfx_option_int("m128_blocks", 256), (hipStream_t)stream),
                          "mlp128_xyz_fwd");
}

size_t nfx_lvis_workspace_bytes(int64_t n) { return n > 0 ? (size_t)n * 256 * sizeof(float) : 0; }

int nfx_lvis_fwd(const float* xyz, const float* xyz_dir, int64_t n, float xyz_scale, const float* lxyz,
                 int n_lights, const void* blob, int prec, void* workspace, size_t workspace_bytes,
                 float* lvis, void* stream) {
    return nfx_lvis_fwd_rows(xyz, xyz_dir, n, xyz_scale, lxyz, n_lights, blob, prec, workspace, workspace_bytes, nullptr,
                             lvis, nullptr, stream);
}

int nfx_zero_rows(float* dst, const int32_t* row_of, int64_t n_all, int d, void* stream) {
    REQUIRE(n_all >= 0 && d >= 1, "nfx_zero_rows: bad shape");
    if (n_all == 0) return NFX_OK;
    REQUIRE(dst && row_of, "nfx_zero_rows: null pointer");
    REQUIRE(d % 4 == 0 && ALIGNED(dst, 16), "nfx_zero_rows: rows of a multiple of 4 floats, 16-byte aligned (d = %d)", d);
    return nfx_hip_result(nfx_launch_zero_rows(dst, row_of, (long long)n_all, d, (hipStream_t)stream), "zero_rows");
}

int nfx_lvis_fwd_rows(const float* xyz, const float* xyz_dir, int64_t n, float xyz_scale, const float* lxyz,
                      int n_lights, const void* blob, int prec, void* workspace, size_t workspace_bytes,
                      const int32_t* out_row, float* lvis, int* nan_flag, void* stream) {
    using namespace nfx::m128;
    REQUIRE(n >= 0, "nfx_lvis_fwd: n < 0");
    REQUIRE(n_lights > 0 && n_lights % 32 == 0, "nfx_lvis_fwd: n_lights (%d) must be a positive multiple of 32",
            n_lights);
    REQUIRE(prec == NFX_PREC_BF16 || prec == NFX_PREC_FP32, "nfx_lvis_fwd: bad prec %d", prec);
    if (n == 0) return NFX_OK;
    const int variant = nfx_option_int("lvis_variant", 8);   // 8 (default) = 8 waves (two per SIMD) x 2 column tiles
    REQUIRE(out_row || !nan_flag, "nfx_lvis_fwd_rows: the NaN flag comes with output rows (pass the identity for a compact output)");
    if (out_row && (prec != NFX_PREC_BF16 || !((variant >= 2 && variant <= 4) || variant == 8)))
        return nfx_fail(NFX_ENOSUP, "nfx_lvis_fwd_rows: output rows / NaN flag exist for the bf16 kernels with the network resident "
                                    "in LDS (lvis_variant 8 | 2 | 3 | 4), not for prec %d / variant %d", prec, variant);
    if (prec == NFX_PREC_FP32) {   // no per-point fold, no workspace
        REQUIRE(xyz && lxyz && blob && lvis, "nfx_lvis_fwd: null pointer");
        if (!ALIGNED(blob, 16)) return nfx_fail(NFX_EALIGN, "nfx_lvis_fwd: blob must be 16-byte aligned");
        return nfx_hip_result(nfx_launch_mlp128_x3(NFX_IN_XYZ_LDIR, xyz, xyz_dir ? xyz_dir : xyz, lxyz, nullptr, nullptr,
                                                   nullptr, 0, n, n_lights, xyz_scale, blob, 1, 2, 1.0f, 0.0f, lvis,
                                                   nfx_option_int("m128_blocks", 256), (hipStream_t)stream),
                              "lvis_fwd(fp32)");
    }
    REQUIRE(xyz && lxyz && blob && lvis && workspace, "nfx_lvis_fwd: null pointer");
    REQUIRE(workspace_bytes >= nfx_lvis_workspace_bytes(n), "nfx_lvis_fwd: workspace too small (%zu < %zu)",
            workspace_bytes, nfx_lvis_workspace_bytes(n));
    if (!ALIGNED(blob, 16) || !ALIGNED(workspace, 16))
        return nfx_fail(NFX_EALIGN, "nfx_lvis_fwd: blob and workspace must be 16-byte aligned");
    const int blocks = nfx_option_int("m128_blocks", 256);
    const char* b = static_cast<const char*>(blob);
    float* pre = static_cast<float*>(workspace);
    int rc = nfx_hip_result(nfx_launch_lvis_pre(xyz, n, xyz_scale, b, pre, blocks, (hipStream_t)stream), "lvis_pre");
    if (rc) return rc;
    // option lvis_variant: 0 = 8 waves x 32 rows with streamed weights (mlp128.hip); 2 | 3 | 4 = network resident in LDS,
    // one wave per SIMD with that many 32-row column tiles (lvis_v2.hip)
    if ((variant >= 2 && variant <= 4) || variant == 8)
        return nfx_hip_result(nfx_launch_lvis_v2(xyz_dir ? xyz_dir : xyz, n, lxyz, n_lights, pre, b + kPreBytes, lvis,
                                                 variant, blocks, (hipStream_t)stream, out_row, nan_flag),
                              "lvis_fwd(v2)");
    return nfx_hip_result(
        nfx_launch_lvis(xyz_dir ? xyz_dir : xyz, n, lxyz, n_lights, pre, b + kPreBytes, lvis, blocks,
                        (hipStream_t)stream),
        "lvis_fwd");
}

int nfx_brdf_spec_fwd(const float* xyz, const float* cam, const float* normal, const float* z, int z_dim,
                      const float* lxyz, int n_lights, const void* blob, int prec, int64_t n, float* spec,
                      void* stream) {
    REQUIRE(n >= 0, "nfx_brdf_spec_fwd: n < 0");
    REQUIRE(z_dim >= 1 && z_dim <= nfx::m128::kMaxZDim, "nfx_brdf_spec_fwd: z_dim %d not in [1, %d]", z_dim,
            nfx::m128::kMaxZDim);
    REQUIRE(n_lights > 0 && n_lights % 32 == 0, "nfx_brdf_spec_fwd: n_lights (%d) must be a multiple of 32",
            n_lights);
    REQUIRE(prec == NFX_PREC_BF16 || prec == NFX_PREC_FP32, "nfx_brdf_spec_fwd: bad prec %d", prec);
    if (n == 0) return NFX_OK;
    REQUIRE(xyz && cam && normal && z && lxyz && blob && spec, "nfx_brdf_spec_fwd: null pointer");
    if (!ALIGNED(blob, 16)) return nfx_fail(NFX_EALIGN, "nfx_brdf_spec_fwd: blob must be 16-byte aligned");
    if (prec == NFX_PREC_FP32)
        return nfx_hip_result(nfx_launch_mlp128_x3(NFX_IN_Z_RUSINK, xyz, nullptr, lxyz, cam, normal, z, z_dim, n, n_lights,
                                                   1.0f, blob, 1, 3, 1.0f, 0.0f, spec,
                                                   nfx_option_int("m128_blocks", 256), (hipStream_t)stream),
                              "brdf_spec_fwd(fp32)");
    // option brdf_variant: 0 / 2 / 3 / 4 as lvis_variant (every row evaluated, back-lit rows zeroed afterwards);
    // 5 = front-lit rows only (LDS row queue per wave), per-row geometry as in the dense kernels (bit-identical);
    // 6 (default) = 5 with closed-form Rusinkiewicz angles.  Option brdf_ct of variants 5 / 6: 2 | 3 | 4 (default) column
    // tiles per wave, one wave per SIMD.  8 = 8 waves x 2 column tiles, two waves per SIMD, variant 6 only and NOT the
    // default: r03 shipped it (12 % faster), and its per-row-geometry sibling <2, 0, 8> then failed bit identity on a
    // fresh MI355X for a reason that is still not established (profiles/HISTORY.md section 2c) — a kernel form whose sibling's
    // bits depend on the box is opt-in until the mechanism is known.
    int variant = nfx_option_int("brdf_variant", 6);
    if (variant >= 5) {
        int ct = nfx_option_int("brdf_ct", 4);
#ifndef NFX_EXPERIMENT_BUILD
        if (ct == 8 && variant != 6) ct = 2;   // <2, 0, 8> is not built
#endif
        int rc = nfx_launch_brdf_spec_v3(xyz, cam, normal, z, z_dim, lxyz, n_lights, blob, n, spec, ct, variant == 6,
                                         nfx_option_int("m128_blocks", 256), (hipStream_t)stream);
        if (rc == -1 && ct == 8)   // more lights than the 8-wave row queues hold: the 4-wave form
            rc = nfx_launch_brdf_spec_v3(xyz, cam, normal, z, z_dim, lxyz, n_lights, blob, n, spec, 4, variant == 6,
                                         nfx_option_int("m128_blocks", 256), (hipStream_t)stream);
        if (rc != -1) return nfx_hip_result(rc, "brdf_spec_fwd(v3)");
        variant = 3;   // shape outside the row queue's limits: dense kernel
    }
    if (variant >= 2 && variant <= 4)
        return nfx_hip_result(nfx_launch_brdf_spec_v2(xyz, cam, normal, z, z_dim, lxyz, n_lights, blob, n, spec, variant,
                                                      nfx_option_int("m128_blocks", 256), (hipStream_t)stream),
                              "brdf_spec_fwd(v2)");
    return nfx_hip_result(nfx_launch_brdf_spec(xyz, cam, normal, z, z_dim, lxyz, n_lights, blob, n, spec,
                                               nfx_option_int("m128_blocks", 256), (hipStream_t)stream),
                          "brdf_spec_fwd");
}

// ------------------------------------------------------------------------------ shading
static int check_shade(const char* who, const float* xyz, const float* cam, const float* normal,
                       const float* albedo, const float* rough, const float* spec, const float* lvis,
                       const float* lxyz, const float* lareas, int64_t n, int n_lights) {
    REQUIRE(n >= 0 && n_lights > 0, "%s: bad shape", who);
    if (n == 0) return NFX_OK;
    REQUIRE(xyz && cam && normal && albedo && lvis && lxyz && lareas, "%s: null pointer", who);
    REQUIRE(rough || spec, "%s: need either roughness (microfacet) or a specular term (learned BRDF)", who);
    return NFX_OK;
}

int nfx_shade_fwd(const float* xyz, const float* cam, const float* normal, const float* albedo,
                  const float* rough, const float* spec, float spec_scale, float f0, const float* lvis,
                  const float* lxyz, const float* lareas, const float* lights, int64_t n, int n_lights,
                  int n_probes, int linear2srgb, float* rgb, void* stream) {
    return nfx_shade_fwd_rows(xyz, cam, normal, albedo, rough, spec, spec_scale, f0, lvis, nullptr, lxyz, lareas, lights, n,
                              n_lights, n_probes, linear2srgb, rgb, stream);
}
int nfx_shade_fwd_rows(const float* xyz, const float* cam, const float* normal, const float* albedo,
                       const float* rough, const float* spec, float spec_scale, float f0, const float* lvis,
                       const int32_t* lvis_row, const float* lxyz, const float* lareas, const float* lights, int64_t n,
                       int n_lights, int n_probes, int linear2srgb, float* rgb, void* stream) {
    int rc = check_shade("nfx_shade_fwd", xyz, cam, normal, albedo, rough, spec, lvis, lxyz, lareas, n, n_lights);
    if (rc) return rc;
    REQUIRE(n_probes >= 1, "nfx_shade_fwd: n_probes must be >= 1");
    REQUIRE(nfx_shade_lds_bytes(n_lights, n_probes) <= 160 * 1024,
            "nfx_shade_fwd: %d probes x %d lights do not fit the 160 KiB LDS; split the probes", n_probes,
            n_lights);
    if (n == 0) return NFX_OK;
    REQUIRE(lights && rgb, "nfx_shade_fwd: null pointer");
    return nfx_hip_result(nfx_launch_shade(xyz, cam, normal, albedo, rough, spec, spec_scale, f0, lvis, lxyz,
                                           lareas, lights, n, n_lights, n_probes, linear2srgb, rgb,
                                           (hipStream_t)stream, lvis_row),
                          "shade_fwd");
}

int nfx_shade_olat_fwd(const float* xyz, const float* cam, const float* normal, const float* albedo,
                       const float* rough, const float* spec, float spec_scale, float f0, const float* lvis,
                       const float* lxyz, const float* lareas, float olat_inten, float ambient, int64_t n,
                       int n_lights, int linear2srgb, float* rgb_olat, void* stream) {
    return nfx_shade_olat_fwd_rows(xyz, cam, normal, albedo, rough, spec, spec_scale, f0, lvis, nullptr, lxyz, lareas,
                                   olat_inten, ambient, n, n_lights, linear2srgb, nullptr, rgb_olat, nullptr, stream);
}
int nfx_shade_olat_fwd_rows(const float* xyz, const float* cam, const float* normal, const float* albedo,
                            const float* rough, const float* spec, float spec_scale, float f0, const float* lvis,
                            const int32_t* lvis_row, const float* lxyz, const float* lareas, float olat_inten,
                            float ambient, int64_t n, int n_lights, int linear2srgb, const int32_t* out_row,
                            float* rgb_olat, int* nan_flag, void* stream) {
    int rc =
        check_shade("nfx_shade_olat_fwd", xyz, cam, normal, albedo, rough, spec, lvis, lxyz, lareas, n, n_lights);
    if (rc) return rc;
    REQUIRE(nfx_shade_olat_lds_bytes(n_lights) <= 160 * 1024, "nfx_shade_olat_fwd: too many lights (%d)",
            n_lights);
    if (n == 0) return NFX_OK;
    REQUIRE(rgb_olat, "nfx_shade_olat_fwd: null output");
    return nfx_hip_result(nfx_launch_shade_olat(xyz, cam, normal, albedo, rough, spec, spec_scale, f0, lvis, lxyz,
                                                lareas, olat_inten, ambient, n, n_lights, linear2srgb, rgb_olat,
                                                (hipStream_t)stream, lvis_row, out_row, nan_flag),
                          "shade_olat_fwd");
}

static int rows_geom_check(const char* who, int64_t n, int n_lights, int z_dim, int n_freqs, int ld) {
    REQUIRE(n >= 0 && n_lights >= 1, "%s: bad sizes", who);
    if (z_dim < 1 || z_dim > nfx::rowsgeom::kMaxZ) return nfx_fail(NFX_ENOSUP, "%s: z_dim %d (1 .. %d)", who, z_dim, nfx::rowsgeom::kMaxZ);
    if (n_freqs < 0 || n_freqs > nfx::rowsgeom::kMaxFreqs) return nfx_fail(NFX_ENOSUP, "%s: %d bands (0 .. %d)", who, n_freqs, nfx::rowsgeom::kMaxFreqs);
    REQUIRE(ld >= z_dim + 3 + 6 * n_freqs, "%s: rows of %d floats, %d needed", who, ld, z_dim + 3 + 6 * n_freqs);
    return NFX_OK;
}
int nfx_brdf_rows_geom_fwd(const float* xyz, const float* cam, const float* normal, const float* z, int z_dim, const float* lxyz,
                           int n_lights, int64_t n, int n_freqs, float* rows, int ld_rows, float* front, void* stream) {
    if (int e = rows_geom_check("nfx_brdf_rows_geom_fwd", n, n_lights, z_dim, n_freqs, ld_rows)) return e;
    if (n == 0) return NFX_OK;
    REQUIRE(xyz && cam && normal && z && lxyz && rows && front, "nfx_brdf_rows_geom_fwd: null pointer");
    nfx::rowsgeom::Args a{xyz, cam, normal, z, lxyz, n, n_lights, z_dim, n_freqs, rows, front, nullptr, ld_rows, nullptr, nullptr};
    return nfx_hip_result(nfx_launch_brdf_rows_geom(&a, 0, (hipStream_t)stream), "brdf_rows_geom_fwd");
}
int nfx_brdf_rows_geom_bwd(const float* xyz, const float* cam, const float* normal, int z_dim, const float* lxyz, int n_lights,
                           int64_t n, int n_freqs, const float* d_rows, int ld_rows, float* d_normal, float* d_z, void* stream) {
    if (int e = rows_geom_check("nfx_brdf_rows_geom_bwd", n, n_lights, z_dim, n_freqs, ld_rows)) return e;
    if (n == 0) return NFX_OK;
    REQUIRE(xyz && cam && normal && lxyz && d_rows && d_normal && d_z, "nfx_brdf_rows_geom_bwd: null pointer");
    nfx::rowsgeom::Args a{xyz, cam, normal, nullptr, lxyz, n, n_lights, z_dim, n_freqs, nullptr, nullptr, d_rows, ld_rows, d_normal, d_z};
    return nfx_hip_result(nfx_launch_brdf_rows_geom(&a, 1, (hipStream_t)stream), "brdf_rows_geom_bwd");
}

int nfx_dir2rusink(const float* a, const float* b, int64_t n, float* rusink, void* stream) {
    REQUIRE(n >= 0 && (n == 0 || (a && b && rusink)), "nfx_dir2rusink: bad arguments");
    return nfx_hip_result(nfx_launch_dir2rusink(a, b, n, rusink, (hipStream_t)stream), "dir2rusink");
}

}  // extern "C"
