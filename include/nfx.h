/*
 * nfx.h — C-ABI of libnfx.so: the MI355X (gfx950) implementation of NeRFactor's
 * per-ray rendering hot path.
 *
 * Every entry point replaces a cluster of TensorFlow ops in google/nerfactor; the
 * reference site is cited per function as `file:line` relative to the reference
 * tree.  Conventions (all entry points):
 *
 *   - plain pointers and sizes only; every pointer marked `dev` is DEVICE memory
 *     owned by the caller (the host framework's allocator).  The library never
 *     allocates, frees or synchronises device memory and enqueues work only on
 *     the `stream` argument (a hipStream_t passed as void*; NULL = default stream).
 *   - return value: NFX_OK (0) or a negative NFX_E* code; nfx_last_error() gives a
 *     thread-local human-readable message for the last failure.
 *   - all floating-point tensors are fp32, C-contiguous, ray-major, exactly the
 *     layouts of the reference's flattened batch tuples
 *     (nerfactor/datasets/nerf.py:96-104, nerfactor/datasets/nerf_shape.py:72-82).
 *   - `prec`: NFX_PREC_BF16 = bf16 operands / fp32 accumulate on the MFMA path;
 *     NFX_PREC_FP32 = fp32-class accuracy on the same pipe: every operand a bf16 hi/lo
 *     pair (16 significant bits), three MFMAs per product, fp32 accumulate (stated
 *     tolerance 2e-4 on rgb; ~2x the v_mfma_f32_32x32x2_f32 peak).  Built for the
 *     forward kernels: nfx_nerf_pack_weights / nfx_nerf_mlp_fwd and nfx_mlp128_pack_weights /
 *     nfx_mlp128_xyz_fwd / nfx_lvis_fwd (workspace unused) / nfx_brdf_spec_fwd, and the
 *     geometry pair nfx_nerf_pack_geom_weights / nfx_nerf_sigma_fwd / nfx_nerf_sigma_grad; the
 *     tuned training-blob and backward entry points return NFX_ENOSUP for it.  The runtime-shaped
 *     family (nfx_mlp_generic_*) takes it forward AND backward (fp32 activations, gradients and
 *     workspace; hi/lo pairs only as MFMA operands) — the backward of `precision = fp32` — and
 *     additionally NFX_PREC_FP32_NATIVE: true fp32 operands on the fp32 matrix instruction.
 *   - re-entrant: no global mutable state besides the option table of nfx_set_option
 *     (atomic integers); concurrent calls on different streams are legal.
 */
#ifndef NFX_H_
#define NFX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Exported symbols: the library is built with -fvisibility=hidden; exactly the functions declared here with NFX_API
 * are visible to the dynamic linker (tests/test_cpu_capi_exports.py compares `nm -D` with this header). */
#if defined(__GNUC__)
#define NFX_API __attribute__((visibility("default")))
#else
#define NFX_API
#endif

#define NFX_OK 0
#define NFX_EINVAL (-1)   /* bad shape / null pointer / unsupported dimension */
#define NFX_EALIGN (-2)   /* pointer not aligned as required                  */
#define NFX_EHIP (-3)     /* a HIP runtime call failed (see nfx_last_error)    */
#define NFX_ENOSUP (-4)   /* configuration not supported by this build         */

#define NFX_PREC_BF16 0
#define NFX_PREC_FP32 1
#define NFX_PREC_FP32_NATIVE 2 /* nfx_mlp_generic_* only: fp32 operands, v_mfma_f32_32x32x2_f32 */

#define NFX_ACT_NONE 0
#define NFX_ACT_RELU 1
#define NFX_ACT_SIGMOID 2
#define NFX_ACT_SOFTPLUS 3

NFX_API int nfx_version(void);
/* Copies the calling thread's last error message (NUL-terminated) into buf. */
NFX_API int nfx_last_error(char *buf, size_t len);

/* Process-wide integer options: kernel-variant selectors kept for A/B measurements and the identity tests, and the
 * persistent-grid sizes.  The library reads NO environment variable; a host sets what it wants before the calls it
 * should affect (set / unset are atomic, but not ordered against calls in flight on other threads).  Keys:
 *   nerf_variant (7)  nerf_blocks (256)  m128_blocks (256)  lvis_variant (8)  brdf_variant (6)  brdf_ct (4)
 *   nerf_bwd (1)  nerf_bwd_nw (8)  m128_bwd (1)  wgrad_lds / wgrad_slabs / wgrad_narrow (by row count)
 *   wgrad_fused (1)  wgrad_map (1)  wgrad_splits (256)
 * (defaults in parentheses; every variant of a selector computes the same function, most of them bit-identically —
 * DESIGN.md).  An unknown key is NFX_EINVAL.  nfx_unset_option returns a key to its default. */
NFX_API int nfx_set_option(const char *key, int value);
NFX_API int nfx_unset_option(const char *key);
NFX_API int nfx_get_option(const char *key, int *value, int *is_set);

/* ------------------------------------------------------------------------ */
/* Weight packing (host side, no GPU needed).                                */
/* Keras Dense layout in: kernel [in, out] row-major fp32, bias [out]        */
/* (nerfactor/networks/mlp.py:32-36).  Out: an opaque blob of MFMA A-operand  */
/* fragments in consumption order + permuted biases, to be copied to the      */
/* device verbatim.                                                           */
/* ------------------------------------------------------------------------ */

/* NeRF net of nerfactor/models/nerf.py:53-71 at the shipped configuration
 * (config/nerf.ini:55-70): enc = 8x Dense(256, relu) with the input re-concatenated
 * after layer 4 (mlp.py:47-48 puts y first: [y, x]); sigma_out Dense(1);
 * bottleneck Dense(256); rgb_out Dense(128, relu) -> Dense(3).
 * kernels[i]/biases[i], i = 0..7 enc layers, 8 sigma_out, 9 bottleneck,
 * 10 rgb_out[0], 11 rgb_out[1].  n_freqs_xyz = 10, n_freqs_view = 4.        */
NFX_API size_t nfx_nerf_packed_bytes(int prec);
NFX_API int nfx_nerf_pack_weights(const float *const kernels[12], const float *const biases[12],
                          int prec, void *blob, size_t blob_bytes);

/* Width-128 surface MLP of nerfactor/models/shape.py:79-94 and
 * nerfactor/models/nerfactor.py:128-143, models/brdf.py:57-66:
 * `mlp` = 4x Dense(128, relu), input re-concatenated after layer 2 ([y, x]),
 * followed by `out` = Dense(out_dim, act).
 * in_kind selects the input encoding the kernel computes on the fly:
 *   NFX_IN_XYZ        posenc10(xyz)                       63 dims (normal/albedo/brdf_z)
 *   NFX_IN_XYZ_LDIR   [posenc10(xyz), posenc4(ldir)]      90 dims (light visibility)
 *   NFX_IN_Z_RUSINK   [z(z_dim), posenc2(rusink)]         z_dim+15 dims (learned BRDF)
 * kernels[0..3] the mlp layers, kernels[4] the out layer.                    */
#define NFX_IN_XYZ 0
#define NFX_IN_XYZ_LDIR 1
#define NFX_IN_Z_RUSINK 2
NFX_API size_t nfx_mlp128_packed_bytes(int in_kind, int z_dim, int out_dim, int prec);
NFX_API int nfx_mlp128_pack_weights(const float *const kernels[5], const float *const biases[5],
                            int in_kind, int z_dim, int out_dim, int prec, void *blob,
                            size_t blob_bytes);

/* ------------------------------------------------------------------------ */
/* NeRF ray marching (nerfactor/models/nerf.py).                              */
/* ------------------------------------------------------------------------ */

/* rayd <- rayd * rsqrt(max(sum(rayd^2), eps))   (nerf.py:157, tf.linalg.l2_normalize,
 * eps = 1e-12; shape.py:131,140 use eps = 1e-6 via util/math.py:63-64).      */
NFX_API int nfx_l2_normalize3(const float *dev_in, float *dev_out, int64_t n, float eps, void *stream);

/* tf.debugging.check_numerics (nerfactor.py:205-233, shape.py:222-232, ...) in one pass: ORs 1 into *dev_flag (an
 * int32 the caller zeroed) if any of the n floats is Inf or NaN.  dev_x must be 16-byte aligned.               */
NFX_API int nfx_any_nonfinite(const float *dev_x, int64_t n, int *dev_flag, void *stream);

/* tf.scatter_nd of the alpha > 0 rows into a zero tensor (nerfactor.py:295-306, shape.py:171-176) in one pass:
 * dev_dst[i, :] = dev_src[dev_row_of[i], :] where dev_row_of[i] >= 0, zeros elsewhere.  dev_src [m, d], dev_row_of
 * [n_all] int32 (compact row of every full row, or -1), dev_dst [n_all, d].  Every output element is written.  */
NFX_API int nfx_scatter_rows(const float *dev_src, const int32_t *dev_row_of, int64_t n_all, int d, float *dev_dst,
                     void *stream);

/* Stratified depths, Model.gen_z (nerf.py:120-136).  z [n_rays, n_samples].
 * dev_u: NULL (no perturbation) or uniform [0,1) randoms [n_rays, n_samples]
 * drawn by the caller (the reference draws them with tf.random.uniform).     */
NFX_API int nfx_gen_z(float near, float far, int n_samples, int64_t n_rays, int lin_in_disp,
              const float *dev_u, float *dev_z, void *stream);

/* Fused point generation + positional encoding + NeRF MLP,
 * Model._eval_nerf_at (nerf.py:256-290) with pts = rayo + rayd * z (nerf.py:162-164)
 * and views = rayd never materialised.  rayd must already be normalised.
 * out: rgbs [n_rays, n_samples, 4] = (raw rgb, raw sigma) exactly as nerf.py:282. */
NFX_API int nfx_nerf_mlp_fwd(const float *dev_rayo, const float *dev_rayd, const float *dev_z,
                     int64_t n_rays, int n_samples, const void *dev_blob, int prec,
                     float *dev_rgbs, void *stream);

/* Volumetric compositing, Model.accumulate_sigma + Model._accumulate
 * (nerf.py:184-254; util/math.py:67-68 safe_cumprod; util/img.py:76-95 alpha_blend).
 * dev_noise: NULL or N(0,1)*noise_std already scaled, [n_rays, n_samples].
 * Outputs (any may be NULL): rgb [n,3] (already blended onto bg = white_bg ? 1 : 0),
 * occu [n], depth [n], disp [n], weights [n, n_samples].                     */
NFX_API int nfx_composite_fwd(const float *dev_rgbs, const float *dev_z, const float *dev_rayd,
                      const float *dev_noise, int64_t n_rays, int n_samples, int white_bg,
                      float *dev_rgb, float *dev_occu, float *dev_depth, float *dev_disp,
                      float *dev_weights, void *stream);

/* Hierarchical re-sampling, Model.gen_z_fine + inv_transform_sample
 * (nerf.py:138-147; util/math.py:71-94): pdf over weights[:,1:-1], cdf,
 * searchsorted(side='right'), lerp, then sort(concat(z_coarse, z_fine)).
 * dev_u: NULL => deterministic u = linspace(0,1,n_fine); else [n_rays, n_fine].
 * out: z_all [n_rays, n_coarse + n_fine] ascending.                          */
NFX_API int nfx_sample_fine(const float *dev_z, const float *dev_weights, int64_t n_rays, int n_coarse,
                    int n_fine, const float *dev_u, float *dev_z_all, void *stream);

/* ------------------------------------------------------------------------ */
/* NeRFactor surface shading (nerfactor/models/{shape,nerfactor,*_microfacet}.py). */
/* ------------------------------------------------------------------------ */

/* Width-128 MLP on posenc10(xyz_scale * xyz): _pred_normal_at (shape.py:196-211),
 * _pred_albedo_at (nerfactor.py:377-396), _pred_brdf_at (nerfactor.py:398-411).
 * out [n, out_dim] = post_scale * act(out(mlp(pe))) + post_bias.             */
NFX_API int nfx_mlp128_xyz_fwd(const float *dev_xyz, int64_t n, float xyz_scale, const void *dev_blob,
                       int out_dim, int out_act, float post_scale, float post_bias, int prec,
                       float *dev_out, void *stream);

/* Light-visibility MLP over the light sphere, _pred_lvis_at (shape.py:213-237) with the
 * light directions of _calc_ldir (shape.py:128-135) recomputed in registers:
 * lvis[n, l] = sigmoid(out(mlp([pe10(xyz_scale*xyz_n), pe4(normalize(lxyz_l - xyz_dir_n))]))).
 * dev_xyz_dir: the points the DIRECTIONS are taken from; NULL = dev_xyz.  (The reference's
 * smoothness term evaluates the MLP at jittered points but keeps the directions of the
 * un-jittered ones: nerfactor.py:195,226.)  dev_lxyz [n_lights, 3].  out [n, n_lights].   */
/* The posenc(xyz) rows of layers 0 and 3 are evaluated once per point into a caller-provided
 * workspace of nfx_lvis_workspace_bytes(n) bytes (16-byte aligned); n_lights % 32 == 0.      */
NFX_API size_t nfx_lvis_workspace_bytes(int64_t n);
/* nfx_lvis_fwd_rows (round 6) = nfx_lvis_fwd with the tf.scatter_nd of shape.py:171-176 and the tf.debugging.check_numerics
 * of shape.py:222-232 done by the kernel's own stores: dev_out_row [n] int32 (or NULL) = the row of point i in dev_lvis,
 * which is then the caller's FULL [n_all, n_lights] buffer (background rows: nfx_zero_rows); dev_nan_flag (or NULL) = an int32
 * the caller zeroed, 1 is OR-ed into it when a visibility is NaN.  bf16 kernels with the network resident in LDS only
 * (lvis_variant 8 | 2 | 3 | 4): NFX_ENOSUP otherwise.  nfx_zero_rows: dev_dst[i, :] = 0 where dev_row_of[i] < 0
 * (dev_row_of [n_all] int32 as for nfx_scatter_rows; d % 4 == 0, 16-byte aligned); other rows are left alone.       */
NFX_API int nfx_lvis_fwd_rows(const float *dev_xyz, const float *dev_xyz_dir, int64_t n, float xyz_scale,
                      const float *dev_lxyz, int n_lights, const void *dev_blob, int prec, void *dev_workspace,
                      size_t workspace_bytes, const int32_t *dev_out_row, float *dev_lvis, int *dev_nan_flag,
                      void *stream);
NFX_API int nfx_zero_rows(float *dev_dst, const int32_t *dev_row_of, int64_t n_all, int d, void *stream);
NFX_API int nfx_lvis_fwd(const float *dev_xyz, const float *dev_xyz_dir, int64_t n, float xyz_scale,
                 const float *dev_lxyz, int n_lights, const void *dev_blob, int prec,
                 void *dev_workspace, size_t workspace_bytes, float *dev_lvis, void *stream);

/* Fused shading integral, Model._render.integrate (nerfactor.py:315-365) with the
 * analytic microfacet BRDF of brdf/microfacet/microfacet.py:30-111 evaluated in
 * registers (nerfactor_microfacet.py:116-124), directions per shape.py:128-144:
 *   rgb[n,p,c] = tonemap( sum_l brdf[n,l,c] * lvis[n,l]*[cos>0] * cos[n,l] * area[l] * light[p,l,c] )
 * tonemap = clip(0,1) then optional linear2srgb (util/img.py:140-163).
 * dev_spec: NULL => GGX microfacet with roughness dev_rough [n] and Fresnel f0;
 *           else achromatic specular term [n, n_lights] (learned BRDF,
 *           nerfactor.py:453-460) scaled by spec_scale; brdf = albedo/pi + spec.
 * dev_lights [n_probes, n_lights, 3]: probe 0 is usually the trained light; the
 * relighting loops of nerfactor.py:348-364 become the n_probes axis.
 * out rgb [n, n_probes, 3].                                                  */
/* Dynamic LDS the shading kernels need for a given sphere / probe count (must be <= 160 KiB). */
NFX_API size_t nfx_shade_lds_bytes(int n_lights, int n_probes);
/* nfx_shade_fwd_rows / nfx_shade_olat_fwd_rows (round 6): dev_lvis_row [n] int32 (or NULL) = the row of point i in
 * dev_lvis — the visibilities may live in the full [n_all, n_lights] buffer nfx_lvis_fwd_rows wrote.  The OLAT form also
 * takes dev_out_row [n] int32 (or NULL) = the row of point i in dev_rgb_olat, then the caller's full [n_all, n_lights, 3]
 * buffer (background rows: nfx_zero_rows with d = 3 n_lights), and dev_nan_flag (or NULL): 1 is OR-ed into the int32 when a
 * radiance is NaN before the clip to [0, 1] (tf.debugging.check_numerics of nerfactor.py:363 without its pass).        */
NFX_API int nfx_shade_fwd_rows(const float *dev_xyz, const float *dev_cam, const float *dev_normal,
                       const float *dev_albedo, const float *dev_rough, const float *dev_spec, float spec_scale,
                       float f0, const float *dev_lvis, const int32_t *dev_lvis_row, const float *dev_lxyz,
                       const float *dev_lareas, const float *dev_lights, int64_t n, int n_lights, int n_probes,
                       int linear2srgb, float *dev_rgb, void *stream);
NFX_API int nfx_shade_olat_fwd_rows(const float *dev_xyz, const float *dev_cam, const float *dev_normal,
                            const float *dev_albedo, const float *dev_rough, const float *dev_spec,
                            float spec_scale, float f0, const float *dev_lvis, const int32_t *dev_lvis_row,
                            const float *dev_lxyz, const float *dev_lareas, float olat_inten, float ambient,
                            int64_t n, int n_lights, int linear2srgb, const int32_t *dev_out_row,
                            float *dev_rgb_olat, int *dev_nan_flag, void *stream);
NFX_API int nfx_shade_fwd(const float *dev_xyz, const float *dev_cam, const float *dev_normal,
                  const float *dev_albedo, const float *dev_rough, const float *dev_spec,
                  float spec_scale, float f0, const float *dev_lvis, const float *dev_lxyz,
                  const float *dev_lareas, const float *dev_lights, int64_t n, int n_lights,
                  int n_probes, int linear2srgb, float *dev_rgb, void *stream);

/* One-light-at-a-time relighting (nerfactor.py:79-84,348-354): light_l = inten*onehot(l)+ambient.
 * out rgb_olat [n, n_lights, 3].                                             */
NFX_API int nfx_shade_olat_fwd(const float *dev_xyz, const float *dev_cam, const float *dev_normal,
                       const float *dev_albedo, const float *dev_rough, const float *dev_spec,
                       float spec_scale, float f0, const float *dev_lvis, const float *dev_lxyz,
                       const float *dev_lareas, float olat_inten, float ambient, int64_t n,
                       int n_lights, int linear2srgb, float *dev_rgb_olat, void *stream);

/* Learned-BRDF specular term, Model._eval_brdf_at (nerfactor.py:413-458):
 * gen_world2local (util/geom.py:119-149), dir2rusink (util/geom.py:152-192),
 * frozen BRDF MLP (models/brdf.py:57-66) on [z, pe2(rusink)], softplus;
 * 0 for back-lit (local l.z <= 0) directions.  out spec [n, n_lights].       */
NFX_API int nfx_brdf_spec_fwd(const float *dev_xyz, const float *dev_cam, const float *dev_normal,
                      const float *dev_z, int z_dim, const float *dev_lxyz, int n_lights,
                      const void *dev_blob, int prec, int64_t n, float *dev_spec, void *stream);

/* Rusinkiewicz coordinates (phi_d, theta_h, theta_d), util/geom.py:152-192. a,b [n,3]. */
NFX_API int nfx_dir2rusink(const float *dev_a, const float *dev_b, int64_t n, float *dev_rusink,
                   void *stream);

/* The learned BRDF's input rows made explicit, in fp32 (nerfactor/models/nerfactor.py:413-436 + embedder.py:23-47):
 * for every (point, light): rows[(i n_lights + l) ld_rows ...] = [z[i] | rusink | sin(2^k rusink), cos(2^k rusink) k < n_freqs]
 * with rusink = dir2rusink(R_i l, R_i v) in the local frame R_i = gen_world2local(normal[i]) and front[i n_lights + l] =
 * 1.0 where the light is in front of the surface (local l.z > 0), else 0.0 — what `precision = fp32` feeds to
 * nfx_mlp_generic_fwd for the frozen BRDF prior (spec = front * softplus(...)).  EVERY row is written (no compaction:
 * no data-dependent size, the step stays capturable).  _bwd: given dLoss/d rows (from nfx_mlp_generic_bwd's dx) writes
 * d_normal[n, 3] (through the local frame and the Rusinkiewicz angles, with the reference's custom gradients of
 * safe_acos / safe_atan2, util/math.py:24-60) and d_z[n, z_dim], each summed over the point's FRONT-LIT lights in a fixed
 * order (bit-reproducible, no atomics).  z_dim <= 8, n_freqs <= 8 (NFX_ENOSUP beyond).                                  */
NFX_API int nfx_brdf_rows_geom_fwd(const float *dev_xyz, const float *dev_cam, const float *dev_normal, const float *dev_z,
                                   int z_dim, const float *dev_lxyz, int n_lights, int64_t n, int n_freqs, float *dev_rows,
                                   int ld_rows, float *dev_front, void *stream);
NFX_API int nfx_brdf_rows_geom_bwd(const float *dev_xyz, const float *dev_cam, const float *dev_normal, int z_dim,
                                   const float *dev_lxyz, int n_lights, int64_t n, int n_freqs, const float *dev_d_rows,
                                   int ld_rows, float *dev_d_normal, float *dev_d_z, void *stream);

/* ------------------------------------------------------------------------ */
/* Training (replaces tape.gradient / optimizer.apply_gradients,             */
/* nerfactor/trainvali.py:278-285, for the pieces built so far).             */
/* ------------------------------------------------------------------------ */

/* Backward of a width-128 surface MLP call (NFX_IN_XYZ or NFX_IN_XYZ_LDIR): given
 * dout [rows, out_dim] = dLoss / d(post_scale * act(out) + post_bias), ACCUMULATES the gradients of
 * the 5 Keras kernels / biases into dev_dkernels[i] ([in, out] fp32) / dev_dbiases[i] ([out] fp32).
 * rows = n (NFX_IN_XYZ) or n * n_lights (NFX_IN_XYZ_LDIR, row = point * n_lights + light).
 * The forward is re-computed inside; `blob` is the TRAIN blob (forward + dgrad fragments).
 * Workspace: nfx_mlp128_bwd_workspace_bytes() bytes, 16-byte aligned.  No input gradients.    */
NFX_API size_t nfx_mlp128_train_packed_bytes(int in_kind);
NFX_API int nfx_mlp128_pack_train_weights(const float *const kernels[5], const float *const biases[5],
                                  int in_kind, int out_dim, int prec, void *blob, size_t blob_bytes);
NFX_API size_t nfx_mlp128_bwd_workspace_bytes(int in_kind, int64_t n, int n_lights);
NFX_API int nfx_mlp128_bwd(int in_kind, const float *dev_xyz, const float *dev_xyz_dir, int64_t n,
                   float xyz_scale, const float *dev_lxyz, int n_lights, const void *dev_blob,
                   int out_dim, int out_act, float post_scale, const float *dev_dout,
                   void *dev_workspace, size_t workspace_bytes, float *const dev_dkernels[5],
                   float *const dev_dbiases[5], int prec, void *stream);
/* The same for up to NFX_MLP128_MAX_HEADS networks over the SAME rows in ONE launch pair and reduction (round 5): the three xyz
 * heads of a NeRFactor step (nerfactor.py:377-411: normal, albedo, BRDF code; 2048 rows each at 1024 rays) are 16 workgroups per
 * launch on their own.  Head i: dev_blobs[i], out_dims[i], out_acts[i], post_scales[i], dev_douts[i]; gradients accumulated into
 * dev_dkernels[5 i .. 5 i + 5) / dev_dbiases[5 i ..) — no buffer may belong to two heads.  Workspace: n_heads x
 * nfx_mlp128_bwd_workspace_bytes().  Bit-identical to n_heads calls of nfx_mlp128_bwd.                                   */
#define NFX_MLP128_MAX_HEADS 4
NFX_API int nfx_mlp128_bwd_heads(int in_kind, const float *dev_xyz, const float *dev_xyz_dir, int64_t n, float xyz_scale,
                                 const float *dev_lxyz, int n_lights, int n_heads, const void *const *dev_blobs,
                                 const int *out_dims, const int *out_acts, const float *post_scales,
                                 const float *const *dev_douts, void *dev_workspace, size_t workspace_bytes,
                                 float *const *dev_dkernels, float *const *dev_dbiases, int prec, void *stream);

/* Backward of nfx_composite_fwd w.r.t. the raw network outputs (nerf.py:184-254): given dev_d_rgb [n_rays, 3] =
 * dLoss/d rgb (the composited, background-blended colour), writes dev_d_rgbs [n_rays, S, 4] = dLoss/d rgbs
 * (sigmoid and relu derivatives applied).  Same rgbs / z / rayd / noise / white_bg as the forward call.  Sample
 * positions carry no gradient (nerf.py:145).  16-byte aligned rgbs / d_rgbs.                                  */
NFX_API int nfx_composite_bwd(const float *dev_rgbs, const float *dev_z, const float *dev_rayd,
                      const float *dev_noise, int64_t n_rays, int n_samples, int white_bg,
                      const float *dev_d_rgb, float *dev_d_rgbs, void *stream);

/* Backward of nfx_nerf_mlp_fwd (trainvali.py:284 through nerf.py:256-290): given dev_d_rgbs [n_rays, S, 4],
 * ACCUMULATES the gradients of the 12 Keras kernels / biases (order of nfx_nerf_pack_weights) into
 * dev_dkernels[i] ([in, out] fp32) / dev_dbiases[i].  The forward is re-computed inside; `blob` is the TRAIN
 * blob (forward + dgrad fragments).  Workspace: nfx_nerf_bwd_workspace_bytes() bytes (feature-major bf16
 * activations and pre-activation gradients, ~10 KB per sample point), 16-byte aligned.  No input gradients.
 * Round 6: a point whose dev_d_rgbs row is four zeros (a sample the composite gave no weight: nerf.py:236-239) adds nothing
 * to any gradient and is skipped — the library lists the other points on the device, in ascending order, and works on
 * those (calls of >= 16384 points; option nerf_bwd_rows = 0: every point).  The sums are the same, their fp32 order
 * differs from the every-point form's; run-to-run the bits are identical.                                        */
NFX_API size_t nfx_nerf_train_packed_bytes(int prec);
NFX_API int nfx_nerf_pack_train_weights(const float *const kernels[12], const float *const biases[12],
                                int prec, void *blob, size_t blob_bytes);
NFX_API size_t nfx_nerf_bwd_workspace_bytes(int64_t n_rays, int n_samples);
NFX_API int nfx_nerf_mlp_bwd(const float *dev_rayo, const float *dev_rayd, const float *dev_z, int64_t n_rays,
                     int n_samples, const void *dev_blob, int prec, const float *dev_d_rgbs,
                     void *dev_workspace, size_t workspace_bytes, float *const dev_dkernels[12],
                     float *const dev_dbiases[12], void *stream);

/* Backward of nfx_brdf_spec_fwd (frozen prior, so no weight gradients): given dev_dspec [n, L] =
 * dLoss/d spec, ADDS dLoss/d z to dev_d_z [n, z_dim] and dLoss/d normal to dev_d_normal [n, 3] (zero or pre-fill
 * them).  The sums over a point's lights are taken in 64-bit fixed point inside `dev_workspace`
 * (nfx_brdf_spec_bwd_workspace_bytes(z_dim, n), 8-byte aligned): independent of the order the waves arrive in, hence
 * bit-reproducible.  `blob` is the BRDF train blob (nfx_brdf_train_packed_bytes / nfx_brdf_pack_train_weights).  */
NFX_API size_t nfx_brdf_train_packed_bytes(void);
NFX_API int nfx_brdf_pack_train_weights(const float *const kernels[5], const float *const biases[5],
                                int z_dim, int prec, void *blob, size_t blob_bytes);
NFX_API size_t nfx_brdf_spec_bwd_workspace_bytes(int z_dim, int64_t n);
NFX_API int nfx_brdf_spec_bwd(const float *dev_xyz, const float *dev_cam, const float *dev_normal,
                      const float *dev_z, int z_dim, const float *dev_lxyz, int n_lights,
                      const void *dev_blob, int prec, int64_t n, const float *dev_dspec,
                      float *dev_d_z, float *dev_d_normal, void *dev_workspace, size_t workspace_bytes,
                      void *stream);
/* nfx_brdf_spec_bwd_rows (round 6) = nfx_brdf_spec_bwd over the rows with dev_dspec != 0 ONLY: a row whose upstream gradient
 * is zero contributes exactly nothing, and the shading backward (nfx_shade_bwd) zeroes d spec of every back-facing light — the
 * half of the (point, light) rows the forward never evaluates either (nerfactor.py:429-434).  dev_list_workspace:
 * nfx_brdf_spec_bwd_list_bytes(n, n_lights) bytes, 16-byte aligned (the row list and its length stay on the device; 0 = more
 * than 2^31 rows: use the dense call), or NULL = the dense call.  Both forms sum in fixed point from the first addition on and
 * return the same bits.                                                                                                     */
NFX_API size_t nfx_brdf_spec_bwd_list_bytes(int64_t n, int n_lights);
NFX_API int nfx_brdf_spec_bwd_rows(const float *dev_xyz, const float *dev_cam, const float *dev_normal,
                           const float *dev_z, int z_dim, const float *dev_lxyz, int n_lights,
                           const void *dev_blob, int prec, int64_t n, const float *dev_dspec,
                           float *dev_d_z, float *dev_d_normal, void *dev_workspace, size_t workspace_bytes,
                           void *dev_list_workspace, size_t list_workspace_bytes, void *stream);

/* The BRDF prior on EXPLICIT rows — evaluation and one training step's backward of models/brdf.py
 * (reference nerfactor/models/brdf.py:57-66 `_eval_brdf_at`, :87-136 `call`/`compute_loss`, trained by
 * trainvali.py:273-295).  Row r < n evaluates (z[r, :z_dim], rusink[r, :3] = (phi_d, theta_h, theta_d)); with
 * reci != 0 there are 2 n rows and row n + r repeats the inputs with phi_d + pi (brdf.py:103-106).  `blob` is the
 * BRDF train blob (nfx_brdf_pack_train_weights).
 *   nfx_brdf_rows_fwd: dev_out[rows] = softplus(out(mlp([z, posenc2(rusink)]))).
 *   nfx_brdf_rows_bwd: given dev_dout[rows] = dLoss/d out, WRITES dev_d_z[rows, z_dim] = dLoss/d z per row and
 *     ACCUMULATES the weight gradients into dev_dkernels / dev_dbiases (Keras layout, as nfx_mlp128_bwd).    */
NFX_API int nfx_brdf_rows_fwd(const float *dev_z, int z_dim, const float *dev_rusink, int64_t n, int reci,
                      const void *dev_blob, int prec, float *dev_out, void *stream);
NFX_API size_t nfx_brdf_rows_bwd_workspace_bytes(int z_dim, int64_t n, int reci);
NFX_API int nfx_brdf_rows_bwd(const float *dev_z, int z_dim, const float *dev_rusink, int64_t n, int reci,
                      const void *dev_blob, int prec, const float *dev_dout, void *dev_workspace,
                      size_t workspace_bytes, float *dev_d_z, float *const dev_dkernels[5],
                      float *const dev_dbiases[5], void *stream);

/* Backward of nfx_shade_fwd for ONE light (n_probes = 1, the trained light): given dev_drgb [n,3] =
 * dLoss/d rgb, writes d_albedo [n,3], d_normal [n,3], d_lvis [n,L] and either d_rough [n]
 * (microfacet, dev_spec == NULL) or d_spec [n,L] (given specular term); ADDS the light's gradient to d_light [L,3]
 * when that is non-NULL (zero it before the first call of a step) — summed over the points in 64-bit fixed point
 * inside `dev_workspace` (nfx_shade_bwd_workspace_bytes(n_lights), 8-byte aligned), so the result is independent of
 * the order of the atomics: bit-reproducible.  Gradients flow to the normal both through cos = l.n and through the
 * BRDF; none to positions, camera or light geometry.  */
NFX_API size_t nfx_shade_bwd_workspace_bytes(int n_lights);
NFX_API int nfx_shade_bwd(const float *dev_xyz, const float *dev_cam, const float *dev_normal,
                  const float *dev_albedo, const float *dev_rough, const float *dev_spec,
                  float spec_scale, float f0, const float *dev_lvis, const float *dev_lxyz,
                  const float *dev_lareas, const float *dev_light, int64_t n, int n_lights,
                  int linear2srgb, const float *dev_drgb, float *dev_d_albedo, float *dev_d_rough,
                  float *dev_d_spec, float *dev_d_normal, float *dev_d_lvis, float *dev_d_light,
                  void *dev_workspace, size_t workspace_bytes, void *stream);

/* Device-side re-packing of any blob produced by the nfx_*_pack_*weights functions (all of them are pure gathers of
 * the parameters).  One map entry (two int32) per 32-bit word of the blob: (a, -2) = the fp32 value src[a];
 * (a, b) = the bf16 pair {src[a], src[b]}, low half first; a negative index gives 0.  `src` is the concatenation
 * of the network's fp32 parameters on the device, the map a device int32 array built once from the host packer
 * (nerfactor_amd/ops.py:DevicePacker).  Lets a training step refresh its blobs without a device->host round trip. */
NFX_API int nfx_pack_gather(const float *dev_src, const int32_t *dev_map, int64_t n_words, void *dev_blob, void *stream);

/* tf.keras.optimizers.Adam(amsgrad=True) dense update on flat fp32 buffers (trainvali.py:116-127):
 * lr_t = lr * sqrt(1 - beta2^step) / (1 - beta1^step); m, v, vhat updated in place;
 * p -= lr_t * m / (sqrt(vhat) + eps).  `step` is 1-based.                                      */
NFX_API int nfx_amsgrad_step(float *dev_p, const float *dev_g, float *dev_m, float *dev_v, float *dev_vhat,
                     int64_t n, float lr, float beta1, float beta2, float eps, int64_t step,
                     void *stream);
/* The same update with lr_t read from device memory (dev_lr_t[0] = nfx_amsgrad_step_size(lr, beta1, beta2, step),
 * written by the host before the launch): a training step captured in a hipGraph replays with the step size of the
 * current step, not the one of the step it was captured at (nerfactor_amd/optim.py:GraphedTrainStep).            */
NFX_API float nfx_amsgrad_step_size(float lr, float beta1, float beta2, int64_t step);
NFX_API int nfx_amsgrad_step_dev(float *dev_p, const float *dev_g, float *dev_m, float *dev_v, float *dev_vhat,
                         int64_t n, const float *dev_lr_t, float beta1, float beta2, float eps, void *stream);

/* tf.linalg.l2_normalize over the rows of x[n, d] (d <= 16; util/math.py:63-64 of the reference: x * rsqrt(max(sum x^2, eps)))
 * and its pull-back dx = (d y / d x)^T dy — the predicted normals and BRDF codes of a NeRFactor training step
 * (nerfactor.py:205-206, 266-270), one launch each way instead of 5 + 12 elementwise ones.                       */
NFX_API int nfx_l2_normalize_rows(const float *dev_x, float *dev_y, int64_t n, int d, float eps, void *stream);
NFX_API int nfx_l2_normalize_rows_bwd(const float *dev_x, const float *dev_dy, float *dev_dx, int64_t n, int d, float eps,
                                      void *stream);
/* The light probe's smoothness penalties (nerfactor.py:526-539): dev_loss[0] = tv_weight * sum((L - roll(L, 1, 1))^2 +
 * (L - roll(L, 1, 0))^2) + achro_weight * sum((L - roll(L, 1, 2))^2) over dev_light[h, w, 3], and dev_grad[h, w, 3] =
 * d loss / d L.  One block, fixed reduction order.                                                                */
NFX_API int nfx_light_smoothness(const float *dev_light, int h, int w, float tv_weight, float achro_weight, float *dev_loss,
                                 float *dev_grad, void *stream);

/* The per-ray training losses of the surface models (nerfactor.py:463-541, shape.py:239-277 `compute_loss`) as one
 * launch: loss[ray] = sum_t w_t * mean_d f_t(A_t[ray, d] - B_t[ray, d]), f = square (keras MSE) or abs (MAE), A / B
 * optionally alpha-blended onto the background first (util/img.py:alpha_blend: x * alpha + bg * (1 - alpha)).
 * nfx_pair_loss_bwd: given dev_dloss[n] writes d loss / d A into term.ga and d loss / d B into term.gb where those
 * are non-NULL (NFX_LOSS_ACCUM_*: add to what an earlier term of the SAME call wrote — a tensor used in two terms). */
#define NFX_LOSS_MAX_TERMS 8
#define NFX_LOSS_MSE 0
#define NFX_LOSS_MAE 1
#define NFX_LOSS_BLEND_A 1
#define NFX_LOSS_BLEND_B 2
#define NFX_LOSS_ACCUM_A 4
#define NFX_LOSS_ACCUM_B 8
typedef struct nfx_loss_term {
    const float *a, *b; /* [n, d] each */
    float *ga, *gb;     /* backward outputs [n, d] or NULL */
    int d;
    float w;
    int kind;  /* NFX_LOSS_MSE | NFX_LOSS_MAE */
    int flags; /* NFX_LOSS_BLEND_* | NFX_LOSS_ACCUM_* */
} nfx_loss_term;
NFX_API int nfx_pair_loss_fwd(const nfx_loss_term *terms, int n_terms, const float *dev_alpha, float bg, int64_t n,
                      float *dev_loss, void *stream);
NFX_API int nfx_pair_loss_bwd(const nfx_loss_term *terms, int n_terms, const float *dev_alpha, float bg, int64_t n,
                      const float *dev_dloss, void *stream);

/* ------------------------------------------------------------------------ */
/* Geometry extraction from a trained NeRF (geometry_from_nerf.py:177-350).   */
/* ------------------------------------------------------------------------ */

/* sigma[n_rays, S] = sigma_out(enc(posenc(rayo + rayd z))) BEFORE the relu (eval_sigma_mlp, :322-350), from the
 * GEOM blob (nfx_nerf_pack_geom_weights, whose first 65 chunks are the encoder + the sigma tile): the bottleneck /
 * rgb head is neither evaluated nor streamed.  Bit-identical to the sigma of nfx_nerf_mlp_fwd.                 */
NFX_API int nfx_nerf_sigma_fwd(const float *dev_rayo, const float *dev_rayd, const float *dev_z, int64_t n_rays,
                       int n_samples, const void *dev_geom_blob, int prec, float *dev_sigma, void *stream);

/* out[n_rays, S, 4] = (n_x, n_y, n_z, sigma_raw) with n = -l2_normalize(d relu(sigma_raw)/dx, eps 1e-12): the
 * per-sample normal of compute_depth_and_normal (:280-297, GradientTape.batch_jacobian there).  `geom_blob` =
 * nfx_nerf_pack_geom_weights (encoder + sigma tile, transposed encoder, input-gradient tiles).  16-byte aligned out. */
NFX_API size_t nfx_nerf_geom_packed_bytes(int prec);
NFX_API int nfx_nerf_pack_geom_weights(const float *const kernels[12], const float *const biases[12], int prec,
                               void *blob, size_t blob_bytes);
NFX_API int nfx_nerf_sigma_grad(const float *dev_rayo, const float *dev_rayd, const float *dev_z, int64_t n_rays,
                        int n_samples, const void *dev_geom_blob, int prec, float *dev_normal_sigma,
                        void *stream);

/* The same output with the reverse sweep run only where it is not zero (round 6): d relu(sigma_raw)/dx of a sample with
 * sigma_raw <= 0 — empty space, most samples of a fitted scene — is zero; its normal is written as (-0, -0, -0) (the
 * every-sample kernel writes zeros with the sign of g * 0 there), every other sample gets the same bits.  Three
 * steps on the stream, nothing visits the host: the forward-only density of every sample (nfx_nerf_sigma_fwd's kernel), the
 * ascending list of the samples with a density (every other sample's output row is written by that pass), the gradient
 * kernel over the list.  `workspace`: nfx_nerf_sigma_grad_workspace_bytes bytes, 16-byte aligned; n_rays * S < 2^31.   */
NFX_API size_t nfx_nerf_sigma_grad_workspace_bytes(int64_t n_rays, int n_samples);
NFX_API int nfx_nerf_sigma_grad_rows(const float *dev_rayo, const float *dev_rayd, const float *dev_z, int64_t n_rays,
                             int n_samples, const void *dev_geom_blob, int prec, float *dev_normal_sigma,
                             void *dev_workspace, size_t workspace_bytes, void *stream);

/* Selective fp32-class refinement of the COARSE densities of a bf16 render (round 6; nerf.py:138-147 with
 * util/math.py:71-94: the coarse weights place the fine samples, and on a fitted network's silhouette rays a bf16
 * density error of 0.1-0.3 moves them across the edge).  Two launches, no host round trip:
 *   nfx_nerf_refine_select  lists the samples that decide (same scan as nfx_composite_fwd): visible (T_i > t_min) and
 *                           either not saturated (a_lo < alpha_i < a_hi) or undecided (|sigma_i| < sigma_margin: the side
 *                           of the relu is within the bf16 error), grown by `dilate` neighbours either way, never the last
 *                           sample of a ray; dev_list[n_rays * S] int32 receives flat indices ray * S + s in arbitrary
 *                           order, dev_count[1] their number (zeroed by this call);
 *   nfx_nerf_sigma_refine   overwrites rgbs[i, 3] (the density channel of rgbs[n_rays, S, 4]) of every listed sample i
 *                           with the fp32-class density (NFX_PREC_FP32 blob of nfx_nerf_pack_geom_weights), bit-identical
 *                           to nfx_nerf_sigma_fwd(prec = NFX_PREC_FP32) at those samples; reads the count on the device. */
NFX_API int nfx_nerf_refine_select(const float *dev_rgbs, const float *dev_z, const float *dev_rayd, int64_t n_rays,
                           int n_samples, float t_min, float a_lo, float a_hi, float sigma_margin, int dilate,
                           int *dev_list, int *dev_count, void *stream);
NFX_API int nfx_nerf_sigma_refine(const float *dev_rayo, const float *dev_rayd, const float *dev_z, int64_t n_rays,
                          int n_samples, const void *dev_geom_blob_fp32, const int *dev_list, const int *dev_count,
                          float *dev_rgbs, void *stream);

/* ------------------------------------------------------------------------ */
/* Diagnostics.                                                              */
/* ------------------------------------------------------------------------ */

/* D[32,32] = A[32,16] * B[16,32] through one v_mfma_f32_32x32x16_bf16 with the operand
 * lane maps documented in DESIGN.md section 3 / csrc/mlp_engine.hpp; used by the GPU tests to pin the fragment layout. */
NFX_API int nfx_selftest_mfma_bf16(const float *dev_a, const float *dev_b, float *dev_d, void *stream);
/* out[i] = (which ? cos : sin)(in[i]) with the kernel's own range reduction.  */
/* ------------------------------------------------------------------------ */
/* Runtime-shaped MLP (csrc/mlp_generic.hip): every mlp.Network the reference */
/* can build (nerfactor/networks/mlp.py:24-50: widths, activations, skip_at    */
/* anywhere; nerfactor/models/nerf.py:53-90: mlp_width, enc_depth,             */
/* use_views = False, pos_enc = False) that the tuned kernels above do not     */
/* cover — one fused kernel, bf16 operands / fp32 accumulate.                  */
/* widths[i] = units of Dense layer i, acts[i] = NFX_ACT_*, skip_input[i] != 0  */
/* <=> layer i reads concat(output of layer i - 1, network input) (i - 1 is in  */
/* the reference's skip_at).  Limits: d_in <= 320, widths <= 256, <= 16 layers  */
/* (NFX_ENOSUP beyond).  kernels[i]: Keras layout [in_i, widths[i]].            */
/* x: [n, ld_x] fp32, first d_in columns; y: [n, ld_y], columns                 */
/* [col0, col0 + widths[n_layers - 1]) are written (so that a caller can        */
/* assemble concat(features, embedded view) without a copy).                    */
/* ------------------------------------------------------------------------ */
/* prec here: NFX_PREC_BF16 as above; NFX_PREC_FP32 = fp32-class as in the tuned kernels (every MFMA operand a    */
/* bf16 hi / lo pair, three v_mfma_f32_32x32x16_bf16 per product) with fp32 activations, gradients and workspace,     */
/* forward AND backward, for any shape; NFX_PREC_FP32_NATIVE = fp32 operands on the native fp32 matrix instruction    */
/* (v_mfma_f32_32x32x2_f32: the reference's own arithmetic bit class, ~5x the matrix time).  2-KiB fragments in both  */
/* fp32 modes; a blob is packed for one prec.  nfx_mlp_generic_split_hilo turns the fragments of a                    */
/* NFX_PREC_FP32_NATIVE blob (a pure gather of the parameters: nfx_pack_gather can rebuild it on the device) into     */
/* NFX_PREC_FP32 fragments IN PLACE on `stream` (train != 0: a train blob, forward + transposed fragments).           */
NFX_API int nfx_mlp_generic_split_hilo(void *dev_blob, int d_in, int n_layers, const int *widths, const int *skip_input,
                                       int train, void *stream);
NFX_API size_t nfx_mlp_generic_packed_bytes(int d_in, int n_layers, const int *widths, const int *skip_input, int prec);
NFX_API int nfx_mlp_generic_pack(const float *const *kernels, const float *const *biases, int d_in, int n_layers,
                                 const int *widths, const int *skip_input, int prec, void *blob, size_t blob_bytes);
NFX_API int nfx_mlp_generic_fwd(const float *dev_x, int64_t n, int ld_x, int d_in, int n_layers, const int *widths,
                                const int *acts, const int *skip_input, const void *dev_blob, int prec, float *dev_y,
                                int ld_y, int col0, void *stream);
/* Backward of the same networks (what trainvali.py:278-285 needs for a non-shipped shape: tf.GradientTape over
 * mlp.Network).  The train blob = the forward blob followed by the transposed (dgrad) fragments; nfx_mlp_generic_fwd
 * accepts it too.  Given dy = dLoss / d(activated output) [n, ld_dy] (columns col0_dy ...), ADDS dLoss/dW into
 * dkernels[i] ([in_i, widths[i]], Keras layout) and dLoss/db into dbiases[i], and — dx != NULL — writes
 * dLoss / d(network input) to dx[n, ld_dx] (first d_in columns), for chaining networks (bottleneck -> rgb_out,
 * nerfactor/models/nerf.py:277-287).  dkernels = dbiases = NULL: only dx is computed (input gradients of a frozen network).  The kernels re-compute the forward in bf16 from x; activation derivatives of
 * hidden layers are taken from their bf16 outputs.  Deterministic (no atomics; the row split of the weight-gradient
 * contraction depends on the problem shape only).  Workspace: nfx_mlp_generic_bwd_workspace_bytes (about
 * 2 (bf16) or 4 (fp32) bytes x n x (d_in + 2 x sum of widths)), 16-byte aligned device memory.  With NFX_PREC_FP32
 * nothing is rounded to bf16 anywhere: this is the backward at the reference's arithmetic (trainvali.py:273-285). */
NFX_API size_t nfx_mlp_generic_train_packed_bytes(int d_in, int n_layers, const int *widths, const int *skip_input,
                                                  int prec);
NFX_API int nfx_mlp_generic_pack_train(const float *const *kernels, const float *const *biases, int d_in, int n_layers,
                                       const int *widths, const int *skip_input, int prec, void *blob,
                                       size_t blob_bytes);
NFX_API size_t nfx_mlp_generic_bwd_workspace_bytes(int64_t n, int d_in, int n_layers, const int *widths,
                                                   const int *skip_input, int prec);
NFX_API int nfx_mlp_generic_bwd(const float *dev_x, int64_t n, int ld_x, int d_in, int n_layers, const int *widths,
                                const int *acts, const int *skip_input, const void *dev_train_blob, int prec,
                                const float *dev_dy, int ld_dy, int col0_dy, float *dev_dx, int ld_dx,
                                float *const *dev_dkernels, float *const *dev_dbiases, void *dev_workspace,
                                size_t workspace_bytes, void *stream);
/* Embedder (nerfactor/networks/embedder.py:23-47) as a kernel: out[row, col0 ...] = [v, sin(2^0 v), cos(2^0 v), ...]
 * (incl_input, n_freqs log-sampled bands; n_freqs = 0 = identity) of a 3-vector v per row:
 *   mode 0: v = x[row / per_ray]        mode 1: v = x[row / per_ray] + dir[row / per_ray] * z[row]  (points along rays,
 *   nerfactor/models/nerf.py:162-164)    mode 2: v = dir[row / per_ray]
 *   mode 3: v = safe_l2_normalize(dir[row % per_ray] - x[row / per_ray]): the unit direction from surface point
 *   row / per_ray to light row % per_ray (nerfactor/models/shape.py:128-131; per_ray = number of lights). */
/* The embedding's pull-back for explicit vectors (mode 0): dv[n, 3] = (d embedding / d v)^T d_out[row, col0 ...] — with
 * nfx_mlp_generic_bwd's dx it gives d(network output)/d(point), the density gradient geometry_from_nerf.py:288-297 takes
 * by tf.GradientTape, for a NeRF of any shape. */
NFX_API int nfx_embed_bwd(const float *dev_v, int64_t n, int n_freqs, int incl_input, const float *dev_d_out, int ld_out,
                          int col0, float *dev_dv, void *stream);
NFX_API int nfx_embed(const float *dev_x, const float *dev_dir, const float *dev_z, int64_t n, int per_ray, int mode,
                      int n_freqs, int incl_input, float *dev_out, int ld_out, int col0, void *stream);

/* D[32][32] = H^T Z (fp32 in, bf16 operands, fp32 out) for two row-major [16 rows][32 slots] tiles, contracted over the
 * ROW axis through LDS and ds_read_b64_tr_b16 exactly as the fused weight-gradient kernels do (csrc/tr16.hpp).
 * mode 1: D[256] = the raw lane map of the instruction (lane l reads LDS bytes 8 l .. 8 l + 7 of a tile holding its own
 * 16-bit element indices). */
NFX_API int nfx_selftest_tr16(const float *dev_h, const float *dev_z, float *dev_d, int mode, void *stream);
NFX_API int nfx_selftest_sincos(const float *dev_in, int64_t n, int which, float *dev_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NFX_H_ */
