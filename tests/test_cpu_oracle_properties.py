"""Property tests of the oracle (hypothesis): invariants of the reference's formulas that hold for ANY input, as a guard
against a restatement that only matches the fixtures."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import nerf_ref, nerfactor_ref

SETTINGS = dict(max_examples=40, deadline=None)


def _rng(seed):
    return np.random.default_rng(seed)


@given(st.integers(0, 10 ** 6), st.integers(3, 48), st.integers(1, 64))      # >= 3 coarse samples: the pdf lives on the interior bins
@settings(**SETTINGS)
def test_fine_sampling_is_sorted_bounded_and_keeps_the_coarse_depths(seed, n_coarse, n_fine):
    rng = _rng(seed)
    z = nerf_ref.gen_z(2., 6., n_coarse, 5, u=rng.uniform(0, 1, (5, n_coarse)).astype(np.float32))
    w = (rng.uniform(0, 1, (5, n_coarse)) ** rng.integers(1, 9)).astype(np.float32)
    w[0] = 0                                              # a ray that hit nothing
    z_all = nerf_ref.gen_z_fine(z, w, n_fine)
    assert z_all.shape == (5, n_coarse + n_fine)
    assert np.all(np.diff(z_all, axis=1) >= 0)
    assert z_all.min() >= z.min() - 1e-5 and z_all.max() <= z.max() + 1e-5
    for r in range(5):                                    # the coarse depths are a sub-multiset of the merged ones
        merged = list(z_all[r])
        for v in z[r]:
            merged.remove(v)
        assert len(merged) == n_fine


@given(st.integers(0, 10 ** 6), st.integers(2, 64))
@settings(**SETTINGS)
def test_compositing_weights_form_a_sub_probability(seed, n):
    rng = _rng(seed)
    z = np.sort(rng.uniform(2, 6, (7, n)).astype(np.float32), 1)
    sigma = rng.normal(0, 5, (7, n)).astype(np.float32)
    rd = rng.normal(size=(7, 3)).astype(np.float32)
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    w = nerf_ref.accumulate_sigma(sigma, z, rd)
    assert np.all(w >= 0)
    assert np.all(w.sum(1) <= 1 + n * 2e-6)               # safe_cumprod adds 1e-6 per factor
    # a ray whose last sample has positive density terminates there (dist = 1e10): the weights then sum to ~1
    hit = sigma[:, -1] > 1e-3
    np.testing.assert_allclose(w.sum(1)[hit], 1., atol=n * 2e-6 + 1e-5)
    # no density, no weight
    assert np.all(nerf_ref.accumulate_sigma(-np.abs(sigma), z, rd) == 0)


@given(st.integers(0, 10 ** 6))
@settings(**SETTINGS)
def test_rusinkiewicz_angles_do_not_depend_on_the_azimuth_of_the_local_frame(seed):
    rng = _rng(seed)
    a = rng.normal(size=(16, 3))
    b = rng.normal(size=(16, 3))
    a[:, 2], b[:, 2] = np.abs(a[:, 2]) + .1, np.abs(b[:, 2]) + .1
    # keep away from the theta_d = 0 pole, where phi_d is undefined
    keep = np.linalg.norm(np.cross(a, b), axis=1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1)) > .2
    phi = rng.uniform(0, 2 * np.pi)
    rot = np.array([[np.cos(phi), -np.sin(phi), 0], [np.sin(phi), np.cos(phi), 0], [0, 0, 1]])
    r0 = nerfactor_ref.dir2rusink(a.astype(np.float32), b.astype(np.float32))
    r1 = nerfactor_ref.dir2rusink((a @ rot.T).astype(np.float32), (b @ rot.T).astype(np.float32))
    d = np.abs(r0 - r1)
    d[:, 0] = np.minimum(d[:, 0], np.pi - d[:, 0])
    assert np.all(d[keep] < 2e-3), d[keep].max()
    assert np.all((r0[:, 1:] >= 0) & (r0[:, 1:] <= np.pi / 2 + 1e-3)) and np.all((r0[:, 0] >= 0) & (r0[:, 0] < np.pi + 1e-6))
    # swapping light and view keeps theta_h and theta_d (reciprocity of the parametrisation)
    r2 = nerfactor_ref.dir2rusink(b.astype(np.float32), a.astype(np.float32))
    np.testing.assert_allclose(r2[:, 1:], r0[:, 1:], atol=2e-3)


@given(st.integers(0, 10 ** 6), st.floats(0.05, 1.0))
@settings(**SETTINGS)
def test_microfacet_brdf_is_finite_non_negative_and_diffuse_when_asked(seed, rough):
    rng = _rng(seed)
    l = rng.normal(size=(6, 9, 3)).astype(np.float32)
    v = rng.normal(size=(6, 3)).astype(np.float32)
    n = rng.normal(size=(6, 3)).astype(np.float32)
    alb = rng.uniform(0, 1, (6, 3)).astype(np.float32)
    r = np.full((6, 1), rough, np.float32)
    brdf = nerfactor_ref.microfacet(l, v, n, alb, r, f0=0.04)
    assert brdf.shape == (6, 9, 3) and np.all(np.isfinite(brdf)) and np.all(brdf >= 0)
    lam = nerfactor_ref.microfacet(l, v, n, alb, r, lambert_only=True)
    np.testing.assert_allclose(lam, np.broadcast_to((alb / np.float32(np.pi))[:, None, :], lam.shape), rtol=1e-6)
    assert np.all(brdf >= lam - 1e-7)                     # the specular lobe only adds


@given(st.integers(0, 10 ** 6))
@settings(**SETTINGS)
def test_render_integral_is_linear_in_the_light_below_the_clip(seed):
    rng = _rng(seed)
    n, nl = 5, 32
    brdf = rng.uniform(0, .3, (n, nl, 3)).astype(np.float32)
    lvis = rng.uniform(0, 1, (n, nl)).astype(np.float32)
    s2l = rng.normal(size=(n, nl, 3)).astype(np.float32)
    s2l /= np.linalg.norm(s2l, axis=2, keepdims=True)
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    areas = rng.uniform(.01, .02, (4, 8)).astype(np.float32)
    la, lb = (rng.uniform(0, 1, (4, 8, 3)).astype(np.float32) for _ in range(2))
    f = lambda light: nerfactor_ref.integrate(brdf, lvis, s2l, nrm, light, areas, to_srgb=False)
    np.testing.assert_allclose(f(la + lb), f(la) + f(lb), atol=2e-6)       # everything stays far below the clip at 1
    back = np.einsum('ijk,ik->ij', s2l, nrm) <= 0
    assert np.all(f(la) >= 0) and back.any()
    lit_only_from_behind = np.where(back, lvis, 0)                          # back-lit lights contribute nothing
    assert np.all(nerfactor_ref.integrate(brdf, lit_only_from_behind, s2l, nrm, la, areas, to_srgb=False) == 0)
