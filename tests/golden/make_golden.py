"""Generates tests/golden/reference_anchors.npz by running the pieces of the REAL reference
(google/nerfactor) that import without TensorFlow.  Run in the build container only
(/root/reference does not exist on the GPU box):

    python tests/golden/make_golden.py

Anchors:
  brdf/renderer.py:184-219                     gen_light_xyz(16, 32), gen_light_xyz(4, 8)
  third_party/xiuminglib/.../geometry/sph.py:157-198   sph2cart
  third_party/nielsen2015on/coordinateFunctions.py:117-129  DirectionsToRusink
  third_party/xiuminglib/.../metric.py:103-149  PSNR('uint8')
  third_party/xiuminglib/.../img.py:597-611,635-660  rgb2lum, linear2srgb
"""
import os
import sys

import numpy as np

REF = os.environ.get('NERFACTOR_REFERENCE', '/root/reference')
sys.path.insert(0, REF)

from brdf.renderer import gen_light_xyz  # noqa: E402
from third_party.nielsen2015on.coordinateFunctions import DirectionsToRusink  # noqa: E402
from third_party.xiuminglib import xiuminglib as xm  # noqa: E402


def main():
    rng = np.random.default_rng(20240925)
    out = {}
    for h in (16, 4):
        xyz, areas = gen_light_xyz(h, 2 * h)
        out['lxyz_%d' % h] = xyz
        out['lareas_%d' % h] = areas
    sph = np.stack([rng.uniform(.5, 3, 64), rng.uniform(-np.pi / 2, np.pi / 2, 64),
                    rng.uniform(-np.pi, np.pi, 64)], -1)
    out['sph_in'] = sph
    out['sph_cart'] = xm.geometry.sph.sph2cart(sph)
    # Rusinkiewicz coordinates: a = light dir, b = view dir, both in the local frame, upper hemi
    a = rng.normal(size=(256, 3))
    a[:, 2] = np.abs(a[:, 2]) + 1e-3
    b = rng.normal(size=(256, 3))
    b[:, 2] = np.abs(b[:, 2]) + 1e-3
    out['rusink_a'] = a
    out['rusink_b'] = b
    out['rusink_out'] = DirectionsToRusink(a, b)
    # PSNR / luma / sRGB
    im1 = rng.uniform(0, 1, (24, 32, 3))
    im2 = np.clip(im1 + rng.normal(0, .02, im1.shape), 0, 1)
    u1 = (im1 * 255).astype(np.uint8)
    u2 = (im2 * 255).astype(np.uint8)
    out['psnr_im1'] = im1
    out['psnr_im2'] = im2
    out['psnr_value'] = np.float64(xm.metric.PSNR('uint8')(u1, u2))
    out['lum'] = xm.img.rgb2lum(im1)
    lin = np.concatenate([np.linspace(0, 0.01, 64), rng.uniform(0, 1, 192)])
    out['srgb_in'] = lin
    out['srgb_out'] = xm.img.linear2srgb(lin)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_anchors.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == '__main__':
    main()
