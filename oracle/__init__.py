"""CPU oracle for the NeRFactor per-ray rendering hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``nerfactor_amd/`` may import this package:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
do, and only as the checker.  The product path fails loudly when libnfx.so is missing.

The reference (google/nerfactor) is 100 % Python on TensorFlow 2.2, ships no tests, golden
vectors or checkpoints for this path, and TensorFlow is not installable in this environment.
This package restates the reference's op sequence in NumPy (``*_ref.py``, dtype-generic:
float32 mirrors TF's compute type, float64 is the anchor) and in torch-CPU fp32
(``torch_ref.py``, the timed CPU baseline).  How it is pinned:

* **Against the reference's own model code, run here.**  ``tests/golden/make_reference_golden.py``
  imports the unmodified ``nerfactor/models/{nerf,shape,brdf,nerfactor,nerfactor_microfacet}.py`` and
  ``nerfactor/geometry_from_nerf.py`` (+ ``networks/``, ``util/``, ``brdf/microfacet``) from the reference tree, configures them from the
  reference's own ``config/*.ini``, and executes ``Model.call`` / ``compute_loss`` on a NumPy stand-in
  for the TensorFlow API (``tests/golden/tf_shim``); the outputs are committed as
  ``tests/golden/reference_models.npz`` and ``tests/test_cpu_reference_golden.py`` holds every oracle
  function to them at float32 tolerance.  This pins the ALGORITHM — operation order, concatenation
  orders, epsilons, clipping, masking, chunking, loss weighting — to the reference's Python.
* **Against the pieces of the reference that import without TensorFlow**
  (``tests/golden/make_golden.py`` -> ``reference_anchors.npz``): ``brdf.renderer.gen_light_xyz``,
  ``xiuminglib.geometry.sph.sph2cart``, ``nielsen2015on.coordinateFunctions.DirectionsToRusink``,
  ``xiuminglib.metric.PSNR``, ``xiuminglib.img.rgb2lum`` / ``linear2srgb``.
* **What stays UNPINNED**: the TensorFlow kernels themselves (the shim implements their documented
  semantics in NumPy, so a TF-internal rounding or an undocumented behaviour is not captured), the
  training gradients (the shim's ``tf.GradientTape`` is forward-mode and covers only geometry_from_nerf's
  d sigma / dx, which IS pinned; weight gradients are checked against torch autograd of the restatement), and the TF checkpoint
  reader (no TF-written file available).

Every function cites the reference ``file:line`` (relative to the reference tree) it follows.
"""
