"""CPU oracle for the NeRFactor per-ray rendering hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``nerfactor_amd/`` may import this package:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
do, and only as the checker.  The product path fails loudly when libnfx.so is missing.

The reference (google/nerfactor) is 100 % Python on TensorFlow 2.2.  TensorFlow is not
installable in this environment and the reference ships no tests, golden vectors or
checkpoints for this path, so parity with the TF arithmetic itself is **UNPINNED**
("parity unpinned"): this package restates the reference's op sequence in
NumPy (``*_ref.py``, dtype-generic: float32 mirrors TF's compute type, float64 is the
anchor) and in torch-CPU fp32 (``torch_ref.py``, the timed CPU baseline), and the two
restatements are cross-checked against each other and against the pieces of the
reference that DO import here (``tests/golden/make_golden.py``):
``brdf.renderer.gen_light_xyz``, ``xiuminglib.geometry.sph.sph2cart``,
``nielsen2015on.coordinateFunctions.DirectionsToRusink``, ``xiuminglib.metric.PSNR``,
``xiuminglib.img.rgb2lum`` / ``linear2srgb``.

Every function cites the reference ``file:line`` (relative to the reference tree) it follows.
"""
