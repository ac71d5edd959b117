"""The NumPy TensorFlow stand-in behind tests/golden/reference_models.npz, checked primitive by primitive against an
independent implementation (torch) of the semantics TensorFlow documents — the fixtures are only as good as the shim.
Covers the non-trivial ones: searchsorted(side='right'), gather(batch_dims), scatter_nd (accumulating), exclusive
cumprod, l2_normalize(epsilon under the root), divide_no_nan, floormod, linspace, where/boolean_mask, Dense, and the
forward-mode GradientTape used for geometry_from_nerf's normals."""
import os
import sys

import numpy as np
import pytest
import torch

SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tf_shim')


@pytest.fixture(scope='module')
def tf():
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == 'tensorflow' or k.startswith('tensorflow.')}
    sys.path.insert(0, SHIM)
    try:
        import tensorflow as shim
        assert 'numpy-shim' in shim.__version__
        yield shim
    finally:
        sys.path.remove(SHIM)
        for k in list(sys.modules):
            if k == 'tensorflow' or k.startswith('tensorflow.'):
                del sys.modules[k]
        sys.modules.update(saved)


def test_dtype_inference_and_shapes(tf):
    assert tf.convert_to_tensor(1.5).dtype == np.float32 and tf.convert_to_tensor([1, 2]).dtype == np.int32
    assert tf.broadcast_to([1e10], (3, 1)).dtype == np.float32
    x = tf.zeros((4, 3))
    assert (x * 2.5 + 1).dtype == np.float32 and (2. ** tf.linspace(0., 3., 4)).dtype == np.float32
    assert x.shape[:1] + (7,) == (4, 7) and tuple(tf.shape(x)) == (4, 3)
    np.testing.assert_array_equal(tf.linspace(0., 9., 10), np.arange(10, dtype=np.float32))
    np.testing.assert_allclose(tf.linspace(0., 1., 64), torch.linspace(0, 1, 64).numpy(), atol=1e-7)


def test_searchsorted_gather_scatter(tf):
    rng = np.random.default_rng(0)
    cdf = np.sort(rng.uniform(0, 1, (5, 9)).astype(np.float32), -1)
    u = rng.uniform(0, 1, (5, 13)).astype(np.float32)
    u[0, :3] = cdf[0, [0, 4, 8]]                           # ties: side='right' counts entries <= u
    got = tf.searchsorted(cdf, u, side='right')
    np.testing.assert_array_equal(got, torch.searchsorted(torch.from_numpy(cdf), torch.from_numpy(u), right=True).numpy())
    ind = np.stack((np.maximum(0, got - 1), np.minimum(got, 8)), -1)            # [5, 13, 2]
    g = tf.gather(cdf, ind, axis=-1, batch_dims=len(ind.shape) - 2)
    want = torch.gather(torch.from_numpy(cdf), 1, torch.from_numpy(ind.reshape(5, -1)).long()).reshape(5, 13, 2).numpy()
    np.testing.assert_array_equal(g, want)
    # scatter_nd accumulates duplicates; where(cond) lists indices row-major; boolean_mask keeps row order
    idx = np.array([[0], [2], [2]])
    np.testing.assert_array_equal(tf.scatter_nd(idx, np.float32([[1, 1], [2, 2], [3, 3]]), (4, 2)),
                                  np.float32([[1, 1], [0, 0], [5, 5], [0, 0]]))
    mask = np.array([True, False, True, True])
    np.testing.assert_array_equal(tf.where(mask), [[0], [2], [3]])
    v = rng.normal(size=(4, 3)).astype(np.float32)
    np.testing.assert_array_equal(tf.scatter_nd(tf.where(mask), tf.boolean_mask(v, mask), (4, 3)), v * mask[:, None])
    np.testing.assert_array_equal(tf.gather_nd(v, np.array([[3], [0]])), v[[3, 0]])
    np.testing.assert_array_equal(tf.tensor_scatter_nd_update(v, np.array([[1]]), np.zeros((1, 3), np.float32))[1], 0)


def test_numeric_primitives(tf):
    rng = np.random.default_rng(1)
    x = rng.uniform(0.1, 1, (6, 7)).astype(np.float32)
    t = torch.from_numpy(x)
    excl = torch.cat((torch.ones(6, 1), torch.cumprod(t, 1)[:, :-1]), 1).numpy()
    np.testing.assert_allclose(tf.math.cumprod(x, axis=-1, exclusive=True), excl, rtol=1e-6)
    np.testing.assert_allclose(tf.cumsum(x, -1), torch.cumsum(t, 1).numpy(), rtol=1e-6)
    v = rng.normal(size=(5, 3)).astype(np.float32)
    v[0] = 0
    want = v / np.sqrt(np.maximum((v ** 2).sum(1, keepdims=True), 1e-6))      # x * rsqrt(max(sum x^2, eps))
    np.testing.assert_allclose(tf.linalg.l2_normalize(v, axis=1, epsilon=1e-6), want, rtol=1e-6)
    np.testing.assert_allclose(tf.linalg.norm(v, axis=1), np.linalg.norm(v, axis=1), rtol=1e-6)
    a, b = np.float32([1, 2, 3]), np.float32([2, 0, -4])
    np.testing.assert_array_equal(tf.math.divide_no_nan(a, b), np.float32([0.5, 0, -0.75]))
    np.testing.assert_allclose(tf.math.floormod(np.float32([-0.5, 3.5, 7.]), np.pi),
                               torch.remainder(torch.tensor([-0.5, 3.5, 7.]), np.pi).numpy(), rtol=1e-6)
    np.testing.assert_allclose(tf.math.sigmoid(v), torch.sigmoid(torch.from_numpy(v)).numpy(), rtol=1e-6)
    np.testing.assert_allclose(tf.math.softplus(v), torch.nn.functional.softplus(torch.from_numpy(v)).numpy(), rtol=1e-6)
    np.testing.assert_allclose(tf.linalg.cross(v, v[::-1]), torch.linalg.cross(torch.from_numpy(v),
                                                                               torch.from_numpy(v[::-1].copy())).numpy(),
                               rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(tf.einsum('ijk,ik->ij', rng.normal(size=(2, 3, 4)).astype(np.float32),
                                         np.ones((2, 4), np.float32)).shape, (2, 3))
    np.testing.assert_array_equal(tf.clip_by_value(np.float32([-1, .5, 2]), 0., 1.), [0, .5, 1])
    np.testing.assert_array_equal(tf.sort(np.float32([[3, 1, 2]]), -1), [[1, 2, 3]])
    np.testing.assert_array_equal(tf.roll(np.arange(4), 1, 0), [3, 0, 1, 2])
    np.testing.assert_allclose(tf.keras.losses.MSE(a, b), np.mean((a - b) ** 2), rtol=1e-6)
    np.testing.assert_allclose(tf.keras.losses.MAE(a, b), np.mean(np.abs(a - b)), rtol=1e-6)


def test_dense_layer_and_custom_gradient_forward(tf):
    rng = np.random.default_rng(2)
    k, b = rng.normal(size=(5, 4)).astype(np.float32), rng.normal(size=4).astype(np.float32)
    layer = tf.keras.layers.Dense(4, activation=tf.keras.layers.Activation('relu'))
    layer.set_weights([k, b])
    x = rng.normal(size=(3, 5)).astype(np.float32)
    np.testing.assert_allclose(layer(tf.convert_to_tensor(x)), np.maximum(x @ k + b, 0), rtol=1e-5, atol=1e-6)
    fresh = tf.keras.layers.Dense(8)
    fresh(tf.convert_to_tensor(x))                                   # builds: glorot-uniform kernel, zero bias
    assert fresh.kernel.shape == (5, 8) and np.abs(fresh.kernel).max() <= np.sqrt(6 / 13) and not fresh.bias.any()

    @tf.custom_gradient
    def f(x, eps=1e-6):
        return tf.acos(tf.clip_by_value(x, -1., 1.)), lambda dy: dy
    np.testing.assert_allclose(f(np.float32([2., 0.])), [0, np.pi / 2], rtol=1e-6)


def test_forward_mode_gradient_tape_matches_autograd(tf):
    """d relu(Dense(relu(Dense([x, sin(2^k x), cos(2^k x)])))) / dx, as geometry_from_nerf.py:289-295 asks for it."""
    rng = np.random.default_rng(3)
    k1, b1 = rng.normal(size=(15, 16)).astype(np.float32), rng.normal(size=16).astype(np.float32)
    k2, b2 = rng.normal(size=(16 + 15, 1)).astype(np.float32), rng.normal(size=1).astype(np.float32)
    d1, d2 = tf.keras.layers.Dense(16, activation='relu'), tf.keras.layers.Dense(1)
    d1.set_weights([k1, b1])
    d2.set_weights([k2, b2])
    pts = rng.normal(size=(9, 3)).astype(np.float32)

    def embed(x, mod):
        return mod.concat([x] + [f(x * 2. ** i) for i in (0, 1) for f in (mod.sin, mod.cos)], -1)

    x = tf.convert_to_tensor(pts)
    with tf.GradientTape() as g:
        g.watch(x)
        e = embed(x, tf)
        y = tf.nn.relu(d2(tf.concat((d1(e + 0), e), -1)))
    jac = np.asarray(g.batch_jacobian(y, x)).reshape(9, 3)

    class T:
        sin, cos = staticmethod(torch.sin), staticmethod(torch.cos)
        concat = staticmethod(torch.cat)
    xt = torch.tensor(pts, dtype=torch.float64, requires_grad=True)
    e = embed(xt, T)
    h = torch.relu(e @ torch.from_numpy(k1).double() + torch.from_numpy(b1).double())
    yt = torch.relu(torch.cat((h, e), -1) @ torch.from_numpy(k2).double() + torch.from_numpy(b2).double())
    (want,) = torch.autograd.grad(yt.sum(), xt)
    np.testing.assert_allclose(np.asarray(y), yt.detach().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(jac, want.numpy(), rtol=1e-4, atol=1e-4)
    with pytest.raises(RuntimeError):
        with tf.GradientTape() as g:
            g.watch(x)
            z = tf.exp(x)                      # not a tangent-carrying operation: must fail loudly, not return zeros
        g.batch_jacobian(z, x)


# ----------------------------------------------------------------------------------------------------------------------
# Every remaining tf.* symbol the reference's executed model / dataset / geometry code calls (grep of
# nerfactor/{models,networks,util,datasets}/*.py, geometry_from_nerf.py, brdf/), on BOTH stand-ins: the NumPy one behind
# reference_models.npz and the torch-backed one behind reference_grads.npz.  Expected values come from NumPy / torch
# written here from the TensorFlow documentation, not from the shims.
SHIM_TORCH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tf_shim_torch')


@pytest.fixture(scope='module', params=['numpy', 'torch'])
def anytf(request):
    path = SHIM if request.param == 'numpy' else SHIM_TORCH
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == 'tensorflow' or k.startswith('tensorflow.')}
    sys.path.insert(0, path)
    try:
        import tensorflow as shim
        assert ('%s-shim' % request.param) in shim.__version__
        yield shim
    finally:
        sys.path.remove(path)
        for k in list(sys.modules):
            if k == 'tensorflow' or k.startswith('tensorflow.'):
                del sys.modules[k]
        sys.modules.update(saved)


def _np(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def test_elementwise_reduction_and_shape_symbols(anytf):
    tf = anytf
    rng = np.random.default_rng(5)
    a = rng.uniform(0.2, 2., (4, 5)).astype(np.float32)
    b = rng.uniform(0.2, 2., (4, 5)).astype(np.float32)
    s = rng.normal(size=(4, 5)).astype(np.float32)
    A, B, S = (tf.convert_to_tensor(x) for x in (a, b, s))
    cases = {
        'abs': (lambda: tf.abs(S), np.abs(s)),
        'cos': (lambda: tf.cos(S), np.cos(s)), 'sin': (lambda: tf.sin(S), np.sin(s)),
        'math.cos': (lambda: tf.math.cos(S), np.cos(s)), 'math.sin': (lambda: tf.math.sin(S), np.sin(s)),
        'sqrt': (lambda: tf.sqrt(A), np.sqrt(a)),
        'square': (lambda: tf.square(S), s * s), 'math.log': (lambda: tf.math.log(A), np.log(a)),
        'math.pow': (lambda: tf.math.pow(A, 2.4), a ** np.float32(2.4)), 'exp': (lambda: tf.exp(S), np.exp(s)),
        'maximum': (lambda: tf.maximum(A, B), np.maximum(a, b)), 'minimum': (lambda: tf.minimum(A, B), np.minimum(a, b)),
        'math.minimum': (lambda: tf.math.minimum(A, 1.), np.minimum(a, 1)),
        'multiply': (lambda: tf.multiply(A, B), a * b),
        'matmul': (lambda: tf.matmul(A, tf.transpose(B)), a @ b.T),
        'transpose perm': (lambda: tf.transpose(tf.reshape(A, (2, 2, 5)), (2, 0, 1)), a.reshape(2, 2, 5).transpose(2, 0, 1)),
        'equal': (lambda: tf.equal(tf.convert_to_tensor([1, 2, 3]), 2), np.array([False, True, False])),
        'logical_and': (lambda: tf.logical_and(A > 1, B > 1), (a > 1) & (b > 1)),
        'logical_or': (lambda: tf.logical_or(A > 1, B > 1), (a > 1) | (b > 1)),
        'cast': (lambda: tf.cast(A > 1, tf.float32), (a > 1).astype(np.float32)),
        'expand_dims': (lambda: tf.expand_dims(A, 1), a[:, None, :]), 'newaxis': (lambda: A[:, tf.newaxis, :], a[:, None, :]),
        'identity': (lambda: tf.identity(A), a), 'ones': (lambda: tf.ones((2, 3)), np.ones((2, 3), np.float32)),
        'ones_like': (lambda: tf.ones_like(A), np.ones_like(a)), 'zeros_like': (lambda: tf.zeros_like(A), np.zeros_like(a)),
        'range': (lambda: tf.range(5), np.arange(5)), 'reshape -1': (lambda: tf.reshape(A, (-1, 2)), a.reshape(-1, 2)),
        'stack': (lambda: tf.stack((A, B), axis=-1), np.stack((a, b), -1)),
        'tile': (lambda: tf.tile(A[:1], (3, 2)), np.tile(a[:1], (3, 2))),
        'concat': (lambda: tf.concat((A, B), 1), np.concatenate((a, b), 1)),
        'reduce_sum': (lambda: tf.reduce_sum(A, axis=1), a.sum(1)),
        'reduce_sum keepdims': (lambda: tf.reduce_sum(A, axis=1, keepdims=True), a.sum(1, keepdims=True)),
        'reduce_mean all': (lambda: tf.reduce_mean(A), a.mean()), 'reduce_mean axes': (lambda: tf.reduce_mean(A, axis=(0, 1)), a.mean()),
        'reduce_max': (lambda: tf.reduce_max(A, axis=0), a.max(0)), 'reduce_min': (lambda: tf.reduce_min(A), a.min()),
        'clip_by_value': (lambda: tf.clip_by_value(S, -0.5, 0.5), np.clip(s, -0.5, 0.5)),
        'nn.relu': (lambda: tf.nn.relu(S), np.maximum(s, 0)),
    }
    for name, (fn, want) in cases.items():
        obj = tf
        for part in name.split(' ')[0].split('.'):
            if part in ('perm', 'newaxis'):
                break
            assert hasattr(obj, part), "shim lacks tf.%s" % name
            obj = getattr(obj, part)
        got = _np(fn())
        assert got.shape == np.asarray(want).shape, (name, got.shape, np.asarray(want).shape)
        if np.asarray(want).dtype == bool:
            np.testing.assert_array_equal(got, want, err_msg=name)
        else:
            np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-7, err_msg=name)
    assert int(tf.rank(A)) == 2 and tf.is_tensor(A) and not tf.is_tensor(a.tolist())
    for kw in ({}, {'indexing': 'ij'}):          # default 'xy' (util/img.py:45), 'ij' (datasets/nerf.py:115)
        gi, gj = tf.meshgrid(tf.range(3), tf.range(4), **kw)
        wi, wj = np.meshgrid(np.arange(3), np.arange(4), **kw)
        np.testing.assert_array_equal(_np(gi), wi)
        np.testing.assert_array_equal(_np(gj), wj)
    assert _np(tf.constant(3.5)).dtype == np.float32 and _np(tf.constant([1, 2])).dtype == np.int32


def test_losses_random_and_debugging_symbols(anytf):
    tf = anytf
    rng = np.random.default_rng(6)
    a = rng.normal(size=(7, 3)).astype(np.float32)
    b = rng.normal(size=(7, 3)).astype(np.float32)
    A, B = tf.convert_to_tensor(a), tf.convert_to_tensor(b)
    np.testing.assert_allclose(_np(tf.keras.losses.MSE(A, B)), ((a - b) ** 2).mean(-1), rtol=1e-6)   # mean over the last axis
    np.testing.assert_allclose(_np(tf.keras.losses.MAE(A, B)), np.abs(a - b).mean(-1), rtol=1e-6)
    u = _np(tf.random.uniform((500, 2)))
    assert u.shape == (500, 2) and u.dtype == np.float32 and 0 <= u.min() and u.max() < 1 and 0.4 < u.mean() < 0.6
    g = _np(tf.random.normal((4000,), mean=1., stddev=2.))
    assert g.dtype == np.float32 and abs(g.mean() - 1) < 0.15 and abs(g.std() - 2) < 0.15
    ui = _np(tf.random.uniform((200,), minval=0, maxval=7, dtype=tf.int32))
    assert ui.dtype == np.int32 and ui.min() >= 0 and ui.max() <= 6
    tf.debugging.check_numerics(A, "fine")
    with pytest.raises(Exception, match="Albedo"):
        tf.debugging.check_numerics(tf.convert_to_tensor(np.float32([1., np.nan])), "Albedo")
    tf.debugging.assert_greater(tf.convert_to_tensor(2.), 1.)
    with pytest.raises(Exception):
        tf.debugging.assert_greater(tf.convert_to_tensor(0.5), 1.)
    # (tf.debugging.assert_equal is only in datasets/brdf_merl.py:131, which the fixtures do not run: not shimmed)
    tf.debugging.Assert(tf.equal(tf.convert_to_tensor(3), 3), ['shape'])
    with pytest.raises(Exception):
        tf.debugging.Assert(tf.equal(tf.convert_to_tensor(3), 1), ['shape'])
    v = tf.Variable(a)
    assert _np(v).shape == (7, 3) and tf.is_tensor(v)


def test_torch_shim_reverse_mode_rules():
    """What tf.GradientTape().gradient must do for the reference's training step, on the torch stand-in: stop_gradient,
    custom_gradient's backward function is honoured (util/math.py:24-60), tf.maximum sends a tie's gradient to the FIRST
    argument, divide_no_nan has zero gradient where the divisor is zero, gather_nd of a Variable scatter-adds
    (networks/layers.py:54), reduce_mean over axis=() is a no-op, compute_average_loss divides by the global batch."""
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == 'tensorflow' or k.startswith('tensorflow.')}
    sys.path.insert(0, SHIM_TORCH)
    try:
        import tensorflow as tf
        x = tf.Variable(np.float32([0.3, 1.0, -2.0]))
        with tf.GradientTape() as tape:
            y = tf.reduce_sum(tf.stop_gradient(x) * x)          # d/dx = stop_gradient(x)
        np.testing.assert_allclose(_np(tape.gradient(y, [x])[0]), [0.3, 1.0, -2.0], rtol=1e-6)

        @tf.custom_gradient
        def clipped_acos(t):
            def grad(dy):
                return dy * 7.                                   # deliberately NOT the analytic derivative
            return tf.acos(tf.clip_by_value(t, -1., 1.)), grad
        with tf.GradientTape() as tape:
            y = tf.reduce_sum(clipped_acos(x))
        np.testing.assert_allclose(_np(tape.gradient(y, [x])[0]), [7., 7., 7.])
        p, q = tf.Variable(np.float32([1., 2., 3.])), tf.Variable(np.float32([1., 5., 0.]))
        with tf.GradientTape() as tape:
            y = tf.reduce_sum(tf.maximum(p, q))
        gp, gq = tape.gradient(y, [p, q])
        np.testing.assert_array_equal(_np(gp), [1., 0., 1.])     # tie at element 0: everything to the first argument
        np.testing.assert_array_equal(_np(gq), [0., 1., 0.])
        with tf.GradientTape() as tape:
            y = tf.reduce_sum(tf.math.divide_no_nan(p, q))
        gp, gq = tape.gradient(y, [p, q])
        np.testing.assert_allclose(_np(gp), [1., 0.2, 0.], rtol=1e-6)
        np.testing.assert_allclose(_np(gq), [-1., -2. / 25., 0.], rtol=1e-6)
        table = tf.Variable(np.float32([[1., 2.], [3., 4.], [5., 6.]]))
        ind = tf.convert_to_tensor(np.int32([2, 0, 2]))
        with tf.GradientTape() as tape:
            rows = tf.gather_nd(table, ind[:, None])
            y = tf.reduce_sum(rows * tf.convert_to_tensor(np.float32([[1., 1.], [2., 2.], [3., 3.]])))
        np.testing.assert_array_equal(_np(tape.gradient(y, [table])[0]), [[2., 2.], [0., 0.], [4., 4.]])
        per = tf.convert_to_tensor(np.float32([1., 2., 3., 6.]))
        np.testing.assert_array_equal(_np(tf.reduce_mean(per, axis=())), [1., 2., 3., 6.])
        assert float(tf.nn.compute_average_loss(per, global_batch_size=8)) == 1.5
    finally:
        sys.path.remove(SHIM_TORCH)
        for k in list(sys.modules):
            if k == 'tensorflow' or k.startswith('tensorflow.'):
                del sys.modules[k]
        sys.modules.update(saved)
