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
