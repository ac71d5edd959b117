"""Generates tests/golden/reference_grads.npz: the REAL google/nerfactor training step, differentiated.

    python tests/golden/make_reference_grad_golden.py        (build container only: needs /root/reference)

The reference's unmodified nerfactor/models/{nerf,nerfactor_microfacet,nerfactor,brdf}.py run on tests/golden/tf_shim_torch
(the TensorFlow calls they make, implemented on torch-CPU tensors), so `tf.GradientTape().gradient` is reverse-mode
autodiff through the reference's own Python — including its tf.custom_gradient backward functions
(nerfactor/util/math.py:24-60: safe_acos, safe_atan2), its tf.stop_gradient on the fine samples (models/nerf.py:143),
the frozen BRDF prior (models/nerfactor.py:60) and tf.nn.compute_average_loss — and the optimizer is the TF 2.2
Adam(amsgrad=True) update.  One training step of trainvali.py:273-285 is replayed 10 times on a fixed batch:

    with tf.GradientTape() as tape:
        pred, gt, loss_kwargs, _ = model(batch, mode='train'); loss_kwargs['keep_batch'] = True
        weighted_loss = tf.nn.compute_average_loss(model.compute_loss(pred, gt, **loss_kwargs), global_batch_size=n)
    grads = tape.gradient(weighted_loss, model.trainable_variables)
    optimizer.apply_gradients(zip(grads, model.trainable_variables))

Stored per model: the loss of every step, the jitter / perturbation noise the reference drew (so the parity tests
replay it), and for EVERY trainable tensor the gradient at step 1 and the parameter after 1 and 10 steps — whole
tensors up to 2048 elements (biases, output layers, the light); Frobenius norm + sum + a fixed strided sample of
1024 elements for the larger kernels (tests/golden/golden_inputs.py:summary) to keep the fixture under 1 MB.
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('NERFACTOR_REFERENCE', '/root/reference')
sys.path[:0] = [os.path.join(HERE, 'tf_shim_torch'), REF, os.path.join(REF, 'nerfactor'), REPO]

import tensorflow as tf  # noqa: E402  (the torch shim)
import torch  # noqa: E402

assert 'torch-shim' in tf.__version__
from nerfactor.util import io as ioutil  # noqa: E402
from nerfactor.models.nerf import Model as NerfModel  # noqa: E402
from nerfactor.models.nerfactor import Model as NerfactorModel  # noqa: E402
from nerfactor.models.nerfactor_microfacet import Model as MicrofacetModel  # noqa: E402
from nerfactor.models.brdf import Model as BrdfModel  # noqa: E402

from tests import common  # noqa: E402
from tests.golden import golden_inputs as gi  # noqa: E402

OUT = {}
N_STEPS = 10
SAMPLE = 1024


def put(key, value):
    a = np.asarray(value.detach().numpy() if isinstance(value, torch.Tensor) else value)
    assert a.dtype != np.float64, (key, 'float64 leaked out of the fp32 model code')
    assert np.all(np.isfinite(a)) if a.dtype.kind == 'f' else True, key
    OUT[key] = a


def put_tensor(key, t):
    """Whole tensor when small, (norm, sum, strided sample) otherwise — see tests/golden/golden_inputs.py:summary."""
    a = t.detach().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    if a.size <= 2048:
        put(key, a.astype(np.float32))
    else:
        put(key + ':summary', gi.summary(a, SAMPLE))


def set_layers(network, pairs):
    assert len(network.layers) == len(pairs)
    for layer, (k, b) in zip(network.layers, pairs):
        layer.set_weights([k, b])


def ref_config(name, **override):
    cfg = ioutil.read_config(os.path.join(REF, 'nerfactor', 'config', name))
    for k, v in override.items():
        cfg.set('DEFAULT', k, str(v))
    return cfg


def named_variables(model):
    """name -> variable for everything tape.gradient is asked about, named like the product's state_dict keys."""
    out = {}
    for net_name, net in model.net.items():
        for i, layer in enumerate(net.layers):
            if layer.trainable and any(v is layer.kernel for v in model.trainable_variables):
                out['net_%s_layer%d.kernel' % (net_name, i)] = layer.kernel
                out['net_%s_layer%d.bias' % (net_name, i)] = layer.bias
    light = getattr(model, '_light', None)
    if light is not None and any(v is light for v in model.trainable_variables):
        out['_light'] = light
    code = getattr(model, 'latent_code', None)
    if code is not None and any(v is code._z for v in model.trainable_variables):
        out['latent_code._z'] = code._z
    assert len(out) == len(model.trainable_variables), (len(out), len(model.trainable_variables))
    return out


class Record:
    """Wraps tf.random.normal / uniform so the noise the reference draws can be replayed by the parity tests."""
    def __init__(self, name):
        self.name, self.draws, self.orig = name, [], getattr(tf.random, name)

    def __call__(self, shape, **kw):
        x = self.orig(shape, **kw)
        self.draws.append(x.numpy().copy())
        return x


def make_optimizer(config):
    """trainvali.py:110-127, verbatim in structure."""
    lr = config.getfloat('DEFAULT', 'lr')
    lr_decay_steps = config.getint('DEFAULT', 'lr_decay_steps', fallback=-1)
    if lr_decay_steps > 0:
        lr_decay_rate = config.getfloat('DEFAULT', 'lr_decay_rate')
        lr = tf.keras.optimizers.schedules.ExponentialDecay(lr, decay_steps=lr_decay_steps, decay_rate=lr_decay_rate)
    kwargs = {'learning_rate': lr, 'amsgrad': True}
    clipnorm = config.getfloat('DEFAULT', 'clipnorm')
    clipvalue = config.getfloat('DEFAULT', 'clipvalue')
    if clipnorm > 0:
        assert clipvalue < 0
        kwargs['clipnorm'] = clipnorm
    if clipvalue > 0:
        assert clipnorm < 0
        kwargs['clipvalue'] = clipvalue
    return tf.keras.optimizers.Adam(**kwargs)


def train(tag, model, batch, n, config):
    model.register_trainable()
    names = named_variables(model)
    optimizer = make_optimizer(config)
    rec_n, rec_u = Record('normal'), Record('uniform')
    tf.random.normal, tf.random.uniform = rec_n, rec_u
    losses = []
    try:
        for step in range(N_STEPS):
            with tf.GradientTape() as tape:
                pred, gt, loss_kwargs, _ = model(batch, mode='train')
                loss_kwargs['keep_batch'] = True
                per_example_loss = model.compute_loss(pred, gt, **loss_kwargs)
                weighted_loss = tf.nn.compute_average_loss(per_example_loss, global_batch_size=n)
            variables = model.trainable_variables
            grads = tape.gradient(weighted_loss, variables)
            if step == 0:
                by_id = {id(v): g for v, g in zip(variables, grads)}
                for name, v in names.items():
                    g = by_id[id(v)]
                    assert g is not None, name
                    put_tensor('%s/grad/%s' % (tag, name), g)
                put('%s/per_example_loss' % tag, per_example_loss)
            optimizer.apply_gradients(zip(grads, variables))
            losses.append(float(weighted_loss.detach()))
            if step in (0, N_STEPS - 1):
                for name, v in names.items():
                    put_tensor('%s/param_after_%d/%s' % (tag, step + 1, name), v)
    finally:
        tf.random.normal, tf.random.uniform = rec_n.orig, rec_u.orig
    put('%s/loss' % tag, np.float32(losses))
    for kind, rec in (('normal', rec_n), ('uniform', rec_u)):
        per_step = len(rec.draws) // N_STEPS
        assert per_step * N_STEPS == len(rec.draws)
        put('%s/draws_per_step_%s' % (tag, kind), np.int32(per_step))
        for i, d in enumerate(rec.draws):
            put('%s/%s_%03d' % (tag, kind, i), d.astype(np.float32))
    print('%-6s losses %s' % (tag, ' '.join('%.6f' % l for l in losses)))


# ------------------------------------------------------------------------------------------------ NeRF
def run_nerf():
    cfg = ref_config('nerf.ini')        # perturb = True, noise_std = 1 in training (config/nerf.ini)
    model = NerfModel(cfg)
    nets = common.nerf_nets(seed=gi.NERF_SEED)
    for pref, net in zip(('coarse_', 'fine_'), nets):
        for part in ('enc', 'sigma_out', 'bottleneck', 'rgb_out'):
            set_layers(model.net[pref + part], net[part])
    rayo, rayd, gt = gi.nerf_rays()
    rayo, rayd, gt = rayo[:gi.GRAD_NERF_RAYS], rayd[:gi.GRAD_NERF_RAYS], gt[:gi.GRAD_NERF_RAYS]
    n = rayo.shape[0]
    batch = (np.array([b'x'] * n), np.tile(np.int32([[4, n // 4]]), (n, 1))) + tuple(
        tf.convert_to_tensor(a) for a in (rayo, rayd, gt))
    train('nerf', model, batch, n, cfg)


# ------------------------------------------------------------------------------------------------ NeRFactor
def shape_batch(n_lights):
    rayo, rgb, alpha, xyz, normal, lvis = gi.surface_batch(n_lights)
    n = rayo.shape[0]
    return (np.array([b'x'] * n), np.tile(np.int32([[4, n // 4]]), (n, 1))) + tuple(
        tf.convert_to_tensor(a) for a in (rayo, np.zeros_like(rayo), rgb, alpha, xyz, normal, lvis)), n


def workdir(tmp, brdf_root):
    paths = {}
    for name, ini, over in (('shape', 'shape.ini', {}), ('brdf', 'brdf.ini', {'data_root': brdf_root})):
        root = os.path.join(tmp, name)
        os.makedirs(os.path.join(root, 'lr1e-2', 'checkpoints'))
        ioutil.write_config(ref_config(ini, **over), os.path.join(root, 'lr1e-2.ini'))
        paths[name] = os.path.join(root, 'lr1e-2', 'checkpoints', 'ckpt-1')
    envdir = os.path.join(tmp, 'envmaps')
    os.makedirs(envdir)
    return paths, envdir


def run_nerfactor(tmp, learned):
    brdf_root = os.path.join(tmp, 'merl_npz')
    os.makedirs(brdf_root)
    for name in gi.BRDF_NAMES:
        open(os.path.join(brdf_root, 'train_%s.npz' % name), 'wb').close()
    paths, envdir = workdir(tmp, brdf_root)
    tag = 'nfl' if learned else 'nfm'
    ini = 'nerfactor.ini' if learned else 'nerfactor_microfacet.ini'
    over = dict(shape_model_ckpt=paths['shape'], test_envmap_dir=envdir, embed_light_h=16, light_tv_weight=2e-4,
                light_achro_weight=1e-4)
    if learned:
        over['brdf_model_ckpt'] = paths['brdf']
    cls = NerfactorModel if learned else MicrofacetModel
    cfg = ref_config(ini, **over)
    model = cls(cfg, debug=True)
    z_dim = 3 if learned else 1
    net = gi.nerfactor_net(z_dim)
    for part in net:
        set_layers(model.net[part], net[part])
    if learned:
        bnet = gi.brdf_net()
        set_layers(model.brdf_model.net['brdf_mlp'], bnet['brdf_mlp'])
        set_layers(model.brdf_model.net['brdf_out'], bnet['brdf_out'])
    model._light = tf.Variable(gi.light_probe(gi.LIGHT_SCALE[tag]))
    batch, n = shape_batch(512)
    train(tag, model, batch, n, cfg)


# ------------------------------------------------------------------------------------------------ BRDF prior
def run_brdf(tmp):
    """models/brdf.py:87-136 on the 192 (identity, Rusinkiewicz, reflectance) rows of golden_inputs.brdf_batch():
    gradients of the MLP and of the latent codes (tf.gather_nd of a variable), loss on log reflectance."""
    root = os.path.join(tmp, 'merl_npz')
    os.makedirs(root)
    for name in gi.BRDF_NAMES:
        open(os.path.join(root, 'train_%s.npz' % name), 'wb').close()
    cfg = ref_config('brdf.ini', data_root=root)
    model = BrdfModel(cfg)
    bnet = gi.brdf_net()
    set_layers(model.net['brdf_mlp'], bnet['brdf_mlp'])
    set_layers(model.net['brdf_out'], bnet['brdf_out'])
    model.latent_code.z = gi.latent_codes()
    i, rusink, refl = gi.brdf_batch()
    n = rusink.shape[0]
    batch = (np.array([b'x'] * n), tf.convert_to_tensor(i), None, None, None,
             tf.convert_to_tensor(rusink), tf.convert_to_tensor(refl))
    train('brdf', model, batch, n, cfg)


def main():
    tf.random.set_seed(11)
    run_nerf()
    with tempfile.TemporaryDirectory() as tmp:
        run_nerfactor(os.path.join(tmp, 'a'), learned=False)
        run_nerfactor(os.path.join(tmp, 'b'), learned=True)
        run_brdf(os.path.join(tmp, 'c'))
    path = os.path.join(HERE, 'reference_grads.npz')
    np.savez_compressed(path, **OUT)
    print('wrote %s (%.1f KiB, %d arrays)' % (path, os.path.getsize(path) / 1024, len(OUT)))


if __name__ == '__main__':
    main()
