"""Host logic of the precision / shape routing (no GPU): which models count as the shipped ("tuned") shapes, what
`precision` / `grad_precision` combinations exist, and that nothing but the documented cases raises at construction
(INTEGRATION.md "Shapes"; reference nerfactor/models/{nerf,shape,brdf}.py build their networks from the ini)."""
import pytest

from nerfactor_amd.nerfactor.config import make_config
from nerfactor_amd.nerfactor.models import get_model_class


def build(name, **ov):
    return get_model_class(name)(make_config(name, **ov))


def test_shipped_configurations_are_tuned():
    assert build('nerf').tuned and build('brdf').tuned
    shape = build('shape')
    assert shape._net_tuned('normal_mlp') and shape._net_tuned('lvis_mlp')


@pytest.mark.parametrize("name,ov", [
    ('nerf', dict(mlp_width='128')), ('nerf', dict(enc_depth='6')), ('nerf', dict(use_views='False')),
    ('nerf', dict(pos_enc='False')), ('nerf', dict(n_freqs_xyz='8')),
    ('brdf', dict(mlp_width='64')), ('brdf', dict(n_freqs='3')), ('brdf', dict(pos_enc='False')), ('brdf', dict(mlp_skip_at='1'))])
def test_other_shapes_construct_and_take_the_runtime_shaped_path(name, ov):
    for prec in ('bf16', 'fp32'):
        assert not build(name, precision=prec, **ov).tuned


@pytest.mark.parametrize("ov", [dict(mlp_width='64'), dict(mlp_depth='3', mlp_skip_at='1'), dict(n_freqs_xyz='6'),
                                dict(n_freqs_ldir='2')])
def test_surface_model_shapes_are_checked_per_network(ov):
    m = build('shape', **ov)
    assert not m._net_tuned('normal_mlp') or not m._net_tuned('lvis_mlp')


def test_limits_of_the_runtime_shaped_kernels_raise_with_the_numbers():
    with pytest.raises(NotImplementedError, match='mlp_width'):
        build('nerf', mlp_width='640')
    with pytest.raises(NotImplementedError, match='mlp_width'):
        build('shape', mlp_width='600')
    assert not build('brdf', mlp_depth='4', mlp_skip_at='3').tuned      # (a skip behind the body's last layer: the head reads concat(y, x))
    with pytest.raises(NotImplementedError, match='skip'):
        build('brdf', mlp_depth='4', mlp_skip_at='4')
    assert not build('shape', pos_enc='False').tuned and build('shape', pos_enc='False').embedder['xyz'].n_freqs == 0
    enc2 = build('nerf', enc_depth='2', mlp_width='64')                # the skip sits behind the last encoder layer
    assert tuple(enc2.net['coarse_sigma_out'].layers[0].kernel.shape) == (64 + 63, 1)


def test_grad_precision_follows_precision_and_rejects_the_impossible_pair():
    assert build('nerf').grad_precision == 'bf16'
    assert build('nerf', precision='fp32').grad_precision == 'fp32'
    assert build('nerf', precision='fp32', grad_precision='bf16').grad_precision == 'bf16'
    with pytest.raises(ValueError, match='grad_precision'):
        build('nerf', grad_precision='fp32')
    with pytest.raises(ValueError, match='grad_precision'):
        build('shape', precision='fp32', grad_precision='fp16')


def test_fp32_matrix_defaults_per_model_and_is_an_ini_key():
    """precision = fp32 runs the runtime-shaped kernels with bf16 hi / lo operand pairs where that keeps the gradients
    within the stated 1e-3 of the reference's (surface models, BRDF prior) and with the native fp32 matrix instruction for
    NeRF; `fp32_matrix` overrides either way; precision = bf16 never reaches an fp32 instantiation."""
    assert build('nerf', precision='fp32').fp32_matrix == 'native' and build('nerf', precision='fp32').generic_prec == 'fp32_native'
    for name in ('shape', 'brdf'):
        m = build(name, precision='fp32')
        assert m.fp32_matrix == 'pairs' and m.generic_prec == 'fp32'
    assert build('nerf', precision='fp32', fp32_matrix='pairs').generic_prec == 'fp32'
    assert build('shape', precision='fp32', fp32_matrix='native').generic_prec == 'fp32_native'
    assert build('shape').generic_prec == 'bf16' and build('nerf', fp32_matrix='native').generic_prec == 'bf16'
    with pytest.raises(ValueError, match='fp32_matrix'):
        build('shape', precision='fp32', fp32_matrix='tf32')
