"""How far are the training gradients from the REFERENCE's (tests/golden/reference_grads.npz: trainvali.py:273-285 of the unmodified
reference, every gradient tensor of step 1) in each precision mode the plugin offers, and what does a step cost there?
VERDICT r05 weak #2: bf16 training is 7-35 % (relative Frobenius, worst tensor) from the reference and the mode that matches it
costs 5-12 x — this prints the mode in between as well:

    precision = bf16                              bf16 operands forward and backward (the default)
    precision = fp32, grad_precision = bf16       fp32-class forward (hi / lo operand pairs), bf16-operand backward kernels
    precision = fp32                              the model's default fp32 matrix mode (pairs | native)

    python scripts/grad_modes.py      (one MI355X)"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import reference_steps as R  # noqa: E402

cuda = torch.device('cuda:0')
MODES = [("bf16", dict(precision='bf16')),
         ("fp32 forward + bf16 gradients", dict(precision='fp32', grad_precision='bf16')),
         ("fp32", dict(precision='fp32'))]
out = {}
for name in ('nerfactor_microfacet', 'nerfactor', 'nerf'):
    tag = R.TAG_OF[name]
    out[name] = {}
    for label, kw in MODES:
        kw = dict(kw)
        prec = kw.pop('precision')
        t0 = time.perf_counter()
        model, losses, grad1 = (R.run_nerf(cuda, prec, **kw) if tag == 'nerf' else R.run_nerfactor(tag, cuda, prec, **kw))
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        fro = lambda a, b: float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))
        rep = {}
        for pname, p in model.named_parameters():
            if p.requires_grad:
                want, got = R.elements('%s/grad/%s' % (tag, pname), grad1[pname])
                rep[pname] = fro(got, want)
        want_losses = np.asarray(R.FIX[tag + '/loss'], dtype=np.float64)
        worst = max(rep.items(), key=lambda kv: kv[1])
        vals = np.sort(np.asarray(list(rep.values())))
        out[name][label] = {"grad_rel_frobenius_vs_reference_worst": worst[1], "worst_tensor": worst[0],
                            "grad_rel_frobenius_median_tensor": float(np.median(vals)), "gradient_tensors": len(rep),
                            "loss_step1_rel_err": float(abs(losses[0] / want_losses[0] - 1)),
                            "loss_trajectory_max_rel_err": float(np.max(np.abs(np.asarray(losses) / want_losses - 1))),
                            "wall_s_10_steps_with_setup": round(wall, 2)}
print(json.dumps(out, indent=1))
