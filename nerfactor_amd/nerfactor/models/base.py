"""models.base.Model — the plugin contract of the reference (nerfactor/models/base.py:25-143) on
torch.nn.Module: `config` (ConfigParser), `net` dict of networks, `register_trainable()`,
`call(batch, mode)` -> (pred, gt, loss_kwargs, to_vis), `compute_loss`, `vis_batch`,
`compile_batch_vis`, `trainable_variables`.
"""
import torch

from .. import losses
from ..networks import base as basenet


class Model(torch.nn.Module):
    def __init__(self, config, debug=False):
        super().__init__()
        self.config = config
        self.debug = debug
        self.net = {'main': basenet.Network()}
        self.trainable_registered = False
        self.wloss = self._init_loss()
        self.precision = config.get('DEFAULT', 'precision', fallback='bf16')  # MFMA operand type
        self._blobs = {}  # packed-weight cache: key -> (versions, device blob)

    # ------------------------------------------------------------------ loss string parsing
    def _init_loss(self):
        wloss = []
        for item in self.config.get('DEFAULT', 'loss').split(','):
            name, weight = self._parse_loss_and_weight(item)
            if name == 'l1':
                fn = losses.L1()
            elif name == 'l2':
                fn = losses.L2()
            else:  # the reference names lpips/elpips/ssim too; no shipped config uses them
                raise NotImplementedError(name)
            wloss.append((weight, fn))
        return wloss

    @staticmethod
    def _parse_loss_and_weight(weight_loss_str):
        """'1e+2l2' -> ('l2', 100.); 'l2' -> ('l2', 1.): longest float-parsable prefix wins."""
        s = weight_loss_str.strip()
        for cut in range(len(s), 0, -1):
            try:
                return s[cut:], float(s[:cut])
            except ValueError:
                pass
        return s, 1.

    # ------------------------------------------------------------------ trainable bookkeeping
    def register_trainable(self):
        """Expose every built layer of every network in `self.net` as a direct submodule named
        net_<netname>_layer<i> — the names the reference's checkpoints are keyed on
        (models/base.py:81-104)."""
        for net_name, net in self.net.items():
            attr = 'net_' + net_name
            if not attr.isidentifier():
                raise ValueError("network name %r does not make a valid attribute" % net_name)
            for i, layer in enumerate(net.layers):
                full = '%s_layer%d' % (attr, i)
                if hasattr(self, full):
                    if getattr(self, full) is layer:
                        continue
                    raise ValueError("Can't register `%s`: attribute exists" % full)
                self.add_module(full, layer)
        self.trainable_registered = True

    @property
    def trainable_variables(self):
        return [p for p in self.parameters() if p.requires_grad]

    @staticmethod
    def _validate_mode(mode):
        if mode not in ('train', 'vali', 'test'):
            raise ValueError(mode)

    # ------------------------------------------------------------------ packed-weight cache
    def _packed(self, key, tensors, pack_fn):
        """Device blob for `tensors` (Keras-layout parameters), re-packed only when one of them
        was modified in place (optimizer step, checkpoint restore)."""
        versions = tuple((t.data_ptr(), t._version) for t in tensors)
        hit = self._blobs.get(key)
        dev = tensors[0].device
        if hit is None or hit[0] != versions or hit[1].device != dev:
            blob = pack_fn().to(dev)
            self._blobs[key] = (versions, blob)
            return blob
        return hit[1]

    # ------------------------------------------------------------------ contract
    def forward(self, batch, mode='train', **kwargs):
        return self.call(batch, mode=mode, **kwargs)

    def call(self, batch, mode='train'):
        raise NotImplementedError

    def compute_loss(self, pred, gt, **kwargs):
        raise NotImplementedError

    def vis_batch(self, data_dict, outdir, mode='train', dump_raw_to=None):
        raise NotImplementedError

    def compile_batch_vis(self, batch_vis_dirs, outpref, mode='train'):
        raise NotImplementedError
