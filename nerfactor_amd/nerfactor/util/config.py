"""util.config / util.io helpers the models need (reference: nerfactor/util/config.py:25-26,
nerfactor/util/io.py:36-45, 48-52)."""
from configparser import ConfigParser
from os.path import exists

import numpy as np
import torch


def get_config_ini(ckpt_path):
    """<run dir>/checkpoints/ckpt-N  ->  <run dir>.ini (the effective config trainvali.py dumps)."""
    return '/'.join(ckpt_path.split('/')[:-2]) + '.ini'


def read_config(path):
    config = ConfigParser()
    with open(path, 'r') as h:
        config.read_file(h)
    return config


def restore_model(model, ckpt_path, strict=False):
    """Load a checkpoint into `model`: a torch file written by this framework's trainvali ({'net': state_dict, ...}),
    or — when `<ckpt_path>.index` exists — a TensorFlow checkpoint of the reference (util/tf_ckpt.py; Keras kernels are
    already [in, out], the layout of networks.layers.Dense)."""
    model.register_trainable()
    from . import tf_ckpt
    if tf_ckpt.is_tf_checkpoint(ckpt_path):
        state = {k: torch.from_numpy(np.ascontiguousarray(v))
                 for k, v in tf_ckpt.to_state_dict(tf_ckpt.load_tensors(ckpt_path)).items()}
        own = model.state_dict()
        for k in list(state):          # scalars / shape mismatches are reported, not silently reshaped
            if k in own and tuple(own[k].shape) != tuple(state[k].shape):
                raise ValueError("checkpoint tensor %s has shape %s, the model expects %s" % (
                    k, tuple(state[k].shape), tuple(own[k].shape)))
    else:
        state = torch.load(ckpt_path, map_location='cpu')
        state = state.get('net', state)
    missing, unexpected = model.load_state_dict(state, strict=strict)
    return missing, unexpected


def ckpt_available(path):
    return bool(path) and path.lower() not in ('none', 'null', '') and (exists(path) or exists(path + '.index'))
