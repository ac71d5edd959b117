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
    already [in, out], the layout of networks.layers.Dense).  A TensorFlow checkpoint that leaves ANY parameter of the
    model unset raises: `tf.train.Checkpoint.restore` matches by object graph, so a silent partial load here would
    mean the names were resolved wrongly and e.g. a frozen BRDF prior stays at its random initialisation."""
    model.register_trainable()
    from . import tf_ckpt
    if tf_ckpt.is_tf_checkpoint(ckpt_path):
        own = model.state_dict()
        blob = tf_ckpt.read_string_tensor(ckpt_path, '_CHECKPOINTABLE_OBJECT_GRAPH')
        graph = tf_ckpt.parse_object_graph(blob) if blob else None
        state = {k: torch.from_numpy(np.ascontiguousarray(v))
                 for k, v in tf_ckpt.to_state_dict(tf_ckpt.load_tensors(ckpt_path), graph=graph, wanted=list(own)).items()
                 if k in own}
        for k in list(state):          # scalars / shape mismatches are reported, not silently reshaped
            if tuple(own[k].shape) != tuple(state[k].shape):
                raise ValueError("checkpoint tensor %s has shape %s, the model expects %s" % (
                    k, tuple(state[k].shape), tuple(own[k].shape)))
        missing, unexpected = model.load_state_dict(state, strict=False)
        missing = [k for k in missing if k not in _DERIVED_BUFFERS]
        if missing:
            raise KeyError("TensorFlow checkpoint %s sets none of: %s%s" % (
                ckpt_path, ', '.join(missing[:8]), ' ...' if len(missing) > 8 else ''))
        return missing, unexpected
    state = torch.load(ckpt_path, map_location='cpu')
    state = state.get('net', state)
    missing, unexpected = model.load_state_dict(state, strict=strict)
    return missing, unexpected


# buffers the models compute themselves (light positions / solid angles): never part of a reference checkpoint
_DERIVED_BUFFERS = ('lxyz', 'lareas')


def ckpt_available(path):
    return bool(path) and path.lower() not in ('none', 'null', '') and (exists(path) or exists(path + '.index'))
