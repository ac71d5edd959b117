"""util.config / util.io helpers the models need (reference: nerfactor/util/config.py:25-26,
nerfactor/util/io.py:36-45, 48-52)."""
from configparser import ConfigParser
from os.path import exists

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
    """Load a torch checkpoint written by this framework's trainvali ({'net': state_dict, ...})."""
    model.register_trainable()
    state = torch.load(ckpt_path, map_location='cpu')
    state = state.get('net', state)
    missing, unexpected = model.load_state_dict(state, strict=strict)
    return missing, unexpected


def ckpt_available(path):
    return bool(path) and path.lower() not in ('none', 'null', '') and exists(path)
