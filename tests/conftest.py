import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "determinism: soak / bit-identity / launch-to-launch determinism tests; "
                                       "collected LAST so that a failure there cannot hide a parity test under -x")


# Order of the GPU suite: the tests that hold the HIP path to the REFERENCE's own outputs first, then the
# per-kernel oracle tests, then training, the drivers, and the determinism / soak tests at the very end.
_FILE_ORDER = ('test_gpu_reference_golden', 'test_gpu_reference_grads', 'test_gpu_nerf', 'test_gpu_nerfactor',
               'test_gpu_train', 'test_gpu_drivers')


def pytest_collection_modifyitems(config, items):
    def key(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        rank = _FILE_ORDER.index(mod) if mod in _FILE_ORDER else len(_FILE_ORDER)
        return (1 if item.get_closest_marker('determinism') else 0, rank)
    items.sort(key=key)   # stable: the order inside a file is kept


@pytest.fixture(scope="session")
def nfx_lib():
    """libnfx.so, built on demand (hipcc cross-compiles gfx950 without a GPU)."""
    from nerfactor_amd import build
    build.build()
    from nerfactor_amd import _capi
    return _capi


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


@pytest.fixture
def nfx_opt(nfx_lib):
    """Library options (nfx_set_option) for the duration of one test: .set(key, value) / .unset(key); whatever the
    test changed is put back afterwards.  (The library reads no environment variable.)"""
    class Opt:
        def __init__(self):
            self.saved = {}

        def _remember(self, key):
            if key not in self.saved:
                self.saved[key] = nfx_lib.get_option(key)

        def set(self, key, value):
            self._remember(key)
            nfx_lib.set_option(key, int(value))

        def unset(self, key):
            self._remember(key)
            nfx_lib.unset_option(key)

    o = Opt()
    yield o
    for key, prev in o.saved.items():
        if prev is None:
            nfx_lib.unset_option(key)
        else:
            nfx_lib.set_option(key, prev)
