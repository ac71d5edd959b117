"""The dynamic symbol table of libnfx.so is exactly the C-ABI of include/nfx.h: the objects are built
with -fvisibility=hidden, the header marks its functions NFX_API, and the linker's version script makes hipcc's host-side
kernel handles local.  No GPU needed."""
import re
import subprocess

import pytest

from tests.conftest import ROOT


def _header_symbols():
    src = open(ROOT + '/include/nfx.h').read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'^NFX_API\s+[a-z_ ]+?\**\s*\**(nfx_[a-z0-9_]+)\s*\(', src, flags=re.M)))


def test_exported_symbols_are_the_header(nfx_lib):
    out = subprocess.run(['nm', '-D', '--defined-only', nfx_lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    header = _header_symbols()
    assert len(header) >= 54
    assert exported == header, (sorted(set(exported) - set(header)), sorted(set(header) - set(exported)))
    # and the ctypes binding declares every one of them
    assert sorted(nfx_lib.SIGNATURES) == header


def test_options_round_trip_and_unknown_keys(nfx_lib):
    for key in nfx_lib.OPTION_KEYS:
        before = nfx_lib.get_option(key)
        with nfx_lib.option(key, 3):
            assert nfx_lib.get_option(key) == 3
            with nfx_lib.option(key, None):
                assert nfx_lib.get_option(key) is None
            assert nfx_lib.get_option(key) == 3
        assert nfx_lib.get_option(key) == before
    with pytest.raises(nfx_lib.NfxError, match='unknown option'):
        nfx_lib.set_option('no_such_option', 1)
    with pytest.raises(nfx_lib.NfxError):
        nfx_lib.get_option('no_such_option')


def test_library_reads_no_environment_variable(nfx_lib):
    """Per-call getenv dispatch is gone: no getenv reference in the library's dynamic imports."""
    out = subprocess.run(['nm', '-D', '--undefined-only', nfx_lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    assert 'getenv' not in out
