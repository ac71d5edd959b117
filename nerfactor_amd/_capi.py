"""ctypes binding of libnfx.so (include/nfx.h).  The library is mandatory: there is no Python or
torch fallback for any hot-path op — a missing or stale .so raises at import of this module."""
import ctypes
import os

# torch first: it brings its own HIP runtime (torch/lib/libamdhip64.so); libnfx.so must bind to
# that copy, not to a second runtime from /opt/rocm loaded earlier (two runtimes in one process
# cannot both own the device: "no ROCm-capable device is detected").
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libnfx.so')
if os.environ.get('NFX_LIB_PATH'):   # experiment builds (python -m nerfactor_amd.build --out ...)
    LIB_PATH = os.environ['NFX_LIB_PATH']

PREC_BF16, PREC_FP32, PREC_FP32_NATIVE = 0, 1, 2    # (PREC_FP32_NATIVE: the runtime-shaped kernels only)
ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SOFTPLUS = 0, 1, 2, 3
IN_XYZ, IN_XYZ_LDIR, IN_Z_RUSINK = 0, 1, 2

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "%s not found: build it with `python -m nerfactor_amd.build` (or __graft_entry__.build()); "
        "nerfactor_amd has no fallback path." % LIB_PATH)

lib = ctypes.CDLL(LIB_PATH)

_p = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_f = ctypes.c_float
_sz = ctypes.c_size_t
_pp = ctypes.POINTER(ctypes.c_void_p)

# name -> (restype, argtypes); every symbol declared in include/nfx.h
SIGNATURES = {
    'nfx_version': (_i, []),
    'nfx_last_error': (_i, [ctypes.c_char_p, _sz]),
    'nfx_set_option': (_i, [ctypes.c_char_p, _i]),
    'nfx_unset_option': (_i, [ctypes.c_char_p]),
    'nfx_get_option': (_i, [ctypes.c_char_p, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    'nfx_nerf_packed_bytes': (_sz, [_i]),
    'nfx_nerf_pack_weights': (_i, [_pp, _pp, _i, _p, _sz]),
    'nfx_mlp128_packed_bytes': (_sz, [_i, _i, _i, _i]),
    'nfx_mlp128_pack_weights': (_i, [_pp, _pp, _i, _i, _i, _i, _p, _sz]),
    'nfx_l2_normalize3': (_i, [_p, _p, _i64, _f, _p]),
    'nfx_any_nonfinite': (_i, [_p, _i64, _p, _p]),
    'nfx_scatter_rows': (_i, [_p, _p, _i64, _i, _p, _p]),
    'nfx_gen_z': (_i, [_f, _f, _i, _i64, _i, _p, _p, _p]),
    'nfx_nerf_mlp_fwd': (_i, [_p, _p, _p, _i64, _i, _p, _i, _p, _p]),
    'nfx_composite_fwd': (_i, [_p, _p, _p, _p, _i64, _i, _i, _p, _p, _p, _p, _p, _p]),
    'nfx_sample_fine': (_i, [_p, _p, _i64, _i, _i, _p, _p, _p]),
    'nfx_mlp128_xyz_fwd': (_i, [_p, _i64, _f, _p, _i, _i, _f, _f, _i, _p, _p]),
    'nfx_lvis_workspace_bytes': (_sz, [_i64]),
    'nfx_lvis_fwd': (_i, [_p, _p, _i64, _f, _p, _i, _p, _i, _p, _sz, _p, _p]),
    'nfx_shade_lds_bytes': (_sz, [_i, _i]),
    'nfx_shade_fwd': (_i, [_p, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _i64, _i, _i, _i, _p, _p]),
    'nfx_shade_olat_fwd': (_i, [_p, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p, _f, _f, _i64, _i, _i, _p,
                                _p]),
    'nfx_brdf_spec_fwd': (_i, [_p, _p, _p, _p, _i, _p, _i, _p, _i, _i64, _p, _p]),
    'nfx_dir2rusink': (_i, [_p, _p, _i64, _p, _p]),
    'nfx_mlp128_train_packed_bytes': (_sz, [_i]),
    'nfx_mlp128_pack_train_weights': (_i, [_pp, _pp, _i, _i, _i, _p, _sz]),
    'nfx_mlp128_bwd_workspace_bytes': (_sz, [_i, _i64, _i]),
    'nfx_mlp128_bwd': (_i, [_i, _p, _p, _i64, _f, _p, _i, _p, _i, _i, _f, _p, _p, _sz, _pp, _pp, _i, _p]),
    'nfx_mlp128_bwd_heads': (_i, [_i, _p, _p, _i64, _f, _p, _i, _i, _pp, _p, _p, _p, _pp, _p, _sz, _pp, _pp, _i, _p]),
    'nfx_composite_bwd': (_i, [_p, _p, _p, _p, _i64, _i, _i, _p, _p, _p]),
    'nfx_nerf_train_packed_bytes': (_sz, [_i]),
    'nfx_nerf_pack_train_weights': (_i, [_pp, _pp, _i, _p, _sz]),
    'nfx_nerf_bwd_workspace_bytes': (_sz, [_i64, _i]),
    'nfx_nerf_mlp_bwd': (_i, [_p, _p, _p, _i64, _i, _p, _i, _p, _p, _sz, _pp, _pp, _p]),
    'nfx_brdf_train_packed_bytes': (_sz, []),
    'nfx_brdf_pack_train_weights': (_i, [_pp, _pp, _i, _i, _p, _sz]),
    'nfx_brdf_spec_bwd_workspace_bytes': (_sz, [_i, _i64]),
    'nfx_brdf_spec_bwd': (_i, [_p, _p, _p, _p, _i, _p, _i, _p, _i, _i64, _p, _p, _p, _p, _sz, _p]),
    'nfx_brdf_spec_bwd_list_bytes': (_sz, [_i64, _i]),
    'nfx_brdf_spec_bwd_rows': (_i, [_p, _p, _p, _p, _i, _p, _i, _p, _i, _i64, _p, _p, _p, _p, _sz, _p, _sz, _p]),
    'nfx_brdf_rows_fwd': (_i, [_p, _i, _p, _i64, _i, _p, _i, _p, _p]),
    'nfx_brdf_rows_bwd_workspace_bytes': (_sz, [_i, _i64, _i]),
    'nfx_brdf_rows_bwd': (_i, [_p, _i, _p, _i64, _i, _p, _i, _p, _p, _sz, _p, _pp, _pp, _p]),
    'nfx_shade_bwd_workspace_bytes': (_sz, [_i]),
    'nfx_shade_bwd': (_i, [_p, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _i64, _i, _i, _p, _p, _p, _p, _p, _p,
                           _p, _p, _sz, _p]),
    'nfx_l2_normalize_rows': (_i, [_p, _p, _i64, _i, _f, _p]),
    'nfx_l2_normalize_rows_bwd': (_i, [_p, _p, _p, _i64, _i, _f, _p]),
    'nfx_light_smoothness': (_i, [_p, _i, _i, _f, _f, _p, _p, _p]),
    'nfx_pair_loss_fwd': (_i, [_p, _i, _p, _f, _i64, _p, _p]),
    'nfx_pair_loss_bwd': (_i, [_p, _i, _p, _f, _i64, _p, _p]),
    'nfx_pack_gather': (_i, [_p, _p, _i64, _p, _p]),
    'nfx_amsgrad_step': (_i, [_p, _p, _p, _p, _p, _i64, _f, _f, _f, _f, _i64, _p]),
    'nfx_amsgrad_step_size': (_f, [_f, _f, _f, _i64]),
    'nfx_amsgrad_step_dev': (_i, [_p, _p, _p, _p, _p, _i64, _p, _f, _f, _f, _p]),
    'nfx_nerf_sigma_fwd': (_i, [_p, _p, _p, _i64, _i, _p, _i, _p, _p]),
    'nfx_lvis_fwd_rows': (_i, [_p, _p, _i64, _f, _p, _i, _p, _i, _p, _sz, _p, _p, _p, _p]),
    'nfx_zero_rows': (_i, [_p, _p, _i64, _i, _p]),
    'nfx_shade_fwd_rows': (_i, [_p, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _p, _i64, _i, _i, _i, _p, _p]),
    'nfx_shade_olat_fwd_rows': (_i, [_p, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _f, _f, _i64, _i, _i, _p, _p, _p, _p]),
    'nfx_nerf_refine_select': (_i, [_p, _p, _p, _i64, _i, _f, _f, _f, _f, _i, _p, _p, _p]),
    'nfx_nerf_sigma_refine': (_i, [_p, _p, _p, _i64, _i, _p, _p, _p, _p, _p]),
    'nfx_nerf_geom_packed_bytes': (_sz, [_i]),
    'nfx_nerf_pack_geom_weights': (_i, [_pp, _pp, _i, _p, _sz]),
    'nfx_nerf_sigma_grad': (_i, [_p, _p, _p, _i64, _i, _p, _i, _p, _p]),
    'nfx_nerf_sigma_grad_workspace_bytes': (_sz, [_i64, _i]),
    'nfx_nerf_sigma_grad_rows': (_i, [_p, _p, _p, _i64, _i, _p, _i, _p, _p, _sz, _p]),
    'nfx_selftest_mfma_bf16': (_i, [_p, _p, _p, _p]),
    'nfx_selftest_sincos': (_i, [_p, _i64, _i, _p, _p]),
    'nfx_selftest_tr16': (_i, [_p, _p, _p, _i, _p]),
    'nfx_mlp_generic_packed_bytes': (_sz, [_i, _i, _p, _p, _i]),
    'nfx_mlp_generic_pack': (_i, [_pp, _pp, _i, _i, _p, _p, _i, _p, _sz]),
    'nfx_mlp_generic_fwd': (_i, [_p, _i64, _i, _i, _i, _p, _p, _p, _p, _i, _p, _i, _i, _p]),
    'nfx_mlp_generic_train_packed_bytes': (_sz, [_i, _i, _p, _p, _i]),
    'nfx_mlp_generic_pack_train': (_i, [_pp, _pp, _i, _i, _p, _p, _i, _p, _sz]),
    'nfx_mlp_generic_bwd_workspace_bytes': (_sz, [_i64, _i, _i, _p, _p, _i]),
    'nfx_mlp_generic_split_hilo': (_i, [_p, _i, _i, _p, _p, _i, _p]),
    'nfx_brdf_rows_geom_fwd': (_i, [_p, _p, _p, _p, _i, _p, _i, _i64, _i, _p, _i, _p, _p]),
    'nfx_brdf_rows_geom_bwd': (_i, [_p, _p, _p, _i, _p, _i, _i64, _i, _p, _i, _p, _p, _p]),
    'nfx_embed_bwd': (_i, [_p, _i64, _i, _i, _p, _i, _i, _p, _p]),
    'nfx_mlp_generic_bwd': (_i, [_p, _i64, _i, _i, _i, _p, _p, _p, _p, _i, _p, _i, _i, _p, _i, _pp, _pp, _p, _sz, _p]),
    'nfx_embed': (_i, [_p, _p, _p, _i64, _i, _i, _i, _i, _p, _i, _i, _p]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = header/library out of sync
    _fn.restype = _res
    _fn.argtypes = _args


class NfxError(RuntimeError):
    """Raised for any non-zero return of the C-ABI; mirrors how the reference surfaces failures as
    Python exceptions (tf.errors.InvalidArgumentError from tf.debugging.*)."""


def last_error():
    buf = ctypes.create_string_buffer(512)
    lib.nfx_last_error(buf, 512)
    return buf.value.decode()


def check(rc, what):
    if rc != 0:
        raise NfxError("%s failed (%d): %s" % (what, rc, last_error()))


# ------------------------------------------------------------------------------- options
OPTION_KEYS = ('nerf_variant', 'nerf_blocks', 'm128_blocks', 'lvis_variant', 'brdf_variant', 'brdf_ct', 'nerf_bwd',
               'nerf_bwd_nw', 'm128_bwd', 'wgrad_lds', 'wgrad_slabs', 'wgrad_rounds', 'wgrad_narrow', 'wgrad_fused', 'lvis_verify', 'lvis_rows', 'brdf_bwd_rows', 'nerf_bwd_rows', 'sigma_grad_rows', 'sigma_variant')


def set_option(key, value):
    """nfx_set_option: a process-wide integer option of the library (kernel-variant selectors, grid sizes)."""
    check(lib.nfx_set_option(key.encode(), int(value)), 'nfx_set_option(%s)' % key)


def unset_option(key):
    check(lib.nfx_unset_option(key.encode()), 'nfx_unset_option(%s)' % key)


def get_option(key):
    """-> the set value, or None when the option is at its default."""
    v, isset = _i(0), _i(0)
    check(lib.nfx_get_option(key.encode(), ctypes.byref(v), ctypes.byref(isset)), 'nfx_get_option(%s)' % key)
    return v.value if isset.value else None


class option:
    """with option('brdf_ct', 8): ...   — sets an option for the block and restores what was there before."""

    def __init__(self, key, value):
        self.key, self.value = key, value

    def __enter__(self):
        self.prev = get_option(self.key)
        if self.value is None:
            unset_option(self.key)
        else:
            set_option(self.key, self.value)
        return self

    def __exit__(self, *exc):
        if self.prev is None:
            unset_option(self.key)
        else:
            set_option(self.key, self.prev)
        return False


# The library reads no environment variable; the binding forwards NFX_<KEY> once, here, so that
# `NFX_BRDF_CT=8 python bench.py` keeps working for A/B runs from a shell.
for _k in OPTION_KEYS:
    _v = os.environ.get('NFX_' + _k.upper())
    if _v not in (None, ''):
        set_option(_k, int(_v))
