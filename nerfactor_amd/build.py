"""Builds nerfactor_amd/libnfx.so (gfx950 only) with hipcc.  No torch involved: the library is a
plain HIP shared object behind the C-ABI of include/nfx.h.

    python -m nerfactor_amd.build [--force] [--verbose]

Objects are cached under build/ (git-ignored) keyed on source + header mtimes; the .so stays
in-tree so it travels to the GPU box with the repository snapshot.
"""
import fcntl
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
OBJDIR = os.path.join(ROOT, 'build', 'obj')
LIB = os.path.join(HERE, 'libnfx.so')

HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# -ffp-contract=off: the fp32 stages follow the reference's op order (mul then add), fused
# multiply-adds appear only where written as fmaf().
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-fvisibility=hidden',
         '-Wno-unused-result', '-I' + os.path.join(ROOT, 'include')]
if os.environ.get('NFX_EXTRA_DEFS'):  # experiment switches, e.g. NFX_EXTRA_DEFS='-DNFX_V5_BIAS_COPY'
    FLAGS += os.environ['NFX_EXTRA_DEFS'].split()


# Per-source extra flags.  -amdgpu-mfma-vgpr-form: MFMA accumulators in ArchVGPRs (the epilogue reads them without
# v_accvgpr_read; activations move to AccVGPRs, which MFMA takes as B operands).  NFX_VGPR_FORM_FILES overrides the list.
VGPR_FORM = ['-mllvm', '-amdgpu-mfma-vgpr-form']
# r01: lvis 21.16 -> 20.79 ms, NeRF render 1228 -> 1239 TFLOP/s (instruction count of the lvis kernel 5380 -> 4554),
# variant 6 1293 -> 1330 TFLOP/s, density-gradient kernel 72.5 -> 68.4 ms per 256 x 256 view; no effect on the
# backward kernels (nerf_bwd, mlp128_bwd, brdf_bwd)
# r04: mlp128_bwd_fused.hip — 1113 of ~3850 VALU instructions per tile of its PART 1 kernel were v_accvgpr_read / _write
# of the chain's accumulators (the persistent weight-gradient blocks still end up in AccVGPRs: only MFMAs touch them)
# r05: shade.hip — the shading kernels are VALU-bound by the per-light GGX term, not HBM-bound (static count: 4364 / 3070 VALU
# instructions in shade_kernel / shade_olat_kernel, ~3000 executed per point and wave); a quarter of them were the IEEE
# division / square-root sequences (v_div_scale, v_div_fmas, v_div_fixup around v_rcp / v_rsq: 528 v_div_scale in the file).
# The 2.5-ulp forms (v_rcp_f32 + one Newton step) leave the render inside its 3e-4 of the fp32 oracle
# (tests/test_gpu_nerfactor.py::test_shade_microfacet_vs_oracle): 4364 -> 3263 and 3070 -> 2348 instructions.
FAST_DIV = ['-fno-hip-fp32-correctly-rounded-divide-sqrt']
PER_FILE_FLAGS = {'lvis_v2.hip': VGPR_FORM, 'nerf_mlp_v6.hip': VGPR_FORM, 'nerf_sigma_v6.hip': VGPR_FORM,
                  'nerf_geom.hip': VGPR_FORM, 'mlp128_bwd_fused.hip': VGPR_FORM, 'shade.hip': FAST_DIV}
if os.environ.get('NFX_VGPR_FORM_FILES') is not None:      # A/B experiment libraries: only the VGPR-form entries are overridden —
    # shade.hip keeps FAST_DIV, so an experiment build divides exactly like the shipped library
    PER_FILE_FLAGS = dict({f: v for f, v in PER_FILE_FLAGS.items() if v is not VGPR_FORM},
                          **{f: VGPR_FORM for f in os.environ['NFX_VGPR_FORM_FILES'].split(',') if f})


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(('.hip', '.cpp')))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hpp')]
    hs.append(os.path.join(ROOT, 'include', 'nfx.h'))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, force, verbose):
    obj = os.path.join(OBJDIR, src + '.o')
    spath = os.path.join(CSRC, src)
    newest = max(os.path.getmtime(spath), _headers_mtime())
    for line in open(spath):       # a translation unit that includes another .hip (mlp_generic_{x3,native}.hip) follows its changes
        if line.startswith('#include "') and line.rstrip().endswith('.hip"'):
            newest = max(newest, os.path.getmtime(os.path.join(CSRC, line.split('"')[1])))
    cmd = [HIPCC] + FLAGS + PER_FILE_FLAGS.get(src, []) + ['-x', 'hip', '-c', spath, '-o', obj]
    stamp = obj + '.cmd'     # the command line is part of the staleness check (per-file flags, NFX_EXTRA_DEFS)
    same_cmd = os.path.exists(stamp) and open(stamp).read() == ' '.join(cmd)
    if not force and same_cmd and os.path.exists(obj) and os.path.getmtime(obj) >= newest:
        return obj, False
    if verbose:
        print(' '.join(cmd), flush=True)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError('hipcc failed on %s:\n%s' % (src, res.stdout))
    with open(stamp, 'w') as h:
        h.write(' '.join(cmd))
    return obj, True


def build(force=False, verbose=False, out=None):
    """out: alternative .so path (experiment builds with NFX_EXTRA_DEFS; objects go to build/obj_<name>)."""
    global OBJDIR, LIB
    if out:
        LIB = os.path.abspath(out)
        OBJDIR = os.path.join(ROOT, 'build', 'obj_' + os.path.splitext(os.path.basename(out))[0])
    os.makedirs(OBJDIR, exist_ok=True)
    # one builder at a time: the N ranks of a multi-GPU launch all call build(); the first compiles, the others wait
    # on the lock and then find everything up to date
    with open(os.path.join(OBJDIR, '.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        srcs = _sources()
        with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
            results = list(ex.map(lambda s: _compile(s, force, verbose), srcs))
        objs = [o for o, _ in results]
        rebuilt = any(r for _, r in results)
        if rebuilt or not os.path.exists(LIB):
            tmp = LIB + '.tmp.%d' % os.getpid()
            cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC',
                   '-Wl,--version-script=' + os.path.join(CSRC, 'libnfx.map')] + objs + ['-o', tmp]
            if verbose:
                print(' '.join(cmd), flush=True)
            res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if res.returncode != 0:
                raise RuntimeError('link failed:\n' + res.stdout)
            os.replace(tmp, LIB)      # a process that already mapped the old file keeps its inode
    return LIB


if __name__ == '__main__':
    out = sys.argv[sys.argv.index('--out') + 1] if '--out' in sys.argv else None
    path = build(force='--force' in sys.argv, verbose='--verbose' in sys.argv or '-v' in sys.argv, out=out)
    print(path)
