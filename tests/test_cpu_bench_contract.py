"""The bench line the driver parses.  Round 5's line (20.7 KB, an `Infinity` in it) came back from the driver as
`parsed: null`; since round 6 bench.py prints a COMPACT strict-JSON line (<= 6 KB) and writes everything else to
bench_detail.json.  Checked here WITHOUT a GPU and without a hand-kept copy of the line:

  * bench.assemble / compact / emit — the very functions main() ends with — are run on leg dictionaries of the real
    shape (the committed detail of an MI355X run, and a doctored one full of inf / nan / NumPy scalars), and the text they
    print is parsed with a parser that REJECTS the non-standard constants;
  * the committed evidence of the round (profiles/r06/bench_line.json = stdout of the driver's command
    `python bench.py --gpus 1 --steps 20 --warmup 5`, profiles/r06/bench_detail.json) is held to the same contract and to
    the arithmetic a reader can redo from the line itself."""
import copy
import io
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

R06_LINE = os.path.join(ROOT, 'profiles', 'r06', 'bench_line.json')
R06_DETAIL = os.path.join(ROOT, 'profiles', 'r06', 'bench_detail.json')
R05_FULL = os.path.join(ROOT, 'profiles', 'r05', 'bench.json')       # round 5's one-line FULL object: leg dictionaries of the real shape

CONTRACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def _reject(name):
    raise ValueError("non-standard JSON constant %s on the bench line" % name)


def parse_strict(text):
    assert '\n' not in text.strip(), "ONE line"
    assert len(text.encode()) <= bench.LINE_MAX_BYTES, "line of %d bytes" % len(text.encode())
    return json.loads(text, parse_constant=_reject)


def legs_of(full):
    """Split a full result object back into what the legs return (what main() hands to assemble)."""
    top = ("metric", "unit", "n_gpus", "steps", "warmup", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
           "world_size", "collective_backend", "psnr_db", "max_abs")
    sub = ('nerfactor', 'train', 'olat', 'relight', 'fp32_class', 'geometry', 'wall_s')
    legs = {k: full[k] for k in sub if k in full}
    legs['nerf'] = {k: v for k, v in full.items() if k not in top and k not in sub}
    return legs


def check_roofline(r, bound):
    assert r['bound'] == bound
    assert r['unit'] == ('TFLOP/s' if bound == 'mfma' else 'GB/s')
    assert r['peak'] == (2500.0 if bound == 'mfma' else 8000.0)     # dense bf16 MFMA / HBM3E (MI355X_MICROARCH.md)
    assert 0 < r['achieved'] < r['peak']
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-5
    assert 'traffic' in r


def check_line(line, steps, warmup):
    assert CONTRACT_KEYS <= set(line), CONTRACT_KEYS - set(line)
    base = json.load(open(os.path.join(ROOT, 'BASELINE.json')))
    assert line['metric'].split(' ')[0] == base['metric'].split(' ')[0] == 'rays/sec'
    assert line['unit'] == 'rays/s' and line['higher_is_better'] is True
    assert (line['n_gpus'], line['steps'], line['warmup']) == (1, steps, warmup)
    assert line['scaling'] == 'weak' and line['data'] == 'synthetic' and line['dtype'] == 'bf16'
    assert line['vs_baseline'] is None                 # BASELINE.md holds no published number for this metric
    cfg = line['config']
    assert 'workload' in cfg and 'model' not in cfg and 'configs[1]' in cfg['workload']
    rays = cfg['views_per_step'] * cfg['rays_per_view']                          # value = rays of all ranks / time of one step
    assert abs(line['value'] - rays / (line['ms_per_step'] * 1e-3)) < 1e-4 * line['value']
    assert (cfg['n_samples_coarse'], cfg['n_samples_fine'], cfg['rays_per_view']) == (64, 128, 800 * 800)
    r = line['roofline']
    check_roofline(r, 'mfma')
    assert r['flop_per_launch_pair'] == 800 * 800 * (64 + 64 + 128) * 1186816      # FLOP per sample point
    assert abs(r['achieved'] - r['flop_per_launch_pair'] / (r['avg_launch_pair_ms'] * 1e-3) / 1e12) < 1e-4 * r['achieved']
    assert r['avg_launch_pair_ms'] <= line['ms_per_step']
    assert r['traffic'] is None or r['traffic'] >= r['algorithmic_hbm_gb']
    assert 'traffic_source' in r
    c = line['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['unit'] == 'rays/s' and c['cores'] >= 1 and c['value'] > 0 and c['sample']
    p = line['parity']
    assert set(p) >= {'psnr_db', 'max_abs', 'rays_compared', 'rays_above_tol'}
    assert (p['psnr_db'] is None and p.get('identical')) or p['psnr_db'] >= 40
    assert p['max_abs'] <= 3e-2 and p['rays_compared'] >= 4096
    for name, leg in line['legs'].items():
        small = {k: v for k, v in leg.items() if isinstance(v, (dict, list))}
        assert not small, "legs.%s must be flat: %s" % (name, list(small))
    assert line['detail'] == bench.DETAIL_NAME


@pytest.fixture(scope='module')
def r05_full():
    text = open(R05_FULL).read().strip().splitlines()
    assert len(text) == 1
    return json.loads(text[0])


def test_compact_line_from_real_leg_dictionaries(r05_full, tmp_path):
    """main()'s own assembly code on the leg dictionaries of an MI355X run: the printed text is one strict-JSON line
    under the byte limit with the contract's keys; the detail file holds everything."""
    legs = legs_of(r05_full)
    legs['wall_s'] = {'nerf': 31.2}
    full = bench.assemble(20, 3, 'weak', 'bf16', 1, "none (single process)", False, legs)
    buf = io.StringIO()
    text = bench.emit(full, stream=buf, detail_dir=str(tmp_path))
    assert buf.getvalue() == text + '\n'
    line = parse_strict(text)
    check_line(line, 20, 3)
    assert len(text) <= 4096, "the r05-shaped line should sit well under the limit: %d" % len(text)
    assert set(line['legs']) >= {'nerfactor_microfacet', 'nerfactor', 'train_nerfactor_microfacet', 'train_nerfactor', 'train_nerf',
                                 'olat', 'relight', 'fp32_class', 'geometry'}
    assert line['legs']['train_nerf']['ms_per_step'] == pytest.approx(r05_full['train']['nerf']['ms_per_step'], rel=1e-5)
    assert line['roofline']['frac'] == pytest.approx(r05_full['roofline']['frac'], rel=1e-5)
    detail = json.load(open(tmp_path / bench.DETAIL_NAME), parse_constant=_reject)
    assert detail['train']['nerf']['parity'] == bench.strict(r05_full['train']['nerf']['parity'])
    assert detail['wall_s'] == {'nerf': 31.2}


def test_non_finite_values_never_reach_the_line(r05_full, tmp_path):
    """inf / nan / NumPy scalars anywhere in a leg: the line and the detail file stay strict JSON; an identical frame is
    `psnr_db: null, identical: true`."""
    legs = copy.deepcopy(legs_of(r05_full))
    legs['nerf']['parity']['psnr_db'] = float('inf')
    legs['nerf']['parity']['max_abs'] = np.float32(0.)
    legs['nerf']['parity_fitted_weights']['psnr_db'] = float('inf')
    legs['nerf']['roofline']['traffic'] = float('nan')
    legs['geometry']['parity']['lvis_max_abs'] = np.float64('nan')
    legs['olat']['roofline']['frac'] = np.float32(0.33)
    legs['train']['nerf'] = {"error": "check_numerics raised " + "x" * 500}
    full = bench.assemble(np.int64(20), 5, 'weak', 'bf16', 1, "nccl", False, legs)
    text = bench.emit(full, stream=io.StringIO(), detail_dir=str(tmp_path))
    assert 'Infinity' not in text and 'NaN' not in text
    line = parse_strict(text)
    assert line['parity']['psnr_db'] is None and line['parity']['identical'] is True
    assert line['parity']['fitted_weights']['identical'] is True
    assert line['roofline']['traffic'] is None and line['collective_backend'] == 'nccl'
    assert len(line['legs']['train_nerf']['error']) <= 120
    json.load(open(tmp_path / bench.DETAIL_NAME), parse_constant=_reject)


def test_an_oversized_line_sheds_the_legs_not_the_contract(r05_full, tmp_path):
    legs = legs_of(r05_full)
    legs['nerfactor'] = {('model_%03d' % i): legs['nerfactor']['nerfactor'] for i in range(120)}
    text = bench.emit(bench.assemble(20, 5, 'weak', 'bf16', 1, "none", False, legs), stream=io.StringIO(), detail_dir=str(tmp_path))
    line = parse_strict(text)
    assert 'dropped' in line['legs'] and CONTRACT_KEYS <= set(line)


def test_committed_r06_evidence():
    """profiles/r06/bench_line.json is the stdout of the DRIVER's command on an MI355X; bench_detail.json its detail file."""
    if not os.path.exists(R06_LINE):
        pytest.skip("no round-6 line committed yet")
    text = open(R06_LINE).read().strip()
    line = parse_strict(text)
    check_line(line, 20, 5)
    assert line['parity']['rays_above_tol'] == 0
    # the same frame against outputs of the reference's own Python (tests/golden/reference_large.npz), glorot and fitted weights
    for block in (line['parity']['vs_reference_python'], line['parity']['fitted_weights']['vs_reference_python']):
        assert block['rays_compared'] == 8192 and block['rays_above_tol'] == 0 and block['max_abs'] <= 3e-2 and block['psnr_db'] >= 40
    assert line['parity']['fitted_weights']['rays_above_tol'] == 0                 # DEFAULT ini: coarse_precision = auto
    assert line['parity']['fitted_weights']['refine_extra_frame_time'] <= 0.15
    legs = line['legs']
    assert legs['train_nerfactor_microfacet']['ms_per_step'] <= 1.35 and legs['train_nerfactor_microfacet']['frac'] >= 0.13
    assert legs['nerfactor_microfacet']['max_abs'] <= 3e-2 and legs['nerfactor']['max_abs'] <= 3e-2
    assert legs['fp32_class']['max_abs'] <= 2e-3
    assert legs['geometry']['depth_rel_of_range'] <= 0.04                           # depth: every ray; normal: >= 98 % of the rays inside 8e-2,
    #   counted on the line (tests/test_gpu_reference_golden.py::_excluded holds the fixtures to the same 2 %)
    detail = json.load(open(R06_DETAIL), parse_constant=_reject)
    assert legs['geometry']['rays_above_8e-2'] <= 0.02 * legs['geometry']['rays_compared']
    assert bench.compact(detail) == line                                           # the line IS the digest of the detail file
    for name in ('nerfactor_microfacet', 'nerfactor', 'nerf'):                    # configs[3]
        leg = detail['train'][name]
        check_roofline(leg['roofline'], 'mfma')
        assert leg['final_loss'] < leg['first_loss'] and 'error' not in leg
        par, f32 = leg['parity'], leg['fp32']
        tol = par['tolerance']
        assert par['grad_rel_frobenius_vs_bf16_oracle_worst'] <= tol['grad_vs_bf16_oracle']
        assert par['loss_step1_rel_err'] <= tol['loss_step1'] and par['loss_trajectory_max_rel_err'] <= tol['loss_trajectory']
        assert f32['parity']['grad_rel_frobenius_vs_reference_worst'] <= 1e-3
    assert sum(detail['wall_s'].values()) <= 120, detail['wall_s']


def test_bench_defaults_and_cpu_exit():
    """No flags = one GPU and a step count that finishes in minutes; without a GPU the script says so and exits
    non-zero instead of measuring anything else; the counter children are opt-in."""
    src = open(os.path.join(ROOT, 'bench.py')).read()
    assert "add_argument('--gpus', type=int, default=1)" in src
    assert "add_argument('--steps', type=int, default=5)" in src and "add_argument('--warmup', type=int, default=2)" in src
    assert "add_argument('--measure-traffic', action='store_true'" in src and bench.MEASURE_TRAFFIC is False
    assert "allow_nan=False" in src
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '0'],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert res.returncode != 0
    assert not any(l.startswith('{"metric"') for l in res.stdout.splitlines())


def test_more_gpus_than_the_box_has_is_a_json_error_line():
    """`python bench.py --gpus 8` with no torchrun around it launches its own ranks; on a box with fewer GPUs it prints
    ONE JSON line with an `error` key and exits non-zero (here: 0 GPUs)."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        pytest.skip("8 GPUs present")
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'NFX_BENCH_REHEARSAL', 'NFX_REHEARSAL')}
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '1'], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert res.returncode == 2, res.stderr[-400:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    err = json.loads(lines[0], parse_constant=_reject)
    assert 'error' in err and err['n_gpus'] == 8 and err['value'] is None
