"""The bench line the driver parses, checked on the line committed as this round's evidence (profiles/r05/bench.json =
the stdout of `python bench.py --gpus 1 --steps 20 --warmup 3` on an MI355X): keys and types of the contract, the
roofline / cpu_baseline objects, and the arithmetic a reader can redo from the line itself.  No GPU needed; a change of
bench.py's output format that forgets the contract, or evidence that no longer matches it, fails here."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = os.path.join(ROOT, 'profiles', 'r05', 'bench.json')


@pytest.fixture(scope='module')
def line():
    text = open(LINE).read().strip().splitlines()
    assert len(text) == 1, "ONE JSON line"
    return json.loads(text[0])


def check_roofline(r, bound):
    assert r['bound'] == bound
    assert r['unit'] == ('TFLOP/s' if bound == 'mfma' else 'GB/s')
    assert r['peak'] == (2500.0 if bound == 'mfma' else 8000.0)     # dense bf16 MFMA / HBM3E (MI355X_MICROARCH.md)
    assert 0 < r['achieved'] < r['peak']
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    assert 'traffic' in r


def test_top_level_contract(line):
    base = json.load(open(os.path.join(ROOT, 'BASELINE.json')))
    assert line['metric'].split(' ')[0] == base['metric'].split(' ')[0] == 'rays/sec'
    assert line['unit'] == 'rays/s' and line['higher_is_better'] is True
    assert (line['n_gpus'], line['steps'], line['warmup']) == (1, 20, 3)
    assert line['scaling'] == 'weak' and line['data'] == 'synthetic' and line['dtype'] == 'bf16'
    assert line['vs_baseline'] is None                 # BASELINE.md holds no published number for this metric
    assert line['world_size'] == 1 and 'collective_backend' in line
    cfg = line['config']
    assert 'workload' in cfg and 'model' not in cfg
    # value = rays of all ranks / time of one step
    rays = cfg['views_per_step'] * cfg['rays_per_view']
    assert abs(line['value'] - rays / (line['ms_per_step'] * 1e-3)) < 1e-6 * line['value']
    assert (cfg['n_samples_coarse'], cfg['n_samples_fine'], cfg['rays_per_view']) == (64, 128, 800 * 800)


def test_roofline_and_cpu_baseline(line):
    r = line['roofline']
    check_roofline(r, 'mfma')
    # achieved = algorithmic FLOPs of a coarse + fine launch pair / its HIP-event time
    assert abs(r['achieved'] - r['flop_per_launch_pair'] / (r['avg_launch_pair_ms'] * 1e-3) / 1e12) < 1e-6 * r['achieved']
    assert r['flop_per_launch_pair'] == 800 * 800 * (64 + 64 + 128) * 1186816      # FLOP per sample point
    assert r['avg_launch_pair_ms'] <= line['ms_per_step']
    assert r['traffic'] >= r['algorithmic_hbm_gb'] and 'traffic_source' in r
    assert r['traffic_source'].startswith('measured by this run')                 # counters of THIS command, not a committed digest
    c = line['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['unit'] == 'rays/s' and c['cores'] >= 1 and c['value'] > 0
    assert 'sample' in c
    p = line['parity']
    assert p['psnr_db'] >= 40 and p['max_abs'] <= 3e-2                             # north_star's tolerance ...
    assert p['rays_excluded_from_max_abs'] == 0 and p['max_abs_all_rays'] == p['max_abs']   # ... on EVERY ray (r04)
    assert p['frac_rays_above_3e-2'] == 0 and p['rays_compared'] >= 4096
    pf = line['parity_fitted_weights']
    assert pf['psnr_db'] >= 40 and pf['rays_excluded_from_max_abs'] == 0 and pf['frac_rays_above_3e-2'] <= 0.01


def test_every_leg_of_the_metric_is_on_the_line(line):
    for name in ('nerfactor_microfacet', 'nerfactor'):                            # BASELINE.json configs[2]
        leg = line['nerfactor'][name]
        check_roofline(leg['roofline'], 'mfma')
        assert leg['ms_per_step'] > 0 and leg['cpu_baseline']['kind'] == 'port'
        par = leg['parity']
        assert par['max_abs'] <= 3e-2 and par['points_above_3e-2'] == 0 and par['grazing_points_excluded_from_max_abs'] == 0
    for name in ('nerfactor_microfacet', 'nerfactor', 'nerf'):                    # configs[3]
        leg = line['train'][name]
        check_roofline(leg['roofline'], 'mfma')
        assert leg['steps'] >= 20 and leg['final_loss'] < leg['first_loss'] and 'collective' in leg and 'error' not in leg
        assert leg['ms_per_step'] <= leg['ms_per_step_eager'] * 1.02 and 'step' in leg
        par = leg['parity']                          # the reference's own ten steps through the same train path (r04)
        tol = par['tolerance']
        assert par['grad_rel_frobenius_vs_bf16_oracle_worst'] <= tol['grad_vs_bf16_oracle']
        assert par['loss_step1_rel_err'] <= tol['loss_step1'] and par['loss_trajectory_max_rel_err'] <= tol['loss_trajectory']
        assert par['params_after_10_steps_mean_dev_in_lr_steps'] <= tol['params_mean_dev'] and par['gradient_tensors'] >= 10
        f32 = leg['fp32']                            # the same step at the reference's own arithmetic (VERDICT r03 missing #1)
        assert f32['ms_per_step'] > leg['ms_per_step'] and 'fp32' in f32['what']
        p32 = f32['parity']
        assert p32['gradient_tensors'] == par['gradient_tensors']
        assert p32['grad_rel_frobenius_vs_reference_worst'] <= p32['tolerance']['grad_vs_reference']
        assert p32['grad_rel_frobenius_vs_reference_worst'] <= 1e-3
        assert p32['loss_trajectory_max_rel_err'] <= (1e-3 if name != 'nerfactor' else 5e-3)
        assert p32['fp32_matrix'] == f32['fp32_matrix_default'] and set(f32['ms_per_step_by_mode']) >= {'pairs_eager', 'native_eager'}
    assert line['train']['nerfactor_microfacet']['ms_per_step'] <= 1.35           # VERDICT r04 #2 (r04: 1.59 ms)
    assert line['train']['nerfactor_microfacet']['roofline']['frac'] >= 0.13
    assert line['train']['nerfactor_microfacet']['fp32']['ms_per_step'] <= 6.5    # <= 6 ms (r04: 16.8; profiles/r05/bench_train_fp32.jsonl: 5.8)
    assert line['train']['nerfactor_microfacet']['fp32']['fp32_matrix_default'] == 'pairs'
    geo = line['geometry']                                                        # the geometry stage is measured
    check_roofline(geo['depth_normal']['roofline'], 'mfma')
    check_roofline(geo['light_visibility']['roofline'], 'mfma')
    assert geo['finite'] and geo['depth_normal']['rays_per_s'] > 0 and geo['light_visibility']['pairs_per_s'] > 0
    assert geo['cpu_baseline']['kind'] == 'port' and geo['parity']['lvis_max_abs'] <= 3e-2 and geo['parity']['occu_max_abs'] <= 3e-2
    olat = line['olat']                                                           # configs[4], OLAT half
    check_roofline(olat['roofline'], 'hbm')
    assert olat['ms_per_step'] > 0
    sweep = line['relight']                                                       # configs[4], probe half: 4 views x 8 probes
    assert sweep['views_per_step'] == 4 and sweep['probes'] == 8 and sweep['ms_per_view'] > 0
    f32 = line['fp32_class']                                                      # the headline at the reference's precision
    check_roofline(f32['roofline'], 'mfma')
    assert f32['parity']['rays_excluded_from_max_abs'] == 0 and f32['parity']['max_abs_all_rays'] <= 2e-3
    assert f32['parity']['q99_abs'] <= 2e-4 and f32['parity']['psnr_db'] >= 55


def test_bench_defaults_and_cpu_exit():
    """No flags = one GPU and a step count that finishes in minutes; without a GPU the script says so and exits
    non-zero instead of measuring anything else."""
    src = open(os.path.join(ROOT, 'bench.py')).read()
    assert "add_argument('--gpus', type=int, default=1)" in src
    assert "add_argument('--steps', type=int, default=5)" in src and "add_argument('--warmup', type=int, default=2)" in src
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '0'],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert res.returncode != 0
    assert not any(l.startswith('{"metric"') for l in res.stdout.splitlines())
