#!/usr/bin/env python
"""bench.py — rays/sec (+ PSNR vs the CPU oracle) of the NeRFactor per-ray rendering hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong] [--legs nerf,nerfactor_microfacet,nerfactor]

BASELINE.json's metric is "rays/sec (64 samples/ray, 512 light dirs) + PSNR vs TF ref"; one run times, back to back,

  * leg `nerf` (the headline `value`; BASELINE.json configs[1], SURVEY.md §8d "C2"): 800x800 view, 64 coarse + 128
    fine samples per ray, white background, perturb off, synthetic glorot weights (opaque variant), rays from the
    reference's pin-hole generator.  A "step" = one full pass of the hot path over one view's 640 000 rays per GPU:
    normalise -> gen_z -> coarse MLP -> composite -> hierarchical resample -> fine MLP -> composite;
  * legs `nerfactor_microfacet` and `nerfactor` (configs[2], "C3", reported under the "nerfactor" key of the same
    JSON object): full NeRFactor render of 800x800 surface points (60 % foreground), 512 light directions, the
    trained light + 8 novel probes, through the model plugin (`Model.call(mode='test', relight_probes=True)`):
    normals / albedo / BRDF-latent MLPs -> light-visibility MLP over the light sphere -> (learned BRDF MLP |
    microfacet BRDF) -> BRDF x visibility x lighting integral -> pixels.

  * leg `train` (configs[3], "C4", under the "train" key): optim.train_step of `nerfactor_microfacet`, `nerfactor` and
    `nerf` at 1024 rays per GPU and step (weak scaling, `n_rays_per_step` of config/*.ini) — forward, fused loss,
    backward through the libnfx kernels, ONE all-reduce of [gradients | loss] (RCCL when N > 1), fused AMSGrad;
  * leg `olat` (the OLAT half of configs[4], under the "olat" key): one 800x800 view with `relight_olat=True`
    (512 one-light-at-a-time renders per point, nerfactor.py:348-364), HBM-bound on the rows it writes.

Inputs are resident in HBM before every timed region; every timed region is bracketed by barrier +
torch.cuda.synchronize() and EXACTLY K steps long.

N > 1 (one process per GPU under torchrun; rays are independent, no data-path collective): `--scaling weak`
(default) renders N views per step, EVERY view sharded by contiguous ray ranges over all N ranks (SURVEY §8e:
rank r owns rays [r n/N, (r+1) n/N) of each view), so per-GPU work stays 640 000 rays per step; `--scaling strong`
renders ONE view per step sharded the same way.  value = all rays of all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0.  `roofline` describes the dominant kernel of the headline leg (fused NeRF MLP,
MFMA-bound: algorithmic FLOPs / HIP-event time of its launches on the launch stream); each NeRFactor leg carries
its own `roofline` (light-visibility kernel) and at N == 1 every leg carries a `cpu_baseline` (torch-CPU fp32 port of
the reference op sequence, bounded sample) whose output doubles as the PSNR / max-abs reference for the GPU frame.
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this driver stack shares memory through dmabuf only: without it RCCL's start-up fails with
# "hipIpcGetMemHandle: invalid argument".  Read by the runtime when it starts — so before torch is imported; the ranks
# bench.py launches for --gpus N inherit it.
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W = 800
N_COARSE, N_FINE = 64, 128
FLOP_PER_POINT = 2 * 593408          # 593 408 MAC per NeRF sample point
LVIS_MAC, BRDF_MAC = 72320, 53888    # MAC per (point, light) row
HEAD_MAC = 65664 + 65664             # normal + albedo heads per point (+ 65 408 | 65 920 for the BRDF latent)
PEAK_BF16_TFLOPS = 2500.0            # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
N_LIGHTS, N_PROBES = 512, 8
PEAK_HBM_GBS = 8000.0                # MI355X HBM3E (MI355X_MICROARCH.md)
PMC_RENDER_LEGS = 'nerf,nerfactor_microfacet,olat'      # the legs of measure_traffic's counter run (one pair of passes)


# ------------------------------------------------------------------------------------------------ helpers
def host_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def psnr_uint8_luma(a, b):
    """xiuminglib.metric.PSNR('uint8') semantics (the reference's per-view PSNR, models/nerf.py:389-391): luma
    0.2126/0.7152/0.0722 of the frames quantised by truncation to uint8."""
    def q(x):
        return (np.clip(x, 0, 1) * 255).astype(np.uint8).astype(np.float64)
    w = np.array([0.2126, 0.7152, 0.0722])
    mse = float(np.mean((q(a) @ w - q(b) @ w) ** 2))
    return float('inf') if mse == 0 else 10 * np.log10(255. ** 2 / mse)


class Shards:
    """The contiguous ray ranges this rank owns of each view of one step."""

    def __init__(self, n_per_view, rank, world, scaling):
        from nerfactor_amd import dist as nfx_dist
        self.n_views = world if scaling == 'weak' else 1
        self.lo, self.hi = nfx_dist.shard_range(n_per_view, rank, world)
        self.rays_per_step_all_ranks = self.n_views * n_per_view


def timed(step, steps, warmup, barrier):
    for _ in range(warmup):
        step(None)
    barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        out = step(k)
    barrier()
    return time.perf_counter() - t0, out


# ------------------------------------------------------------------------------------------------ NeRF leg
COARSE_REFINE_GATE = 4e-3      # models/nerf.py `coarse_refine_gate` (ini default)


def coarse_refine_decision(ops, view, nets, refine, prec='bf16'):
    """What the plugin's `coarse_precision = auto` decides for these weights (models/nerf.py:_coarse_refine_on): the measured
    bf16 density error as an alpha error, and whether the selective fp32-class refinement of the coarse pass is on.  4096 rays of
    the view through the bf16 and the fp32-class DENSITY kernels (the bf16 one is bit-identical to the MLP kernel's sigma): the
    headline kernel is not launched here, so its coarse and fine launches stay 1 : 1 in every profile of this command."""
    if refine is None or prec != 'bf16':
        return False, (None, 0.)
    from nerfactor_amd import synth
    o, d_raw = view
    d = ops.l2_normalize3(d_raw, 1e-12)
    z = ops.gen_z(2., 6., N_COARSE, o.shape[0], device=o.device)
    g16 = ops.pack_nerf_geom_weights(*synth.nerf_layers(nets[0]), prec='bf16').to(o.device)
    err_a, err_s = ops.nerf_coarse_error(o, d, z, None, refine[0], geom_blob_bf16=g16)
    return err_a > COARSE_REFINE_GATE, (err_a, err_s)


def nerf_render_step(ops, views, blobs, ev=None, prec='bf16', refine=None, coarse_prec=None, refine_coarse=False, margin=0.):
    """views: [(rayo, rayd)] device tensors (this rank's shard of every view).  ev: per-view 4 events around the two
    MLP launches.  refine: the two fp32-class density blobs — the last sample of every ray is re-evaluated with them
    (what models/nerf.py does when rendering with precision = bf16: ops.nerf_refine_last_sample); inside the timed step,
    outside the dominant kernel's event pairs."""
    rgb = None
    for i, (o, d_raw) in enumerate(views):
        e = None if ev is None else ev[i]
        d = ops.l2_normalize3(d_raw, 1e-12)
        z = ops.gen_z(2., 6., N_COARSE, o.shape[0], device=o.device)
        if e is not None:
            e[0].record()
        raw = ops.nerf_mlp_fwd(o, d, z, blobs[0], coarse_prec or prec)     # (coarse_prec: the plugin's ini key coarse_precision)
        if e is not None:
            e[1].record()
        if refine is not None and (coarse_prec or prec) == 'bf16':
            ops.nerf_refine_last_sample(o, d, z, raw, refine[0])
            if refine_coarse:       # (the plugin's coarse_precision = auto / select: the deciding coarse samples fp32-class)
                ops.nerf_refine_coarse(o, d, z, raw, refine[0], sigma_margin=margin)
        _, _, _, _, w = ops.composite_fwd(raw, z, d, white_bg=True)
        z_all = ops.sample_fine(z, w, N_FINE)
        if e is not None:
            e[2].record()
        raw = ops.nerf_mlp_fwd(o, d, z_all, blobs[1], prec)
        if e is not None:
            e[3].record()
        if refine is not None:
            ops.nerf_refine_last_sample(o, d, z_all, raw, refine[1])
        rgb = ops.composite_fwd(raw, z_all, d, white_bg=True, want_weights=False)[0]
    return rgb


def nerf_cpu_reference(nets, rayo, rayd, budget_s, timed_run=True):
    """torch-CPU fp32 port of the reference op sequence (oracle/torch_ref.py) on a random subset of the view.
    Returns (cpu_baseline dict | None, ray indices, reference rgb, |sigma_last| of the coarse and fine pass)."""
    from oracle import torch_ref
    tn = [torch_ref.to_torch_net(n) for n in nets]
    idx = np.random.default_rng(0).permutation(rayo.shape[0])
    avail = host_cores()
    cands = sorted({c for c in (avail, avail // 2, 64, 32, 16, 8) if 1 <= c <= avail}, reverse=True)

    def run(n, threads):
        torch.set_num_threads(threads)
        o, d = torch.from_numpy(rayo[idx[:n]]), torch.from_numpy(rayd[idx[:n]])
        t0 = time.perf_counter()
        res = torch_ref.render_rays(o, d, tn[0], tn[1])
        return time.perf_counter() - t0, res

    with torch.no_grad():
        if not timed_run:      # N > 1: reference frame only (1024 rays), no baseline timing
            dt, res = run(1024, cands[0])
            return None, idx[:1024], res
        run(64, cands[-1])  # warm the allocator / BLAS
        probe = {c: run(256, c)[0] for c in cands}
        best = min(probe, key=probe.get)
        n = int(min(65536, max(256, 256 * budget_s / max(probe[best], 1e-3))))
        n = max(256, (n // 256) * 256)
        dt, res = run(n, best)
    base = {"value": n / dt, "unit": "rays/s", "cores": best, "kind": "port", "cores_available": avail,
            "sample": "%d rays of the same 800x800 view, 64+128 samples, torch-CPU fp32 (oracle/torch_ref.py, "
                      "mlp_chunk=65536), %.1f s; thread probe %s" % (n, dt, {k: round(v, 2) for k, v in probe.items()})}
    return base, idx[:n], res



_LIVE_TRAFFIC = {}     # legs -> digest of a counter run of THIS command (measure_traffic)
MEASURE_TRAFFIC = False     # --measure-traffic: spawn the two rocprofv3 counter children (off by default: ~50 s of a driver-timed run)


def measure_traffic(legs, extra=()):
    """HBM traffic per launch of every nfx kernel of `legs`, MEASURED by this run: two child runs of
    this very script (1 step, no CPU work) under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `WRITE_SIZE` — separate passes,
    kernel trace only, as MI355X_MICROARCH.md prescribes (the two counters do not fit one pass) — digested by
    scripts/pmc_digest.py (FETCH_SIZE x 2: the gfx950 correction for wide coalesced reads).  Returns {} when rocprofv3 is
    not on PATH, when this process is itself a child or already runs under a profiler, or when a pass fails or times out:
    the caller then falls back to the committed constant and the line says which it is."""
    import shutil
    import subprocess
    import tempfile
    key = (legs,) + tuple(extra)
    if key in _LIVE_TRAFFIC:
        return _LIVE_TRAFFIC[key]
    _LIVE_TRAFFIC[key] = {}
    if not MEASURE_TRAFFIC:
        return {}
    if (os.environ.get('NFX_BENCH_CHILD') or os.environ.get('NFX_BENCH_NO_PMC') or not shutil.which('rocprofv3')
            or any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ)):
        return {}
    sys.path.insert(0, os.path.join(ROOT, 'scripts'))
    try:
        import pmc_digest
        with tempfile.TemporaryDirectory(prefix='nfx_pmc_', dir='/tmp') as tmp:
            env = dict(os.environ, NFX_BENCH_CHILD='1', TMPDIR='/tmp')
            for name, counter in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
                cmd = ['rocprofv3', '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', os.path.join(tmp, name),
                       '-o', 'p', '--', sys.executable, os.path.join(ROOT, 'bench.py'), '--legs', legs, '--steps', '1',
                       '--warmup', '1', '--no-cpu-baseline', '--no-hip-graph'] + list(extra)
                res = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240)
                if res.returncode != 0:
                    return {}
                # rocprofv3 writes <dir>/<host>/<pid>_... : hand pmc_digest one directory per pass
                found = [os.path.join(r, f) for r, _, fs in os.walk(os.path.join(tmp, name)) for f in fs if f.endswith('counter_collection.csv')]
                if not found:
                    return {}
                os.makedirs(os.path.join(tmp, 'digest', name), exist_ok=True)
                shutil.copy(found[0], os.path.join(tmp, 'digest', name, 'p_counter_collection.csv'))
            _LIVE_TRAFFIC[key] = pmc_digest.digest(os.path.join(tmp, 'digest'))
    except Exception:        # (a profiler problem must never take the bench line down)
        _LIVE_TRAFFIC[key] = {}
    return _LIVE_TRAFFIC[key]


def live_traffic(legs, kernel_substr, scale=1, extra=()):
    """(GB per launch, source) from measure_traffic, or (None, None)."""
    for k, v in measure_traffic(legs, extra).items():
        if kernel_substr in k and 'hbm_write_bytes' in v and 'hbm_read_bytes_corrected' in v:
            return (scale * (v['hbm_read_bytes_corrected'] + v['hbm_write_bytes']) / 1e9,
                    "measured by this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of a 1-step child of this command "
                    "(FETCH_SIZE x 2 for gfx950), average of %d dispatches" % int(v.get('dispatches', 0)))
    return None, None


def committed_traffic(kernel_substr, scale=1):
    """(GB per launch, source) of a kernel from the newest committed PMC digest (scripts/gpu_r03_final.sh ->
    scripts/pmc_digest.py: FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, averaged over the kernel's dispatches of
    this very command at N = 1).  rocprofv3 counter passes cannot run inside this process, so the figure is a committed
    constant and the line says so."""
    for rnd, name in (('r06', 'pmc_digest.json'), ('r06', 'pmc_train_digest.json'), ('r05', 'pmc_digest.json'), ('r05', 'pmc_train_digest.json'), ('r04', 'pmc_digest.json'), ('r04', 'pmc_train_digest.json'), ('r03', 'pmc_digest.json'),
                      ('r02', 'pmc_digest.json'), ('r01', 'pmc_variant7_digest.json'), ('r01', 'pmc_nerfactor_digest.json')):
        dig = os.path.join(ROOT, 'profiles', rnd, name)
        if not os.path.exists(dig):
            continue
        for k, v in json.load(open(dig)).items():
            if isinstance(v, dict) and kernel_substr in k and 'hbm_write_bytes' in v:
                return (scale * (v.get('hbm_read_bytes_corrected', 0) + v['hbm_write_bytes']) / 1e9,
                        "committed PMC digest profiles/%s/%s (not measured in this run)" % (rnd, name))
    return None, None

def nerf_leg(args, ops, dev, rank, world, barrier, max_over_ranks):
    from nerfactor_amd import synth
    nets = synth.nerf_nets(seed=0)
    blobs = [ops.pack_nerf_weights(*synth.nerf_layers(n), prec=args.precision).to(dev) for n in nets]
    # precision = bf16 renders re-evaluate every ray's last sample fp32-class (models/nerf.py, last_sample_precision)
    refine = None if args.precision == 'fp32' or args.no_last_sample_refine else \
        [ops.pack_nerf_geom_weights(*synth.nerf_layers(n), prec='fp32').to(dev) for n in nets]
    sh = Shards(H * W, rank, world, args.scaling)
    views, host_views = [], []
    for v in range(sh.n_views):
        ang = 0.7 * v
        cam = (4 * np.cos(ang) * 0.8, 4 * np.sin(ang) * 0.8 - 0.1, 4 * 0.6)
        rayo, rayd = synth.camera_rays(H, W, cam_loc=cam)
        host_views.append((rayo, rayd))
        views.append((torch.from_numpy(rayo[sh.lo:sh.hi]).to(dev), torch.from_numpy(rayd[sh.lo:sh.hi]).to(dev)))
    evs = [[[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in views] for _ in range(args.steps)]
    # coarse_precision = auto: measured once per weight version, as the plugin does — outside the timed region
    refine_coarse, (alpha_err, sigma_err) = coarse_refine_decision(ops, views[0], nets, refine, args.precision)
    margin = ops.REFINE_MARGIN_FACTOR * sigma_err
    elapsed, rgb = timed(lambda k: nerf_render_step(ops, views, blobs, None if k is None else evs[k], args.precision,
                                                    refine, refine_coarse=refine_coarse, margin=margin),
                         args.steps, args.warmup, barrier)
    elapsed = max_over_ranks(elapsed)
    assert torch.isfinite(rgb).all()

    # dominant kernel: the fused NeRF MLP (two launches per view: coarse + fine), HIP events on the launch stream
    n_local = sh.hi - sh.lo
    pair_ms = [e[0].elapsed_time(e[1]) + e[2].elapsed_time(e[3]) for step in evs for e in step]
    fine_ms = [e[2].elapsed_time(e[3]) for step in evs for e in step]
    pts_per_pair = n_local * (N_COARSE + N_COARSE + N_FINE)
    pair_s = float(np.mean(pair_ms)) * 1e-3
    achieved = pts_per_pair * FLOP_PER_POINT / pair_s / 1e12
    fine_tf = n_local * (N_COARSE + N_FINE) * FLOP_PER_POINT / (float(np.mean(fine_ms)) * 1e-3) / 1e12

    # HBM traffic of the dominant kernel: rocprofv3 PMC passes cannot run inside this process, so the figure is the
    # committed digest of the same workload at N = 1 (scripts/gpu_pmc.sh -> scripts/pmc_digest.py): FETCH_SIZE x 2
    # (gfx950 correction) + WRITE_SIZE, per launch pair of a whole 640 000-ray view.
    variant = ops._capi.get_option("nerf_variant")
    variant = str(7 if variant is None else variant)
    # digest = average over the coarse and the fine dispatch; a launch pair = both
    fp32 = args.precision == 'fp32'
    traffic, traffic_source = (None, None)
    if n_local == H * W and variant == "7" and not fp32:
        if rank == 0 and world == 1 and not args.no_cpu_baseline:      # measured by a counter run of this very command
            traffic, traffic_source = live_traffic(PMC_RENDER_LEGS, 'nerf_mlp_bf16_v6', scale=2)
        if traffic is None:
            traffic, traffic_source = committed_traffic('nerf_mlp', scale=2)
    out = {
        "value": sh.rays_per_step_all_ranks * args.steps / elapsed,
        "ms_per_step": elapsed / args.steps * 1e3,
        "config": {
            "workload": "lego_3072-shaped NeRF coarse+fine MLP render, 800x800 rays per view, 64+128 samples "
                        "(BASELINE.json configs[1]); NeRFactor 512-light render (configs[2]) under \"nerfactor\"",
            "views_per_step": sh.n_views, "rays_per_view": H * W, "rays_per_step_per_gpu": sh.n_views * n_local,
            "partition": "each view's rays in contiguous ranges over the ranks",
            "n_samples_coarse": N_COARSE, "n_samples_fine": N_FINE, "weights": "glorot seed 0, opaque variant",
            "kernel_variant": variant,
            "coarse_precision": "auto: measured bf16 alpha error %s %s gate %.0e -> selective fp32-class coarse refinement %s" % (
                "n/a" if alpha_err is None else "%.2e" % alpha_err, ">" if refine_coarse else "<=", COARSE_REFINE_GATE,
                "ON" if refine_coarse else "off")},
        "roofline": {
            "bound": "mfma", "kernel": "%s (coarse + fine launches)" % (
                "nerf_mlp_x3_kernel, 3 MFMAs per product: achieved counts the algorithmic FLOPs once" if fp32 else {
                    "1": "nerf_mlp_bf16_kernel<1, 8>",
                    "7": "nerf_mlp_bf16_v6_kernel<0, 1>, LDS-DMA weight stream"}.get(variant, "variant %s" % variant)),
            "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS,
            "traffic": traffic, "traffic_unit": "GB per launch pair", "traffic_source": traffic_source,
            "algorithmic_hbm_gb": pts_per_pair * 20 / 1e9, "flop_per_launch_pair": pts_per_pair * FLOP_PER_POINT,
            "avg_launch_pair_ms": pair_s * 1e3, "fine_launch_tflops": fine_tf,
            "mlp_share_of_step": pair_s * sh.n_views / (elapsed / args.steps)},
    }
    if rank == 0 and not args.no_cpu_baseline:
        # parity of the timed frame: the CPU port's rays vs the same rays of view 0 of the last timed step
        base, idx, ref = nerf_cpu_reference(nets, *host_views[0], budget_s=args.cpu_budget, timed_run=(world == 1))
        sel = idx[(idx >= sh.lo) & (idx < sh.hi)]
        keep = np.isin(idx, sel)
        o, d = (torch.from_numpy(a[sel]).to(dev) for a in host_views[0])
        got = nerf_render_step(ops, [(o, d)], blobs, prec=args.precision, refine=refine, refine_coarse=refine_coarse, margin=margin).cpu().numpy()
        want = ref[1]['rgb'].numpy()[keep]
        # r04: NO ray is excused.  Rays decided by the sign of a near-zero logit at the dist = 1e10 last sample are a
        # discontinuity of the reference formula; the render evaluates that one sample fp32-class, so
        # they are held to the tolerance like every other ray.  The band count stays, for information.
        sig = np.minimum(ref[2]['sigma_last_coarse'].numpy()[keep], ref[2]['sigma_last_fine'].numpy()[keep])
        err = np.abs(got - want).max(1)
        out["parity"] = {
            "psnr_db": psnr_uint8_luma(got, want), "max_abs": float(err.max()),
            "max_abs_all_rays": float(err.max()), "rays_compared": int(len(sel)),
            "rays_excluded_from_max_abs": 0, "frac_rays_above_3e-2": float((err > 3e-2).mean()),
            "rays_above_3e-2": int((err > 3e-2).sum()),
            "rays_with_abs_sigma_last_below_0.06": int((sig <= 0.06).sum()),
            "last_sample": "bf16 (no refine)" if refine is None else "fp32-class density (ops.nerf_refine_last_sample)",
            "reference": "oracle/torch_ref.py (fp32) on the same rays; tolerance PSNR >= 40 dB, max-abs <= 3e-2"}
        if args.precision == 'bf16':
            out["parity_vs_reference"] = reference_fixture_parity(
                ops, dev, host_views[0], 'glorot', lambda oo, dd: nerf_render_step(
                    ops, [(oo, dd)], blobs, prec=args.precision, refine=refine, refine_coarse=refine_coarse, margin=margin))
        if base is not None:
            out["cpu_baseline"] = base
            out["gpu_over_cpu"] = out["value"] / base["value"]
        out["parity_fitted_weights"] = nerf_fitted_parity(args, ops, dev, host_views[0], refine is not None, views[0] if world == 1 else None)
    return out


REFERENCE_LARGE = os.path.join(ROOT, 'tests', 'golden', 'reference_large.npz')


def reference_fixture_parity(ops, dev, host_view, tag, render):
    """The timed view's 8192 fixture rays against the outputs of the REFERENCE'S OWN Python (tests/golden/reference_large.npz:
    nerfactor/models/nerf.py Model.call on the NumPy TensorFlow stand-in, make_reference_large_golden.py; weights `tag` =
    glorot | fitted).  `render(o, d)` -> rgb of the fine level.  None when the fixture is not there."""
    if not os.path.exists(REFERENCE_LARGE):
        return None
    gold = np.load(REFERENCE_LARGE)
    idx = gold['nerfbig_ray_index'].astype(np.int64)
    o, d = host_view[0][idx], host_view[1][idx]
    if not np.allclose([o.astype(np.float64).sum(), d.astype(np.float64).sum()], gold['nerfbig_ray_checksum'], rtol=1e-6):
        return {"error": "the bench view is not the view of the fixture"}
    got = render(torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)).cpu().numpy()
    want = gold['nerfbig_%s_fine_rgb' % tag]
    err = np.abs(got - want).max(1)
    return {"psnr_db": psnr_uint8_luma(got, want), "max_abs": float(err.max()), "rays_compared": int(len(idx)),
            "rays_above_3e-2": int((err > 3e-2).sum()), "median_abs": float(np.median(err)),
            "reference": "tests/golden/reference_large.npz: the reference's own nerf.py Model.call on these rays of the timed view"}


def nerf_fitted_parity(args, ops, dev, host_view, refine_last, full_view=None, n=2048):
    """The same render on the NeRF weights FITTED to a scene (tests/golden/nerf_trained_fp16.npz, the networks of the
    reference fixtures; empty space sits at a robustly negative density): the glorot "opaque variant" weights of the
    timed frame put ~9 % of the rays on the reference formula's own discontinuity (alpha_last = [sigma_last > 0],
    DESIGN.md §3.4), fitted weights < 2 % — max-abs is reported over ALL rays and outside the counted band."""
    from oracle import torch_ref
    from tests.golden import golden_inputs as gi
    nets = gi.trained_nerf_nets()
    from nerfactor_amd import synth
    blobs = [ops.pack_nerf_weights(*synth.nerf_layers(net), prec=args.precision).to(dev) for net in nets]
    refine = [ops.pack_nerf_geom_weights(*synth.nerf_layers(net), prec='fp32').to(dev) for net in nets] if refine_last else None
    idx = np.sort(np.random.default_rng(1).permutation(host_view[0].shape[0])[:n])
    o, d = host_view[0][idx], host_view[1][idx]
    torch.set_num_threads(host_cores())
    with torch.no_grad():
        ref = torch_ref.render_rays(torch.from_numpy(o), torch.from_numpy(d), *[torch_ref.to_torch_net(x) for x in nets])
    view = [(torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev))]
    refine_coarse, (alpha_err, sigma_err) = coarse_refine_decision(ops, full_view or view[0], nets, refine, args.precision)
    margin = ops.REFINE_MARGIN_FACTOR * sigma_err
    got = nerf_render_step(ops, view, blobs, prec=args.precision, refine=refine, refine_coarse=refine_coarse, margin=margin).cpu().numpy()
    want = ref[1]['rgb'].numpy()
    cost = None
    if full_view is not None and refine_coarse:       # what the refinement costs on THIS scene: the whole 800 x 800 view both ways
        def frame_ms(on, reps=3):
            nerf_render_step(ops, [full_view], blobs, prec=args.precision, refine=refine, refine_coarse=on, margin=margin)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                nerf_render_step(ops, [full_view], blobs, prec=args.precision, refine=refine, refine_coarse=on, margin=margin)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3
        o_f, d_f = full_view
        d_n = ops.l2_normalize3(d_f, 1e-12)
        z_f = ops.gen_z(2., 6., N_COARSE, o_f.shape[0], device=dev)
        raw_f = ops.nerf_mlp_fwd(o_f, d_n, z_f, blobs[0], args.precision)
        ops.nerf_refine_last_sample(o_f, d_n, z_f, raw_f, refine[0])
        _, cnt = ops.nerf_refine_coarse(o_f, d_n, z_f, raw_f, refine[0], sigma_margin=margin, want_count=True)
        t_off, t_on = frame_ms(False), frame_ms(True)
        cost = {"frame_ms_bf16_coarse": t_off, "frame_ms_with_refinement": t_on, "extra_frame_time": t_on / t_off - 1.,
                "coarse_samples_refined_frac": float(cnt.item()) / float(z_f.numel())}
    no_refine = None
    if refine_coarse:      # the same rays with coarse_precision = bf16 (the r05 default), for the record
        g0 = nerf_render_step(ops, view, blobs, prec=args.precision, refine=refine).cpu().numpy()
        e0 = np.abs(g0 - want).max(1)
        no_refine = {"max_abs_all_rays": float(e0.max()), "rays_above_3e-2": int((e0 > 3e-2).sum())}
    sig = np.minimum(ref[2]['sigma_last_coarse'].numpy(), ref[2]['sigma_last_fine'].numpy())
    err = np.abs(got - want).max(1)
    coarse32 = None
    if args.precision == 'bf16':      # the same rays with the plugin's `coarse_precision = fp32` (coarse pass fp32-class, fine pass bf16)
        b32 = [ops.pack_nerf_weights(*synth.nerf_layers(nets[0]), prec='fp32').to(dev), blobs[1]]
        g32 = nerf_render_step(ops, view, b32, prec='bf16', refine=refine, coarse_prec='fp32').cpu().numpy()
        e32 = np.abs(g32 - want).max(1)
        coarse32 = {"psnr_db": psnr_uint8_luma(g32, want), "max_abs_all_rays": float(e32.max()),
                    "rays_above_3e-2": int((e32 > 3e-2).sum()), "what": "ini key coarse_precision = fp32: +67 % frame time (r04 call B)"}
    vs_ref = reference_fixture_parity(ops, dev, host_view, 'fitted', lambda oo, dd: nerf_render_step(
        ops, [(oo, dd)], blobs, prec=args.precision, refine=refine, refine_coarse=refine_coarse, margin=margin)) if args.precision == 'bf16' else None
    return {"psnr_db": psnr_uint8_luma(got, want), "max_abs": float(err.max()), "coarse_precision_fp32": coarse32,
            "vs_reference": vs_ref,
            "coarse_precision": "auto: measured bf16 alpha error %s -> selective refinement %s" % (
                "n/a" if alpha_err is None else "%.2e" % alpha_err, "ON" if refine_coarse else "off"),
            "coarse_refinement_cost": cost, "coarse_precision_bf16": no_refine,
            "max_abs_all_rays": float(err.max()), "rays_compared": int(n),
            "rays_excluded_from_max_abs": 0, "rays_with_abs_sigma_last_below_0.06": int((sig <= 0.06).sum()),
            "frac_rays_above_3e-2": float((err > 3e-2).mean()), "rays_above_3e-2": int((err > 3e-2).sum()),
            "weights": "tests/golden/nerf_trained_fp16.npz (fitted to the unit-sphere scene)",
            "reference": "oracle/torch_ref.py (fp32) on the same rays of the timed view"}


# ------------------------------------------------------------------------------------------------ NeRFactor legs
class KernelTimer:
    """HIP-event timing of selected `ops` entry points as the model calls them (events on torch's current stream,
    which is the stream ops launches on)."""

    def __init__(self, ops, names):
        self.ops, self.names, self.orig, self.events, self.on = ops, names, {}, {n: [] for n in names}, False

    def __enter__(self):
        for n in self.names:
            self.orig[n] = getattr(self.ops, n)
            setattr(self.ops, n, self._wrap(n, self.orig[n]))
        return self

    def __exit__(self, *exc):
        for n, f in self.orig.items():
            setattr(self.ops, n, f)

    def _wrap(self, name, fn):
        def call(*a, **kw):
            if not self.on:
                return fn(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **kw)
            e1.record()
            self.events[name].append((e0, e1))
            return r
        return call

    def mean_ms(self, name):
        ev = self.events[name]
        return float(np.mean([a.elapsed_time(b) for a, b in ev])) if ev else None

    def all_ms(self, name):
        return [a.elapsed_time(b) for a, b in self.events[name]]


def nerfactor_nets_of(model, variant):
    """The model's weights as the oracle's net dict (CPU tensors)."""
    def pairs(net, name):
        ks, bs = net[name].kernels_and_biases()
        return [(k.detach().cpu(), b.detach().cpu()) for k, b in zip(ks, bs)]
    net = {k: pairs(model.net, k) for k in ('normal_mlp', 'normal_out', 'lvis_mlp', 'lvis_out', 'albedo_mlp',
                                            'albedo_out', 'brdf_z_mlp', 'brdf_z_out')}
    brdf_net = None
    if variant == 'learned':
        brdf_net = {k: pairs(model.brdf_model.net, k) for k in ('brdf_mlp', 'brdf_out')}
    return net, brdf_net


def nerfactor_cpu_reference(model, variant, batch_host, lights, budget_s, timed_run):
    from oracle import torch_ref
    net, brdf_net = nerfactor_nets_of(model, variant)
    cfg = model.config
    kw = dict(variant=variant, brdf_net=brdf_net, f0=cfg.getfloat('DEFAULT', 'fresnel_f0', fallback=0.04),
              brdf_scale=cfg.getfloat('DEFAULT', 'learned_brdf_scale', fallback=1.),
              albedo_slope=cfg.getfloat('DEFAULT', 'albedo_slope'), albedo_bias=cfg.getfloat('DEFAULT', 'albedo_bias'),
              to_srgb=cfg.getboolean('DEFAULT', 'linear2srgb'))
    lxyz, lareas = model.lxyz.cpu(), model.lareas.cpu()
    lights = lights.cpu()
    rayo, alpha, xyz = (torch.from_numpy(batch_host[i]) for i in (2, 5, 6))
    torch.set_num_threads(host_cores())

    def run(n):
        t0 = time.perf_counter()
        res = torch_ref.nerfactor_render((rayo[:n], alpha[:n], xyz[:n]), net, lxyz, lareas, lights, **kw)
        return time.perf_counter() - t0, res

    with torch.no_grad():
        if not timed_run:
            return None, 512, run(512)[1]
        run(16)
        t_probe = run(128)[0]
        n = int(min(16384, max(128, 128 * budget_s / max(t_probe, 1e-3))))
        n = (n // 128) * 128
        dt, res = run(n)
    base = {"value": n / dt, "unit": "points/s", "cores": host_cores(), "kind": "port",
            "sample": "first %d surface points of the same batch (%d foreground), 512 lights, 1+%d lights, torch-CPU "
                      "fp32 (oracle/torch_ref.py:nerfactor_render, mlp_chunk=65536), %.1f s" % (
                          n, int(alpha[:n].sum()), lights.shape[0] - 1, dt)}
    return base, n, res


def nerfactor_leg(name, args, ops, dev, rank, world, barrier, max_over_ranks):
    from nerfactor_amd import synth
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.models import get_model_class
    variant = 'microfacet' if name == 'nerfactor_microfacet' else 'learned'
    torch.manual_seed(5)
    cfg = make_config(name, shape_mode='finetune', shape_model_ckpt='none', brdf_model_ckpt='none',
                      test_envmap_dir='', xyz_jitter_std='0', precision=args.precision)
    model = get_model_class(name)(cfg).to(dev)
    for i, p in enumerate(synth.probes(N_PROBES, seed=20)):
        model.add_probe('p%d' % i, p)
    n = H * W
    sh = Shards(n, rank, world, args.scaling)
    host_batches, batches, n_fg_local = [], [], 0
    for v in range(sh.n_views):
        hb = synth.surface_batch(n, seed=1 + 10 * v, n_lights=N_LIGHTS)
        host_batches.append(hb)
        batches.append(tuple(None if a is None else torch.from_numpy(a[sh.lo:sh.hi]).to(dev) for a in hb))
        n_fg_local += int(hb[5][sh.lo:sh.hi].sum())
    names = ['lvis_fwd'] + (['brdf_spec_fwd'] if variant == 'learned' else [])
    with KernelTimer(ops, names) as kt:
        def step(k):
            kt.on = k is not None
            out = None
            for b in batches:
                out = model(b, mode='test', relight_probes=True)[0]
            return out
        elapsed, pred = timed(step, args.steps, args.warmup, barrier)
        kt.on = False
    elapsed = max_over_ranks(elapsed)
    assert torch.isfinite(pred['rgb_probes']).all()
    fg_per_call = n_fg_local / sh.n_views
    lvis_s = kt.mean_ms('lvis_fwd') * 1e-3
    lvis_tf = fg_per_call * N_LIGHTS * 2 * LVIS_MAC / lvis_s / 1e12
    lvis_variant = ops._capi.get_option("lvis_variant")
    lvis_variant = str(8 if lvis_variant is None else lvis_variant)
    # algorithmic bytes per foreground point: 512 visibilities written, the 1-KiB pre-activation row written by
    # lvis_pre and read once by the main kernel, the position read
    lv_traffic, lv_source = None, None
    if world == 1 and args.precision == 'bf16':
        a, src, b = None, None, None
        if rank == 0 and not args.no_cpu_baseline:
            a, src = live_traffic(PMC_RENDER_LEGS, 'resident128_kernel')
            b, _ = live_traffic(PMC_RENDER_LEGS, 'lvis_pre_kernel')
        if a is None or b is None:
            a, src = committed_traffic('resident128_kernel')
            b, _ = committed_traffic('lvis_pre_kernel')
        if a is not None and b is not None:
            lv_traffic, lv_source = a + b, src
    out = {
        "workload": "%s full render (BASELINE.json configs[2]): 800x800 surface points per view (60 %% foreground), "
                    "512 lights, trained light + %d probes, Model.call(mode='test', relight_probes=True)" % (
                        name, N_PROBES),
        "points_per_s": sh.rays_per_step_all_ranks * args.steps / elapsed,
        "ms_per_step": elapsed / args.steps * 1e3, "ms_per_view_per_gpu": elapsed / args.steps * 1e3 * world / sh.n_views,
        "foreground_points_per_view": int(host_batches[0][5].sum()),
        "flop_per_foreground_point_algorithmic": 2 * (N_LIGHTS * LVIS_MAC + HEAD_MAC + 65408),
        "roofline": {
            "bound": "mfma", "kernel": ("mlp128_x3_kernel<1> (light visibility, %d x 512 rows; 3 MFMAs per product, "
                                        "plain 90-dim input: achieved counts the bf16 path's algorithmic FLOPs)"
                                        if args.precision == 'fp32' else
                                        "lvis_pre_kernel + %s (light visibility, %%d x 512 rows)" % {
                "8": "resident128_kernel<2, 0, 8>", "4": "resident128_kernel<4, 0, 4>"}.get(
                    lvis_variant, "variant %s" % lvis_variant))
                                       % int(fg_per_call),
            "achieved": lvis_tf, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": lvis_tf / PEAK_BF16_TFLOPS,
            "avg_launch_ms": lvis_s * 1e3, "flop_per_launch": fg_per_call * N_LIGHTS * 2 * LVIS_MAC,
            "executed_tflops": fg_per_call * N_LIGHTS * 2 * 61440 / lvis_s / 1e12,
            "share_of_step": lvis_s * sh.n_views / (elapsed / args.steps),
            "traffic": lv_traffic, "traffic_unit": "GB per launch (lvis_pre + light-visibility kernel)",
            "traffic_source": lv_source,
            "algorithmic_hbm_gb": fg_per_call * (N_LIGHTS * 4 + 2 * 1024 + 12) / 1e9},
    }
    if variant == 'learned':
        spec_s = kt.mean_ms('brdf_spec_fwd') * 1e-3
        out["brdf_spec"] = {"avg_launch_ms": spec_s * 1e3,
                            "tflops_all_rows": fg_per_call * N_LIGHTS * 2 * BRDF_MAC / spec_s / 1e12}
    if rank == 0 and not args.no_cpu_baseline:
        lights = torch.stack([model.light.detach().cpu().reshape(-1, 3)] +
                             [p.cpu().reshape(-1, 3) for p in model.novel_probes.values()])
        base, n_s, ref = nerfactor_cpu_reference(model, variant, host_batches[0], lights, args.cpu_budget / 4,
                                                 timed_run=(world == 1))
        hi = min(n_s, sh.hi)
        if hi > 0 and sh.lo == 0:
            b = tuple(None if a is None else torch.from_numpy(a[:hi]).to(dev) for a in host_batches[0])
            p = model(b, mode='test', relight_probes=True)[0]
            got = torch.cat((p['rgb'][:, None], p['rgb_probes']), 1).cpu().numpy()
            want = ref['rgb'].numpy()[:hi]
            # No point is excused: max_abs is over every foreground point and all nine lights.  (Round 2 excluded view
            # directions grazing the predicted normal, |n.v| < 0.05, where spec / (4 |l.n| |v.n|) of microfacet.py:57
            # amplified the bf16 error of the normal head; that head now runs with fp32-class operands.)  The grazing
            # subset is still reported, as information.
            nrm = ref['normal'].numpy()[:hi]
            vdir = host_batches[0][2][:hi] - host_batches[0][6][:hi]
            vdir /= np.maximum(np.linalg.norm(vdir, axis=1, keepdims=True), 1e-6)
            fg = host_batches[0][5][:hi, 0] > 0
            grazing = fg & (np.abs((nrm * vdir).sum(1)) <= 0.05)
            err = np.abs(got - want).max((1, 2))
            out["parity"] = {
                "psnr_db": psnr_uint8_luma(got.reshape(-1, 3), want.reshape(-1, 3)),
                "max_abs": float(err[fg].max()), "max_abs_all_points": float(err.max()),
                "max_abs_trained_light": float(np.abs(got[:, 0] - want[:, 0])[fg].max()),
                "frac_points_above_3e-2": float((err[fg] > 3e-2).mean()),
                "points_above_3e-2": int((err[fg] > 3e-2).sum()),
                "grazing_points_excluded_from_max_abs": 0,
                "grazing_points_abs_n_dot_v_below_0.05": int(grazing.sum()),
                "max_abs_outside_grazing": float(err[fg & ~grazing].max()),
                "normal_head_precision": model.normal_precision,
                "max_abs_lvis": float(np.abs(p['lvis'].cpu().numpy() - ref['lvis'].numpy()[:hi]).max()),
                "max_abs_albedo": float(np.abs(p['albedo'].cpu().numpy() - ref['albedo'].numpy()[:hi]).max()),
                "max_abs_normal": float(np.abs(p['normal'].cpu().numpy() - nrm).max()),
                "points_compared": int(hi), "foreground_points_compared": int(fg.sum()),
                "lights_compared": int(got.shape[1]),
                "reference": "oracle/torch_ref.py:nerfactor_render (fp32) on the same points; probes are HDR "
                             "(exp N(0,1)) and the render clips to [0, 1]"}
            if variant == 'learned':
                fl = float((ref.get('front_lit_frac') or 0.))
                out["brdf_spec"]["front_lit_fraction"] = fl
                out["brdf_spec"]["tflops_algorithmic_front_lit_rows"] = out["brdf_spec"]["tflops_all_rows"] * fl
        if base is not None:
            out["cpu_baseline"] = base
            out["gpu_over_cpu"] = out["points_per_s"] / base["value"]
    return out


# ------------------------------------------------------------------------------------------------ train leg (configs[3])
TRAIN_RAYS = 1024          # n_rays_per_step of config/*.ini; per GPU (weak scaling)
TRAIN_STEPS_PER_BENCH_STEP = 20
TRAIN_STEPS_MAX = 100      # one synthetic batch with noise targets, lr 5e-3: the fit turns unstable after a few hundred
                           # steps (check_numerics raised at --steps 20 = 400 steps in round 3), so the leg is capped


def train_leg(name, args, ops, dev, rank, world, barrier, max_over_ranks):
    """K x 20 eager optim.train_step calls of one model at 1024 rays per GPU (SURVEY.md §8d "C4", trainvali.py:273-295):
    forward (clean + jittered points), fused loss, backward through the libnfx kernels, one all-reduce of the flat
    [gradients | loss] bucket over all ranks, fused AMSGrad.  The batch is resident in HBM and the same every step
    (the loader is measured by scripts/bench_loader.py)."""
    from nerfactor_amd import dist as nfx_dist, optim
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.datasets.nerf_shape import mark_all_foreground
    from nerfactor_amd.nerfactor.models import get_model_class
    torch.manual_seed(5)                                   # identical initial weights on every rank
    extra = dict(shape_mode='finetune', shape_model_ckpt='none', test_envmap_dir='') if 'nerfactor' in name else {}
    cfg = make_config(name, precision=args.precision, **extra)
    model = get_model_class(name)(cfg).to(dev)
    opt = optim.make_optimizer(model, cfg)
    nfx_dist.broadcast_model(model, opt)
    rng = np.random.default_rng(100 + rank)
    n = TRAIN_RAYS
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    xyz = t(rng.uniform(-1, 1, size=(n, 3)))
    nrm = torch.nn.functional.normalize(t(rng.normal(size=(n, 3))), dim=1)
    cam = t(np.broadcast_to([2.2, -2.4, 1.7], (n, 3)))
    if name == 'nerf':       # rays from the camera towards the unit cube
        batch = (None, None, cam, xyz - cam, t(rng.uniform(size=(n, 3))))
    else:
        batch = (None, None, cam, t(np.zeros((n, 3))), t(rng.uniform(size=(n, 3))),
                 mark_all_foreground(torch.ones(n, 1, device=dev)), xyz, nrm, t(rng.uniform(size=(n, 512))))
    global_bs = n * world
    steps = min(args.steps * TRAIN_STEPS_PER_BENCH_STEP, TRAIN_STEPS_MAX)
    bwd_names = ['nerf_mlp_bwd'] if name == 'nerf' else ['mlp128_bwd']
    losses, errors = [], []
    with KernelTimer(ops, bwd_names) as kt:
        def step(k):
            kt.on = k is not None
            try:
                loss, _ = optim.train_step(model, batch, opt, global_bs)
            except FloatingPointError as e:
                # check_numerics verdicts are raised by train_step's FIRST statement (flush_numerics), before any
                # collective of the step, and only on the rank that saw the non-finite tensor.  Leaving the loop here
                # would strand the other ranks in this step's all_reduce: record it, run the step, and let
                # all ranks agree on the failure after the timed region.
                errors.append(str(e))
                loss, _ = optim.train_step(model, batch, opt, global_bs)
            if k is not None:
                losses.append(loss)
            return loss
        elapsed, _ = timed(step, steps, max(3, args.warmup), barrier)
        kt.on = False
    elapsed = max_over_ranks(elapsed)
    try:
        model.flush_numerics(block=True)
    except FloatingPointError as e:
        errors.append(str(e))
    if max_over_ranks(1. if errors else 0.) > 0:      # collective: every rank leaves the leg together
        return {"error": "check_numerics raised during the timed steps on at least one rank%s" % (
            ": " + errors[0] if errors else " (not this one)")}
    losses = torch.stack(losses).cpu().numpy()
    assert np.isfinite(losses).all(), "non-finite training loss in the timed steps"
    dt_eager = elapsed / steps
    # One process, foreground-tagged batch: the step as ONE hipGraph replay (optim.GraphedTrainStep = ini key
    # `hip_graph = true`; bit-identical to the eager step, tests/test_gpu_train.py).  The eager NeRFactor step issues
    # ~170 launches and sits at the host's issue time (2.2-2.8 ms depending on the box's CPU), whatever its kernels
    # take: the replay is what shows the GPU time.  N > 1 ranks keep the eager step (the all-reduce is not captured).
    graph_dt, graph_note = None, None
    if (world == 1 or os.environ.get('NFX_GRAPH_COLLECTIVE') == '1') and not args.no_hip_graph:      # (NeRF too since r05: torch's generator hands a replay the stratified draws of an eager step)
        gstep = optim.GraphedTrainStep(model, opt, global_bs, capture_collective=(True if world == 1 else None))
        glosses = []

        def gs(k):
            loss, _ = gstep(batch)
            if k is not None:
                glosses.append(loss)
            return loss
        try:
            g_steps = min(steps, 60)       # (the leg is capped at TRAIN_STEPS_MAX optimizer steps on one noise batch)
            elapsed_g, _ = timed(gs, g_steps, 6, barrier)
            model.flush_numerics(block=True)
            if np.isfinite(torch.stack(glosses).cpu().numpy()).all():
                graph_dt = elapsed_g / g_steps
            else:
                graph_note = "non-finite loss in the replayed steps"
        except FloatingPointError as e:
            graph_note = "check_numerics raised in the replayed steps: %s" % e
    if graph_dt is not None and graph_dt > dt_eager:      # (a GPU-bound step — NeRF: ~60 launches — gains nothing from the replay)
        graph_dt, graph_note = None, "the hipGraph replay was not faster: %.3f ms" % (graph_dt * 1e3)
    dt = graph_dt if graph_dt is not None else dt_eager
    grad_frac = None
    if name == 'nerf':
        # per ray 64 coarse + 192 fine points; forward + re-computed forward + dgrad + wgrad = 4 x the forward MACs — the last
        # three only for the points with a gradient (nfx_nerf_mlp_bwd skips a point whose d_rgbs is four zeros: a sample the
        # composite gave no weight; the reference's gradient of it is zero too).  Their share: one untimed step, counted by
        # the library's own device-side list
        ops.NERF_BWD_STATS = []
        try:
            optim.train_step(model, batch, opt, global_bs)
            torch.cuda.synchronize()
            stats = [(int(c.item()), m) for c, m in ops.NERF_BWD_STATS]
        finally:
            ops.NERF_BWD_STATS = None
        listed = ops._capi.get_option('nerf_bwd_rows') != 0
        grad_frac = sum(c for c, _ in stats) / max(1, sum(m for _, m in stats)) if listed and stats else 1.
        pts = n * (N_COARSE + N_COARSE + N_FINE)
        flops = pts * FLOP_PER_POINT * (1 + 3 * grad_frac)
        what = ("64+128 samples per ray, perturb on; FLOPs = forward over every point + 3 x forward (re-computed forward, dgrad, "
                "wgrad) over the %.1f %% of the points with a gradient" % (100 * grad_frac))
        dom, per_step = 'nerf_mlp_bwd', 2
    else:
        rows = n * N_LIGHTS * 2                                # clean + jittered visibility rows
        flops = 3 * 2 * (rows * LVIS_MAC + 2 * 3 * n * 65664)   # 2 (jitter) x 3 (fwd + dgrad + wgrad) x forward
        what = "512 lights, xyz jitter on; FLOPs = 2 (clean + jittered) x 3 (forward, dgrad, wgrad) x the four trainable MLPs' forward"
        dom, per_step = 'mlp128_bwd', 4
    fitted = None
    if name == 'nerf' and world == 1 and args.precision == 'bf16' and not args.no_hip_graph:
        fitted = nerf_fitted_train_step(ops, dev, n, optim, get_model_class, make_config)
    calls = np.asarray(kt.all_ms(dom))
    if calls.size != steps * per_step:
        per_step = max(1, calls.size // steps)
    calls = calls[:steps * per_step].reshape(steps, per_step)  # per step: the calls in launch order
    big = calls.max(1)                                          # the largest call of a step (light visibility / fine net)
    tf = flops / dt / 1e12
    # HBM traffic of the step's largest backward call from the committed PMC digest of the training legs (separate
    # rocprofv3 --pmc passes of `bench.py --legs train`): per launch, averaged over the kernel's dispatches
    traffic, traffic_source, traffic_kernels = None, None, (
        ['nerf_bwd_ring_kernel', 'nfx::wgrad_lds_kernel', 'nfx::wgrad_reduce_kernel'] if name == 'nerf' else
        ['mlp128_bwd_fused_kernel<1, 0>', 'mlp128_bwd_fused_kernel<1, 1>', 'mlp128_wgrad_reduce_kernel<6>'])
    if args.precision == 'bf16':
        parts = [committed_traffic(k) for k in traffic_kernels]
        if all(p[0] is not None for p in parts):
            traffic, traffic_source = sum(p[0] for p in parts), parts[0][1]
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # (one process only: these are training steps of their own — with a process group their all-reduce would wait for ranks
        #  that are not here; found by the full-leg two-rank rehearsal of round 6)
        # the reference's own ten steps (tests/golden/reference_grads.npz) through the same train path: step-1
        # gradients, loss trajectory, parameters after the steps (tests/reference_steps.py; CPU oracle outside the timing)
        from tests import reference_steps
        tag, rmodel, rlosses, rgrad1 = reference_steps.run(name, dev)
        parity = reference_steps.summary(tag, reference_steps.metrics(tag, rmodel, rlosses, rgrad1))
    fp32_block = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.precision == 'bf16':
        # the same step at the reference's arithmetic class (precision = fp32 -> grad_precision = fp32: fp32 activations and
        # gradients, every MFMA operand a bf16 hi / lo pair, forward and backward: csrc/mlp_generic.hip, round 5) — its
        # time, its gradients held to the reference's fp32 gradients directly, and the time of the same step on the native
        # fp32 matrix instruction (fp32_matrix = native, round 4's path) beside it
        def time_fp32(fp32_matrix, graph, **more):
            torch.manual_seed(5)
            m32 = get_model_class(name)(make_config(name, precision='fp32', fp32_matrix=fp32_matrix, **extra, **more)).to(dev)
            o32 = optim.make_optimizer(m32, m32.config)
            step32 = optim.GraphedTrainStep(m32, o32, global_bs) if graph else (lambda b: optim.train_step(m32, b, o32, global_bs))
            for _ in range(6 if graph else 3):
                step32(batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                step32(batch)
            torch.cuda.synchronize()
            dt32 = (time.perf_counter() - t0) / 10
            m32.flush_numerics(block=True)
            return dt32 * 1e3
        default_mode = get_model_class(name).DEFAULT_FP32_MATRIX      # pairs for the surface models, native for NeRF
        graph32 = not args.no_hip_graph and default_mode == 'pairs'
        ms32 = {}
        if graph32:
            try:
                ms32["pairs_hip_graph"] = time_fp32('pairs', True)
            except Exception as e:      # (a capture failure must not take the whole line down)
                ms32["pairs_hip_graph_error"] = str(e)[:200]
        if not any(isinstance(v, float) for v in ms32.values()):
            ms32[default_mode + "_eager"] = time_fp32(default_mode, False)
        if args.fp32_modes:             # every matrix mode beside the default one (--fp32-modes: ~8 s per model)
            for mode in ('pairs', 'native'):
                ms32.setdefault(mode + "_eager", time_fp32(mode, False))
            # (tests/test_gpu_convergence.py: the floor of bf16 training comes from the bf16 FORWARD; fp32-class forward kernels with
            #  the bf16-operand backward kernels — precision = fp32, grad_precision = bf16 — recover most of it)
            ms32["fp32_forward_bf16_grads_eager"] = time_fp32('pairs', False, grad_precision='bf16')
        tag, rmodel, rlosses, rgrad1 = reference_steps.run(name, dev, 'fp32')
        p32 = reference_steps.metrics_fp32(tag, rmodel, rlosses, rgrad1)
        p32.pop('grads')
        fp32_block = {"what": "the same step with precision = fp32: every network forward and backward through nfx_mlp_generic_fwd / "
                              "_bwd with fp32 activations, gradients and workspace; fp32_matrix = pairs: bf16 hi / lo operand "
                              "pairs, 3 x v_mfma_f32_32x32x16_bf16 per product; native: v_mfma_f32_32x32x2_f32" + (
                                  "; the frozen learned BRDF on explicit fp32 rows (nfx_brdf_rows_geom_fwd / _bwd) instead of "
                                  "inside the bf16 shading kernels" if name == 'nerfactor' else ""),
                      "fp32_matrix_default": default_mode,
                      "ms_per_step": min(v for k, v in ms32.items() if k.startswith(default_mode) and isinstance(v, float)),
                      "ms_per_step_by_mode": ms32, "parity": p32}
    return {
        "workload": "%s optim.train_step, %d rays per GPU and step (weak), %s" % (name, n, what),
        "parity": parity, "fp32": fp32_block,
        "steps": steps, "ms_per_step": dt * 1e3, "rays_per_s": n * world / dt,
        "step": ("one hipGraph replay per step (optim.GraphedTrainStep, ini hip_graph = true)" if graph_dt is not None
                 else "eager optim.train_step" + (" (%s)" % graph_note if graph_note else "")),
        "ms_per_step_eager": dt_eager * 1e3,
        "first_loss": float(losses[0]), "final_loss": float(losses[-1]),
        "points_with_gradient_frac": grad_frac, "fitted_scene": fitted,
        "collective": ("%s all_reduce of one flat fp32 bucket (%d floats) per step over %d ranks" % (
            torch.distributed.get_backend(), opt.bucket.flat.numel(), world)) if (world > 1 or nfx_dist.run_collectives_on_one_rank())
        else "none (one rank, no process group)",
        "roofline": {"bound": "mfma", "kernel": "whole step; largest backward call = ops.%s (fused backward + batched "
                                                "weight-gradient launches)" % dom,
                     "achieved": tf, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_BF16_TFLOPS,
                     "flop_per_step_per_gpu": flops, "traffic": traffic,
                     "traffic_unit": "GB per largest backward call (%s)" % " + ".join(traffic_kernels),
                     "traffic_source": traffic_source,
                     "largest_backward_call_ms": float(big.mean()),
                     "backward_calls_ms_per_step": float(calls.sum(1).mean())},
    }


def nerf_fitted_train_step(ops, dev, n, optim, get_model_class, make_config, steps=40):
    """The NeRF training step on networks FITTED to a scene (tests/golden/nerf_trained_fp16.npz) — what a training run spends most
    of its steps on: empty space has a negative raw density there, so most points carry no gradient and nfx_nerf_mlp_bwd skips
    them.  `n` rays of the view the render legs time, targets = the networks' own render (the densities stay
    where they are), one hipGraph replay per step; one process only."""
    from nerfactor_amd import synth
    from tests.golden import golden_inputs as gi
    torch.manual_seed(0)
    model = get_model_class('nerf')(make_config('nerf'))
    with torch.no_grad():
        for pref, net in zip(('coarse_', 'fine_'), gi.trained_nerf_nets()):
            for part in ('enc', 'sigma_out', 'bottleneck', 'rgb_out'):
                for layer, (k, b) in zip(model.net[pref + part].layers, net[part]):
                    layer.kernel.copy_(torch.from_numpy(np.asarray(k, np.float32)))
                    layer.bias.copy_(torch.from_numpy(np.asarray(b, np.float32)))
    model = model.to(dev)
    opt = optim.make_optimizer(model, model.config)
    rayo, rayd = synth.camera_rays(800, 800, cam_loc=(4 * 0.8, -0.1, 4 * 0.6))
    idx = np.random.default_rng(7).choice(rayo.shape[0], n, replace=False)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    o, d = t(rayo[idx]), t(rayd[idx])
    with torch.no_grad():
        rgb = model((None, None, o, d, torch.zeros_like(o)), mode='test')[0]['fine'].clamp(0, 1)
    batch = (None, None, o, d, rgb)

    def count():
        ops.NERF_BWD_STATS = []
        try:
            optim.train_step(model, batch, opt, n)
            torch.cuda.synchronize()
            return [(int(c.item()), m) for c, m in ops.NERF_BWD_STATS]
        finally:
            ops.NERF_BWD_STATS = None
    listed = ops._capi.get_option('nerf_bwd_rows') != 0
    before = count()
    step = optim.GraphedTrainStep(model, opt, n, capture_collective=True)      # (one process; --force-group: the all-reduce is captured too)
    for _ in range(6):
        step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _ = step(batch)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    after = count()
    frac = lambda st: (sum(c for c, _ in st) / max(1, sum(m for _, m in st))) if listed and st else 1.
    model.flush_numerics(block=True)
    return {"what": "the same step on the networks fitted to a scene, %d rays of the rendered view, targets = their own render, "
                    "one hipGraph replay per step" % n,
            "ms_per_step": ms, "steps": steps, "loss": float(loss),
            "points_with_gradient_frac": frac(before), "points_with_gradient_frac_after_the_steps": frac(after)}


# ------------------------------------------------------------------------------------------------ geometry leg
GEO_MAC = 63 * 256 + 6 * 256 * 256 + 319 * 256 + 256      # encoder (8 x 256, skip behind layer 4) + sigma_out: MAC per density sample
GEO_LVIS_POINTS = 8192                                    # surface points of the shadow-ray stage's bounded sample


def geometry_leg(args, ops, dev, rank, world, barrier, max_over_ranks):
    """geometry_from_nerf.py:93-246 on the driver line: the timed 800 x 800 view of the bench through
    `compute_depth_and_normal` (64 + 64 coarse samples of the coarse density, 64 + 64 + 128 + 64 = 320 samples of the fine
    density WITH its gradient: expected depth and normal per ray) on the NeRF weights fitted to a scene, then
    `compute_light_visibility` (the same 128 + 320 density march along the shadow ray to every front-lit one of 512 lights)
    on a bounded sample of that view's surface points — the reference's heaviest offline stage, N x L x 448 density
    evaluations per view.  Rays of the view are split over the ranks.  Roofline: MFMA, from the HIP-event time of the
    density kernels; FLOPs = 2 x GEO_MAC per density sample, twice that where the gradient is taken (forward + reverse sweep)."""
    from nerfactor_amd import synth
    from nerfactor_amd.nerfactor import geometry_from_nerf as G
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.models import get_model_class
    from tests.golden import golden_inputs as gi
    cfg = make_config('nerf', precision=args.precision)
    model = get_model_class('nerf')(cfg)
    nets = gi.trained_nerf_nets()
    with torch.no_grad():
        for pref, net in zip(('coarse_', 'fine_'), nets):
            for part in ('enc', 'sigma_out', 'bottleneck', 'rgb_out'):
                for layer, (k, b) in zip(model.net[pref + part].layers, net[part]):
                    layer.kernel.copy_(torch.from_numpy(np.asarray(k, np.float32)))
                    layer.bias.copy_(torch.from_numpy(np.asarray(b, np.float32)))
    model = model.to(dev)
    rayo_h, rayd_h = synth.camera_rays(H, W)
    rayd_h = rayd_h / np.maximum(np.linalg.norm(rayd_h, axis=1, keepdims=True), 1e-12)
    sh = Shards(H * W, rank, world, 'strong')
    rayo, rayd = torch.from_numpy(rayo_h[sh.lo:sh.hi]).to(dev), torch.from_numpy(rayd_h[sh.lo:sh.hi].astype(np.float32)).to(dev)
    n_c, n_f, _ = G._sample_counts(cfg)          # 128, 192: 320 fine-network samples per ray
    steps = max(1, min(args.steps, 3))
    state = {}
    with torch.no_grad(), KernelTimer(ops, ['nerf_sigma_fwd', 'nerf_sigma_grad']) as kt:
        def march(k):
            kt.on = k is not None
            state['out'] = G.compute_depth_and_normal(model, rayo, rayd, cfg)
            return state['out'][0]
        elapsed, _ = timed(march, steps, 1, barrier)
        kt.on = False
        elapsed = max_over_ranks(elapsed)
        occu, depth, normal = state['out']
        fwd_ms, grad_ms = kt.all_ms('nerf_sigma_fwd'), kt.all_ms('nerf_sigma_grad')
    kernel_ms = (sum(fwd_ms) + sum(grad_ms)) / steps
    n_local = sh.hi - sh.lo
    # executed FLOPs: the coarse density, the fine density of every sample, and forward + reverse sweep of the samples with a
    # positive density only (ops.nerf_sigma_grad -> nfx_nerf_sigma_grad_rows: every other sample's gradient is zero); their
    # count comes from the library's device-side list on one untimed march
    listed_frac = 1.
    if ops._capi.get_option("sigma_grad_rows") != 0:
        ops.SIGMA_GRAD_STATS = []
        try:
            with torch.no_grad():
                G.compute_depth_and_normal(model, rayo, rayd, cfg)
            torch.cuda.synchronize()
            st = [(int(c.item()), m) for c, m in ops.SIGMA_GRAD_STATS]
        finally:
            ops.SIGMA_GRAD_STATS = None
        listed_frac = sum(c for c, _ in st) / max(1, sum(m for _, m in st)) if st else 1.
        flop_dn = n_local * (n_c + (n_c + n_f) * (1 + 2 * listed_frac)) * 2 * GEO_MAC
    else:
        flop_dn = n_local * (n_c + 2 * (n_c + n_f)) * 2 * GEO_MAC
    # ---- shadow rays: a bounded sample of this rank's surface points (foreground: occupancy > 0.5) x 512 lights
    fg = torch.nonzero(occu > 0.5)[:, 0]
    n_pts = min(GEO_LVIS_POINTS // world, int(fg.numel()))
    pick = fg[torch.linspace(0, fg.numel() - 1, n_pts, device=dev).long()] if n_pts else fg[:0]
    surf = (rayo[pick] + rayd[pick] * depth[pick, None]).contiguous()
    nrm = torch.nn.functional.normalize(normal[pick], dim=1).contiguous()
    lv = {}
    with torch.no_grad(), KernelTimer(ops, ['nerf_sigma_fwd']) as kt:
        def shadow(k):
            kt.on = k is not None
            lv['lvis'] = G.compute_light_visibility(model, surf, nrm, cfg)
            return lv['lvis']
        elapsed_l, _ = timed(shadow, 1, 1, barrier)
        kt.on = False
        elapsed_l = max_over_ranks(elapsed_l)
        lvis_kernel_ms = sum(kt.all_ms('nerf_sigma_fwd'))
    lxyz, _ = G.gen_light_xyz(16, 32)
    lx = torch.as_tensor(lxyz.reshape(-1, 3).astype(np.float32), device=dev)
    s2l = torch.nn.functional.normalize(lx[None] - surf[:, None], dim=2, eps=1e-12)
    pairs = int(((s2l * nrm[:, None]).sum(-1) > 0).sum())
    flop_lv = pairs * (n_c + n_c + n_f) * 2 * GEO_MAC
    finite = bool(torch.isfinite(normal).all() and torch.isfinite(lv['lvis']).all())
    out = {
        "workload": "geometry_from_nerf of one %d x %d view on the fitted NeRF (tests/golden/nerf_trained_fp16.npz): depth + normal "
                    "march with %d coarse + %d fine-network samples per ray (density gradient on the %d), then the shadow-ray march "
                    "to the front-lit ones of 512 lights for a sample of %d surface points per GPU" % (
                        H, W, n_c, n_c + n_f, n_c + n_f, n_pts),
        "precision": args.precision, "steps": steps, "finite": finite,
        "depth_normal": {
            "ms_per_view": elapsed / steps * 1e3, "rays_per_s": H * W * steps / elapsed,
            "foreground_rays_this_rank": int(fg.numel()), "samples_with_density_frac": listed_frac,
            "roofline": {"bound": "mfma", "kernel": "nerf_sigma_geo_kernel (coarse density; fine density of every sample) + "
                                                    "nerf_sigma_grad_kernel (forward + reverse sweep of the samples with a density), "
                                                    "all launches of a view; FLOPs as executed",
                         "achieved": flop_dn / (kernel_ms * 1e-3) / 1e12, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": flop_dn / (kernel_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, "traffic": None,
                         "flop_per_view_per_gpu": flop_dn, "kernels_ms_per_view": kernel_ms,
                         "mac_per_density_sample": GEO_MAC,
                         "share_of_stage": kernel_ms / (elapsed / steps * 1e3)}},
        "light_visibility": {
            "surface_points_per_gpu": n_pts, "front_lit_pairs_per_gpu": pairs, "ms": elapsed_l * 1e3,
            "pairs_per_s": pairs * world / elapsed_l if elapsed_l > 0 else None,      # (this rank's rate x ranks)
            "full_view_estimate_s": (float(fg.numel()) * world / max(n_pts * world, 1)) * elapsed_l,
            "roofline": {"bound": "mfma", "kernel": "nerf_sigma_geo_kernel (density only: the encoder + sigma tile), every launch of the stage",
                         "achieved": flop_lv / max(lvis_kernel_ms * 1e-3, 1e-9) / 1e12, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": flop_lv / max(lvis_kernel_ms * 1e-3, 1e-9) / 1e12 / PEAK_BF16_TFLOPS, "traffic": None,
                         "flop_per_gpu": flop_lv, "kernels_ms": lvis_kernel_ms,
                         "share_of_stage": lvis_kernel_ms / (elapsed_l * 1e3) if elapsed_l > 0 else None}}}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out.update(geometry_cpu_reference(nets, rayo_h, rayd_h, occu, depth, normal, surf, nrm, lv['lvis'], lxyz))
    return out


def geometry_cpu_reference(nets, rayo_h, rayd_h, occu, depth, normal, surf, nrm, lvis, lxyz, n_rays=512, n_pts=8):
    """oracle/geometry_ref.py (NumPy / torch-CPU restatement of geometry_from_nerf.py:93-246) on a bounded sample: the CPU
    baseline of both stages and the parity of the timed outputs on those rays / points."""
    from oracle import geometry_ref as GR
    torch.set_num_threads(host_cores())
    fgi = torch.nonzero(occu > 0.5)[:, 0].cpu().numpy()
    idx = np.sort(np.concatenate([fgi[np.linspace(0, len(fgi) - 1, n_rays // 2).astype(int)] if len(fgi) else np.zeros(0, int),
                                  np.linspace(0, rayo_h.shape[0] - 1, n_rays - n_rays // 2).astype(int)]))
    t0 = time.perf_counter()
    r_occu, r_depth, r_normal = GR.compute_depth_and_normal(rayo_h[idx], rayd_h[idx].astype(np.float32), nets[0], nets[1])
    t_dn = time.perf_counter() - t0
    g_occu, g_depth, g_normal = occu[idx].cpu().numpy(), depth[idx].cpu().numpy(), normal[idx].cpu().numpy()
    hit = r_occu > 0.5
    # the tests' terms (tests/test_gpu_reference_golden.py): the normal error as a VECTOR (the expected normal sum_i w_i n_i of a
    # fitted field is short, |n| 0.01-0.25, so a cosine between two such vectors says nothing), depth against the near-far range
    err_n = np.abs(g_normal - r_normal).max(1)
    z_range = 6. - 2.
    sp, sn = surf[:n_pts].cpu().numpy(), nrm[:n_pts].cpu().numpy()
    t0 = time.perf_counter()
    r_lvis = GR.compute_light_visibility(sp, sn, lxyz.reshape(-1, 3).astype(np.float32), nets[0], nets[1])
    t_lv = time.perf_counter() - t0
    g_lvis = lvis[:n_pts].cpu().numpy()
    pairs = int((r_lvis > 0).sum() + ((r_lvis == 0) & (g_lvis > 0)).sum())
    depth_err = float(np.abs(g_depth - r_depth)[hit].max()) if hit.any() else None
    return {
        "cpu_baseline": {"kind": "port", "cores": host_cores(), "unit": "rays/s (depth + normal) | pairs/s (light visibility)",
                         "value": len(idx) / t_dn, "light_visibility_pairs_per_s": max(pairs, 1) / t_lv,
                         "sample": "oracle/geometry_ref.py on %d rays of the view (%.1f s) and %d surface points x 512 lights (%.1f s)" % (
                             len(idx), t_dn, n_pts, t_lv)},
        "parity": {"rays_compared": int(len(idx)), "occu_max_abs": float(np.abs(g_occu - r_occu).max()),
                   "depth_max_abs_on_hits": depth_err,
                   "depth_rel_of_range": None if depth_err is None else depth_err / z_range,
                   "normal_vec_max_abs": float(err_n.max()), "rays_above_8e-2": int((err_n > 8e-2).sum()),
                   "lvis_points_compared": int(n_pts), "lvis_max_abs": float(np.abs(g_lvis - r_lvis).max()),
                   "tolerance": "occupancy 3e-2, depth 4 % of the near-far range, normal 8e-2 as a vector, visibility 4e-2 "
                                "(tests/test_gpu_reference_golden.py)",
                   "reference": "oracle/geometry_ref.py (fp32 / float64 autograd) on the same rays and points"}}


# ------------------------------------------------------------------------------------------------ OLAT leg (configs[4])
def olat_leg(args, ops, dev, rank, world, barrier, max_over_ranks):
    """One-light-at-a-time relighting of an 800 x 800 view (test.py:182, nerfactor.py:348-364): the render of the
    nerfactor_microfacet leg plus shade_olat_kernel, which writes 512 x 3 floats per foreground point — HBM-bound."""
    from nerfactor_amd import synth
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.models import get_model_class
    name = 'nerfactor_microfacet'
    torch.manual_seed(5)
    cfg = make_config(name, shape_mode='finetune', shape_model_ckpt='none', brdf_model_ckpt='none',
                      test_envmap_dir='', xyz_jitter_std='0', precision=args.precision)
    model = get_model_class(name)(cfg).to(dev)
    n = H * W
    sh = Shards(n, rank, world, args.scaling)
    batches, n_fg_local = [], 0
    for v in range(sh.n_views):
        hb = synth.surface_batch(n, seed=1 + 10 * v, n_lights=N_LIGHTS)
        batches.append(tuple(None if a is None else torch.from_numpy(a[sh.lo:sh.hi]).to(dev) for a in hb))
        n_fg_local += int(hb[5][sh.lo:sh.hi].sum())
    with KernelTimer(ops, ['shade_olat_fwd']) as kt:
        def step(k):
            kt.on = k is not None
            out = None
            for b in batches:
                out = model(b, mode='test', relight_olat=True)[0]['rgb_olat'][:1]
            return out
        elapsed, out = timed(step, args.steps, args.warmup, barrier)
        kt.on = False
    elapsed = max_over_ranks(elapsed)
    assert torch.isfinite(out).all()
    fg_per_call = n_fg_local / sh.n_views
    k_s = kt.mean_ms('shade_olat_fwd') * 1e-3
    # algorithmic bytes: 512 x 12 B written per foreground point; read: the point's visibility row (2 KiB) + 52 B
    alg = fg_per_call * (N_LIGHTS * 12 + N_LIGHTS * 4 + 52)
    gbs = alg / k_s / 1e9
    o_traffic, o_source = live_traffic(PMC_RENDER_LEGS, 'shade_olat_kernel') if (rank == 0 and world == 1 and not args.no_cpu_baseline) \
        else (None, None)
    if o_traffic is None:
        o_traffic, o_source = committed_traffic('shade_olat_kernel')
    return {
        "workload": "nerfactor_microfacet OLAT relighting (BASELINE.json configs[4], OLAT half): 800x800 surface points "
                    "per view (60 % foreground), 512 one-light renders per point, Model.call(mode='test', relight_olat=True)",
        "points_per_s": sh.rays_per_step_all_ranks * args.steps / elapsed,
        "ms_per_step": elapsed / args.steps * 1e3, "ms_per_view_per_gpu": elapsed / args.steps * 1e3 * world / sh.n_views,
        "olat_images_per_s": N_LIGHTS * sh.n_views * args.steps / elapsed,
        "roofline": {"bound": "hbm", "kernel": "shade_olat_kernel (%d foreground points x 512 lights x 3 floats written)"
                                               % int(fg_per_call),
                     "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                     "avg_launch_ms": k_s * 1e3, "algorithmic_bytes_per_launch": alg, "traffic": o_traffic,
                     "traffic_unit": "GB per launch", "traffic_source": o_source,
                     # (static count of the kernel: 2348 VALU instructions, ~2000 executed per point and wave — the per-light GGX
                     #  term, 512 of them per point: the kernel sits at its VALU issue rate, not at the HBM roof)
                     "note": "VALU-bound by the per-light microfacet term: the HBM roofline is the bound by bytes, not the binding one",
                     "share_of_step": k_s * sh.n_views / (elapsed / args.steps)},
    }



# ------------------------------------------------------------------------------------------------ relighting sweep (configs[4])
SWEEP_VIEWS = 4


def relight_sweep_leg(args, ops, dev, rank, world, barrier, max_over_ranks):
    """BASELINE.json configs[4], probe half: 8 novel environment maps x 4 test views through
    Model.call(mode='test', relight_probes=True) (test.py:163-204): one step = the 4 views (x world under weak scaling),
    every view's rays in contiguous ranges over the ranks, all 8 probes on every rank (sharding by
    probe would repeat the light-visibility MLP 8 times)."""
    from nerfactor_amd import synth
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.models import get_model_class
    name = 'nerfactor_microfacet'
    torch.manual_seed(5)
    cfg = make_config(name, shape_mode='finetune', shape_model_ckpt='none', brdf_model_ckpt='none',
                      test_envmap_dir='', xyz_jitter_std='0', precision=args.precision)
    model = get_model_class(name)(cfg).to(dev)
    for i, p in enumerate(synth.probes(N_PROBES, seed=20)):
        model.add_probe('p%d' % i, p)
    n = H * W
    sh = Shards(n, rank, world, args.scaling)
    n_views = SWEEP_VIEWS * sh.n_views
    batches = []
    for v in range(n_views):
        hb = synth.surface_batch(n, seed=10 + v, n_lights=N_LIGHTS)      # seeds 10-13
        batches.append(tuple(None if a is None else torch.from_numpy(a[sh.lo:sh.hi]).to(dev) for a in hb))
    steps = max(1, args.steps // 2)

    def step(k):
        out = None
        for b in batches:
            out = model(b, mode='test', relight_probes=True)[0]['rgb_probes']
        return out
    elapsed, out = timed(step, steps, min(args.warmup, 2), barrier)
    elapsed = max_over_ranks(elapsed)
    assert torch.isfinite(out).all() and out.shape[1] == N_PROBES
    return {
        "workload": "nerfactor_microfacet relighting sweep (BASELINE.json configs[4]): %d test views x %d novel "
                    "environment maps per step, 800x800 surface points per view (60 %% foreground), 512 lights, "
                    "Model.call(mode='test', relight_probes=True)" % (n_views, N_PROBES),
        "steps": steps, "views_per_step": n_views, "probes": N_PROBES,
        "ms_per_step": elapsed / steps * 1e3, "ms_per_view": elapsed / steps * 1e3 / n_views * world,
        "relit_images_per_s": n_views * N_PROBES * steps / elapsed,
        "points_per_s": n_views * n * steps / elapsed,
    }


def nerf_fp32_class_leg(args, ops, dev, rank, world, barrier, max_over_ranks):
    """The headline render at the reference's own arithmetic class (VERDICT r03 #6 i): precision = fp32 = every MLP
    operand a bf16 hi / lo pair, three MFMAs per product (nerf_mlp_x3.hip); a few steps, its own parity."""
    from nerfactor_amd import synth
    nets = synth.nerf_nets(seed=0)
    blobs = [ops.pack_nerf_weights(*synth.nerf_layers(n), prec='fp32').to(dev) for n in nets]
    sh = Shards(H * W, rank, world, args.scaling)
    views, host_views = [], []
    for v in range(sh.n_views):
        ang = 0.7 * v
        rayo, rayd = synth.camera_rays(H, W, cam_loc=(4 * np.cos(ang) * 0.8, 4 * np.sin(ang) * 0.8 - 0.1, 4 * 0.6))
        host_views.append((rayo, rayd))
        views.append((torch.from_numpy(rayo[sh.lo:sh.hi]).to(dev), torch.from_numpy(rayd[sh.lo:sh.hi]).to(dev)))
    steps = max(1, min(3, args.steps))
    elapsed, rgb = timed(lambda k: nerf_render_step(ops, views, blobs, None, 'fp32'), steps, 1, barrier)
    elapsed = max_over_ranks(elapsed)
    assert torch.isfinite(rgb).all()
    n_local = sh.hi - sh.lo
    tf = sh.n_views * n_local * (N_COARSE + N_COARSE + N_FINE) * FLOP_PER_POINT * steps / elapsed / 1e12
    out = {"workload": "the headline NeRF render with precision = fp32 (bf16 hi / lo operand pairs, 3 MFMAs per product)",
           "dtype": "bf16x3 (fp32-class)", "steps": steps, "rays_per_s": sh.rays_per_step_all_ranks * steps / elapsed,
           "ms_per_step": elapsed / steps * 1e3,
           "roofline": {"bound": "mfma", "kernel": "nerf_mlp_x3_kernel (whole step; achieved counts the ALGORITHMIC FLOPs "
                                                   "once, the kernel executes three MFMAs per product)",
                        "achieved": tf, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_BF16_TFLOPS,
                        "executed_tflops": 3 * tf, "traffic": None}}
    if rank == 0 and not args.no_cpu_baseline:
        _, idx, ref = nerf_cpu_reference(nets, *host_views[0], budget_s=0, timed_run=False)   # 1024 rays, no timing
        sel = idx[(idx >= sh.lo) & (idx < sh.hi)]
        keep = np.isin(idx, sel)
        o, d = (torch.from_numpy(a[sel]).to(dev) for a in host_views[0])
        got = nerf_render_step(ops, [(o, d)], blobs, prec='fp32').cpu().numpy()
        want = ref[1]['rgb'].numpy()[keep]
        err = np.abs(got - want).max(1)
        sig = np.minimum(ref[2]['sigma_last_coarse'].numpy()[keep], ref[2]['sigma_last_fine'].numpy()[keep])
        out["parity"] = {"psnr_db": psnr_uint8_luma(got, want), "max_abs_all_rays": float(err.max()),
                         "q99_abs": float(np.quantile(err, 0.99)), "rays_compared": int(len(sel)),
                         "rays_excluded_from_max_abs": 0, "rays_above_2e-4": int((err > 2e-4).sum()),
                         "rays_with_abs_sigma_last_below_0.01": int((sig <= 1e-2).sum()),
                         "reference": "oracle/torch_ref.py (fp32) on the same rays; stated tolerance 2e-4 on rgb (SURVEY.md "
                                      "section 8d) for rays off the two discontinuities of the formula (alpha_last = "
                                      "[sigma_last > 0], inverse-CDF bin edges), which are COUNTED here, not excluded"}
    return out


# ------------------------------------------------------------------------------------------------ the line
LINE_MAX_BYTES = 6144          # the driver reads ONE line; r05's 20.7 KB line (with an `Infinity` in it) came back `parsed: null`
DETAIL_NAME = 'bench_detail.json'


def strict(x):
    """`x` as strict JSON data: NumPy scalars / arrays -> Python, non-finite floats -> None (json.dumps(allow_nan=False)
    would raise on them; `Infinity` / `NaN` are not JSON)."""
    if isinstance(x, dict):
        return {str(k): strict(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [strict(v) for v in x]
    if isinstance(x, np.ndarray):
        return strict(x.tolist())
    if isinstance(x, (np.bool_, bool)):
        return bool(x)
    if isinstance(x, (np.integer, int)):
        return int(x)
    if isinstance(x, (np.floating, float)):
        x = float(x)
        return x if np.isfinite(x) else None
    if x is None or isinstance(x, str):
        return x
    return str(x)


def _sig(x, n=6):
    """Floats of the line rounded to n significant digits (the detail file keeps every digit)."""
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, list):
        return [_sig(v, n) for v in x]
    if isinstance(x, float) and x != 0:
        return float('%.*g' % (n, x))
    return x


def _pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _parity_block(p, tol_key, count_key, n_key):
    """psnr / max-abs / how many were compared / how many are above the stated tolerance.  An identical frame has an
    infinite PSNR: `psnr_db: null, identical: true`."""
    if not isinstance(p, dict):
        return None
    out = {"psnr_db": p.get("psnr_db"), "max_abs": p.get("max_abs", p.get("max_abs_all_rays")),
           n_key: p.get(n_key), tol_key: p.get(count_key)}
    if "psnr_db" in p and (p["psnr_db"] is None or not np.isfinite(p["psnr_db"])):
        out["psnr_db"], out["identical"] = None, True
    return out


def assemble(args_steps, args_warmup, scaling, precision, world, backend, rehearsal, legs):
    """The full result object (-> bench_detail.json) from the legs' dictionaries.  `legs`: {'nerf': {...} | None,
    'nerfactor': {name: {...}}, 'train': {name: {...}}, 'olat', 'relight', 'fp32_class', 'geometry', 'wall_s': {...}}."""
    out = {"metric": "rays/sec (NeRF coarse+fine render, 64+128 samples/ray)", "value": None,
           "unit": "rays/s", "n_gpus": world, "steps": args_steps, "warmup": args_warmup,
           "ms_per_step": None, "higher_is_better": True, "scaling": scaling,
           "vs_baseline": None, "dtype": "bf16" if precision == "bf16" else "bf16x3 (fp32-class: hi/lo operand pairs)",
           "data": "synthetic", "world_size": world, "collective_backend": backend}
    if rehearsal:
        out["rehearsal"] = "all ranks share GPU 0 over gloo: a functional check of the N > 1 path, not a measurement"
    nerf = legs.get('nerf')
    if nerf is not None:   # (profiling runs may time the NeRFactor legs alone: --legs nerfactor)
        out.update(nerf)
        if "parity" in nerf:
            out["psnr_db"], out["max_abs"] = nerf["parity"]["psnr_db"], nerf["parity"]["max_abs"]
    for k in ('nerfactor', 'train', 'olat', 'relight', 'fp32_class', 'geometry', 'wall_s'):
        if legs.get(k):
            out[k] = legs[k]
    return strict(out)


def compact(full):
    """The ONE line the driver parses: the contract's keys, `roofline`, `cpu_baseline`, `parity` of the
    headline leg, and one small object per further leg (step time, roofline fraction, worst error).  Everything else
    is in bench_detail.json next to bench.py."""
    keys = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "world_size", "collective_backend", "rehearsal", "error")
    line = _pick(full, *keys)
    cfg = full.get("config")
    if cfg:
        line["config"] = _pick(cfg, "workload", "views_per_step", "rays_per_view", "rays_per_step_per_gpu",
                               "n_samples_coarse", "n_samples_fine", "kernel_variant")
        line["config"]["workload"] = "lego_3072-shaped NeRF coarse+fine MLP render, 800x800 rays/view, 64+128 samples (BASELINE.json configs[1])"
    r = full.get("roofline")
    if r:
        line["roofline"] = _pick(r, "bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_unit",
                                 "algorithmic_hbm_gb", "avg_launch_pair_ms", "flop_per_launch_pair")
        src = r.get("traffic_source") or ""
        line["roofline"]["traffic_source"] = ("measured by this run (rocprofv3 --pmc children)" if src.startswith("measured")
                                              else src[:80] if src else None)
    c = full.get("cpu_baseline")
    if c:
        line["cpu_baseline"] = _pick(c, "value", "unit", "cores", "kind")
        line["cpu_baseline"]["sample"] = (c.get("sample") or "").split(';')[0][:160]
        if "gpu_over_cpu" in full:
            line["gpu_over_cpu"] = full["gpu_over_cpu"]
    if "parity" in full:
        line["parity"] = _parity_block(full["parity"], "rays_above_tol", "rays_above_3e-2", "rays_compared")
        line["parity"]["against"] = "CPU port (oracle/torch_ref.py, fp32) on rays of the timed frame"
        line["parity"]["tolerance"] = "PSNR >= 40 dB, max-abs <= 3e-2 on every ray (BASELINE.md section 4)"
        pr = full.get("parity_vs_reference")
        if isinstance(pr, dict) and "error" not in pr:     # the same frame against outputs of the reference's own Python
            line["parity"]["vs_reference_python"] = _parity_block(pr, "rays_above_tol", "rays_above_3e-2", "rays_compared")
        pf = full.get("parity_fitted_weights")
        if pf:
            line["parity"]["fitted_weights"] = _parity_block(pf, "rays_above_tol", "rays_above_3e-2", "rays_compared")
            if isinstance(pf.get("vs_reference"), dict) and "error" not in pf["vs_reference"]:
                line["parity"]["fitted_weights"]["vs_reference_python"] = _parity_block(
                    pf["vs_reference"], "rays_above_tol", "rays_above_3e-2", "rays_compared")
            cost = pf.get("coarse_refinement_cost") or {}
            if cost:       # what holding the tolerance on every ray of a FITTED scene costs there (the timed frame: glorot weights)
                line["parity"]["fitted_weights"]["refine_extra_frame_time"] = cost.get("extra_frame_time")
                line["parity"]["fitted_weights"]["coarse_samples_refined_frac"] = cost.get("coarse_samples_refined_frac")
            if pf.get("coarse_precision_bf16"):
                line["parity"]["fitted_weights"]["rays_above_tol_without_refinement"] = pf["coarse_precision_bf16"].get("rays_above_3e-2")
    legs = {}
    for name, leg in (full.get("nerfactor") or {}).items():
        e = {"ms_per_step": leg.get("ms_per_step"), "frac": (leg.get("roofline") or {}).get("frac")}
        if "parity" in leg:
            e["max_abs"], e["points_above_tol"] = leg["parity"].get("max_abs"), leg["parity"].get("points_above_3e-2")
        if "brdf_spec" in leg:
            e["brdf_spec_ms"] = leg["brdf_spec"].get("avg_launch_ms")
        if "cpu_baseline" in leg:
            e["gpu_over_cpu"] = leg.get("gpu_over_cpu")
        legs[name] = e
    for name, leg in (full.get("train") or {}).items():
        if "error" in leg:
            legs["train_" + name] = {"error": str(leg["error"])[:120]}
            continue
        e = {"ms_per_step": leg.get("ms_per_step"), "ms_per_step_eager": leg.get("ms_per_step_eager"),
             "frac": (leg.get("roofline") or {}).get("frac")}
        if leg.get("points_with_gradient_frac") is not None:
            e["points_with_gradient_frac"] = leg["points_with_gradient_frac"]
        if leg.get("fitted_scene"):
            e["fitted_scene_ms_per_step"] = leg["fitted_scene"].get("ms_per_step")
            e["fitted_scene_points_with_gradient_frac"] = leg["fitted_scene"].get("points_with_gradient_frac")
        if leg.get("parity"):
            e["grad_rel_vs_bf16_oracle"] = leg["parity"].get("grad_rel_frobenius_vs_bf16_oracle_worst")
            e["grad_rel_vs_reference"] = leg["parity"].get("grad_rel_frobenius_vs_reference_worst")
        if leg.get("fp32"):
            e["fp32_ms_per_step"] = leg["fp32"].get("ms_per_step")
            e["fp32_grad_rel_vs_reference"] = (leg["fp32"].get("parity") or {}).get("grad_rel_frobenius_vs_reference_worst")
        legs["train_" + name] = e
    if full.get("olat"):
        legs["olat"] = {"ms_per_step": full["olat"].get("ms_per_step"), "frac": (full["olat"].get("roofline") or {}).get("frac"),
                        "bound": "hbm"}
    if full.get("relight"):
        legs["relight"] = _pick(full["relight"], "ms_per_step", "ms_per_view", "views_per_step", "probes")
    if full.get("fp32_class"):
        f = full["fp32_class"]
        legs["fp32_class"] = {"ms_per_step": f.get("ms_per_step"), "frac": (f.get("roofline") or {}).get("frac"),
                              "max_abs": (f.get("parity") or {}).get("max_abs_all_rays")}
    if full.get("geometry"):
        g = full["geometry"]
        e = {"ms_per_view": (g.get("depth_normal") or {}).get("ms_per_view"),
             "frac": ((g.get("depth_normal") or {}).get("roofline") or {}).get("frac"),
             "lvis_frac": ((g.get("light_visibility") or {}).get("roofline") or {}).get("frac")}
        if (g.get("depth_normal") or {}).get("samples_with_density_frac") is not None:
            e["samples_with_density_frac"] = g["depth_normal"]["samples_with_density_frac"]
        e.update(_pick(g.get("parity") or {}, "normal_vec_max_abs", "rays_above_8e-2", "rays_compared", "depth_rel_of_range", "lvis_max_abs"))
        legs["geometry"] = e
    if legs:
        line["legs"] = legs
    line["detail"] = DETAIL_NAME
    return _sig(strict(line))


def emit(full, stream=None, detail_dir=None):
    """Write bench_detail.json (every leg, every digit) and print the compact line — strict JSON, one line, <= 6 KB."""
    detail_dir = ROOT if detail_dir is None else detail_dir
    try:
        with open(os.path.join(detail_dir, DETAIL_NAME), 'w') as h:
            json.dump(full, h, allow_nan=False, indent=1)
    except OSError:                      # (a read-only tree must not take the line down)
        pass
    line = compact(full)
    text = json.dumps(line, allow_nan=False, separators=(',', ':'))
    if len(text) > LINE_MAX_BYTES:       # never silently: drop the per-leg objects before anything of the contract
        line["legs"] = {"dropped": "line over %d bytes; see %s" % (LINE_MAX_BYTES, DETAIL_NAME)}
        text = json.dumps(line, allow_nan=False, separators=(',', ':'))
    print(text, file=stream or sys.stdout, flush=True)
    return text


# ------------------------------------------------------------------------------------------------ launching N ranks
def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def self_launch(args, argv):
    """`python bench.py --gpus N` with no torchrun around it (the reference's MirroredStrategy is one process that uses every
    visible GPU, trainvali.py:263-266): re-execute this script as N ranks, one per GPU, under torch.distributed.run on
    127.0.0.1; rank 0 prints the line.  Refuses — with a JSON error line and a non-zero exit — when the box has fewer GPUs."""
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    rehearsal = os.environ.get('NFX_BENCH_REHEARSAL') == '1' or os.environ.get('NFX_REHEARSAL') == '1'
    if have < args.gpus and not (rehearsal and have >= 1):
        print(json.dumps({"error": "--gpus %d but this box has %d GPU(s)" % (args.gpus, have), "n_gpus": args.gpus,
                          "gpus_visible": have, "metric": "rays/sec (NeRF coarse+fine render, 64+128 samples/ray)",
                          "value": None}, allow_nan=False), flush=True)
        raise SystemExit(2)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


# ------------------------------------------------------------------------------------------------ main
def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak')
    ap.add_argument('--legs', default='nerf,nerfactor_microfacet,nerfactor,train,olat,relight,fp32_class,geometry')
    ap.add_argument('--train-models', default='nerfactor_microfacet,nerfactor,nerf')
    ap.add_argument('--cpu-budget', type=float, default=10., help="seconds of CPU work for the NeRF baseline sample")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-hip-graph', action='store_true', help="train leg: time the eager step only")
    ap.add_argument('--no-last-sample-refine', action='store_true',
                    help="A/B: skip the fp32-class re-evaluation of every ray's last sample (the r03 render)")
    ap.add_argument('--precision', choices=('bf16', 'fp32'), default='bf16',
                    help="MLP operand type: bf16 (the headline) or fp32 = bf16 hi/lo pairs, 3 MFMAs per product")
    ap.add_argument('--measure-traffic', action='store_true',
                    help="roofline.traffic from two rocprofv3 --pmc child runs of this command (default: the committed digest, labelled)")
    ap.add_argument('--fp32-modes', action='store_true',
                    help="train leg: time the precision = fp32 step in every matrix mode (default: the model's default mode only)")
    ap.add_argument('--force-group', action='store_true',
                    help="one rank: initialise the nccl (RCCL) process group anyway, so that the training legs' all-reduce runs on RCCL")
    argv = sys.argv[1:] if argv is None else argv
    args = ap.parse_args(argv)
    global MEASURE_TRAFFIC
    MEASURE_TRAFFIC = args.measure_traffic

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        self_launch(args, argv)          # does not return
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libnfx has no CPU path")
    # NFX_BENCH_REHEARSAL=1: every rank on GPU 0 over gloo — exercises the whole N > 1 code path (sharding, barriers,
    # max over ranks, rank-0 assembly) on a one-GPU box; the numbers mean nothing (the ranks share one GPU)
    rehearsal = os.environ.get('NFX_BENCH_REHEARSAL') == '1' or os.environ.get('NFX_REHEARSAL') == '1'
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    import torch.distributed as dist
    from nerfactor_amd import dist as nfx_dist
    nfx_dist.init_from_env(backend='gloo' if rehearsal else 'nccl', device=None if rehearsal else dev,
                           force=args.force_group)
    from nerfactor_amd import build
    build.build()
    from nerfactor_amd import ops

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        return nfx_dist.max_over_ranks(x, device=dev)

    names = [s for s in args.legs.split(',') if s]
    wall, legs = {}, {}

    def run(key, fn, *a):
        t0 = time.perf_counter()
        res = fn(*a, args, ops, dev, rank, world, barrier, max_over_ranks)
        wall[key] = round(time.perf_counter() - t0, 2)
        return res

    if 'nerf' in names:
        legs['nerf'] = run('nerf', nerf_leg)
    legs['nerfactor'] = {n: run(n, nerfactor_leg, n) for n in names if n in ('nerfactor_microfacet', 'nerfactor')}
    legs['train'] = {}
    if 'train' in names and args.precision == 'bf16':
        for n in (m for m in args.train_models.split(',') if m):
            # (a check_numerics failure comes back as {"error": ...}, agreed on by all ranks inside the leg)
            legs['train'][n] = run('train_' + n, train_leg, n)
    if 'olat' in names:
        legs['olat'] = run('olat', olat_leg)
    if 'relight' in names:
        legs['relight'] = run('relight', relight_sweep_leg)
    if 'fp32_class' in names and args.precision == 'bf16':
        legs['fp32_class'] = run('fp32_class', nerf_fp32_class_leg)
    if 'geometry' in names:
        legs['geometry'] = run('geometry', geometry_leg)
    legs['wall_s'] = wall
    if rank == 0:
        backend = dist.get_backend() if dist.is_initialized() else "none (single process)"
        emit(assemble(args.steps, args.warmup, args.scaling, args.precision, world, backend, rehearsal, legs))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
