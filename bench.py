#!/usr/bin/env python
"""bench.py — rays/sec of the NeRF coarse+fine render hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY.md §8d "C2"): one 800x800 view per GPU, 64 coarse +
128 fine samples per ray, white background, perturb off, synthetic glorot weights (opaque
variant), rays from the reference's pin-hole generator.  A "step" = one full pass of the hot path
over one view's 640 000 rays per GPU: normalise -> gen_z -> coarse MLP -> composite ->
hierarchical resample -> fine MLP -> composite.  Inputs are resident in HBM before the timed
region.  N > 1: one process per GPU (torchrun), each rank renders its own view (rays are
independent; no data-path collective) => weak scaling; value = all rays of all ranks / max time.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel = fused NeRF MLP, MFMA-bound,
algorithmic FLOPs / HIP-event kernel time) and, at N == 1, `cpu_baseline` (torch-CPU fp32 port of
the reference op sequence on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W = 800
N_COARSE, N_FINE = 64, 128
FLOP_PER_POINT = 2 * 593408          # 593 408 MAC per sample point
PEAK_BF16_TFLOPS = 2500.0            # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)


def synth_inputs(rank):
    from tests import common
    nets = common.nerf_nets(seed=0)
    ang = 0.7 * rank
    cam = (4 * np.cos(ang) * 0.8, 4 * np.sin(ang) * 0.8 - 0.1, 4 * 0.6)
    rayo, rayd = common.camera_rays(H, W, cam_loc=cam)
    return nets, rayo, rayd


def render_step(ops, o, d_raw, blobs, ev=None):
    d = ops.l2_normalize3(d_raw, 1e-12)
    z = ops.gen_z(2., 6., N_COARSE, o.shape[0], device=o.device)
    if ev is not None:
        ev[0].record()
    raw = ops.nerf_mlp_fwd(o, d, z, blobs[0])
    if ev is not None:
        ev[1].record()
    _, _, _, _, w = ops.composite_fwd(raw, z, d, white_bg=True)
    z_all = ops.sample_fine(z, w, N_FINE)
    if ev is not None:
        ev[2].record()
    raw = ops.nerf_mlp_fwd(o, d, z_all, blobs[1])
    if ev is not None:
        ev[3].record()
    rgb, occu, depth, disp, _ = ops.composite_fwd(raw, z_all, d, white_bg=True, want_weights=False)
    return rgb


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


def cpu_baseline(nets, rayo, rayd, budget_s=20.):
    """torch-CPU fp32 port of the reference op sequence (oracle/torch_ref.py) on the host cores.
    The thread count is probed (all usable cores is not always fastest for 65 536-row GEMMs);
    `cores` reports the count actually used for the timed sample."""
    from oracle import torch_ref
    tn = [torch_ref.to_torch_net(n) for n in nets]
    idx = np.random.default_rng(0).permutation(rayo.shape[0])
    avail = host_cores()
    cands = sorted({c for c in (avail, avail // 2, 64, 32, 16, 8) if 1 <= c <= avail}, reverse=True)

    def run(n, threads):
        torch.set_num_threads(threads)
        o, d = torch.from_numpy(rayo[idx[:n]]), torch.from_numpy(rayd[idx[:n]])
        t0 = time.perf_counter()
        torch_ref.render_rays(o, d, tn[0], tn[1])
        return time.perf_counter() - t0

    with torch.no_grad():
        run(64, cands[-1])  # warm the allocator / BLAS
        probe = {}
        for c in cands:
            probe[c] = run(256, c)
            if probe[c] > 8.:
                continue
        best = min(probe, key=probe.get)
        n = int(min(65536, max(256, 256 * budget_s / max(probe[best], 1e-3))))
        n = max(256, (n // 256) * 256)
        dt = run(n, best)
    return {
        "value": n / dt, "unit": "rays/s", "cores": best, "kind": "port",
        "cores_available": avail,
        "sample": "%d rays of the same 800x800 view, 64+128 samples, torch-CPU fp32 "
                  "(oracle/torch_ref.py, mlp_chunk=65536), %.1f s; thread probe %s" % (
                      n, dt, {k: round(v, 2) for k, v in probe.items()})}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (
            args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libnfx has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    import torch.distributed as dist
    from nerfactor_amd import dist as nfx_dist
    nfx_dist.init_from_env(backend='nccl', device=dev)

    from nerfactor_amd import build
    build.build()
    from nerfactor_amd import ops

    nets, rayo, rayd = synth_inputs(rank)
    from tests import common
    blobs = [ops.pack_nerf_weights(*common.nerf_layers(n)).to(dev) for n in nets]
    o = torch.from_numpy(rayo).to(dev)
    d = torch.from_numpy(rayd).to(dev)
    n_rays = o.shape[0]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        render_step(ops, o, d, blobs)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        rgb = render_step(ops, o, d, blobs, evs[k])
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = nfx_dist.max_over_ranks(elapsed, device=dev)
    assert torch.isfinite(rgb).all()

    # dominant kernel: the fused NeRF MLP (two launches per step: coarse + fine)
    mlp_ms = [e[0].elapsed_time(e[1]) + e[2].elapsed_time(e[3]) for e in evs]
    fine_ms = [e[2].elapsed_time(e[3]) for e in evs]
    pts_per_step = n_rays * (N_COARSE + N_COARSE + N_FINE)
    mlp_avg_s = float(np.mean(mlp_ms)) * 1e-3
    achieved = pts_per_step * FLOP_PER_POINT / mlp_avg_s / 1e12
    fine_tf = n_rays * (N_COARSE + N_FINE) * FLOP_PER_POINT / (float(np.mean(fine_ms)) * 1e-3) / 1e12

    # HBM traffic of the dominant kernel: rocprofv3 PMC passes cannot run inside this process, so
    # the figure comes from the committed digest of the same workload (scripts/gpu_pmc.sh ->
    # scripts/pmc_digest.py): FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, per launch pair.
    traffic = None
    variant = os.environ.get("NFX_NERF_VARIANT", "7")
    dig = os.path.join(ROOT, 'profiles', 'r01', 'pmc_variant%s_digest.json' % variant)
    if os.path.exists(dig):
        for k, v in json.load(open(dig)).items():
            if 'nerf_mlp' in k and 'hbm_write_bytes' in v:
                # digest = average over the coarse and the fine dispatch; a launch pair = both
                traffic = 2 * (v.get('hbm_read_bytes_corrected', 0) + v['hbm_write_bytes']) / 1e9
    if rank == 0:
        out = {
            "metric": "rays/sec (NeRF coarse+fine render, 64+128 samples/ray)",
            "value": world * n_rays * args.steps / elapsed,
            "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {
                "workload": "lego_3072-shaped NeRF coarse+fine MLP render, 800x800 rays per GPU, "
                            "64+128 samples (BASELINE.json configs[1])",
                "rays_per_step_per_gpu": n_rays, "n_samples_coarse": N_COARSE,
                "n_samples_fine": N_FINE, "weights": "glorot seed 0, opaque variant",
                "kernel_variant": variant},
            "roofline": {
                "bound": "mfma", "kernel": "%s (coarse + fine launches)" % {
                    "0": "nerf_mlp_bf16_kernel<2, 4>", "1": "nerf_mlp_bf16_kernel<1, 8>",
                    "7": "nerf_mlp_bf16_v6_kernel<0, 1>, LDS-DMA weight stream",
                    "8": "nerf_mlp_bf16_v6_kernel<0, 2>"}.get(variant, "nerf_mlp_bf16_v%s_kernel" % variant),
                "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_unit": "GB per launch pair",
                "algorithmic_hbm_gb": pts_per_step * 20 / 1e9,
                "flop_per_launch_pair": pts_per_step * FLOP_PER_POINT,
                "avg_launch_pair_ms": mlp_avg_s * 1e3, "fine_launch_tflops": fine_tf,
                "mlp_share_of_step": mlp_avg_s / (elapsed / args.steps)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(nets, rayo, rayd)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
