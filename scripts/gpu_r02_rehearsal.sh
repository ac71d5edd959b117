#!/bin/bash
# Two ranks on ONE GPU over gloo (NFX_REHEARSAL=1): the multi-process paths of the drivers, end to end — trainvali
# (parameter broadcast, sharded rays, one [gradients | loss] all-reduce per step, rank-0 checkpoints), test.py (each
# view's rays split over the ranks, uint8 rows gathered on rank 0) — against the same commands run by one process.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out/rehearsal
mkdir -p $OUT
S=/tmp/nfx_reh
rm -rf $S; mkdir -p $S
python - <<PY
import sys; sys.path.insert(0, '.')
from tests import synth_scene
synth_scene.write_scene('$S', imh=24, imw=24, n_train=3, n_val=1, n_test=2)
PY
OV="data_root=$S/data,data_nerf_root=$S/nerf,imh=24,n_rays_per_step=128,vali_batches=1,vis_train_batches=1,epochs=4,ckpt_period=2,vali_period=2,use_nerf_alpha=False,shape_mode=finetune,shape_model_ckpt=none,test_envmap_dir=,seed=3,xyz_jitter_std=0"
ls $S
run() {  # name, launcher...
  local name=$1; shift
  "$@" -m nerfactor_amd.nerfactor.trainvali --config=nerfactor_microfacet.ini --config_override="$OV,outroot=$S/out_$name" > $OUT/train_$name.log 2>&1 || { echo "trainvali $name FAILED"; tail -5 $OUT/train_$name.log; }
  "$@" -m nerfactor_amd.nerfactor.test --ckpt=$S/out_$name/lr5e-3/checkpoints/ckpt-2 > $OUT/test_$name.log 2>&1 || { echo "test $name FAILED"; tail -5 $OUT/test_$name.log; }
}
run one python
NFX_REHEARSAL=1 run two python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517
python - <<PY
import glob, numpy as np, torch
from PIL import Image
a = torch.load('$S/out_one/lr5e-3/checkpoints/ckpt-2', map_location='cpu')['net']
b = torch.load('$S/out_two/lr5e-3/checkpoints/ckpt-2', map_location='cpu')['net']
d = max(float((a[k].float() - b[k].float()).abs().max()) for k in a)
print('parameters after 4 epochs, one process vs two ranks: max |diff| %.3e over %d tensors' % (d, len(a)))
allv = torch.cat([(a[k].float() - b[k].float()).abs().reshape(-1) for k in a if a[k].dtype.is_floating_point])
print('  mean |diff| %.3e, 99th percentile %.3e  (Adam moves every element by about lr = 5e-3 per step whatever the size of '
      'its gradient, so elements whose gradient is at summation-order noise level diverge by up to 2 lr x 12 steps)' % (
          float(allv.mean()), float(allv.quantile(0.99))))
top = sorted(((float((a[k].float() - b[k].float()).abs().max()), k) for k in a), reverse=True)[:5]
print('  largest:', [(k, '%.2e' % v) for v, k in top])
import csv
for name in ('one', 'two'):
    rows = list(csv.DictReader(open('$S/out_%s/lr5e-3/summary_train/scalars.csv' % name)))
    print('  loss_train', name, [(r['step'], round(float(r['value']), 6)) for r in rows if r['tag'] == 'loss_train'])
ims = sorted(glob.glob('$S/out_one/lr5e-3/vis_test/ckpt-2/batch*/pred_rgb.png'))
worst = 0
for f in ims:
    g = f.replace('out_one', 'out_two')
    x, y = np.asarray(Image.open(f)).astype(int), np.asarray(Image.open(g)).astype(int)
    worst = max(worst, int(np.abs(x - y).max()))
print('test renders: %d views, max |uint8 diff| between the two trained models %d' % (len(ims), worst))
PY
# ---- stage 1 / 2 of the workflow: a NeRF trained once (one process), then nerf_test and geometry_from_nerf by one
# process and by two ranks (rays of every view split over the ranks; float maps written by both ranks into one
# memory-mapped file, uint8 previews gathered on rank 0)
OVN="data_root=$S/data,imh=24,n_rays_per_step=128,vali_batches=1,vis_train_batches=1,epochs=2,ckpt_period=2,vali_period=2,n_samples_coarse=16,n_samples_fine=32,lr=5e-4,outroot=$S/out_nerf"
python -m nerfactor_amd.nerfactor.trainvali --config=nerf.ini --config_override="$OVN" > $OUT/train_nerf.log 2>&1 || { echo "nerf trainvali FAILED"; tail -5 $OUT/train_nerf.log; }
geo() {  # name, launcher...
  local name=$1; shift
  "$@" -m nerfactor_amd.nerfactor.geometry_from_nerf --trained_nerf=$S/out_nerf/lr5e-4 --out_root=$S/surf_$name --lvis_far=1 --scene_bbox=-1.5,1.5,-1.5,1.5,-1.5,1.5 > $OUT/geo_$name.log 2>&1 || { echo "geometry_from_nerf $name FAILED"; tail -5 $OUT/geo_$name.log; }
}
geo one python
NFX_REHEARSAL=1 geo two python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519
python - <<PY
import glob, numpy as np
from PIL import Image
views = sorted(glob.glob('$S/surf_one/*/xyz.npy'))
worst = {}
for f in views:
    for name in ('xyz.npy', 'normal.npy', 'lvis.npy'):
        a, b = np.load(f.replace('xyz.npy', name)), np.load(f.replace('surf_one', 'surf_two').replace('xyz.npy', name))
        assert a.shape == b.shape, (f, name, a.shape, b.shape)
        worst[name] = max(worst.get(name, 0.), float(np.abs(a - b).max()))
    for name in ('alpha.png', 'normal.png', 'lvis.png'):
        a = np.asarray(Image.open(f.replace('xyz.npy', name))).astype(int)
        b = np.asarray(Image.open(f.replace('surf_one', 'surf_two').replace('xyz.npy', name))).astype(int)
        worst[name] = max(worst.get(name, 0), int(np.abs(a - b).max()))
print('geometry_from_nerf, %d views, one process vs two ranks: max |diff|' % len(views), worst)
PY
