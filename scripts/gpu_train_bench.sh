cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/train
for m in nerf shape nerfactor_microfacet nerfactor; do timeout 300 python scripts/bench_train.py --model $m --steps 20 2>&1 | tail -1 | tee gpurun_out/train/bench_train_$m.json; done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/train/prof_nerf -o nerf_train -- python $GRAFT_REPO_ROOT/scripts/bench_train.py --model nerf --steps 10 > $GRAFT_REPO_ROOT/gpurun_out/train/prof_nerf.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/train/prof_nerf -name "*kernel_stats*" | head -1 | xargs -r head -14
