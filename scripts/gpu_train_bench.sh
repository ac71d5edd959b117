cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/train
for m in ${MODELS-nerf shape nerfactor_microfacet nerfactor}; do timeout 300 python scripts/bench_train.py --model $m --steps 20 2>&1 | tail -1 | tee gpurun_out/train/bench_train_$m.json; done
timeout 600 python scripts/bench_geometry.py 2>&1 | tail -1 | tee gpurun_out/train/bench_geometry.json
if [ "${PROF:-0}" = "1" ]; then
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/train/prof_${PROF_MODEL:-nerf} -o train -- python $GRAFT_REPO_ROOT/scripts/bench_train.py --model ${PROF_MODEL:-nerf} --steps 10 > $GRAFT_REPO_ROOT/gpurun_out/train/prof.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/train/prof_${PROF_MODEL:-nerf} -name "*kernel_stats*" | head -1 | xargs -r head -8
fi
