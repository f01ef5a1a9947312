cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pack_prof -o t -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pack_prof.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/pack_prof.log | cut -c1-120
grep "pack_weights" $GRAFT_REPO_ROOT/gpurun_out/pack_prof/t_kernel_stats.csv | cut -c1-200
