cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ws_prof -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_conv_ws.py 16 > $GRAFT_REPO_ROOT/gpurun_out/ws_prof.log 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/ws_prof.log | tail -7
grep -E "conv_ws|conv_fwd" $GRAFT_REPO_ROOT/gpurun_out/ws_prof/t_kernel_stats.csv | cut -c1-200
