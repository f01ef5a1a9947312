# rocprofv3 kernel statistics of tools/bench_bn.py (BatchNorm backward shapes of the training step): bash tools/prof_bn.sh (through gpurun)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/bn_prof -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_bn.py 16 > $GRAFT_REPO_ROOT/gpurun_out/bn_prof.log 2>&1
grep -v "^[WE]2026" $GRAFT_REPO_ROOT/gpurun_out/bn_prof.log | tail -9
python3 - <<'PY'
import csv, os
for r in csv.DictReader(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/bn_prof/t_kernel_stats.csv')):
    if 'bn_' in r['Name'] or 'reduce' in r['Name']:
        print('%6d %9.1f min %8.1f max %8.1f  %s' % (int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, r['Name'][:90]))
PY
