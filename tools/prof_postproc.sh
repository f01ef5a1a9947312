cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/tools/bench_postproc.py 64 50
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pp_prof -o t -- python3 $R/tools/bench_postproc.py 64 20 > /dev/null 2>&1
python3 - <<'PY'
import csv, os
rows = list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/pp_prof/t_kernel_stats.csv')))
for r in rows[:32]:
    print('%6.2f%% %6d %9.1f  %s' % (float(r['Percentage']), int(r['Calls']), float(r['AverageNs']) / 1e3, r['Name'][:110]))
PY
rm -f $R/gpurun_out/pp_prof/t_kernel_trace.csv
