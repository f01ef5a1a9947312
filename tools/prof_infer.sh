# rocprofv3 kernel statistics of the inference step (64 tiles): bash tools/prof_infer.sh (through gpurun)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/infer_prof -o t -- python3 $GRAFT_REPO_ROOT/bench.py --mode infer --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/infer_prof.log 2>&1
python3 - <<'PY'
import csv, os, json
print(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/infer_prof.log').read().strip().splitlines()[-1][:160])
rows = list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/infer_prof/t_kernel_stats.csv')))
for r in rows[:9]:
    print('%6.2f%% %6d %9.1f  %s' % (float(r['Percentage']), int(r['Calls']), float(r['AverageNs']) / 1e3, r['Name'][:100]))
PY
