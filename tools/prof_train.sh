# rocprofv3 kernel statistics of the training step (16 tiles): bash tools/prof_train.sh (through gpurun)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/train_prof -o t -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/train_prof.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py $GRAFT_REPO_ROOT/gpurun_out/train_prof/t_kernel_trace.csv
python3 - <<'PY'
import csv, os, json
print(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/train_prof.log').read().strip().splitlines()[-1][:160])
rows = list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/train_prof/t_kernel_stats.csv')))
for r in rows[:40]:
    print('%6.2f%% %6d %9.1f  %s' % (float(r['Percentage']), int(r['Calls']), float(r['AverageNs']) / 1e3, r['Name'][:100]))
PY
