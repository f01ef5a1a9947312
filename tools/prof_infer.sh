# rocprofv3 kernel statistics of the inference step (64 tiles): bash tools/prof_infer.sh [bf16|fp32] (through gpurun)
DT=${1:-bf16}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/infer_prof_$DT -o t -- python3 $GRAFT_REPO_ROOT/bench.py --mode infer --dtype $DT --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/infer_prof_$DT.log 2>&1
python3 - $DT <<'PY'
import csv, os, json, sys
dt = sys.argv[1]
root = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/'
print(open(root + 'infer_prof_%s.log' % dt).read().strip().splitlines()[-1][:160])
rows = list(csv.DictReader(open(root + 'infer_prof_%s/t_kernel_stats.csv' % dt)))
# per forward: calls of the head kernel = forwards; the roofline kernel's own launches inflate the dominant kernel's count
fw = max(1, max(int(r['Calls']) for r in rows if 'dam_head_' in r['Name']))
for r in rows[:24]:
    print('%6.2f%% %6d  %7.2f/fw %9.1f us  %s' % (float(r['Percentage']), int(r['Calls']), int(r['Calls']) / fw, float(r['AverageNs']) / 1e3, r['Name'][:110]))
PY
