# rocprofv3 kernel statistics + main-stream timeline of the fp32-mode training step (16 tiles): bash tools/prof_train32.sh (through gpurun)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/train32_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --dtype fp32 --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $OUT.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py $OUT/t_kernel_trace.csv > $OUT/timeline.txt
cat $OUT/timeline.txt
python3 - <<'PY'
import csv, os, collections
root = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/train32_prof'
print(open(root + '.log').read().strip().splitlines()[-1][:160])
rows = list(csv.DictReader(open(root + '/t_kernel_stats.csv')))
for r in rows[:34]:
    print('%6.2f%% %6d %9.1f  %s' % (float(r['Percentage']), int(r['Calls']), float(r['AverageNs']) / 1e3, r['Name'][:110]))
# per-stream totals by kernel family over the last step of the main stream
tr = list(csv.DictReader(open(root + '/t_kernel_trace.csv')))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
by = collections.defaultdict(list)
for r in tr:
    by[r['Queue_Id']].append(r)
dur = lambda r: int(r['End_Timestamp']) - int(r['Start_Timestamp'])
main = max(by.values(), key=lambda rs: sum(dur(r) for r in rs))
idx = [i for i, r in enumerate(main) if 'adam_kernel' in r['Kernel_Name']]
step = main[idx[-2] + 1:idx[-1] + 1]
fam = collections.defaultdict(float)
for r in step:
    n = r['Kernel_Name']
    n = (n.split('(anonymous namespace)::')[1] if '(anonymous namespace)::' in n else n).split('<')[0].split('(')[0]
    fam[n] += dur(r) / 1e3
print('main stream, last step, by kernel family (us):')
for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:16]:
    print('  %-34s %8.1f' % (k, v))
PY
cp $OUT/t_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/train32_prof_kernel_stats.csv
