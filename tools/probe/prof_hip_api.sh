# kernel trace + HIP API trace of the training step: is the main stream starved by the host in the deep layers?
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --hip-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/api_prof -o t -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --steps 6 --warmup 3 --no-extras --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/api_prof.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/api_prof/
python3 - <<'PY'
import csv, os, collections
d = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/api_prof/'
k = list(csv.DictReader(open(d + 't_kernel_trace.csv')))
a = list(csv.DictReader(open(d + 't_hip_api_trace.csv')))
print(a[0].keys())
api = {r['Correlation_Id']: r for r in a if 'Launch' in r['Function']}
k.sort(key=lambda r: int(r['Start_Timestamp']))
by = collections.defaultdict(list)
for r in k: by[r['Queue_Id']].append(r)
main = max(by.values(), key=lambda rs: sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs))
idx = [i for i, r in enumerate(main) if 'adam_kernel' in r['Kernel_Name']]
step = main[idx[-2] + 1:idx[-1] + 1]
t0 = int(step[0]['Start_Timestamp'])
prev = None
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev) / 1e3 if prev else 0
    ar = api.get(r['Correlation_Id'])
    if gap > 12 and ar:
        print('kernel at %7.1f us gap %5.1f: launch call issued at %7.1f us (ended %7.1f), previous kernel ended %7.1f   %s' % (
            (s - t0) / 1e3, gap, (int(ar['Start_Timestamp']) - t0) / 1e3, (int(ar['End_Timestamp']) - t0) / 1e3, (prev - t0) / 1e3, r['Kernel_Name'][25:60]))
    prev = e
# host lead over the GPU along the step: for every 10th main-stream kernel, kernel start minus launch-call time
for i, r in enumerate(step):
    ar = api.get(r['Correlation_Id'])
    if ar and i % 12 == 0:
        print('launch %3d: host lead %8.1f us' % (i, (int(r['Start_Timestamp']) - int(ar['End_Timestamp'])) / 1e3))
PY
