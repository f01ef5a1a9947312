# HBM bytes and stand-alone durations per kernel of one HRNet18_rev1 training step (4 tiles of 512x512), from two PMC passes (FETCH_SIZE x 2
# on gfx950, WRITE_SIZE; the profiler serialises the launches, so the durations are each kernel's own):
#   bash tools/prof_hrnet_traffic.sh [bf16|fp32]      (through gpurun)
DT=${1:-bf16}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/hr_train.py <<'PY'
import os, sys
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import torch
from cdnet_amd import trainer, runtime
from cdnet_amd.models.dam.seg_hrnet_rev1 import HighResolutionNet
class O:
    model = {'out_c': 3}
runtime.set_precision(sys.argv[1])
torch.manual_seed(0)
m = HighResolutionNet(O()).cuda().train()
tr = trainer.Trainer(m)
batch = trainer.synthetic_batch(4, torch.device('cuda:0'), seed=5, H=512, W=512)
for _ in range(4):
    tr.train_step(*batch)
torch.cuda.synchronize()
PY
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/hr_pmc_$c -o t -- python3 /tmp/hr_train.py $DT > $GRAFT_REPO_ROOT/gpurun_out/hr_pmc_$c.log 2>&1
done
python3 - <<'PY'
import csv, os, collections, json, re
root = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/'
per = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
tot = [0.0, 0.0]
for ci, c in enumerate(('FETCH_SIZE', 'WRITE_SIZE')):
    by, name, grid = collections.defaultdict(float), {}, {}
    for r in csv.DictReader(open(root + 'hr_pmc_%s/t_counter_collection.csv' % c)):
        if r['Counter_Name'] == c:
            i = int(r['Dispatch_Id'])
            by[i] += float(r['Counter_Value'])
            name[i] = r['Kernel_Name']
            grid[i] = int(r['Grid_Size']) // max(1, int(r['Workgroup_Size']))
    dur = {}
    for r in csv.DictReader(open(root + 'hr_pmc_%s/t_kernel_trace.csv' % c)):
        dur[int(r['Dispatch_Id'])] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    ids = sorted(by)
    adam = [i for i in ids if 'adam_kernel' in name[i]]
    lo, hi = adam[-2], adam[-1]
    for i in ids:
        if lo < i <= hi:
            n = re.sub(r'\(anonymous namespace\)::', '', name[i])
            n = re.sub(r'^void ', '', n).split('(')[0][:52]
            k = (n, grid[i])
            b = by[i] * (2048 if ci == 0 else 1024)
            per[k][2 + ci] += b
            tot[ci] += b
            if ci == 0:
                per[k][0] += 1
                per[k][1] += dur.get(i, 0.0)
print(json.dumps({'hbm_read_GB_per_step': tot[0] / 1e9, 'hbm_write_GB_per_step': tot[1] / 1e9, 'hbm_GB_per_step': (tot[0] + tot[1]) / 1e9,
                  'standalone_kernel_ms_per_step': sum(v[1] for v in per.values()) / 1e3, 'launches': sum(v[0] for v in per.values())}))
print('%-52s %6s %4s %9s %8s %8s %7s' % ('kernel', 'wgs', 'n', 'total us', 'read MB', 'write MB', 'GB/s'))
for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:70]:
    print('%-52s %6d %4d %9.1f %8.1f %8.1f %7.0f' % (k[0], k[1], v[0], v[1], v[2] / 1e6, v[3] / 1e6, (v[2] + v[3]) / 1e3 / max(v[1], 1e-9)))
PY
