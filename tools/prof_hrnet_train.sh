# rocprofv3 kernel statistics of the HRNet18_rev1 training step (4 tiles of 512x512): bash tools/prof_hrnet_train.sh  (through gpurun)
cd /tmp && export TMPDIR=/tmp
cat > /tmp/hr_train.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import torch
from cdnet_amd.models.dam.seg_hrnet_rev1 import HighResolutionNet
from cdnet_amd import trainer
class O:
    model = {'out_c': 3}
torch.manual_seed(0)
m = HighResolutionNet(O()).cuda().train()
tr = trainer.Trainer(m)
batch = trainer.synthetic_batch(4, torch.device('cuda:0'), seed=5, H=512, W=512)
for _ in range(3):
    tr.train_step(*batch)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    tr.train_step(*batch)
torch.cuda.synchronize()
print('ms per step %.2f' % ((time.perf_counter() - t) * 100))
PY
python3 /tmp/hr_train.py 2>/dev/null | grep "ms per step" | sed "s/ms per step/ms per step (no profiler)/"
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/hrnet_train_prof -o t -- python3 /tmp/hr_train.py > $GRAFT_REPO_ROOT/gpurun_out/hrnet_train_prof.log 2>&1
grep "ms per step" $GRAFT_REPO_ROOT/gpurun_out/hrnet_train_prof.log | sed "s/ms per step/ms per step under rocprofv3 (its interception slows each of the ~1 600 launch calls)/"
python3 - <<'PY'
import csv, os
rows = list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/hrnet_train_prof/t_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
calls = sum(int(r['Calls']) for r in rows)
print('total busy ms per step %.2f, launches per step %.0f' % (tot / 1e6 / 13, calls / 13))
for r in rows[:45]:
    print('%6.2f%% %7.1f %9.1f  %s' % (float(r['Percentage']), int(r['Calls']) / 13, float(r['AverageNs']) / 1e3, r['Name'][:120]))
PY
