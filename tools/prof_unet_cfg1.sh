# rocprofv3 kernel statistics of the plain UNet training step (BASELINE config 1 at its own size: 4 tiles of 256x256, fp32 mode):
#   bash tools/prof_unet_cfg1.sh   (through gpurun)
cd /tmp && export TMPDIR=/tmp
cat > /tmp/unet_cfg1.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import torch
import cdnet_amd
from cdnet_amd import trainer
from cdnet_amd.models.unet import UNet
cdnet_amd.set_precision(os.environ.get('PREC', 'fp32'))
torch.manual_seed(0)
dev = torch.device('cuda:0')
B = int(os.environ.get('B', '4'))
tr = trainer.UNetTrainer(UNet(num_classes=3).to(dev))
x, lab, _, _, weight = trainer.synthetic_batch(B, dev, seed=2022)
for _ in range(3):
    tr.train_step(x, lab, weight)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    tr.train_step(x, lab, weight)
torch.cuda.synchronize()
print('ms per step %.3f' % ((time.perf_counter() - t) * 100))
PY
python3 /tmp/unet_cfg1.py 2>/dev/null | grep "ms per step"
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/unet_cfg1_prof -o t -- python3 /tmp/unet_cfg1.py > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py $GRAFT_REPO_ROOT/gpurun_out/unet_cfg1_prof/t_kernel_trace.csv 2>&1 | head -5
python3 - <<'PY'
import csv, os
rows = list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/unet_cfg1_prof/t_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total busy ms per step %.2f' % (tot / 1e6 / 13))
for r in rows[:30]:
    print('%6.2f%% %7.1f %9.1f  %s' % (float(r['Percentage']), int(r['Calls']) / 13, float(r['AverageNs']) / 1e3, r['Name'][:110].replace('(anonymous namespace)::', '')))
PY
rm -f $GRAFT_REPO_ROOT/gpurun_out/unet_cfg1_prof/t_kernel_trace.csv
