# main-stream timeline of one HRNet18_rev1 training step (4 x 512x512): busy / gaps, and how much of the stream's time sits in launches that
# leave most of the chip idle (few workgroups, short) - what branch-parallel streams could hide.   bash tools/hrnet_timeline.sh   (through gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/hr_tl
mkdir -p $O
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
for _ in range(6):
    tr.train_step(*batch)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --output-format csv -d $O/raw -o t -- python3 /tmp/hr_train.py > /dev/null 2>&1
python3 $R/tools/step_timeline.py $O/raw/t_kernel_trace.csv > $O/hrnet_timeline.txt
python3 - <<'PY' >> $O/hrnet_timeline.txt
import csv, os, collections
rows = list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/hr_tl/raw/t_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
by = collections.defaultdict(list)
for r in rows:
    by[r['Queue_Id']].append(r)
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
main = max(by.values(), key=lambda rs: sum(dur(r) for r in rs))
idx = [i for i, r in enumerate(main) if 'adam_kernel' in r['Kernel_Name']]
step = main[idx[-2] + 1:idx[-1] + 1]
wgs = lambda r: int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // (int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z']))
for lim in (256, 512, 1024):
    sel = [r for r in step if wgs(r) <= lim]
    print('main-stream launches with <= %4d workgroups: %4d launches, %7.1f us' % (lim, len(sel), sum(dur(r) for r in sel)))
for lim in (10, 20, 40):
    sel = [r for r in step if dur(r) <= lim]
    print('main-stream launches of <= %2d us: %4d launches, %7.1f us' % (lim, len(sel), sum(dur(r) for r in sel)))
PY
rm -rf $O/raw
cat $O/hrnet_timeline.txt
