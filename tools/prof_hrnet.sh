# rocprofv3 kernel statistics of the HRNet18_rev1 inference forward (4 tiles of 512x512): bash tools/prof_hrnet.sh  (through gpurun)
cd /tmp && export TMPDIR=/tmp
cat > /tmp/hr_inf.py <<'PY'
import os, sys
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import torch
from cdnet_amd.models.dam.seg_hrnet_rev1 import HighResolutionNet
class O:
    model = {'out_c': 3}
torch.manual_seed(0)
m = HighResolutionNet(O()).cuda().eval()
x = torch.rand((4, 3, 512, 512), device='cuda')
with torch.no_grad():
    for _ in range(10):
        m(x)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/hrnet_prof -o t -- python3 /tmp/hr_inf.py > $GRAFT_REPO_ROOT/gpurun_out/hrnet_prof.log 2>&1
python3 - <<'PY'
import csv, os
rows = list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/hrnet_prof/t_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
calls = sum(int(r['Calls']) for r in rows)
print('total busy ms per forward %.2f, launches per forward %.0f' % (tot / 1e6 / 10, calls / 10))
for r in rows[:25]:
    print('%6.2f%% %6d %9.1f  %s' % (float(r['Percentage']), int(r['Calls']) / 10, float(r['AverageNs']) / 1e3, r['Name'][:120]))
PY
