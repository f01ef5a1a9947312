cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_img
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests/test_gpu_postproc.py tests/test_gpu_inference.py tests/test_gpu_tile_postproc.py tests/test_gpu_cdm.py tests/test_gpu_watershed.py -x -q -m gpu 2>&1 | tail -6 | tee $O/tests.txt
cd /tmp
cat > /tmp/imgpp.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import torch
from cdnet_amd import postproc, synth
dev = torch.device('cuda:0')
probs, points, dcms = synth.postproc_case(1000, 1000, 500, 5)
t = lambda a: torch.from_numpy(a).to(dev)[None]
pr, po, dc = t(probs), t(points), t(dcms)
for _ in range(5):
    r = postproc.postprocess_views(pr, po, dc)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    r = postproc.postprocess_views(pr, po, dc, check=False)
torch.cuda.synchronize()
print('image_postproc 1000x1000 x 8 views: %.3f ms per image, %d nuclei' % ((time.perf_counter() - t0) / 30 * 1e3, int(r['counts'][0])))
PY
python3 /tmp/imgpp.py 2>&1 | grep -v amdgpu | tee $O/imgpp.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python3 /tmp/imgpp.py > /dev/null 2>&1
python3 - <<'PY' | tee -a $O/imgpp.txt
import csv, os
rows = list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r6_img/prof/t_kernel_stats.csv')))
for r in rows[:26]:
    print('%6.2f%% %6d %9.1f  %s' % (float(r['Percentage']), int(r['Calls']), float(r['AverageNs']) / 1e3, r['Name'][:110]))
PY
rm -f $O/prof/t_kernel_trace.csv
