# round 6: the fused tile post-processing chain - parity tests, then timing alone and per kernel (fused vs per-step)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_pp
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_tile_postproc.py tests/test_abi.py -x -q -m gpu 2>&1 | tail -15 > $O/tests.txt
cat $O/tests.txt
cd /tmp
python3 $R/tools/bench_postproc.py 64 50 2>&1 | tail -2 | tee $O/fused.txt
POSTPROC_FUSED=0 python3 $R/tools/bench_postproc.py 64 50 2>&1 | tail -2 | tee $O/perstep.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python3 $R/tools/bench_postproc.py 64 20 > /dev/null 2>&1
python3 - <<'PY' | tee -a $O/fused.txt
import csv, os
rows = list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r6_pp/prof/t_kernel_stats.csv')))
for r in rows[:12]:
    print('%6.2f%% %6d %9.1f  %s' % (float(r['Percentage']), int(r['Calls']), float(r['AverageNs']) / 1e3, r['Name'][:110]))
PY
rm -f $O/prof/t_kernel_trace.csv
python3 - <<'PY' | tee $O/box.txt
import os, sys, json
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import torch, bench
print(json.dumps(bench.box_calibration(torch, torch.device('cuda:0'))))
PY
