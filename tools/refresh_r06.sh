#!/bin/bash
# Round 6: every artifact of profiles/r06/ on ONE box, on the tree as it is (through gpurun from the repo root):
#   bash tools/refresh_r06.sh            -> gpurun_out/refresh_r06/<file names of profiles/r06/>
# = tools/refresh_profiles.sh r06 (bench line, per-kernel statistics, timelines, dominant-layer counters, step traffic, CDM, HRNet, label gate)
# + the round's additions: the post-processing chains alone (three-launch tile chain with its phase stamps, the per-step chain beside it, the
#   multi-view image chain per kernel), the CU-partition table with clocks, the serial per-call listings of the training step.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export GRAFT_REPO_ROOT=$ROOT
OUT=$ROOT/gpurun_out/refresh_r06
mkdir -p $OUT
bash $ROOT/tools/refresh_profiles.sh r06 > $OUT/refresh.log 2>&1
cd /tmp && export TMPDIR=/tmp
# post-processing of a 64-tile batch alone
python3 $ROOT/tools/bench_postproc.py 64 50 2>&1 | grep -v amdgpu.ids > $OUT/postproc_alone_64tiles.txt
POSTPROC_FUSED=0 python3 $ROOT/tools/bench_postproc.py 64 50 2>&1 | grep -v amdgpu.ids | sed 's/^/per-step chain of rounds 1-5 (POSTPROC_FUSED=0): /' >> $OUT/postproc_alone_64tiles.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw/pp -o t -- python3 $ROOT/tools/bench_postproc.py 64 20 > /dev/null 2>&1
python3 - <<'PY' >> $OUT/postproc_alone_64tiles.txt
import csv, os
rows = list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/refresh_r06/raw/pp/t_kernel_stats.csv')))
print('per kernel (rocprofv3 --kernel-trace --stats, 25 batches):')
for r in rows[:6]:
    print('%6.2f%% %6d %9.1f  %s' % (float(r['Percentage']), int(r['Calls']), float(r['AverageNs']) / 1e3, r['Name'][:110]))
PY
if [ -f $ROOT/cdnet_amd/libcdnet_hip_tstamps.so ]; then
  CDNET_LIB_PATH=$ROOT/cdnet_amd/libcdnet_hip_tstamps.so python3 $ROOT/tools/tile_stamps.py 64 2>&1 | grep -v amdgpu.ids > $OUT/tile_chain_stamps_64tiles.txt
fi
# post-processing of one 1000 x 1000 image with 8 views, per kernel
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
python3 /tmp/imgpp.py 2>&1 | grep -v amdgpu > $OUT/image_postproc_alone.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw/ipp -o t -- python3 /tmp/imgpp.py > /dev/null 2>&1
python3 - <<'PY' >> $OUT/image_postproc_alone.txt
import csv, os
rows = list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/refresh_r06/raw/ipp/t_kernel_stats.csv')))
for r in rows[:24]:
    print('%6.2f%% %6d %9.1f  %s' % (float(r['Percentage']), int(r['Calls']), float(r['AverageNs']) / 1e3, r['Name'][:110]))
PY
# CU-partitioned concurrency
python3 $ROOT/tools/cu_partition.py 16 2>&1 | grep -v amdgpu.ids > $OUT/cu_partition.txt
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/raw/cu_clk -o t -- python3 $ROOT/tools/cu_partition.py 16 solo > /dev/null 2>&1
python3 $ROOT/tools/cu_partition.py --clocks $OUT/raw/cu_clk >> $OUT/cu_partition.txt
# the training step on one stream, call by call
for DT in fp32 bf16; do
  bash $ROOT/tools/prof_step_calls.sh $DT train serial > /dev/null 2>&1
  cp $ROOT/gpurun_out/step_calls.txt $OUT/train_b16_${DT}_step_calls_serial.txt
done
# the four 1000 x 1000 gate draws on this tree
cd $ROOT
CDNET_GATE_DRAW_SIZE=1000 timeout 2400 python3 -m pytest tests/test_gpu_label_gate.py::test_second_draw -q -s -m gpu 2>&1 | grep -E "label gate draw|passed|failed" > $OUT/label_gate_1000x1000_draws.log
rm -rf $OUT/raw
du -sh $OUT; ls $OUT
