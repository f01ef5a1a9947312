cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_full
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_inference.py tests/test_gpu_graphs.py tests/test_gpu_label_gate.py -q -m gpu 2>&1 | tail -8 | tee $O/tests_inf.txt
cd /tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/cu_clk -o t -- python3 $R/tools/cu_partition.py 16 solo > /dev/null 2>&1
python3 $R/tools/cu_partition.py --clocks $O/cu_clk | tee $O/cu_partition_clocks.txt
rm -rf $O/cu_clk
# inference step per kernel: fused post-processing on the side stream
rocprofv3 --kernel-trace --stats --output-format csv -d $O/infer_prof -o t -- python3 $R/bench.py --mode infer --dtype bf16 --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $O/infer_bf16_line.json 2>/dev/null
python3 - <<'PY' | tee $O/infer_bf16_kernels.txt
import csv, os
rows = list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r6_full/infer_prof/t_kernel_stats.csv')))
for r in rows[:28]:
    print('%6.2f%% %6d %9.1f  %s' % (float(r['Percentage']), int(r['Calls']), float(r['AverageNs']) / 1e3, r['Name'][:120]))
PY
rm -f $O/infer_prof/t_kernel_trace.csv
