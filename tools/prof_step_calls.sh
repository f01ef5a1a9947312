# every kernel call of ONE step in launch order (name, stream, duration, grid): bash tools/prof_step_calls.sh [fp32|bf16] [train|infer] [serial]  (through gpurun)
DT=${1:-fp32}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
if [ "$3" = serial ]; then    # the training step on ONE stream: every duration is the kernel's own
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/step_calls -o t -- python3 $R/tools/step_serial.py $DT 6 > /dev/null 2>&1
else
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/step_calls -o t -- python3 $R/bench.py --mode ${2:-train} --dtype $DT --no-extras --no-cpu-baseline --steps 6 --warmup 3 > /dev/null 2>&1
fi
python3 - <<'PY'
import csv, os, re
rows = list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/step_calls/t_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
mark = 'adam_kernel' if any('adam_kernel' in r['Kernel_Name'] for r in rows) else 'input_pack'      # (a step ends with Adam / an inference batch starts with the input pack)
adam = [i for i, r in enumerate(rows) if mark in r['Kernel_Name']]
seg = rows[adam[-3] + 1:adam[-2] + 1] if mark == 'adam_kernel' else rows[adam[-3]:adam[-2]]
t0 = int(seg[0]['Start_Timestamp'])
out = open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/step_calls.txt', 'w')
for r in seg:
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    n = n.split('(')[0][:70]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    out.write('%9.1f q%s %8.1f us  grid %6d x %d x %d  %s\n' % ((int(r['Start_Timestamp']) - t0) / 1e3, r['Queue_Id'], d,
              int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']), n))
out.close()
PY
rm -rf $R/gpurun_out/step_calls
