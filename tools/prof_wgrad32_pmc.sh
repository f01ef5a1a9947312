#!/bin/bash
# PMC passes over the fp32 weight-gradient kernel on the dominant layer (64 -> 64 @ 256 x 256 x 16), one set per ablation mode
# (CDNET_WGRAD_DEBUG: 0 full, 1 movers alone, 2 consumers alone):   bash tools/prof_wgrad32_pmc.sh   (through gpurun)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/wgrad32_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export WG32_ONLY=1
for D in 0 1 2; do
  export CDNET_WGRAD_DEBUG=$D
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/d${D}_stats -o t -- python3 $ROOT/tools/bench_wgrad32.py 16 40 > $OUT/d${D}_stats.log 2>&1
  i=0
  for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/d${D}_pmc$i -o t -- python3 $ROOT/tools/bench_wgrad32.py 16 40 > $OUT/d${D}_pmc$i.log 2>&1 || echo "mode $D pass $i ($C) failed" >> $OUT/failed.txt
  done
done
python3 - <<'PY'
import csv, glob, os, json
root = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/wgrad32_pmc'
for d in (0, 1, 2):
    out = {}
    for f in glob.glob('%s/d%d_stats/*kernel_stats.csv' % (root, d)):
        for r in csv.DictReader(open(f)):
            if 'wgrad_ws32_kernel' in r['Name']:
                out['avg_us'] = float(r['AverageNs']) / 1e3
    for f in sorted(glob.glob('%s/d%d_pmc*/*counter_collection.csv' % (root, d))):
        acc = {}
        for r in csv.DictReader(open(f)):
            if 'wgrad_ws32_kernel' in r['Kernel_Name']:
                acc.setdefault(r['Counter_Name'], {}).setdefault(r['Dispatch_Id'], 0.0)
                acc[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
        for n, per in acc.items():
            out[n] = sum(per.values()) / len(per)
    if 'GRBM_GUI_ACTIVE' in out and 'avg_us' in out:
        out['clock_mhz'] = out['GRBM_GUI_ACTIVE'] / 8.0 / out['avg_us']
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in out:
            out['mfma_busy_frac'] = out['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * out['GRBM_GUI_ACTIVE'] / 8.0)
    print('mode', d, json.dumps({k: round(v, 3) for k, v in out.items()}))
PY
rm -rf $OUT/*/*kernel_trace.csv
