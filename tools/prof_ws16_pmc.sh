#!/bin/bash
# Counters of conv_ws16_kernel on the dominant layer at 64 tiles (working set 1.07 GB: beyond the 256 MB Infinity Cache), write path included:
#   bash tools/prof_ws16_pmc.sh [tiles=64] [out-dir]        (through gpurun; separate --pmc passes, the program right after "--")
# variants: random / zero operands x full (debug 64) / no stores (64|8)
B=${1:-64}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${2:-$ROOT/gpurun_out/ws16_pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_avail.txt 2>&1
for V in "0 64 random_full" "1 64 zero_full" "0 72 random_nostore" "1 72 zero_nostore"; do
  set -- $V; Z=$1; D=$2; NAME=$3
  i=0
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM" \
           "TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCC_EA0_WR_UNCACHED_32B_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum" \
           "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${NAME}_p$i -o t -- python3 $ROOT/tools/run_ws16.py $B $Z $D 12 > $OUT/${NAME}_p$i.log 2>&1 || echo "$NAME pass $i ($C) failed" >> $OUT/failed.txt
  done
done
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections, json
out = sys.argv[1]
res = collections.OrderedDict()
for name in ('random_full', 'zero_full', 'random_nostore', 'zero_nostore'):
    vals = collections.OrderedDict()
    for d in sorted(glob.glob(os.path.join(out, name + '_p*'))):
        if not os.path.isdir(d):
            continue
        dur = {}
        for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                if 'conv_ws16_kernel' in r['Kernel_Name']:
                    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            per = collections.defaultdict(lambda: collections.defaultdict(float))
            for r in csv.DictReader(open(f)):
                if 'conv_ws16_kernel' in r['Kernel_Name']:
                    per[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
            for c, byd in per.items():
                ids = sorted(byd, key=int)[2:]                      # (skip the first two launches)
                vals[c] = sum(byd[i] for i in ids) / max(1, len(ids))
                if dur:
                    vals.setdefault('_us_under_' + c, sum(dur[i] for i in ids if i in dur) / max(1, len([i for i in ids if i in dur])))
    res[name] = vals
json.dump(res, open(os.path.join(out, 'summary.json'), 'w'), indent=1)
keys = []
for v in res.values():
    for k in v:
        if k not in keys and not k.startswith('_'):
            keys.append(k)
print('%-34s' % 'counter (mean per launch)' + ''.join('%18s' % n for n in res))
for k in keys:
    print('%-34s' % k + ''.join('%18.4g' % res[n].get(k, float('nan')) for n in res))
print('%-34s' % 'launch us (under FETCH_SIZE pass)' + ''.join('%18.1f' % res[n].get('_us_under_FETCH_SIZE', float('nan')) for n in res))
PY
rm -rf $OUT/*_p*/ 2>/dev/null
