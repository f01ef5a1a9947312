#!/bin/bash
# PMC passes over the dominant-kernel measurement `bench.py --mode roofline --dtype <fp32|bf16>` (through gpurun):
#   bash tools/prof_roofline_pmc.sh fp32 [out-dir]
# Separate rocprofv3 --pmc passes with --kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; no
# --stats / sys-trace beside --pmc), the program itself right after "--".  tools/make_roofline_pmc.py folds them into one JSON.
DT=${1:-fp32}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${2:-$ROOT/gpurun_out/roofline_pmc_$DT}
B=${3:-16}                      # tiles per launch (64: the inference batch, beyond the Infinity Cache)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o t -- python3 $ROOT/bench.py --mode roofline --dtype $DT --batch $B --steps 40 > $OUT/roofline_line.json 2> $OUT/stats.err
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc$i -o t -- python3 $ROOT/bench.py --mode roofline --dtype $DT --batch $B --steps 40 > $OUT/pmc$i.log 2>&1 || echo "pass $i ($C) failed" >> $OUT/failed.txt
done
python3 $ROOT/tools/make_roofline_pmc.py $OUT $DT $B > $OUT/summary.json
cat $OUT/summary.json
