#!/bin/bash
# Re-collect every artifact under profiles/<round>/ on the MI355X box (run through gpurun from the repo root):
#   bash tools/refresh_profiles.sh r03        -> gpurun_out/refresh_r03/<the file names of profiles/r03/>
# rocprofv3 runs from /tmp (TMPDIR=/tmp), the program itself follows "--"; PMC passes are separate runs without --stats traces
# (tools/prof_roofline_pmc.sh, tools/prof_step_traffic.sh).
R=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export GRAFT_REPO_ROOT=$ROOT
OUT=$ROOT/gpurun_out/refresh_$R
RAW=$OUT/raw
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench_default_line.json 2> $RAW/bench_default.err
prof() {   # prof <name> <bench.py args...>
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/$name -o t -- python3 $ROOT/bench.py "$@" --no-extras --no-cpu-baseline > $OUT/${name}_bench_line.json 2> $RAW/$name.err
  cp $RAW/$name/t_kernel_stats.csv $OUT/${name}_kernel_stats.csv
}
prof train_b16_fp32 --mode train --dtype fp32 --steps 10 --warmup 3
python3 $ROOT/tools/step_timeline.py $RAW/train_b16_fp32/t_kernel_trace.csv > $OUT/train_b16_fp32_timeline.txt 2>&1
prof train_b16_bf16 --mode train --dtype bf16 --steps 20 --warmup 5
python3 $ROOT/tools/step_timeline.py $RAW/train_b16_bf16/t_kernel_trace.csv > $OUT/train_b16_bf16_timeline.txt 2>&1
prof infer_b64_fp32 --mode infer --dtype fp32 --steps 5 --warmup 2
prof infer_b64_bf16 --mode infer --dtype bf16 --steps 5 --warmup 2
prof image_1000_fp32 --mode image --dtype fp32 --steps 3 --warmup 1
prof image_1000_bf16 --mode image --dtype bf16 --steps 3 --warmup 1
rm -rf $RAW/*/t_kernel_trace.csv
# the dominant kernel alone: --stats pass + the PMC passes, folded into dominant_conv_<dtype>_pmc.json
for V in "fp32 16" "bf16 16" "bf16 64"; do
  set -- $V; DT=$1; B=$2
  SUF=$([ "$B" = 16 ] && echo "" || echo "_${B}tiles")
  bash $ROOT/tools/prof_roofline_pmc.sh $DT $RAW/roofline_$DT$SUF $B > /dev/null 2>&1
  cp $RAW/roofline_$DT$SUF/summary.json $OUT/dominant_conv_${DT}${SUF}_pmc.json
  cp $RAW/roofline_$DT$SUF/roofline_line.json $OUT/dominant_conv_${DT}${SUF}_roofline_line.json
  cp $RAW/roofline_$DT$SUF/stats/t_kernel_stats.csv $OUT/dominant_conv_${DT}${SUF}_kernel_stats.csv
  rm -rf $RAW/roofline_$DT$SUF/pmc*/*kernel_trace.csv $RAW/roofline_$DT$SUF/stats/*kernel_trace.csv
done
cd /tmp
for DT in fp32 bf16; do
  bash $ROOT/tools/prof_step_traffic.sh $DT > $OUT/train_b16_${DT}_step_traffic.txt 2>&1
  bash $ROOT/tools/prof_step_traffic.sh $DT infer > $OUT/infer_b64_${DT}_step_traffic.txt 2>&1
done
bash $ROOT/tools/prof_cdm.sh > /dev/null 2>&1
cp $ROOT/gpurun_out/cdm_kernel_stats.csv $OUT/cdm_kernel_stats.csv
rm -rf $ROOT/gpurun_out/step_pmc_*
bash $ROOT/tools/prof_hrnet_train.sh > $OUT/hrnet_train_b4_512_summary.txt 2>&1
cp $ROOT/gpurun_out/hrnet_train_prof/t_kernel_stats.csv $OUT/hrnet_train_b4_512_kernel_stats.csv
rm -rf $ROOT/gpurun_out/hrnet_train_prof
bash $ROOT/tools/prof_hrnet_traffic.sh bf16 > $OUT/hrnet_train_b4_512_traffic.txt 2>&1
rm -rf $ROOT/gpurun_out/hr_pmc_*
cd $ROOT
python3 -m pytest tests/test_gpu_label_gate.py -q -s -m gpu 2>&1 | tail -60 > $OUT/label_gate.log
du -sh $OUT; ls $OUT
