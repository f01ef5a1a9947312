R=r05
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export GRAFT_REPO_ROOT=$ROOT
OUT=$ROOT/gpurun_out/refresh_train
RAW=$OUT/raw
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench_default_line.json 2> $RAW/bench_default.err
prof() {
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/$name -o t -- python3 $ROOT/bench.py "$@" --no-extras --no-cpu-baseline > $OUT/${name}_bench_line.json 2> $RAW/$name.err
  cp $RAW/$name/t_kernel_stats.csv $OUT/${name}_kernel_stats.csv
}
prof train_b16_fp32 --mode train --dtype fp32 --steps 10 --warmup 3
python3 $ROOT/tools/step_timeline.py $RAW/train_b16_fp32/t_kernel_trace.csv > $OUT/train_b16_fp32_timeline.txt 2>&1
prof train_b16_bf16 --mode train --dtype bf16 --steps 20 --warmup 5
python3 $ROOT/tools/step_timeline.py $RAW/train_b16_bf16/t_kernel_trace.csv > $OUT/train_b16_bf16_timeline.txt 2>&1
rm -rf $RAW/*/t_kernel_trace.csv
cd /tmp
for DT in fp32 bf16; do
  bash $ROOT/tools/prof_step_traffic.sh $DT > $OUT/train_b16_${DT}_step_traffic.txt 2>&1
done
rm -rf $ROOT/gpurun_out/step_pmc_* $RAW
ls $OUT
