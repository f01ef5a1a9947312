#!/bin/bash
# Re-collect every artifact under profiles/<round>/ on the MI355X box (run through gpurun from the repo root):
#   bash tools/refresh_profiles.sh r02
# rocprofv3 runs from /tmp (TMPDIR=/tmp), the program itself follows "--"; PMC passes are separate runs without --stats traces.
R=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/refresh_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench_default_line.json 2> $OUT/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -o t -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/train_b16_bench_line.json 2> $OUT/train.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train32 -o t -- python3 $ROOT/bench.py --dtype fp32 --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $OUT/train_b16_fp32_bench_line.json 2> $OUT/train32.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/infer -o t -- python3 $ROOT/bench.py --mode infer --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $OUT/infer_b64_bench_line.json 2> $OUT/infer.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/infer32 -o t -- python3 $ROOT/bench.py --mode infer --dtype fp32 --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $OUT/infer_b64_fp32_bench_line.json 2> $OUT/infer32.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/image -o t -- python3 $ROOT/bench.py --mode image --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $OUT/image_1000_bench_line.json 2> $OUT/image.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dom -o t -- python3 $ROOT/bench.py --mode roofline > $OUT/dominant_conv_roofline_line.json 2> $OUT/dom.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dom32 -o t -- python3 $ROOT/bench.py --mode roofline --dtype fp32 > $OUT/dominant_conv_fp32_roofline_line.json 2> $OUT/dom32.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o t -- python3 $ROOT/bench.py --mode roofline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o t -- python3 $ROOT/bench.py --mode roofline > $OUT/pmc_write.log 2>&1
ls -R $OUT | head -80
