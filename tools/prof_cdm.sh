#!/bin/bash
# per-kernel times of the centripetal-direction-map generation (16 label images of 256x256, 60 nuclei):  bash tools/prof_cdm.sh  (through gpurun)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/cdm_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $ROOT/tools/run_cdm.py 20 > $OUT/run.log 2>&1
find $OUT -name 't_kernel_stats.csv' -exec cp {} $ROOT/gpurun_out/cdm_kernel_stats.csv \;
head -30 $ROOT/gpurun_out/cdm_kernel_stats.csv | cut -c1-160
rm -rf $OUT
