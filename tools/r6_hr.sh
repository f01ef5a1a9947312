cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_hr
mkdir -p $O
python3 $R/tools/bench_hrnet.py 4 2>&1 | grep -v amdgpu.ids | tee $O/bench_hrnet.txt
