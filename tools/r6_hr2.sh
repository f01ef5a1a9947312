cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_hr
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_hrnet.py tests/test_gpu_hrnet_train.py -x -q -m gpu 2>&1 | tail -6 | tee $O/tests_hr.txt
cd /tmp
python3 $R/tools/bench_hrnet.py 4 2>&1 | grep -v amdgpu.ids | head -2 | tee $O/bench_hrnet2.txt
bash $R/tools/prof_hrnet_traffic.sh bf16 > $O/hrnet_traffic.txt 2>&1
rm -rf $R/gpurun_out/hr_pmc_*
head -45 $O/hrnet_traffic.txt
