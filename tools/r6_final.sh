cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/refresh_r06
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/refresh_r06/gpu_suite.txt
tail -3 gpurun_out/refresh_r06/gpu_suite.txt
bash tools/refresh_r06.sh 2>&1 | tail -40
