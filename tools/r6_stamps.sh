cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_pp
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_tile_postproc.py -x -q -m gpu 2>&1 | tail -30 | tee $O/tests2.txt
CDNET_LIB_PATH=$R/cdnet_amd/libcdnet_hip_tstamps.so python3 tools/tile_stamps.py 64 2>&1 | grep -v amdgpu.ids | tee $O/stamps.txt
