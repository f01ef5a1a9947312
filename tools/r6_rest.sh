cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_full
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu --deselect tests/test_gpu_label_gate.py 2>&1 | tail -30 > $O/tests_all.txt
tail -8 $O/tests_all.txt
python3 tools/cu_partition.py 16 2>&1 | grep -v amdgpu.ids | tee $O/cu_partition.txt
