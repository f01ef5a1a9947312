ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/refresh_hr; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
bash $ROOT/tools/prof_hrnet_train.sh > $OUT/hrnet_train_b4_512_summary.txt 2>&1
cp $ROOT/gpurun_out/hrnet_train_prof/t_kernel_stats.csv $OUT/hrnet_train_b4_512_kernel_stats.csv
rm -rf $ROOT/gpurun_out/hrnet_train_prof
bash $ROOT/tools/prof_hrnet_traffic.sh bf16 > $OUT/hrnet_train_b4_512_traffic.txt 2>&1
rm -rf $ROOT/gpurun_out/hr_pmc_*
head -3 $OUT/hrnet_train_b4_512_summary.txt; head -1 $OUT/hrnet_train_b4_512_traffic.txt
