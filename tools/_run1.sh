O=$GRAFT_REPO_ROOT/gpurun_out/r4h; mkdir -p $O; cd $GRAFT_REPO_ROOT
(python -m pytest tests -m gpu -q 2>&1 | grep -E "^FAILED|passed|failed") > $O/gpu_suite.txt
(CRNERF_TEST_COLD_L2=1 python -m pytest tests -m gpu -q -k "not multiproc and not fullsize and not converges" 2>&1 | grep -E "^FAILED|passed|failed") > $O/gpu_suite_cold_l2.txt
bash tools/train_step_trace.sh $O/train_config4_1024 grid_batch $GRAFT_REPO_ROOT/tools/train_config4_bench.py 1024 > $O/train_config4_1024.log 2>&1
bash tools/train_step_trace.sh $O/train_config4_65536 grid_batch $GRAFT_REPO_ROOT/tools/train_config4_bench.py 65536 > $O/train_config4_65536.log 2>&1
: > $O/steps.txt
for i in 1 2 3; do echo -n "[default, 40 steps] " >> $O/steps.txt; CRNERF_TRAIN_BENCH_STEPS=5,40 python tools/train_config4_bench.py 1024 2>&1 | tail -1 >> $O/steps.txt; done
for r in 16384 65536; do echo -n "[default] " >> $O/steps.txt; CRNERF_TRAIN_BENCH_STEPS=3,10 python tools/train_config4_bench.py $r 2>&1 | tail -1 >> $O/steps.txt; done
python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
cat $O/gpu_suite.txt $O/gpu_suite_cold_l2.txt $O/steps.txt; head -3 $O/train_config4_1024_kernels.txt
