set -x
O=$PWD/gpurun_out
: > $O/train_config3_steps.txt
for mode in "" "CRNERF_TRAIN_FWD_X3=1 CRNERF_WGRAD_BF16X3=1" "CRNERF_WGRAD_BF16X3=1" "CRNERF_WGRAD_BF16=1" "CRNERF_TRAIN_BF16=1" "CRNERF_TRAIN_BF16=1 CRNERF_TRAIN_RECOMPUTE=1"; do
  for r in 1024 16384 65536; do
    echo -n "[$mode] " >> $O/train_config3_steps.txt
    env $mode python tools/train_config4_bench.py $r 2>&1 | tail -1 >> $O/train_config3_steps.txt
  done
done
cat $O/train_config3_steps.txt
export CRNERF_TRAIN_BF16=1
bash tools/train_step_trace.sh $O/train_mixed_fused_16384 grid_batch $PWD/tools/train_config4_bench.py 16384 > $O/train_mixed_fused_16384.log 2>&1
head -3 $O/train_mixed_fused_16384.log
bash tools/train_step_trace.sh $O/train_mixed_fused_1024 grid_batch $PWD/tools/train_config4_bench.py 1024 > $O/train_mixed_fused_1024.log 2>&1
unset CRNERF_TRAIN_BF16
bash tools/kstats.sh $O/mixed_fwd_bench_kstats.txt $PWD/tools/mixed_fwd_bench.py > $O/mixed_fwd_bench.log 2>&1
bash tools/kstats.sh $O/mlp_train_bench_kstats.txt $PWD/tools/mlp_train_bench.py > $O/mlp_train_bench.log 2>&1
python tools/train_step_bench.py 2>&1 | tail -3
CRNERF_TRAIN_BF16=1 python tools/train_step_bench.py 2>&1 | tail -3
