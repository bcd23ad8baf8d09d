#!/bin/bash
# GPU box: the batched weight-gradient launch of the 1,024-ray training step (wgrad_h2_batch_kernel) under values of one environment switch;
# average kernel time from rocprofv3 --kernel-trace --stats over 13 steps.   usage: tools/wgb_sweep.sh <ENV_VAR> <value> [<value> ...]
VAR=$1; shift
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6/wgb_sweep_$VAR.txt; : > $O
for w in "$@"; do
  rm -rf /tmp/kp
  env CRNERF_TRAIN_BENCH_STEPS=3,10 $VAR=$w rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o t -- python $GRAFT_REPO_ROOT/tools/train_config4_bench.py 1024 > /dev/null 2>&1
  echo "$VAR=$w: wgrad_h2_batch_kernel avg $(grep -h "wgrad_h2_batch_kernel" $(find /tmp/kp -name "*kernel_stats.csv") | cut -d, -f4) ns   wgrad_reduce_batch_kernel avg $(grep -h "wgrad_reduce_batch" $(find /tmp/kp -name "*kernel_stats.csv") | cut -d, -f4) ns" >> $O
done
cat $O
