#!/bin/bash
# GPU box: the round-6 evidence in one call -> gpurun_out/r6e/ (copied into profiles/r6/ afterwards).
# usage: bash tools/r6_evidence.sh
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6e
mkdir -p $O
cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_f32.json 2> $O/bench_f32.err      # the driver's command line
(python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep -E "^FAILED|passed|failed") > $O/gpu_suite.txt
(CRNERF_TEST_COLD_L2=1 python -m pytest tests -m gpu -q -k "not multiproc and not fullsize and not converges" 2>&1 | grep -E "^FAILED|passed|failed") > $O/gpu_suite_cold_l2.txt
python bench.py --precision f32h2 --no-cpu-baseline > $O/bench_f32h2.json 2> /dev/null
python bench.py --precision bf16 --no-cpu-baseline > $O/bench_bf16.json 2> /dev/null
python bench.py --scaling strong --workload configs2 --steps 3 --warmup 1 > $O/strong_configs2_n1.json 2> /dev/null
python bench.py --scaling strong --workload configs3 --steps 4 --warmup 2 > $O/strong_configs3_n1.json 2> /dev/null
python bench.py --scaling strong --workload configs3 --train-precision f32 --steps 3 --warmup 1 > $O/strong_configs3_f32_n1.json 2> /dev/null
# one rank over RCCL (the driver's launcher with --nproc-per-node 1): the collectives of both strong workloads timed on the final tree
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_rccl_1rank.json 2> $O/rccl.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 1 --scaling strong --workload configs2 --steps 3 --warmup 1 > $O/strong_configs2_rccl_1rank.json 2>> $O/rccl.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 1 --scaling strong --workload configs3 --steps 3 --warmup 1 > $O/strong_configs3_rccl_1rank.json 2>> $O/rccl.err
: > $O/train_config3_steps.txt
for mode in "CRNERF_TRAIN_FWD=f32 CRNERF_WGRAD_F32=1" "" "CRNERF_TRAIN_CHUNK_POINTS=1048576 CRNERF_BRANCH_STREAMS=0" "CRNERF_TRAIN_BF16=1" "CRNERF_TRAIN_RECOMPUTE=1"; do
  for r in 1024 16384 65536; do
    echo -n "[$mode] " >> $O/train_config3_steps.txt
    env $mode CRNERF_TRAIN_BENCH_STEPS=3,10 python tools/train_config4_bench.py $r 2>&1 | tail -1 >> $O/train_config3_steps.txt
  done
done
for i in 1 2 3; do for c in 0 1; do echo -n "[default, 60 steps, CRNERF_PIN_HOST=$c] " >> $O/train_config3_steps.txt; CRNERF_PIN_HOST=$c CRNERF_TRAIN_BENCH_STEPS=10,60 python tools/train_config4_bench.py 1024 2>&1 | grep "config-4" >> $O/train_config3_steps.txt; done; done
for mode in "CRNERF_TRAIN_FWD=f32 CRNERF_WGRAD_F32=1" "CRNERF_TRAIN_BF16=1" "CRNERF_TRAIN_RECOMPUTE=1" "CRNERF_BRANCH_STREAMS=0"; do
  echo -n "[$mode CRNERF_PIN_HOST=1] " >> $O/train_config3_steps.txt; env $mode CRNERF_PIN_HOST=1 CRNERF_TRAIN_BENCH_STEPS=10,40 python tools/train_config4_bench.py 1024 2>&1 | grep "config-4" >> $O/train_config3_steps.txt
done
CRNERF_PIN_HOST=1 bash tools/train_step_trace.sh $O/train_config4_1024 grid_batch $GRAFT_REPO_ROOT/tools/train_config4_bench.py 1024 > $O/train_config4_1024.log 2>&1
bash tools/train_step_trace.sh $O/train_config4_65536 grid_batch $GRAFT_REPO_ROOT/tools/train_config4_bench.py 65536 > $O/train_config4_65536.log 2>&1
bash tools/hbm_profile.sh mlp_train_bench.py r6e/hbm_train > /dev/null 2>&1
bash tools/profile.sh r6e/prof_f32 > /dev/null 2>&1
bash tools/profile.sh r6e/prof_h2 --precision f32h2 > /dev/null 2>&1
cat $O/gpu_suite.txt $O/gpu_suite_cold_l2.txt $O/train_config3_steps.txt
head -c 400 $O/bench_f32.json
