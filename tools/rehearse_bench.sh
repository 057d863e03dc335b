#!/bin/bash
# Rehearsal of bench.py's multi-rank path on a ONE-GPU box (not a measurement).
#   A  2 ranks, gloo, collectives staged through host tensors
#   C  1 rank, RCCL ("nccl") group of size one forced down the data-parallel path: the first
#      execution of the RCCL branch on hardware, and the host enqueue time per epoch
set -u
mkdir -p gpurun_out
run() { python -m torch.distributed.run --nnodes=1 --nproc-per-node "$1" --master-addr 127.0.0.1 --master-port "$2" bench.py --gpus "$1" --steps 2 --warmup 1 --k-epochs 8 --no-cpu-baseline; }
echo "== A gloo host-staged"; PPO_BENCH_BACKEND=gloo PPO_BENCH_WATCHDOG=60 timeout -k 10 120 bash -c "$(declare -f run); run 2 29541" > gpurun_out/rehearse_A.txt 2>&1; echo "rc=$?" | tee -a gpurun_out/rehearse_A.txt
echo "== C nccl world=1 forced"; PPO_DIST_FORCE=1 PPO_BENCH_FORCE_GROUP=1 PPO_BENCH_WATCHDOG=60 timeout -k 10 120 python bench.py --gpus 1 --steps 2 --warmup 1 --k-epochs 80 --no-cpu-baseline > gpurun_out/rehearse_C.txt 2>&1; echo "rc=$?" | tee -a gpurun_out/rehearse_C.txt
exit 0
