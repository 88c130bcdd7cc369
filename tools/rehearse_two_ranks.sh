#!/bin/bash
# Rehearsal of bench.py's N > 1 path on a ONE-GPU box (run from the repo root through gpurun): two ranks share cuda:0 and
# exchange the gradient buckets through gloo (RCCL refuses two ranks on one device).  Everything but the RCCL transport runs
# as on the 8-GPU node: process group set-up from the torchrun environment, the bucketed all-reduce launched from inside
# backward on a side stream, the barrier / max-over-ranks timing, and the replica check in the JSON line
# ("replicas_identical_after_run": true -- each rank trains on its own shard, so identical parameters prove the exchange).
# The real multi-GPU numbers come from the driver's runs on the node; this only proves the code path.
set -o pipefail
KDCC_DIST_BACKEND=gloo KDCC_DIST_SHARE_GPU=1 timeout -k 10 400 \
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port "${1:-29513}" \
  bench.py --gpus 2 --steps 3 --warmup 1 --batch 1 --no-batch-sweep
