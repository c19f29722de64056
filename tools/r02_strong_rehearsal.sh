#!/bin/bash
# N>1 code path of bench.py --scaling strong rehearsed with 2 ranks sharing the one GPU of this box (gloo instead of RCCL)
cd "$GRAFT_REPO_ROOT"
P2E_DIST_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --steps 5 --warmup 2 --scaling strong --batch-log2 13 --no-limb-split --allgather-reps 1 2>&1 | grep -v amdgpu.ids | tail -3
P2E_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 \
  bench.py --gpus 2 --steps 5 --warmup 2 --batch-log2 12 --no-limb-split 2>&1 | grep -v amdgpu.ids | tail -2
