#!/bin/bash
# development aid: the queue of the fused kernel changes no result -- static pairs (grid >= pairs), the chip-sized
# grid and a tiny grid (every half refills many times) give bit-identical dumps
set -e
mkdir -p gpurun_out
RMPC_FUSED_GRID=100000 timeout -k 10 400 python tests/tools/ab_dump.py gpurun_out/ab_static.npz > gpurun_out/ab_static.log 2>&1 || { tail -5 gpurun_out/ab_static.log; exit 1; }
timeout -k 10 400 python tests/tools/ab_dump.py gpurun_out/ab_queue.npz > gpurun_out/ab_queue.log 2>&1 || { tail -5 gpurun_out/ab_queue.log; exit 1; }
RMPC_FUSED_GRID=48 timeout -k 10 400 python tests/tools/ab_dump.py gpurun_out/ab_q48.npz > gpurun_out/ab_q48.log 2>&1 || { tail -5 gpurun_out/ab_q48.log; exit 1; }
python tests/tools/ab_compare.py gpurun_out/ab_static.npz gpurun_out/ab_queue.npz
python tests/tools/ab_compare.py gpurun_out/ab_static.npz gpurun_out/ab_q48.npz
