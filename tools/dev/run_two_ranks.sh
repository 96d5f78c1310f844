#!/bin/bash
# row (e) rehearsal on a one-GPU box: bench.py --gpus 2 starts its own two ranks, which share the GPU and exchange over gloo
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout -k 10 600 python bench.py --gpus 2 --scale 14 --edges 200000 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/two_ranks.log 2>&1; echo "rc=$?"
grep "^{" gpurun_out/two_ranks.log | tail -1 | cut -c1-3000
