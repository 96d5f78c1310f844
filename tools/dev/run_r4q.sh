#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout -k 10 200 python tools/dev/edge_path_ab.py > gpurun_out/r4q_edge_ab.log 2>&1; cat gpurun_out/r4q_edge_ab.log | tail -5
timeout -k 10 300 python -m pytest tests/test_gpu_gat_heads.py -x -q -k "dense_half_matches" > gpurun_out/r4q_tests.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r4q_tests.log
