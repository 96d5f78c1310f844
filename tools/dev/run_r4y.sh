#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "piece_gemm or transition_mlp or tiled_gemm" > gpurun_out/r4y_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r4y_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/dev/pgemm_bench.py --sweep > gpurun_out/pgemm_sweep3.log 2>&1; cat gpurun_out/pgemm_sweep3.log | grep -v amdgpu
