#!/bin/bash
# round 4: equal-share work division of the piece GEMM - parity, then the edge-count sweep
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "piece_gemm or transition_mlp or tiled_gemm or spmm" > gpurun_out/r4j_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r4j_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/dev/pgemm_bench.py --sweep > gpurun_out/pgemm_sweep.log 2>&1 && cat gpurun_out/pgemm_sweep.log &&
timeout -k 10 200 python tools/dev/pgemm_bench.py > gpurun_out/pgemm_bench5.log 2>&1; tail -12 gpurun_out/pgemm_bench5.log
timeout -k 10 400 python -m pytest tests/test_gpu_gat_heads.py -x -q > gpurun_out/r4j_gat.log 2>&1; echo "gat_heads rc=$?"; tail -5 gpurun_out/r4j_gat.log
timeout -k 10 200 python tools/gat_bench.py 8:64:rk4 1:16:rk4 1:64:rk4 > gpurun_out/r4j_gat8.log 2>&1; tail -4 gpurun_out/r4j_gat8.log
timeout -k 10 500 python -m pytest tests/test_gpu_gcn.py tests/test_gpu_variants.py -x -q > gpurun_out/r4j_gcn.log 2>&1; echo "gcn rc=$?"; tail -5 gpurun_out/r4j_gcn.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs > gpurun_out/r4j_bench.log 2>&1; tail -1 gpurun_out/r4j_bench.log | cut -c1-600
