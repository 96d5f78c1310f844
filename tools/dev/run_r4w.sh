#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "adam or rectangular or tiled_gemm or first_nonzero or transition_mlp" > gpurun_out/r4w_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r4w_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python -m pytest tests/test_gpu_gat_qc.py tests/test_gpu_harness.py -x -q > gpurun_out/r4w_tests2.log 2>&1; rc=$?
tail -3 gpurun_out/r4w_tests2.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python tools/config_bench.py C4 --no-cpu > gpurun_out/r4w_c4.log 2>&1; grep -E "ms_per_step|C4_" gpurun_out/r4w_c4.log
timeout -k 10 500 python tools/config_bench.py C4 --no-cpu > gpurun_out/r4w_c4b.log 2>&1; grep -E "ms_per_step|C4_" gpurun_out/r4w_c4b.log
