#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_partition.py tests/test_gpu_bench.py tests/test_gpu_variants.py tests/test_gpu_harness.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r4s_tests.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r4s_tests.log
