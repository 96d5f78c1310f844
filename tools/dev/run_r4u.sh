#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "first_nonzero or assignment_csr" > gpurun_out/r4u_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r4u_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python -m pytest tests/test_gpu_gat_qc.py -x -q > gpurun_out/r4u_tests2.log 2>&1; rc=$?
tail -3 gpurun_out/r4u_tests2.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python tools/config_bench.py C4 --no-cpu > gpurun_out/r4u_c4.log 2>&1; grep -E "ms_per_step|C4_" gpurun_out/r4u_c4.log
