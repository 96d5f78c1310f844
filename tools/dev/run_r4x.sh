#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_gcn.py -x -q -k "small_feval or native_dopri5 or fused_small or on_cora" > gpurun_out/r4x_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r4x_tests.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2; do timeout -k 10 300 python tools/config_bench.py C1 C2 --no-cpu > gpurun_out/r4x_c12_$i.log 2>&1; grep -E "ms_per_step|\"C" gpurun_out/r4x_c12_$i.log; done
