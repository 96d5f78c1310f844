#!/bin/bash
# round 4: message kernel with flat LDS tiles; C4 numbers after the gradient drop
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_gat_qc.py -x -q -k "qc or edge or mpnn or c4" > gpurun_out/r4m_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r4m_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python tools/config_bench.py C4 --no-cpu > gpurun_out/r4m_c4.log 2>&1; tail -12 gpurun_out/r4m_c4.log | cut -c1-400
bash tools/dev/qc_prof.sh EdgeGCN_K_Sum --prepared --no-cpu-baseline > gpurun_out/r4m_qcprof.log 2>&1; head -40 gpurun_out/r4m_qcprof.log | cut -c1-150
