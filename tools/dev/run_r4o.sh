#!/bin/bash
# round 4: message kernel v3, dopri5 step closes its stages in one launch
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_gat_qc.py -x -q -k "qc or edge or mpnn or c4" > gpurun_out/r4o_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r4o_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -m pytest tests/test_gpu_gcn.py -x -q -k "native_dopri5 or fused_small or pubmed or dopri5" > gpurun_out/r4o_tests2.log 2>&1; rc=$?
tail -3 gpurun_out/r4o_tests2.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/pubmed_bench.py > gpurun_out/r4o_pubmed.log 2>&1; tail -2 gpurun_out/r4o_pubmed.log
timeout -k 10 500 python tools/config_bench.py C4 --no-cpu > gpurun_out/r4o_c4.log 2>&1; grep -E "ms_per_step|C4_" gpurun_out/r4o_c4.log
bash tools/dev/qc_prof.sh MPNN_ENN_K_Set2Set --prepared > gpurun_out/r4o_qcprof.log 2>&1; head -30 gpurun_out/r4o_qcprof.log | cut -c1-150
