#!/bin/bash
# round 4: GRU chain + double-buffered GRU backward, message step as block per edge + SpMM from 256 edges
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_gat_qc.py -x -q > gpurun_out/r4r_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r4r_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python tools/config_bench.py C4 --no-cpu > gpurun_out/r4r_c4.log 2>&1; grep -E "ms_per_step|C4_" gpurun_out/r4r_c4.log
bash tools/dev/qc_prof.sh MPNN_ENN_K_Set2Set --prepared > gpurun_out/r4r_qcprof.log 2>&1; head -26 gpurun_out/r4r_qcprof.log | cut -c1-150
