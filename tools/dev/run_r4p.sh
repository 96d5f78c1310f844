#!/bin/bash
# round 4: GAT d = 64 projection on the matrix instruction; unrolled LDS dot products of the QC message kernels; dA of a message chain in one pass
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_gat_heads.py -q > gpurun_out/r4p_tests.log 2>&1; echo "heads rc=$?"
tail -6 gpurun_out/r4p_tests.log
timeout -k 10 400 python -m pytest tests/test_gpu_gat_qc.py -x -q > gpurun_out/r4p_tests2.log 2>&1; rc=$?
tail -3 gpurun_out/r4p_tests2.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/gat_bench.py 8:64:rk4 1:16:rk4 1:64:rk4 > gpurun_out/r4p_gat.log 2>&1; tail -3 gpurun_out/r4p_gat.log
timeout -k 10 500 python tools/config_bench.py C4 --no-cpu > gpurun_out/r4p_c4.log 2>&1; grep -E "ms_per_step|C4_" gpurun_out/r4p_c4.log
bash tools/dev/qc_prof.sh EdgeGCN_K_Sum --prepared > gpurun_out/r4p_qcprof.log 2>&1; grep -E "edge_|total kernel" gpurun_out/r4p_qcprof.log | cut -c1-150
