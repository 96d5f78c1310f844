#!/bin/bash
# round 4: chained stage inputs of the launch-bound dopri5 step; QC shapes; QC step after the gradient-drop
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_gcn.py -x -q -k "small_feval or native_dopri5 or fused_small or pubmed or dopri5" > gpurun_out/r4l_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r4l_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/pubmed_bench.py > gpurun_out/r4l_pubmed.log 2>&1; tail -3 gpurun_out/r4l_pubmed.log
timeout -k 10 200 python tools/dev/qc_gemm_shapes.py EdgeGCN_K_Sum > gpurun_out/r4l_shapes.log 2>&1; tail -25 gpurun_out/r4l_shapes.log
timeout -k 10 200 python tools/qc_bench.py --model EdgeGCN_K_Sum --steps 100 --warmup 20 --prepared --no-cpu-baseline > gpurun_out/r4l_qc_edge.log 2>&1; tail -1 gpurun_out/r4l_qc_edge.log | cut -c1-200
timeout -k 10 200 python tools/qc_bench.py --model MPNN_ENN_K_Set2Set --steps 100 --warmup 20 --prepared --no-cpu-baseline > gpurun_out/r4l_qc_mpnn.log 2>&1; tail -1 gpurun_out/r4l_qc_mpnn.log | cut -c1-200
timeout -k 10 300 python -m pytest tests/test_gpu_gat_qc.py -x -q -k "qc or set2set or c4" > gpurun_out/r4l_qc_tests.log 2>&1; echo "qc tests rc=$?"; tail -3 gpurun_out/r4l_qc_tests.log
