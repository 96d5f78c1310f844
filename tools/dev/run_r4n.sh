#!/bin/bash
# round 4: the small dense products of the QC step (prefetched rect_gemm, per-wave wgrad partials, split-K gemm)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "rectangular or tiled_gemm or transition_mlp" > gpurun_out/r4n_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r4n_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python -m pytest tests/test_gpu_gat_qc.py tests/test_gpu_gcn.py -x -q -k "qc or c4 or graph_convolution or gcn3 or depth" > gpurun_out/r4n_tests2.log 2>&1; rc=$?
tail -3 gpurun_out/r4n_tests2.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python tools/config_bench.py C4 --no-cpu > gpurun_out/r4n_c4.log 2>&1; grep -E "ms_per_step|C4_" gpurun_out/r4n_c4.log
bash tools/dev/qc_prof.sh EdgeGCN_K_Sum --prepared > gpurun_out/r4n_qcprof.log 2>&1; head -30 gpurun_out/r4n_qcprof.log | cut -c1-150
timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs > gpurun_out/r4n_bench.log 2>&1; tail -1 gpurun_out/r4n_bench.log | cut -c1-250
