#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_gcn.py -x -q -k "pubmed or fused_small or small_feval or native_dopri5 or citeseer_and_pubmed" > gpurun_out/r4t_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r4t_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/pubmed_bench.py > gpurun_out/r4t_pubmed.log 2>&1; tail -2 gpurun_out/r4t_pubmed.log
bash tools/dev/pubmed_prof.sh > gpurun_out/pubmedprof.log 2>&1; head -6 gpurun_out/pubmed_kernel_stats.txt | cut -c1-150; grep ms_per_step gpurun_out/prof_pubmed/run.log | cut -c1-200
