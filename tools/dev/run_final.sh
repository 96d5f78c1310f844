#!/bin/bash
# final: the contract bench line (default flags), then the QC kernel-stats profiles with the final binaries
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/profiles_r04
timeout -k 10 900 python bench.py > gpurun_out/final_bench3.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/final_bench3.log | cut -c1-200
bash tools/dev/qc_prof.sh MPNN_ENN_K_Set2Set --prepared > gpurun_out/qcprof_a.log 2>&1
cp gpurun_out/profiles_qc/MPNN_ENN_K_Set2Set_kernel_stats.txt gpurun_out/profiles_r04/r04_qc_mpnn_kernel_stats.txt
bash tools/dev/qc_prof.sh EdgeGCN_K_Sum --prepared > gpurun_out/qcprof_b.log 2>&1
cp gpurun_out/profiles_qc/EdgeGCN_K_Sum_kernel_stats.txt gpurun_out/profiles_r04/r04_qc_edgegcn_kernel_stats.txt
head -3 gpurun_out/profiles_r04/r04_qc_edgegcn_kernel_stats.txt | cut -c1-120; head -3 gpurun_out/profiles_r04/r04_qc_mpnn_kernel_stats.txt | cut -c1-120
