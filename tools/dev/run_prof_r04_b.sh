#!/bin/bash
# round-4 evidence, part B: QC kernel stats (default mode), pgemm counters and timings, GAT kernel stats
cd "$(dirname "$0")/../.."
R=$(pwd)
mkdir -p gpurun_out/profiles_r04
bash tools/dev/qc_prof.sh MPNN_ENN_K_Set2Set --prepared > gpurun_out/qcprof_a.log 2>&1
cp gpurun_out/profiles_qc/MPNN_ENN_K_Set2Set_kernel_stats.txt gpurun_out/profiles_r04/r04_qc_mpnn_kernel_stats.txt
bash tools/dev/qc_prof.sh EdgeGCN_K_Sum --prepared > gpurun_out/qcprof_b.log 2>&1
cp gpurun_out/profiles_qc/EdgeGCN_K_Sum_kernel_stats.txt gpurun_out/profiles_r04/r04_qc_edgegcn_kernel_stats.txt
bash tools/dev/pgemm_pmc.sh > gpurun_out/pgemm_pmc.log 2>&1
python tools/dev/pgemm_bench.py 2>/dev/null | grep -v amdgpu > gpurun_out/profiles_r04/r04_pgemm_bench.txt
( echo "# python tools/dev/pgemm_bench.py --sweep on MI355X: the three edge-encoder products for edge counts of real mini-batches (piece kernel only; us and fp32-equivalent TFLOP/s)"; python tools/dev/pgemm_bench.py --sweep 2>/dev/null | grep -v amdgpu ) > gpurun_out/profiles_r04/r04_pgemm_sweep.txt
python tools/dev/spmm_colsum_probe.py 2>/dev/null | grep -v amdgpu > gpurun_out/profiles_r04/r04_spmm_colsum_probe.txt
bash tools/dev/pubmed_prof.sh > gpurun_out/pubmedprof.log 2>&1
cp gpurun_out/pubmed_kernel_stats.txt gpurun_out/profiles_r04/r04_pubmed_kernel_stats.txt; grep ms_per_step gpurun_out/prof_pubmed/run.log | cut -c1-300 >> gpurun_out/profiles_r04/r04_pubmed_kernel_stats.txt
bash tools/dev/gat_prof.sh 1:16:rk4 > gpurun_out/gatprof_a.log 2>&1
( echo "# ---- one head, hidden 16"; cat gpurun_out/gat_kernel_stats.txt; grep citeseer gpurun_out/prof_gat/run.log ) > gpurun_out/profiles_r04/r04_gat_citeseer_kernel_stats.txt
bash tools/dev/gat_prof.sh 8:64:rk4 > gpurun_out/gatprof_b.log 2>&1
( echo "# ---- eight heads, hidden 64"; cat gpurun_out/gat_kernel_stats.txt; grep citeseer gpurun_out/prof_gat/run.log ) >> gpurun_out/profiles_r04/r04_gat_citeseer_kernel_stats.txt
ls -la gpurun_out/profiles_r04/; head -8 gpurun_out/profiles_r04/r04_pgemm_pmc.txt; tail -12 gpurun_out/profiles_r04/r04_pgemm_bench.txt
