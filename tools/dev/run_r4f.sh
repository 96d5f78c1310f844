#!/bin/bash
cd "$(dirname "$0")/../.."
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gat_qc.py tests/test_gpu_harness.py -x -q -m gpu -k "set2set or qc or c4 or segment or lstm or assignment" > gpurun_out/t6.log 2>&1; echo rc=$? >> gpurun_out/t6.log
python - > gpurun_out/c4d.json 2> gpurun_out/c4d.err <<'PY'
import json, sys, torch
sys.path.insert(0, "tools")
import config_bench as cb
d = torch.device("cuda:0")
print(json.dumps({"edge": cb.c4_qc(d, "EdgeGCN_K_Sum", cpu=False), "mpnn": cb.c4_qc(d, "MPNN_ENN_K_Set2Set", cpu=False)}, indent=1))
PY
bash tools/dev/qc_prof.sh MPNN_ENN_K_Set2Set --prepared > gpurun_out/qcprof_mpnn3.log 2>&1
tail -5 gpurun_out/t6.log; grep ms_per gpurun_out/c4d.json; tail -3 gpurun_out/c4d.err; head -12 gpurun_out/profiles_qc/MPNN_ENN_K_Set2Set_kernel_stats.txt
