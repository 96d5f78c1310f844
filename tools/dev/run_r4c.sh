#!/bin/bash
# round 4, GPU call 4: one-node Set2Set, bench self-launch test again, QC profiles in the default (prepared) mode
cd "$(dirname "$0")/../.."
python -m pytest tests/test_gpu_gat_qc.py tests/test_gpu_harness.py tests/test_gpu_bench.py -x -q -m gpu -k "set2set or qc or c4 or segment or lstm or bench" > gpurun_out/t4.log 2>&1; echo rc=$? >> gpurun_out/t4.log
bash tools/dev/qc_prof.sh MPNN_ENN_K_Set2Set --prepared > gpurun_out/qcprof_mpnn.log 2>&1
bash tools/dev/qc_prof.sh EdgeGCN_K_Sum --prepared > gpurun_out/qcprof_edge.log 2>&1
python - > gpurun_out/c4.json 2> gpurun_out/c4.err <<'PY'
import json, sys, torch
sys.path.insert(0, "tools")
import config_bench as cb
d = torch.device("cuda:0")
print(json.dumps({"edge": cb.c4_qc(d, "EdgeGCN_K_Sum", cpu=False), "mpnn": cb.c4_qc(d, "MPNN_ENN_K_Set2Set", cpu=False)}, indent=1))
PY
tail -5 gpurun_out/t4.log; cat gpurun_out/c4.json | grep ms_per
