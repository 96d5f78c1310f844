#!/bin/bash
cd "$(dirname "$0")/../.."
python tools/dev/pgemm_bench.py > gpurun_out/pgemm_bench3.log 2>&1; echo rc=$? >> gpurun_out/pgemm_bench3.log
python - > gpurun_out/c4b.json 2> gpurun_out/c4b.err <<'PY'
import json, sys, torch
sys.path.insert(0, "tools")
import config_bench as cb
d = torch.device("cuda:0")
print(json.dumps({"edge": cb.c4_qc(d, "EdgeGCN_K_Sum", cpu=False), "mpnn": cb.c4_qc(d, "MPNN_ENN_K_Set2Set", cpu=False)}, indent=1))
PY
tail -9 gpurun_out/pgemm_bench3.log; grep ms_per gpurun_out/c4b.json; tail -3 gpurun_out/c4b.err
