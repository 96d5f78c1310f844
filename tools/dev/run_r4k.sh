#!/bin/bash
# round 4: block policy of the piece GEMM, assign_csr rank scan, y2_colsum option A/B, QC shapes
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "piece_gemm or assignment_csr or spmm_leaves" > gpurun_out/r4k_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r4k_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/dev/pgemm_bench.py --sweep > gpurun_out/pgemm_sweep2.log 2>&1 && cat gpurun_out/pgemm_sweep2.log &&
timeout -k 10 300 python -m pytest tests/test_gpu_fullsize.py -x -q -k "adjoint_bias or semigroup" > gpurun_out/r4k_full.log 2>&1; echo "fullsize rc=$?"; tail -3 gpurun_out/r4k_full.log
timeout -k 10 200 python tools/dev/qc_gemm_shapes.py EdgeGCN_K_Sum > gpurun_out/r4k_shapes.log 2>&1; tail -25 gpurun_out/r4k_shapes.log
timeout -k 10 400 python bench.py --no-cpu-baseline --no-configs > gpurun_out/r4k_bench.log 2>&1; python - <<'PY'
import json
for line in open("gpurun_out/r4k_bench.log"):
    if line.startswith("{"):
        d = json.loads(line)
        print(d["ms_per_step"], {k: v for k, v in d.get("secondary", {}).items() if k.startswith("steps_per_s")})
PY
timeout -k 10 200 python tools/qc_bench.py --model EdgeGCN_K_Sum --steps 100 --warmup 20 --prepared > gpurun_out/r4k_qc_edge.log 2>&1; tail -1 gpurun_out/r4k_qc_edge.log | cut -c1-300
