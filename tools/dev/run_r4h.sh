#!/bin/bash
cd "$(dirname "$0")/../.."
python -m pytest tests/test_gpu_gat_qc.py tests/test_gpu_gat_heads.py -x -q -m gpu -k "gat or heads" > gpurun_out/t8.log 2>&1; echo rc=$? >> gpurun_out/t8.log
tail -5 gpurun_out/t8.log
python tools/gat_bench.py 1:16:rk4 8:64:rk4 1:16:dopri5 > gpurun_out/gat_bench_r04.log 2>&1
cat gpurun_out/gat_bench_r04.log | grep citeseer
python - <<'PY'
import sys
sys.argv = ["x", "1:16:rk4", "8:64:rk4"]
from graph_odenet_amd import gat_ode, gat_heads
gat_ode.GatOdeAdjointField.DEFER_SMALL = False
gat_heads.GatHeadsAdjointField.DEFER_SMALL = False
print("closing launch per STAGE (round 3):")
exec(compile(open("tools/gat_bench.py").read(), "tools/gat_bench.py", "exec"), {"__file__": "tools/gat_bench.py", "__name__": "__main__"})
PY
