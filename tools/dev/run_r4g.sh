#!/bin/bash
cd "$(dirname "$0")/../.."
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "one_pass or vjp_from_exact or weight_gradient_from_exact" > gpurun_out/t7.log 2>&1; echo rc=$? >> gpurun_out/t7.log
tail -15 gpurun_out/t7.log
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_gcn.py -x -q -m gpu -k "fullsize or full_size or native_rk4 or renumbered" > gpurun_out/t7b.log 2>&1; echo rc=$? >> gpurun_out/t7b.log
tail -5 gpurun_out/t7b.log
GODE_BWD_WGRAD=1 python bench.py --no-configs --no-cpu-baseline --steps 3 > gpurun_out/bench_bw1.json 2> gpurun_out/bench_bw1.err
GODE_BWD_WGRAD=0 python bench.py --no-configs --no-cpu-baseline --steps 3 > gpurun_out/bench_bw0.json 2> gpurun_out/bench_bw0.err
python - <<'PY'
import json
for k in ("bw1", "bw0"):
    try:
        d = json.loads([l for l in open("gpurun_out/bench_%s.json" % k) if l.startswith("{")][-1])
        print(k, d["ms_per_step"], d["value"], d["loss"], {n: v["avg_launch_ms"] for n, v in d["roofline_dense"]["families"].items()})
    except Exception as e:
        print(k, "failed", e)
PY
