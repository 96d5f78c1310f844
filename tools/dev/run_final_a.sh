#!/bin/bash
# final: smoke + the contract bench line (default flags)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/final_smoke.log
timeout -k 10 900 python bench.py > gpurun_out/final_bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/final_bench.log | cut -c1-300
