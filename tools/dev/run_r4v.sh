#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout -k 10 200 python tools/dev/target_index_probe.py > gpurun_out/r4v_probe.log 2>&1; tail -4 gpurun_out/r4v_probe.log
timeout -k 10 500 python tools/config_bench.py C4 --no-cpu > gpurun_out/r4v_c4.log 2>&1; grep -E "ms_per_step|C4_" gpurun_out/r4v_c4.log
timeout -k 10 500 python tools/config_bench.py C4 --no-cpu > gpurun_out/r4v_c4b.log 2>&1; grep -E "ms_per_step|C4_" gpurun_out/r4v_c4b.log
