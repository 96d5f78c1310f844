#!/bin/bash
# final: the contract bench line (default flags), then the round's profile evidence, part A (bench kernel stats + PMC traffic)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout -k 10 900 python bench.py > gpurun_out/final_bench2.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/final_bench2.log | cut -c1-200
bash tools/profile_round.sh r04 > gpurun_out/profile_round_r04.log 2>&1
tail -12 gpurun_out/profile_round_r04.log
ls gpurun_out/profiles_r04/
