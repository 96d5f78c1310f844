#!/bin/bash
# round-4 evidence, part A: bench kernel stats + PMC traffic (tools/profile_round.sh)
cd "$(dirname "$0")/../.."
bash tools/profile_round.sh r04 > gpurun_out/profile_round_r04.log 2>&1
tail -30 gpurun_out/profile_round_r04.log
ls gpurun_out/profiles_r04/
