#!/bin/bash
# the whole GPU suite in one process, log to gpurun_out/full_r04.log (the last step after the last binary rebuild)
cd "$(dirname "$0")/../.."
python -m pytest tests/ -x -q -m gpu > gpurun_out/full_r04.log 2>&1; echo rc=$? >> gpurun_out/full_r04.log
tail -6 gpurun_out/full_r04.log
