#!/bin/bash
# round 4, GPU call 2: pgemm parity + timing, QC two-rank rehearsal with a traceback
cd "$(dirname "$0")/../.."
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "piece_gemm or three_way_cut or tiled_gemm or transition_mlp" > gpurun_out/t2.log 2>&1; echo rc=$? >> gpurun_out/t2.log
HIP_LAUNCH_BLOCKING=1 AMD_SERIALIZE_KERNEL=3 python tools/qc_bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/qc2.log 2>&1; echo rc=$? >> gpurun_out/qc2.log
python tools/dev/pgemm_bench.py > gpurun_out/pgemm_bench.log 2>&1; echo rc=$? >> gpurun_out/pgemm_bench.log
tail -4 gpurun_out/t2.log; tail -12 gpurun_out/pgemm_bench.log
