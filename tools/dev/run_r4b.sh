#!/bin/bash
# round 4, GPU call 3: pgemm v2 (split-K, read pipelining), bench --gpus 2 rehearsal, new tests
cd "$(dirname "$0")/../.."
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gcn.py -x -q -m gpu -k "piece_gemm or three_way_cut or tiled_gemm or transition_mlp or sparse_features or adam" > gpurun_out/t3.log 2>&1; echo rc=$? >> gpurun_out/t3.log
python tools/dev/pgemm_bench.py > gpurun_out/pgemm_bench2.log 2>&1; echo rc=$? >> gpurun_out/pgemm_bench2.log
python bench.py --gpus 2 --scale 12 --edges 50000 --steps 2 --warmup 1 --no-cpu-baseline --qc-steps 3 > gpurun_out/bench_g2.json 2> gpurun_out/bench_g2.err; echo rc=$? >> gpurun_out/bench_g2.err
python -m pytest tests/test_gpu_gat_qc.py tests/test_gpu_harness.py -x -q -m gpu -k "qc or c4" > gpurun_out/t3b.log 2>&1; echo rc=$? >> gpurun_out/t3b.log
tail -4 gpurun_out/t3.log; tail -12 gpurun_out/pgemm_bench2.log; tail -3 gpurun_out/t3b.log; tail -3 gpurun_out/bench_g2.err
