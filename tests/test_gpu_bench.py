"""bench.py the way the driver starts it: `python bench.py --gpus N ...` with no torch.distributed.run around it.

The ranks are started by bench.py itself (graph_odenet_amd/launch.py) as fresh child processes before any GPU call; on
this one-GPU box both ranks share the card and exchange over gloo (with a GPU per rank the same code runs RCCL).  The
command is started from a helper process that has never touched the GPU (tests/conftest.py starts the forkserver before
anything else): a process that has initialised the GPU starts no programs on this pool."""
import json
import multiprocessing as mp
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, timeout, q):
    try:
        cp = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
        q.put((cp.returncode, cp.stdout, cp.stderr[-3000:]))
    except BaseException as e:                       # a timeout included: surface it in the parent
        q.put((-1, "", repr(e)))


def _from_clean_process(cmd, timeout=900):
    try:
        ctx = mp.get_context("forkserver")
    except ValueError:
        pytest.skip("no forkserver start method")
    q = ctx.Queue()
    p = ctx.Process(target=_run, args=(cmd, timeout, q))
    p.start()
    rc, out, err = q.get(timeout=timeout + 60)
    p.join(60)
    return rc, out, err


def test_bench_starts_its_own_ranks_and_reports_both_multi_gpu_workloads():
    rc, out, err = _from_clean_process([sys.executable, "bench.py", "--gpus", "2", "--scale", "12", "--edges", "50000",
                                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--qc-steps", "3"])
    assert rc == 0, err
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out                                         # ONE JSON line, from rank 0
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["rccl_ranks"] == 2 and res["steps"] == 2 and res["scaling"] == "weak"
    assert res["value"] > 0 and res["backend"] in ("nccl", "gloo")
    sec = res["secondary"]
    qc, strong = sec["qc_data_parallel"], sec["strong_scaling"]
    assert "error" not in qc and "error" not in strong, sec
    assert qc["n_gpus"] == 2 and qc["unit"] == "graphs/s" and qc["value"] > 0
    assert qc["config"]["gradient_bytes_allreduced_per_step"] > 57e6    # the 14.4 M-parameter model's bucketed exchange
    assert strong["n_gpus"] == 2 and strong["scaling"] == "strong" and strong["value"] > 0
    assert strong["exchange"]["bytes_received_per_rank_per_step"] > 0 and strong["exchange"]["exchange_ms_per_step"] > 0


def test_single_gpu_line_keeps_the_contract_fields():
    rc, out, err = _from_clean_process([sys.executable, "bench.py", "--scale", "12", "--edges", "50000", "--steps", "2",
                                        "--warmup", "1", "--cpu-evals", "1", "--no-configs"])
    assert rc == 0, err
    res = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in res, k
    assert res["n_gpus"] == 1 and res["cpu_baseline"]["kind"] == "port" and res["cpu_baseline"]["cores"] >= 1
