"""HIP-graph capture on this platform: one fact the product has to know.

Finding (round 2, tools/dev/memset_node_repro.py, memset_node_repro2.py; ROCm 7.2 / HIP 7.0.5 as shipped in this
image): a `hipMemsetAsync` captured into a HIP graph (a memset node) is NOT reliably ordered against the kernel nodes
around it when the graph is replayed - 199 of 200 replays of `memset(buf); buf += x; out += sum(buf)` give wrong sums, at
every size from 64 B to 1 MB.  PyTorch's own captured reductions are hit as well: a captured `tensor.sum()` that takes
the multi-block path (its semaphore buffer is cleared with cudaMemsetAsync) is wrong in 199 of 200 replays.  With the
HIP runtime's graph fast path switched off (environment variable DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, read when the HIP
runtime initialises) every replay is right.

Consequences here:
  * libgraphode launches no memset at all (tests/test_abi.py checks the sources), so captured solves of the fused
    fields (odeint._GraphedSolve: our kernels plus copies / fills / small single-block reductions) are safe either way;
  * captures that contain arbitrary PyTorch autograd (qc_step.CapturedQCStep) are only taken when `memset_nodes_ok()`
    says so: the variable reads 0 and a replay self-test passes - otherwise the step stays on the eager path.
`prefer_safe_graphs()` sets the variable when the HIP runtime has not been initialised yet.
"""
import ctypes
import os

import torch

_ok = None
ENV = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"


def prefer_safe_graphs():
    """Ask the HIP runtime for the graph path on which memset nodes work; only effective before the first HIP call of
    the process.  Returns True when the variable is (now) set to 0."""
    if os.environ.get(ENV) is None and not torch.cuda.is_initialized():
        os.environ[ENV] = "0"
    return os.environ.get(ENV) == "0"


def memset_nodes_ok(device=None):
    """True when this process may rely on replayed memset nodes: the variable must read 0 (the only setting under which
    every replay of every probe was right) AND a self-test must pass (once per process, ~1 ms: capture
    memset(buf); buf += 1; out += sum(buf)  ten times in a row, replay 20 times).  The self-test alone is NOT
    sufficient: on the fast path it fails in a fresh process but was seen to pass after other GPU work had run in the
    process (round 2, `pytest -k "captured_qc or fused_gru"` with the variable set to 1), i.e. the fast path's
    behaviour depends on process state and a toy graph cannot certify another graph."""
    global _ok
    if _ok is not None:
        return _ok
    if os.environ.get(ENV) != "0":
        _ok = False
        return _ok
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("memset_nodes_ok() must be called outside a capture")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    hip.hipMemsetAsync.restype = ctypes.c_int
    n = 4096
    buf = torch.full((n,), 3.0, device=dev)
    out = torch.zeros(1, device=dev, dtype=torch.float64)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(10):
            if hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), 0, n * 4, st) != 0:
                raise RuntimeError("hipMemsetAsync failed during the capture self-test")
            buf.add_(1.0)
            out.add_(buf.double().sum())
    good = True
    for _ in range(20):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        good = good and abs(float(out) - 10.0 * n) < 0.5
    _ok = good
    return _ok
