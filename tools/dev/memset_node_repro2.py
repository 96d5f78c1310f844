"""Follow-up to memset_node_repro.py: (1) the same chain with sizes 64 B .. 1 MB; (2) torch's own reductions that zero
a semaphore buffer with cudaMemsetAsync (global reduce: sum of a large tensor to a scalar, argmax over a long strided
dimension) captured and replayed with changing inputs."""
import ctypes
import os
import sys

import torch

dev = torch.device("cuda:0")
print("env DEBUG_CLR_GRAPH_PACKET_CAPTURE =", os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE"))
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int
for numel in (16, 1024, 26624, 262144):
    buf = torch.full((numel,), 3.0, device=dev)
    x = torch.ones(numel, device=dev)
    out = torch.zeros(1, device=dev, dtype=torch.float64)
    pad = torch.zeros(1 << 16, device=dev)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for i in range(100):
            pad.add_(1.0)
            if i % 10 == 5:
                assert hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), 0, buf.numel() * 4, st) == 0
                buf.add_(x)
                out.add_(buf.double().sum())
    bad = 0
    for it in range(200):
        out.zero_(); x.fill_(float(it + 1)); g.replay(); torch.cuda.synchronize()
        want = float(it + 1) * numel * 10
        if abs(float(out) - want) > 1e-6 * want:
            bad += 1
    print("memset node of %8d bytes: %3d bad replays of 200" % (numel * 4, bad), flush=True)

# torch reductions under capture
big = torch.randn(1 << 24, device=dev)
mat = torch.zeros(448, 896, device=dev)
s_out = torch.zeros((), device=dev)
a_out = torch.zeros(896, dtype=torch.int64, device=dev)
c_out = torch.zeros(73, device=dev)
rows = torch.randn(448, 73, device=dev)
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(g):
    for _ in range(5):
        s_out.copy_(big.sum())
        a_out.copy_((mat != 0).to(torch.uint8).argmax(0))
        c_out.copy_(rows.sum(0))
bad = [0, 0, 0]
gen = torch.Generator(device=dev).manual_seed(0)
for it in range(200):
    big.normal_(generator=gen)
    rows.normal_(generator=gen)
    tgt = torch.randint(0, 448, (896,), device=dev, generator=gen)
    mat.zero_(); mat[tgt, torch.arange(896, device=dev)] = 1.0
    g.replay(); torch.cuda.synchronize()
    if abs(float(s_out) - float(big.double().sum())) > 1e-2 + 1e-4 * abs(float(big.double().sum())):
        bad[0] += 1
    if not torch.equal(a_out, tgt):
        bad[1] += 1
    if (c_out - rows.sum(0)).abs().max() > 1e-3:
        bad[2] += 1
print("captured torch reductions, bad replays of 200: sum(16M) %d, argmax(448x896, dim 0) %d, sum(448x73, dim 0) %d" % tuple(bad))
