"""Does a hipMemsetAsync captured into a HIP graph (a memset node) keep its place between the kernels around it?
Mimics the pattern that produced non-finite GroupNorm gradients in captured adjoint solves (round 2): a long chain of
kernels, in it  memset(buf) ; kernel(buf += x)  ; kernel(out = sum(buf)), replayed with changing x."""
import ctypes
import sys

import torch

dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int
rows, d = 416, 64
buf = torch.full((rows, d), 3.0, device=dev)
x = torch.ones(rows, d, device=dev)
out = torch.zeros(d, device=dev)
pad = torch.zeros(1 << 16, device=dev)
n_chain = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(g):
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for i in range(n_chain):
        pad.add_(1.0)
        if i % 10 == 5:
            rc = hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), 0, buf.numel() * 4, st)
            assert rc == 0, rc
            buf.add_(x)
            out.add_(buf.sum(0))
bad = 0
n_sets = len([i for i in range(n_chain) if i % 10 == 5])
for it in range(300):
    out.zero_()
    x.fill_(float(it + 1))
    g.replay()
    torch.cuda.synchronize()
    want = float(it + 1) * rows * n_sets
    got = out.cpu()
    if not torch.isfinite(got).all() or (got - want).abs().max() > 1e-3 * want:
        bad += 1
        if bad <= 5:
            print("replay %d: want %.1f got min %.1f max %.1f" % (it, want, float(got.min()), float(got.max())))
print("memset-node repro: %d bad replays of 300 (chain %d, %d memset nodes)" % (bad, n_chain, n_sets))
