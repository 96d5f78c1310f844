#!/usr/bin/env python3
"""Which dense products does a QC training step launch (csrc/mlp.hip gode_gemm_f32, csrc/rect.hip), and what does each
cost alone?  Records the shapes of three steps of a model on fresh batches, then times every distinct one.
usage: python tools/dev/qc_gemm_shapes.py [EdgeGCN_K_Sum|MPNN_ENN_K_Set2Set]"""
import collections
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from graph_odenet_amd import _lib, ops
import qc_bench

dev = torch.device("cuda:0")
model = sys.argv[1] if len(sys.argv) > 1 else "EdgeGCN_K_Sum"
net, batches = qc_bench.build(dev, model, 0, 4, 20)
lib = _lib.load()
seen = collections.Counter()
real = lib.gode_gemm_f32


def spy(ta, tb, M, N, K, *rest):
    seen[(int(ta), int(tb), int(M), int(N), int(K))] += 1
    return real(ta, tb, M, N, K, *rest)


lib.gode_gemm_f32 = spy
from graph_odenet_amd.optim import Adam
opt = Adam(net.parameters(), lr=1e-3)
for x, ef, Esrc, Etgt, batch, tgt in batches[:3]:
    opt.zero_grad(set_to_none=False)
    F.mse_loss(net(x, ef, Esrc, Etgt, batch), tgt).backward()
    opt.step()
torch.cuda.synchronize()
lib.gode_gemm_f32 = real
print("%s: gode_gemm_f32 calls in 3 steps (pgemm-sized products excluded)" % model)
for (ta, tb, M, N, K), c in sorted(seen.items(), key=lambda kv: -kv[1]):
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    out = torch.empty(M, N, device=dev)

    def call():
        _lib.check(real(ta, tb, M, N, K, _lib.ptr(A), A.stride(0), _lib.ptr(B), B.stride(0), _lib.ptr(out), out.stride(0), None, 0, None, 0,
                        _lib.stream_ptr()), "gemm")
    for _ in range(3):
        call()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize(); ev[0].record()
    for _ in range(50):
        call()
    ev[1].record(); torch.cuda.synchronize()
    print("  trans %d%d   M %5d  N %5d  K %5d   x%d per 3 steps   %6.1f us" % (ta, tb, M, N, K, c, ev[0].elapsed_time(ev[1]) / 50 * 1e3))
