#!/usr/bin/env python3
"""The three products of the QC edge encoder (760 x 2667 x 5329): exact-fp32 MFMA kernel (gode_gemm_f32) against the
bf16-piece kernel (gode_cut_bf16x3_f32 + gode_pgemm_bf16x3), same process, interleaved, HIP-event timed."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graph_odenet_amd import _lib, ops

D = torch.device("cuda:0")
lib = _lib.load()
E, Hd, O = 760, 2667, 5329
if "--sweep" in sys.argv:
    # edge counts of real mini-batches vary (650 .. 950 for 20 molecules): 6, 7 or 8 row tiles - 252 / 294 / 336 tiles of
    # the forward product on 256 CUs.  Piece kernel only, per product.
    def t_us(fn, n=20):
        for _ in range(3):
            fn()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        torch.cuda.synchronize(); ev[0].record()
        for _ in range(n):
            fn()
        ev[1].record(); torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]) / n * 1e3
    g = torch.Generator(device=D).manual_seed(0)
    W2 = torch.randn(Hd, O, device=D, generator=g) / Hd ** 0.5
    Wc = ops.cut3(W2)
    for e in (640, 700, 760, 768, 800, 850, 896, 950, 1024, 1500):
        H = torch.relu(torch.randn(e, Hd, device=D, generator=g)); dA = torch.randn(e, O, device=D, generator=g)
        Hc, dc = ops.cut3(H), ops.cut3(dA)
        t = (t_us(lambda: ops.pgemm(Hc, Wc)), t_us(lambda: ops.pgemm(dc, Wc, trans_b=True, mask=H)),
             t_us(lambda: ops.pgemm(Hc, dc, trans_a=True)))
        fl = 2.0 * e * Hd * O
        print("E %5d   H W2 %7.1f us (%5.1f)   dA W2^T %7.1f us (%5.1f)   H^T dA %7.1f us (%5.1f)   [fp32-equiv TFLOP/s]   ws bytes %s"
              % (e, t[0], fl / t[0] / 1e6, t[1], fl / t[1] / 1e6, t[2], fl / t[2] / 1e6,
                 [lib.gode_pgemm_workspace_bytes(e, O, Hd), lib.gode_pgemm_workspace_bytes(e, Hd, O),
                  lib.gode_pgemm_workspace_bytes(Hd, O, e)]), flush=True)
    sys.exit(0)
g = torch.Generator(device=D).manual_seed(0)
H = torch.relu(torch.randn(E, Hd, device=D, generator=g))
W2 = torch.randn(Hd, O, device=D, generator=g) / Hd ** 0.5
dA = torch.randn(E, O, device=D, generator=g)


def fp32(A, B, ta, tb, mask=None):
    M, K = (A.shape[1], A.shape[0]) if ta else A.shape
    N = B.shape[0] if tb else B.shape[1]
    out = torch.empty(M, N, device=D)
    _lib.check(lib.gode_gemm_f32(int(ta), int(tb), M, N, K, _lib.ptr(A), A.stride(0), _lib.ptr(B), B.stride(0), _lib.ptr(out),
                                 out.stride(0), None, 0, _lib.ptr(mask), mask.stride(0) if mask is not None else 0,
                                 _lib.stream_ptr()), "gemm")
    return out


def timed(fn, n=20):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n * 1e3


FLUSH = torch.empty(1 << 28, device=D)            # 1 GiB: four times the Infinity Cache


def timed_cold(fn, n=8):
    """every call behind a 2 GiB read-modify-write of another buffer: operands come from HBM, as inside a training step"""
    fn()
    tot = 0.0
    for _ in range(n):
        FLUSH.add_(1.0)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record(); fn(); ev[1].record()
        torch.cuda.synchronize()
        tot += ev[0].elapsed_time(ev[1])
    return tot / n * 1e3


Hc, Wc, dc = ops.cut3(H), ops.cut3(W2), ops.cut3(dA)
if "--once" in sys.argv:                 # counter passes: a few launches of each kernel, nothing timed
    for _ in range(3):
        Hc, Wc, dc = ops.cut3(H), ops.cut3(W2), ops.cut3(dA)
        ops.pgemm(Hc, Wc); ops.pgemm(dc, Wc, trans_b=True, mask=H); ops.pgemm(Hc, dc, trans_a=True)
        fp32(H, W2, False, False)
    torch.cuda.synchronize()
    sys.exit(0)
flop = 2.0 * E * Hd * O
rows = []
for name, f32fn, pfn in (
        ("H W2", lambda: fp32(H, W2, False, False), lambda p: ops.pgemm(Hc, Wc, products=p)),
        ("dA W2^T * mask", lambda: fp32(dA, W2, False, True, H), lambda p: ops.pgemm(dc, Wc, trans_b=True, mask=H, products=p)),
        ("H^T dA", lambda: fp32(H, dA, True, False), lambda p: ops.pgemm(Hc, dc, trans_a=True, products=p))):
    for rep in range(2):
        t32 = timed(f32fn)
        t8 = timed(lambda: pfn(8))
        t6 = timed(lambda: pfn(6))
        c32, c8 = timed_cold(f32fn), timed_cold(lambda: pfn(8))
        print("%-16s fp32-MFMA %7.1f us (%5.1f TFLOP/s)   pieces x8 %7.1f us (%5.1f fp32-equiv TFLOP/s, %4.2f of the bf16 peak)   x6 %7.1f us"
              "   | operands from HBM: fp32 %7.1f us, pieces x8 %7.1f us"
              % (name, t32, flop / t32 / 1e6, t8, flop / t8 / 1e6, 8 * flop / t8 / 1e6 / 2500.0, t6, c32, c8), flush=True)
for name, X in (("cut W2 (2667 x 5329)", W2), ("cut dA (760 x 5329)", dA), ("cut H (760 x 2667)", H)):
    t = timed(lambda: ops.cut3(X))
    byt = X.numel() * 4 + 6 * lib.gode_cut_pad(X.shape[0]) * lib.gode_cut_pad(X.shape[1])
    print("%-22s %7.1f us  (%.2f TB/s)" % (name, t, byt / t / 1e6), flush=True)
# error against float64 on a row block
ref = (H[:64].double() @ W2.double()).cpu()
for p in (8, 6):
    got = ops.pgemm(Hc, Wc, products=p)[:64].double().cpu()
    print("H W2 rows 0-63, %d products: max err %.2e of max|C| %.2e" % (p, (got - ref).abs().max().item(), ref.abs().max().item()))
got = fp32(H, W2, False, False)[:64].double().cpu()
print("fp32-MFMA kernel:            max err %.2e" % (got - ref).abs().max().item())
