#!/usr/bin/env python3
"""Same-process A/B of the dense kernel variants at 2^20 x 128 (box-to-box spread is +-10 %, so variants are timed
interleaved in ONE process: median of `--reps` rounds, each round timing every variant once)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graph_odenet_amd import _lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=7)
    a = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    n, d = 1 << 20, 128
    X, K1, K2, K3, Y, OUT, XO, PRE = [torch.randn(n, d, device=dev) for _ in range(8)]
    W = torch.randn(d + 1, d, device=dev) / d ** 0.5
    gam, bet = torch.rand(d, device=dev) + 0.5, torch.rand(d, device=dev) - 0.5
    T = [(1.0, X), (0.1, K1), (-0.1, K2), (0.1, K3)]
    cases = {
        "fwd 1": lambda: ops.gn_time_gemm(T[:1], n, d, 32, 1e-5, gam, bet, W, True, 0.3, out=OUT),
        "fwd 2": lambda: ops.gn_time_gemm(T[:2], n, d, 32, 1e-5, gam, bet, W, True, 0.3, out=OUT),
        "fwd 3": lambda: ops.gn_time_gemm(T[:3], n, d, 32, 1e-5, gam, bet, W, True, 0.3, out=OUT),
        "fwd 4": lambda: ops.gn_time_gemm(T, n, d, 32, 1e-5, gam, bet, W, True, 0.3, out=OUT),
        "fwd 4+xout": lambda: ops.gn_time_gemm(T, n, d, 32, 1e-5, gam, bet, W, True, 0.3, out=OUT, x_out=XO),
        "bwd 1": lambda: ops.gn_time_gemm_bwd(T[:1], n, d, 32, 1e-5, gam, W, True, Y, out=OUT),
        "bwd 2": lambda: ops.gn_time_gemm_bwd(T[:2], n, d, 32, 1e-5, gam, W, True, Y, out=OUT),
        "bwd 1+pre4": lambda: ops.gn_time_gemm_bwd(T[:1], n, d, 32, 1e-5, gam, W, True, Y, out=OUT, out_scale=0.1,
                                                   pre_terms=[(1.0, PRE), (0.1, K1), (0.2, K2), (0.3, K3)]),
        "wgrad 1": lambda: ops.wgrad(T[:1], n, d, 32, 1e-5, gam, bet, Y, True),
    }
    variants = {"round 2": dict(fwd_pc=0, bwd_pc=0), "producer/consumer": dict(fwd_pc=3, bwd_pc=1), "round 2 again": dict(fwd_pc=0, bwd_pc=0)}
    res = {c: {v: [] for v in variants} for c in cases}
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for rep in range(a.reps + 1):
        for cname, fn in cases.items():
            for vname, opts in variants.items():
                for k, v in opts.items():
                    assert lib.gode_set_option(k.encode(), v) == 0
                fn()
                ev[0].record()
                for _ in range(5):
                    fn()
                ev[1].record()
                torch.cuda.synchronize()
                if rep:
                    res[cname][vname].append(ev[0].elapsed_time(ev[1]) / 5)
    print("%-12s" % "case" + "".join("%16s" % v for v in variants))
    for c in cases:
        print("%-12s" % c + "".join("%13.4f ms" % sorted(res[c][v])[len(res[c][v]) // 2] for v in variants))


if __name__ == "__main__":
    main()
