#!/usr/bin/env python3
"""gode_gemm_f32 against the library GEMM on the three large products of the QC edge encoder (E x 2667 x 5329), for a
760-edge and an 800-edge batch; same process, interleaved, medians."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graph_odenet_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    hid, out = 2667, 5329
    for E in (760, 800):
        H, W2, dA = torch.randn(E, hid, device=dev), torch.randn(hid, out, device=dev) / 50, torch.randn(E, out, device=dev)
        cases = {
            "fwd  H W2        (NN, M=%d N=%d K=%d)" % (E, out, hid): (lambda: ops.gemm(H, W2), lambda: torch.mm(H, W2)),
            "dH   dA W2^T     (NT, M=%d N=%d K=%d)" % (E, hid, out): (lambda: ops.gemm(dA, W2, trans_b=True, mask=H), lambda: torch.mm(dA, W2.t())),
            "dW2  H^T dA      (TN, M=%d N=%d K=%d)" % (hid, out, E): (lambda: ops.gemm(H, dA, trans_a=True), lambda: torch.mm(H.t(), dA)),
        }
        res = {k: ([], []) for k in cases}
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for rep in range(6):
            for k, fns in cases.items():
                for i, fn in enumerate(fns):
                    fn()
                    ev[0].record()
                    for _ in range(3):
                        fn()
                    ev[1].record()
                    torch.cuda.synchronize()
                    if rep:
                        res[k][i].append(ev[0].elapsed_time(ev[1]) / 3)
        gf = 2.0 * E * hid * out / 1e9
        for k, (a, b) in res.items():
            ma, mb = sorted(a)[len(a) // 2], sorted(b)[len(b) // 2]
            print("%s  own %.3f ms (%.0f TFLOP/s)   library %.3f ms (%.0f TFLOP/s)" % (k, ma, gf / ma, mb, gf / mb), flush=True)


if __name__ == "__main__":
    main()
