"""Root-cause hunt for the non-finite loss of tests/test_gpu_gat_heads.py::test_eight_head_model_trains_on_citeseer_edges
(round-1 driver run: 7 finite Adam steps, then NaN).  One process = one run of the same training loop with per-step
diagnostics; flags switch the suspects on and off.

  --poison GB     allocate GB of NaN-filled device memory and free it first, so the caching allocator hands poisoned
                  blocks to every later torch.empty
  --patch-empty   torch.empty / empty_like / new_empty return NaN-filled float tensors (any read-before-write shows)
  --no-capture    odeint.GRAPH_CAPTURE_MAX_ELEMS = 0 (no HIP-graph replay)
  --no-overlap    gode_set_option("overlap", 0)
  --steps N       Adam steps (default 8, as the test)
  --nhid D --heads H --seed S
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--poison", type=float, default=0.0)
    ap.add_argument("--patch-empty", action="store_true")
    ap.add_argument("--no-capture", action="store_true")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--nhid", type=int, default=64)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--tag", default="")
    ap.add_argument("--lib", default="", help="A/B: load this build of libgraphode.so instead of the in-tree one")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.poison > 0:
        chunks = [torch.full((int(256 * 2**20 / 4),), float("nan"), device=dev) for _ in range(int(a.poison * 4))]
        small = [torch.full((sz,), float("nan"), device=dev) for sz in (64, 512, 4096, 65536, 1 << 20) for _ in range(32)]
        torch.cuda.synchronize()
        del chunks, small
    if a.patch_empty:
        nan = float("nan")
        o_empty, o_like = torch.empty, torch.empty_like

        def empty(*s, **k):
            t = o_empty(*s, **k)
            return t.fill_(nan) if t.is_floating_point() and t.is_cuda else t

        def empty_like(*s, **k):
            t = o_like(*s, **k)
            return t.fill_(nan) if t.is_floating_point() and t.is_cuda else t
        torch.empty, torch.empty_like = empty, empty_like
    from graph_odenet_amd import _lib, gat_heads, odeint
    if a.lib:
        _lib.LIB_PATH = os.path.abspath(a.lib)
    lib = _lib.load()
    if a.no_capture:
        odeint.GRAPH_CAPTURE_MAX_ELEMS = 0
    if a.no_overlap:
        lib.gode_set_option(b"overlap", 0)
    ge = dict(np.load(os.path.join(ROOT, "tests", "golden", "citeseer_gat_edges.npz")))
    n = int(ge["n"])
    T = lambda v: torch.from_numpy(np.asarray(v))          # noqa: E731
    src, tgt = T(ge["src"]).long().to(dev), T(ge["tgt"]).long().to(dev)
    Mtgt = torch.sparse_coo_tensor(torch.stack([T(ge["m_rows"]).long(), T(ge["m_cols"]).long()]), T(ge["m_vals"]),
                                   (n, src.numel())).to(dev)
    zoo = gat_heads.zoo(a.heads)
    torch.manual_seed(a.seed)
    m = zoo.ODEGCN3(nfeat=50, nhid=a.nhid, nclass=6, dropout=0.0, method="rk4", step_size=0.25).to(dev)
    x = torch.randn(n, 50, device=dev)
    y = torch.randint(0, 6, (n,), device=dev)
    opt = torch.optim.Adam(m.parameters(), lr=0.01)
    bad = False
    losses = []
    for step in range(a.steps):
        opt.zero_grad()
        out = m(x, src, tgt, Mtgt)
        loss = torch.nn.functional.nll_loss(out, y)
        loss.backward()
        lv = float(loss.detach())
        losses.append(lv)
        gmax, worst = 0.0, None
        for nm, p in m.named_parameters():
            if p.grad is None:
                continue
            fin = bool(torch.isfinite(p.grad).all())
            g = float(p.grad.abs().max()) if fin else float("inf")
            if g > gmax:
                gmax, worst = g, nm
            if not fin:
                k = int((~torch.isfinite(p.grad)).sum())
                print("  step %d: NON-FINITE grad in %s (%d of %d entries)" % (step, nm, k, p.grad.numel()))
                bad = True
        if a.verbose or not np.isfinite(lv) or bad:
            print("  step %d loss %.9g  max|grad| %.4g (%s)" % (step, lv, gmax, worst))
        if not np.isfinite(lv):
            bad = True
            print("  step %d: NON-FINITE loss; out finite: %s" % (step, bool(torch.isfinite(out).all())))
        if bad:
            break
        opt.step()
    torch.cuda.synchronize()
    print("%s %s losses %s" % (a.tag, "BAD" if bad else "ok", " ".join("%.7f" % v for v in losses)))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
