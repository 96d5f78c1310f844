#!/usr/bin/env python3
"""Is the difference between the one-launch and the multi-launch dense half of the GAT ODE function (d = 64: two channels
per GroupNorm group) rounding noise amplified by GroupNorm's conditioning, or an error?  Both paths against the float64
oracle on the same inputs: one evaluation, and the rk4 solve.  Development aid (uses oracle/ as the checker)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_odenet_amd import _lib, gat_models
from graph_odenet_amd.models import ODEBlock
from oracle import layers_ref as R, solver_ref as S
lib = _lib.load()
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "citeseer_gat_edges.npz")))
n = int(g["n"]); src = torch.from_numpy(g["src"].astype(np.int64)); tgt = torch.from_numpy(g["tgt"].astype(np.int64))
E = src.numel()
Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(E)]), torch.ones(E), (n, E))
D = torch.device("cuda:0")
for d in (16, 64):
    x0 = torch.randn(n, d, generator=torch.Generator().manual_seed(3)) * 0.5
    for steps in (1, 4):
        out = {}
        for fused in (0, 1):
            lib.gode_set_option(b"small_fused", fused)
            torch.manual_seed(9)
            blk = ODEBlock(gat_models.ODEfunc(d), method="rk4", step_size=1.0 / steps)
            sd = {k: v.clone() for k, v in blk.state_dict().items()}
            blk = blk.to(D)
            with torch.no_grad():
                out[fused] = blk(x0.to(D), src.to(D), tgt.to(D), Mtgt.to(D)).cpu().double()
        P = {k: v.double() for k, v in sd.items()}
        f64 = lambda t, x: R.gat_odefunc(t, x, src, tgt, Mtgt.double(), P["odefunc.norm1.weight"], P["odefunc.norm1.bias"],
                                         P["odefunc.gc1.f.weight"], P["odefunc.gc1.f.bias"], P["odefunc.gc1.w.weight"], P["odefunc.gc1.w.bias"])
        with torch.no_grad():
            ref = S.odeint(f64, x0.double(), torch.tensor([0., 1.], dtype=torch.float64), method="rk4", options={"step_size": 1.0 / steps})[1]
        sc = ref.abs().max().item()
        print("d=%d rk4 steps=%d: |multi-launch - fp64| = %.2e   |one-launch - fp64| = %.2e   |one - multi| = %.2e   (max |y| %.2f)" % (
            d, steps, (out[0] - ref).abs().max().item(), (out[1] - ref).abs().max().item(), (out[1] - out[0]).abs().max().item(), sc), flush=True)
lib.gode_set_option(b"small_fused", 1)
