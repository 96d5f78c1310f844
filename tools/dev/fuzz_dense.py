"""Randomised cross-check of the fused dense kernels (all three, fast / narrow / generic paths) against fp64 torch."""
import os, sys, itertools, random, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_odenet_amd import ops
dev = torch.device("cuda:0")
random.seed(0); torch.manual_seed(0)
bad = 0
cases = []
for d in (16, 32, 64, 128, 24, 73):
    for groups in {0, min(32, d), d} | ({d // 4} if d % 4 == 0 else set()):
        if groups and d % groups:
            continue
        for dout in {d, 2, 7}:
            cases.append((d, groups, dout))
for d, groups, dout in cases:
    for n in (1, 15, 16, 17, random.randint(100, 5000), random.randint(60000, 70000)):
        nt = random.randint(1, 5)
        has_time = random.random() < 0.8
        terms_c = [(1.0 if j == 0 else random.uniform(-0.5, 0.5), torch.randn(n, d, dtype=torch.float64)) for j in range(nt)]
        x = sum(c * t for c, t in terms_c).requires_grad_(True)
        gam = (torch.rand(d, dtype=torch.float64) + 0.5).requires_grad_(True); bet = (torch.rand(d, dtype=torch.float64) - 0.5).requires_grad_(True)
        W = (torch.randn(d + has_time, dout, dtype=torch.float64) / d ** 0.5).requires_grad_(True)
        tt = 0.37
        if n == 1 and groups == d:
            continue
        xn = F.group_norm(x, groups, gam, bet, 1e-5) if groups else x
        S = (torch.cat([torch.full((n, 1), tt, dtype=torch.float64), xn], 1) if has_time else xn) @ W
        dS = torch.randn(n, dout, dtype=torch.float64)
        S.backward(dS)
        terms = [(c, t.float().to(dev)) for c, t in terms_c]
        g_, b_ = (gam.detach().float().to(dev), bet.detach().float().to(dev)) if groups else (None, None)
        Wd, dSd = W.detach().float().to(dev), dS.float().to(dev)
        cg = d // groups if groups else 0
        tol = {0: 2e-5, 1: 5e-4, 2: 1e-4}.get(cg, 2e-5)
        def rel(a, b):
            return float((a.double().cpu() - b).abs().max() / max(1.0, float(b.abs().max())))
        e1 = rel(ops.gn_time_gemm(terms, n, d, groups, 1e-5, g_, b_, Wd, has_time, tt), S.detach())
        dx, dgp, dbp = ops.gn_time_gemm_bwd(terms, n, d, groups, 1e-5, g_, Wd, has_time, dSd)
        e2 = rel(dx, x.grad)
        part = ops.wgrad(terms, n, d, groups, 1e-5, g_, b_, dSd, has_time)
        gW = part.double().sum(0).view(d + has_time, dout).cpu()
        if has_time:
            gW[0] *= tt
        e3 = float((gW - W.grad).abs().max() / max(1.0, float(W.grad.abs().max())))
        e4 = rel(dgp.double().sum(0), gam.grad) if groups else 0.0
        lim = (tol, tol * 20 if cg in (1, 2) else 5e-5, max(tol, 5e-5), 5e-3 if cg == 1 else 1e-4)
        if e1 > lim[0] or e2 > lim[1] or e3 > lim[2] or e4 > lim[3]:
            bad += 1
            print("d=%d g=%d dout=%d n=%d nt=%d time=%d: fwd %.1e dx %.1e dW %.1e dgamma %.1e" % (d, groups, dout, n, nt, has_time, e1, e2, e3, e4))
print("dense fuzz:", "FAILURES %d" % bad if bad else "all within tolerance", "(%d shape classes)" % len(cases))
