"""The solver restatement (oracle/solver_ref.py) against analytic solutions and scipy RK45.
Parity with torchdiffeq itself is UNPINNED (dependency absent, no reference fixtures)."""
import math

import numpy as np
import torch
from scipy.integrate import solve_ivp
from scipy.linalg import expm

from oracle import solver_ref as S


def test_dopri5_harmonic_oscillator():
    A = torch.tensor([[0., 1.], [-1., 0.]])
    y = S.odeint(lambda t, y: y @ A.T, torch.tensor([1., 0.]), torch.tensor([0., 1.]), 1e-5, 1e-5)[1]
    assert abs(y[0].item() - math.cos(1)) < 5e-5 and abs(y[1].item() + math.sin(1)) < 5e-5


def test_rk4_38_order():
    """3/8 rule is 4th order: halving h cuts the error ~16x on y' = -2y."""
    f = lambda t, y: -2.0 * y   # noqa: E731
    y0 = torch.tensor([1.0], dtype=torch.float64)
    t = torch.tensor([0., 1.], dtype=torch.float64)
    errs = []
    for n in (4, 8, 16):
        y = S.odeint(f, y0, t, method="rk4", options={"step_size": 1.0 / n})[1]
        errs.append(abs(y.item() - math.exp(-2)))
    assert 12 < errs[0] / errs[1] < 20 and 12 < errs[1] / errs[2] < 20


def test_dopri5_vs_scipy_graph_diffusion():
    """y' = (A_hat - I) y on a small graph: both must match expm to O(tol) and use similar NFE."""
    g = torch.Generator().manual_seed(0)
    n = 30
    A = (torch.rand(n, n, generator=g) < 0.15).float() + torch.eye(n)
    A = A / A.sum(1, keepdim=True)
    L = (A - torch.eye(n)).double()
    y0 = torch.randn(n, 4, generator=g).double()
    exact = torch.from_numpy(expm(L.numpy())) @ y0
    st = {}
    ours = S.odeint(lambda t, y: L @ y, y0, torch.tensor([0., 1.], dtype=torch.float64), 1e-5, 1e-5, stats=st)[1]
    ref = solve_ivp(lambda t, y: (L.numpy() @ y.reshape(n, 4)).ravel(), (0, 1), y0.numpy().ravel(),
                    method="RK45", rtol=1e-5, atol=1e-5)
    assert (ours - exact).abs().max() < 1e-4
    assert np.abs(ref.y[:, -1].reshape(n, 4) - exact.numpy()).max() < 1e-4
    assert st["accepted"] <= 2 * len(ref.t)


def test_adjoint_matches_backprop():
    torch.manual_seed(0)

    class F(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.W = torch.nn.Parameter(torch.randn(3, 3) * 0.5)

        def forward(self, t, y):
            return torch.tanh(y @ self.W) * (1 + t)
    m = F()
    t = torch.tensor([0., 1.])
    for method, opt, tol in ((None, None, 5e-5), ("rk4", {"step_size": 1 / 16}, 1e-5)):
        y0 = torch.randn(4, 3, requires_grad=True)
        m.zero_grad()
        S.odeint_adjoint(m, y0, t, 1e-6, 1e-6, method, opt)[1].pow(2).sum().backward()
        g1, gy1 = m.W.grad.clone(), y0.grad.clone()
        m.zero_grad(); y0.grad = None
        S.odeint(m, y0, t, 1e-6, 1e-6, method, opt)[1].pow(2).sum().backward()
        assert (g1 - m.W.grad).abs().max() < tol and (gy1 - y0.grad).abs().max() < tol
