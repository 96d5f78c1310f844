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


def test_dopri5_step_equals_scipy_rk45_step():
    """One Dormand-Prince 5(4) step of the oracle (all seven stages and the 5th-order solution) against scipy's own
    RK45 step routine on the same (t, y, h): the same published tableau, so the numbers agree to fp64 rounding.  Pins
    the stage / solution arithmetic to an independent implementation.  The embedded ERROR estimate is deliberately not
    compared: the oracle follows torchdiffeq's Dormand-Prince-Shampine weights (c_error = b - [1951/21600, 0,
    22642/50085, 451/720, -12231/42400, 649/6300, 1/60]), scipy the original b* of Dormand & Prince (5179/57600, ...);
    both are 4th-order companions of the same 5th-order solution and differ at O(h^5).  The controller and the dense
    output stay pinned only by the sources named in solver_ref.py."""
    from scipy.integrate._ivp import rk
    g = torch.Generator().manual_seed(3)
    n = 12
    M = (torch.randn(n, n, generator=g) * 0.4).double()
    fun = lambda t, y: np.tanh(M.numpy() @ y) * (1.0 + 0.3 * t)          # noqa: E731
    f_t = lambda t, y: (torch.tanh(y[0] @ M.T) * (1.0 + 0.3 * t),)        # noqa: E731  (tuple state, as the solver calls it)
    y0 = torch.randn(n, generator=g).double()
    for t0, h in ((0.0, 0.05), (0.4, 0.37), (1.0, -0.2)):
        f0 = fun(t0, y0.numpy())
        K = np.empty((rk.RK45.n_stages + 1, n))
        y_new, f_new = rk.rk_step(fun, t0, y0.numpy(), f0, h, rk.RK45.A, rk.RK45.B, rk.RK45.C, K)
        y1, f1, err, k = S._dopri5_step(f_t, (y0,), (torch.from_numpy(f0),), torch.tensor(t0, dtype=torch.float64),
                                        torch.tensor(h, dtype=torch.float64))
        assert np.abs(y1[0].numpy() - y_new).max() < 1e-14
        assert np.abs(f1[0].numpy() - f_new).max() < 1e-14
        # both error estimates are O(h^5) and of the same order of magnitude
        e_scipy, e_ours = np.abs(K.T @ rk.RK45.E * h).max(), err[0].abs().max().item()
        assert 0.05 < e_ours / e_scipy < 20
        for s in range(7):
            assert np.abs(k[s][0].numpy() - K[s]).max() < 1e-14


def test_initial_step_norm_per_tensor_vs_pooled_on_a_four_tensor_state():
    """The adjoint solve integrates FOUR tensors (y, a, a_t, a_theta).  The initial-step heuristic's norm is an
    explicit option: "per_tensor" (default; torchdiffeq 0.0.x as publicly understood: per-tensor
    RMS values, h0 = 0.01 max_i d0_i/d1_i, h1 = (0.01 / max(d1 + d2))^(1/5)) against "pooled" (one RMS over everything,
    rounds 1-2).  A state whose tensors differ in size and scale separates them; the expected numbers are formed by
    hand from the definition.  graph_odenet_amd/solver.py mirrors the same switch (GPU test:
    tests/test_gpu_gcn.py::test_initial_step_norm_option_matches_oracle)."""
    rtol = atol = 1e-5
    # linear field f_i(y) = lam_i * y_i per tensor: f(t0 + h0) - f0 = lam_i * h0 * f0_i, everything in closed form
    sizes = (1000, 1000, 1, 50)
    vals = (1.0, 1e-2, 1.0, 0.5)
    lam = (-0.5, -40.0, -400.0, -2.0)             # the stiff direction sits in the ONE-element tensor (as a_t does)
    y0 = tuple(torch.full((n,), v, dtype=torch.float64) for n, v in zip(sizes, vals))
    func = lambda t, ys: tuple(l * y for l, y in zip(lam, ys))          # noqa: E731
    t0 = torch.tensor(0.0, dtype=torch.float64)
    f0 = func(t0, y0)
    got = {nm: S._select_initial_step(func, t0, y0, 4, rtol, atol, f0, norm=nm) for nm in ("per_tensor", "pooled")}
    # by hand
    scale = [atol + abs(v) * rtol for v in vals]
    d0 = [abs(v) / s for v, s in zip(vals, scale)]
    d1 = [abs(l * v) / s for l, v, s in zip(lam, vals, scale)]
    h0 = 0.01 * max(a / b for a, b in zip(d0, d1))                          # = 0.01 / min |lam| = 0.02
    assert abs(h0 - 0.02) < 1e-12
    d2 = [abs(l * h0 * l * v) / s / h0 for l, v, s in zip(lam, vals, scale)]
    want_pt = min(100 * h0, (0.01 / max(d1 + d2)) ** 0.2)
    n_all = sum(sizes)
    pool = lambda ds: math.sqrt(sum(n * d * d for n, d in zip(sizes, ds)) / n_all)      # noqa: E731
    p0, p1 = pool(d0), pool(d1)
    h0p = 0.01 * p0 / p1
    p2 = pool([abs(l * h0p * l * v) / s for l, v, s in zip(lam, vals, scale)]) / h0p
    want_pool = min(100 * h0p, (0.01 / max(p1, p2)) ** 0.2)
    assert abs(got["per_tensor"] - want_pt) < 1e-9 * want_pt
    assert abs(got["pooled"] - want_pool) < 1e-9 * want_pool
    assert abs(got["per_tensor"] - got["pooled"]) > 0.2 * got["pooled"]    # they really differ on such a state
    # one tensor: identical
    one = (y0[0],)
    f1 = lambda t, ys: (lam[0] * ys[0],)                                   # noqa: E731
    a = S._select_initial_step(f1, t0, one, 4, rtol, atol, f1(t0, one), norm="per_tensor")
    b = S._select_initial_step(f1, t0, one, 4, rtol, atol, f1(t0, one), norm="pooled")
    assert a == b
    assert S.INITIAL_STEP_NORM == "per_tensor"
    # the option reaches the solves: a backward solve over a tuple state takes a different first step
    seqs = {}
    for nm in ("per_tensor", "pooled"):
        S.INITIAL_STEP_NORM, S.TRACE = nm, []
        try:
            S.odeint(func, y0, torch.tensor([0.0, 0.3], dtype=torch.float64), rtol, atol)
            seqs[nm] = S.TRACE[0][0][0]
        finally:
            S.INITIAL_STEP_NORM, S.TRACE = "per_tensor", None
    assert abs(seqs["per_tensor"] - want_pt) < 1e-9 and abs(seqs["pooled"] - want_pool) < 1e-9
