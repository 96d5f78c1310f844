"""C5 decomposition of the dense product: bare product, + time row, + GroupNorm, forward and VJP (HISTORY.md, dense kernels)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_odenet_amd import ops
dev = torch.device("cuda:0")
n, d = 1 << 20, 128
x = torch.randn(n, d, device=dev); out = torch.empty(n, d, device=dev); dS = torch.randn(n, d, device=dev)
W = torch.randn(d + 1, d, device=dev) / d ** 0.5; W0 = W[1:].contiguous()
gam, bet = torch.rand(d, device=dev) + 0.5, torch.rand(d, device=dev) - 0.5
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
T = [(1.0, x)]
print("fwd  GN+time %.3f | no GN, time %.3f | no GN, no time %.3f ms" % (
    t(lambda: ops.gn_time_gemm(T, n, d, 32, 1e-5, gam, bet, W, True, 0.3, out=out)),
    t(lambda: ops.gn_time_gemm(T, n, d, 0, 0.0, None, None, W, True, 0.3, out=out)),
    t(lambda: ops.gn_time_gemm(T, n, d, 0, 0.0, None, None, W0, False, 0.0, out=out))))
print("bwd  GN %.3f | no GN %.3f ms" % (
    t(lambda: ops.gn_time_gemm_bwd(T, n, d, 32, 1e-5, gam, W, True, dS, out=out)),
    t(lambda: ops.gn_time_gemm_bwd(T, n, d, 0, 0.0, None, W, True, dS, out=out, want_affine_grads=False))))
print("wgrad GN %.3f | no GN %.3f ms" % (
    t(lambda: ops.wgrad(T, n, d, 32, 1e-5, gam, bet, dS, True)),
    t(lambda: ops.wgrad(T, n, d, 0, 0.0, None, None, dS, True))))
print("torch.mm %.3f ms ; copy %.3f ms" % (t(lambda: torch.mm(x, W0, out=out)), t(lambda: out.copy_(x))))
