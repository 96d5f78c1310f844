import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graph_odenet_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0"); n, d = 1 << 20, 128
X, K1, K2, K3 = [torch.randn(n, d, device=dev) for _ in range(4)]
W = torch.randn(d + 1, d, device=dev) / d ** .5
gam, bet = torch.rand(d, device=dev) + .5, torch.rand(d, device=dev) - .5
out = torch.empty(n, d, device=dev)
for terms in ([(1., X)], [(1., X), (.1, K1), (-.1, K2), (.1, K3)]):
    for _ in range(3):
        ops.gn_time_gemm(terms, n, d, 32, 1e-5, gam, bet, W, True, .3, out=out)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (2048 * 8))()
    lib._handle  # noqa
    f = ctypes.CDLL(_lib.LIB_PATH).gode_debug_read
    f.argtypes = [ctypes.c_void_p, ctypes.c_int]
    print("rc", f(buf, 2048 * 8))
    import numpy as np
    a = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 8).astype(np.float64)
    tiles = 65536 / 2048
    print("terms=%d per-tile cycles (s_memtime @100MHz? raw units): load %.0f  valu %.0f  mfma %.0f  store %.0f | wave total %.0f (%.0f per tile)" % (
        len(terms), *(a[:, q].mean() / tiles for q in range(4)), a[:, 4].mean(), a[:, 4].mean() / tiles))
