"""Row-partitioned solve: gradient differences rank-count 1 vs 2 at tol 1e-4 vs 1e-6 (HISTORY.md, partition section)."""
import os, sys
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_partition as T
from graph_odenet_amd import models, solver
from graph_odenet_amd.partition import PartitionedGraph, RowPartition
dev = torch.device("cuda:0")
A, x, y, train = T._problem()
def run(tol, graph, native=True):
    solver.DOPRI5_NATIVE = native
    torch.manual_seed(11)
    m = models.ODEGCN3(T.NFEAT, T.NHID, T.NCLASS, 0.0, method="dopri5", tol=tol).to(dev)
    out = m(x.to(dev), graph)
    F.nll_loss(out[train.to(dev)], y.to(dev)[train.to(dev)]).backward()
    return out.detach(), torch.cat([p.grad.reshape(-1) for p in m.parameters()]), m.nfe
Ad = A.to(dev)
pg = PartitionedGraph.from_adj(Ad, RowPartition(T.N, 1, 0))
ref = run(1e-7, Ad)
for tol in (1e-4, 1e-5, 1e-6):
    a = run(tol, Ad); b = run(tol, Ad, native=False); c = run(tol, pg)
    g = ref[1].abs().max()
    print("tol %.0e: whole-native vs tight: out %.1e grad %.1e (rel %.1e) | per-stage vs native: grad %.1e | part1 vs per-stage: grad %.1e | nfe %s %s %s" % (
        tol, (a[0]-ref[0]).abs().max(), (a[1]-ref[1]).abs().max(), (a[1]-ref[1]).abs().max()/g, (b[1]-a[1]).abs().max(), (c[1]-b[1]).abs().max(), a[2], b[2], c[2]), flush=True)
