"""Row-partitioned solve: two node orders on ONE GPU differ by the same rounding amounts as two ranks (cited by tests/test_gpu_partition.py)."""
import os, sys
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_partition as T
from graph_odenet_amd.partition import PartitionedGraph, RowPartition
from graph_odenet_amd import odeint
dev = torch.device("cuda:0")
A, x, y, train = T._problem()
def whole(native=True):
    odeint.NATIVE_RK4 = native
    m = T._model("ODEGCN3", dev)
    out = m(x.to(dev), A.to(dev))
    F.nll_loss(out[train.to(dev)], y.to(dev)[train.to(dev)]).backward()
    return out.detach(), {n: p.grad.clone() for n, p in m.named_parameters()}
def parted(world):
    # emulate `world` ranks sequentially in one process: gather by hand
    outs, grads = [], None
    parts = [RowPartition(T.N, world, r) for r in range(world)]
    if world == 1:
        pg = PartitionedGraph.from_adj(A.to(dev), parts[0])
        m = T._model("ODEGCN3", dev)
        out = m(parts[0].take(x).to(dev), pg)
        pos = parts[0].local_positions(train).to(dev)
        (F.nll_loss(out[pos], parts[0].take(y).to(dev)[pos], reduction="sum") / train.numel()).backward()
        return out.detach(), {n: p.grad.clone() for n, p in m.named_parameters()}
o0, g0 = whole(True)
o1, g1 = whole(False)
o2, g2 = parted(1)
for n in g0:
    print("%-28s max|g| %.4f  native-vs-stage %.2e  stage-vs-part1 %.2e" % (n, g0[n].abs().max(), (g0[n]-g1[n]).abs().max(), (g1[n]-g2[n]).abs().max()))
print("out", (o0-o1).abs().max().item(), (o1-o2).abs().max().item())
# whole graph with the nodes renumbered by the 2-rank cyclic map, on one GPU: isolates summation-order effects
p2 = RowPartition(T.N, 2, 0)
ids = torch.arange(T.N)
new = p2.renumber(ids) if T.N % 2 == 0 else None
if new is not None:
    inv = torch.empty_like(new); inv[new] = ids
    Ad = A.to_dense()[inv][:, inv]
    odeint.NATIVE_RK4 = True
    m = T._model("ODEGCN3", dev)
    out = m(x[inv].to(dev), Ad.to_sparse().to(dev))
    tr = new[train]
    F.nll_loss(out[tr.to(dev)], y[inv].to(dev)[tr.to(dev)]).backward()
    g3 = {n: p.grad.clone() for n, p in m.named_parameters()}
    for n in g0:
        print("%-28s renumbered-vs-original %.2e" % (n, (g0[n]-g3[n]).abs().max()))
    print("out", (out.detach()[new.to(dev)] - o0).abs().max().item())
