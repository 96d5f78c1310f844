"""Dev check (build container only): the reference's GAT GCN3 trained on CPU for a few epochs vs ours on the same init,
dropout 0 - are the loss trajectories the same, i.e. is the poor accuracy a property of the reference model?"""
import importlib, os, sys, types, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_odenet_amd.data import load_captured_gat
stub = types.ModuleType("torchdiffeq"); stub.odeint_adjoint = None; sys.modules["torchdiffeq"] = stub
sys.path.insert(0, "/root/reference/GAT")
ref_models = importlib.import_module("models")
sys.path.pop(0)
src, tgt, Mtgt, x, y, itr, iva, ite = load_captured_gat("cora")
torch.manual_seed(0)
m = ref_models.GCN3(nfeat=x.shape[1], nhid=16, nclass=7, dropout=0.0)
sd = {k: v.clone() for k, v in m.state_dict().items()}
opt = torch.optim.Adam(m.parameters(), lr=0.01, weight_decay=5e-4)
losses = []
for ep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    m.train(); opt.zero_grad()
    out = m(x, src, tgt, Mtgt)
    loss = F.nll_loss(out[itr], y[itr]); loss.backward(); opt.step()
    losses.append(float(loss))
m.eval()
out = m(x, src, tgt, Mtgt)
acc = float((out[ite].argmax(1) == y[ite]).float().mean())
tg = torch.zeros(x.shape[0]).index_add_(0, tgt, torch.ones(tgt.numel()))
print("reference GAT GCN3 on cora (CPU): loss %s ... %s, test acc %.3f; nodes without incoming edge: %d of %d (train: %d of %d)" % (
    ["%.4f" % l for l in losses[:3]], ["%.4f" % l for l in losses[-3:]], acc, int((tg == 0).sum()), x.shape[0], int((tg[itr] == 0).sum()), itr.numel()))
torch.save({"sd": sd, "losses": losses}, "/tmp/gat_ref_train.pt")
