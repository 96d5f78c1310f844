"""Training harness for the citation graphs with the reference's command line and output format
(GCN/train_res.py:17-158), on the MI355X path.

    python -m graph_odenet_amd.train_res --model ode3 --dataset cora [--runs N] [--method rk4 --step_size 0.0625]
    python -m graph_odenet_amd.train_res --variant gat --model ode3 --dataset citeseer        (GAT/train_res.py)

Same flags, defaults, seeding rule (seed applied only when --runs 1), per-epoch line, `Run #i Test --`
line and closing summary, so GCN/results/basic/{stats,make_table}.py parse its stdout unchanged.
Additions: --method/--step_size/--tol for the ODE block, --data_dir for raw Planetoid files (default:
the loader outputs captured in tests/golden), and, when launched with torch.distributed.run, the
independent `--runs` are sharded over the ranks (one GPU each; the only collective is the final sum
of loss / accuracy / time).
"""
import argparse
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import gat_models, models
from .data import load_captured, load_captured_gat, load_planetoid, load_planetoid_gat
from .optim import Adam
from .parallel import shard_range

def _model_dict(mod):
    return {"gcn2": mod.GCN, "gcn3": mod.GCN3, "gcn3norm": mod.GCN3, "res3": mod.RGCN3,
            "ode3": mod.ODEGCN3, "res3norm": mod.RGCN3norm, "res3fullnorm": mod.RGCN3fullnorm,
            "ode3norm": mod.ODEGCN3fullnorm}      # "gcn3norm" -> GCN3 as in the reference's model_dict (Q2)


MODELS = _model_dict(models)
VARIANTS = {"gcn": MODELS, "gat": _model_dict(gat_models)}      # GCN/train_res.py and GAT/train_res.py (same CLI)


def accuracy(output, labels):
    return (output.max(1)[1] == labels).double().mean()


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--no-cuda', action='store_true', default=False)
    p.add_argument('--fastmode', action='store_true', default=False)
    p.add_argument('--seed', type=int, default=42)
    p.add_argument('--epochs', type=int, default=200)
    p.add_argument('--runs', type=int, default=1)
    p.add_argument('--lr', type=float, default=0.01)
    p.add_argument('--weight_decay', type=float, default=5e-4)
    p.add_argument('--hidden', type=int, default=16)
    p.add_argument('--dropout', type=float, default=0.5)
    p.add_argument('--dataset', choices=["cora", "citeseer", "pubmed"], default="cora")
    p.add_argument('--model', choices=sorted(MODELS), default="res3")
    p.add_argument('--method', choices=["dopri5", "rk4"], default=None)
    p.add_argument('--step_size', type=float, default=None)
    p.add_argument('--tol', type=float, default=1e-5)
    p.add_argument('--data_dir', default=None)
    p.add_argument('--norm', choices=["row", "sym", "sum"], default="row")
    p.add_argument('--partition', action='store_true', default=False,
                   help="under torch.distributed.run: ONE model on the graph split by rows over the ranks "
                        "(partition.py) instead of sharding --runs over them")
    p.add_argument('--dist_backend', choices=["nccl", "gloo"], default="nccl",
                   help="gloo: rehearsal with several ranks sharing one GPU (exchange staged through the host)")
    p.add_argument('--heads', type=int, default=1,
                   help="gat variant: H reference attention heads side by side in every layer whose width H divides "
                        "(gat_heads.py; --hidden must be a multiple of H)")
    p.add_argument('--variant', choices=sorted(VARIANTS), default="gcn",
                   help="gcn: models over a normalised adjacency (GCN/train_res.py); gat: edge attention over "
                        "(src, tgt, Mtgt) (GAT/train_res.py)")
    return p


class Trainer:
    def __init__(self, args, data, device, verbose):
        self.args, self.verbose, self.device = args, verbose, device
        *graph, self.x, self.y, self.itr, self.iva, self.ite = (t.to(device) for t in data)
        self.graph = tuple(graph)                  # (adj,) or (src, tgt, Mtgt)
        self.is_ode = "ode" in args.model

    def new_model(self):
        a = self.args
        kw = dict(nfeat=self.x.shape[1], nhid=a.hidden, nclass=int(self.y.max().item()) + 1, dropout=a.dropout)
        if self.is_ode:
            kw.update(method=a.method, step_size=a.step_size, tol=a.tol)
        table = VARIANTS[a.variant]
        if a.heads > 1:
            if a.variant != "gat":
                raise SystemExit("--heads applies to --variant gat")
            from . import gat_heads
            table = _model_dict(gat_heads.zoo(a.heads))
        model = table[a.model](**kw).to(self.device)
        opt = Adam(model.parameters(), lr=a.lr, weight_decay=a.weight_decay)       # optim.py: torch.optim.Adam's update, one launch
        return model, opt

    # hooks of the row-partitioned trainer below
    def loss(self, out, idx):
        return F.nll_loss(out[idx], self.y[idx])

    def acc(self, out, idx):
        return accuracy(out[idx], self.y[idx])

    def value(self, t):
        return t.item()

    def sync_grads(self, model):
        pass

    def epoch(self, model, opt, ep):
        t0 = time.time()
        model.nfe = 0
        model.train()
        opt.zero_grad()
        out = model(self.x, *self.graph)
        nfe_f = model.nfe
        model.nfe = 0
        loss = self.loss(out, self.itr)
        acc = self.acc(out, self.itr)
        loss.backward()
        self.sync_grads(model)
        opt.step()
        nfe_b = model.nfe
        model.nfe = 0
        if not self.args.fastmode:
            model.eval()
            with torch.no_grad():
                out = model(self.x, *self.graph)
        lv = self.loss(out, self.iva)
        av = self.acc(out, self.iva)
        loss, acc, lv, av = (_Val(self.value(t)) for t in (loss, acc, lv, av))
        if self.verbose:
            print('Epoch: {:04d}'.format(ep + 1), 'loss_train: {:.4f}'.format(loss.item()),
                  'acc_train: {:.4f}'.format(acc.item()), 'loss_val: {:.4f}'.format(lv.item()),
                  'acc_val: {:.4f}'.format(av.item()), 'time: {:.4f}s'.format(time.time() - t0),
                  "" if not self.is_ode else 'nfe_f: {}'.format(nfe_f),
                  "" if not self.is_ode else 'nfe_b: {}'.format(nfe_b))
        return lv.item(), av.item()

    def test(self, model):
        model.eval()
        with torch.no_grad():
            out = model(self.x, *self.graph)
        lt = self.value(self.loss(out, self.ite))
        at = self.value(self.acc(out, self.ite))
        if self.verbose:
            print("Test set results:", "loss= {:.4f}".format(lt), "accuracy= {:.4f}".format(at))
        return lt, at


class _Val(float):
    """A python float that still answers .item() (the print block above is the reference's)."""

    def item(self):
        return float(self)


class PartitionedTrainer(Trainer):
    """--partition: ONE model on ONE graph split by rows over the ranks (graph_odenet_amd/partition.py).  Every rank
    holds its rows of the features / labels and its row blocks of A and A^T; losses and accuracies are sums over the
    local members of an index set divided by the set's global size (summed over the ranks for printing), and the
    parameter gradients - partial sums - are added over the ranks before the optimiser step."""

    def __init__(self, args, data, device, verbose):
        from .parallel import GradBucket
        from .partition import PartitionedGraph, RowPartition
        if args.variant != "gcn":
            raise SystemExit("--partition is built for the gcn variant")
        self.args, self.verbose, self.device = args, verbose, device
        adj, x, y, itr, iva, ite = data
        adj = adj.coalesce() if adj.is_sparse else adj.to_sparse().coalesce()
        n = adj.shape[0]
        rows, cols = adj.indices()
        self.part = RowPartition.balanced(n, rows, cols)
        self.graph = (PartitionedGraph.from_coo(rows, cols, adj.values(), n, self.part, device=device),)
        self.x, self.y = self.part.take(x).to(device), self.part.take(y).to(device)
        self.n_class = int(y.max().item()) + 1
        self.itr, self.iva, self.ite = ((self.part.local_positions(i).to(device), i.numel()) for i in (itr, iva, ite))
        self.is_ode = "ode" in args.model
        self._bucket_of, self._GradBucket = {}, GradBucket

    def new_model(self):
        a = self.args
        kw = dict(nfeat=self.x.shape[1], nhid=a.hidden, nclass=self.n_class, dropout=a.dropout)
        if self.is_ode:
            kw.update(method=a.method, step_size=a.step_size, tol=a.tol)
        model = VARIANTS[a.variant][a.model](**kw).to(self.device)       # same seed on every rank: same initial weights
        opt = Adam(model.parameters(), lr=a.lr, weight_decay=a.weight_decay)       # optim.py: torch.optim.Adam's update, one launch
        self._bucket_of = {id(model): self._GradBucket(model)}
        return model, opt

    def loss(self, out, idx):
        pos, count = idx
        return F.nll_loss(out[pos], self.y[pos], reduction="sum") / count

    def acc(self, out, idx):
        pos, count = idx
        return (out[pos].max(1)[1] == self.y[pos]).double().sum() / count

    def value(self, t):
        from .partition import global_sum
        return float(global_sum(t.detach().double().reshape(1).clone(), self.part.group))

    def sync_grads(self, model):
        self._bucket_of[id(model)].allreduce_sum()


def main(argv=None):
    from . import hipgraph
    hipgraph.prefer_safe_graphs()                        # an explicit program-level choice, not an import side effect
    args = build_parser().parse_args(argv)
    if args.no_cuda or not torch.cuda.is_available():
        raise SystemExit("graph_odenet_amd.train_res needs the GPU: the hot path has no CPU implementation")
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) if args.dist_backend == "nccl" else 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    if args.runs == 1 or args.partition:
        np.random.seed(args.seed)
        torch.manual_seed(args.seed)
        torch.cuda.manual_seed(args.seed)
    if args.variant == "gat":
        data = load_planetoid_gat(args.dataset, args.data_dir) if args.data_dir else load_captured_gat(args.dataset)
    else:
        data = load_planetoid(args.dataset, args.data_dir, args.norm) if args.data_dir else load_captured(args.dataset)
    if args.partition:
        tr = PartitionedTrainer(args, data, device, verbose=(args.runs == 1 and rank == 0))
        lo, hi = 0, args.runs                      # every rank takes part in every run
    else:
        tr = Trainer(args, data, device, verbose=(args.runs == 1 and rank == 0))
        lo, hi = shard_range(args.runs, rank, world)
    tot = torch.zeros(3, dtype=torch.float64, device=device)
    model = None
    for run in range(lo, hi):
        if args.runs > 1:
            torch.manual_seed(args.seed + run)          # independent replicas: seed = base + run
            torch.cuda.manual_seed(args.seed + run)
        model, opt = tr.new_model()
        t0 = time.time()
        for ep in range(args.epochs):
            tr.epoch(model, opt, ep)
        torch.cuda.synchronize()
        dt = time.time() - t0
        lt, at = tr.test(model)
        if args.runs > 1 and (rank == 0 or not args.partition):
            print("Run #{run} Test -- time: {time}s acc: {acc:.2f}%".format(run=run, time=dt, acc=100 * at), flush=True)
        tot += torch.tensor([lt, at, dt], dtype=torch.float64, device=device)
    if world > 1 and not args.partition:
        dist.all_reduce(tot)
    tot = (tot / max(args.runs, 1)).tolist()
    if rank == 0:
        print("Optimization on dataset \"{dataset}\" Finished!".format(dataset=args.dataset))
        if model is not None:
            print("#Parameters: {param_count}".format(param_count=sum(p.numel() for p in model.parameters() if p.requires_grad)))
        print("Average time elapsed: {:.4f}s".format(tot[2]))
        print("Test set results:", "avg loss= {:.4f}".format(tot[0]), "avg accuracy= {:.4f}".format(tot[1]))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
