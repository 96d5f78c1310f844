"""Depth sweep with early stopping, with the reference's command line, stdout lines and pickle layout
(GCN/train_layers.py:19-183), on the MI355X path.

    python -m graph_odenet_amd.train_layers --dataset cora --runs 5 --layers_min 3 --layers_max 5 [--models ODEK1,RESK1]

For every model family (GCNK, GCNKnorm, RESK1, RESK2, RESK1norm, RESK2norm, ODEK1, ODEK2) and every depth in
[layers_min, layers_max] it trains `--runs` fresh models until the validation accuracy / loss cross the
dataset's thresholds (or `--epochs`), then tests; a depth a family cannot be built with is skipped and raises the
family's `min_layers` (the reference's `except ValueError`).  One `{dataset}_{model}.pickle` per family holds the
arrays `layer_val_acc`, `layer_val_loss` [depth, run, epoch], `layer_convergence`, `layer_test_acc`,
`layer_test_loss` [depth, run], `min_layers`, `max_layers`, as GCN/plot_layers.py reads them.
Additions: --models (subset), --method/--step_size for the ODE blocks, --out_dir, --data_dir/--norm; under
torch.distributed.run the runs of every (family, depth) cell are sharded over the ranks and the arrays are summed.
"""
import argparse
import os
import pickle

import numpy as np
import torch
import torch.nn.functional as F

from . import models
from .data import load_captured, load_planetoid
from .optim import Adam
from .parallel import shard_range
from .train_res import accuracy

FAMILIES = ("GCNK", "GCNKnorm", "RESK1", "RESK2", "RESK1norm", "RESK2norm", "ODEK1", "ODEK2")
# validation thresholds of the reference (GCN/train_layers.py:119-127): 90 % of the accuracy and 110 % of the loss
# of its 2-layer GCN baseline
THRESHOLDS = {"cora": (0.7782 * 0.9, 0.7929 * 1.1), "citeseer": (0.6443 * 0.9, 1.2454 * 1.1),
              "pubmed": (0.7726 * 0.9, 0.7136 * 1.1)}


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--no-cuda', action='store_true', default=False)
    p.add_argument('--fastmode', action='store_true', default=False)
    p.add_argument('--seed', type=int, default=42)
    p.add_argument('--epochs', type=int, default=200)
    p.add_argument('--runs', type=int, default=500)
    p.add_argument('--lr', type=float, default=0.01)
    p.add_argument('--weight_decay', type=float, default=5e-4)
    p.add_argument('--hidden', type=int, default=16)
    p.add_argument('--layers_min', type=int, default=3)
    p.add_argument('--layers_max', type=int, default=5)
    p.add_argument('--dropout', type=float, default=0.5)
    p.add_argument('--dataset', choices=sorted(THRESHOLDS), default="cora")
    p.add_argument('--early_stopping_epochs', type=int, default=10)          # parsed and unused, as in the reference
    p.add_argument('--early_stopping_threshold', type=float, default=1e-10)  # idem
    p.add_argument('--models', default=",".join(FAMILIES))
    p.add_argument('--method', choices=["dopri5", "rk4"], default=None)
    p.add_argument('--step_size', type=float, default=None)
    p.add_argument('--out_dir', default=".")
    p.add_argument('--data_dir', default=None)
    p.add_argument('--norm', choices=["row", "sym", "sum"], default="row")
    return p


class Sweep:
    def __init__(self, args, data, device):
        self.args, self.device = args, device
        self.adj, self.x, self.y, self.itr, self.iva, self.ite = (t.to(device) for t in data)
        self.acc_thr, self.loss_thr = THRESHOLDS[args.dataset]

    def build(self, family, nlayers):
        a = self.args
        kw = dict(nfeat=self.x.shape[1], nhid=a.hidden, nclass=int(self.y.max().item()) + 1, dropout=a.dropout,
                  nlayers=nlayers)
        if family.startswith("ODE"):
            kw.update(method=a.method, step_size=a.step_size)
        return getattr(models, family)(**kw).to(self.device)

    def epoch(self, model, opt):
        model.train()
        opt.zero_grad()
        out = model(self.x, self.adj)
        F.nll_loss(out[self.itr], self.y[self.itr]).backward()
        opt.step()
        if not self.args.fastmode:
            model.eval()
            with torch.no_grad():
                out = model(self.x, self.adj)
        return (F.nll_loss(out[self.iva], self.y[self.iva]).item(), accuracy(out[self.iva], self.y[self.iva]).item())

    def test(self, model):
        model.eval()
        with torch.no_grad():
            out = model(self.x, self.adj)
        return F.nll_loss(out[self.ite], self.y[self.ite]).item(), accuracy(out[self.ite], self.y[self.ite]).item()

    def one_run(self, model, nlayers, rec, run):
        """Trains one model; fills rec[...][nlayers, run] and returns (epochs used, test accuracy)."""
        a = self.args
        opt = Adam(model.parameters(), lr=a.lr, weight_decay=a.weight_decay)
        for ep in range(a.epochs):
            lv, av = self.epoch(model, opt)
            rec["layer_val_loss"][nlayers, run, ep] = lv
            rec["layer_val_acc"][nlayers, run, ep] = av
            if av > self.acc_thr and lv < self.loss_thr:
                rec["layer_convergence"][nlayers, run] = ep
                # the reference back-fills the tail with the value of the epoch BEFORE the crossing
                rec["layer_val_loss"][nlayers, run, ep:] = rec["layer_val_loss"][nlayers, run, ep - 1]
                rec["layer_val_acc"][nlayers, run, ep:] = rec["layer_val_acc"][nlayers, run, ep - 1]
                break
        lt, at = self.test(model)
        rec["layer_test_loss"][nlayers, run] = lt
        rec["layer_test_acc"][nlayers, run] = at
        return int(rec["layer_convergence"][nlayers, run]), at


def new_record(args, hi):
    return {"layer_val_acc": np.zeros([hi, args.runs, args.epochs]), "layer_val_loss": np.zeros([hi, args.runs, args.epochs]),
            "layer_convergence": np.zeros([hi, args.runs]), "layer_test_acc": np.zeros([hi, args.runs]),
            "layer_test_loss": np.zeros([hi, args.runs]), "min_layers": args.layers_min, "max_layers": hi}


def main(argv=None):
    from . import hipgraph
    hipgraph.prefer_safe_graphs()                        # an explicit program-level choice, not an import side effect
    args = build_parser().parse_args(argv)
    if args.no_cuda or not torch.cuda.is_available():
        raise SystemExit("graph_odenet_amd.train_layers needs the GPU: the hot path has no CPU implementation")
    if not args.layers_min < args.layers_max:
        raise SystemExit("--layers_min must be smaller than --layers_max")
    families = [f for f in args.models.split(",") if f]
    for f in families:
        if f not in FAMILIES:
            raise SystemExit("unknown model family %r (known: %s)" % (f, ", ".join(FAMILIES)))
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed + rank)
    torch.cuda.manual_seed(args.seed + rank)
    data = load_planetoid(args.dataset, args.data_dir, args.norm) if args.data_dir else load_captured(args.dataset)
    sweep = Sweep(args, data, device)
    hi = args.layers_max + 1                      # the reference turns the inclusive maximum into a range end
    lo_run, hi_run = shard_range(args.runs, rank, world)
    for family in families:
        rec = new_record(args, hi)
        # not-run cells keep the reference's initial value (convergence = epochs); ranks fill disjoint run slices
        rec["layer_convergence"][:, lo_run:hi_run] = args.epochs
        for nlayers in range(args.layers_min, hi):
            for run in range(lo_run, hi_run):
                try:
                    model = sweep.build(family, nlayers)
                except ValueError:
                    rec["min_layers"] = nlayers + 1        # this family cannot be built that shallow
                    continue
                epochs, acc = sweep.one_run(model, nlayers, rec, run)
                print("{nlayers} layers's run #{run} Test -- epochs: {epochs:d} acc: {acc:.2f}%".format(
                    nlayers=nlayers, run=run, epochs=epochs, acc=100 * acc), flush=True)
        if world > 1:
            for k, v in rec.items():
                if isinstance(v, np.ndarray):
                    t = torch.from_numpy(v).to(device)
                    dist.all_reduce(t)
                    rec[k] = t.cpu().numpy()
            m = torch.tensor([rec["min_layers"]], device=device)
            dist.all_reduce(m, op=dist.ReduceOp.MAX)
            rec["min_layers"] = int(m.item())
        else:
            rec["layer_convergence"][:, :lo_run] = args.epochs
            rec["layer_convergence"][:, hi_run:] = args.epochs
        if rank == 0:
            print("Optimization with model \"{model}\" on dataset \"{dataset}\" Finished!".format(
                model=family, dataset=args.dataset), flush=True)
            with open(os.path.join(args.out_dir, "{dataset}_{model}.pickle".format(dataset=args.dataset, model=family)), "wb") as f:
                pickle.dump(rec, f, protocol=pickle.HIGHEST_PROTOCOL)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
