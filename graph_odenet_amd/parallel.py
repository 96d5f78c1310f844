"""Multi-GPU data parallelism for the hot path: one process per GPU (torch.distributed, backend
"nccl" == RCCL over xGMI).  The reference has no distributed code; the natural
shard is a batch of independent graphs (QC/QM9 mini-batches, or one full graph per rank): each
rank runs the whole forward/adjoint pass on its own graph and the only exchange step is ONE
flattened fp32 all-reduce of the parameter gradients per step (<= 100 KB for the GCN models,
57 MB for QC's enn-s2s: latency-bound vs one ring pass over 153 GB/s links).
"""
import torch
import torch.distributed as dist


FORCE_COLLECTIVES = False      # rehearsal (bench.py --force-dist): issue the collectives even with a single rank


def _alone():
    return not dist.is_initialized() or (dist.get_world_size() == 1 and not FORCE_COLLECTIVES)


def broadcast_parameters(module, src=0):
    """Make every rank start from rank `src`'s weights (one flat broadcast)."""
    if _alone():
        return
    ps = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    if not ps:
        return
    flat = torch.cat([p.reshape(-1).float() for p in ps])
    if dist.get_backend() == "gloo" and flat.is_cuda:               # one-box rehearsal: staged through the host
        host = flat.cpu()
        dist.broadcast(host, src)
        flat = host.to(flat.device)
    else:
        dist.broadcast(flat, src)
    o = 0
    for p in ps:
        n = p.numel()
        p.copy_(flat[o:o + n].view_as(p))
        o += n


class GradBucket:
    """Persistent flat buffer so that the all-reduce is a single collective per step.  The parameters' `.grad`
    tensors are made VIEWS of the buffer (autograd accumulates into an existing `.grad` in place), so no gradient is
    copied in or out around the collective as long as the optimiser keeps them (`zero_grad(set_to_none=False)`);
    a gradient that was dropped or replaced in the meantime is copied in and re-attached."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.views = []
        o = 0
        for p in self.params:
            k = p.numel()
            v = self.flat[o:o + k].view_as(p)
            if p.grad is not None:
                v.copy_(p.grad)
            p.grad = v
            self.views.append(v)
            o += k

    def allreduce_mean(self):
        """grad <- mean over ranks.  Parameters without a gradient on this rank count as zero."""
        self._allreduce(True)

    def allreduce_sum(self):
        """grad <- sum over ranks: the ranks hold partial sums of ONE model's gradient (partition.py)."""
        self._allreduce(False)

    def _allreduce(self, mean):
        if _alone():
            return
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
                p.grad = v
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
                p.grad = v
        if dist.get_backend() == "gloo" and self.flat.is_cuda:      # one-box rehearsal: staged through the host
            host = self.flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            self.flat.copy_(host)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        if mean:
            self.flat.div_(dist.get_world_size())


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items work units (graphs, runs) for `rank`."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
