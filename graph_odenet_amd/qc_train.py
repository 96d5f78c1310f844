"""The body of the reference's QC training loop (QC/util.py:146-211: zero_grad, forward, criterion, backward,
optimizer.step on a NEW mini-batch every iteration) as one object, with the mode this platform runs it fastest in as
the default (BASELINE.json configs[3]).

    step = qc_train.TrainStep(model, optimizer, criterion)            # mode="auto"
    for batch_size, g, b, x, e_d, e_src, e_tgt, target in train_loader:
        loss = step(x, e_d, e_src, e_tgt, b, target)                  # a device tensor; read it when you log

`e_tgt` is the reference collate's dense N x E incidence or - cheaper, what a loader has before it builds that matrix -
the per-edge target index vector (int64[E]); `b` the sorted graph index of every node (collate order).

Modes:
  "eager"     the reference's loop body as it stands: per-batch conversion of the dense matrix (a column arg-max, a
              validity check = one host synchronisation), ~250-450 launches from Python;
  "prepared"  the edges go in as index vectors (qc_batch.prepare): handed over by the loader, or taken off the dense
              matrix by a column arg-max on the device - no validity check, no host synchronisation, the GPU queue
              stays full;
  "captured"  the batch is padded to its shape bucket (qc_batch.pad_batch: multiples of 64 atoms / 128 edges, one dummy
              graph) and the whole step - conversion, forward, loss, backward, Adam - is ONE HIP-graph replay per
              bucket (qc_step.CapturedQCStep); needs hipgraph.memset_nodes_ok();
  "auto"      = "prepared".  Measured on never-repeating batches of 20 molecules (tools/config_bench.py, end of round 4):
              EdgeGCN_K_Sum eager 1.6-1.8 ms / prepared 1.12 ms / captured 1.23 ms per step, MPNN_ENN_K_Set2Set 1.5-1.8 / 1.29 /
              1.41 ms.  The captured step still loses on this platform: replayed memset nodes are only reliable with the HIP
              runtime's graph fast path OFF (hipgraph.py), on that path a replay issues its kernel nodes one by one from the
              host - its time is the step's kernel time, without the overlap of the eager launches - and the padding adds
              work.  (On a host too slow to issue ~120 launches per millisecond the captured step is the steadier one.)
The number of graphs of a batch is taken from `target.shape[0]` (one row per graph, as the reference's collate emits),
never read back from the device.

Multi-GPU (one rank per GPU, parallel.GradBucket): pass `exchange=lambda: bucket.allreduce_mean(assume_all=True)`; then
forward + backward are captured and the exchange and the optimiser step run after the replay.

A learning-rate change between epochs (QC/train_egcn.py:160-164) reaches a captured step only through a new capture:
call `step.reset()` after changing `param_groups[...]["lr"]`.
"""
import torch

from . import hipgraph
from .qc_batch import pad_batch, prepare
from .qc_step import CapturedQCStep


def _target_index(Etgt):
    """The per-edge target vector off the collate's dense N x E matrix: (Etgt != 0).to(uint8).argmax(0), in one launch
    where the matrix is a 2-D fp32 GPU tensor with unit column stride (csrc/convert.hip)."""
    if Etgt.is_cuda and Etgt.dtype == torch.float32 and Etgt.dim() == 2 and Etgt.stride(1) == 1 and Etgt.shape[0] > 0:
        from . import ops
        return ops.dense_first_nonzero(Etgt)
    return (Etgt != 0).to(torch.uint8).argmax(0)


class TrainStep:
    def __init__(self, model, optimizer, criterion, mode="auto", exchange=None):
        if mode not in ("auto", "eager", "prepared", "captured"):
            raise ValueError("TrainStep: mode must be auto / eager / prepared / captured")
        self.model, self.opt, self.criterion, self.exchange = model, optimizer, criterion, exchange
        self.requested = mode
        self.mode = None                 # decided at the first call (needs the device)
        self._captured = None

    def reset(self):
        """Drop the captured graphs (after a hyper-parameter change that a captured step bakes in)."""
        self._captured = None

    def _decide(self, x, by_index):
        m = self.requested
        if m == "auto":
            m = "prepared" if x.is_cuda else "eager"
        return m

    def __call__(self, x, edge_feat, Esrc, Etgt, batch, target):
        by_index = Etgt.dim() == 1 and not Etgt.is_floating_point()
        if self.mode is None:
            self.mode = self._decide(x, by_index)
        n_graphs = target.shape[0]
        if self.mode == "captured":
            if not by_index:
                Etgt = _target_index(Etgt)                           # one entry per edge column (the collate's layout)
            x, edge_feat, Esrc, Etgt, batch, _ = pad_batch(x, edge_feat, Esrc, Etgt, batch, n_graphs=n_graphs)
            if self._captured is None:
                self._captured = CapturedQCStep(self.model, self.opt, self.criterion, exchange=self.exchange)
            return self._captured(x, edge_feat, Esrc, Etgt, batch, target, n_graphs=n_graphs + 1)
        if self.mode == "prepared":
            if not by_index:
                Etgt = _target_index(Etgt)                           # one entry per edge column (the collate's layout)
            Etgt, batch = prepare(Esrc, Etgt, batch, x.shape[0], n_graphs)
        # Without an exchange the gradients are DROPPED, not zeroed: autograd then hands every parameter its gradient
        # tensor as it is (no fill launch before the step, no `grad += new` launch after it - ~2 x 25 launches of a
        # ~300-launch step); optim.Adam takes the new addresses by value in its kernel arguments.  With an exchange the
        # gradients are views of parallel.GradBucket's flat buffer and have to stay in place.
        self.opt.zero_grad(set_to_none=self.exchange is None)
        loss = self.criterion(self.model(x, edge_feat, Esrc, Etgt, batch), target)
        loss.backward()
        if self.exchange is not None:
            self.exchange()
        self.opt.step()
        return loss.detach()
