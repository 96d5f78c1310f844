"""A QC training step as ONE HIP-graph replay per shape bucket (BASELINE.json configs[3]).

A batch-20 QM9 step is ~200 small launches: 1.5 ms of host work for ~1.2 ms of kernels, and the per-batch graph
conversion synchronises twice, so host and GPU time add up (3.4 ms measured).  Batches are padded to shape buckets
(qc_batch.pad_batch); for every bucket the whole step - conversion of (Esrc, Etgt, batch) to the kernels' CSR form,
forward, loss, backward, Adam - is captured once on static input buffers and replayed afterwards: a step is a handful of
copies into the static buffers plus one graph launch.  Nothing in the captured region synchronises, allocates outside the
graph's pool, or launches a memset (tests/test_abi.py).

    step = CapturedQCStep(model, optimizer, loss_fn)          # optimizer: torch.optim.Adam(..., capturable=True)
    loss = step(x, edge_feat, Esrc, Etgt, batch, target)      # tensors of a padded batch; returns the loss tensor
                                                              # (Etgt: the dense N x E matrix or the int64[E] target index)

The capture contains arbitrary PyTorch autograd, whose reductions may clear buffers with memset nodes: it is only taken
when hipgraph.memset_nodes_ok() passes in this process (see hipgraph.py: start the process with
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 or call hipgraph.prefer_safe_graphs() before the first GPU call); otherwise every step
runs eagerly and `capture_disabled_reason` says why.

With torch.distributed (one rank per GPU) only forward + backward are captured; the gradient all-reduce and the optimiser
step run eagerly after the replay (`exchange=` callback, e.g. `lambda: bucket.allreduce_mean(assume_all=True)`).  A
`parallel.GradBucket` on the same model must not launch anything inside the capture: its hooks notice a capturing
stream and only record (parallel.py), so either `overlap` setting is safe; the gradients of a captured bucket are
graph-owned tensors, copied into the exchange buffer by `allreduce_mean` after the replay.

`n_graphs` (optional): number of graphs of the batch including the dummy graph of a padded batch; when omitted it is
taken from `qc_batch.prepare`'s annotation of `batch`, else read from `batch[-1]` (one host synchronisation per call).
"""
import torch


class _Bucket:
    def __init__(self, x, ef, Esrc, Etgt, batch, target, n_graphs):
        self.x, self.ef, self.target = torch.empty_like(x), torch.empty_like(ef), torch.empty_like(target)
        self.Esrc, self.batch = torch.empty_like(Esrc), torch.empty_like(batch)
        self.Etgt = torch.empty_like(Etgt)
        self.batch._gode_n_graphs = n_graphs
        self.graph = None
        self.loss = None
        self.seen = 0

    def load(self, x, ef, Esrc, Etgt, batch, target):
        self.x.copy_(x); self.ef.copy_(ef); self.Esrc.copy_(Esrc); self.Etgt.copy_(Etgt); self.batch.copy_(batch)
        self.target.copy_(target)


class CapturedQCStep:
    """See the module docstring.  `loss_fn(output, target)`; outputs are cut to target.shape[0] rows (the dummy graph of
    a padded batch is dropped).  A bucket is run eagerly `warmup` times before it is captured."""

    def __init__(self, model, optimizer, loss_fn, exchange=None, warmup=2):
        self.model, self.opt, self.loss_fn, self.exchange, self.warmup = model, optimizer, loss_fn, exchange, warmup
        self.buckets = {}
        self.capture_optimizer = exchange is None
        self.capture_disabled_reason = None
        if self.capture_optimizer and not all(g.get("capturable", False) for g in optimizer.param_groups):
            raise ValueError("CapturedQCStep: the optimiser must be built with capturable=True")

    def _run(self, b):
        Etgt = b.Etgt
        if Etgt.dim() == 1:            # per-edge target index (qc_batch.prepare): converted without a dense matrix
            from .qc_layers import prepared_edges
            Etgt = prepared_edges(b.Esrc, Etgt, b.x.shape[0])
        out = self.model(b.x, b.ef, b.Esrc, Etgt, b.batch)
        loss = self.loss_fn(out[:b.target.shape[0]], b.target)
        loss.backward()
        if self.capture_optimizer:
            self.opt.step()
        return loss

    def _clear_caches(self):
        from . import qc_layers, qc_models
        qc_layers._cache.clear()
        qc_models._seg_cache.clear()

    def __call__(self, x, ef, Esrc, Etgt, batch, target, n_graphs=None):
        # a bucket's static buffers and captured graph are valid for exactly one layout: shapes, the Etgt FORM (dense
        # N x E matrix or int64[E] index), index dtypes, and the number of graphs incl. the dummy graph of a padded batch
        if n_graphs is None:
            n_graphs = getattr(batch, "_gode_n_graphs", None)       # qc_batch.prepare leaves it on the tensor
        if n_graphs is None:
            n_graphs = target.shape[0] + (1 if batch.numel() and int(batch[-1]) >= target.shape[0] else 0)
        key = (tuple(x.shape), tuple(ef.shape), tuple(target.shape), tuple(Etgt.shape), Etgt.dtype, Esrc.dtype,
               batch.dtype, int(n_graphs))
        b = self.buckets.get(key)
        if b is None:
            b = self.buckets[key] = _Bucket(x, ef, Esrc, Etgt, batch, target, int(n_graphs))
        b.load(x, ef, Esrc, Etgt, batch, target)
        if b.graph is None:
            if self.capture_disabled_reason is None and b.seen >= self.warmup:
                from . import hipgraph
                if not hipgraph.memset_nodes_ok(x.device):
                    import warnings
                    self.capture_disabled_reason = ("replayed memset nodes are unreliable in this process (set %s=0 "
                                                    "before the first GPU call)" % hipgraph.ENV)
                    warnings.warn("graph_odenet_amd: QC steps stay eager - " + self.capture_disabled_reason)
            if b.seen < self.warmup or self.capture_disabled_reason is not None:
                # eager: also lets the libraries pick their kernels for the bucket
                b.seen += 1
                self._clear_caches()
                self.opt.zero_grad(set_to_none=True)
                loss = self._run(b)
                if self.exchange is not None:
                    self.exchange()
                    self.opt.step()
                return loss.detach()
            self._clear_caches()                         # the conversion must run (and be captured) inside the graph
            self.opt.zero_grad(set_to_none=True)         # gradients become graph-owned tensors, rewritten by every replay
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                b.loss = self._run(b).detach()
            b.graph = g
            b.grads = [p.grad for p in self.model.parameters()]
            self._clear_caches()
            # the capture itself executed nothing: run the step it recorded
        else:
            for p, gr in zip(self.model.parameters(), b.grads):
                p.grad = gr                              # another bucket's replay may have re-pointed them
        b.graph.replay()
        if self.exchange is not None:
            self.exchange()
            self.opt.step()
        return b.loss
