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


def _capturing(p):
    return p.is_cuda and torch.cuda.is_current_stream_capturing()


class GradBucket:
    """Gradient exchange of data-parallel training: ONE persistent flat fp32 buffer, cut into buckets of about
    `bucket_bytes` in REVERSE registration order (the order backward produces gradients in), each bucket all-reduced
    by one collective.  The parameters' `.grad` tensors are VIEWS of the buffer (autograd accumulates into an existing
    `.grad` in place), so nothing is copied in or out around the collectives as long as the optimiser keeps them
    (`zero_grad(set_to_none=False)`); a gradient that was dropped or replaced is copied in and re-attached.

    overlap=True (default): a post-accumulate hook on every parameter marks its bucket; a bucket's all-reduce is
    launched (async) once its last gradient has been written AND every bucket before it has been launched, i.e. WHILE
    backward is still computing the earlier layers - SURVEY.md section 8(e): the 57 MB gradient of the QC model is
    dominated by the edge encoder, whose bucket can travel while the message rounds are still in backward.  One
    backward per step in this mode (gradient accumulation over several backward calls: overlap=False).
    `allreduce_mean()` / `allreduce_sum()` finish the step: they launch what is still pending, wait, and scale.

    Launch order is the bucket index order ON EVERY RANK, whatever order the gradients arrive in and whichever
    parameters received none on this rank: RCCL pairs collectives by issue order, so a bucket that is complete here
    but behind one that is still open (a parameter unused on this rank) waits for `_finish`, which launches the rest
    in index order.  (Round 2 launched a bucket the moment it completed: ranks with different unused parameters then
    issued equal-sized buckets in different orders and summed the wrong slices - tests/test_parallel_gloo.py,
    `test_rank_local_parameter_alone_in_a_middle_bucket`.)

    Inside a HIP-graph capture (qc_step.CapturedQCStep) the hooks only record which parameters were written: no
    re-pointing of `.grad`, no collective and no host read may be baked into the captured region; the exchange then
    runs after the replay (`allreduce_mean(assume_all=True)`).

    Parameters without a gradient: counted as zero on this rank; a parameter that received no gradient on ANY rank
    gets `.grad = None` back, as in a single-process run (the optimiser then skips it instead of applying weight
    decay / momentum to it) - every bucket's slice ends with one flag per parameter, summed by the same collective."""

    def __init__(self, module, bucket_bytes=16 << 20, overlap=True):
        self.params = [p for p in module.parameters() if p.requires_grad]
        dev = self.params[0].device if self.params else torch.device("cpu")
        # buckets in reverse registration order; the slice of a bucket in the flat buffer is its gradients followed by
        # one "received a gradient" flag per parameter, so that ONE collective per bucket carries both
        groups, cur, size = [], [], 0
        for i in reversed(range(len(self.params))):
            cur.append(i)
            size += self.params[i].numel() * 4
            if size >= bucket_bytes:
                groups.append(cur)
                cur, size = [], 0
        if cur:
            groups.append(cur)
        total = sum(sum(self.params[i].numel() for i in g) + len(g) for g in groups)
        self.buf = torch.zeros(total, dtype=torch.float32, device=dev)
        self.views = [None] * len(self.params)
        self.buckets = []                     # (lo, hi_grads, hi, [param indices]) over self.buf, in launch order
        o = 0
        for g in groups:
            lo = o
            for i in g:
                p = self.params[i]
                v = self.buf[o:o + p.numel()].view_as(p)
                if p.grad is not None:
                    v.copy_(p.grad)
                p.grad = v
                self.views[i] = v
                o += p.numel()
            self.buckets.append((lo, o, o + len(g), g))
            o += len(g)
        self._bucket_of = {i: b for b, (_, _, _, idx) in enumerate(self.buckets) for i in idx}
        self.overlap = bool(overlap)
        self._reset()
        # the hooks also tell which parameters received a gradient in this step (a kept `.grad` view looks the same
        # whether or not backward wrote to it), so they are registered in both modes
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(i)) for i, p in enumerate(self.params)]

    @property
    def flat(self):
        """The exchanged buffer (gradients + per-parameter flags)."""
        return self.buf

    def _reset(self):
        self._ready = [0] * len(self.buckets)
        self._seen = [False] * len(self.params)
        self._work = [None] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._next = 0                        # first bucket not launched yet: launches go strictly in index order
        self.launch_log = []                  # bucket indices in issue order of the current step (tests read it)

    def _attach(self, i):
        p, v = self.params[i], self.views[i]
        if p.grad is None:
            return False
        if p.grad.data_ptr() != v.data_ptr():
            v.copy_(p.grad)
            p.grad = v
        return True

    def _make_hook(self, i):
        def hook(_param):
            if _alone() or self._seen[i]:
                return
            self._seen[i] = True
            if not self.overlap or _capturing(_param) or (_param.is_cuda and dist.get_backend() == "gloo"):
                # gloo with GPU tensors is the one-box rehearsal (ranks sharing a GPU, exchange staged through the host):
                # its blocking host round trip stays on the main thread (_finish), not on autograd's worker thread
                return
            self._attach(i)
            b = self._bucket_of[i]
            self._ready[b] += 1
            # drain: bucket b goes out only once 0 .. b-1 have gone out; a completed bucket behind an open one waits
            while self._next < len(self.buckets) and self._ready[self._next] == len(self.buckets[self._next][3]):
                self._launch(self._next)
        return hook

    def _launch(self, b):
        lo, hg, hi, idx = self.buckets[b]
        flags = [1.0 if (self._seen[i] and self._attach(i)) else 0.0 for i in idx]
        for i, f in zip(idx, flags):
            if not f:
                self.views[i].zero_()                     # no gradient here: contributes zero to the sum
        if all(flags):
            self.buf[hg:hi].fill_(1.0)
        else:
            self.buf[hg:hi].copy_(torch.tensor(flags, dtype=torch.float32))
        chunk = self.buf[lo:hi]
        assert b == self._next, "bucket all-reduces are issued in index order on every rank"
        self._launched[b] = True
        self._next = b + 1
        self.launch_log.append(b)
        if dist.get_backend() == "gloo" and chunk.is_cuda:      # one-box rehearsal: staged through the host
            host = chunk.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            chunk.copy_(host)
        else:
            self._work[b] = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, async_op=True)

    def allreduce_mean(self, assume_all=False):
        """grad <- mean over ranks.  Parameters without a gradient on this rank count as zero.  assume_all: every
        parameter that HAS a `.grad` received it in this step although no hook fired (gradients written by a HIP-graph
        replay, qc_step.py)."""
        self._finish(True, assume_all)

    def allreduce_sum(self, assume_all=False):
        """grad <- sum over ranks: the ranks hold partial sums of ONE model's gradient (partition.py)."""
        self._finish(False, assume_all)

    def _finish(self, mean, assume_all=False):
        if _alone():
            return
        if assume_all:
            self._seen = [p.grad is not None for p in self.params]
        for b in range(self._next, len(self.buckets)):
            self._launch(b)
        for w in self._work:
            if w is not None:
                w.wait()
        if mean:
            self.buf.div_(dist.get_world_size())
        if all(self._seen):
            # every parameter has a gradient on THIS rank, hence somewhere: no need to read the flags back (the common
            # case stays free of host synchronisation)
            for i, p in enumerate(self.params):
                p.grad = self.views[i]
        else:
            flags = torch.cat([self.buf[hg:hi] for (_, hg, hi, _) in self.buckets]).tolist()
            k = 0
            for (_, _, _, idx) in self.buckets:
                for i in idx:
                    self.params[i].grad = self.views[i] if flags[k] > 0 else None
                    k += 1
        self._reset()


def run_timed(step, steps, warmup, device=None):
    """The timing contract of bench.py / tools/qc_bench.py: `warmup` untimed steps, then exactly `steps` steps
    bracketed by barrier + device synchronisation on both sides; returns the MAX over ranks of the elapsed seconds
    (and the value of the last step).  Works without a process group (one rank) and on CPU tensors (gloo tests)."""
    import time
    use_dist = dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)
    cuda = device is not None and torch.device(device).type == "cuda"

    def barrier():
        if use_dist:
            dist.barrier()
        if cuda:
            torch.cuda.synchronize(device)
    last = None
    for _ in range(warmup):
        last = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if cuda else "cpu")
        if dist.get_backend() == "gloo" and t.is_cuda:
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, last


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items work units (graphs, runs) for `rank`."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
