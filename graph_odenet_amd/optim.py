"""`optimizer.step()` of the reference's training loops (QC/train_egcn.py, GCN/train_res.py:97: torch.optim.Adam) as ONE
launch over the whole parameter list (csrc/mlp.hip: gode_adam_f32).

    opt = graph_odenet_amd.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-4)

Same update as torch.optim.Adam (no amsgrad / maximize; L2 weight decay folded into the gradient; bias corrections
from the step count), same constructor keywords, `zero_grad`, `param_groups` (lr can be changed between steps) and a
`state` with torch's keys (`step`, `exp_avg`, `exp_avg_sq`), so that checkpoints written with `state_dict()` load into
torch.optim.Adam and back.  At QM9 batch sizes a step of the 14.3 M-parameter model spends 0.33 ms in torch's
multi-tensor Adam (~30 launches with foreach); here it is two launches, and both are kernels, so the step can be
captured into a HIP graph (qc_step.CapturedQCStep: the step counter lives on the device and is advanced by the launch).
Parameters whose `.grad` is None are skipped, as torch does.  GPU float32 parameters only: there is no CPU path.

One deviation from torch.optim.Adam: the step count (hence the bias corrections) is kept PER PARAMETER GROUP, on the
device, not per parameter.  A parameter that receives its first gradient k steps after the others of its group (a head
that joins late, a rank-local unused parameter) is corrected with the group's count where torch would start it at 1,
and its exported `state['step']` is the group's count.  Every model of the reference gives all parameters a gradient
in every step, so the two coincide there; put late joiners in their own param group to get torch's behaviour.
"""
import ctypes

import torch

from . import _lib
from ._lib import check, stream_ptr


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("Adam: invalid hyper-parameter")
        # capturable: the update reads its step count from device memory (what qc_step.CapturedQCStep asks for)
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, capturable=True))
        self._plans = {}          # group index -> (key, [(AdamArgs, n_tensors, chunks tensor, n_chunks)], state tensor)

    def _plan(self, gi, group, active):
        """Launch plan of one group for the parameters that have a gradient: argument blocks of <= 64 tensors with their
        chunk lists.  Rebuilt when the set of tensors or one of their addresses changes (host-side bookkeeping only)."""
        lib = _lib.load()
        # the argument blocks bake in the addresses of all four arrays of a tensor: all four are part of the key.  The
        # gradients are keyed apart: a loop that DROPS its gradients between steps (zero_grad(set_to_none=True),
        # qc_train.TrainStep) brings new gradient tensors every step, and then only their addresses are patched into the
        # blocks - rebuilding the plan (a 1 750-entry work list for the 14.3 M parameters of the QC models) every step cost
        # more host time than the launches the dropped gradients saved
        key = tuple((p.data_ptr(), p.numel(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr())
                    for p in active)
        gkey = tuple(p.grad.data_ptr() for p in active)
        hit = self._plans.get(gi)
        dev = active[0].device
        if hit is not None and hit[0] == key:
            if hit[3] != gkey:
                for idx, gp in enumerate(gkey):
                    hit[1][idx // _lib.GODE_ADAM_MAX_TENSORS][0].grad[idx % _lib.GODE_ADAM_MAX_TENSORS] = gp
                hit = (key, hit[1], hit[2], gkey)
                self._plans[gi] = hit
            return hit
        state_t = hit[2] if hit is not None else None
        if state_t is None:
            step0 = 0.0
            for p in group["params"]:                       # a loaded state_dict brings its own step count
                st = self.state.get(p)
                if st and "step" in st:
                    step0 = float(st["step"])
                    break
            state_t = torch.tensor([step0, 0.0, 0.0], dtype=torch.float32, device=dev)
        chunk = lib.gode_adam_chunk()
        blocks = []
        for lo in range(0, len(active), _lib.GODE_ADAM_MAX_TENSORS):
            part = active[lo:lo + _lib.GODE_ADAM_MAX_TENSORS]
            args = _lib.AdamArgs()
            rows = []
            for i, p in enumerate(part):
                st = self.state[p]
                args.param[i], args.grad[i] = p.data_ptr(), p.grad.data_ptr()
                args.exp_avg[i], args.exp_avg_sq[i] = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                args.len[i] = p.numel()
                rows += [(i, s) for s in range(0, p.numel(), chunk)]
            # AdamChunk {int32 tensor; int32 0; int64 start} = two little-endian int64 words
            ck = getattr(self, "_chunk_cache", {}).get(tuple(p.numel() for p in part))
            if ck is None or ck.device != dev:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("optim.Adam: run one step outside the HIP-graph capture first (it uploads the work list)")
                ck = torch.tensor(rows, dtype=torch.int64).reshape(-1, 2).to(dev)
                self.__dict__.setdefault("_chunk_cache", {})[tuple(p.numel() for p in part)] = ck
            blocks.append((args, len(part), ck, len(rows)))
        plan = (key, blocks, state_t, gkey)
        self._plans[gi] = plan
        return plan

    def load_state_dict(self, state_dict):
        """torch's loader replaces the moment tensors and the step counts: every launch plan (which holds their
        addresses and the device step counter) is dropped, so that the next step re-seeds the counter from the loaded
        `step` and points the kernel at the loaded moments."""
        super().load_state_dict(state_dict)
        for g in self.param_groups:
            g["capturable"] = True        # a torch.optim.Adam checkpoint says False; this update always reads a device counter
        self._plans.clear()

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        if hasattr(self, "_plans"):
            self._plans.clear()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            active = [p for p in group["params"] if p.grad is not None]
            if not active:
                continue
            for p in active:
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()) or p.grad.is_sparse \
                        or not p.grad.is_contiguous() or p.grad.dtype != torch.float32:
                    raise RuntimeError("graph_odenet_amd.optim.Adam: contiguous float32 GPU parameters and gradients only")
                st = self.state[p]
                if "exp_avg" not in st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            _, blocks, state_t, _ = self._plan(gi, group, active)
            b1, b2 = group["betas"]
            check(lib.gode_adam_tick_f32(ctypes.c_void_p(state_t.data_ptr()), float(b1), float(b2), stream_ptr()),
                  "gode_adam_tick_f32")
            for args, n_t, ck, n_c in blocks:
                check(lib.gode_adam_f32(ctypes.byref(args), n_t, ctypes.c_void_p(ck.data_ptr()), n_c,
                                        ctypes.c_void_p(state_t.data_ptr()), float(group["lr"]), float(b1), float(b2),
                                        float(group["eps"]), float(group["weight_decay"]), stream_ptr()), "gode_adam_f32")
            for p in active:
                self.state[p]["step"] = state_t[0]             # a view of the device counter (torch keeps a tensor here too)
        return loss
