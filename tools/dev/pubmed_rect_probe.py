#!/usr/bin/env python3
"""Pubmed rk4 step: gradient error against the fp64 oracle with the rectangular layer products on (a) csrc/rect.hip,
(b) the library GEMM (torch.mm patched in) - is the rect path the source of the 1e-4 deviation of the ODE weight?"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_gcn as T  # noqa: E402
from graph_odenet_amd import models, ops  # noqa: E402


def golden(name):
    return dict(np.load(os.path.join(ROOT, "tests", "golden", name)))


def main():
    dev = torch.device("cuda:0")
    adj, feats, labels, idx, ncls = T._citation(golden, "pubmed")
    torch.manual_seed(7)
    m = models.ODEGCN3(nfeat=feats.shape[1], nhid=128, nclass=ncls, dropout=0.0, method="rk4", step_size=1 / 16)
    with torch.no_grad():
        m.gc2.odefunc.norm1.weight.uniform_(0.5, 1.5)
        m.gc2.odefunc.norm1.bias.uniform_(-0.5, 0.5)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    opts = {"step_size": 1 / 16}
    ref_out, ref_g, _ = T.oracle_odegcn3(sd, feats, adj, "rk4", opts, 1e-5, labels, idx)
    out64, g64, _ = T.oracle_odegcn3(sd, feats, adj, "rk4", opts, 1e-5, labels, idx, dtype=torch.float64)
    m = m.to(dev)
    real = (ops.rect_gemm, ops.rect_gemm_nt, ops.rect_wgrad)

    def lib_gemm(x, W, pad_to=None):
        y = torch.mm(x, W)
        return y if pad_to in (None, W.shape[1]) else torch.nn.functional.pad(y, (0, pad_to - W.shape[1]))
    for name in ("rect", "library", "rect"):
        if name == "library":
            ops.rect_gemm, ops.rect_gemm_nt, ops.rect_wgrad = lib_gemm, (lambda dS, W: torch.mm(dS, W.t())), (lambda x, dS: torch.mm(x.t(), dS))
        else:
            ops.rect_gemm, ops.rect_gemm_nt, ops.rect_wgrad = real
        m.zero_grad(set_to_none=True)
        out = m(feats.to(dev), adj.to(dev))
        torch.nn.functional.nll_loss(out[idx.to(dev)], labels.to(dev)[idx.to(dev)]).backward()
        print(name, "logits err %.3e (oracle32 %.3e)" % ((out.detach().cpu().double() - out64).abs().max(), (ref_out.double() - out64).abs().max()))
        for k, p in m.named_parameters():
            e_got = (p.grad.cpu().double() - g64[k]).abs().max().item()
            e_ref = (ref_g[k].double() - g64[k]).abs().max().item()
            print("   %-28s got %.3e  oracle32 %.3e  |g| %.3e" % (k, e_got, e_ref, g64[k].abs().max().item()))
    # how much of that is the conditioning of the problem?  the LIBRARY path with the first layer's output perturbed by
    # one part in 10^7 (what any other fp32 summation order of X W does to it)
    ops.rect_gemm, ops.rect_gemm_nt, ops.rect_wgrad = lib_gemm, (lambda dS, W: torch.mm(dS, W.t())), (lambda x, dS: torch.mm(x.t(), dS))
    gen = torch.Generator(device=dev).manual_seed(0)
    for trial in range(6):
        m.zero_grad(set_to_none=True)
        hook = m.gc1.register_forward_hook(lambda mod, inp, out: out * (1 + 1e-7 * torch.randn(out.shape, generator=gen, device=dev)))
        out = m(feats.to(dev), adj.to(dev))
        hook.remove()
        torch.nn.functional.nll_loss(out[idx.to(dev)], labels.to(dev)[idx.to(dev)]).backward()
        k = "gc2.odefunc.gc1.weight"
        print("library + 1e-7 noise on gc1's output, trial %d: %s err %.3e" % (trial, k, (dict(m.named_parameters())[k].grad.cpu().double() - g64[k]).abs().max().item()))
    ops.rect_gemm, ops.rect_gemm_nt, ops.rect_wgrad = real


if __name__ == "__main__":
    main()
