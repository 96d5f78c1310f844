#!/usr/bin/env python3
"""Pubmed (config C2) step time: fused small-graph path on / off (development aid)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import config_bench as cb
from graph_odenet_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
for fused in (1, 0, 1, 0):
    lib.gode_set_option(b"small_fused", fused)
    r = next(cb.c2_pubmed(dev, cpu=False))
    print("small_fused", fused, r["ms_per_step"], r["nfe_f"], r["nfe_b"], flush=True)
lib.gode_set_option(b"small_fused", 1)
