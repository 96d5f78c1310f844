#!/usr/bin/env python3
"""A second model of the same kind in one process: does its solve get captured like the first one's?  (development aid)"""
import json, os, sys, warnings
import torch
warnings.simplefilter("always")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import config_bench as cb
dev = torch.device("cuda:0")
for name, fn in (("gat 1x16", lambda: next(cb.c3_citeseer_gat(dev, 1, 16, cpu=False))), ("gat 1x16 again", lambda: next(cb.c3_citeseer_gat(dev, 1, 16, cpu=False))),
                 ("gat 8x8", lambda: next(cb.c3_citeseer_gat(dev, 8, 64, cpu=False))), ("gat 1x16 third", lambda: next(cb.c3_citeseer_gat(dev, 1, 16, cpu=False))),
                 ("cora", lambda: next(cb.c1_cora(dev, cpu=False))), ("cora again", lambda: next(cb.c1_cora(dev, cpu=False)))):
    print(name, fn()["ms_per_step"], flush=True)
