"""Initial weights as a deterministic function of the parameter NAMES: the same tensors for the reference's classes
(tests/golden/make_golden.py, in the build container) and for ours (the tests, on the GPU box) without storing them.
U(-b, b) with b = 1 / sqrt(last dimension), from a CPU generator seeded by a hash of the name."""
import hashlib
import math

import torch


def fill_by_name(module):
    with torch.no_grad():
        for name, p in sorted(module.state_dict().items()):
            if not p.is_floating_point():
                continue
            seed = int.from_bytes(hashlib.sha256(name.encode()).digest()[:6], "little")
            g = torch.Generator().manual_seed(seed)
            bound = 1.0 / math.sqrt(max(p.shape[-1], 1)) if p.dim() > 0 else 1.0
            p.copy_((torch.rand(p.shape, generator=g, dtype=torch.float32) * 2 - 1) * bound)
    return module
