"""Deterministic, platform-independent state_dict filler shared by the fixture generator and the tests.

The 2D branch has 46 M parameters: the wiring fixture cannot carry them, so both sides regenerate them from the key
name (numpy ``default_rng`` seeded with crc32(key) is bit-reproducible everywhere) and the fixture holds only inputs
and the reference composition's outputs.
"""
import zlib

import numpy as np
import torch


def fill_state_dict(sd):
    """Returns a new dict with the same keys/shapes/dtypes as ``sd`` and key-seeded values."""
    out = {}
    for k in sorted(sd.keys()):
        v = sd[k]
        if not v.dtype.is_floating_point:
            out[k] = v.clone()
            continue
        g = np.random.default_rng(zlib.crc32(k.encode()))
        shape = tuple(v.shape)
        if k.endswith("running_var"):
            a = g.uniform(0.5, 1.5, shape)
        elif k.endswith("running_mean"):
            a = g.normal(0, 0.1, shape)
        elif v.dim() >= 2:
            fan_in = int(np.prod(shape[1:]))
            a = g.normal(0, np.sqrt(2.0 / fan_in), shape)
        elif k.endswith("weight"):
            a = g.uniform(0.5, 1.5, shape)
        else:
            a = g.normal(0, 0.1, shape)
        out[k] = torch.from_numpy(a.astype(np.float32)).to(v.dtype)
    return out
