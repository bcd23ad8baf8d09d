"""Deterministic parameters for the mask-network parity tests: the same numbers for the reference module (in
tests/golden/make_golden_cgnet.py) and for ours, derived per state_dict key from a seed -- so the fixture holds only
inputs and expected outputs, not a megabyte of weights."""
import zlib

import numpy as np
import torch


def seeded_state(module, seed):
    """Fill every entry of module.state_dict() from numpy RNG streams keyed by (seed, name); returns the dict loaded."""
    state = {}
    for key, ref in module.state_dict().items():
        rng = np.random.default_rng([seed, zlib.crc32(key.encode())])
        shape = tuple(ref.shape)
        if key.endswith("num_batches_tracked"):
            v = np.zeros(shape, np.int64)
        elif key.endswith("running_var") or key.endswith("bn.weight"):
            v = rng.uniform(0.5, 1.5, shape)
        elif key.endswith("running_mean") or key.endswith("bias"):
            v = rng.normal(0.0, 0.1, shape)
        elif key.endswith("act.weight"):
            v = rng.uniform(0.1, 0.4, shape)
        else:                                           # conv / linear weights
            fan_in = int(np.prod(shape[1:]))
            v = rng.normal(0.0, np.sqrt(2.0 / fan_in), shape)
        state[key] = torch.from_numpy(np.asarray(v)).to(ref.dtype)
    module.load_state_dict(state)
    return state


def probe_positions(shape, key, n=48):
    """A fixed set of flat positions per tensor at which the goldens keep gradient values."""
    size = int(np.prod(shape))
    rng = np.random.default_rng([99, zlib.crc32(key.encode())])
    return np.sort(rng.choice(size, size=min(n, size), replace=False))
