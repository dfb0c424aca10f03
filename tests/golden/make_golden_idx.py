"""Index sampler shared by make_golden.py and the tests (no reference import)."""
import numpy as np


def sample_idx(numel, k=64, seed=1234):
    rs = np.random.RandomState(seed + numel % 9973)
    return rs.randint(0, numel, size=k).astype(np.int64)
