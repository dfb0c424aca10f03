"""Host-side helpers of the training loop that are pure Python in the reference:
`get_mask_ratio_fn` (train_utils/helper.py:9-27), `get_one_hot` (:30-33), the lr warm-up
factor (train.py:223-225) and a minimal attribute-dict YAML loader standing in for OmegaConf
(train.py:37; omegaconf is not part of this image)."""
from __future__ import annotations

import math

import torch
import yaml


def get_mask_ratio_fn(name='constant', ratio_scale=0.5, ratio_min=0.0):
    """Mask ratio as a function of training progress x in [0, 1]: cosine^k, exp, linear, constant."""
    span = ratio_scale - ratio_min
    if name.startswith('cosine') and name[6:] in ('2', '3', '4', '5', '6'):
        k = int(name[6:])
        return lambda x: span * math.cos(math.pi * x / 2) ** k + ratio_min
    if name == 'exp':
        return lambda x: span * math.exp(-x * 7) + ratio_min
    if name == 'linear':
        return lambda x: span * x + ratio_min
    if name == 'constant':
        return lambda x: ratio_scale
    raise ValueError('Unknown mask ratio function: {}'.format(name))


def get_one_hot(labels: torch.Tensor, num_classes: int = 1000) -> torch.Tensor:
    out = torch.zeros(labels.shape[0], num_classes, device=labels.device)
    out.scatter_(1, labels.view(-1, 1), 1)
    return out


def lr_rampup_factor(step: int, global_batch: int, lr_rampup_kimg: float) -> float:
    """train.py:223-225: min(step * global_batch / max(rampup_kimg * 1000, 1e-8), 1)."""
    return min(step * global_batch / max(lr_rampup_kimg * 1000, 1e-8), 1.0)


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    __setattr__ = dict.__setitem__


def _wrap(x):
    if isinstance(x, dict):
        return AttrDict({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def load_config(path: str) -> AttrDict:
    """The reference's YAML files (configs/**) parse unchanged; `50_000`-style ints are handled
    by PyYAML, the string 'None' is left as the reference leaves it."""
    with open(path) as f:
        return _wrap(yaml.safe_load(f))
