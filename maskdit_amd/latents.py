"""Host mirror of the two small per-batch transforms in front of the loss (train.py:203,208-209):
`utils.sample` (moments -> scaled latent) and label dropout.  The random numbers are drawn with
torch in the reference's order; the arithmetic runs in HIP."""
from __future__ import annotations

import torch

from . import _lib
from ._lib import call


def _st():
    return torch.cuda.current_stream().cuda_stream


def sample(moments: torch.Tensor, scale_factor: float = 0.18215) -> torch.Tensor:
    """utils.py:59-65.  moments [B, 2C, H, W] fp32 -> z [B, C, H, W]."""
    if not moments.is_cuda:
        raise _lib.MaskDiTLibError('maskdit_amd.sample: moments are not on a HIP device; there is no CPU path')
    moments = moments.float().contiguous()
    B, C2, H, W = moments.shape
    assert C2 % 2 == 0
    rn = torch.randn(B, C2 // 2, H, W, device=moments.device)  # == torch.randn_like(mean)
    z = torch.empty_like(rn)
    call('mdt_sample_moments', moments.data_ptr(), rn.data_ptr(), z.data_ptr(), B, (C2 // 2) * H * W, float(scale_factor), _st())
    return z


def class_dropout_(y: torch.Tensor, class_dropout_prob: float) -> torch.Tensor:
    """train.py:208-209 in place: y = y * (torch.rand(B, 1) >= p)."""
    if not y.is_cuda:
        raise _lib.MaskDiTLibError('maskdit_amd.class_dropout_: labels are not on a HIP device; there is no CPU path')
    assert y.dtype == torch.float32 and y.is_contiguous() and y.dim() == 2
    u = torch.rand(y.shape[0], 1, device=y.device)
    call('mdt_class_dropout', y.data_ptr(), u.data_ptr(), float(class_dropout_prob), y.shape[0], y.shape[1], _st())
    return y
