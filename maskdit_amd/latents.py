"""Host mirror of the two small per-batch transforms in front of the loss (train.py:203,208-209):
`utils.sample` (moments -> scaled latent) and label dropout -- the random numbers are drawn with
torch in the reference's order, the arithmetic runs in HIP -- and of the per-seed random source of the
sampling entry point (`utils.StackedRandomGenerator`, utils.py:119-133; sample.py:260-264)."""
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


class StackedRandomGenerator:
    """Per-sample-seed random source (utils.py:119-133): sample k of a batch draws from its OWN torch.Generator seeded
    with seeds[k] mod 2^32, so an image depends on its seed only, not on the batch it was generated in.  Same call
    surface as the reference (`randn(size, **kw)`, `randn_like(x)`, `randint(*args, size=..., **kw)`); pinned against
    reference-generated draws by tests/test_host_cpu.py::test_stacked_random_generator_matches_reference."""

    def __init__(self, device, seeds):
        self.device = torch.device(device)
        self.seeds = [int(s) & 0xFFFFFFFF for s in seeds]
        self.generators = []
        for s in self.seeds:
            g = torch.Generator(self.device)
            g.manual_seed(s)
            self.generators.append(g)

    def _per_sample(self, draw, size):
        if size[0] != len(self.generators):
            raise AssertionError(f'leading dimension {size[0]} != number of seeds {len(self.generators)}')
        tail = tuple(size[1:])
        return torch.stack([draw(tail, g) for g in self.generators], dim=0)

    def randn(self, size, **kwargs):
        return self._per_sample(lambda shp, g: torch.randn(shp, generator=g, **kwargs), size)

    def randn_like(self, input):
        return self.randn(input.shape, dtype=input.dtype, layout=input.layout, device=input.device)

    def randint(self, *args, size, **kwargs):
        return self._per_sample(lambda shp, g: torch.randint(*args, size=shp, generator=g, **kwargs), size)


def seed_batches(seeds, max_batch_size: int, rank: int = 0, world: int = 1):
    """sample.py:233-235: the seed list is split into `num_batches` (a multiple of the world size, each at most
    max_batch_size long) contiguous chunks; rank r takes chunks r, r + world, ...  No exchange between ranks."""
    seeds = torch.as_tensor(list(seeds))
    num_batches = ((len(seeds) - 1) // (max_batch_size * world) + 1) * world
    return [b.tolist() for b in seeds.tensor_split(num_batches)[rank::world]]
