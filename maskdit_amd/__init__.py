"""maskdit_amd -- MI355X (gfx950) native engine for the MaskDiT training / sampling hot path.

Only what the hot path needs lives here: `csrc/` (hand-written HIP kernels + the C ABI of
libmaskdit_hip.so, declared in include/maskdit_hip.h) and the host-side mirror of the
reference's Python surface:

    Precond_models['edm'] / EDMPrecond, DiT_models, get_mask     <- models/maskdit.py
    Losses['edm'] / EDMLoss, unwrap_model                        <- train_utils/loss.py, helper.py
    FusedAdam, update_ema                                        <- apex.optimizers, train_utils/helper.py
    edm_sampler                                                  <- sample.py
    DataParallel, ShardedFusedAdam (ZeRO-1)                      <- accelerate / DDP (train.py:178), train.py:226-230
    sample, class_dropout_                                       <- utils.py:59-65, train.py:208-209
    StackedRandomGenerator, seed_batches                         <- utils.py:119-133, sample.py:233-235
    data.{WdsTarLatents, LmdbLatents, LatentPrefetcher}          <- train_wds.py:58-97, train_utils/datasets.py:240-304

There is no non-HIP fallback: computing without libmaskdit_hip.so or off-GPU raises.
"""
__version__ = '0.1.0'

from ._lib import MaskDiTLibError, build  # noqa: F401
from .precond import DiT_models, EDMPrecond, Precond_models, get_mask  # noqa: F401
from .loss import EDMLoss, Losses, unwrap_model  # noqa: F401
from .optim import FusedAdam, update_ema  # noqa: F401
from .sampler import edm_sampler  # noqa: F401
from .ddp import DataParallel, GradSlabReducer  # noqa: F401
from .zero import ShardedFusedAdam  # noqa: F401
from .latents import StackedRandomGenerator, class_dropout_, sample, seed_batches  # noqa: F401
