#!/usr/bin/env python
"""train_wds.py -- the reference's WebDataset training entry point (train_wds.py:103-400; BASELINE configs[3]: XL/2 on
ImageNet-512 latents, 64x64x4, global batch 1024 over 8 GPUs).  Identical step body to train.py; the data source is the
reference's tar-shard layout (`<key>.latent` pickled moments + `<key>.cls`, shards split over the ranks) read by
maskdit_amd.data.WdsTarLatents through the pinned-memory prefetcher, and the integer labels become one-hot rows on the
device (train_wds.py:266 `get_one_hot`).

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_wds.py \
        --config configs/xl2-512-wds.yaml --data_path /data/imagenet512-wds

`data.category` defaults to `wds` here (a YAML saying `webdataset`, as the reference's imagenet512-latent.yaml does, is
accepted); `synthetic` still works for a dry run without shards."""
from __future__ import annotations

import torch.distributed as dist

import train as T


def main(argv=None):
    args = T.parse(argv)
    args.default_category = 'wds'
    T.train_loop(args)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
