"""Latent input pipeline: readers for the reference's two on-disk latent formats and a
pinned-memory prefetcher that overlaps the host->device copy with the training step.

Reference formats (both store the VAE *moments* [2C, R, R] float32 = mean | logvar, so that `utils.sample`
draws a fresh latent every step, train.py:203):
  * LMDB (train_utils/datasets.py:240-304): keys `z-{i}` = raw float32 bytes of the moments, `y-{i}` = the class
    index as utf-8 text, `length`; one environment per split under `<path>/<split>`.
  * WebDataset tar shards (train_wds.py:58-97): per sample a member `<key>.latent` = pickle of the numpy moments and
    `<key>.cls` = the class index as utf-8 text; shards are split across ranks (`nodesplitter`), shuffled through a
    buffer and batched without partial batches.

Feeding rate: 3 850 img/s/GPU x 32 KiB (256^2) = 126 MB/s, 0.5 GB/s at 512^2 -- far below PCIe, so ONE reader
thread + double-buffered pinned staging suffices; what matters is that the copy never sits on the compute
stream.  The device side of the pipeline (sample() + class-dropout + one-hot) is `maskdit_amd.latents`.

`lmdb` / `webdataset` are not dependencies: the LMDB reader imports `lmdb` lazily (and says so when it is
missing); the shard reader uses only `tarfile` + `pickle` from the standard library.
"""
from __future__ import annotations

import glob
import os
import pickle
import queue
import random
import tarfile
import threading
from typing import Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch


class SyntheticMoments:
    """Endless synthetic batches of the dataset's shape: mean ~ 2.745 * N(0,1), logvar = -10, so that
    sample() has std ~ sigma_data = 0.5; labels uniform over the classes.  Host-side (exercises the same prefetch
    path as the file readers)."""

    def __init__(self, batch: int, channels: int, resolution: int, num_classes: int, seed: int = 0):
        self.batch, self.C, self.R, self.num_classes = batch, channels, resolution, num_classes
        self.rng = np.random.default_rng(seed)

    def __iter__(self) -> Iterator[Tuple[np.ndarray, np.ndarray]]:
        while True:
            mean = (2.745 * self.rng.standard_normal((self.batch, self.C, self.R, self.R))).astype(np.float32)
            mom = np.concatenate([mean, np.full_like(mean, -10.0)], axis=1)
            yield mom, self.rng.integers(0, self.num_classes, (self.batch,), dtype=np.int64)


class WdsTarLatents:
    """Batches from WebDataset-style tar shards in the reference's layout (train_wds.py:58-97).

    shards      : directory, glob pattern or list of .tar paths
    rank/world  : shard split `shards[rank::world]` (train_wds.py `nodesplitter`)
    shuffle_buf : sample-level shuffle buffer (the reference uses 1000, initial 100); 0 = file order
    partial batches are dropped (`.batched(batch_size, partial=False)`); `epochs=None` repeats forever."""

    def __init__(self, shards, batch: int, rank: int = 0, world: int = 1, shuffle_buf: int = 1000, seed: int = 0,
                 epochs: Optional[int] = 1):
        if isinstance(shards, (list, tuple)):
            paths = list(shards)
        elif os.path.isdir(shards):
            paths = sorted(glob.glob(os.path.join(shards, '**', '*.tar'), recursive=True))
        else:
            paths = sorted(glob.glob(shards))
        if len(paths) < world:
            raise ValueError(f'{len(paths)} shard(s) for {world} rank(s): every rank needs at least one (train_wds.py:50-55)')
        self.paths = paths[rank::world]
        self.batch, self.shuffle_buf, self.epochs = batch, shuffle_buf, epochs
        self.rng = random.Random(seed + rank)

    @staticmethod
    def _samples(path: str) -> Iterator[Tuple[np.ndarray, int]]:
        """Members of one sample share the key (name up to the first dot of the base name) and are adjacent."""
        key, latent, label = None, None, None
        with tarfile.open(path, 'r') as tf:
            for m in tf:
                if not m.isfile():
                    continue
                base = os.path.basename(m.name)
                k, _, ext = base.partition('.')
                k = os.path.join(os.path.dirname(m.name), k)
                if k != key:
                    if latent is not None and label is not None:
                        yield latent, label
                    key, latent, label = k, None, None
                data = tf.extractfile(m).read()
                if ext == 'latent':
                    latent = np.asarray(pickle.loads(data), dtype=np.float32)  # train_wds.py:60 decode_data
                elif ext == 'cls':
                    label = int(data.decode('utf-8'))
            if latent is not None and label is not None:
                yield latent, label

    def _stream(self) -> Iterator[Tuple[np.ndarray, int]]:
        ep = 0
        while self.epochs is None or ep < self.epochs:
            order = list(self.paths)
            if self.shuffle_buf:
                self.rng.shuffle(order)
            for p in order:
                yield from self._samples(p)
            ep += 1

    def __iter__(self) -> Iterator[Tuple[np.ndarray, np.ndarray]]:
        buf: List[Tuple[np.ndarray, int]] = []
        xs, ys = [], []
        for s in self._stream():
            if self.shuffle_buf:
                buf.append(s)
                if len(buf) < self.shuffle_buf:
                    continue
                s = buf.pop(self.rng.randrange(len(buf)))
            xs.append(s[0])
            ys.append(s[1])
            if len(xs) == self.batch:
                yield np.stack(xs), np.asarray(ys, dtype=np.int64)
                xs, ys = [], []
        self.rng.shuffle(buf)
        for s in buf:
            xs.append(s[0])
            ys.append(s[1])
            if len(xs) == self.batch:
                yield np.stack(xs), np.asarray(ys, dtype=np.int64)
                xs, ys = [], []


class LmdbLatents:
    """Batches from the reference's latent LMDB (train_utils/datasets.py:240-304): records `z-{i}` / `y-{i}`,
    rank-strided, reshuffled every epoch.  Needs the `lmdb` module."""

    def __init__(self, path: str, batch: int, resolution: int, split: str = 'train', rank: int = 0, world: int = 1, seed: int = 0,
                 epochs: Optional[int] = 1):
        try:
            import lmdb  # noqa: F401
        except ImportError as e:
            raise ImportError('LmdbLatents needs the `lmdb` module (the reference\'s ImageNetLatentDataset does too); '
                              'convert to tar shards or install it') from e
        import lmdb
        self.env = lmdb.open(os.path.join(path, split), readonly=True, lock=False, create=False)
        self.txn = self.env.begin(write=False)
        self.length = int(self.txn.get(b'length').decode('utf-8'))
        self.batch, self.R, self.rank, self.world, self.epochs = batch, resolution, rank, world, epochs
        self.rng = np.random.default_rng(seed)

    def __iter__(self):
        ep = 0
        while self.epochs is None or ep < self.epochs:
            order = self.rng.permutation(self.length)[self.rank::self.world]
            for i in range(0, len(order) - self.batch + 1, self.batch):
                xs, ys = [], []
                for idx in order[i:i + self.batch]:
                    z = np.frombuffer(self.txn.get(f'z-{idx}'.encode()), dtype=np.float32).reshape(-1, self.R, self.R)
                    xs.append(z)
                    ys.append(int(self.txn.get(f'y-{idx}'.encode()).decode('utf-8')))
                yield np.stack(xs), np.asarray(ys, dtype=np.int64)
            ep += 1


class LatentPrefetcher:
    """Host -> HBM feeder.  A reader thread pulls (moments, labels) numpy batches from `source`, copies them into
    one of `depth` PINNED staging slots and enqueues an asynchronous H2D copy on a dedicated copy stream; the consumer
    gets device tensors whose readiness the COMPUTE stream waits on with an event (no host synchronisation), and the
    slot is recycled once the compute stream has passed the point where the batch was last used (the consumer calls
    next() again).  On a CPU `device` it degrades to plain tensors (host-logic tests).

    Yields (moments f32 [B, 2C, R, R] on `device`, labels int64 [B] on `device`)."""

    def __init__(self, source: Iterable, device, depth: int = 3):
        self.source, self.device, self.depth = source, torch.device(device), max(2, depth)
        self.cuda = self.device.type == 'cuda'
        self.q: 'queue.Queue' = queue.Queue(maxsize=self.depth - 1)
        self.free: 'queue.Queue' = queue.Queue()
        self.slots: List[Optional[dict]] = [None] * self.depth
        for i in range(self.depth):
            self.free.put(i)
        self.copy_stream = torch.cuda.Stream(self.device) if self.cuda else None
        self.err: Optional[BaseException] = None
        self._stop = False
        self._last: Optional[int] = None
        self.thread = threading.Thread(target=self._reader, name='latent-prefetch', daemon=True)
        self.thread.start()

    def _slot(self, i: int, mom: np.ndarray, lab: np.ndarray) -> dict:
        s = self.slots[i]
        if s is None or s['hm'].shape != mom.shape:
            hm, hl = torch.from_numpy(np.empty(mom.shape, np.float32)), torch.from_numpy(np.empty(lab.shape, np.int64))
            if self.cuda:
                hm, hl = hm.pin_memory(), hl.pin_memory()
            s = {'hm': hm, 'hl': hl,
                 'dm': torch.empty(mom.shape, dtype=torch.float32, device=self.device),
                 'dl': torch.empty(lab.shape, dtype=torch.int64, device=self.device),
                 'ready': torch.cuda.Event() if self.cuda else None, 'done': torch.cuda.Event() if self.cuda else None}
            self.slots[i] = s
        return s

    def _reader(self):
        try:
            for mom, lab in self.source:
                if self._stop:
                    return
                i = self.free.get()
                if self._stop:
                    return
                s = self._slot(i, mom, lab)
                if self.cuda and s['done'].query() is False:
                    s['done'].synchronize()  # the compute stream is still reading this slot's device tensors
                s['hm'].copy_(torch.from_numpy(np.ascontiguousarray(mom, dtype=np.float32)))
                s['hl'].copy_(torch.from_numpy(np.ascontiguousarray(lab, dtype=np.int64)))
                if self.cuda:
                    with torch.cuda.stream(self.copy_stream):
                        s['dm'].copy_(s['hm'], non_blocking=True)
                        s['dl'].copy_(s['hl'], non_blocking=True)
                        s['ready'].record(self.copy_stream)
                else:
                    s['dm'].copy_(s['hm'])
                    s['dl'].copy_(s['hl'])
                self.q.put(i)
        except BaseException as e:  # noqa: BLE001  (surfaced in the consumer)
            self.err = e
        finally:
            self.q.put(None)

    def __iter__(self):
        return self

    def __next__(self):
        if self._last is not None:  # the previous batch has been consumed by kernels already enqueued
            if self.cuda:
                self.slots[self._last]['done'].record(torch.cuda.current_stream(self.device))
            self.free.put(self._last)
            self._last = None
        i = self.q.get()
        if i is None:
            self.q.put(None)
            if self.err is not None:
                raise self.err
            raise StopIteration
        s = self.slots[i]
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_event(s['ready'])
        self._last = i
        return s['dm'], s['dl']

    def close(self):
        self._stop = True
        try:
            self.free.put_nowait(0)
        except Exception:  # noqa: BLE001
            pass
        while True:  # unblock a reader waiting on q.put
            try:
                self.q.get_nowait()
            except queue.Empty:
                break


def write_wds_shard(path: str, moments: Sequence[np.ndarray], labels: Sequence[int], start_index: int = 0):
    """Write samples in the reference's shard layout (`<key>.latent` pickle + `<key>.cls` text): used by the tests
    and by anyone converting an LMDB."""
    import io
    with tarfile.open(path, 'w') as tf:
        for j, (z, y) in enumerate(zip(moments, labels)):
            key = f'{start_index + j:09d}'
            for ext, data in (('latent', pickle.dumps(np.asarray(z, dtype=np.float32))), ('cls', str(int(y)).encode('utf-8'))):
                ti = tarfile.TarInfo(f'{key}.{ext}')
                ti.size = len(data)
                tf.addfile(ti, io.BytesIO(data))
