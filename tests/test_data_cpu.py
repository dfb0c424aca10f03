"""Host logic of the latent input pipeline (maskdit_amd/data.py): the reference's WebDataset shard layout
(train_wds.py:58-97: `<key>.latent` = pickled moments, `<key>.cls` = class index text) written and read back, rank
split, partial-batch drop, shuffle buffer = a permutation, and the prefetcher's ordering / error propagation.  CPU only."""
import os
import pickle
import tarfile

import numpy as np
import pytest
import torch

from maskdit_amd import data as D


def _make(tmp_path, n_shards=4, per=10, R=8):
    rng = np.random.default_rng(0)
    allz, ally, paths = [], [], []
    for s in range(n_shards):
        z = rng.standard_normal((per, 8, R, R)).astype(np.float32)
        y = rng.integers(0, 1000, per)
        p = os.path.join(tmp_path, f'shard-{s:03d}.tar')
        D.write_wds_shard(p, z, y, start_index=s * per)
        paths.append(p)
        allz.append(z)
        ally.append(y)
    return paths, np.concatenate(allz), np.concatenate(ally)


def test_shard_layout_is_the_references(tmp_path):
    paths, z, y = _make(str(tmp_path), 1, 3)
    with tarfile.open(paths[0]) as tf:
        names = tf.getnames()
        assert names == ['000000000.latent', '000000000.cls', '000000001.latent', '000000001.cls', '000000002.latent', '000000002.cls']
        # train_wds.py:58-65 decode_data: pickle.loads(item['latent']), int(item['cls'].decode('utf-8'))
        assert np.array_equal(pickle.loads(tf.extractfile('000000001.latent').read()), z[1])
        assert int(tf.extractfile('000000001.cls').read().decode('utf-8')) == int(y[1])


def test_read_back_in_order_and_drop_partial(tmp_path):
    paths, z, y = _make(str(tmp_path))
    got = list(D.WdsTarLatents(str(tmp_path), batch=16, shuffle_buf=0))
    assert len(got) == 2  # 40 samples -> 2 full batches, the partial one is dropped (batched(partial=False))
    assert np.array_equal(np.concatenate([g[0] for g in got]), z[:32]) and np.array_equal(np.concatenate([g[1] for g in got]), y[:32])
    assert got[0][0].dtype == np.float32 and got[0][1].dtype == np.int64 and got[0][0].shape == (16, 8, 8, 8)


def test_rank_split_and_shuffle_is_a_permutation(tmp_path):
    paths, z, y = _make(str(tmp_path))
    seen = []
    for rank in range(2):
        ds = D.WdsTarLatents(paths, batch=5, rank=rank, world=2, shuffle_buf=7, seed=3)
        assert ds.paths == paths[rank::2]
        for zz, yy in ds:
            seen.append(zz)
    seen = np.concatenate(seen)
    assert seen.shape[0] == 40
    key = lambda a: sorted(map(float, a.reshape(a.shape[0], -1)[:, 0]))  # noqa: E731
    assert key(seen) == key(z)  # every sample exactly once across the two ranks
    with pytest.raises(ValueError):
        D.WdsTarLatents(paths[:1], batch=4, rank=0, world=2)


def test_prefetcher_preserves_order_and_surfaces_errors(tmp_path):
    paths, z, y = _make(str(tmp_path))
    pf = D.LatentPrefetcher(D.WdsTarLatents(paths, batch=8, shuffle_buf=0), 'cpu', depth=3)
    out = [(m.clone(), l.clone()) for m, l in pf]
    assert len(out) == 5 and all(m.dtype == torch.float32 and l.dtype == torch.int64 for m, l in out)
    assert np.array_equal(torch.cat([m for m, _ in out]).numpy(), z) and np.array_equal(torch.cat([l for _, l in out]).numpy(), y)

    def bad():
        yield z[:4], y[:4]
        raise OSError('shard vanished')

    pf = D.LatentPrefetcher(bad(), 'cpu')
    next(pf)
    with pytest.raises(OSError, match='shard vanished'):
        next(pf)


def test_synthetic_source_statistics():
    it = iter(D.SyntheticMoments(64, 4, 32, 1000, seed=1))
    mom, lab = next(it)
    assert mom.shape == (64, 8, 32, 32) and lab.shape == (64,) and lab.min() >= 0 and lab.max() < 1000
    assert abs(mom[:, :4].std() - 2.745) < 0.05 and (mom[:, 4:] == -10.0).all()


def test_lmdb_reader_says_what_is_missing(tmp_path):
    try:
        import lmdb  # noqa: F401
        pytest.skip('lmdb is installed here')
    except ImportError:
        with pytest.raises(ImportError, match='lmdb'):
            D.LmdbLatents(str(tmp_path), 8, 32)
