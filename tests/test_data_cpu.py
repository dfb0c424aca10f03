"""Host logic of the latent input pipeline (maskdit_amd/data.py): the reference's WebDataset shard layout
(train_wds.py:58-97: `<key>.latent` = pickled moments, `<key>.cls` = class index text) written and read back, rank
split, partial-batch drop, shuffle buffer = a permutation, and the prefetcher's ordering / error propagation.  CPU only."""
import os
import pickle
import sys
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


def _lmdb_standin(monkeypatch):
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_standins')
    monkeypatch.syspath_prepend(here)
    sys.modules.pop('lmdb', None)
    import lmdb
    assert hasattr(lmdb, 'write_db'), 'the real lmdb is installed: this test drives the dict-backed stand-in'
    return lmdb


def _write_latent_db(lmdb, root, n, R, seed=3):
    """Records in the reference's layout (train_utils/datasets.py:261-277): `z-{i}` = float32 bytes of the [2 C, R, R]
    moments, `y-{i}` = the class as text, `length` = the record count as text."""
    rs = np.random.RandomState(seed)
    z = rs.randn(n, 8, R, R).astype(np.float32)
    y = rs.randint(0, 1000, size=n)
    tab = {b'length': str(n).encode('utf-8')}
    for i in range(n):
        tab[f'z-{i}'.encode('utf-8')] = z[i].tobytes()
        tab[f'y-{i}'.encode('utf-8')] = str(int(y[i])).encode('utf-8')
    lmdb.write_db(os.path.join(root, 'train'), tab)
    return z, y


def test_lmdb_reader_on_the_reference_record_layout(tmp_path, monkeypatch):
    """maskdit_amd.data.LmdbLatents (the loader the shipped 256^2 config uses: train_utils/datasets.py:240-304 behind
    train.py:166-176) EXECUTED against a dict-backed stand-in `lmdb` (the class had never run a line):
    record decoding ([8, R, R] float32 moments + integer class), rank-strided disjoint shards whose union is one
    permutation of the data set, a different order every epoch, whole batches only, determinism per seed -- and, where
    /root/reference exists (build container), the REFERENCE's own ImageNetLatentDataset reading the same records
    through the same stand-in returns identical (z, y) for every index."""
    lmdb = _lmdb_standin(monkeypatch)
    n, R, B, W = 53, 8, 4, 2
    z, y = _write_latent_db(lmdb, str(tmp_path), n, R)
    shards = [list(D.LmdbLatents(str(tmp_path), B, R, rank=r, world=W, seed=7, epochs=2)) for r in range(W)]
    seen_epochs = [[], []]
    for r in range(W):
        share = len(range(r, n, W))
        assert len(shards[r]) == 2 * (share // B), (r, len(shards[r]))
        for k, (xs, ys) in enumerate(shards[r]):
            assert xs.shape == (B, 8, R, R) and xs.dtype == np.float32 and ys.shape == (B,) and ys.dtype == np.int64
            ep = k // (share // B)
            for x1, y1 in zip(xs, ys):
                hit = np.flatnonzero((z.reshape(n, -1) == x1.reshape(1, -1)).all(1))
                assert len(hit) == 1 and int(y[hit[0]]) == int(y1)  # the record it claims to be, label attached to ITS latent
                seen_epochs[ep].append((r, int(hit[0])))
    for ep in range(2):
        idx = [i for _, i in seen_epochs[ep]]
        assert len(idx) == len(set(idx)), 'a record was served twice within one epoch (ranks must be disjoint)'
    order0 = [i for r, i in seen_epochs[0] if r == 0]
    order1 = [i for r, i in seen_epochs[1] if r == 0]
    assert order0 != order1, 'the epoch reshuffle did not change the order'
    again = list(D.LmdbLatents(str(tmp_path), B, R, rank=0, world=W, seed=7, epochs=2))
    assert all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(again, shards[0]))
    other = list(D.LmdbLatents(str(tmp_path), B, R, rank=0, world=W, seed=8, epochs=1))
    assert not all(np.array_equal(a[1], b[1]) for a, b in zip(other, shards[0]))
    # the prefetcher path train.py uses: device tensors out (CPU degradation), same content
    pf = D.LatentPrefetcher(D.LmdbLatents(str(tmp_path), B, R, seed=7, epochs=1), 'cpu')
    direct = list(D.LmdbLatents(str(tmp_path), B, R, seed=7, epochs=1))
    got = [(m.clone(), l.clone()) for m, l in pf]
    assert len(got) == n // B and all(np.array_equal(g[0].numpy(), d[0]) and np.array_equal(g[1].numpy(), d[1]) for g, d in zip(got, direct))
    # a missing database is the stand-in's (= lmdb's) error, not a silent empty loader
    with pytest.raises(Exception):
        D.LmdbLatents(str(tmp_path / 'nowhere'), B, R)
    ref_root = os.environ.get('MASKDIT_REFERENCE', '/root/reference')
    if os.path.isdir(ref_root):
        import types
        tv = types.ModuleType('torchvision')
        tvd = types.ModuleType('torchvision.datasets')
        tvd.ImageFolder = tvd.VisionDataset = object
        tv.datasets = tvd
        monkeypatch.setitem(sys.modules, 'torchvision', tv)
        monkeypatch.setitem(sys.modules, 'torchvision.datasets', tvd)
        monkeypatch.syspath_prepend(ref_root)
        for k in [k for k in sys.modules if k == 'train_utils' or k.startswith('train_utils.')]:
            monkeypatch.delitem(sys.modules, k)
        from train_utils.datasets import ImageNetLatentDataset  # the reference itself
        ds = ImageNetLatentDataset(str(tmp_path), resolution=R, num_channels=4, split='train')
        assert len(ds) == n
        for i in range(n):
            zr, yr = ds._load_raw_data(i)
            assert np.array_equal(zr, z[i]) and yr == int(y[i])
        ds.env.close()  # (the reference's own close() dereferences feat_env, which only the feature-conditioned path sets)
