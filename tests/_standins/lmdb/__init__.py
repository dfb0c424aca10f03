"""Dict-backed stand-in for the `lmdb` module (not installed in this image, no network): just the calls the reference's
ImageNetLatentDataset (train_utils/datasets.py:261-277) and maskdit_amd.data.LmdbLatents make -- `lmdb.open(path,
readonly=True, lock=False, create=False)`, `env.begin(write=False)`, `txn.get(key_bytes)`, `env.close()`.  A "database"
is a directory holding `data.pkl` = {key bytes: value bytes}; `write_db` creates one.  TEST INFRASTRUCTURE ONLY
(tests/test_data_cpu.py puts this directory on sys.path)."""
import io
import os
import pickle


class Error(Exception):
    pass


class _Txn:
    def __init__(self, table):
        self._t = table

    def get(self, key, default=None):
        return self._t.get(bytes(key), default)


class Environment:
    def __init__(self, path, readonly=True, lock=False, create=False, **kw):
        f = os.path.join(path, 'data.pkl')
        if not os.path.exists(f):
            raise Error(f'{path}: No such file or directory')
        with io.open(f, 'rb') as fh:
            self._t = pickle.load(fh)

    def begin(self, write=False):
        return _Txn(self._t)

    def close(self):
        self._t = None


def open(path, **kw):  # noqa: A001  (the module-level name lmdb exports)
    return Environment(path, **kw)


def write_db(path, table):
    os.makedirs(path, exist_ok=True)
    with io.open(os.path.join(path, 'data.pkl'), 'wb') as fh:
        pickle.dump(dict(table), fh)
