"""Empty stand-in: the reference's sample.py imports `lmdb` at module top (sample.py:14)
but the sampler path never touches it.  Used only by tests/golden/make_golden.py."""
