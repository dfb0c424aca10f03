"""Per-kernel HBM traffic from rocprofv3 --pmc runs (rocpd sqlite): FETCH_SIZE / WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes for wide coalesced reads
(/opt/skills/guides/MI355X_MICROARCH.md, section HBM), so the read side is doubled.
    python tools/pmc_summary.py fetch.db write.db out.json out.txt [model resolution micro_batch "command"]
The optional trailing arguments are recorded in out.json: bench.py only reports `roofline.traffic` when they match the
benchmarked workload."""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute('pragma table_info(counters_collection)')]
    q = ('select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? group by kernel_name'
         if 'kernel_name' in cols else None)
    if q is None:
        raise SystemExit(f'unexpected schema: {cols}')
    return {n: (c, v) for n, c, v in db.execute(q, (counter,))}


def _source_hash():
    from maskdit_amd import _lib
    return _lib.source_hash()


def main(fetch_db, write_db, out_json, out_txt, model=None, resolution=None, micro_batch=None, command=None):
    f = per_kernel(fetch_db, 'FETCH_SIZE')
    w = per_kernel(write_db, 'WRITE_SIZE')
    rows = []
    for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, (0, 0))[1] * f.get(k, (0, 0))[0])):
        fc, fv = f.get(k, (0, 0.0))
        wc, wv = w.get(k, (0, 0.0))
        rows.append((k, fc, fv * 1024 * 2, wv * 1024))
    lines = ['# HBM traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), averaged per kernel name',
             '# read = FETCH_SIZE[KiB] * 1024 * 2 (gfx950 half-count correction for wide coalesced reads); write = WRITE_SIZE[KiB] * 1024 (uncalibrated)',
             f"{'kernel':70s} {'launches':>8s} {'read MB':>10s} {'write MB':>10s}"]
    for k, c, rd, wr in rows:
        lines.append(f'{k[:70]:70s} {c:8d} {rd / 1e6:10.2f} {wr / 1e6:10.2f}')
    open(out_txt, 'w').write('\n'.join(lines) + '\n')
    nt = [(c, rd, wr) for k, c, rd, wr in rows if 'gemm_nt' in k]
    n = sum(c for c, _, _ in nt)
    tot = sum(c * (rd + wr) for c, rd, wr in nt)
    json.dump({'kernel': 'gemm_nt8_kernel + gemm_nt_kernel (all mdt_gemm_nt launches)', 'launches': n,
               'hbm_bytes_per_launch': round(tot / max(n, 1)),
               'model': model, 'resolution': int(resolution) if resolution else None,
               'micro_batch': int(micro_batch) if micro_batch else None, 'command': command, 'source_hash': _source_hash(),
               'note': 'read side = FETCH_SIZE x 2 (gfx950 correction), write side = WRITE_SIZE (uncalibrated)'}, open(out_json, 'w'), indent=1)
    print('\n'.join(lines[:14]))


if __name__ == '__main__':
    main(*sys.argv[1:9])
