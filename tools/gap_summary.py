"""GPU idle time between kernels from a rocprofv3 --kernel-trace database: union of kernel intervals vs wall span.
    python tools/gap_summary.py results.db [skip_first_n_kernels]"""
import sqlite3
import sys


def main(db_path, skip=0):
    db = sqlite3.connect(db_path)
    rows = list(db.execute('select start, end, name from kernels order by start'))[int(skip):]
    span = rows[-1][1] - rows[0][0]
    busy = 0
    cur_s, cur_e = rows[0][0], rows[0][1]
    gaps = []
    for s, e, n in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, n))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f'kernels {len(rows)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms  idle {(span - busy) / 1e6:.2f} ms '
          f'({100.0 * (span - busy) / span:.1f} %)')
    gaps.sort(reverse=True)
    print('largest gaps (us, before kernel):')
    for g, n in gaps[:12]:
        print(f'  {g / 1e3:9.1f}  {n[:70]}')
    small = [g for g, _ in gaps if g < 50e3]
    print(f'gaps < 50 us: {len(small)}, total {sum(small) / 1e6:.2f} ms, mean {sum(small) / max(len(small), 1) / 1e3:.2f} us')


if __name__ == '__main__':
    main(*sys.argv[1:])
