"""Summarise a rocprofv3 --kernel-trace result database (rocpd sqlite) into a per-kernel table.
    python tools/rocprof_summary.py gpurun_out/prof2/r2_results.db profiles/r1_xxx.txt "header text" """
import sqlite3
import sys


def main(db_path, out_path, header):
    db = sqlite3.connect(db_path)
    rows = list(db.execute('select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) '
                           'from kernels group by name order by 3 desc'))
    tot = sum(r[2] for r in rows)
    out = [f'# {header}', f'# total kernel time {tot / 1e6:.1f} ms', '',
           f"{'kernel':72s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}"]
    for n, c, t, a, mn, mx in rows:
        out.append(f'{n[:72]:72s} {c:7d} {t / 1e6:10.2f} {a / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * t / tot:6.2f}')
    open(out_path, 'w').write('\n'.join(out) + '\n')
    print('\n'.join(out[:26]))


if __name__ == '__main__':
    main(*sys.argv[1:4])
