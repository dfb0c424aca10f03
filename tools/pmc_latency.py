"""Mean L2-miss latency per kernel from a rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum
run (rocpd sqlite):   python tools/pmc_latency.py <dir> [name filter]
latency = TCC_EA0_RDREQ_LEVEL_sum / TCC_EA0_RDREQ_sum  (L2 clock cycles a read request to the fabric stays outstanding);
see tools/mall_probe.py for the HBM / Infinity-Cache calibration points."""
import glob
import sqlite3
import sys


def main(d, flt=''):
    rows = {}
    for path in glob.glob(d + '/**/*.db', recursive=True):
        db = sqlite3.connect(path)
        q = 'select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name'
        for k, c, n, v in db.execute(q):
            rows.setdefault(k, {})[c] = (n, v)
    print(f"{'kernel':72s} {'launches':>8s} {'RDREQ / launch':>15s} {'to DRAM':>8s} {'miss latency (cycles)':>22s}")
    out = []
    for k, cs in rows.items():
        if flt and flt not in k:
            continue
        rd, lv, dr = cs.get('TCC_EA0_RDREQ_sum'), cs.get('TCC_EA0_RDREQ_LEVEL_sum'), cs.get('TCC_EA0_RDREQ_DRAM_sum')
        if not rd or not lv or rd[1] < 1e4:
            continue
        out.append((rd[1] * rd[0], k, rd[0], rd[1], (dr[1] / rd[1]) if dr else float('nan'), lv[1] / rd[1]))
    for _, k, n, r, fr, lat in sorted(out, reverse=True):
        name = k[:72] if 'reduce_kernel' not in k else '...' + k[k.find('ReduceOp'):][:69]
        print(f'{name:72s} {n:8d} {r:15.4g} {fr:8.3f} {lat:22.1f}')


if __name__ == '__main__':
    main(*sys.argv[1:3])
