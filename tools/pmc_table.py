"""Per-kernel table of rocprofv3 --pmc passes (rocpd sqlite): python tools/pmc_table.py dir [dir ...]
Every counter is averaged per launch and kernel name; derived columns where the inputs are present:
  L2 hit rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
  MFMA busy   = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CYCLES-normalised): reported as the ratio
                SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs): matrix-pipe cycles per SIMD-cycle of the
                launch.  GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (a 0.78 ms launch reads 1.17e7 = 8 x 1.47e6
                cycles); SQ_VALU_MFMA_BUSY_CYCLES is the chip total (16 cycles per 16x16x32 bf16 MFMA, checked against
                SQ_INSTS_VALU_MFMA_MOPS_BF16 and the launch's FLOPs)
  LDS busy    = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE / 8 x 256 CUs): LDS-array cycles per CU-cycle (bank-conflict cycles included)
  wave states = SQ_WAIT_ANY | SQ_WAIT_INST_ANY | SQ_ACTIVE_INST_ANY as fractions of SQ_WAVE_CYCLES (MI355X_MICROARCH.md:
                parked on s_waitcnt / barrier | issue stalls | issuing)."""
import glob
import sqlite3
import sys


def load(d):
    out = {}
    for path in glob.glob(d + '/**/*.db', recursive=True):
        db = sqlite3.connect(path)
        for k, c, n, v in db.execute('select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name'):
            out.setdefault(k, {})[c] = (n, v)
    return out


def main(dirs):
    allk = {}
    for d in dirs:
        for k, cs in load(d).items():
            allk.setdefault(k, {}).update(cs)
    rows = []
    for k, cs in allk.items():
        g = lambda n: cs.get(n, (0, None))[1]  # noqa: E731
        n = max(c[0] for c in cs.values())
        hit, miss = g('TCC_HIT_sum'), g('TCC_MISS_sum')
        wc = g('SQ_WAVE_CYCLES')
        gui = g('GRBM_GUI_ACTIVE')
        rows.append((gui or 0, k, n, {
            'L2 hit': None if hit is None or miss is None or hit + miss == 0 else hit / (hit + miss),
            'MFMA busy': None if not gui or g('SQ_VALU_MFMA_BUSY_CYCLES') is None else g('SQ_VALU_MFMA_BUSY_CYCLES') / (gui / 8 * 256 * 4),
            'wait': None if not wc or g('SQ_WAIT_ANY') is None else g('SQ_WAIT_ANY') / wc,
            'stall': None if not wc or g('SQ_WAIT_INST_ANY') is None else g('SQ_WAIT_INST_ANY') / wc,
            'issue': None if not wc or g('SQ_ACTIVE_INST_ANY') is None else g('SQ_ACTIVE_INST_ANY') / wc,
            'lds stall': None if not wc or g('SQ_WAIT_INST_LDS') is None else g('SQ_WAIT_INST_LDS') / wc,
            'LDS busy': None if not gui or g('SQ_LDS_IDX_ACTIVE') is None else g('SQ_LDS_IDX_ACTIVE') / (gui / 8 * 256),
            'bank conflict': None if not g('SQ_LDS_IDX_ACTIVE') or g('SQ_LDS_BANK_CONFLICT') is None else g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE'),
            'GUI cycles': gui, 'MFMA cycles': g('SQ_VALU_MFMA_BUSY_CYCLES'), 'mfma bf16 mops': g('SQ_INSTS_VALU_MFMA_MOPS_BF16'),
        }))
    rows.sort(key=lambda r: -r[0] * r[2])
    cols = ['L2 hit', 'MFMA busy', 'wait', 'stall', 'issue', 'lds stall', 'LDS busy', 'bank conflict', 'GUI cycles', 'MFMA cycles', 'mfma bf16 mops']
    print(f"{'kernel':62s} {'n':>5s} " + ' '.join(f'{c:>13s}' for c in cols))
    for _, k, n, d in rows:
        cells = []
        for c in cols:
            v = d[c]
            cells.append(f'{"-":>13s}' if v is None else (f'{v:13.3f}' if v < 100 else f'{v:13.4g}'))
        print(f'{k[:62]:62s} {n:5d} ' + ' '.join(cells))


if __name__ == '__main__':
    main(sys.argv[1:])
