import sqlite3, sys, glob
for path in glob.glob(sys.argv[1] + '/**/*.db', recursive=True):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
    by = {}
    for k, c, n, v in rows:
        by.setdefault(k, {})[c] = (n, v)
    for k, d in by.items():
        if 'tn8' not in k and 'nt8' not in k: continue
        print(k[:70], {c: f'{v:.3g}' for c, (n, v) in d.items()}, 'launches', next(iter(d.values()))[0])
