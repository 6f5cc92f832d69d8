"""Summarise rocprofv3 rocpd sqlite outputs (counters_collection / kernels views) per kernel.
usage: python tools/pmc_summary.py <dir-with-*_results.db> [name-filter]"""
import glob
import sqlite3
import sys
from collections import defaultdict

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for db in sorted(glob.glob(root + '/**/*_results.db', recursive=True)):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    name_col = 'kernel_name' if 'kernel_name' in cols else [x for x in cols if 'name' in x and 'kernel' in x][0]
    acc = defaultdict(lambda: defaultdict(float))
    ndisp = defaultdict(set)
    for r in c.execute("select %s, counter_name, value, dispatch_id from counters_collection" % name_col):
        acc[r[0]][r[1]] += r[2]
        ndisp[r[0]].add(r[3])
    for k, d in acc.items():
        if flt in k:
            n = len(ndisp[k])
            print('%s  [%s]  dispatches=%d' % (db.split('/')[-2], k[:70], n))
            for cn, v in sorted(d.items()):
                print('    %-28s %16.0f  (per dispatch %14.1f)' % (cn, v, v / n))
    try:
        for r in c.execute("select name, count(*), avg(end-start), min(end-start) from kernels group by name"):
            if flt in r[0]:
                print('    kernel-trace: %s calls=%d avg=%.1f us min=%.1f us' % (r[0][:60], r[1], r[2] / 1e3, r[3] / 1e3))
    except Exception as e:
        pass
