#!/usr/bin/env python3
"""Kernel-by-kernel table of ONE replayed batch-64 training step (bench.py's epoch_bs64), from two rocprofv3 kernel traces of
tools/bench_epoch.py that differ only in the number of replays:

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace -d gpurun_out/ep_a -- python3 tools/bench_epoch.py --replays 100
    rocprofv3 --kernel-trace -d gpurun_out/ep_b -- python3 tools/bench_epoch.py --replays 200
    python3 tools/epoch_trace.py gpurun_out/ep_a gpurun_out/ep_b 100 > profiles/rNN_epoch_bs64_kernel_trace.md

(per kernel: (calls_b - calls_a) / extra replays, (time_b - time_a) / extra replays -- warm-up, capture and set-up cancel)."""
import glob
import os
import sqlite3
import sys


def table(d):
    dbs = glob.glob(os.path.join(d, '**', '*.db'), recursive=True)
    assert dbs, 'no rocprofv3 database under ' + d
    db = sqlite3.connect(dbs[0])
    cur = db.cursor()
    names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table', 'view')")]
    kt = [n for n in names if n == 'kernels'] or [n for n in names if 'kernel_dispatch' in n and 'rocpd' in n] or [n for n in names if 'kernel' in n]
    t = kt[0]
    cols = [r[1] for r in cur.execute('pragma table_info(%s)' % t)]
    name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    out = {}
    for n, c, tot in cur.execute('select %s, count(*), sum(end-start) from %s group by %s' % (name_col, t, name_col)):
        out[n] = (c, tot)
    return out


def main(a, b, extra):
    ta, tb = table(a), table(b)
    rows = []
    for n, (cb, tbt) in tb.items():
        ca, tat = ta.get(n, (0, 0))
        if cb - ca > 0:
            rows.append((n, (cb - ca) / extra, (tbt - tat) / extra / 1e3))
    rows.sort(key=lambda r: -r[2])
    nl, us = sum(r[1] for r in rows), sum(r[2] for r in rows)
    print('launches per step: %.1f   kernel time per step under the profiler: %.1f us\n' % (nl, us))
    print('| kernel | launches / step | us / step | avg us |\n|---|---|---|---|')
    for n, c, u in rows:
        print('| `%s` | %.1f | %.1f | %.1f |' % (n if len(n) < 100 else n[:97] + '...', c, u, u / c))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]))
