#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace results .db into a per-kernel table (markdown) for profiles/."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
    total = sum(r[2] for r in rows)
    lines = ['| kernel | calls | total ms | avg us | min us | max us | % |', '|---|---|---|---|---|---|---|']
    for n, c, tot, avg, mn, mx in rows:
        short = n if len(n) < 110 else n[:107] + '...'
        lines.append('| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.1f |' % (short, c, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3,
                                                                         100.0 * tot / total))
    lines.append('')
    lines.append('total kernel time: %.3f ms over %d dispatches' % (total / 1e6, sum(r[1] for r in rows)))
    text = '\n'.join(lines)
    if out:
        open(out, 'w').write(text + '\n')
    print(text)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
