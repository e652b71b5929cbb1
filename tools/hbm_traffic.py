#!/usr/bin/env python3
"""HBM bytes per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; MI355X_MICROARCH.md HBM section):
   python tools/hbm_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out prefix> [title]
Counters are KiB; FETCH_SIZE is doubled (gfx950 reports half of wide coalesced reads).  Writes <prefix>.md/.json."""
import collections
import csv
import json
import sys


def avg(path, counter):
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            by[r['Kernel_Name']].append(float(r['Counter_Value']))
    return {k: (sum(v) / len(v), len(v)) for k, v in by.items()}


def main():
    fpath, wpath, prefix = sys.argv[1:4]
    title = sys.argv[4] if len(sys.argv) > 4 else ''
    f, w = avg(fpath, 'FETCH_SIZE'), avg(wpath, 'WRITE_SIZE')
    out = {'workload': title,
           'method': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; counters are KiB; '
                     'FETCH_SIZE doubled (gfx950); per-launch averages', 'kernels': {}}
    lines = ['| kernel | launches | FETCH_SIZE KiB | x2 corrected MB | WRITE_SIZE KiB | MB | HBM MB / launch |',
             '|---|---|---|---|---|---|---|']
    for k in sorted(f, key=lambda k: -(2 * f[k][0] + w.get(k, (0, 0))[0]) * f[k][1]):
        if 'gml_k_' not in k:
            continue
        rd, wr = 2 * f[k][0] * 1024, w.get(k, (0, 0))[0] * 1024
        if rd + wr < 1e6:
            continue
        short = k.split('(')[0].replace('void ', '')
        out['kernels'][short] = {'fetch_bytes_corrected': rd, 'write_bytes': wr, 'hbm_bytes_per_launch': rd + wr,
                                 'launches': f[k][1]}
        lines.append('| `%s` | %d | %.0f | %.1f | %.0f | %.1f | %.1f |' % (short, f[k][1], f[k][0], rd / 1e6,
                                                                        w.get(k, (0, 0))[0], wr / 1e6, (rd + wr) / 1e6))
    json.dump(out, open(prefix + '.json', 'w'), indent=1)
    open(prefix + '.md', 'w').write('# HBM traffic per launch (PMC)\n\n%s\n\n%s\n\n%s\n' % (title, out['method'], '\n'.join(lines)))
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
