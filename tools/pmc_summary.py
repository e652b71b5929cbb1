#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files (one row per kernel)."""
import collections
import csv
import sys


def load(path):
    by = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        d = by.setdefault((r['Kernel_Name'], r['Dispatch_Id']), {'dur': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
                                                                 'vgpr': r['VGPR_Count'], 'lds': r['LDS_Block_Size']})
        d[r['Counter_Name']] = float(r['Counter_Value'])
    agg = collections.OrderedDict()
    for (name, _), d in by.items():
        a = agg.setdefault(name, collections.defaultdict(float))
        a['n'] += 1
        for k, v in d.items():
            if k not in ('vgpr', 'lds'):
                a[k] += v
        a['vgpr'], a['lds'] = d['vgpr'], d['lds']
    return agg


def main(paths, filt):
    for path in paths:
        agg = load(path)
        print('##', path)
        for name, a in agg.items():
            if filt and not any(f in name for f in filt):
                continue
            n = a['n']
            keys = [k for k in a if k not in ('n', 'dur', 'vgpr', 'lds')]
            print('%-60s n=%d dur=%.1fus vgpr=%s lds=%s' % (name[:60], n, a['dur'] / n, a['vgpr'], a['lds']))
            print('    ' + '  '.join('%s=%.3g' % (k, a[k] / n) for k in keys))


if __name__ == '__main__':
    paths = [p for p in sys.argv[1:] if p.endswith('.csv')]
    filt = [p for p in sys.argv[1:] if not p.endswith('.csv')]
    main(paths, filt)
