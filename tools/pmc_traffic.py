#!/usr/bin/env python
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM bytes per kernel launch.

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md); both
counters are in KiB-like units of 1024 B... rocprofv3 reports kilobytes.
"""
import csv
import json
import sys
from collections import defaultdict


def load(path, counter):
    acc = defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row.get('Counter_Name') != counter:
            continue
        name = row['Kernel_Name'].split('(')[0]
        name = name.replace('void ', '').split('<')[0]
        acc[name].append(float(row['Counter_Value']))
    return acc


def main():
    f = load(sys.argv[1], 'FETCH_SIZE')
    w = load(sys.argv[2], 'WRITE_SIZE')
    out = {'note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes (python bench.py '
                   '--steps 6 --warmup 2 --no-cpu-baseline --timing-steps 1), cartpole_nomm config; FETCH_SIZE '
                   'doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE uncalibrated',
           'kernels': {}}
    for k in f:
        if not k.startswith('pm_'):
            continue
        fk = sum(f[k]) / len(f[k])
        wk = sum(w[k]) / len(w[k]) if k in w else 0.0
        out['kernels'][k] = {'FETCH_SIZE_KB_mean': fk, 'WRITE_SIZE_KB_mean': wk,
                             'hbm_bytes_per_launch': (2.0 * fk + wk) * 1024.0, 'launches': len(f[k])}
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    for k, v in out['kernels'].items():
        print('%-28s fetch %.1f MB  write %.1f MB' % (k, 2 * v['FETCH_SIZE_KB_mean'] / 1024, v['WRITE_SIZE_KB_mean'] / 1024))


if __name__ == '__main__':
    main()
