#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter_collection CSVs into profiles/<round>_pmc_traffic.json.

  python tools/pmc_summary.py OUT.json DIR_OR_CSV [DIR_OR_CSV ...]
Each argument is one rocprofv3 pass (a directory is searched for *counter_collection.csv).  Per kernel name the script averages
every counter over the kernel's dispatches and derives the HBM-side traffic per launch the way MI355X_MICROARCH.md prescribes
for gfx950: FETCH_SIZE (KB) reports half the bytes of wide coalesced reads, so traffic = (2*FETCH_SIZE + WRITE_SIZE) KB."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)
    return name.replace(', ', ',')


def main():
    out, srcs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(list))
    for s in srcs:
        files = [s] if os.path.isfile(s) else glob.glob(os.path.join(s, '**', '*counter_collection.csv'), recursive=True)
        for f in files:
            for row in csv.DictReader(open(f)):
                acc[short(row['Kernel_Name'])][row['Counter_Name']].append(float(row['Counter_Value']))
    kernels = {}
    for k, cs in acc.items():
        if not (k.startswith('k_') or k.startswith('void k_')):
            continue
        e = {'launches': max(len(v) for v in cs.values())}
        for c, v in cs.items():
            e[c + '_avg'] = round(sum(v) / len(v), 3)
        if 'FETCH_SIZE' in cs and 'WRITE_SIZE' in cs:
            e['traffic_bytes_per_launch'] = round((2.0 * e['FETCH_SIZE_avg'] + e['WRITE_SIZE_avg']) * 1024.0, 1)
        kernels[k] = e
    doc = {'source': 'rocprofv3 --pmc <counter> (one counter group per pass) over `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline`, MI355X',
           'note': 'FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE on gfx950 reports half the bytes of wide coalesced reads '
                   '(MI355X_MICROARCH.md, HBM section): traffic = (2*FETCH_SIZE + WRITE_SIZE) KB; infinity-cache hits are counted in FETCH_SIZE',
           'kernels': dict(sorted(kernels.items(), key=lambda kv: -kv[1].get('traffic_bytes_per_launch', 0) * kv[1]['launches']))}
    json.dump(doc, open(out, 'w'), indent=1)
    for k, e in doc['kernels'].items():
        print(k, e)


if __name__ == '__main__':
    main()
