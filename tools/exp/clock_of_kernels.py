#!/usr/bin/env python3
"""effective clock per kernel symbol from one rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace pass: GRBM_GUI_ACTIVE / 8 / duration (MI355X_MICROARCH.md, DVFS give-back)
   python tools/exp/clock_of_kernels.py DIR"""
import csv, glob, os, sys, re
from collections import defaultdict
d = sys.argv[1]
cc = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
kt = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
dur = {}
for f in kt:
    for r in csv.DictReader(open(f)):
        dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name'])
acc = defaultdict(list)
for f in cc:
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != 'GRBM_GUI_ACTIVE': continue
        did = r['Dispatch_Id']
        if did in dur and dur[did][0] > 0:
            ns, name = dur[did]
            acc[re.sub(r'\(.*$', '', name)[:90]].append((float(r['Counter_Value']) / 8.0 / ns, ns))
for k, v in sorted(acc.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
    if sum(x[1] for x in v) < 2e6: continue
    ghz = sorted(x[0] for x in v)
    print(f'{k:90s} n={len(v):4d} avg {sum(x[1] for x in v)/len(v)/1e3:9.1f} us  clock med {ghz[len(ghz)//2]:.3f} GHz (min {ghz[0]:.3f} max {ghz[-1]:.3f})')
