#!/usr/bin/env python3
"""Per-step view of a rocprofv3 --kernel-trace CSV of bench.py: which kernels run INSIDE one sampling step, how often, for how long.

  python tools/trace_steps.py <kernel_trace.csv> [--context NAME] [--head N]

A step begins at `k_first_map` (the prologue kernel of SamplingEngine.sample, one launch per call).  Prints, for the last complete
step, launches / total time per kernel name, and the same for everything BEFORE the first step (weight init, packing).  --context NAME:
for the first 12 launches of a kernel whose name contains NAME inside that step, the two kernels before and after (who asked for it)."""
import csv
import sys
from collections import OrderedDict


def main(argv):
    path = argv[0]
    ctx = argv[argv.index('--context') + 1] if '--context' in argv else None
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if r[2].startswith('k_first_map')]
    if len(starts) < 2:
        print('fewer than two steps in the trace'); return 1

    def table(seg, title):
        agg = OrderedDict()
        for s, e, n in seg:
            key = n.split('(')[0][:90]
            a = agg.setdefault(key, [0, 0]); a[0] += 1; a[1] += e - s
        tot = sum(v[1] for v in agg.values())
        span = (seg[-1][1] - seg[0][0]) if seg else 0
        print(f'== {title}: {len(seg)} launches, kernel time {tot / 1e6:.3f} ms, span {span / 1e6:.3f} ms')
        for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
            print(f'  {c:6d} x {t / 1e3 / max(c, 1):9.1f} us = {t / 1e6:8.3f} ms  {k}')

    table(rows[:starts[0]], 'before the first step')
    a, b = starts[-2], starts[-1]
    table(rows[a:b], 'last complete step')
    gaps = sorted(((rows[i + 1][0] - rows[i][1]) for i in range(a, b - 1)), reverse=True)
    idle = sum(g for g in gaps if g > 0)
    print(f'== idle between kernels inside that step: {idle / 1e6:.3f} ms in total; ten largest gaps (us): {[round(g / 1e3, 1) for g in gaps[:10]]}')
    if '--head' in argv:                                   # the first N launches of that step: duration and the idle gap in front of each
        n = int(argv[argv.index('--head') + 1])
        print(f'== first {n} launches of the step (us: duration, gap before)')
        for i in range(a, min(a + n, b)):
            gap = (rows[i][0] - rows[i - 1][1]) / 1e3 if i > a else 0.0
            print(f'  {(rows[i][1] - rows[i][0]) / 1e3:8.1f} {gap:7.1f}  {rows[i][2].split("(")[0][:70]}')
    if ctx:
        hits = [i for i in range(a, b) if ctx in rows[i][2]][:12]
        for i in hits:
            print('  ...', ' | '.join(rows[j][2].split('(')[0][:40] for j in range(max(a, i - 2), min(b, i + 3))))
    return 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
