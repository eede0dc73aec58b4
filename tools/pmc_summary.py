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


_FILT = next((f for f in ('/opt/rocm/lib/llvm/bin/llvm-cxxfilt', '/usr/bin/c++filt') if os.path.exists(f)), None)
_DEMANGLED = {}


def demangle(name: str) -> str:
    """rocprofv3 prints some symbols mangled (_ZN6vh_f1612k_gn16_apply...): run them through llvm-cxxfilt once each"""
    if not name.startswith('_Z') or _FILT is None:
        return name
    if name not in _DEMANGLED:
        import subprocess
        d = subprocess.run([_FILT, name], capture_output=True, text=True).stdout.strip() or name
        _DEMANGLED[name] = d if not d.startswith('_Z') else _nested_name(name)
    return _DEMANGLED[name]


def _nested_name(name: str) -> str:
    """fallback for symbols the installed c++filt cannot read (_Float16 / __bf16 parameter types, `DF16_` / `DF16b`): the nested name
    `_ZN<len><id><len><id>...[I<integer literal arguments>E]E` alone, which is all this script needs — e.g. _ZN6vh_f168k_attn16ILi4EEEvPKDF16_... ->
    vh_f16::k_attn16<4>()"""
    if not name.startswith('_ZN'):
        return name
    parts, rest = [], name[3:]
    while rest and rest[0].isdigit():
        n = re.match(r'\d+', rest)
        k = int(n.group(0)); parts.append(rest[n.end():n.end() + k]); rest = rest[n.end() + k:]
    if not parts:
        return name
    targs = ''
    if rest.startswith('I'):                                   # template arguments: integer / bool literals only (L<type><value>E)
        vals, rest2 = [], rest[1:]
        while rest2.startswith('L'):
            m = re.match(r'L([a-z])(n?\d+)E', rest2)
            if not m: return name
            v = m.group(2).replace('n', '-')
            vals.append({'0': 'false', '1': 'true'}.get(v, v) if m.group(1) == 'b' else v)
            rest2 = rest2[m.end():]
        if not rest2.startswith('E'): return name
        targs = '<' + ', '.join(vals) + '>'
    return '::'.join(parts) + targs + '()'


def short(name: str) -> str:
    """'void vh_f16::k_gemm16<8, 4, 2, 4>(vh_f16::Params)' -> 'k_gemm16<8,4,2,4>': no return type, no argument list, no namespace (the 16-bit
    kernels live in vh_f16:: / vh_bf16::, one flavour per profiled run; the namespace is kept beside the entry)"""
    name = demangle(name.strip())
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)
    head, lt, rest = name.partition('<')
    head = head.split('::')[-1]
    rest = re.sub(r'\b\w+::', '', rest)
    return (head + lt + rest).replace(', ', ',')


def namespace(name: str) -> str:
    name = re.sub(r'^void ', '', demangle(name.strip()))
    head = name.partition('<')[0].partition('(')[0]
    return '::'.join(head.split('::')[:-1])


def main():
    out, srcs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(list))
    spaces = {}
    for s in srcs:
        files = [s] if os.path.isfile(s) else glob.glob(os.path.join(s, '**', '*counter_collection.csv'), recursive=True)
        for f in files:
            for row in csv.DictReader(open(f)):
                acc[short(row['Kernel_Name'])][row['Counter_Name']].append(float(row['Counter_Value']))
                spaces.setdefault(short(row['Kernel_Name']), namespace(row['Kernel_Name']))
    kernels = {}
    for k, cs in acc.items():
        if not (k.startswith('k_') or k.startswith('void k_')):
            continue
        e = {'launches': max(len(v) for v in cs.values())}
        if spaces.get(k): e['namespace'] = spaces[k]
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
