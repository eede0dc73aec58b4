#!/usr/bin/env python3
"""Compile one .hip file of var_amd/csrc to gfx950 assembly and print, per kernel, the registers / scratch / occupancy lines and
the instruction mix of every basic block (MFMA, packed VALU, VALU, SALU, LDS, VMEM, scratch).  A development aid: the attention and
GEMM loops are tuned by instruction count, and a `scratch_` access inside a loop is a spill that must not be there.

    python tools/asm_mix.py var_amd/csrc/attn.hip [kernel-name-substring] [--min 20] [--dump LABEL]
"""
import argparse
import collections
import os
import re
import subprocess
import tempfile

ap = argparse.ArgumentParser()
ap.add_argument('src')
ap.add_argument('kernel', nargs='?', default='')
ap.add_argument('--min', type=int, default=20, help='only blocks with at least this many instructions')
ap.add_argument('--dump', default=None, help='print the instructions of this basic block label (e.g. .LBB3_12)')
ap.add_argument('--flags', default='')
args = ap.parse_args()

tmp = tempfile.mkdtemp(prefix='asm_mix_')
src = os.path.abspath(args.src)
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-fPIC', '-std=c++17', '-Wno-unused-result', '-Wno-inline-asm',
       '-save-temps', '-c', src, '-o', os.path.join(tmp, 'x.o')] + args.flags.split()
subprocess.run(cmd, cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
asm = [f for f in os.listdir(tmp) if f.endswith('gfx950.s')][0]
lines = open(os.path.join(tmp, asm)).read().split('\n')


def kind(op):
    if op.startswith('v_mfma') or op.startswith('v_smfma'): return 'mfma'
    if op.startswith('v_pk_'): return 'vpk'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_barrier'): return 'barrier'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('scratch_'): return 'SCRATCH'
    if op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_'): return 'vmem'
    return op


i = 0
while i < len(lines):
    m = re.match(r'^(_Z\w+):', lines[i])
    if not m or args.kernel not in m.group(1):
        i += 1; continue
    name = m.group(1)
    j = i
    while j < len(lines) and 's_endpgm' not in lines[j]: j += 1
    k = j
    info = []
    while k < len(lines) and not lines[k].startswith('_Z'):
        if re.search(r'; (NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|LDSByteSize|NumSgprs)', lines[k]): info.append(lines[k].strip('; ').strip())
        k += 1
    print(f'== {name}\n   ' + ', '.join(info))
    blocks, cur, label = [], [], 'entry'
    for l in lines[i + 1:j + 1]:
        mm = re.match(r'^(\.LBB\d+_\d+):', l)
        if mm:
            blocks.append((label, cur)); label, cur = mm.group(1), []
        else:
            t = l.strip()
            if t and not t.startswith(';') and not t.startswith('.'): cur.append(t)
    blocks.append((label, cur))
    for label, b in blocks:
        if args.dump and label == args.dump:
            print('\n'.join('      ' + x for x in b))
        if len(b) < args.min: continue
        c = collections.Counter(kind(x.split()[0]) for x in b)
        print(f'   {label:10s} {len(b):4d}  ' + '  '.join(f'{k}={v}' for k, v in sorted(c.items())) + f'   | ends: {b[-1]}')
    i = j + 1
