#!/usr/bin/env python3
"""Build-time checks on the gfx950 assembly hipcc emits for the hand-scheduled kernels (run by var_amd/csrc/Makefile, target `asmcheck`,
on `hipcc -S --cuda-device-only` output; tests/test_build_cpu.py runs it too).

  python tools/check_kernel_asm.py [--no-scratch] [--mfma-hazard] file.s ...

--no-scratch   every kernel of the file must have .private_segment_fixed_size 0 and .vgpr_spill_count 0.
               Why: k_gemm16 / k_conv16 / k_conv16h / k_attn* keep LDS-DMA tiles in flight behind hand-counted `s_waitcnt vmcnt(N)`.
               vmcnt counts every vector-memory operation of the wave in issue order, so a compiler-inserted scratch access can only make
               such a wait cover MORE (never fewer) of the older requests — results stay right — but every scratch reload also brings the
               compiler's own `s_waitcnt vmcnt(0)`, which drains the request pipeline the kernel was built around: a silent slowdown.
               Several instantiations sit at their register cap, so a compiler update could start spilling without any source change.
  --scratch-outside-loops-ok SUBSTR   kernels whose mangled name contains SUBSTR may spill OUTSIDE loops (prologue / epilogue code that runs
               once per tile); scratch instructions inside a loop still fail.
  --scratch-ok SUBSTR                 kernels that are known to spill inside their loop (listed with the reason in the Makefile).

--mfma-hazard  no instruction of an inline-asm block may touch the destination registers of an MFMA whose result has not been
               "settled".  hipcc pads the MFMA -> VALU read/write hazard (up to 18 wait states behind a 16-pass v_mfma) for ITS OWN
               instructions only; an `asm("v_max3_f32 ...")` placed first behind the score MFMAs of the attention kernels read the
               accumulators before the last MFMA had written them (round 2: run-to-run differences of the tile maximum that no
               tolerance test caught).  An MFMA is settled once a compiler-generated (non-asm, non-MFMA) instruction has read or written
               one of its destination registers (the compiler put the wait states in front of that instruction, and instructions issue
               in order), or once 20 wait states' worth of instructions have issued behind it (`s_nop N` counts N + 1, anything else 1).
               Control flow: the listing is walked linearly, and at every branch taken with MFMAs still unsettled the branch target is
               walked as well with those MFMAs pending (loop back-edges included), one level deep.
"""
import re
import sys

SETTLE_STATES = 20          # >= the largest MFMA -> VALU requirement on gfx950 (16-pass XDL op: 18) with margin

_REG1 = re.compile(r'\b([va])(\d+)\b')
_REGR = re.compile(r'\b([va])\[(\d+):(\d+)\]')
_FUNC = re.compile(r'^([A-Za-z_][\w$]*):')          # a function label at column 0 (block labels start with .L)


def regs_of(text):
    out = set()
    for m in _REGR.finditer(text):
        out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
    for m in _REG1.finditer(text):
        out.add((m.group(1), int(m.group(2))))
    return out


def _scan(lines, start, pending, fname, bad, labels, follow):
    """walk the listing from line index `start`; pending = unsettled MFMAs [dest regs, wait states since, line, text], oldest first.
    follow=True (the linear pass over the file): at every branch with MFMAs still pending the branch TARGET is scanned too (follow=False,
    until they are settled), so an MFMA at the bottom of a loop body followed by an inline-asm reader at the top of the next iteration — the
    shape of the round-2 attention bug, across the back-edge — is seen.  A target scan ends when nothing is pending or the kernel ends."""
    in_asm = False
    seen = set()
    for i in range(start, len(lines)):
        ln, raw = i + 1, lines[i]
        s = raw.strip()
        if s.startswith(';;#ASMSTART'):
            in_asm = True; continue
        if s.startswith(';;#ASMEND'):
            in_asm = False; continue
        if not s or s.startswith((';', '.', '//')) or s.endswith(':'):
            if s.startswith('.amdhsa_kernel') or s.startswith('.end_amdhsa_kernel') or _FUNC.match(raw):
                if not follow: return
                pending = []
            continue
        code = s.split(';')[0].strip()
        if not code:
            continue
        mnem = code.split()[0]
        if mnem == 's_endpgm':
            if not follow: return
            pending = []; continue
        states = 1
        if mnem == 's_nop':
            try: states = int(code.split()[1], 0) + 1
            except (IndexError, ValueError): states = 1
        is_mfma = mnem.startswith(('v_mfma', 'v_smfmac'))
        touched = regs_of(code[len(mnem):]) if mnem[0] in 'vdgbfs' else set()
        if pending and touched:
            hit = [i2 for i2, p in enumerate(pending) if p[0] & touched]
            if hit and in_asm and is_mfma:
                # an inline-asm MFMA that accumulates in place (`v_mfma d, a, b, d`, VH16_MFMA_16x16x32_INPLACE): its read of an older MFMA's result is
                # the src C of the same register tuple — the accumulate chain, which the hardware interlocks (no software wait states for dst == src C
                # of an identical tuple).  Only that exact overlap is exempt: an A / B operand or a partial overlap is still a violation.
                ops_ = [o.strip() for o in code[len(mnem):].split(',')]
                if len(ops_) >= 4 and ops_[0] == ops_[3]:
                    own = regs_of(ops_[0]); ab = regs_of(ops_[1]) | regs_of(ops_[2])
                    hit = [i2 for i2 in hit if not (pending[i2][0] == own and not (pending[i2][0] & ab))]
            if hit:
                if in_asm:
                    p = pending[hit[-1]]
                    msg = (f'{fname}:{ln}: inline-asm `{code}` touches the result of the MFMA at line {p[2]} (`{p[3]}`) only {p[1]} wait '
                           f'state(s) behind it and before any compiler-generated reader' + ('' if follow else ' (reached through a branch)'))
                    if msg not in bad: bad.append(msg)
                elif not is_mfma:
                    pending = pending[hit[-1] + 1:]          # the compiler padded this read: that MFMA and every older one have completed
        for p in pending:
            p[1] += states
        pending = [p for p in pending if p[1] < SETTLE_STATES]
        if is_mfma:
            ops = code[len(mnem):].split(',')
            dest = regs_of(ops[0]) if ops else set()
            if dest:
                pending.append([dest, 0, ln, code])
        if not follow and not pending:
            return
        if mnem.startswith(('s_cbranch', 's_branch')) and pending:
            tgt = code.split()[-1]
            if follow and tgt in labels and (tgt, tuple(p[2] for p in pending)) not in seen:
                seen.add((tgt, tuple(p[2] for p in pending)))
                _scan(lines, labels[tgt], [[set(p[0]), p[1], p[2], p[3]] for p in pending], fname, bad, labels, False)
            if not follow and mnem == 's_branch':
                return                                       # (an unconditional jump inside a target scan: one level of following only)


def check_hazards(lines, fname='<asm>'):
    """-> list of violation strings"""
    bad = []
    labels = {m.group(1): i + 1 for i, raw in enumerate(lines) for m in [re.match(r'^(\.L[\w$]+):', raw.strip())] if m}
    _scan(lines, 0, [], fname, bad, labels, True)
    return bad


def check_scratch(lines, fname='<asm>', outside_ok=(), ok=()):
    """kernels with scratch / spills: refused, unless allowed outside loops (then every scratch instruction must sit in a block that is
    not part of a loop) or allowed altogether"""
    bad, name = [], '?'
    spilling = {}
    for raw in lines:
        s = raw.strip()
        if s.startswith('.name:'):
            name = s.split(':', 1)[1].strip()
        # (.sgpr_spill_count alone is harmless here: with .private_segment_fixed_size 0 the SGPRs went to lanes of a VGPR —
        # v_writelane / v_readlane, no memory operation, nothing vmcnt sees)
        for key in ('.private_segment_fixed_size:', '.vgpr_spill_count:'):
            if s.startswith(key) and int(s.split(':', 1)[1]) != 0:
                spilling.setdefault(name, []).append(s)
    in_loop_scratch = {}
    cur, in_loop = None, False
    for ln, raw in enumerate(lines, 1):
        s = raw.strip()
        m = _FUNC.match(raw)
        if m:
            cur, in_loop = m.group(1), False
        elif s.startswith('.LBB'):
            in_loop = ('in Loop:' in s) or ('Loop Header' in s)
        elif s.startswith('scratch_') and in_loop and cur is not None:
            in_loop_scratch.setdefault(cur, []).append(ln)
    for name, what in spilling.items():
        if any(t in name for t in ok):
            continue
        if any(t in name for t in outside_ok):
            if name in in_loop_scratch:
                bad.append(f'{fname}: kernel {name}: scratch access inside a loop at line(s) {in_loop_scratch[name][:6]} (spills are tolerated outside loops only)')
            continue
        bad.append(f'{fname}: kernel {name}: ' + ', '.join(what) + '  (scratch reloads drain the LDS-DMA pipeline: vmcnt(0))')
    return bad


def main(argv):
    want_scratch = '--no-scratch' in argv
    want_hazard = '--mfma-hazard' in argv
    outside_ok, ok, files, it = [], [], [], iter(argv)
    for a in it:
        if a == '--scratch-outside-loops-ok': outside_ok.append(next(it))
        elif a == '--scratch-ok': ok.append(next(it))
        elif not a.startswith('--'): files.append(a)
    if not files or not (want_scratch or want_hazard):
        print(__doc__); return 2
    bad = []
    for f in files:
        lines = open(f).read().splitlines()
        if want_scratch: bad += check_scratch(lines, f, outside_ok, ok)
        if want_hazard: bad += check_hazards(lines, f)
    for b in bad:
        print('[asmcheck] ' + b, file=sys.stderr)
    if not bad:
        print(f'[asmcheck] ok: {len(files)} file(s)' + (' no scratch/spills' if want_scratch else '') + (' no asm reader of an unsettled MFMA result' if want_hazard else ''))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
