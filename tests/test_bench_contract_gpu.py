"""bench.py's output contract on a real GPU: one JSON line with the driver's keys plus the `roofline` object (and `cpu_baseline`
when not switched off).  Runs the real script as a child process on a small batch so it stays within half a minute."""
import json
import os
import subprocess
import sys

import pytest

from tests import util

pytestmark = pytest.mark.gpu


def _run(*extra):
    cmd = [sys.executable, os.path.join(util.ROOT, 'bench.py'), '--batch', '2', '--steps', '1', '--warmup', '0', *extra]
    out = subprocess.run(cmd, cwd=util.ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, f'expected exactly one JSON line, got {len(lines)}'
    return json.loads(lines[0])


def test_bench_line_has_the_contract_keys():
    j = _run('--no-cpu-baseline')
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert k in j, k
    assert j['metric'].startswith('256x256 images/sec (CFG=1.5) VAR-d16') and j['unit'] == 'images/sec'
    assert j['n_gpus'] == 1 and j['steps'] == 1 and j['warmup'] == 0 and j['higher_is_better'] is True and j['scaling'] == 'weak'
    assert j['vs_baseline'] is None and j['dtype'] == 'f32' and j['data'] == 'synthetic' and 'workload' in j['config']
    assert j['value'] > 0 and abs(j['value'] - 2 / (j['ms_per_step'] * 1e-3)) < 1e-2 * j['value']
    r = j['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'avg_launch_ms'):
        assert k in r, k
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and r['peak'] == 157.3 and 0 < r['frac'] < 1
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    assert 'cpu_baseline' not in j
