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
    assert 'cpu_baseline' not in j and 'ranks' not in j
    # the 16-bit throughput mode, timed in the same process after the f32 region: its own value, steps, dominant kernel and roofline
    for dt16 in ('f16', 'bf16'):
        m = j['modes'][dt16]
        assert m['dtype'] == dt16 and m['steps'] >= 10 and m['warmup'] == 2 and m['value'] > 0 and m['unit'] == 'images/sec'
        assert abs(m['value'] - 2 / (m['ms_per_step'] * 1e-3)) < 1e-2 * m['value'] and abs(m['speedup_vs_headline'] - m['value'] / j['value']) < 1e-2
        r16 = m['roofline']
        assert r16['peak'] == 2500.0 and r16['family'] in ('gemm16', 'conv16h', 'attn16') and 0 < r16['frac'] < 1 and r16['launches'] > 0
        assert m['whole_path']['peak_tflops'] == 2500.0 and set(m['whole_path']['mfma_time_weighted']['families']) >= {'gemm16_small', 'gemm_small'}
        t2 = m['two_calls_in_flight']                       # informational: the same calls on two alternating streams; `value` stays one call at a time
        assert t2['steps'] == m['steps'] and t2['value'] > 0 and abs(t2['speedup_vs_one_call_at_a_time'] - t2['value'] / m['value']) < 1e-2


def test_bench_line_f16_headline_and_no_modes():
    j = _run('--no-cpu-baseline', '--dtype', 'f16')
    assert j['dtype'] == 'f16' and 'modes' not in j and j['roofline']['peak'] == 2500.0
    j = _run('--no-cpu-baseline', '--no-modes')
    assert j['dtype'] == 'f32' and 'modes' not in j


_RCCL_CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.environ["VAR_AMD_ROOT"])
from var_amd import dist
dist.initialize(backend="nccl")
assert dist.initialized() and dist.get_world_size() == 1 and dist.get_rank() == 0
x = torch.arange(24, dtype=torch.float32, device="cuda").view(2, 3, 4)
y = dist.allgather(x)                                   # all_gather_into_tensor through RCCL
assert y.shape == (2, 3, 4) and torch.equal(x, y)
parts = dist.allgather(x, cat=False)
assert len(parts) == 1 and torch.equal(parts[0], x)
t = torch.ones(5, device="cuda"); dist.allreduce(t); assert float(t.sum()) == 5.0
dist.barrier(); dist.finalize(); print("rccl-ok")
'''


def test_rccl_process_group_on_one_rank():
    """the collectives of the N>1 path (init, barrier, all-gather, all-reduce) through RCCL itself, as far as one GPU allows: a
    1-rank process group (the 2-rank logic runs under gloo in tests/test_host_cpu.py)"""
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(util.free_port()),
               VAR_AMD_ROOT=util.ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, '-c', _RCCL_CHILD], cwd=util.ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'rccl-ok' in out.stdout, out.stderr[-2000:]


def test_bench_under_a_one_rank_torchrun_launch():
    """the driver's N>1 command line with one rank: bench.py joins the RCCL group, times between barriers and still prints one line"""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(util.free_port()), os.path.join(util.ROOT, 'bench.py'), '--gpus', '1', '--batch', '2', '--steps', '1',
           '--warmup', '0', '--no-cpu-baseline']
    out = subprocess.run(cmd, cwd=util.ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1 and json.loads(lines[0])['n_gpus'] == 1


def test_bench_two_ranks_on_one_card_through_the_launcher():
    """bench.py's own world > 1 branches executed as a whole before the driver's 8-GPU run does: `bench.py --gpus 2` outside torchrun starts two ranks
    through var_amd/launch.py; with the rehearsal backend (gloo: RCCL refuses two ranks on one device) both share this card.  Closing barriers,
    rank_stats, the all-gather of the decoded images and the modes.* legs (one stream, no two-stream collectives) all run; one JSON line comes out."""
    cmd = [sys.executable, os.path.join(util.ROOT, 'bench.py'), '--gpus', '2', '--batch', '2', '--steps', '1', '--warmup', '0', '--no-cpu-baseline', '--backend', 'gloo']
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run(cmd, cwd=util.ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, f'expected exactly one JSON line, got {len(lines)}'
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['config']['global_batch'] == 4 and j['scaling'] == 'weak' and j['dtype'] == 'f32'
    assert abs(j['value'] - 4 / (j['ms_per_step'] * 1e-3)) < 1e-2 * j['value']
    r = j['ranks']
    assert 0 < r['rank_ms_min'] <= r['rank_ms_max'] <= j['ms_per_step'] * 1.05
    assert abs(r['allgather_mbytes'] - 4 * 3 * 256 * 256 * 4 / 1e6) < 0.06 and r['allgather_ms'] >= r['allgather_ms_min'] > 0
    assert 'cpu_baseline' not in j
    for dt16 in ('f16', 'bf16'):
        m = j['modes'][dt16]
        assert 'error' not in m and m['value'] > 0 and 'two_calls_in_flight' not in m and m['ranks']['rank_ms_min'] > 0


_SHARD_CHILD = r'''
import contextlib, io, os, sys, torch
import torch.distributed as tdist
sys.path.insert(0, os.environ["VAR_AMD_ROOT"])
from var_amd import dist, detinit
from var_amd.multi import sample_sharded
from models import build_vae_var
dist.initialize(backend="gloo")                          # two ranks on ONE card: RCCL refuses a duplicate GPU, the shards meet on the host
rank, world = dist.get_rank(), dist.get_world_size()
assert world == 2
pns, depth, B = (1, 2, 3, 4, 5, 6, 8, 10, 13, 16), 16, 4
with contextlib.redirect_stdout(io.StringIO()):
    vae, var = build_vae_var(device="cuda", patch_nums=pns, depth=depth, ch=160)
detinit.fill_module_device_(var, depth, 0, "var."); detinit.fill_module_device_(vae, depth, 0, "vae.")
var.eval(); vae.eval(); var.rng = torch.Generator(device="cuda")
labels = torch.tensor([3, 980, 207, 88], device="cuda")
for prec in ("f32", "f16"):
    var.set_hip_precision(prec)
    mine = sample_sharded(var, B, labels, g_seed=7, cfg=1.5, top_k=900, top_p=0.96, rng_mode="exact", gather=False)      # this rank's two images
    parts = [torch.empty(B // 2, 3, 256, 256) for _ in range(world)]
    tdist.all_gather(parts, mine.cpu())
    if rank == 0:
        with torch.inference_mode():
            whole = var.autoregressive_infer_cfg(B, labels, g_seed=7, cfg=1.5, top_k=900, top_p=0.96).cpu()              # the 1-GPU stream
        got = torch.cat(parts, dim=0)
        if prec == "f32": assert torch.equal(got, whole), float((got - whole).abs().max())
        else: assert float((got - whole).abs().max()) <= 5e-3          # (16-bit decoder: GroupNorm partial order follows the batch, tests/test_f16_gpu.py)
dist.barrier(); dist.finalize()
if rank == 0: print("shard-ok")
'''


def test_two_ranks_on_one_card_reproduce_the_one_gpu_stream():
    """BASELINE.json configs[2]'s partition on hardware as far as one card allows: two processes (ranks 0 and 1 of a gloo group) share the GPU,
    each samples its half of the batch with the 'exact' RNG mode (every rank draws the whole Exp(1) fill and keeps its rows), the shards are
    gathered on the host: rank order == batch order, and the images are the single-process call's — bit for bit in f32"""
    port = str(util.free_port())
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK='0', WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=port,
                   VAR_AMD_ROOT=util.ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, '-c', _SHARD_CHILD], cwd=util.ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join(o[1][-1500:] for o in outs)
    assert 'shard-ok' in outs[0][0]
