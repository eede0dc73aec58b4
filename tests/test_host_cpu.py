"""CPU-side checks: the C ABI surface, the drop-in `models` boundary, host-side tables, and the multi-rank sharding logic
(gloo, world_size 2).  No GPU compute happens here; the reference (if present in this container) is only used to pin the
boundary (state-dict layout, teacher-forced forward) — these tests skip that part where /root/reference is absent."""
import contextlib
import ctypes
import io
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import util

ROOT = util.ROOT
REF = '/root/reference'


def _quiet_build(**kw):
    from models import build_vae_var
    with contextlib.redirect_stdout(io.StringIO()):
        return build_vae_var(device='cpu', **kw)


# ---- C ABI ----------------------------------------------------------------------------------------------------------------
def _header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'var_hip.h')).read()
    return sorted(set(re.findall(r'\b(varhip_\w+)\s*\(', txt)))


def test_hip_library_exports_every_declared_symbol():
    so_path = os.path.join(ROOT, 'var_amd', 'libvar_hip.so')
    if not os.path.exists(so_path):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'var_amd', 'csrc'), '-j8'], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    so = ctypes.CDLL(so_path)
    syms = _header_symbols()
    assert len(syms) >= 24
    missing = [s for s in syms if not hasattr(so, s)]
    assert not missing, f'declared in include/var_hip.h but not exported: {missing}'
    from var_amd import abi, hip
    assert set('varhip_' + k for k in abi.SIGNATURES) <= set(syms)
    assert 'gfx950' in hip.lib().version()


def test_oracle_exports_a_twin_for_every_compute_entry_point():
    util.ensure_oracle_built()
    so = ctypes.CDLL(os.path.join(ROOT, 'oracle', 'libvar_oracle.so'))
    from var_amd import abi
    for name in abi.SIGNATURES:
        assert hasattr(so, 'varref_' + name), name


def test_product_never_imports_the_oracle():
    """the oracle is test infrastructure: no file of the product tree may mention it as an import"""
    bad = []
    for base in ('var_amd', 'models'):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith('.py'):
                    src = open(os.path.join(dp, fn)).read()
                    if re.search(r'^\s*(from|import)\s+oracle\b', src, re.M) or 'libvar_oracle' in src:
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_sampling_path_fails_loudly_without_a_gpu():
    vae, var = _quiet_build(depth=2, ch=32, patch_nums=(1, 2, 3))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        var.autoregressive_infer_cfg(2, torch.tensor([1, 2]), g_seed=0)
    from models.helpers import sample_with_top_k_top_p_
    with pytest.raises(NotImplementedError):
        sample_with_top_k_top_p_(torch.randn(1, 2, 4096), top_k=10, top_p=0.5)
    from var_amd import hip
    with pytest.raises(hip.VarHipError):                      # engine refuses CPU parameters before any launch
        var.engine().refresh()


# ---- boundary -------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('kw', [dict(depth=2, ch=32, patch_nums=(1, 2, 3)), dict(depth=3, ch=32, patch_nums=(1, 2, 4), shared_aln=True, attn_l2_norm=False)])
def test_state_dict_layout_matches_the_documented_one(kw):
    from var_amd import shapes
    vae, var = _quiet_build(**kw)
    want = shapes.var_shapes(kw['depth'], kw['patch_nums'], shared_aln=kw.get('shared_aln', False), attn_l2_norm=kw.get('attn_l2_norm', True))
    got = {k: tuple(v.shape) for k, v in var.state_dict().items()}
    assert list(got.items()) == list(want.items())
    assert {k: tuple(v.shape) for k, v in vae.state_dict().items()} == dict(shapes.vae_shapes(ch=kw['ch'], patch_nums=kw['patch_nums']))
    assert isinstance(var.vae_proxy, tuple) and isinstance(var.vae_quant_proxy, tuple) and not any(k.startswith('vae') for k in got)
    for attr in ('patch_nums', 'cond_drop_rate', 'shared_ada_lin', 'lvl_1L', 'uniform_prob', 'num_stages_minus_1', 'rng', 'prog_si', 'begin_ends'):
        assert hasattr(var, attr), attr
    var.blocks[0].attn.kv_caching(True); var.blocks[0].attn.kv_caching(False)
    from models.var import AdaLNSelfAttn, gumbel_softmax_with_rng, sample_with_top_k_top_p_   # noqa: F401  (names the notebooks import)
    import dist
    assert dist.get_device() is not None and dist.initialized() is False


def _import_reference():
    """the reference's `models` package under a private name (it must not shadow ours)"""
    if not os.path.isdir(REF):
        pytest.skip('reference not present on this machine')
    import importlib
    import typing
    torch.Optional = typing.Optional
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == 'models' or k.startswith('models.') or k == 'dist'}
    sys.path.insert(0, REF)
    try:
        ref_models = importlib.import_module('models')
        ref_mods = {k: v for k, v in sys.modules.items() if k == 'models' or k.startswith('models.') or k == 'dist'}
    finally:
        sys.path.remove(REF)
        for k in list(sys.modules):
            if k == 'models' or k.startswith('models.') or k == 'dist':
                del sys.modules[k]
        sys.modules.update(saved)
    return ref_models, ref_mods


def test_against_reference_statedict_and_teacher_forced_forward():
    ref_models, _ = _import_reference()
    from var_amd.detinit import fill_module_
    kw = dict(depth=2, ch=32, patch_nums=(1, 2, 3, 4))
    with contextlib.redirect_stdout(io.StringIO()):
        rvae, rvar = ref_models.build_vae_var(device='cpu', **kw)
    vae, var = _quiet_build(**kw)
    fill_module_(rvar, 2, 0, 'var.'); fill_module_(rvae, 2, 0, 'vae.')
    var.load_state_dict(rvar.state_dict(), strict=True)           # their checkpoint into our module
    vae.load_state_dict(rvae.state_dict(), strict=True)
    rvar.load_state_dict(var.state_dict(), strict=True)           # and back
    rvar.eval(); var.eval(); rvae.eval(); vae.eval()
    rvar.cond_drop_rate = var.cond_drop_rate = 0.0
    g = torch.Generator().manual_seed(3)
    img = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    with torch.no_grad():
        ridx, idx = rvae.img_to_idxBl(img), vae.img_to_idxBl(img)                       # encode side (PyTorch, API kept)
        for a, b in zip(ridx, idx):
            assert torch.equal(a, b)
        x_in = vae.quantize.idxBl_to_var_input(idx)
        assert torch.allclose(x_in, rvae.quantize.idxBl_to_var_input(ridx), atol=1e-6)
        lab = torch.tensor([3, 999])
        assert torch.allclose(var(lab, x_in), rvar(lab, x_in), atol=2e-5, rtol=1e-5)   # teacher-forced logits
        f = torch.randn(2, 32, 4, 4, generator=g)
        assert torch.allclose(vae.fhat_to_img(f), rvae.fhat_to_img(f), atol=1e-5)       # CPU tensors take the PyTorch modules
        ms = [vae.quantize.embedding(i).transpose(1, 2).reshape(2, 32, pn, pn) for i, pn in zip(idx, kw['patch_nums'])]
        assert torch.allclose(vae.quantize.embed_to_fhat(ms, last_one=True), rvae.quantize.embed_to_fhat(ms, last_one=True), atol=1e-6)


def test_encode_side_against_golden_fixture(golden_dir):
    """f_to_idxBl_or_fhat / idxBl_to_var_input (reference quant.py:135-184) vs tests/golden/nearest_code.npz"""
    import json
    z = np.load(f'{golden_dir}/nearest_code.npz')
    meta = json.loads(str(z['meta']))
    from var_amd import shapes
    from var_amd.detinit import make_state_dict
    vae, _ = _quiet_build(depth=meta['depth'], ch=meta['ch'], patch_nums=tuple(meta['patch_nums']))
    sd = make_state_dict(shapes.vae_shapes(ch=meta['ch'], patch_nums=tuple(meta['patch_nums'])), depth=meta['depth'], seed=0, prefix='vae.')
    vae.load_state_dict({**{k: torch.from_numpy(v) for k, v in sd.items()}, 'quantize.ema_vocab_hit_SV': vae.quantize.ema_vocab_hit_SV}, strict=True)
    f = torch.from_numpy(z['f'])
    with torch.no_grad():
        idx = vae.quantize.f_to_idxBl_or_fhat(f, to_fhat=False)
        fh = vae.quantize.f_to_idxBl_or_fhat(f, to_fhat=True)
        for si in range(len(idx)):
            assert np.array_equal(idx[si].numpy().astype(np.int32), z[f'idx_s{si}']), si
            assert np.allclose(fh[si].numpy(), z[f'f_hat_s{si}'], atol=1e-5)
        assert np.allclose(vae.quantize.idxBl_to_var_input(idx).numpy(), z['var_input'], atol=1e-5)
    # the oracle's nearest-code twin on the first scale's query (area-pooled to 1x1 == spatial mean)
    util.ensure_oracle_built()
    from oracle.var_oracle import lib, _p
    zq = np.ascontiguousarray(z['f'].mean(axis=(2, 3)), dtype=np.float32)
    out = np.zeros(zq.shape[0], np.int64)
    assert lib()['nearest_code_f32'](_p(zq), _p(np.ascontiguousarray(sd['quantize.embedding.weight'])), _p(out), zq.shape[0], 4096, 32) == 0
    assert np.array_equal(out.astype(np.int32), z['idx_s0'][:, 0])


# ---- host tables ------------------------------------------------------------------------------------------------------------
def test_bicubic_taps_and_phi_schedule():
    from var_amd.engine import bicubic_taps, phi_index
    util.ensure_oracle_built()
    from oracle import var_oracle
    for pn in (1, 2, 3, 4, 5, 6, 8, 10, 13):
        ti, tw = bicubic_taps(pn, 16)
        oi, ow = var_oracle.bicubic_taps(pn, 16)
        assert np.array_equal(ti, oi) and np.array_equal(tw, ow)
        h = torch.randn(1, 1, pn, pn, dtype=torch.float64)
        M = np.zeros((16, pn))
        for o in range(16):
            for k in range(4): M[o, ti[o, k]] += tw[o, k]
        up = torch.nn.functional.interpolate(h, size=(16, 16), mode='bicubic').numpy()[0, 0]
        assert np.abs(M @ h.numpy()[0, 0] @ M.T - up).max() < 2e-6
    assert [phi_index(si, 10, 4) for si in range(10)] == [0, 0, 1, 1, 1, 2, 2, 3, 3, 3]
    assert [phi_index(si, 3, 4) for si in range(3)] == [0, 2, 3]
    assert [var_oracle.phi_index(si, 10, 4) for si in range(10)] == [0, 0, 1, 1, 1, 2, 2, 3, 3, 3]


def test_detinit_is_deterministic_and_covers_every_parameter():
    from var_amd import shapes
    from var_amd.detinit import hash_uniform, make_state_dict
    a, b = hash_uniform('x', 1000, 0), hash_uniform('x', 1000, 0)
    assert np.array_equal(a, b) and not np.array_equal(a, hash_uniform('y', 1000, 0)) and not np.array_equal(a, hash_uniform('x', 1000, 1))
    assert a.min() >= -1 and a.max() < 1
    sd = make_state_dict(shapes.var_shapes(2, (1, 2, 3), shared_aln=True), depth=2)
    assert all(np.isfinite(v).all() for v in sd.values())
    assert abs(float(hash_uniform('blocks.0.attn.proj.weight', 4096, 0)[:5].sum()) - float(hash_uniform('blocks.0.attn.proj.weight', 5, 0).sum())) == 0


# ---- multi-rank sharding (gloo, 2 processes) ------------------------------------------------------------------------------------
_WORKER = r'''
import os, sys, torch
sys.path.insert(0, os.environ['VAR_ROOT'])
import torch.distributed as tdist
from var_amd import dist, multi
tdist.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']))
dist._state.update(rank=tdist.get_rank(), world=tdist.get_world_size(), init=True, device='cpu')
class FakeVar:                      # just what sample_sharded touches
    V = 64
    lvl_1L = torch.zeros(1)
    rng = torch.Generator()
    patch_nums = (1, 2, 3)
def fake_sample(B_local, labels, noise_fn):      # "image" = a function of the label and of the rows of noise the rank was handed
    acc = labels.float().view(B_local, 1, 1, 1).expand(B_local, 3, 2, 2).clone()
    for si, pn in enumerate(FakeVar.patch_nums):
        n = noise_fn(si, pn * pn).view(B_local, pn * pn, FakeVar.V)
        acc += n.sum(dim=(1, 2)).view(B_local, 1, 1, 1)
    return acc
B = 6
labels = torch.arange(B) * 10
mode = os.environ['RNG_MODE']
out = multi.sample_sharded(FakeVar, B, labels, g_seed=5, rng_mode=mode, sample_fn=fake_sample)
if mode == 'exact':                 # must equal the single-process result row for row
    ref = multi.sample_sharded(FakeVar, B, labels, g_seed=5, rng_mode='exact', sample_fn=fake_sample, rank=0, world=1)
    assert torch.equal(out, ref), (out[:, 0, 0, 0], ref[:, 0, 0, 0])
else:
    lo, hi = multi.shard_range(B, dist.get_rank(), 2)
    mine = multi.sample_sharded(FakeVar, B, labels, g_seed=5, rng_mode='per_rank', sample_fn=fake_sample, gather=False)
    assert torch.equal(out[lo:hi], mine) and out.shape[0] == B
# bench.py's N > 1 statistics: job time = max over ranks, per-rank own time min / max, all-gather time max / min
r = dist.get_rank()
rs = multi.rank_stats(dt=2.0 + r, dt_own=1.0 + 0.5 * r, gather_ms=3.0 - r, steps=2, gathered_shape=(B, 3, 2, 2), device='cpu')
assert rs['dt_max'] == 3.0 and rs['ranks']['rank_ms_min'] == 500.0 and rs['ranks']['rank_ms_max'] == 750.0, rs
assert rs['ranks']['allgather_ms'] == 3.0 and rs['ranks']['allgather_ms_min'] == 2.0 and abs(rs['ranks']['allgather_mbytes'] - B * 12 * 4 / 1e6) < 0.05, rs
ev = []
out2 = multi.sample_sharded(FakeVar, B, labels, g_seed=5, rng_mode=mode, sample_fn=fake_sample, gather_events=ev)      # CPU tensors: no events, same result
assert torch.equal(out2, out) and ev == []
tdist.barrier()
print('rank', dist.get_rank(), 'ok')
'''


@pytest.mark.parametrize('mode', ['exact', 'per_rank'])
def test_sharded_sampling_two_ranks_gloo(mode, tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER)
    port = 29500 + (os.getpid() % 500) + (0 if mode == 'exact' else 1)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), VAR_ROOT=ROOT, RNG_MODE=mode)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f'rank {r} failed:\n{o}'


def test_eight_rank_partition_reproduces_the_one_gpu_stream():
    """BASELINE.json configs[2]'s partition at its real width (8 ranks, B = 512 -> 64 per rank), as far as one process can show it: every rank's shard of the
    'exact' RNG mode (each rank draws the whole Exp(1) fill and keeps its rows), concatenated in rank order, is the single-rank result row for row; 'per_rank'
    shards differ from it but are reproducible"""
    import torch
    from var_amd import multi

    class FakeVar:
        V = 32
        lvl_1L = torch.zeros(1)
        rng = torch.Generator()
        patch_nums = (1, 2, 3, 4)

    def fake_sample(B_local, labels, noise_fn):
        acc = labels.float().view(B_local, 1).clone()
        for si, pn in enumerate(FakeVar.patch_nums):
            acc = acc + noise_fn(si, pn * pn).view(B_local, -1).sum(dim=1, keepdim=True) * (si + 1)
        return acc

    B, W = 512, 8
    labels = (torch.arange(B) * 7) % 1000
    whole = multi.sample_sharded(FakeVar, B, labels, g_seed=3, rng_mode='exact', sample_fn=fake_sample, rank=0, world=1)
    parts = [multi.sample_sharded(FakeVar, B, labels, g_seed=3, rng_mode='exact', sample_fn=fake_sample, rank=r, world=W, gather=False) for r in range(W)]
    assert all(p.shape[0] == B // W for p in parts) and torch.equal(torch.cat(parts), whole)
    a = [multi.sample_sharded(FakeVar, B, labels, g_seed=3, rng_mode='per_rank', sample_fn=fake_sample, rank=r, world=W, gather=False) for r in range(W)]
    b = [multi.sample_sharded(FakeVar, B, labels, g_seed=3, rng_mode='per_rank', sample_fn=fake_sample, rank=r, world=W, gather=False) for r in range(W)]
    assert all(torch.equal(x, y) for x, y in zip(a, b)) and not torch.equal(torch.cat(a), whole)


def test_shard_range_rejects_ragged_batches():
    from var_amd.multi import shard_range
    assert shard_range(512, 3, 8) == (192, 256)
    with pytest.raises(ValueError):
        shard_range(10, 0, 4)


# ---- rank launcher (what `python bench.py --gpus N` uses outside torchrun) -----------------------------------------------------
_STUB_RANK = r'''
import json, os, sys
import torch.distributed as tdist
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
assert os.environ['MASTER_ADDR'] == '127.0.0.1' and os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'
tdist.init_process_group('gloo')                      # the stub stands in for bench.py's RCCL group; rendezvous must work
assert tdist.get_world_size() == world == int(sys.argv[sys.argv.index('--gpus') + 1])
tdist.barrier()
if 'fail' in sys.argv and rank == world - 1:
    sys.exit(7)
if rank == 0:
    print(json.dumps({'n_gpus': world, 'argv': sys.argv[1:]}), flush=True)
tdist.destroy_process_group()
'''


def test_spawn_ranks_starts_n_ranks_and_relays_rank0_line(tmp_path):
    from var_amd import launch
    script = tmp_path / 'stub_rank.py'
    script.write_text(_STUB_RANK)
    code = f'import sys; sys.path.insert(0, {ROOT!r}); from var_amd import launch; ' \
           f'sys.exit(launch.spawn_ranks({str(script)!r}, ["--gpus", "2", "--steps", "1"], 2, device_count=lambda: 2, timeout=200))'
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout
    import json
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['argv'] == ['--gpus', '2', '--steps', '1']
    # a failing rank makes the launcher's exit code non-zero; too few GPUs is refused before anything starts
    code_fail = code.replace('"--steps", "1"', '"fail"')
    out = subprocess.run([sys.executable, '-c', code_fail], capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode != 0
    assert launch.spawn_ranks(str(script), ['--gpus', '4'], 4, device_count=lambda: 1) == 2
    assert launch.under_launcher({'RANK': '0', 'WORLD_SIZE': '2'}) and not launch.under_launcher({})


def test_launcher_counts_gpus_from_kfd_topology_without_loading_hip(tmp_path):
    """launch._gpu_count reads /sys/class/kfd/kfd/topology/nodes/*/properties (GPU node: simd_count > 0 and an openable render node) and
    the *_VISIBLE_DEVICES lists; the launcher parent never imports torch or loads libamdhip64 for it"""
    from var_amd import launch
    nodes, dri = tmp_path / 'nodes', tmp_path / 'dri'
    dri.mkdir()
    for i, (simd, minor) in enumerate([(0, None), (0, None), (1024, 128), (1024, 129), (1024, 130)]):      # 2 CPU agents, 3 GPUs
        d = nodes / str(i); d.mkdir(parents=True)
        (d / 'properties').write_text(f'cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\n' + (f'drm_render_minor {minor}\n' if minor else ''))
    for minor in (128, 129):                                       # the third GPU's render node is not handed to this container
        (dri / f'renderD{minor}').write_text('')
    assert launch._gpu_count(str(nodes), env={}, dri_dir=str(dri)) == 2
    assert launch._gpu_count(str(nodes), env={'ROCR_VISIBLE_DEVICES': '1'}, dri_dir=str(dri)) == 1
    assert launch._gpu_count(str(nodes), env={'HIP_VISIBLE_DEVICES': '0,1,2,3'}, dri_dir=str(dri)) == 2
    assert launch._gpu_count(str(nodes), env={'CUDA_VISIBLE_DEVICES': ''}, dri_dir=str(dri)) == 0
    assert launch._gpu_count(str(tmp_path / 'absent'), env={}) == 0
    code = (f'import sys; sys.path.insert(0, {ROOT!r}); from var_amd import launch; n = launch._gpu_count(); '
            'maps = open("/proc/self/maps").read(); '
            'assert "torch" not in sys.modules and "libamdhip64" not in maps and "libhsa-runtime" not in maps, "the launcher parent loaded a GPU runtime"; print(n)')
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr[-1500:]


def test_bench_self_launches_when_asked_for_more_gpus_than_ranks(monkeypatch):
    """bench.py --gpus 2 outside torchrun must go through the launcher (and therefore fail here: this container has no GPU), never run
    one rank and report n_gpus 1."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                         capture_output=True, text=True, timeout=240, env=env)
    if torch.cuda.device_count() < 2:
        assert out.returncode == 2 and 'ranks requested' in out.stderr and not any(ln.startswith('{') for ln in out.stdout.splitlines())


def test_device_side_detinit_matches_numpy():
    """var_amd.detinit.fill_module_device_ (torch int64 ops where the parameter lives; what the 2 B-parameter GPU tests use) must
    produce the bits of the numpy generator the fixtures were made with"""
    from var_amd.detinit import fill_module_, fill_module_device_, hash_uniform, hash_uniform_torch
    a = hash_uniform('var.blocks.0.attn.proj.weight', (1 << 18) + 5, 3)
    b = hash_uniform_torch('var.blocks.0.attn.proj.weight', (1 << 18) + 5, 3, 'cpu', chunk=1 << 16).numpy()
    assert np.array_equal(a, b)
    v1, m1 = _quiet_build(depth=2, ch=32, patch_nums=(1, 2, 3), shared_aln=True)
    v2, m2 = _quiet_build(depth=2, ch=32, patch_nums=(1, 2, 3), shared_aln=True)
    fill_module_(m1, 2, 0, 'var.'); fill_module_device_(m2, 2, 0, 'var.')
    fill_module_(v1, 2, 0, 'vae.'); fill_module_device_(v2, 2, 0, 'vae.')
    for x, y in ((m1, m2), (v1, v2)):
        for (k, p), q in zip(x.state_dict().items(), y.state_dict().values()):
            assert torch.equal(p, q), k
