"""Shared helpers of the test-suite: deterministic weights, golden loading, noise regeneration, comparison reports."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from var_amd import shapes                          # noqa: E402
from var_amd.detinit import make_state_dict         # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
_WCACHE = {}


def ensure_oracle_built():
    so = os.path.join(ROOT, 'oracle', 'libvar_oracle.so')
    src = os.path.join(ROOT, 'oracle', 'var_oracle.c')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle')], stdout=subprocess.DEVNULL)


def load_case(name):
    z = np.load(os.path.join(GOLD, f'e2e_{name}.npz'))
    meta = json.loads(str(z['meta']))
    return z, meta


def make_weights(meta, seed=0, include_encoder=False):
    """Deterministic VAR + VQVAE weights in the reference's state-dict layout (numpy), cached per config."""
    key = (meta['depth'], meta['ch'], tuple(meta['patch_nums']), meta['attn_l2_norm'], meta['shared_aln'], seed, include_encoder)
    if key in _WCACHE:
        return _WCACHE[key]
    pns = tuple(meta['patch_nums'])
    vs = shapes.var_shapes(meta['depth'], pns, shared_aln=meta['shared_aln'], attn_l2_norm=meta['attn_l2_norm'])
    var_sd = make_state_dict(vs, depth=meta['depth'], seed=seed, prefix='var.')
    var_sd['lvl_1L'] = np.concatenate([np.full((p * p,), i, dtype=np.int64) for i, p in enumerate(pns)]).reshape(1, -1)
    es = shapes.vae_shapes(ch=meta['ch'], patch_nums=pns, include_encoder=include_encoder)
    vae_sd = make_state_dict(es, depth=meta['depth'], seed=seed, prefix='vae.')
    _WCACHE.clear()          # keep at most one config resident (d16 is 1.6 GB)
    _WCACHE[key] = (var_sd, vae_sd)
    return var_sd, vae_sd


def regen_noise(meta, z=None):
    """The Exp(1) fills torch.multinomial(n=1) consumed in the reference run, one (B*l, V) fill per scale, regenerated
    with a CPU torch.Generator (same wheel on both boxes) and checked against the fixture's head/checksum."""
    import torch
    g = torch.Generator(device='cpu')
    g.manual_seed(meta['seed'])
    out = []
    for si, pn in enumerate(meta['patch_nums']):
        q = torch.empty(meta['B'] * pn * pn, meta['V'], dtype=torch.float32).exponential_(1, generator=g)
        if z is not None:
            assert np.array_equal(q.view(-1)[:8].numpy(), z['noise_head'][si]), 'torch CPU RNG stream differs from the fixture'
            assert abs(q.double().sum().item() - float(z['noise_sum'][si])) <= 1e-6 * abs(float(z['noise_sum'][si]))
        out.append(q.numpy())
    return out


def diff_report(name, got, want, atol=0.0, rtol=0.0):
    """Returns (ok, message) with max abs error, mismatch count and first mismatch index."""
    got = np.asarray(got); want = np.asarray(want)
    if got.shape != want.shape:
        return False, f'{name}: shape {got.shape} != {want.shape}'
    if got.dtype.kind in 'iub':
        bad = got != want
        nbad = int(bad.sum())
        return nbad == 0, f'{name}: {nbad}/{got.size} mismatches' + (f', first at {tuple(np.argwhere(bad)[0])}' if nbad else '')
    g64, w64 = got.astype(np.float64), want.astype(np.float64)
    with np.errstate(invalid='ignore'):
        close = np.abs(g64 - w64) <= atol + rtol * np.abs(w64)
    bad = ~(close | (got == want) | (np.isnan(got) & np.isnan(want)))
    nbad = int(bad.sum())
    d = np.abs(g64 - w64)
    fin = np.isfinite(d)
    mx = float(d[fin].max()) if fin.any() else 0.0
    wfin = np.isfinite(w64)
    msg = f'{name}: max|d|={mx:.3e} (|want| max {float(np.abs(w64[wfin]).max()) if wfin.any() else 0:.3e}), {nbad}/{got.size} outside atol={atol:g} rtol={rtol:g}'
    if nbad:
        i = tuple(np.argwhere(bad)[0]); msg += f', first at {i}: got {got[i]!r} want {want[i]!r}'
    return nbad == 0, msg


def free_port() -> int:
    """a TCP port nobody listens on right now (rendezvous of child process groups on 127.0.0.1)"""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]
