#!/usr/bin/env python3
"""What the 'exact' RNG mode of var_amd.multi costs a rank of an 8-GPU run, measured on ONE GPU (run on the GPU box).

In 'exact' mode every rank draws the full (B_total*l, V) Exp(1) fill per scale and keeps its own rows, so that the token stream
equals the single-GPU stream with batch B_total (reference helpers.py:19 draws that fill row-major from one generator).  One GPU
can time exactly what rank r of W does: sample_sharded(..., rank=r, world=W, gather=False).

    python tools/time_rng_modes.py [--batch 64] [--world 8] [--steps 3] > gpurun_out/rng_modes.json
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64, help='images per rank')
ap.add_argument('--world', type=int, default=8)
ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--depth', type=int, default=16)
args = ap.parse_args()

from models import build_vae_var                       # noqa: E402
from var_amd.detinit import fill_module_device_        # noqa: E402
from var_amd.multi import sample_sharded               # noqa: E402

torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
with contextlib.redirect_stdout(io.StringIO()):
    vae, var = build_vae_var(device=dev, depth=args.depth, ch=160)
fill_module_device_(var, args.depth, 0, 'var.'); fill_module_device_(vae, args.depth, 0, 'vae.')
var.eval(); var.rng = torch.Generator(device=dev)


def run(mode, world, rank):
    B_total = args.batch * world
    labels = ((torch.arange(B_total) * 7) % 1000).to(dev)
    f = lambda i: sample_sharded(var, B_total, labels, g_seed=i, cfg=1.5, top_k=900, top_p=0.96, rng_mode=mode, gather=False, rank=rank, world=world)
    f(0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps): f(100 + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.steps * 1e3


res = {'batch_per_rank': args.batch, 'depth': args.depth, 'steps': args.steps}
res['ms_world1'] = run('exact', 1, 0)
res[f'ms_exact_world{args.world}_rank0'] = run('exact', args.world, 0)
res[f'ms_exact_world{args.world}_last_rank'] = run('exact', args.world, args.world - 1)
res[f'ms_per_rank_world{args.world}'] = run('per_rank', args.world, 0)
res['exact_overhead_pct'] = round(100.0 * (max(res[f'ms_exact_world{args.world}_rank0'], res[f'ms_exact_world{args.world}_last_rank']) / res['ms_world1'] - 1.0), 2)
res['peak_hbm_gib'] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)
print(json.dumps(res))
