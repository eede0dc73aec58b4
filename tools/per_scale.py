#!/usr/bin/env python3
"""Per-scale cost table of one sampling call (SURVEY.md §8d asks for the HBM-bound early scales to be visible): for each of the
10 scales the device time of its transformer pass (16 blocks + head + sampler + quantizer step), the algorithmic TFLOP/s, and the rate at
which the scale streams the block weights (every scale reads all of them once: 1.21 GB fp32 / 0.60 GB fp16 at d16); then the decoder.
Timed with events on the launch stream inside the engine (SamplingEngine.profile_scales), median over the repetitions.

    python tools/per_scale.py [--dtype f16] [--batch 64] [--depth 16] [--reps 5] > gpurun_out/per_scale.json
"""
import argparse
import contextlib
import io
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('--dtype', default='f32', choices=['f32', 'f16', 'bf16'])
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--depth', type=int, default=16)
ap.add_argument('--reps', type=int, default=5)
args = ap.parse_args()

from models import build_vae_var                      # noqa: E402
from var_amd.detinit import fill_module_device_       # noqa: E402

torch.cuda.set_device(0)
pns = (1, 2, 3, 4, 5, 6, 8, 10, 13, 16)
with contextlib.redirect_stdout(io.StringIO()):
    vae, var = build_vae_var(device='cuda', patch_nums=pns, depth=args.depth, ch=160)
fill_module_device_(var, args.depth, 0, 'var.'); fill_module_device_(vae, args.depth, 0, 'vae.')
var.eval(); var.set_hip_precision(args.dtype)
eng = var.engine()
B = args.batch
labels = ((torch.arange(B) * 7) % 1000).cuda()
var.rng = torch.Generator(device='cuda')
var.autoregressive_infer_cfg(B, labels, g_seed=0, cfg=1.5, top_k=900, top_p=0.96)
eng.profile_scales = True
runs = []
for r in range(args.reps):
    var.autoregressive_infer_cfg(B, labels, g_seed=1 + r, cfg=1.5, top_k=900, top_p=0.96)
    runs.append(eng.last_scale_ms)
C, depth, V = var.C, var.depth, var.V
wbytes = depth * 12 * C * C * (2 if args.dtype != 'f32' else 4) + C * V * (2 if args.dtype != 'f32' else 4)     # block + head weights read per scale
rows, cur = [], 0
for si, pn in enumerate(pns):
    l = pn * pn; cur += l
    ms = statistics.median(r[si] for r in runs)
    rows_tok = 2 * B * l
    flops = 2.0 * 12 * C * C * depth * rows_tok + 2.0 * C * V * rows_tok + 4.0 * 2 * B * l * cur * C * depth
    rows.append({'scale': si, 'pn': pn, 'rows': rows_tok, 'ms': round(ms, 3), 'tflops': round(flops / ms / 1e9, 1),
                 'weight_stream_GBps': round(wbytes / ms / 1e6, 1), 'flop_per_weight_byte': round(flops / wbytes, 1)})
dec_ms = statistics.median(r[len(pns)] for r in runs)
dec_fl = eng.dec.flops_per_image_executed(pns[-1]) * B
out = {'dtype': args.dtype, 'batch': B, 'depth': args.depth, 'weight_bytes_per_scale': wbytes, 'scales': rows,
       'decoder': {'ms': round(dec_ms, 3), 'tflops_executed': round(dec_fl / dec_ms / 1e9, 1)},
       'total_ms': round(sum(x['ms'] for x in rows) + dec_ms, 3),
       'note': 'ridge of the machine: 157.3 TF / 6.3 TB/s = 25 FLOP per byte in fp32, 2500 / 6.3 = 400 in fp16: scales whose flop_per_weight_byte is '
               'below it are bound by streaming the weights from HBM, not by the matrix pipe'}
print(json.dumps(out, indent=1))
