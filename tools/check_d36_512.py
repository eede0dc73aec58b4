#!/usr/bin/env python3
"""BASELINE.json configs[4] on ONE GPU: VAR-d36 (C=2304, 36 heads, shared AdaLN, 2.35 B parameters), 512x512 (patch_nums up to 32,
L=2240, KV cache reused across the 10 scales), random-init weights; --dtype f16 is the precision that config names.  Prints one JSON
line: images/s of the timed calls, TFLOP/s at the SURVEY.md §8d count (24.39 TFLOP per image), peak memory.  (The 8-GPU run of this
config is the driver's; per-GPU work is fixed, so this is what every rank does.)

    python tools/check_d36_512.py [--batch 8] [--dtype f16] [--calls 2]
"""
import argparse, contextlib, io, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from models import build_vae_var
from var_amd.detinit import fill_module_device_

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--dtype', default='f32', choices=['f32', 'f16', 'bf16'])
ap.add_argument('--calls', type=int, default=2)
args = ap.parse_args()
B = args.batch
pns = (1, 2, 3, 4, 6, 9, 13, 18, 24, 32)
t0 = time.time()
with contextlib.redirect_stdout(io.StringIO()):
    vae, var = build_vae_var(device='cuda', patch_nums=pns, depth=36, shared_aln=True, ch=160)
fill_module_device_(var, 36, 0, 'var.'); fill_module_device_(vae, 36, 0, 'vae.')
var.eval(); vae.eval(); var.set_hip_precision(args.dtype)
nparam = sum(p.numel() for p in var.parameters())
labels = torch.arange(B, device='cuda') * 37 % 1000
times = []
with torch.inference_mode():
    for it in range(args.calls + 1):
        torch.cuda.synchronize(); t = time.time()
        img = var.autoregressive_infer_cfg(B, labels, g_seed=it, cfg=1.5, top_k=900, top_p=0.96)
        torch.cuda.synchronize(); times.append(time.time() - t)
assert img.shape == (B, 3, 512, 512) and bool(torch.isfinite(img).all()) and float(img.min()) >= 0 and float(img.max()) <= 1
eng = var.engine()
flops_img = eng.flops_per_image() + eng.dec.flops_per_image_reference(pns[-1])
dt = min(times[1:])
peak = 2500.0 if args.dtype != 'f32' else 157.3
print(json.dumps({'config': 'VAR-d36 512x512 (patch_nums up to 32, L=2240), shared AdaLN, CFG=1.5, top_k=900, top_p=0.96, 1 GPU', 'dtype': args.dtype, 'batch': B,
                  'parameters_B': round(nparam / 1e9, 2), 'images_per_sec': round(B / dt, 3), 'sec_per_call': round(dt, 3), 'first_call_sec': round(times[0], 2),
                  'tflop_per_image': round(flops_img / 1e12, 2), 'tflops': round(B / dt * flops_img / 1e12, 1), 'frac_of_mfma_peak': round(B / dt * flops_img / 1e12 / peak, 4),
                  'peak_hbm_gib': round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), 'setup_sec': round(time.time() - t0 - sum(times), 1)}))
