#!/usr/bin/env python3
"""Functional check of the largest published configuration on one GPU: VAR-d36 (C=2304, 36 heads, shared AdaLN), 512x512
(patch_nums up to 32, L=2240), small batch, fp32, random-init weights (BASELINE.json configs[4] names it as an fp16 8-GPU run; this
only shows that every kernel takes those shapes).  Prints the image shape, finiteness and the time of the second call."""
import contextlib, io, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from models import build_vae_var
from var_amd.detinit import fill_module_

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pns = (1, 2, 3, 4, 6, 9, 13, 18, 24, 32)
t0 = time.time()
with contextlib.redirect_stdout(io.StringIO()):
    vae, var = build_vae_var(device='cuda', patch_nums=pns, depth=36, shared_aln=True, ch=160)
fill_module_(var, 36, 0, 'var.'); fill_module_(vae, 36, 0, 'vae.')
var.eval(); vae.eval()
print(f'[d36-512] model ready in {time.time() - t0:.0f} s, {sum(p.numel() for p in var.parameters()) / 1e9:.2f} B parameters', flush=True)
labels = torch.arange(B, device='cuda') * 37 % 1000
with torch.inference_mode():
    for it in range(2):
        torch.cuda.synchronize(); t = time.time()
        img = var.autoregressive_infer_cfg(B, labels, g_seed=it, cfg=1.5, top_k=900, top_p=0.96)
        torch.cuda.synchronize(); dt = time.time() - t
        print(f'[d36-512] call {it}: {tuple(img.shape)} finite={bool(torch.isfinite(img).all())} min={float(img.min()):.3f} max={float(img.max()):.3f} '
              f'{dt:.2f} s = {B / dt:.2f} images/s, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB', flush=True)
