#!/usr/bin/env python3
"""Sampling harness — the build's counterpart of the reference's demo_sample.py:43-73 (row H of SURVEY.md §8a).

Builds VAE + VAR with build_vae_var (same kwargs as the demo), optionally loads the published checkpoints with strict=True,
samples class-conditional images with VAR.autoregressive_infer_cfg on the MI355X HIP path and writes a PNG grid.

  python tools/sample.py --depth 16 --labels 980 980 437 437 22 22 562 562 --cfg 4 --top-k 900 --top-p 0.95 --seed 0 \
         [--precision f16] [--vae-ckpt vae_ch160v4096z32.pth --var-ckpt var_d16.pth] --out sample.png
Without checkpoints (there is no network here) the deterministic random-init weights of var_amd.detinit are used."""
import argparse
import contextlib
import io
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_grid(img_B3HW: torch.Tensor, nrow: int = 8, pad: int = 2) -> np.ndarray:
    """torchvision.utils.make_grid(nrow=8, padding=0 in the demo; here pad is configurable) -> uint8 HWC"""
    B, C, H, W = img_B3HW.shape
    ncol = min(nrow, B); nr = (B + ncol - 1) // ncol
    grid = torch.zeros(C, nr * (H + pad) + pad, ncol * (W + pad) + pad)
    for i in range(B):
        r, c = divmod(i, ncol)
        grid[:, pad + r * (H + pad): pad + r * (H + pad) + H, pad + c * (W + pad): pad + c * (W + pad) + W] = img_B3HW[i].cpu()
    return grid.permute(1, 2, 0).mul(255).clamp(0, 255).round().to(torch.uint8).numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--depth', type=int, default=16)
    ap.add_argument('--labels', type=int, nargs='+', default=[980, 980, 437, 437, 22, 22, 562, 562])
    ap.add_argument('--cfg', type=float, default=4.0)
    ap.add_argument('--top-k', type=int, default=900)
    ap.add_argument('--top-p', type=float, default=0.95)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--more-smooth', action='store_true')
    ap.add_argument('--precision', default='f32', choices=['f32', 'f16', 'bf16', 'auto'],
                    help="f32: token ids bit-identical to the CPU oracle; f16: what the demo's torch.autocast(dtype=float16) asks for (demo_sample.py:66-68); bf16: the same with bfloat16; "
                         "auto: exactly the demo — the call sits inside torch.autocast('cuda', dtype=torch.float16) and the engine follows it")
    ap.add_argument('--vae-ckpt'); ap.add_argument('--var-ckpt')
    ap.add_argument('--out', default='sample.png')
    a = ap.parse_args()

    from models import build_vae_var
    from var_amd.detinit import fill_module_
    patch_nums = (1, 2, 3, 4, 5, 6, 8, 10, 13, 16)
    with contextlib.redirect_stdout(io.StringIO()):
        vae, var = build_vae_var(V=4096, Cvae=32, ch=160, share_quant_resi=4, device='cuda', patch_nums=patch_nums, num_classes=1000,
                                 depth=a.depth, shared_aln=False)
    if a.vae_ckpt and a.var_ckpt:
        vae.load_state_dict(torch.load(a.vae_ckpt, map_location='cpu'), strict=True)
        var.load_state_dict(torch.load(a.var_ckpt, map_location='cpu'), strict=True)
    else:
        print('[sample] no checkpoints given: deterministic random-init weights (the images are noise-like textures)')
        fill_module_(var, a.depth, 0, 'var.'); fill_module_(vae, a.depth, 0, 'vae.')
    vae.eval(); var.eval()
    for p in list(vae.parameters()) + list(var.parameters()): p.requires_grad_(False)
    torch.manual_seed(a.seed); np.random.seed(a.seed)
    labels = torch.tensor(a.labels, device='cuda')
    var.set_hip_precision(a.precision)
    with torch.inference_mode():
        with torch.autocast('cuda', enabled=a.precision == 'auto', dtype=torch.float16, cache_enabled=True):       # (demo_sample.py:66)
            img = var.autoregressive_infer_cfg(B=len(a.labels), label_B=labels, cfg=a.cfg, top_k=a.top_k, top_p=a.top_p, g_seed=a.seed, more_smooth=a.more_smooth)
    grid = make_grid(img)
    try:
        from PIL import Image
        Image.fromarray(grid).save(a.out)
    except ImportError:
        a.out = os.path.splitext(a.out)[0] + '.npy'; np.save(a.out, grid)
    print(f'[sample] {tuple(img.shape)} -> {a.out}')


if __name__ == '__main__':
    main()
