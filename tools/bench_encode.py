#!/usr/bin/env python3
"""Timing of the encode side + teacher-forced likelihood pass (run on the GPU box):

    image --VQVAE.img_to_idxBl--> token maps --idxBl_to_var_input--> teacher-forcing input --VAR.forward--> logits (B, L, V)

i.e. what the fork's eval_prob.py does per image and candidate class (reference eval_prob.py:420-446) and what trainer.eval_ep does per
batch (trainer.py:66-71), at B images of 256x256 on one MI355X, random-init weights (detinit seed 0), fp32 (the parity mode; --dtype f16
switches VAR.forward's transformer to the 16-bit mode, the encoder stays fp32).

    python tools/bench_encode.py [--batch 64] [--iters 5] [--dtype f32]

Prints one JSON object: ms per stage (CUDA events on the launch stream), TFLOP/s of the encoder and of VAR.forward against the MFMA peak of
their arithmetic type, and for the nearest-code kernel (library family timing, HIP events around each launch) its time per call over the
ten scales, TFLOP/s of its score GEMM and the bytes it moves."""
import argparse
import contextlib
import io
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from var_amd import detinit, hip      # noqa: E402

PEAK32, PEAK16 = 157.3, 2500.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--depth', type=int, default=16)
    ap.add_argument('--dtype', default='f32', choices=['f32', 'f16', 'bf16'])
    a = ap.parse_args()
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    from models import build_vae_var
    pns = (1, 2, 3, 4, 5, 6, 8, 10, 13, 16)
    with contextlib.redirect_stdout(io.StringIO()):
        vae, var = build_vae_var(device=dev, patch_nums=pns, depth=a.depth, ch=160)
    detinit.fill_module_device_(var, a.depth, 0, 'var.'); detinit.fill_module_device_(vae, a.depth, 0, 'vae.')
    var.eval(); vae.eval(); var.cond_drop_rate = 0.0
    var.set_hip_precision(a.dtype)
    B = a.batch
    g = torch.Generator(device=dev).manual_seed(0)
    img = (torch.rand(B, 3, 256, 256, device=dev, generator=g) * 2 - 1)
    labels = ((torch.arange(B) * 7) % 1000).to(dev)

    def ev(): return torch.cuda.Event(enable_timing=True)
    stages = {'img_to_post (encoder + quant_conv)': [], 'f_to_idxBl (10 scales: pool, nearest code, residual)': [], 'idxBl_to_var_input': [],
              'VAR.forward (teacher-forced, L=680)': []}
    nc_ms, nc_calls = 0.0, 0
    with torch.inference_mode():
        for it in range(a.iters + 1):
            e = [ev() for _ in range(5)]
            e[0].record()
            f = vae.img_to_post(img)
            e[1].record()
            if it: hip.timing_reset(); hip.timing_enable(True, ['other'])
            idx = vae.quantize.f_to_idxBl_or_fhat(f, to_fhat=False)
            if it:
                torch.cuda.synchronize(); hip.timing_enable(False)
            e[2].record()
            x = vae.quantize.idxBl_to_var_input(idx)
            e[3].record()
            logits = var(labels, x)
            e[4].record()
            torch.cuda.synchronize()
            if it:                                        # iteration 0 warms up (weight packing, workspaces)
                for k, (s, t) in zip(stages, zip(e[:-1], e[1:])): stages[k].append(s.elapsed_time(t))
    assert logits.shape == (B, 680, 4096) and bool(torch.isfinite(logits).all()) and all(i.dtype == torch.int64 for i in idx)
    # the nearest-code kernel alone: the ten per-scale calls of one encode, timed back to back (the family table above also holds the pool /
    # residual kernels of the quantizer loop)
    q = vae.quantize.hip_engine(); q.refresh()
    N_all = [B * pn * pn for pn in pns]
    zs = [torch.randn(n, 32, device=dev, generator=g) for n in N_all]
    outs = [torch.empty(n, dtype=torch.int64, device=dev) for n in N_all]
    for rep in range(3):
        s, t = ev(), ev()
        s.record()
        for z, o, n in zip(zs, outs, N_all): hip.call('nearest_code_f32', z, q.codebook, o, n, 4096, 32)
        t.record(); torch.cuda.synchronize()
        nc_ms = s.elapsed_time(t)
    Ntot = sum(N_all)
    nc_flops = 2.0 * Ntot * 4096 * 32
    nc_bytes = 4.0 * Ntot * 32 + 8.0 * Ntot + 10 * 4096 * 32 * 4.0          # queries + indices + the codebook once per launch
    med = {k: sorted(v)[len(v) // 2] for k, v in stages.items()}
    eng = var.engine()
    fwd_flops = eng.flops_per_image() / 2.0 * B                 # teacher forcing: B rows (no CFG pair)
    enc_flops = enc_flops_per_image(vae) * B
    peak_fwd = PEAK16 if a.dtype != 'f32' else PEAK32
    out = {'workload': f'encode + teacher-forced likelihood, VAR-d{a.depth}, B={B} images 256x256, random-init (detinit seed 0)', 'dtype': a.dtype,
           'ms': {k: round(v, 3) for k, v in med.items()}, 'ms_total': round(sum(med.values()), 3), 'images_per_sec': round(B / sum(med.values()) * 1e3, 1),
           'encoder': {'gflop_per_image': round(enc_flops / B / 1e9, 1), 'tflops': round(enc_flops / med['img_to_post (encoder + quant_conv)'] / 1e9, 1),
                       'frac_of_fp32_mfma_peak': round(enc_flops / med['img_to_post (encoder + quant_conv)'] / 1e9 / PEAK32, 3)},
           'var_forward': {'gflop_per_image': round(fwd_flops / B / 1e9, 1), 'tflops': round(fwd_flops / med['VAR.forward (teacher-forced, L=680)'] / 1e9, 1),
                           'frac_of_mfma_peak': round(fwd_flops / med['VAR.forward (teacher-forced, L=680)'] / 1e9 / peak_fwd, 3), 'peak_tflops': peak_fwd},
           'nearest_code': {'kernel': 'k_nearest_mfma<false>', 'calls': 10, 'queries': Ntot, 'ms_all_ten_scales': round(nc_ms, 4),
                            'tflops_score_gemm': round(nc_flops / nc_ms / 1e9, 2), 'algorithmic_gbytes_per_sec': round(nc_bytes / nc_ms / 1e6, 1),
                            'note': 'scores z.e on the fp32 MFMA (k-ascending chain = the oracle\'s), 128 queries per workgroup, codebook staged through LDS in '
                                    '512-code shards; compute-bound (32 channels): 2 FLOP x 4096 codes x 32 channels per 136 query bytes'}}
    print(json.dumps(out), flush=True)


def enc_flops_per_image(vae) -> float:
    """2 FLOP per MAC over every conv / linear of the encoder + quant_conv at 256x256 (reference basic_vae.py:99-160)"""
    import torch.nn as nn
    tot = [0.0]
    hooks = []

    def hook(m, inp, out):
        if isinstance(m, nn.Conv2d):
            tot[0] += 2.0 * out.numel() / out.shape[0] * m.in_channels * m.kernel_size[0] * m.kernel_size[1]
    import copy
    enc, qc = copy.deepcopy(vae.encoder).to('meta'), copy.deepcopy(vae.quant_conv).to('meta')      # shapes only: no kernel runs
    for m in list(enc.modules()) + [qc]:
        if isinstance(m, nn.Conv2d): hooks.append(m.register_forward_hook(hook))
    with torch.no_grad():
        qc(enc(torch.zeros(1, 3, 256, 256, device='meta')))  # only to count MACs (the attention block's bmm's, 0.5 GMAC, are not counted)
    for h in hooks: h.remove()
    return tot[0]


if __name__ == '__main__':
    main()
