#!/usr/bin/env python3
"""Micro-benchmarks of the hot kernels at the shapes the d16 / B=64 sampling call launches (run on the GPU box).

  python tools/bench_kernels.py gemm|qkv|conv|attn|all [--iters N]
Times each shape with torch CUDA events on the launch stream, interleaving shapes over rounds (cdna guide rule 24), and
prints TFLOP/s against the fp32 MFMA peak.  Used to A/B kernel variants; numbers quoted in DESIGN.md come from bench.py."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from var_amd import hip      # noqa: E402

PEAK = 157.3


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def gemm_shapes():
    B2 = 128
    out = []
    for l in (256, 169, 100, 64, 36, 25, 16, 9, 4, 1):
        M = B2 * l
        out += [(f'qkv  l={l}', M, 3072, 1024, 0), (f'proj l={l}', M, 1024, 1024, 2), (f'fc1  l={l}', M, 4096, 1024, 1), (f'fc2  l={l}', M, 1024, 4096, 2)]
    out.append(('square 4096', 4096, 4096, 4096, 0))
    out.append(('square 8192', 8192, 8192, 8192, 0))
    return out


def run_gemm(iters, rounds=3):
    dev = 'cuda'
    res = {}
    bufs = {}
    for name, M, N, K, epi in gemm_shapes():
        A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.03; b = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev); resid = torch.randn(M, N, device=dev); gamma = torch.randn(128, N, device=dev)
        bufs[name] = (A, W, b, out, resid, gamma)
    for r in range(rounds):
        for name, M, N, K, epi in gemm_shapes():
            A, W, b, out, resid, gamma = bufs[name]
            rpg = max(M // 128, 1)
            fn = lambda: hip.call('gemm_nt_f32', A, K, W, K, b, out, N, M, N, K, epi, resid if epi == 2 else None, N, gamma if epi == 2 else None, N, rpg, 0, 1, 0, 0, 0)
            ms = timeit(fn, iters if M * N * K > 1e10 else iters * 4)
            res.setdefault(name, []).append(ms)
    for name, M, N, K, epi in gemm_shapes():
        ms = min(res[name]); tf = 2.0 * M * N * K / ms / 1e9
        print(f'gemm {name:14s} M={M:6d} N={N:5d} K={K:5d} epi={epi}: {ms*1e3:9.1f} us  {tf:7.1f} TF  {tf/PEAK*100:5.1f}%', flush=True)


def run_qkv(iters, rounds=3):
    """mat_qkv with the fused q/k/v epilogue (varhip_gemm_qkv_f32) at the d16 / B=64 shapes"""
    dev, B2, H, C, K, Lmax = 'cuda', 128, 16, 1024, 1024, 680
    W = torch.randn(3 * C, K, device=dev) * 0.03; bias = torch.randn(3 * C, device=dev); smul = torch.full((H,), 1.386, device=dev)
    kc = torch.empty(B2, H, Lmax, 64, device=dev); vc = torch.empty_like(kc)
    res, pos = {}, 0
    for r in range(rounds):
        pos = 0
        for pn in (1, 2, 3, 4, 5, 6, 8, 10, 13, 16):
            l = pn * pn; M = B2 * l
            A = torch.randn(M, K, device=dev); q = torch.empty(M, C, device=dev); p0 = pos
            fn = lambda: hip.call('gemm_qkv_f32', A, K, W, K, bias, M, C, K, smul, 1.0, 1, q, kc, vc, B2, l, H, p0, Lmax)
            res.setdefault(l, []).append(timeit(fn, iters if l >= 36 else iters * 4))
            pos += l
    tot = 0.0
    for l, v in res.items():
        ms = min(v); tf = 2.0 * B2 * l * 3 * C * K / ms / 1e9; tot += ms
        print(f'qkvfused l={l:3d}: {ms*1e3:9.1f} us  {tf:7.1f} TF  {tf/PEAK*100:5.1f}%', flush=True)
    print(f'qkvfused total {tot*1e3:.1f} us per layer', flush=True)


def run_conv(iters):
    dev = 'cuda'
    for (B, H, W, Cin, Cout, up2, res) in [(64, 256, 256, 160, 160, 0, 1), (64, 256, 256, 160, 160, 1, 0), (64, 128, 128, 320, 160, 0, 0), (64, 128, 128, 160, 160, 0, 1),
                                           (64, 128, 128, 320, 320, 1, 0), (64, 64, 64, 320, 320, 0, 1), (64, 32, 32, 640, 320, 0, 0), (64, 16, 16, 640, 640, 0, 1),
                                           (64, 256, 256, 160, 3, 0, 0)]:
        Hi, Wi = (H // 2, W // 2) if up2 else (H, W)
        x = torch.randn(B, Hi, Wi, Cin, device=dev); w = torch.randn(Cout, 3, 3, Cin, device=dev) * 0.02; b = torch.randn(Cout, device=dev)
        r = torch.randn(B, H, W, Cout, device=dev) if res else None
        out = torch.empty(B, H, W, Cout, device=dev)
        fn = lambda: hip.call('conv3x3_nhwc_f32', x, w, b, r, out, B, H, W, Cin, Cout, up2, 0)
        ms = timeit(fn, max(iters // 4, 2)); tf = 2.0 * B * H * W * Cout * 9 * Cin / ms / 1e9
        print(f'conv {Cin:3d}->{Cout:3d} {H:3d}x{W:3d} up{up2} res{res}: {ms:8.3f} ms  {tf:7.1f} TF  {tf/PEAK*100:5.1f}%', flush=True)
        del x, w, out, r


def run_attn(iters):
    dev = 'cuda'
    B2, H, Lmax = 128, 16, 680
    kc = torch.randn(B2, H, Lmax, 64, device=dev); vc = torch.randn(B2, H, Lmax, 64, device=dev)
    cur = 0
    for pn in (1, 2, 3, 4, 5, 6, 8, 10, 13, 16):
        l = pn * pn; cur += l
        q = torch.randn(B2 * l, H * 64, device=dev); out = torch.empty_like(q)
        fn = lambda: hip.call('attn_cached_f32', q, kc, vc, out, B2, l, H, cur, Lmax)
        ms = timeit(fn, iters); tf = 4.0 * B2 * H * l * cur * 64 / ms / 1e9
        print(f'attn l={l:3d} curL={cur:3d}: {ms*1e3:9.1f} us  {tf:6.1f} TF (algorithmic)', flush=True)


def run_gemm16(iters, rounds=3):
    """the 16-bit mode's GEMM (varhip_gemm_nt_f16) at the d16 / B=64 shapes; TFLOP/s against the 2.5 PF dense fp16 peak"""
    dev = 'cuda'
    res, bufs = {}, {}
    shapes = [s_ for s_ in gemm_shapes() if int(os.environ.get('GEMM16_MINM', 2048)) <= s_[1] <= int(os.environ.get('GEMM16_MAXM', 1 << 30))]
    for name, M, N, K, epi in shapes:
        A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * 0.03).half(); b = torch.randn(N, device=dev)
        out16 = torch.empty(M, N, device=dev, dtype=torch.float16); out32 = torch.empty(M, N, device=dev); resid = torch.randn(M, N, device=dev); gamma = torch.randn(128, N, device=dev)
        bufs[name] = (A, W, b, out16, out32, resid, gamma)
    tiles = [int(t) for t in os.environ.get('GEMM16_TILES', '-1').split(',')]
    for r in range(rounds):
        for name, M, N, K, epi in shapes:
            A, W, b, out16, out32, resid, gamma = bufs[name]
            rpg = max(M // 128, 1)
            o16 = 0 if epi == 2 else 1
            fn = lambda: hip.call('gemm_nt_f16', A, K, W, K, b, out16 if o16 else out32, N, o16, M, N, K, epi, resid if epi == 2 else None, N, 0, gamma if epi == 2 else None, N, rpg, 1, 0, 0, 0)
            for t in tiles:
                hip.lib().so.varhip_gemm16_force_tile(t)
                res.setdefault((name, t), []).append(timeit(fn, iters * 2))
    hip.lib().so.varhip_gemm16_force_tile(-1)
    for name, M, N, K, epi in shapes:
        line = f'gemm16 {name:14s} M={M:6d} N={N:5d} K={K:5d} epi={epi}:'
        for t in tiles:
            ms = min(res[(name, t)]); tf = 2.0 * M * N * K / ms / 1e9
            line += f'  [tile {t:2d}] {ms*1e3:8.1f} us {tf:7.1f} TF ({tf/2500*100:4.1f}%)'
        print(line, flush=True)


def run_conv16(iters):
    """the 16-bit mode's decoder convolutions at the d16 / B=64 shapes, both pixel tiles (CONV16_TILES=2,4; 0 = automatic)"""
    dev = 'cuda'
    tiles = [int(t) for t in os.environ.get('CONV16_TILES', '2,4,8').split(',')]
    for (B, H, W, Cin, Cout, up2, res) in [(64, 256, 256, 160, 160, 0, 1), (64, 256, 256, 160, 160, 1, 0), (64, 128, 128, 320, 160, 0, 0), (64, 128, 128, 160, 160, 0, 1),
                                           (64, 128, 128, 320, 320, 1, 0), (64, 64, 64, 320, 320, 0, 1), (64, 32, 32, 640, 320, 0, 0), (64, 32, 32, 640, 640, 1, 0), (64, 16, 16, 640, 640, 0, 1)]:
        Hi, Wi = (H // 2, W // 2) if up2 else (H, W)
        x = torch.randn(B, Hi, Wi, Cin, device=dev).half(); b = torch.randn(Cout, device=dev)
        w = (torch.randn(4, Cout, 2, 2, Cin, device=dev) * 0.02).half() if up2 else (torch.randn(Cout, 3, 3, Cin, device=dev) * 0.02).half()
        r = torch.randn(B, H, W, Cout, device=dev).half() if res else None
        out = torch.empty(B, H, W, Cout, device=dev, dtype=torch.float16)
        nblk = hip.conv_gn_blocks(H, W, Cout, phase=bool(up2))
        part = torch.zeros(B, max(nblk, 1), Cout, 2, dtype=torch.float64, device=dev)
        if os.environ.get('CONV16_NOGN'): part = None
        if os.environ.get('CONV16_NORES'): r = None
        if up2: fn = lambda: hip.call('upconv_phase_f16', x, w, b, out, part, B, H, W, Cin, Cout)
        else: fn = lambda: hip.call('conv3x3_nhwc_f16', x, w, b, r, out, part, B, H, W, Cin, Cout, 0)
        line = f'conv16 {Cin:3d}->{Cout:3d} {H:3d}x{W:3d} up{up2} res{res}:'
        for t in tiles:
            hip.lib().so.varhip_conv16_force_tile(t)
            ms = timeit(fn, max(iters // 2, 2)); tf = 2.0 * B * H * W * Cout * (4 if up2 else 9) * Cin / ms / 1e9
            line += f'  [wm {t}] {ms:8.3f} ms {tf:7.1f} TF ({tf/2500*100:4.1f}%)'
        hip.lib().so.varhip_conv16_force_tile(0)
        if not up2 and hip.conv16_gn_fusable(B, H, W, Cin, Cout):      # GroupNorm + SiLU inside the conv against apply pass + conv
            gamma, beta = torch.randn(Cin, device=dev) * 0.2 + 1.0, torch.randn(Cin, device=dev) * 0.2
            stats = torch.randn(B, 32, 2, device=dev).abs() + 0.5
            xn = torch.empty_like(x)
            t_apply = timeit(lambda: hip.call('gn_apply_f16', x, stats, gamma, beta, xn, B, H * W, Cin, 32, 1), max(iters // 2, 2))
            t_conv = timeit(fn, max(iters // 2, 2))
            table = torch.randn(B, 2, Cin, device=dev)
            t_fused = timeit(lambda: hip.call('gnconv3x3_nhwc_f16', x, table, 1, w, b, r, out, part, B, H, W, Cin, Cout), max(iters // 2, 2))
            line += f'   | GroupNorm apply {t_apply:6.3f} + conv {t_conv:6.3f} = {t_apply + t_conv:6.3f} ms, fused {t_fused:6.3f} ms'
        print(line, flush=True)
        del x, w, out, r, part


def run_attn16(iters):
    dev = 'cuda'
    B2, H, Lmax = 128, 16, 680
    kc = torch.randn(B2, H, Lmax, 64, device=dev).half(); vc = torch.randn(B2, H, Lmax, 64, device=dev).half()
    cur = 0
    for pn in (1, 2, 3, 4, 5, 6, 8, 10, 13, 16):
        l = pn * pn; cur += l
        q = torch.randn(B2 * l, H * 64, device=dev).half(); out = torch.empty_like(q)
        fn = lambda: hip.call('attn_cached_f16', q, kc, vc, out, B2, l, H, cur, Lmax)
        ms = timeit(fn, iters); tf = 4.0 * B2 * H * l * cur * 64 / ms / 1e9
        print(f'attn16 l={l:3d} curL={cur:3d}: {ms*1e3:9.1f} us  {tf:6.1f} TF (algorithmic)', flush=True)


def run_sampler(iters):
    """CFG + top-k 900 + top-p 0.96 + multinomial (varhip_cfg_sample_f32) at the d16 / B=64 shapes; bytes = two logits rows + one noise row per token"""
    dev, B, V = 'cuda', 64, 4096
    tot_ms, tot_b = 0.0, 0.0
    for pn in (1, 2, 3, 4, 5, 6, 8, 10, 13, 16):
        l = pn * pn
        logits = torch.randn(2 * B * l, V, device=dev) * 2.0; noise = torch.empty(B * l, V, device=dev).exponential_(1)
        idx = torch.empty(B * l, dtype=torch.int64, device=dev)
        for walk in (0, 1):
            hip.lib().so.varhip_sampler_force_walk(walk)
            fn = lambda: hip.call('cfg_sample_f32', logits, noise, idx, None, B, l, V, 0.75, 900, 0.96)
            ms = timeit(fn, iters)
            if walk == 0: tot_ms += ms; tot_b += 12.0 * B * l * V
            print(f'sampler l={l:3d} rows={B * l:6d} {"walk    " if walk else "parallel"}: {ms * 1e3:9.1f} us  {12.0 * B * l * V / ms / 1e6:8.1f} GB/s', flush=True)
        hip.lib().so.varhip_sampler_force_walk(0)
    print(f'sampler total (parallel cut) {tot_ms:.3f} ms per step, {tot_b / tot_ms / 1e6:.1f} GB/s of 8000', flush=True)


def run_tail(iters):
    """the decoder's tail (norm_out -> swish -> conv_out -> clamp) at the d16 / B=64 shape: the one-pass kernels against the two launches they replace"""
    B, H, W, Cin, Cout, omode = 64, 256, 256, 160, 3, 1
    gamma, beta = torch.randn(Cin, device='cuda') * 0.2 + 1.0, torch.randn(Cin, device='cuda') * 0.2
    bias = torch.randn(Cout, device='cuda') * 0.1
    stats = torch.empty(B, 32, 2, dtype=torch.float32, device='cuda')
    out = torch.empty(B, Cout, H, W, device='cuda')
    for flav, dt in (('f32', torch.float32), ('f16', torch.float16), ('bf16', torch.bfloat16)):
        x = (torch.randn(B, H, W, Cin, device='cuda') * 1.3 + 0.2).to(dt)
        w = (torch.randn(Cout, 3, 3, Cin, device='cuda') * 0.05).to(dt)
        scratch = torch.empty(hip.gn_scratch_elems(B, H * W, Cin, 32), dtype=torch.float64, device='cuda')
        hip.call('gn_stats_' + flav, x, stats, scratch, B, H * W, Cin, 32, 1e-6)
        xn = torch.empty_like(x)
        fused = timeit(lambda: hip.call('gn_silu_conv_out_' + flav, x, stats, gamma, beta, w, bias, out, B, H, W, Cin, Cout, 32, omode), iters)
        apply = timeit(lambda: hip.call('gn_apply_' + flav, x, stats, gamma, beta, xn, B, H * W, Cin, 32, 1), iters)
        if flav == 'f32': conv = timeit(lambda: hip.call('conv3x3_nhwc_f32', xn, w, bias, None, out, B, H, W, Cin, Cout, 0, omode), iters)
        else: conv = timeit(lambda: hip.call('conv3x3_nhwc_' + flav, xn, w, bias, None, out, None, B, H, W, Cin, Cout, omode), iters)
        gb = x.numel() * x.element_size() / 1e9
        print(f'tail {flav:4s}: one pass {fused * 1e3:8.1f} us ({gb / (fused * 1e-3):7.1f} GB/s of input)   GroupNorm apply {apply * 1e3:8.1f} us + conv_out {conv * 1e3:8.1f} us', flush=True)
        del x, xn, w, scratch


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('what', nargs='?', default='all')
    ap.add_argument('--iters', type=int, default=10)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    print(hip.lib().version())
    if a.what in ('gemm', 'all'): run_gemm(a.iters)
    if a.what in ('qkv', 'all'): run_qkv(a.iters)
    if a.what in ('conv', 'all'): run_conv(a.iters)
    if a.what in ('attn', 'all'): run_attn(a.iters)
    if a.what == 'conv16': run_conv16(a.iters)
    if a.what in ('gemm16', 'all16'): run_gemm16(a.iters)
    if a.what in ('attn16', 'all16'): run_attn16(a.iters)
    if a.what in ('sampler', 'all'): run_sampler(a.iters)
    if a.what == 'tail': run_tail(a.iters)
