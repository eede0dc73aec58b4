#!/usr/bin/env python3
"""A/B of the GroupNorm-fused 16-bit conv against apply pass + conv at the decoder's shapes (run on the GPU box): interleaved rounds, min / median per arm."""
import os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from var_amd import hip


def t(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    dev = 'cuda'
    for (B, H, W, Cin, Cout, res) in [(64, 256, 256, 160, 160, 1), (64, 128, 128, 320, 160, 0), (64, 128, 128, 160, 160, 1), (64, 64, 64, 320, 320, 1)]:
        x = torch.randn(B, H, W, Cin, device=dev).half(); b = torch.randn(Cout, device=dev)
        w = (torch.randn(Cout, 3, 3, Cin, device=dev) * 0.02).half()
        r = torch.randn(B, H, W, Cout, device=dev).half() if res else None
        out = torch.empty(B, H, W, Cout, device=dev, dtype=torch.float16); xn = torch.empty_like(x)
        nblk = hip.conv_gn_blocks(H, W, Cout)
        part = torch.zeros(B, nblk, Cout, 2, dtype=torch.float64, device=dev)
        gamma, beta = torch.randn(Cin, device=dev) * 0.2 + 1.0, torch.randn(Cin, device=dev) * 0.2
        stats = torch.randn(B, 32, 2, device=dev).abs() + 0.5
        table = torch.randn(B, 2, Cin, device=dev)
        arms = {'conv': lambda: hip.call('conv3x3_nhwc_f16', x, w, b, r, out, part, B, H, W, Cin, Cout, 0),
                'apply': lambda: hip.call('gn_apply_f16', x, stats, gamma, beta, xn, B, H * W, Cin, 32, 1),
                'fused': lambda: hip.call('gnconv3x3_nhwc_f16', x, table, 1, w, b, r, out, part, B, H, W, Cin, Cout)}
        res_ = {k: [] for k in arms}
        for k, fn in arms.items(): fn()
        torch.cuda.synchronize()
        for rnd in range(6):
            for k, fn in arms.items(): res_[k].append(t(fn, 8))
        line = f'{Cin}->{Cout} {H}x{W} res{res}:'
        for k in arms: line += f'  {k} min {min(res_[k]):.3f} med {statistics.median(res_[k]):.3f}'
        line += f'  | apply+conv {min(res_["apply"]) + min(res_["conv"]):.3f} vs fused {min(res_["fused"]):.3f}'
        print(line, flush=True)


if __name__ == '__main__':
    torch.cuda.set_device(0); main()
