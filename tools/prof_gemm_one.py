#!/usr/bin/env python3
"""One GEMM shape, a few launches — the target of `rocprofv3 --pmc ...` runs (counters per dispatch)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from var_amd import hip
M, N, K = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (8192, 8192, 8192)))
epi = int(sys.argv[4]) if len(sys.argv) > 4 else 0
A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda') * 0.03; b = torch.randn(N, device='cuda')
out = torch.empty(M, N, device='cuda'); resid = torch.randn(M, N, device='cuda'); gamma = torch.randn(128, N, device='cuda')
for _ in range(3):
    hip.call('gemm_nt_f32', A, K, W, K, b, out, N, M, N, K, epi, resid if epi == 2 else None, N, gamma if epi == 2 else None, N, max(M // 128, 1), 0, 1, 0, 0, 0)
torch.cuda.synchronize()
print('done')
