import sys, torch
sys.path.insert(0, '.')
from var_amd import hip
B, l, V = 64, 256, 4096
lg = torch.randn(2 * B * l, V, device='cuda') * 3
noise = torch.empty(B * l, V, device='cuda').exponential_(1)
idx = torch.empty(B * l, dtype=torch.int64, device='cuda')
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for tk, tp in ((900, 0.96), (900, 0.0), (0, 0.0), (0, 0.96)):
    print(f'top_k {tk:4d} top_p {tp}: {t(lambda: hip.call("cfg_sample_f32", lg, noise, idx, None, B, l, V, 1.5, tk, tp)):.1f} us')
