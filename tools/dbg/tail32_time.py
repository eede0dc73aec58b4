import sys, torch
sys.path.insert(0, '.')
from var_amd import hip
B, H, W, Cin, Cout, omode = 64, 256, 256, 160, 3, 1
x = torch.randn(B, H, W, Cin, device='cuda') * 1.3 + 0.2
w = torch.randn(Cout, 3, 3, Cin, device='cuda') * 0.05
bias = torch.randn(Cout, device='cuda') * 0.1
gamma, beta = torch.randn(Cin, device='cuda') * 0.2 + 1.0, torch.randn(Cin, device='cuda') * 0.2
stats = torch.empty(B, 32, 2, dtype=torch.float32, device='cuda')
scratch = torch.empty(hip.gn_scratch_elems(B, H * W, Cin, 32), dtype=torch.float64, device='cuda')
hip.call('gn_stats_f32', x, stats, scratch, B, H * W, Cin, 32, 1e-6)
out = torch.empty(B, Cout, H, W, device='cuda'); xn = torch.empty_like(x)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print('fused   %.1f us' % t(lambda: hip.call('gn_silu_conv_out_f32', x, stats, gamma, beta, w, bias, out, B, H, W, Cin, Cout, 32, omode)))
print('apply   %.1f us' % t(lambda: hip.call('gn_apply_f32', x, stats, gamma, beta, xn, B, H * W, Cin, 32, 1)))
print('conv    %.1f us' % t(lambda: hip.call('conv3x3_nhwc_f32', xn, w, bias, None, out, B, H, W, Cin, Cout, 0, omode)))
