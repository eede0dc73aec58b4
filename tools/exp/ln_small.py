import sys, torch
sys.path.insert(0, '.')
from var_amd import hip
C = 1024
for M, rpg in ((128, 1), (512, 4), (1152, 9), (2048, 16), (4608, 36), (32768, 256)):
    x = torch.randn(M, C, device='cuda'); sc = torch.randn(128, 6 * C, device='cuda'); out16 = torch.empty(M, C, dtype=torch.float16, device='cuda'); out32 = torch.empty(M, C, device='cuda')
    def t(fn, n=50):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3
    print(f'M={M:6d}: f16out {t(lambda: hip.call("ln_modulate_f16out", x, sc, 6 * C, sc[:, C:], 6 * C, out16, M, C, rpg, 1e-6)):.1f} us   f32 {t(lambda: hip.call("ln_modulate_f32", x, sc, 6 * C, sc[:, C:], 6 * C, out32, M, C, rpg, 1e-6)):.1f} us')
