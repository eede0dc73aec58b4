import sys, torch
sys.path.insert(0, '.')
from var_amd import hip
flav, dt = 'f16', torch.float16
for (B, H, W, Cin, Cout, omode) in [(2, 8, 32, 32, 3, 1), (1, 64, 64, 160, 3, 2)]:
    g = torch.Generator().manual_seed(H * 7 + W + Cin)
    x = (torch.randn(B, H, W, Cin, generator=g) * 1.3 + 0.2).to(dt).cuda()
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) * (2.0 / (9 * Cin) ** 0.5)).to(dt).cuda()
    bias = (torch.randn(Cout, generator=g) * 0.1).cuda()
    gamma, beta = (torch.randn(Cin, generator=g) * 0.2 + 1.0).cuda(), (torch.randn(Cin, generator=g) * 0.2).cuda()
    stats = torch.empty(B, 32, 2, dtype=torch.float32, device='cuda')
    scratch = torch.empty(hip.gn_scratch_elems(B, H * W, Cin, 32), dtype=torch.float64, device='cuda')
    hip.call('gn_stats_' + flav, x, stats, scratch, B, H * W, Cin, 32, 1e-6)
    fused = torch.full((B, Cout, H, W), float('nan'), dtype=torch.float32, device='cuda')
    hip.call('gn_silu_conv_out_' + flav, x, stats, gamma, beta, w, bias, fused, B, H, W, Cin, Cout, 32, omode)
    xn = torch.empty_like(x)
    hip.call('gn_apply_' + flav, x, stats, gamma, beta, xn, B, H * W, Cin, 32, 1)
    two = torch.empty_like(fused)
    hip.call('conv3x3_nhwc_' + flav, xn, w, bias, None, two, None, B, H, W, Cin, Cout, omode)
    d = (fused != two).nonzero()
    print((B, H, W, Cin), 'ndiff', d.shape[0], 'max', float((fused - two).abs().max()))
    print(d[:40].tolist())
    ref = torch.nn.functional.conv2d(xn.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), bias.double(), padding=1).clamp(-1, 1)
    if omode == 1: ref = (ref + 1) * 0.5
    print('fused vs f64', float((fused.double() - ref).abs().max()), 'two vs f64', float((two.double() - ref).abs().max()))
