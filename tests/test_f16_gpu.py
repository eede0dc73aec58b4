"""The 16-bit throughput mode (include/var_hip.h "f16": fp16 GEMM operands / KV cache, fp32 accumulation on the f16 MFMAs) on a real MI355X.

This mode is NOT under the fp32 bit-exactness contract: the MFMA-internal reduction over k is not a k-ascending fma chain.  It is tested
  (a) kernel by kernel against float64 arithmetic on the same fp16 inputs (tolerances: fp32 accumulation noise, one fp16 rounding of outputs);
  (b) end to end against its CPU twin (oracle.var_oracle.OracleVAR(f16=True): the fp32 restatement with the same rounding points), teacher-forced
      with the reference's tokens so that a flipped token cannot cascade;
  (c) against the reference's own fp32 run (golden fixtures): logits within a stated tolerance, token agreement reported as a rate.
"""
import contextlib
import io
import json

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


def _hip():
    from var_amd import hip
    return hip


@pytest.mark.parametrize('M,N,K', [(128, 1024, 1024), (1152, 4096, 1024), (200, 1024, 4096), (50, 256, 64), (4096, 3072, 1024), (33, 132, 128)])
@pytest.mark.parametrize('mode', ['none32', 'none16', 'gelu16', 'resid32', 'resid16in'])
def test_gemm16_against_float64(M, N, K, mode):
    hip = _hip()
    g = torch.Generator().manual_seed(M * 7 + N + K)
    A = (torch.randn(M, K, generator=g) * 0.7).half(); W = (torch.randn(N, K, generator=g) * (1.5 / K ** 0.5)).half()
    bias = torch.randn(N, generator=g) * 0.2
    rpg = 50 if M % 50 == 0 else M
    gamma = torch.randn((M + rpg - 1) // rpg, N, generator=g) * 0.5
    resid32 = torch.randn(M, N, generator=g)
    ref = A.double() @ W.double().T + bias.double()
    mag = (A.double().abs() @ W.double().abs().T)            # sum |a||w|: the scale of the fp32 accumulation noise
    out16 = mode in ('none16', 'gelu16')
    epi, resid, rf16, gm = 0, None, 0, None
    if mode == 'gelu16':
        epi = 1; ref = torch.nn.functional.gelu(ref, approximate='tanh')
    elif mode == 'resid32':
        epi = 2; resid = resid32.cuda(); gm = gamma.cuda()
        ref = resid32.double() + ref * gamma.double().repeat_interleave(rpg, dim=0)[:M]
    elif mode == 'resid16in':
        epi = 2; resid = resid32.half().cuda(); rf16 = 1
        ref = resid32.half().double() + ref
    out = torch.empty(M, N, dtype=torch.float16 if out16 else torch.float32, device='cuda')
    hip.call('gemm_nt_f16', A.cuda(), K, W.cuda(), K, bias.cuda(), out, N, int(out16), M, N, K, epi, resid, N, rf16, gm, N, rpg, 1, 0, 0, 0)
    got = out.double().cpu()
    tol = 2e-6 * mag + 1e-6 + (ref.abs() * 2.0 ** -10 if out16 else 0)      # fp32 accumulation (+ one fp16 rounding of the result)
    if mode == 'resid32': tol = tol * gamma.double().repeat_interleave(rpg, dim=0)[:M].abs().clamp_min(1.0) + 1e-6 * ref.abs()
    bad = (got - ref).abs() > tol
    assert not bool(bad.any()), f'{mode} {M}x{N}x{K}: {int(bad.sum())} outside tolerance, max err {float((got - ref).abs().max()):.3e}'


@pytest.mark.parametrize('tile', [0, 1, 2, 3])
@pytest.mark.parametrize('M,N,K,mode', [(700, 1024, 256, 'none16'), (513, 520, 128, 'resid32'), (256, 256, 64, 'gelu16'), (1000, 3072, 192, 'none32')])
def test_gemm16_every_tile_on_ragged_shapes(tile, M, N, K, mode):
    """the three tile instantiations (128x128, 64x64, 256x256 with 8 waves) forced onto shapes with partial tiles in both dimensions:
    identical results (same MFMA instruction, same k order), and right against float64"""
    hip = _hip()
    g = torch.Generator().manual_seed(M + N)
    A = (torch.randn(M, K, generator=g) * 0.7).half().cuda(); W = (torch.randn(N, K, generator=g) * (1.5 / K ** 0.5)).half().cuda()
    bias = (torch.randn(N, generator=g) * 0.2).cuda(); resid = torch.randn(M, N, generator=g).cuda()
    out16 = mode in ('none16', 'gelu16'); epi = {'none16': 0, 'none32': 0, 'gelu16': 1, 'resid32': 2}[mode]
    def run(t):
        out = torch.empty(M, N, dtype=torch.float16 if out16 else torch.float32, device='cuda')
        hip.lib().so.varhip_gemm16_force_tile(t)
        try:
            hip.call('gemm_nt_f16', A, K, W, K, bias, out, N, int(out16), M, N, K, epi, resid if epi == 2 else None, N, 0, None, 0, 1, 1, 0, 0, 0)
        finally:
            hip.lib().so.varhip_gemm16_force_tile(-1)
        return out
    got, base = run(tile), run(1)
    assert torch.equal(got, base), f'tile {tile} differs from the 64x64 tile in {int((got != base).sum())} elements'
    ref = A.double() @ W.double().T + bias.double()
    if epi == 1: ref = torch.nn.functional.gelu(ref, approximate='tanh')
    if epi == 2: ref = resid.double() + ref
    tol = 2e-6 * (A.double().abs() @ W.double().abs().T) + 1e-6 + (ref.abs() * 2.0 ** -10 if out16 else 0) + (1e-6 * ref.abs() if epi == 2 else 0)
    assert bool(((got.double() - ref).abs() <= tol).all())


@pytest.mark.parametrize('tile', [0, 1, 2, 3])
@pytest.mark.parametrize('N,K,mode', [(5760, 1920, 'none16'), (1920, 1920, 'resid32'), (7680, 1920, 'gelu16'), (1920, 7680, 'resid32'),
                                      (6912, 2304, 'none16'), (2304, 2304, 'resid32'), (9216, 2304, 'gelu16'), (2304, 9216, 'resid32'), (4096, 2304, 'none32')])
def test_gemm16_at_d30_d36_widths_every_tile(tile, N, K, mode):
    """the GEMM shapes of BASELINE.json configs[3] / configs[4] (VAR-d30: C = 1920, hidden 7680; VAR-d36: C = 2304, hidden 9216; head V = 4096):
    N and K that are multiples of 64 but not of 256 / 1024, partial tiles in N for the 256- and 128-wide tiles, a ragged M; every tile
    instantiation forced, against float64, and all three identical bit for bit"""
    hip = _hip()
    M = 300
    g = torch.Generator().manual_seed(N + K)
    A = (torch.randn(M, K, generator=g) * 0.7).half().cuda(); W = (torch.randn(N, K, generator=g) * (1.5 / K ** 0.5)).half().cuda()
    bias = (torch.randn(N, generator=g) * 0.2).cuda(); resid = torch.randn(M, N, generator=g).cuda(); gamma = (torch.randn(3, N, generator=g) * 0.5).cuda()
    out16 = mode in ('none16', 'gelu16'); epi = {'none16': 0, 'none32': 0, 'gelu16': 1, 'resid32': 2}[mode]
    def run(t):
        out = torch.empty(M, N, dtype=torch.float16 if out16 else torch.float32, device='cuda')
        hip.lib().so.varhip_gemm16_force_tile(t)
        try:
            hip.call('gemm_nt_f16', A, K, W, K, bias, out, N, int(out16), M, N, K, epi, resid if epi == 2 else None, N, 0, gamma if epi == 2 else None, N, 100, 1, 0, 0, 0)
        finally:
            hip.lib().so.varhip_gemm16_force_tile(-1)
        return out
    got = run(tile)
    if tile != 1:
        base = run(1)
        assert torch.equal(got, base), f'tile {tile} differs from the 64x64 tile in {int((got != base).sum())} elements'
    Ad, Wd = A.double().cpu(), W.double().cpu()
    ref = Ad @ Wd.T + bias.double().cpu()
    mag = Ad.abs() @ Wd.abs().T
    if epi == 1: ref = torch.nn.functional.gelu(ref, approximate='tanh')
    tol = 2e-6 * mag + 1e-6 + (ref.abs() * 2.0 ** -10 if out16 else 0)
    if epi == 2:
        gm = gamma.double().cpu().repeat_interleave(100, dim=0)[:M]
        ref = resid.double().cpu() + ref * gm
        tol = tol * gm.abs().clamp_min(1.0) + 1e-6 * ref.abs()
    err = (got.double().cpu() - ref).abs()
    assert bool((err <= tol).all()), f'{mode} {M}x{N}x{K} tile {tile}: {int((err > tol).sum())} outside tolerance, max err {float(err.max()):.3e}'


@pytest.mark.parametrize('M,N,K,mode', [(256, 256, 64, 'none16'), (2048, 768, 192, 'gelu16'), (8192, 3072, 1024, 'none16'), (4096, 1024, 1024, 'resid32'),
                                        (12800, 1024, 4096, 'resid32'), (33024, 512, 64, 'none32'), (4352, 4096, 128, 'gelu16')])
def test_gemm16_persistent_kernel_equals_one_tile_kernel(M, N, K, mode):
    """whole 256x256 tiles on the persistent kernel (k_gemm16p: one workgroup per CU walks a tile list, the next tile's first K tile requested
    during the last K step, stores left in flight) against the same tiles on k_gemm16<8,4,2,4>: identical bits; 1 to 6 tiles per workgroup,
    K of one tile step (the prefetch is issued in the only step) to 64, every epilogue; and right against float64 on a sample of rows"""
    hip = _hip()
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * 0.7).half().cuda(); W = (torch.randn(N, K, generator=g) * (1.5 / K ** 0.5)).half().cuda()
    bias = (torch.randn(N, generator=g) * 0.2).cuda(); resid = torch.randn(M, N, generator=g).cuda(); gamma = (torch.randn((M + 99) // 100, N, generator=g) * 0.5).cuda()
    out16 = mode in ('none16', 'gelu16'); epi = {'none16': 0, 'none32': 0, 'gelu16': 1, 'resid32': 2}[mode]
    def run(persist, tile):
        out = torch.empty(M, N, dtype=torch.float16 if out16 else torch.float32, device='cuda')
        hip.lib().so.varhip_gemm16_persistent(persist); hip.lib().so.varhip_gemm16_force_tile(tile)
        try: hip.call('gemm_nt_f16', A, K, W, K, bias, out, N, int(out16), M, N, K, epi, resid if epi == 2 else None, N, 0, gamma if epi == 2 else None, N, 100, 1, 0, 0, 0)
        finally: hip.lib().so.varhip_gemm16_persistent(1); hip.lib().so.varhip_gemm16_force_tile(-1)
        return out
    a, b, c = run(1, 2), run(0, 2), run(1, 1)
    assert torch.equal(a, b), f'persistent vs one-tile kernel: {int((a != b).sum())} elements differ'
    assert torch.equal(a, c), f'persistent 256x256 vs 64x64 tiles: {int((a != c).sum())} elements differ'
    assert torch.equal(run(1, 2), a)                                      # run to run
    rows = torch.randperm(M, generator=g)[:64]
    ref = A[rows].double().cpu() @ W.double().cpu().T + bias.double().cpu()
    if epi == 1: ref = torch.nn.functional.gelu(ref, approximate='tanh')
    if epi == 2: ref = resid[rows].double().cpu() + ref * gamma.double().cpu()[(rows // 100)]
    err = (a[rows].double().cpu() - ref).abs()
    assert float(err.max()) <= 2e-3 + (float(ref.abs().max()) * 2.0 ** -10 if out16 else 0.0), f'max err {float(err.max()):.3e}'


def test_gemm16_row_split_launches_are_invisible():
    """a GEMM whose last round of 256x256 tiles would be under 70 % full goes out as two launches over row ranges (the rows that fill whole
    rounds on the 256x256 kernel, the rest on a small-tile kernel): same bits as a single-kernel run, right against float64; the row offset
    reaches gamma's row groups (rows_per_group = 300 straddles the cut) and the q/k/v epilogue's (image, position) decode (l = 100)"""
    hip = _hip()
    g = torch.Generator().manual_seed(11)
    M, N, K = 103936, 512, 64                      # 406 x 2 = 812 tiles: three full rounds (98304 rows) + 5632 rows
    A = (torch.randn(M, K, generator=g) * 0.7).half().cuda(); W = (torch.randn(N, K, generator=g) * 0.2).half().cuda()
    bias = (torch.randn(N, generator=g) * 0.2).cuda(); resid = torch.randn(M, N, generator=g).cuda(); gamma = (torch.randn(347, N, generator=g) * 0.5).cuda()
    def run(t, epi, out16):
        out = torch.empty(M, N, dtype=torch.float16 if out16 else torch.float32, device='cuda')
        hip.lib().so.varhip_gemm16_force_tile(t)
        try: hip.call('gemm_nt_f16', A, K, W, K, bias, out, N, int(out16), M, N, K, epi, resid if epi == 2 else None, N, 0, gamma if epi == 2 else None, N, 300, 1, 0, 0, 0)
        finally: hip.lib().so.varhip_gemm16_force_tile(-1)
        return out
    hip.timing_reset(); hip.timing_enable(True)
    auto_res = run(-1, 2, False)
    hip.timing_enable(False)
    t = hip.timing_read()
    assert t['gemm16']['launches'] == 1 and t['gemm16_small']['launches'] == 1, 'expected one 256x256 launch + one small-tile launch'
    assert t['gemm16']['flops'] == 2.0 * 98304 * N * K and t['gemm16_small']['flops'] == 2.0 * 5632 * N * K
    assert torch.equal(auto_res, run(1, 2, False)) and torch.equal(run(-1, 1, True), run(1, 1, True))
    ref = resid.double().cpu() + (A.double().cpu() @ W.double().cpu().T + bias.double().cpu()) * gamma.double().cpu().repeat_interleave(300, dim=0)[:M]
    assert float((auto_res.double().cpu() - ref).abs().max()) <= 1e-4
    # q/k/v: H = 4 heads (N = 768: 3 column tiles), l = 100, B2 = 700 -> 274 x 3 = 822 tiles, the cut at row 65536 falls inside image 655
    B2, l, H, pos0, Lmax = 700, 100, 4, 7, 120
    C = H * 64; M2 = B2 * l
    A2 = torch.randn(M2, C, generator=g).half().cuda(); W2 = (torch.randn(3 * C, C, generator=g) * (1.0 / C ** 0.5)).half().cuda()
    b2 = (torch.randn(3 * C, generator=g) * 0.1).cuda(); smul = (torch.randn(H, generator=g) * 0.3 + 1.4).cuda()
    outs = []
    for tile in (-1, 1):
        q = torch.empty(M2, C, dtype=torch.float16, device='cuda'); kc = torch.zeros(B2, H, Lmax, 64, dtype=torch.float16, device='cuda'); vc = torch.zeros_like(kc)
        hip.lib().so.varhip_gemm16_force_tile(tile)
        try: hip.call('gemm_qkv_f16', A2, C, W2, C, b2, M2, C, C, smul, 0.125, 1, q, kc, vc, B2, l, H, pos0, Lmax)
        finally: hip.lib().so.varhip_gemm16_force_tile(-1)
        outs.append((q, kc, vc))
    for a, b in zip(*outs): assert torch.equal(a, b)
    assert float(outs[0][1][:, :, :pos0].abs().max()) == 0 and float(outs[0][1][:, :, pos0 + l:].abs().max()) == 0 and float(outs[0][1][655].abs().max()) > 0 and float(outs[0][1][699].abs().max()) > 0


@pytest.mark.parametrize('B2,l,H,pos0,l2', [(4, 9, 2, 5, 1), (2, 64, 4, 91, 1), (3, 25, 16, 0, 0), (4, 100, 4, 10, 1), (6, 50, 8, 3, 0),
                                            (2, 81, 30, 30, 1), (2, 169, 36, 55, 1)])       # the head counts / widths of VAR-d30 and VAR-d36
def test_gemm_qkv16_against_float64(B2, l, H, pos0, l2):
    hip = _hip()
    C, K, Lmax = H * 64, H * 64, max(160, pos0 + l + 8)
    M = B2 * l
    g = torch.Generator().manual_seed(B2 * 100 + l)
    A = (torch.randn(M, K, generator=g)).half(); W = (torch.randn(3 * C, K, generator=g) * (1.0 / K ** 0.5)).half()
    bias = torch.randn(3 * C, generator=g) * 0.1
    smul = torch.randn(H, generator=g) * 0.3 + 1.4
    q = torch.empty(M, C, dtype=torch.float16, device='cuda')
    kc = torch.zeros(B2, H, Lmax, 64, dtype=torch.float16, device='cuda'); vc = torch.zeros_like(kc)
    hip.call('gemm_qkv_f16', A.cuda(), K, W.cuda(), K, bias.cuda(), M, C, K, smul.cuda(), 0.125, l2, q, kc, vc, B2, l, H, pos0, Lmax)
    q2 = torch.empty_like(q); kc2 = torch.zeros_like(kc); vc2 = torch.zeros_like(vc)       # the 256x256 tile forced: same bits
    hip.lib().so.varhip_gemm16_force_tile(2)
    try:
        hip.call('gemm_qkv_f16', A.cuda(), K, W.cuda(), K, bias.cuda(), M, C, K, smul.cuda(), 0.125, l2, q2, kc2, vc2, B2, l, H, pos0, Lmax)
    finally:
        hip.lib().so.varhip_gemm16_force_tile(-1)
    q3 = torch.empty_like(q); kc3 = torch.zeros_like(kc); vc3 = torch.zeros_like(vc)       # ... and the 192x256 tile
    hip.lib().so.varhip_gemm16_force_tile(3)
    try:
        hip.call('gemm_qkv_f16', A.cuda(), K, W.cuda(), K, bias.cuda(), M, C, K, smul.cuda(), 0.125, l2, q3, kc3, vc3, B2, l, H, pos0, Lmax)
    finally:
        hip.lib().so.varhip_gemm16_force_tile(-1)
    assert torch.equal(vc, vc3) and torch.equal(q, q3) and torch.equal(kc, kc3)
    assert torch.equal(vc, vc2) and torch.equal(q, q2) and torch.equal(kc, kc2)      # every tile takes the head's sum of squares in the same order: identical bits
    ref = (A.double() @ W.double().T + bias.double()).view(B2, l, 3, H, 64)
    rq, rk, rv = ref[:, :, 0], ref[:, :, 1], ref[:, :, 2]
    if l2:
        rq = rq / rq.norm(dim=-1, keepdim=True).clamp_min(1e-12) * smul.double().clamp_max(np.log(100)).exp().view(1, 1, H, 1)
        rk = rk / rk.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    else:
        rq = rq * 0.125
    def close(a, b, name):
        err = (a.double().cpu() - b).abs(); tol = b.abs() * 2.0 ** -10 + 2e-4
        assert bool((err <= tol).all()), f'{name}: max err {float(err.max()):.3e}'
    for qq, kk, vv, tag in ((q, kc, vc, 'auto tile'), (q2, kc2, vc2, '256x256 tile')):
        close(qq.view(B2, l, H, 64), rq, f'q ({tag})')
        close(kk[:, :, pos0:pos0 + l].permute(0, 2, 1, 3), rk, f'k cache rows ({tag})')
        close(vv[:, :, pos0:pos0 + l].permute(0, 2, 1, 3), rv, f'v cache rows ({tag})')
    assert float(kc[:, :, :pos0].abs().max() if pos0 else 0) == 0 and float(kc[:, :, pos0 + l:].abs().max()) == 0


@pytest.mark.parametrize('B2,l,H,curL', [(2, 1, 2, 1), (3, 9, 2, 14), (2, 36, 3, 91), (2, 169, 2, 424), (1, 256, 2, 680), (2, 40, 1, 33),
                                         # BASELINE.json configs[4] (VAR-d36 512x512, fp16): its three largest scales, one head and all 36
                                         (2, 324, 1, 536), (2, 576, 1, 1112), (2, 1024, 1, 2240), (1, 324, 36, 536), (1, 576, 36, 1112), (1, 1024, 36, 2240)])
def test_attn16_against_twin(B2, l, H, curL):
    """fp16 attention vs its CPU twin (oracle: fp32 chains, p rounded to fp16 for p.v) on the same fp16 q / K / V; peaked scores included"""
    hip = _hip()
    util.ensure_oracle_built()
    from oracle.var_oracle import lib, _p
    Lmax = curL + 7
    g = torch.Generator().manual_seed(l * 1000 + curL)
    q = torch.randn(B2 * l, H * 64, generator=g)
    q = (q.view(B2 * l, H, 64) / q.view(B2 * l, H, 64).norm(dim=-1, keepdim=True) * 6.0).view(B2 * l, H * 64).half()     # |q| = scale_mul-like
    k = torch.randn(B2, H, Lmax, 64, generator=g); k = (k / k.norm(dim=-1, keepdim=True)).half()
    v = torch.randn(B2, H, Lmax, 64, generator=g).half()
    out = torch.empty(B2 * l, H * 64, dtype=torch.float16, device='cuda')
    hip.call('attn_cached_f16', q.cuda(), k.cuda(), v.cuda(), out, B2, l, H, curL, Lmax)
    want = np.empty((B2 * l, H * 64), np.float32)
    assert lib()['attn_cached_p16_f32'](_p(q.float().numpy()), _p(k.float().numpy()), _p(v.float().numpy()), _p(want), B2, l, H, curL, Lmax) == 0
    got = out.float().cpu().numpy()
    err = np.abs(got - want)
    # p is rounded to fp16 at slightly different values (hardware exp2 vs vm_exp: ~1e-6 relative), the output is one fp16 rounding of
    # nearly equal fp32 values: two fp16 ulps of the output magnitude
    tol = np.abs(want) * 2.0 ** -9 + 2e-3
    assert (err <= tol).all(), f'max err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}'
    # and against exact softmax attention in float64 (head by head): the fp16 p rounding is the dominant deviation
    for h in range(H):
        s = torch.einsum('btc,bjc->btj', q.view(B2, l, H, 64)[:, :, h].double(), k[:, h, :curL].double())
        ref = torch.einsum('btj,bjc->btc', s.softmax(-1), v[:, h, :curL].double()).reshape(B2 * l, 64).numpy()
        assert np.abs(got[:, h * 64:(h + 1) * 64] - ref).max() <= 1e-2


@pytest.mark.parametrize('B,H,W,Cin,Cout,res,omode', [(2, 16, 16, 32, 32, 0, 0), (2, 16, 16, 640, 640, 1, 0), (1, 32, 32, 320, 160, 0, 0), (3, 8, 8, 160, 160, 1, 0),
                                                      (2, 32, 32, 160, 3, 0, 1), (1, 16, 16, 64, 3, 0, 2), (1, 24, 40, 96, 64, 1, 0), (1, 24, 16, 64, 128, 1, 0),
                                                      (2, 8, 64, 32, 128, 1, 0), (1, 32, 16, 64, 128, 0, 0), (2, 16, 32, 160, 320, 1, 0), (3, 16, 64, 96, 160, 0, 0)])
@pytest.mark.parametrize('wm', [2, 4, 8])
def test_conv16_against_float64(B, H, W, Cin, Cout, res, omode, wm):
    """the three kernels, forced (the automatic choice takes the large ones only once they fill the chip): 128 pixels x 4 waves, 256 pixels x 8
    waves, and (8) the halo-patch kernel on the shapes it takes — 8x32 or 16x16 patches, 128 | Cout or 160 | Cout — falling back to the first otherwise"""
    hip = _hip()
    hip.lib().so.varhip_conv16_force_tile(wm)
    try:
        _conv16_case(hip, B, H, W, Cin, Cout, res, omode, blocks_2d=(wm == 8))
    finally:
        hip.lib().so.varhip_conv16_force_tile(0)


def _conv16_case(hip, B, H, W, Cin, Cout, res, omode, blocks_2d=False):
    g = torch.Generator().manual_seed(H * 31 + Cin + Cout)
    x = torch.randn(B, H, W, Cin, generator=g).half()
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) * (1.0 / (9 * Cin) ** 0.5)).half()
    bias = torch.randn(Cout, generator=g) * 0.1
    resid = torch.randn(B, H, W, Cout, generator=g).half() if res else None
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), bias.double(), padding=1)     # NCHW
    if res: ref = ref + resid.double().permute(0, 3, 1, 2)
    nblk = hip.conv_gn_blocks(H, W, Cout) if (omode == 0 and Cout % 4 == 0) else 0
    part = torch.zeros(B, nblk, Cout, 2, dtype=torch.float64, device='cuda') if nblk else None
    if omode:
        out = torch.empty(B, Cout, H, W, dtype=torch.float32, device='cuda')
        ref = ref.clamp(-1, 1); ref = (ref + 1) * 0.5 if omode == 1 else ref
    else:
        out = torch.empty(B, H, W, Cout, dtype=torch.float16, device='cuda')
    hip.call('conv3x3_nhwc_f16', x.cuda(), w.cuda(), bias.cuda(), None if resid is None else resid.cuda(), out, part, B, H, W, Cin, Cout, omode)
    got = out.double().cpu() if omode else out.double().cpu().permute(0, 3, 1, 2)
    tol = 1e-5 + (0 if omode else ref.abs() * 2.0 ** -10) + 2e-6 * (9 * Cin) ** 0.5
    err = (got - ref).abs()
    assert bool((err <= tol).all()), f'max err {float(err.max()):.3e}'
    if nblk and blocks_2d:                                       # the halo-patch kernel's blocks are halves of its 2-D patches: any partition of a sample serves the statistics
        o = out.double().cpu().view(B, H * W, Cout)
        assert torch.allclose(part[..., 0].sum(1).cpu(), o.sum(1), rtol=1e-5, atol=1e-3) and torch.allclose(part[..., 1].sum(1).cpu(), (o * o).sum(1), rtol=1e-5, atol=1e-3)
    elif nblk:                                                   # GroupNorm partials: sums of the ROUNDED outputs, per block of 128 pixels
        o = out.double().cpu().view(B, nblk, 128, Cout)
        # fp32 inside a wave (<= 16 of the fp16 values per lane, then 64 / PXI lanes), fp64 across waves, blocks and in the statistics kernel
        assert torch.allclose(part[..., 0].cpu(), o.sum(2), rtol=1e-5, atol=1e-4) and torch.allclose(part[..., 1].cpu(), (o * o).sum(2), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize('B,H,W,Cin,Cout', [(2, 16, 16, 64, 32), (1, 32, 32, 320, 320), (2, 64, 32, 160, 160), (3, 48, 16, 32, 128)])
@pytest.mark.parametrize('wm', [2, 4])
def test_upconv_phase16_against_float64(B, H, W, Cin, Cout, wm):
    """Upsample2x (nearest 2x + conv3x3, basic_vae.py:22-28) in its folded four-phase form on fp16 data, both pixel tiles"""
    hip = _hip()
    hip.lib().so.varhip_conv16_force_tile(wm)
    try:
        _upconv16_case(hip, B, H, W, Cin, Cout)
    finally:
        hip.lib().so.varhip_conv16_force_tile(0)


def _upconv16_case(hip, B, H, W, Cin, Cout):
    g = torch.Generator().manual_seed(H + Cin)
    x = torch.randn(B, H // 2, W // 2, Cin, generator=g).half()
    w = torch.randn(Cout, 3, 3, Cin, generator=g) * (1.0 / (9 * Cin) ** 0.5)
    bias = torch.randn(Cout, generator=g) * 0.1
    wp = torch.empty(4, Cout, 2, 2, Cin, dtype=torch.float32, device='cuda')
    hip.call('upconv_pack_f32', w.cuda(), wp, Cin, Cout)
    wp16 = wp.half()
    nblk = hip.conv_gn_blocks(H, W, Cout, phase=True)
    part = torch.zeros(B, max(nblk, 1), Cout, 2, dtype=torch.float64, device='cuda')
    out = torch.empty(B, H, W, Cout, dtype=torch.float16, device='cuda')
    hip.call('upconv_phase_f16', x.cuda(), wp16, bias.cuda(), out, part if nblk else None, B, H, W, Cin, Cout)
    # reference: the phase form itself in float64 with the fp16-rounded phase weights
    xd = x.double().permute(0, 3, 1, 2)
    ref = torch.empty(B, Cout, H, W, dtype=torch.float64)
    for py in range(2):
        for px in range(2):
            k = wp16[py * 2 + px].double().cpu().permute(0, 3, 1, 2)                 # [Cout][Cin][2][2]
            xp = torch.nn.functional.pad(xd, (1 - px, px, 1 - py, py))                # taps (a, b) read low-res pixel (y + a - 1 + py, x + b - 1 + px)
            ref[:, :, py::2, px::2] = torch.nn.functional.conv2d(xp, k, bias.double())
    got = out.double().cpu().permute(0, 3, 1, 2)
    err = (got - ref).abs()
    assert bool((err <= 1e-5 + ref.abs() * 2.0 ** -10 + 2e-6 * (4 * Cin) ** 0.5).all()), f'max err {float(err.max()):.3e}'
    # and against the definition (nearest 2x then 3x3 conv) with the unrounded weights: fp16 rounding of the folded weights only
    ref2 = torch.nn.functional.conv2d(torch.nn.functional.interpolate(xd, scale_factor=2, mode='nearest'), w.double().permute(0, 3, 1, 2), bias.double(), padding=1)
    assert float((got - ref2).abs().max()) <= 2e-2
    if nblk:
        o = out.double().cpu()
        assert torch.allclose(part[..., 0].sum(1).cpu(), o.sum((1, 2)), rtol=1e-5, atol=1e-3) and torch.allclose(part[..., 1].sum(1).cpu(), (o * o).sum((1, 2)), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize('B,HW,C,silu', [(2, 256, 640, 1), (3, 1024, 160, 1), (1, 100, 32, 0), (2, 4096, 320, 1)])
def test_groupnorm16_against_float64(B, HW, C, silu):
    hip = _hip()
    g = torch.Generator().manual_seed(HW + C)
    x = (torch.randn(B, HW, C, generator=g) * 1.7 + 0.3).half()
    gamma, beta = torch.randn(C, generator=g) * 0.2 + 1.0, torch.randn(C, generator=g) * 0.2
    stats = torch.empty(B, 32, 2, dtype=torch.float32, device='cuda')
    scratch = torch.empty(hip.gn_scratch_elems(B, HW, C, 32), dtype=torch.float64, device='cuda')
    hip.call('gn_stats_f16', x.cuda(), stats, scratch, B, HW, C, 32, 1e-6)
    xd = x.double().view(B, HW, 32, C // 32)
    mean = xd.mean(dim=(1, 3)); var = xd.var(dim=(1, 3), unbiased=False)
    assert torch.allclose(stats[..., 0].double().cpu(), mean, atol=1e-6) and torch.allclose(stats[..., 1].double().cpu(), (var + 1e-6).rsqrt(), rtol=1e-6)
    out = torch.empty(B, HW, C, dtype=torch.float16, device='cuda')
    hip.call('gn_apply_f16', x.cuda(), stats, gamma.cuda(), beta.cuda(), out, B, HW, C, 32, silu)
    ref = torch.nn.functional.group_norm(x.double().permute(0, 2, 1), 32, gamma.double(), beta.double(), eps=1e-6).permute(0, 2, 1)
    if silu: ref = torch.nn.functional.silu(ref)
    err = (out.double().cpu() - ref).abs()
    assert bool((err <= ref.abs() * 2.0 ** -10 + 1e-3).all()), f'max err {float(err.max()):.3e}'
    y32 = torch.empty(B, HW, C, dtype=torch.float32, device='cuda')
    hip.call('cast_f16_to_f32', out, y32, out.numel())
    back = torch.empty_like(out)
    hip.call('cast_f32_to_f16', y32, back, out.numel())
    assert torch.equal(y32, out.float()) and torch.equal(back, out)


@pytest.mark.parametrize('flav', ['f16', 'bf16'])
@pytest.mark.parametrize('B,H,W,Cin,Cout,omode', [(2, 8, 32, 32, 3, 1), (1, 64, 64, 160, 3, 2), (3, 16, 96, 64, 3, 1), (1, 24, 32, 96, 8, 2), (2, 256, 256, 160, 3, 1)])
def test_gn_silu_conv_out_fused_equals_two_launches(flav, B, H, W, Cin, Cout, omode):
    """the decoder's tail (norm_out -> swish -> conv_out -> clamp, basic_vae.py:224-226) in one pass against GroupNorm apply + conv3x3 as two
    launches: identical bits (same arithmetic per element, same MFMA sequence), edge patches and the image border included; and against float64"""
    hip = _hip()
    dt = torch.float16 if flav == 'f16' else torch.bfloat16
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
    assert torch.equal(fused, two), f'fused tail differs from the two launches in {int((fused != two).sum())} elements, max {float((fused - two).abs().max()):.3e}'
    if B * H * W <= 70000:
        ref = torch.nn.functional.conv2d(xn.double().cpu().permute(0, 3, 1, 2), w.double().cpu().permute(0, 3, 1, 2), bias.double().cpu(), padding=1).clamp(-1, 1)
        if omode == 1: ref = (ref + 1) * 0.5
        assert float((fused.double().cpu() - ref).abs().max()) <= 1e-5 + 2e-6 * (9 * Cin) ** 0.5
    # shapes it does not take are refused, not mis-computed
    from var_amd import abi
    assert hip.lib().so.varhip_gn_silu_conv_out_f16 is not None
    with pytest.raises(Exception):
        hip.call('gn_silu_conv_out_' + flav, x, stats, gamma, beta, w, bias, fused, B, H - 1, W, Cin, Cout, 32, omode)


@pytest.mark.parametrize('flav', ['f16', 'bf16'])
@pytest.mark.parametrize('B,H,W,Cin,Cout,res,silu', [(2, 8, 32, 32, 160, 0, 1), (1, 16, 64, 160, 160, 1, 1), (3, 24, 32, 64, 128, 1, 1), (2, 16, 16, 96, 160, 0, 1), (1, 32, 48, 320, 320, 1, 0),
                                                     (2, 32, 32, 640, 640, 1, 1), (1, 64, 64, 320, 160, 0, 1), (64, 32, 32, 160, 160, 1, 1)])
def test_gnconv16_fused_equals_apply_then_conv(flav, B, H, W, Cin, Cout, res, silu):
    """ResnetBlock's conv(swish(norm(x))) (basic_vae.py:57-60) in one launch — GroupNorm + SiLU applied to the halo patch in LDS — against GroupNorm apply +
    conv3x3 as two launches: identical bits in the map and in the GroupNorm partials it leaves (same arithmetic per element, same MFMA sequence); border
    patches, residual, both patch shapes (8 x 32 and 16 x 16), both flavours"""
    hip = _hip()
    dt = torch.float16 if flav == 'f16' else torch.bfloat16
    g = torch.Generator().manual_seed(H * 5 + W + Cin + Cout)
    x = (torch.randn(B, H, W, Cin, generator=g) * 1.3 + 0.2).to(dt).cuda()
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) * (2.0 / (9 * Cin) ** 0.5)).to(dt).cuda()
    bias = (torch.randn(Cout, generator=g) * 0.1).cuda()
    resid = torch.randn(B, H, W, Cout, generator=g).to(dt).cuda() if res else None
    gamma, beta = (torch.randn(Cin, generator=g) * 0.2 + 1.0).cuda(), (torch.randn(Cin, generator=g) * 0.2).cuda()
    stats = torch.empty(B, 32, 2, dtype=torch.float32, device='cuda')
    scratch = torch.empty(hip.gn_scratch_elems(B, H * W, Cin, 32), dtype=torch.float64, device='cuda')
    hip.call('gn_stats_' + flav, x, stats, scratch, B, H * W, Cin, 32, 1e-6)
    nblk = hip.conv_gn_blocks(H, W, Cout)
    hip.lib().so.varhip_conv16_force_tile(8)                    # (small maps: the halo-patch kernel is otherwise chosen from one workgroup per CU on)
    try:
        fusable = hip.conv16_gn_fusable(B, H, W, Cin, Cout)
        assert fusable == (not (Cin == 640 and W % 32 == 0 and H % 8 == 0)), 'the (scale, shift) table of 640 channels does not fit beside 8 x 32 patches'
        xn = torch.empty_like(x)
        hip.call('gn_apply_' + flav, x, stats, gamma, beta, xn, B, H * W, Cin, 32, silu)
        table = torch.empty(B, 2, Cin, dtype=torch.float32, device='cuda')
        hip.call('gn_scale_shift_f32', stats, gamma, beta, table, B, Cin, 32)
        two = torch.empty(B, H, W, Cout, dtype=dt, device='cuda')
        part2 = torch.zeros(B, nblk, Cout, 2, dtype=torch.float64, device='cuda') if nblk else None
        hip.call('conv3x3_nhwc_' + flav, xn, w, bias, resid, two, part2, B, H, W, Cin, Cout, 0)
        if not fusable:
            with pytest.raises(Exception):
                hip.call('gnconv3x3_nhwc_' + flav, x, table, silu, w, bias, resid, two, part2, B, H, W, Cin, Cout)
            return
        fused = torch.full((B, H, W, Cout), float('nan'), dtype=dt, device='cuda')
        part1 = torch.zeros(B, nblk, Cout, 2, dtype=torch.float64, device='cuda') if nblk else None
        hip.call('gnconv3x3_nhwc_' + flav, x, table, silu, w, bias, resid, fused, part1, B, H, W, Cin, Cout)
    finally:
        hip.lib().so.varhip_conv16_force_tile(0)
    assert torch.equal(fused, two), f'fused differs from the two launches in {int((fused != two).sum())} of {fused.numel()} elements, max {float((fused.float() - two.float()).abs().max()):.3e}'
    if nblk: assert torch.equal(part1, part2)
    if B * H * W <= 70000:
        ref = torch.nn.functional.conv2d(xn.double().cpu().permute(0, 3, 1, 2), w.double().cpu().permute(0, 3, 1, 2), bias.double().cpu(), padding=1)
        if res: ref = ref + resid.double().cpu().permute(0, 3, 1, 2)
        err = (fused.double().cpu().permute(0, 3, 1, 2) - ref).abs()
        assert bool((err <= 1e-5 + ref.abs() * 2.0 ** (-10 if flav == 'f16' else -7) + 4e-6 * (9 * Cin) ** 0.5).all()), float(err.max())


def test_decoder16_fused_groupnorm_equals_unfused():
    """the 16-bit decoder with every GroupNorm + SiLU inside the conv that follows it (where the shape allows) against the same decoder with the apply
    passes as launches of their own: the same image bit for bit (d16-size decoder, 256 x 256; B = 64 so that the 128^2 / 256^2 levels take the fused form)"""
    z, meta = util.load_case('d16_full')
    vae, var = _models(meta)
    g = torch.Generator().manual_seed(5)
    f_hat = (torch.randn(64, 32, 16, 16, generator=g) * 1.5).cuda()
    eng = vae._decoder_engine()
    for flav in ('f16', 'bf16'):
        eng.set_precision(flav)
        try:
            with torch.inference_mode():
                a = vae.fhat_to_img(f_hat).clone()
                eng.fuse_gn = False
                b = vae.fhat_to_img(f_hat).clone()
        finally:
            eng.set_precision('f32'); eng.fuse_gn = True
        assert torch.equal(a, b), f'{flav}: max {float((a - b).abs().max()):.3e}'


def test_decoder16_vs_fp32_decoder():
    """VQVAE.fhat_to_img in the 16-bit mode against the fp32 HIP decoder on the same f_hat (d16-size decoder, 256x256, B=2)"""
    z, meta = util.load_case('d16_full')
    vae, var = _models(meta)
    g = torch.Generator().manual_seed(3)
    f_hat = (torch.randn(2, 32, 16, 16, generator=g) * 1.5).cuda()
    with torch.inference_mode():
        a = vae.fhat_to_img(f_hat).clone()
        vae._decoder_engine().set_precision('f16')
        try:
            b = vae.fhat_to_img(f_hat).clone()
            b2 = vae.fhat_to_img(f_hat)
        finally:
            vae._decoder_engine().set_precision('f32')
        c = vae.fhat_to_img(f_hat)
    assert torch.equal(b, b2) and torch.equal(a, c)
    eng = vae._decoder_engine()
    eng.set_precision('f16'); eng.unfused_tail = True           # norm_out / conv_out as two launches: the same image bit for bit
    try:
        with torch.inference_mode(): b3 = vae.fhat_to_img(f_hat).clone()
    finally:
        eng.set_precision('f32'); eng.unfused_tail = False
    assert torch.equal(b, b3)
    d = (a - b).abs()
    print(f'decoder f16 vs f32: max |d| {float(d.max()):.3e}, mean |d| {float(d.mean()):.3e} (range [-1, 1])')
    assert float(d.max()) <= 5e-2 and float(d.mean()) <= 4e-3 and torch.isfinite(b).all()


@pytest.mark.parametrize('kind', ['f32', 'f16'])
def test_attention_kernels_are_run_to_run_deterministic(kind):
    """regression for the round-2 hazard: an inline-asm v_max3 was the first reader of the score MFMAs' result and read it before the last MFMA
    had written it (hipcc pads the MFMA -> VALU hazard only for its own instructions): tile maxima, and with them the rounding of every
    probability, changed from run to run.  Eight back-to-back launches per shape (1 to 4 waves per workgroup, ragged and full) must agree bit for bit."""
    hip = _hip()
    dt = torch.float32 if kind == 'f32' else torch.float16
    B2, H, Lmax = 8, 16, 680
    g = torch.Generator().manual_seed(1)
    kc = torch.randn(B2, H, Lmax, 64, generator=g).to(dt).cuda(); vc = torch.randn(B2, H, Lmax, 64, generator=g).to(dt).cuda()
    cur = 0
    for pn in (1, 2, 3, 4, 5, 6, 8, 10, 13, 16):
        l = pn * pn; cur += l
        q = torch.randn(B2 * l, H * 64, generator=g).to(dt).cuda()
        outs = []
        for _ in range(8):
            out = torch.empty_like(q)
            hip.call('attn_cached_f32' if kind == 'f32' else 'attn_cached_f16', q, kc, vc, out, B2, l, H, cur, Lmax)
            outs.append(out)
        torch.cuda.synchronize()
        assert all(torch.equal(o, outs[0]) for o in outs[1:]), f'{kind} attention l={l} curL={cur}: launches differ'
        assert torch.isfinite(outs[0].float()).all()


_MODELS = {}


def _models(meta):
    key = (meta['depth'], meta['ch'], tuple(meta['patch_nums']), meta['attn_l2_norm'], meta['shared_aln'])
    if key not in _MODELS:
        from models import build_vae_var
        from var_amd.detinit import fill_module_device_
        _MODELS.clear(); torch.cuda.empty_cache()
        with contextlib.redirect_stdout(io.StringIO()):
            vae, var = build_vae_var(device='cuda', patch_nums=tuple(meta['patch_nums']), depth=meta['depth'], ch=meta['ch'],
                                     shared_aln=meta['shared_aln'], attn_l2_norm=meta['attn_l2_norm'])
        fill_module_device_(var, meta['depth'], 0, 'var.'); fill_module_device_(vae, meta['depth'], 0, 'vae.')
        _MODELS[key] = (vae.eval(), var.eval())
    return _MODELS[key]


@pytest.mark.parametrize('name', ['t_pn12345', 't_saln', 't_nol2', 'd16_pn123', 'd30_pn123', 'd36_saln_pn12346'])
def test_f16_mode_vs_twin_and_reference(name):
    """end to end, teacher-forced with the reference's tokens: per-scale logits of the HIP f16 mode vs the CPU twin (same rounding points)
    and vs the reference's fp32 run; own token choices vs the reference's as an agreement rate; image vs the reference's"""
    z, meta = util.load_case(name)
    vae, var = _models(meta)
    pns = meta['patch_nums']
    noise = [torch.from_numpy(n) for n in util.regen_noise(meta, z)]
    labels = torch.tensor(meta['labels'], dtype=torch.int64, device='cuda')
    force = torch.from_numpy(z['idx'].astype(np.int64))
    var.set_hip_precision('f16')
    try:
        eng = var.engine()
        img = eng.sample(len(meta['labels']), labels, None, meta['cfg'], meta['top_k'], meta['top_p'], noises=noise, force_idx=force, trace=True)
        tr = {k: [t.cpu().numpy() for t in v] for k, v in eng.last_trace.items()}
        img = img.cpu().numpy()
        free = eng.sample(len(meta['labels']), labels, None, meta['cfg'], meta['top_k'], meta['top_p'], noises=noise, trace=True)
        idx_free = torch.cat(eng.last_trace['idx'], dim=1).cpu().numpy()
    finally:
        var.set_hip_precision('f32')
    util.ensure_oracle_built()
    from oracle.var_oracle import OracleVAR
    var_sd, vae_sd = util.make_weights(meta)
    twin = OracleVAR(var_sd, vae_sd, pns, meta['depth'], attn_l2_norm=meta['attn_l2_norm'], shared_aln=meta['shared_aln'], f16=True)
    r = twin.run(meta['labels'], util.regen_noise(meta, z), meta['cfg'], meta['top_k'], meta['top_p'], force_idx=z['idx'].astype(np.int64))
    msgs, ok_all = [], True
    for si, pn in enumerate(pns):
        lg = tr['logits'][si]
        scale = float(np.abs(r['logits'][si]).max())
        ok, m = util.diff_report(f'{name} f16 logits s{si} vs twin (|logit| max {scale:.2f})', lg, r['logits'][si], atol=4e-3 * max(scale, 1.0), rtol=0); ok_all &= ok; msgs.append(m)
        want = z[f'logits_s{si}']
        got = lg if meta['full_logits'] else lg[:, sorted({0, pn * pn - 1}), :]
        ok, m = util.diff_report(f'{name} f16 logits s{si} vs fp32 reference', got, want, atol=3e-2 * max(scale, 1.0), rtol=0); ok_all &= ok; msgs.append(m)
    idx = np.concatenate(tr['idx'], axis=1)
    agree_tf = float((idx == z['idx']).mean()); agree_twin = float((idx == r['idx']).mean()); agree_free = float((idx_free == z['idx']).mean())
    msgs.append(f'{name}: token agreement with the fp32 reference, teacher-forced {agree_tf:.3f}, free-running {agree_free:.3f}; with the twin (teacher-forced) {agree_twin:.3f}')
    # pixels: fp16 activations through the ~50 layers of the decoder (statistics / accumulation in fp32): stated tolerance 2e-2 of the [0,1] range
    ok, m = util.diff_report(f'{name} f16 image (teacher-forced tokens) vs reference', img, z['img'], atol=2e-2); ok_all &= ok; msgs.append(m)
    msgs.append(f'{name}: image mean |d| vs reference {float(np.abs(img - z["img"]).mean()):.2e}')
    print('\n'.join(msgs))
    assert ok_all, '\n'.join(msgs)
    assert agree_tf >= 0.9 and agree_twin >= 0.95, msgs[-2]
    assert np.isfinite(free.cpu().numpy()).all()


@pytest.mark.parametrize('name', ['ac16_t_pn12345', 'ac16_t_saln', 'ac16_d16_pn123'])
def test_f16_mode_vs_reference_under_fp16_autocast(name):
    """the 16-bit mode on the GPU against the REFERENCE's own 16-bit execution (fixtures: the reference's autoregressive_infer_cfg under
    torch.autocast(dtype=float16), demo_sample.py:66-68), teacher-forced with the fixture's tokens: logits within 2e-3 x max|logit| (the
    reference's logits carry an fp16 rounding of their own: 4e-3 at |logit| 8..16), token agreement >= 97 %, pixels within 2e-2"""
    z, meta = util.load_case(name)
    assert meta['autocast16'] is True
    vae, var = _models(meta)
    pns = meta['patch_nums']
    noise = [torch.from_numpy(n) for n in util.regen_noise(meta, z)]
    labels = torch.tensor(meta['labels'], dtype=torch.int64, device='cuda')
    var.set_hip_precision('f16')
    try:
        eng = var.engine()
        img = eng.sample(len(meta['labels']), labels, None, meta['cfg'], meta['top_k'], meta['top_p'], noises=noise,
                         force_idx=torch.from_numpy(z['idx'].astype(np.int64)), trace=True).cpu().numpy()
        tr = {k: [t.cpu().numpy() for t in v] for k, v in eng.last_trace.items()}
    finally:
        var.set_hip_precision('f32')
    msgs, ok_all = [], True
    for si, pn in enumerate(pns):
        lg, want = tr['logits'][si], z[f'logits_s{si}']
        got = lg if meta['full_logits'] else lg[:, sorted({0, pn * pn - 1}), :]
        ok, m = util.diff_report(f'{name} f16 logits s{si} vs the reference under fp16 autocast', got, want, atol=2e-3 * max(float(np.abs(want).max()), 1.0)); ok_all &= ok; msgs.append(m)
    agree = float((np.concatenate(tr['idx'], axis=1) == z['idx']).mean())
    msgs.append(f'{name}: token agreement with the reference under fp16 autocast (teacher-forced) {agree:.3f}')
    ok, m = util.diff_report(f'{name} f16 image vs the reference under fp16 autocast', img, z['img'], atol=2e-2); ok_all &= ok; msgs.append(m)
    print('\n'.join(msgs))
    assert ok_all and agree >= 0.97, '\n'.join(msgs)


def test_f16_mode_properties_d16_full():
    """d16, all 10 scales at B=4 in the 16-bit mode: deterministic, batch-slice invariant, finite; the mode switch restores fp32 results bit for bit"""
    z, meta = util.load_case('d16_full')
    vae, var = _models(meta)
    V, B = var.V, 4
    g = torch.Generator().manual_seed(5)
    noise = [torch.empty(B * pn * pn, V).exponential_(1, generator=g) for pn in var.patch_nums]
    labels = torch.tensor([1, 22, 333, 980], device='cuda')
    eng = var.engine()
    base32 = eng.sample(B, labels, None, 1.5, 900, 0.96, noises=noise).clone()
    var.set_hip_precision('f16')
    try:
        a = eng.sample(B, labels, None, 1.5, 900, 0.96, noises=noise, trace=True).clone()
        ia = torch.cat(eng.last_trace['idx'], dim=1)
        b = eng.sample(B, labels, None, 1.5, 900, 0.96, noises=noise, trace=True)
        assert torch.equal(a, b) and torch.equal(ia, torch.cat(eng.last_trace['idx'], dim=1))
        sub = eng.sample(2, labels[1:3], None, 1.5, 900, 0.96, noises=[n.view(B, -1, V)[1:3].reshape(-1, V) for n in noise], trace=True)
        assert torch.equal(torch.cat(eng.last_trace['idx'], dim=1), ia[1:3]), 'tokens must not depend on the batch neighbours'
        # pixels: which conv16 kernel runs depends on the number of tiles, i.e. on the batch, and the kernels sum their GroupNorm partials in
        # different orders (fp32 inside a wave): a statistic moves in its 7th digit, an activation on a rounding boundary moves by one fp16 ulp
        # (5e-4 at magnitude 1) and the layers above carry that on: a few fp16 ulps at the output, far inside the mode's 2e-2 pixel budget
        assert float((sub - a[1:3]).abs().max()) <= 5e-3
        assert torch.isfinite(a).all() and float(a.min()) >= 0 and float(a.max()) <= 1
    finally:
        var.set_hip_precision('f32')
    assert torch.equal(eng.sample(B, labels, None, 1.5, 900, 0.96, noises=noise), base32)


@pytest.mark.parametrize('depth,saln,pns,flav', [(30, False, (1, 2, 3, 4, 5, 6, 8, 10, 13, 16), 'f16'), (36, True, (1, 2, 3, 4, 6, 9, 13, 18, 24, 32), 'f16'),
                                                 (36, True, (1, 2, 3, 4, 6, 9, 13, 18, 24, 32), 'bf16')],
                         ids=['d30_256px_f16', 'd36_512px_f16', 'd36_512px_bf16'])
def test_full_size_wide_models_properties_f16(depth, saln, pns, flav):
    """BASELINE.json configs[3] / configs[4] at FULL size in the precision configs[4] names (fp16): VAR-d30 256x256 (L = 680) and VAR-d36 512x512
    (patch_nums up to 32, L = 2240, KV cache reused across scales, 36 heads), B=2 on one GPU, through the size-independent checks (the
    f32 counterpart is tests/test_e2e_gpu.py::test_full_size_wide_models_properties): determinism, batch-slice invariance of the tokens,
    teacher-forced VAR.forward logits == the AR run's conditional logits bit for bit (same kernels, other row counts and tiles),
    incremental f_hat == embed_to_fhat bit for bit (the quantizer is fp32 in both modes), idxBl_to_img with the fp16 decoder == the AR image
    bit for bit and with the fp32 decoder within the mode's 2e-2 pixel budget; tokens agree with the f32 mode's on most positions."""
    meta = dict(depth=depth, ch=160, patch_nums=list(pns), attn_l2_norm=True, shared_aln=saln)
    vae, var = _models(meta)
    eng = var.engine()
    V, B, L = var.V, 2, var.L
    assert (var.C, var.num_heads, var.depth) == (64 * depth, depth, depth) and L == sum(p * p for p in pns)
    g = torch.Generator().manual_seed(depth)
    noise = [torch.empty(B * pn * pn, V).exponential_(1, generator=g) for pn in pns]
    labels = torch.tensor([207, 980], device='cuda')
    eng.sample(B, labels, None, 1.5, 900, 0.96, noises=noise, trace=True, decode=False)
    idx32 = torch.cat(eng.last_trace['idx'], dim=1)
    var.set_hip_precision(flav)
    try:
        img = eng.sample(B, labels, None, 1.5, 900, 0.96, noises=noise, trace=True).clone()
        tr = eng.last_trace
        idx = torch.cat(tr['idx'], dim=1)
        ar_logits = torch.cat([lg[:B] for lg in tr['logits']], dim=1)
        f_hat = tr['f_hat'][-1].clone()
        P = 16 * pns[-1]
        assert img.shape == (B, 3, P, P) and torch.isfinite(img).all() and float(img.min()) >= 0 and float(img.max()) <= 1
        assert idx.shape == (B, L) and len(torch.unique(idx)) > 64 and torch.isfinite(ar_logits).all()
        img2 = eng.sample(B, labels, None, 1.5, 900, 0.96, noises=noise, trace=True)
        assert torch.equal(img, img2) and torch.equal(idx, torch.cat(eng.last_trace['idx'], dim=1))
        sub = eng.sample(1, labels[1:], None, 1.5, 900, 0.96, noises=[n.view(B, -1, V)[1:].reshape(-1, V) for n in noise], trace=True)
        assert torch.equal(torch.cat(eng.last_trace['idx'], dim=1), idx[1:]), 'tokens must not depend on the batch neighbours'
        assert float((sub - img[1:]).abs().max()) <= (5e-3 if flav == 'f16' else 4e-2)         # (conv16 kernel choice and GroupNorm partial order follow the batch: test_f16_mode_properties_d16_full)
        ms, cur = [], 0
        for pn in pns:
            ms.append(idx[:, cur:cur + pn * pn].contiguous()); cur += pn * pn
        var.cond_drop_rate = 0.0
        with torch.inference_mode():
            tf = var(labels, vae.quantize.idxBl_to_var_input(ms))
        assert tf.shape == (B, L, V) and torch.equal(tf, ar_logits), f'teacher-forced f16 logits differ: max {float((tf - ar_logits).abs().max()):.3e}'
        with torch.inference_mode():
            hs = [vae.quantize.embedding(i).transpose(1, 2).reshape(B, vae.Cvae, pn, pn) for i, pn in zip(ms, pns)]
            assert torch.equal(vae.quantize.embed_to_fhat(hs, all_to_max_scale=True, last_one=True), f_hat)
            im32 = vae.idxBl_to_img(ms, same_shape=True, last_one=True).add_(1).mul_(0.5)           # the VQVAE's own entry points stay fp32 ...
            vae._decoder_engine().set_precision(flav)                                                 # ... unless its owner asks for the 16-bit decoder
            try:
                im16 = vae.idxBl_to_img(ms, same_shape=True, last_one=True).add_(1).mul_(0.5)
            finally:
                vae._decoder_engine().set_precision('f32')
        assert torch.equal(im16, img)
        d = (im32 - img).abs()
        print(f'd{depth} {flav}: decoder {flav} vs f32 on the same tokens max |d| {float(d.max()):.3e} mean {float(d.mean()):.3e}; '
              f'free-running token agreement with the f32 mode {float((idx == idx32).float().mean()):.3f}')
        k = 1.0 if flav == 'f16' else 3.0                                # (bf16: 3 bits fewer per rounding, measured 2.8e-2 / 2.6e-3; tests/test_bf16_gpu.py)
        assert float(d.max()) <= 3e-2 * k and float(d.mean()) <= 2e-3 * k          # [0,1] range: half of test_decoder16_vs_fp32_decoder's [-1,1] bounds
    finally:
        var.set_hip_precision('f32')
