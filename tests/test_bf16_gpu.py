"""The bfloat16 flavour of the 16-bit throughput mode (include/var_hip.h "bf16"; `VAR.set_hip_precision('bf16')`) on a real MI355X: the kernels of
tests/test_f16_gpu.py compiled with the bf16 MFMA opcodes and conversions (var_amd/csrc/elem16.h) — the reference's other 16-bit option
(utils/arg_util.py `fp16: int  # 1: using fp16, 2: bf16`).  Same three layers of checks as the fp16 flavour, with bfloat16's 8 significant bits in
the tolerances (one rounding = 2^-9 relative; the fp16 tests state 2^-10 where these state 2^-7):
  (a) every entry point against float64 arithmetic on the same bf16 inputs, tile instantiations identical bit for bit;
  (b) end to end against the CPU twin OracleVAR(f16='bf16'), teacher-forced;
  (c) against the reference under torch.autocast(dtype=bfloat16) (fixtures e2e_acbf16_*) and the reference's fp32 run.
"""
import numpy as np
import pytest

from tests import util
from tests.test_f16_gpu import _models

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')
BF = torch.bfloat16
ULP = 2.0 ** -7            # a bf16 rounding of a result (half an ulp = 2^-9) with the same factor of slack the fp16 tests carry


def _hip():
    from var_amd import hip
    return hip


@pytest.mark.parametrize('tile', [0, 1, 2, 3])
@pytest.mark.parametrize('M,N,K,mode', [(128, 1024, 1024, 'none16'), (1152, 4096, 1024, 'gelu16'), (200, 1024, 4096, 'resid32'), (513, 520, 128, 'resid16in'),
                                        (1000, 3072, 192, 'none32'), (300, 2304, 9216, 'resid32'), (300, 5760, 1920, 'none16'), (4352, 4096, 128, 'gelu16')])
def test_gemm_bf16_every_tile_against_float64(tile, M, N, K, mode):
    """varhip_gemm_nt_bf16: every epilogue, every tile instantiation forced (256x256 whole tiles on the persistent kernel), ragged M and N,
    the d30 / d36 widths; all tiles identical bit for bit and right against float64 on the same bf16 operands"""
    hip = _hip()
    g = torch.Generator().manual_seed(M * 7 + N + K)
    A = (torch.randn(M, K, generator=g) * 0.7).to(BF).cuda(); W = (torch.randn(N, K, generator=g) * (1.5 / K ** 0.5)).to(BF).cuda()
    bias = (torch.randn(N, generator=g) * 0.2).cuda()
    rpg = 100
    gamma = (torch.randn((M + rpg - 1) // rpg, N, generator=g) * 0.5).cuda()
    resid32 = torch.randn(M, N, generator=g).cuda()
    out16 = mode in ('none16', 'gelu16')
    epi = {'none16': 0, 'none32': 0, 'gelu16': 1, 'resid32': 2, 'resid16in': 2}[mode]
    resid = None if epi != 2 else (resid32.to(BF) if mode == 'resid16in' else resid32)
    gm = gamma if mode == 'resid32' else None
    def run(t):
        out = torch.empty(M, N, dtype=BF if out16 else torch.float32, device='cuda')
        hip.lib().so.varhip_gemm16_force_tile(t)
        try: hip.call('gemm_nt_bf16', A, K, W, K, bias, out, N, int(out16), M, N, K, epi, resid, N, int(mode == 'resid16in'), gm, N, rpg, 1, 0, 0, 0)
        finally: hip.lib().so.varhip_gemm16_force_tile(-1)
        return out
    got = run(tile)
    assert torch.equal(got, run(1)), f'tile {tile} differs from the 64x64 tile'
    Ad, Wd = A.double().cpu(), W.double().cpu()
    ref = Ad @ Wd.T + bias.double().cpu()
    tol = 2e-6 * (Ad.abs() @ Wd.abs().T) + 1e-6
    if epi == 1: ref = torch.nn.functional.gelu(ref, approximate='tanh')
    if mode == 'resid32':
        gd = gamma.double().cpu().repeat_interleave(rpg, dim=0)[:M]
        ref = resid32.double().cpu() + ref * gd; tol = tol * gd.abs().clamp_min(1.0) + 1e-6 * ref.abs()
    elif mode == 'resid16in':
        ref = resid.double().cpu() + ref; tol = tol + 1e-6 * ref.abs()
    if out16: tol = tol + ref.abs() * ULP
    err = (got.double().cpu() - ref).abs()
    assert bool((err <= tol).all()), f'{mode} {M}x{N}x{K} tile {tile}: {int((err > tol).sum())} outside tolerance, max err {float(err.max()):.3e}'


@pytest.mark.parametrize('B2,l,H,pos0,l2', [(4, 9, 2, 5, 1), (2, 64, 4, 91, 1), (3, 25, 16, 0, 0), (4, 100, 4, 10, 1), (2, 169, 36, 55, 1)])
def test_gemm_qkv_bf16_against_float64(B2, l, H, pos0, l2):
    hip = _hip()
    C, K, Lmax = H * 64, H * 64, max(160, pos0 + l + 8)
    M = B2 * l
    g = torch.Generator().manual_seed(B2 * 100 + l)
    A = torch.randn(M, K, generator=g).to(BF); W = (torch.randn(3 * C, K, generator=g) * (1.0 / K ** 0.5)).to(BF)
    bias = torch.randn(3 * C, generator=g) * 0.1
    smul = torch.randn(H, generator=g) * 0.3 + 1.4
    res = []
    for tile in (-1, 2):
        q = torch.empty(M, C, dtype=BF, device='cuda'); kc = torch.zeros(B2, H, Lmax, 64, dtype=BF, device='cuda'); vc = torch.zeros_like(kc)
        hip.lib().so.varhip_gemm16_force_tile(tile)
        try: hip.call('gemm_qkv_bf16', A.cuda(), K, W.cuda(), K, bias.cuda(), M, C, K, smul.cuda(), 0.125, l2, q, kc, vc, B2, l, H, pos0, Lmax)
        finally: hip.lib().so.varhip_gemm16_force_tile(-1)
        res.append((q, kc, vc))
    for a, b in zip(*res): assert torch.equal(a, b)              # one summation order of the head's squares on every tile
    q, kc, vc = res[0]
    ref = (A.double() @ W.double().T + bias.double()).view(B2, l, 3, H, 64)
    rq, rk, rv = ref[:, :, 0], ref[:, :, 1], ref[:, :, 2]
    if l2:
        rq = rq / rq.norm(dim=-1, keepdim=True).clamp_min(1e-12) * smul.double().clamp_max(np.log(100)).exp().view(1, 1, H, 1)
        rk = rk / rk.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    else:
        rq = rq * 0.125
    def close(a, b, name):
        err = (a.double().cpu() - b).abs(); tol = b.abs() * ULP + 2e-4
        assert bool((err <= tol).all()), f'{name}: max err {float(err.max()):.3e}'
    close(q.view(B2, l, H, 64), rq, 'q')
    close(kc[:, :, pos0:pos0 + l].permute(0, 2, 1, 3), rk, 'k cache rows')
    close(vc[:, :, pos0:pos0 + l].permute(0, 2, 1, 3), rv, 'v cache rows')
    assert float(kc[:, :, :pos0].abs().max() if pos0 else 0) == 0 and float(kc[:, :, pos0 + l:].abs().max()) == 0


@pytest.mark.parametrize('B2,l,H,curL', [(2, 1, 2, 1), (3, 9, 2, 14), (2, 36, 3, 91), (2, 169, 2, 424), (1, 256, 2, 680), (2, 40, 1, 33),
                                         (2, 1024, 1, 2240), (1, 576, 36, 1112)])
def test_attn_bf16_against_twin(B2, l, H, curL):
    """bf16 attention vs its CPU twin (fp32 chains, p rounded to bf16 for p.v, bf16 output) on the same bf16 q / K / V; up to config #5's lengths"""
    hip = _hip()
    util.ensure_oracle_built()
    from oracle.var_oracle import lib, _p
    Lmax = curL + 7
    g = torch.Generator().manual_seed(l * 1000 + curL)
    q = torch.randn(B2 * l, H * 64, generator=g)
    q = (q.view(B2 * l, H, 64) / q.view(B2 * l, H, 64).norm(dim=-1, keepdim=True) * 6.0).view(B2 * l, H * 64).to(BF)
    k = torch.randn(B2, H, Lmax, 64, generator=g); k = (k / k.norm(dim=-1, keepdim=True)).to(BF)
    v = torch.randn(B2, H, Lmax, 64, generator=g).to(BF)
    out = torch.empty(B2 * l, H * 64, dtype=BF, device='cuda')
    hip.call('attn_cached_bf16', q.cuda(), k.cuda(), v.cuda(), out, B2, l, H, curL, Lmax)
    out2 = torch.empty_like(out)
    hip.call('attn_cached_bf16', q.cuda(), k.cuda(), v.cuda(), out2, B2, l, H, curL, Lmax)
    assert torch.equal(out, out2)
    want = np.empty((B2 * l, H * 64), np.float32)
    assert lib()['attn_cached_pbf16_f32'](_p(q.float().numpy()), _p(k.float().numpy()), _p(v.float().numpy()), _p(want), B2, l, H, curL, Lmax) == 0
    got = out.float().cpu().numpy()
    err = np.abs(got - want)
    # p is rounded to bf16 at slightly different fp32 values (hardware exp2 vs vm_exp); the output is one bf16 rounding of nearly equal values:
    # two bf16 ulps of the output magnitude
    tol = np.abs(want) * 2.0 ** -6 + 1.6e-2
    assert (err <= tol).all(), f'max err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}'
    for h in range(min(H, 4)):
        s = torch.einsum('btc,bjc->btj', q.view(B2, l, H, 64)[:, :, h].double(), k[:, h, :curL].double())
        ref = torch.einsum('btj,bjc->btc', s.softmax(-1), v[:, h, :curL].double()).reshape(B2 * l, 64).numpy()
        assert np.abs(got[:, h * 64:(h + 1) * 64] - ref).max() <= 8e-2


@pytest.mark.parametrize('B,H,W,Cin,Cout,res,omode', [(2, 16, 16, 32, 32, 0, 0), (2, 16, 16, 640, 640, 1, 0), (1, 32, 32, 320, 160, 0, 0),
                                                      (2, 32, 32, 160, 3, 0, 1), (1, 24, 40, 96, 64, 1, 0), (2, 8, 64, 32, 128, 1, 0), (2, 16, 32, 160, 320, 1, 0)])
@pytest.mark.parametrize('wm', [2, 4, 8])
def test_conv_bf16_against_float64(B, H, W, Cin, Cout, res, omode, wm):
    """varhip_conv3x3_nhwc_bf16 on its three kernels (forced) against float64 on the same bf16 data; GroupNorm partials = sums of the rounded outputs"""
    hip = _hip()
    g = torch.Generator().manual_seed(H * 31 + Cin + Cout)
    x = torch.randn(B, H, W, Cin, generator=g).to(BF)
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) * (1.0 / (9 * Cin) ** 0.5)).to(BF)
    bias = torch.randn(Cout, generator=g) * 0.1
    resid = torch.randn(B, H, W, Cout, generator=g).to(BF) if res else None
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), bias.double(), padding=1)
    if res: ref = ref + resid.double().permute(0, 3, 1, 2)
    nblk = hip.conv_gn_blocks(H, W, Cout) if (omode == 0 and Cout % 4 == 0) else 0
    part = torch.zeros(B, nblk, Cout, 2, dtype=torch.float64, device='cuda') if nblk else None
    if omode:
        out = torch.empty(B, Cout, H, W, dtype=torch.float32, device='cuda')
        ref = ref.clamp(-1, 1); ref = (ref + 1) * 0.5 if omode == 1 else ref
    else:
        out = torch.empty(B, H, W, Cout, dtype=BF, device='cuda')
    hip.lib().so.varhip_conv16_force_tile(wm)
    try: hip.call('conv3x3_nhwc_bf16', x.cuda(), w.cuda(), bias.cuda(), None if resid is None else resid.cuda(), out, part, B, H, W, Cin, Cout, omode)
    finally: hip.lib().so.varhip_conv16_force_tile(0)
    got = out.double().cpu() if omode else out.double().cpu().permute(0, 3, 1, 2)
    tol = 1e-5 + (0 if omode else ref.abs() * ULP) + 2e-6 * (9 * Cin) ** 0.5
    err = (got - ref).abs()
    assert bool((err <= tol).all()), f'max err {float(err.max()):.3e}'
    if nblk:
        o = out.double().cpu().view(B, H * W, Cout)
        assert torch.allclose(part[..., 0].sum(1).cpu(), o.sum(1), rtol=1e-5, atol=1e-3) and torch.allclose(part[..., 1].sum(1).cpu(), (o * o).sum(1), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize('B,H,W,Cin,Cout', [(2, 16, 16, 64, 32), (1, 32, 32, 320, 320), (2, 64, 32, 160, 160)])
@pytest.mark.parametrize('wm', [2, 4])
def test_upconv_phase_bf16_against_float64(B, H, W, Cin, Cout, wm):
    """Upsample2x (basic_vae.py:22-28) in its folded four-phase form on bf16 data against the phase form in float64 with the bf16-rounded phase weights"""
    hip = _hip()
    g = torch.Generator().manual_seed(H + Cin + Cout)
    h2, w2 = H // 2, W // 2
    x = torch.randn(B, h2, w2, Cin, generator=g).to(BF)
    w = torch.randn(Cout, 3, 3, Cin, generator=g) * (1.0 / (9 * Cin) ** 0.5)
    bias = torch.randn(Cout, generator=g) * 0.1
    wp = torch.empty(4, Cout, 2, 2, Cin, dtype=torch.float32, device='cuda')
    hip.call('upconv_pack_f32', w.cuda(), wp, Cin, Cout)
    wp16 = wp.to(BF)
    out = torch.empty(B, H, W, Cout, dtype=BF, device='cuda')
    hip.lib().so.varhip_conv16_force_tile(wm)
    try: hip.call('upconv_phase_bf16', x.cuda(), wp16, bias.cuda(), out, None, B, H, W, Cin, Cout)
    finally: hip.lib().so.varhip_conv16_force_tile(0)
    xd = x.double().permute(0, 3, 1, 2)
    ref = torch.empty(B, Cout, H, W, dtype=torch.float64)
    for py in range(2):
        for px in range(2):
            k = wp16[py * 2 + px].double().cpu().permute(0, 3, 1, 2)                 # [Cout][Cin][2][2]
            xp = torch.nn.functional.pad(xd, (1 - px, px, 1 - py, py))                # taps (a, b) read low-res pixel (y + a - 1 + py, x + b - 1 + px)
            ref[:, :, py::2, px::2] = torch.nn.functional.conv2d(xp, k, bias.double())
    err = (out.double().cpu().permute(0, 3, 1, 2) - ref).abs()
    assert bool((err <= ref.abs() * ULP + 1e-5 + 2e-6 * (4 * Cin) ** 0.5).all()), f'max err {float(err.max()):.3e}'


@pytest.mark.parametrize('B,HW,C,silu', [(2, 256, 640, 1), (3, 1024, 160, 1), (1, 100, 32, 0)])
def test_groupnorm_bf16_against_float64(B, HW, C, silu):
    hip = _hip()
    g = torch.Generator().manual_seed(HW + C)
    x = (torch.randn(B, HW, C, generator=g) * 1.7 + 0.3).to(BF)
    gamma, beta = torch.randn(C, generator=g) * 0.2 + 1.0, torch.randn(C, generator=g) * 0.2
    stats = torch.empty(B, 32, 2, dtype=torch.float32, device='cuda')
    scratch = torch.empty(hip.gn_scratch_elems(B, HW, C, 32), dtype=torch.float64, device='cuda')
    hip.call('gn_stats_bf16', x.cuda(), stats, scratch, B, HW, C, 32, 1e-6)
    xd = x.double().view(B, HW, 32, C // 32)
    mean = xd.mean(dim=(1, 3)); var = xd.var(dim=(1, 3), unbiased=False)
    assert torch.allclose(stats[..., 0].double().cpu(), mean, atol=1e-6) and torch.allclose(stats[..., 1].double().cpu(), (var + 1e-6).rsqrt(), rtol=1e-6)
    out = torch.empty(B, HW, C, dtype=BF, device='cuda')
    hip.call('gn_apply_bf16', x.cuda(), stats, gamma.cuda(), beta.cuda(), out, B, HW, C, 32, silu)
    ref = torch.nn.functional.group_norm(x.double().permute(0, 2, 1), 32, gamma.double(), beta.double(), eps=1e-6).permute(0, 2, 1)
    if silu: ref = torch.nn.functional.silu(ref)
    err = (out.double().cpu() - ref).abs()
    assert bool((err <= ref.abs() * ULP + 1e-3).all()), f'max err {float(err.max()):.3e}'
    y32 = torch.empty(B, HW, C, dtype=torch.float32, device='cuda')
    hip.call('cast_bf16_to_f32', out, y32, out.numel())
    back = torch.empty_like(out)
    hip.call('cast_f32_to_bf16', y32, back, out.numel())
    assert torch.equal(y32, out.float()) and torch.equal(back, out)
    z = torch.randn(1000, generator=g).cuda(); z16 = torch.empty(1000, dtype=BF, device='cuda')       # round-to-nearest-even like torch's
    hip.call('cast_f32_to_bf16', z, z16, 1000)
    assert torch.equal(z16, z.to(BF))
    xn = torch.randn(60, 256, generator=g).cuda(); sc = torch.randn(2, 256, generator=g).cuda(); sh = torch.randn(2, 256, generator=g).cuda()
    o = torch.empty(60, 256, dtype=BF, device='cuda')
    hip.call('ln_modulate_bf16out', xn, sc, 256, sh, 256, o, 60, 256, 30, 1e-6)
    want = torch.nn.functional.layer_norm(xn.double(), (256,), eps=1e-6) * (sc.double().repeat_interleave(30, 0) + 1) + sh.double().repeat_interleave(30, 0)
    assert bool(((o.double() - want).abs() <= want.abs() * ULP + 1e-5).all())


def test_decoder_bf16_vs_fp32_decoder():
    """VQVAE.fhat_to_img with the bf16 decoder against the fp32 HIP decoder on the same f_hat (d16-size decoder, 256x256, B=2): bf16 activations
    through ~50 layers, statistics / accumulation in fp32; stated budget about 4x the fp16 flavour's (3 more bits dropped per rounding)"""
    z, meta = util.load_case('d16_full')
    vae, var = _models(meta)
    g = torch.Generator().manual_seed(3)
    f_hat = (torch.randn(2, 32, 16, 16, generator=g) * 1.5).cuda()
    with torch.inference_mode():
        a = vae.fhat_to_img(f_hat).clone()
        vae._decoder_engine().set_precision('bf16')
        try:
            b = vae.fhat_to_img(f_hat).clone()
            b2 = vae.fhat_to_img(f_hat)
        finally:
            vae._decoder_engine().set_precision('f32')
        c = vae.fhat_to_img(f_hat)
    assert torch.equal(b, b2) and torch.equal(a, c)
    d = (a - b).abs()
    print(f'decoder bf16 vs f32: max |d| {float(d.max()):.3e}, mean |d| {float(d.mean()):.3e} (range [-1, 1])')
    assert float(d.max()) <= 1.6e-1 and float(d.mean()) <= 1.6e-2 and torch.isfinite(b).all()     # measured 4.3e-2 / 5.1e-3


@pytest.mark.parametrize('name', ['t_pn12345', 't_saln', 'd16_pn123', 'd36_saln_pn12346'])
def test_bf16_mode_vs_twin_and_reference(name):
    """end to end, teacher-forced with the reference's tokens: per-scale logits of the HIP bf16 mode vs the CPU twin (same rounding points, fp32
    chains vs MFMA order: stated 1.2e-2 x max|logit| (measured 4.5e-3), 3x the fp16 bar) and vs the reference's fp32 run (stated 1.2e-1 x max|logit|)"""
    z, meta = util.load_case(name)
    vae, var = _models(meta)
    pns = meta['patch_nums']
    noise = [torch.from_numpy(n) for n in util.regen_noise(meta, z)]
    labels = torch.tensor(meta['labels'], dtype=torch.int64, device='cuda')
    force = torch.from_numpy(z['idx'].astype(np.int64))
    var.set_hip_precision('bf16')
    try:
        eng = var.engine()
        img = eng.sample(len(meta['labels']), labels, None, meta['cfg'], meta['top_k'], meta['top_p'], noises=noise, force_idx=force, trace=True).cpu().numpy()
        tr = {k: [t.cpu().numpy() for t in v] for k, v in eng.last_trace.items()}
        free = eng.sample(len(meta['labels']), labels, None, meta['cfg'], meta['top_k'], meta['top_p'], noises=noise, trace=True)
        idx_free = torch.cat(eng.last_trace['idx'], dim=1).cpu().numpy()
    finally:
        var.set_hip_precision('f32')
    util.ensure_oracle_built()
    from oracle.var_oracle import OracleVAR
    var_sd, vae_sd = util.make_weights(meta)
    twin = OracleVAR(var_sd, vae_sd, pns, meta['depth'], attn_l2_norm=meta['attn_l2_norm'], shared_aln=meta['shared_aln'], f16='bf16')
    r = twin.run(meta['labels'], util.regen_noise(meta, z), meta['cfg'], meta['top_k'], meta['top_p'], force_idx=z['idx'].astype(np.int64))
    msgs, ok_all = [], True
    for si, pn in enumerate(pns):
        lg = tr['logits'][si]
        scale = max(float(np.abs(r['logits'][si]).max()), 1.0)
        ok, m = util.diff_report(f'{name} bf16 logits s{si} vs twin (|logit| max {scale:.2f})', lg, r['logits'][si], atol=1.2e-2 * scale, rtol=0); ok_all &= ok; msgs.append(m)
        want = z[f'logits_s{si}']
        got = lg if meta['full_logits'] else lg[:, sorted({0, pn * pn - 1}), :]
        ok, m = util.diff_report(f'{name} bf16 logits s{si} vs fp32 reference', got, want, atol=1.2e-1 * scale, rtol=0); ok_all &= ok; msgs.append(m)
    idx = np.concatenate(tr['idx'], axis=1)
    agree_tf = float((idx == z['idx']).mean()); agree_twin = float((idx == r['idx']).mean()); agree_free = float((idx_free == z['idx']).mean())
    msgs.append(f'{name}: bf16 token agreement with the fp32 reference, teacher-forced {agree_tf:.3f}, free-running {agree_free:.3f}; with the twin (teacher-forced) {agree_twin:.3f}')
    d = np.abs(img - z['img'])
    msgs.append(f'{name}: bf16 image vs reference max |d| {float(d.max()):.3e} mean {float(d.mean()):.2e}')
    print('\n'.join(msgs))
    assert ok_all, '\n'.join(msgs)
    assert agree_tf >= 0.8 and agree_twin >= 0.9, msgs[-2]
    assert float(d.max()) <= 1.6e-1 and float(d.mean()) <= 1.6e-2 and np.isfinite(free.cpu().numpy()).all()


@pytest.mark.parametrize('name', ['acbf16_t_pn12345', 'acbf16_d16_pn123'])
def test_bf16_mode_vs_reference_under_bf16_autocast(name):
    """the bf16 mode on the GPU against the REFERENCE's own bf16 execution (fixtures: autoregressive_infer_cfg under torch.autocast(dtype=bfloat16)),
    teacher-forced with the fixture's tokens; same bars as the CPU twin's test (tests/test_oracle_vs_golden.py): logits within 1.6e-2 x max|logit|
    (the reference's logits carry a bf16 rounding of their own: 3e-2 at |logit| 8..16), token agreement >= 95 %, pixels within 1e-1, mean <= 1e-2"""
    z, meta = util.load_case(name)
    assert meta['autocast_dtype'] == 'bf16'
    vae, var = _models(meta)
    pns = meta['patch_nums']
    noise = [torch.from_numpy(n) for n in util.regen_noise(meta, z)]
    labels = torch.tensor(meta['labels'], dtype=torch.int64, device='cuda')
    var.set_hip_precision('bf16')
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
        ok, m = util.diff_report(f'{name} bf16 logits s{si} vs the reference under bf16 autocast', got, want, atol=1.6e-2 * max(float(np.abs(want).max()), 1.0)); ok_all &= ok; msgs.append(m)
    agree = float((np.concatenate(tr['idx'], axis=1) == z['idx']).mean())
    d = np.abs(img - z['img'])
    msgs.append(f'{name}: token agreement with the reference under bf16 autocast (teacher-forced) {agree:.3f}; image max |d| {float(d.max()):.3e} mean {float(d.mean()):.2e}')
    print('\n'.join(msgs))
    assert ok_all and agree >= 0.95 and float(d.max()) <= 1e-1 and float(d.mean()) <= 1e-2, '\n'.join(msgs)


def test_bf16_mode_properties_d16_full():
    """d16, all 10 scales at B=4 in the bf16 mode: deterministic, tokens batch-slice invariant, finite, teacher-forced VAR.forward == the AR run's
    conditional logits bit for bit; switching back restores the fp32 and the fp16 results bit for bit"""
    z, meta = util.load_case('d16_full')
    vae, var = _models(meta)
    V, B = var.V, 4
    g = torch.Generator().manual_seed(5)
    noise = [torch.empty(B * pn * pn, V).exponential_(1, generator=g) for pn in var.patch_nums]
    labels = torch.tensor([1, 22, 333, 980], device='cuda')
    eng = var.engine()
    base32 = eng.sample(B, labels, None, 1.5, 900, 0.96, noises=noise).clone()
    var.set_hip_precision('f16')
    base16 = eng.sample(B, labels, None, 1.5, 900, 0.96, noises=noise).clone()
    var.set_hip_precision('bf16')
    try:
        a = eng.sample(B, labels, None, 1.5, 900, 0.96, noises=noise, trace=True).clone()
        ia = torch.cat(eng.last_trace['idx'], dim=1)
        ar_logits = torch.cat([lg[:B] for lg in eng.last_trace['logits']], dim=1)
        b = eng.sample(B, labels, None, 1.5, 900, 0.96, noises=noise, trace=True)
        assert torch.equal(a, b) and torch.equal(ia, torch.cat(eng.last_trace['idx'], dim=1))
        sub = eng.sample(2, labels[1:3], None, 1.5, 900, 0.96, noises=[n.view(B, -1, V)[1:3].reshape(-1, V) for n in noise], trace=True)
        assert torch.equal(torch.cat(eng.last_trace['idx'], dim=1), ia[1:3]), 'tokens must not depend on the batch neighbours'
        assert float((sub - a[1:3]).abs().max()) <= 4e-2          # (conv kernel choice and GroupNorm partial order follow the batch; 8x the fp16 bound)
        assert torch.isfinite(a).all() and float(a.min()) >= 0 and float(a.max()) <= 1 and not torch.equal(a, base16)
        ms, cur = [], 0
        for pn in var.patch_nums:
            ms.append(ia[:, cur:cur + pn * pn].contiguous()); cur += pn * pn
        var.cond_drop_rate = 0.0
        with torch.inference_mode():
            tf = var(labels, vae.quantize.idxBl_to_var_input(ms))
        assert torch.equal(tf, ar_logits), f'teacher-forced bf16 logits differ: max {float((tf - ar_logits).abs().max()):.3e}'
    finally:
        var.set_hip_precision('f16')
    assert torch.equal(eng.sample(B, labels, None, 1.5, 900, 0.96, noises=noise), base16)
    var.set_hip_precision('f32')
    assert torch.equal(eng.sample(B, labels, None, 1.5, 900, 0.96, noises=noise), base32)
