"""End-to-end parity of the HIP sampling path on a real MI355X, through the drop-in `models` API.

Three comparisons per case:
  (a) HIP vs CPU oracle on the same weights/noise: tokens, per-scale logits, f_hat and next-scale maps BIT-EXACT
      (free-running: the loop feeds its own tokens back, so one differing bit anywhere would show up as a token flip),
      decoded pixels within 1e-5 (GroupNorm statistics are accumulated in a different fp64 order);
  (b) HIP vs the reference itself (golden fixtures from tools/gen_golden.py): token ids identical, image within 1e-3
      (north_star tolerance), logits to fp32 rounding noise;
  (c) size-independent properties at sizes the oracle cannot reach in seconds (determinism, batch-slice invariance,
      incremental f_hat == non-incremental embed_to_fhat, image range).
The Exp(1) noise is the reference's own stream: regenerated with a CPU torch.Generator and verified against the fixture.
"""
import contextlib
import io

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

TINY = ['t_pn123_base', 't_pn12345', 't_nol2', 't_saln', 't_greedy', 't_nofilter', 't_b3_pn1234']
_MODELS = {}


def build_models(meta):
    """our nn.Modules on the GPU with the deterministic weights of var_amd.detinit (same as the fixtures' reference run)"""
    key = (meta['depth'], meta['ch'], tuple(meta['patch_nums']), meta['attn_l2_norm'], meta['shared_aln'])
    if key in _MODELS:
        return _MODELS[key]
    from models import build_vae_var
    from var_amd.detinit import fill_module_device_
    _MODELS.clear(); torch.cuda.empty_cache()
    with contextlib.redirect_stdout(io.StringIO()):
        vae, var = build_vae_var(device='cuda', patch_nums=tuple(meta['patch_nums']), depth=meta['depth'], ch=meta['ch'],
                                 shared_aln=meta['shared_aln'], attn_l2_norm=meta['attn_l2_norm'])
    # the detinit values computed on the device (bit-identical to the numpy generator: tests/test_host_cpu.py), seconds even for d36
    fill_module_device_(var, meta['depth'], 0, 'var.'); fill_module_device_(vae, meta['depth'], 0, 'vae.')
    _MODELS[key] = (vae.eval(), var.eval())
    return _MODELS[key]


def hip_run(meta, z, force=None):
    vae, var = build_models(meta)
    noise = [torch.from_numpy(n) for n in util.regen_noise(meta, z)]
    labels = torch.tensor(meta['labels'], dtype=torch.int64, device='cuda')
    img = var.engine().sample(len(meta['labels']), labels, None, meta['cfg'], meta['top_k'], meta['top_p'], noises=noise,
                              force_idx=None if force is None else torch.from_numpy(force.astype(np.int64)), trace=True)
    torch.cuda.synchronize()
    tr = var.engine().last_trace
    return img.cpu().numpy(), {k: [t.cpu().numpy() for t in v] for k, v in tr.items()}


def oracle_run(meta, z, force=None):
    util.ensure_oracle_built()
    from oracle.var_oracle import OracleVAR
    var_sd, vae_sd = util.make_weights(meta)
    orc = OracleVAR(var_sd, vae_sd, meta['patch_nums'], meta['depth'], attn_l2_norm=meta['attn_l2_norm'], shared_aln=meta['shared_aln'])
    return orc.run(meta['labels'], util.regen_noise(meta, z), meta['cfg'], meta['top_k'], meta['top_p'],
                   force_idx=None if force is None else force.astype(np.int64))


def _compare(name, with_oracle=True, logit_atol=2e-4):
    z, meta = util.load_case(name)
    pns = meta['patch_nums']
    img, tr = hip_run(meta, z)
    msgs, ok_all = [], True

    def rec(ok, m):
        nonlocal ok_all
        ok_all &= ok; msgs.append(('ok   ' if ok else 'FAIL ') + m)
    idx = np.concatenate(tr['idx'], axis=1)
    # (b) against the reference's fixture
    rec(*util.diff_report(f'{name} tokens vs reference', idx.astype(np.int32), z['idx']))
    rec(*util.diff_report(f'{name} image vs reference', img, z['img'], atol=1e-3))
    for si, pn in enumerate(pns):
        got = tr['logits'][si] if meta['full_logits'] else tr['logits'][si][:, sorted({0, pn * pn - 1}), :]
        rec(*util.diff_report(f'{name} logits s{si} vs reference', got, z[f'logits_s{si}'], atol=logit_atol, rtol=1e-5))
        rec(*util.diff_report(f'{name} f_hat s{si} vs reference', tr['f_hat'][si], z[f'f_hat_s{si}'], atol=2e-5, rtol=1e-5))
    # (a) against the oracle: bit-exact
    if with_oracle:
        r = oracle_run(meta, z)
        rec(*util.diff_report(f'{name} tokens vs oracle (free-running)', idx, r['idx']))
        for si in range(len(pns)):
            rec(*util.diff_report(f'{name} logits s{si} vs oracle (exact)', tr['logits'][si], r['logits'][si]))
            rec(*util.diff_report(f'{name} f_hat s{si} vs oracle (exact)', tr['f_hat'][si], r['f_hat'][si]))
            if si < len(pns) - 1:
                rec(*util.diff_report(f'{name} next map s{si} vs oracle (exact)', tr['pooled'][si], r['pooled'][si]))
        rec(*util.diff_report(f'{name} image vs oracle', img, r['img'], atol=1e-5))
    print('\n'.join(msgs))
    assert ok_all, '\n'.join(m for m in msgs if m.startswith('FAIL'))


@pytest.mark.parametrize('name', TINY)
def test_tiny_cases(name):
    _compare(name)


def test_baseline_config1_d16_pn123():
    """BASELINE.json configs[0]: VAR-d16, patch_nums=(1,2,3), B=2 — vs reference fixture and vs oracle"""
    _compare('d16_pn123', logit_atol=5e-4)


def test_d16_full_pyramid_vs_reference():
    """VAR-d16, 10 scales, 256x256, B=2: 680 token ids per image identical to the reference run; image within 1e-3"""
    _compare('d16_full', with_oracle=False, logit_atol=1e-3)


def test_d30_width_full_depth_vs_reference():
    """BASELINE.json configs[3]'s model, VAR-d30 (C=1920, 30 heads, depth 30, 2.0 B parameters), first three scales, B=2: token ids
    identical to the reference's own run (tests/golden/e2e_d30_pn123.npz), logits / f_hat / image within tolerance, and bit-exact
    against the oracle free-running"""
    _compare('d30_pn123', logit_atol=1e-3)


def test_d36_width_full_depth_shared_aln_vs_reference():
    """BASELINE.json configs[4]'s model, VAR-d36 (C=2304, 36 heads, depth 36, shared AdaLN, 2.3 B parameters), first five scales of the
    512-pixel schedule (1,2,3,4,6), B=2: tokens identical to the reference's run, bit-exact against the oracle"""
    _compare('d36_saln_pn12346', logit_atol=1e-3)


@pytest.mark.parametrize('depth,saln,pns', [(30, False, (1, 2, 3, 4, 5, 6, 8, 10, 13, 16)), (36, True, (1, 2, 3, 4, 6, 9, 13, 18, 24, 32))],
                         ids=['d30_256px', 'd36_512px'])
def test_full_size_wide_models_properties(depth, saln, pns):
    """BASELINE.json configs[3] / configs[4] at FULL size (VAR-d30 256x256, 10 scales, L=680; VAR-d36 512x512, patch_nums up to 32,
    L=2240, KV cache reused across scales), B=2 on one GPU, through the size-independent checks (no oracle / reference run exists at
    these sizes): determinism, batch-slice invariance, teacher-forced VAR.forward logits == the AR run's conditional logits bit for
    bit (SURVEY.md §4 identity (i)), incremental f_hat == embed_to_fhat and idxBl_to_img == the AR image (identity (ii))."""
    meta = dict(depth=depth, ch=160, patch_nums=list(pns), attn_l2_norm=True, shared_aln=saln)
    vae, var = build_models(meta)
    eng = var.engine()
    V, B, L = var.V, 2, var.L
    assert (var.C, var.num_heads, var.depth) == (64 * depth, depth, depth) and L == sum(p * p for p in pns)
    g = torch.Generator().manual_seed(depth)
    noise = [torch.empty(B * pn * pn, V).exponential_(1, generator=g) for pn in pns]
    labels = torch.tensor([207, 980], device='cuda')
    img = eng.sample(B, labels, None, 1.5, 900, 0.96, noises=noise, trace=True).clone()
    tr = eng.last_trace
    idx = torch.cat(tr['idx'], dim=1)
    ar_logits = torch.cat([lg[:B] for lg in tr['logits']], dim=1)                 # conditional rows
    f_hat = tr['f_hat'][-1].clone()
    P = 16 * pns[-1]
    assert img.shape == (B, 3, P, P) and torch.isfinite(img).all() and float(img.min()) >= 0 and float(img.max()) <= 1
    assert idx.shape == (B, L) and len(torch.unique(idx)) > 64
    # determinism
    img2 = eng.sample(B, labels, None, 1.5, 900, 0.96, noises=noise, trace=True)
    assert torch.equal(img, img2) and torch.equal(idx, torch.cat(eng.last_trace['idx'], dim=1))
    # batch-slice invariance: image 1 alone, fed its own noise rows
    sub = eng.sample(1, labels[1:], None, 1.5, 900, 0.96, noises=[n.view(B, -1, V)[1:].reshape(-1, V) for n in noise], trace=True)
    assert torch.equal(torch.cat(eng.last_trace['idx'], dim=1), idx[1:]) and torch.equal(sub, img[1:])
    # identity (i): teacher forcing over the KV cache reproduces the AR logits bit for bit
    ms, cur = [], 0
    for pn in pns:
        ms.append(idx[:, cur:cur + pn * pn].contiguous()); cur += pn * pn
    var.cond_drop_rate = 0.0
    with torch.inference_mode():
        tf = var(labels, vae.quantize.idxBl_to_var_input(ms))
    assert tf.shape == (B, L, V) and torch.equal(tf, ar_logits), f'teacher-forced logits differ: max {float((tf - ar_logits).abs().max()):.3e}'
    # identity (ii): non-incremental f_hat and image
    with torch.inference_mode():
        hs = [vae.quantize.embedding(i).transpose(1, 2).reshape(B, vae.Cvae, pn, pn) for i, pn in zip(ms, pns)]
        assert torch.equal(vae.quantize.embed_to_fhat(hs, all_to_max_scale=True, last_one=True), f_hat)
        im2 = vae.idxBl_to_img(ms, same_shape=True, last_one=True).add_(1).mul_(0.5)
    assert torch.equal(im2, img)


def test_engine_follows_parameter_edits():
    """the engines key their packed weight copies on (address, version counter): autograd-visible edits are picked up by themselves,
    `.data` edits after invalidate_engine() / load_state_dict"""
    z, meta = util.load_case('t_pn12345')
    vae, var = build_models(meta)
    labels = torch.tensor(meta['labels'], device='cuda')
    kw = dict(g_seed=3, cfg=1.5, top_k=900, top_p=0.96)
    base = var.autoregressive_infer_cfg(2, labels, **kw).clone()
    sd_var = {k: v.clone() for k, v in var.state_dict().items()}
    sd_vae = {k: v.clone() for k, v in vae.state_dict().items()}
    with torch.no_grad():
        var.head.weight.mul_(0.5)                                                  # bumps the version counter
    a = var.autoregressive_infer_cfg(2, labels, **kw).clone()
    assert not torch.equal(a, base)
    vae.decoder.conv_in.weight.data.mul_(1.5)                                      # `.data`: no counter — stale until told
    vae.invalidate_engines()
    b = var.autoregressive_infer_cfg(2, labels, **kw).clone()
    assert not torch.equal(b, a)
    var.load_state_dict(sd_var); vae.load_state_dict(sd_vae)                       # both loaders invalidate
    assert torch.equal(var.autoregressive_infer_cfg(2, labels, **kw), base)
    # VAR.forward outside eval() keeps the PyTorch branch (DropPath / dropout are live there), inside eval() the HIP one
    x = torch.randn(2, var.L - var.first_l, var.Cvae, device='cuda')
    var.cond_drop_rate = 0.0
    with torch.no_grad():
        ev = var(labels, x)
        var.train(); tr_ = var(labels, x); var.eval()
    assert ev.shape == tr_.shape == (2, var.L, var.V) and torch.isfinite(tr_).all()


def test_inpainting_more_smooth_vs_reference(golden_dir):
    """VAR.inpainting(more_smooth=True) with no fully kept scale (var.py:332-341): equals autoregressive sampling with more_smooth on the
    same noise bit for bit (the kept tokens never reach f_hat on that branch), and the reference's run within the gumbel tolerance;
    a fully kept scale is refused (undefined in the reference)."""
    import json
    from tests.test_oracle_vs_golden import regen_smooth_noise
    z = np.load(f'{golden_dir}/inpaint_ms_t_pn12345.npz')
    meta = json.loads(str(z['meta']))
    vae, var = build_models(meta)
    n1, n2 = regen_smooth_noise(meta, z)
    n1 = [torch.from_numpy(a) for a in n1]; n2 = [torch.from_numpy(a) for a in n2]
    labels = torch.tensor(meta['labels'], device='cuda')
    gt, mask = torch.from_numpy(z['gt'].astype(np.int64)).cuda(), torch.from_numpy(z['mask']).cuda()
    eng = var.engine()
    img = eng.sample(2, labels, None, meta['cfg'], meta['top_k'], meta['top_p'], noises=n1, gumbel_noises=n2, more_smooth=True, gt_tokens=gt, keep_mask=mask).clone()
    ar = eng.sample(2, labels, None, meta['cfg'], meta['top_k'], meta['top_p'], noises=n1, gumbel_noises=n2, more_smooth=True)
    assert torch.equal(img, ar)
    ok, m = util.diff_report('inpaint more_smooth image vs reference', img.cpu().numpy(), z['img'], atol=2e-3); print(m); assert ok, m
    out = var.inpainting(gt, mask, label=labels, g_seed=5, cfg=1.5, top_k=900, top_p=0.96, more_smooth=True)
    assert out.shape == img.shape and torch.isfinite(out).all()
    full = mask.clone(); full[:, :1] = True
    with pytest.raises(NotImplementedError):
        var.inpainting(gt, full, label=labels, g_seed=5, more_smooth=True)


def test_znorm_quantizer_on_hip_vs_reference(golden_dir):
    """VectorQuantizer2(using_znorm=True).f_to_idxBl_or_fhat on HIP (cosine arg-max kernel, quant.py:151-153) against the reference's
    tokens and f_hat's (tests/golden/nearest_code_cos.npz) and against the oracle twin on every scale's queries"""
    import json
    z = np.load(f'{golden_dir}/nearest_code_cos.npz')
    meta = json.loads(str(z['meta']))
    from models import VQVAE
    from var_amd.detinit import fill_module_device_
    pns = tuple(meta['patch_nums'])
    with contextlib.redirect_stdout(io.StringIO()):
        vae = VQVAE(vocab_size=4096, z_channels=32, ch=meta['ch'], using_znorm=True, test_mode=True, share_quant_resi=4, v_patch_nums=pns).cuda().eval()
    fill_module_device_(vae, meta['depth'], 0, 'vae.')
    f = torch.from_numpy(z['f']).cuda()
    with torch.inference_mode():
        idx = vae.quantize.f_to_idxBl_or_fhat(f, to_fhat=False)
        fh = vae.quantize.f_to_idxBl_or_fhat(f, to_fhat=True)
    for si in range(len(pns)):
        ok, m = util.diff_report(f'znorm tokens s{si} vs reference', idx[si].cpu().numpy().astype(np.int32), z[f'idx_s{si}']); print(m); assert ok, m
        ok, m = util.diff_report(f'znorm f_hat s{si} vs reference', fh[si].cpu().numpy(), z[f'f_hat_s{si}'], atol=2e-5, rtol=1e-5); print(m); assert ok, m
    util.ensure_oracle_built()
    from oracle.var_oracle import lib, _p
    from var_amd import hip
    g = torch.Generator().manual_seed(1)
    q = torch.randn(777, 32, generator=g)
    cb = vae.quantize.embedding.weight.detach()
    out = torch.empty(777, dtype=torch.int64, device='cuda')
    hip.call('nearest_code_cos_f32', q.cuda(), cb, out, 777, 4096, 32)
    want = np.zeros(777, np.int64)
    assert lib()['nearest_code_cos_f32'](_p(q.numpy()), _p(np.ascontiguousarray(cb.cpu().numpy())), _p(want), 777, 4096, 32) == 0
    assert np.array_equal(out.cpu().numpy(), want)


def test_demo_sample_calling_convention():
    """The reference's harness (demo_sample.py:43-68) calls the model inside inference_mode + fp16 autocast with 8 labels,
    cfg=4, top_k=900, top_p=0.95: the HIP path computes in fp32 regardless, so the enclosing autocast must not change a bit;
    VAR.forward (teacher forcing, eval_prob.py:420-446) under the same contexts likewise."""
    z, meta = util.load_case('t_pn12345')
    vae, var = build_models(meta)
    labels = torch.tensor((980, 980, 437, 437, 22, 22, 562, 562), device='cuda')
    B = labels.numel()
    with torch.inference_mode():
        with torch.autocast('cuda', enabled=True, dtype=torch.float16, cache_enabled=True):
            a = var.autoregressive_infer_cfg(B=B, label_B=labels, cfg=4, top_k=900, top_p=0.95, g_seed=0, more_smooth=False)
        b = var.autoregressive_infer_cfg(B=B, label_B=labels, cfg=4, top_k=900, top_p=0.95, g_seed=0, more_smooth=False)
    P = 16 * meta['patch_nums'][-1]
    assert a.shape == (B, 3, P, P) and a.dtype == torch.float32 and float(a.min()) >= 0.0 and float(a.max()) <= 1.0
    assert torch.equal(a, b), 'an enclosing autocast context changed the result'
    var.cond_drop_rate = 0.0                                  # VAR.forward drops labels at random otherwise (var.py:200), also in eval
    x = torch.randn(B, var.L - var.first_l, var.Cvae, device='cuda')
    with torch.inference_mode():
        with torch.autocast('cuda', enabled=True, dtype=torch.float16):
            l1 = var(labels, x)
        l2 = var(labels, x)
    assert l1.dtype == torch.float32 and torch.equal(l1, l2)


def test_precision_auto_follows_the_callers_autocast():
    """set_hip_precision('auto') (opt-in; the default stays 'f32'): the demo's calling convention (demo_sample.py:66-68) under fp16 / bf16 autocast
    equals an explicit set_hip_precision('f16' / 'bf16') bit for bit, outside the context (or with enabled=False) it equals the f32 call; the
    teacher-forced VAR.forward follows the same rule (reference basic_var.py:97 branches on the autocast dtype)."""
    z, meta = util.load_case('t_pn12345')
    vae, var = build_models(meta)
    labels = torch.tensor((980, 980, 437, 437, 22, 22, 562, 562), device='cuda')
    B = labels.numel()
    kw = dict(B=B, label_B=labels, cfg=4, top_k=900, top_p=0.95, g_seed=0, more_smooth=False)
    var.cond_drop_rate = 0.0
    x = torch.randn(B, var.L - var.first_l, var.Cvae, device='cuda')
    try:
        want, want_tf = {}, {}
        with torch.inference_mode():
            for prec in ('f32', 'f16', 'bf16'):
                var.set_hip_precision(prec)
                want[prec] = var.autoregressive_infer_cfg(**kw).clone()
                want_tf[prec] = var(labels, x).clone()
            assert not torch.equal(want['f32'], want['f16']) and not torch.equal(want['f16'], want['bf16'])
            var.set_hip_precision('auto')
            for prec, dt in (('f16', torch.float16), ('bf16', torch.bfloat16), ('f16', torch.float16)):
                with torch.autocast('cuda', enabled=True, dtype=dt, cache_enabled=True):
                    got, got_tf = var.autoregressive_infer_cfg(**kw), var(labels, x)
                assert var.engine().precision == prec
                assert torch.equal(got, want[prec]) and torch.equal(got_tf, want_tf[prec]), f"'auto' under {dt} autocast differs from set_hip_precision('{prec}')"
                got = var.autoregressive_infer_cfg(**kw)                         # outside the context: f32
                assert var.engine().precision == 'f32' and torch.equal(got, want['f32'])
            with torch.autocast('cuda', enabled=False, dtype=torch.float16):
                assert torch.equal(var.autoregressive_infer_cfg(**kw), want['f32'])
            # the default policy ignores the context (test_demo_sample_calling_convention); an explicit mode does too
            var.set_hip_precision('bf16')
            with torch.autocast('cuda', enabled=True, dtype=torch.float16):
                assert torch.equal(var.autoregressive_infer_cfg(**kw), want['bf16'])
    finally:
        var.set_hip_precision('f32')
    with pytest.raises(ValueError):
        var.set_hip_precision('fp8')


def test_first_call_on_a_side_stream_without_warmup():
    """Derived weight copies (16-bit casts, packed ada_lin, phase-packed conv kernels) are built by whichever call comes first, on ITS stream; a
    call issued at once on another stream has to wait for those kernels (an event recorded behind them), not read half-written copies: no warm-up,
    no synchronize between the two calls, results equal to the same calls issued serially afterwards."""
    z, meta = util.load_case('d16_pn123')
    for prec in ('f32', 'f16'):
        _MODELS.clear()                                       # a model no call has touched yet: nothing derived exists
        vae, var = build_models(meta)
        var.rng = torch.Generator(device='cuda')
        var.set_hip_precision(prec)
        B = 4
        labels = ((torch.arange(B) * 91) % 1000).cuda()
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        with torch.inference_mode():
            with torch.cuda.stream(sa): a = var.autoregressive_infer_cfg(B, labels, g_seed=5, cfg=1.5, top_k=900, top_p=0.96)
            with torch.cuda.stream(sb): b = var.autoregressive_infer_cfg(B, labels, g_seed=6, cfg=1.5, top_k=900, top_p=0.96)
            torch.cuda.synchronize()
            a2 = var.autoregressive_infer_cfg(B, labels, g_seed=5, cfg=1.5, top_k=900, top_p=0.96)
            b2 = var.autoregressive_infer_cfg(B, labels, g_seed=6, cfg=1.5, top_k=900, top_p=0.96)
        assert torch.equal(a, a2) and torch.equal(b, b2), f'{prec}: a first call on a side stream read weights that were still being built'
        var.set_hip_precision('f32')


def test_public_api_and_properties():
    """VAR.autoregressive_infer_cfg with the device generator: determinism, output contract, and batch-slice invariance
    under injected noise (images are independent: SURVEY.md §8e), on the d16 (1,2,3) model."""
    z, meta = util.load_case('d16_pn123')
    vae, var = build_models(meta)
    B = 5
    labels = torch.tensor([1, 22, 333, 980, 1000], device='cuda')           # 1000 == num_classes: the unconditional class is a legal label
    with torch.inference_mode():
        a = var.autoregressive_infer_cfg(B, labels, g_seed=7, cfg=1.5, top_k=900, top_p=0.96)
        b = var.autoregressive_infer_cfg(B, labels, g_seed=7, cfg=1.5, top_k=900, top_p=0.96)
        c = var.autoregressive_infer_cfg(B, labels, g_seed=8, cfg=1.5, top_k=900, top_p=0.96)
        d = var.autoregressive_infer_cfg(2, 7, g_seed=1, cfg=4.0, top_k=0, top_p=0.0)               # int label form
    assert a.shape == (B, 3, 48, 48) and a.dtype == torch.float32 and d.shape == (2, 3, 48, 48)
    assert torch.equal(a, b), 'same g_seed must reproduce bit-identical images'
    assert not torch.equal(a, c)
    assert float(a.min()) >= 0.0 and float(a.max()) <= 1.0 and torch.isfinite(a).all()
    # batch-slice invariance: sample 4 images with given noise; images 1..2 alone with the matching noise rows give the same pixels
    eng = var.engine()
    V = var.V
    g = torch.Generator().manual_seed(5)
    noise = [torch.empty(4 * pn * pn, V).exponential_(1, generator=g) for pn in var.patch_nums]
    lab = torch.tensor([5, 6, 7, 8], device='cuda')
    full = eng.sample(4, lab, None, 1.5, 900, 0.96, noises=noise).clone()
    sub_noise = [n.view(4, -1, V)[1:3].reshape(-1, V) for n in noise]
    sub = eng.sample(2, lab[1:3], None, 1.5, 900, 0.96, noises=sub_noise)
    assert torch.equal(full[1:3], sub), 'an image must not depend on its batch neighbours'
    with pytest.raises(ValueError):
        eng.sample(2, torch.tensor([5, 1001], device='cuda'), None, 1.5, 0, 0.0)
    with pytest.raises(ValueError):
        var.smooth_sampling(torch.zeros(2, 3, dtype=torch.long, device='cuda'), 4, label=labels)       # gt_tokens must be (B, L)


def test_incremental_fhat_equals_embed_to_fhat_and_decoder_api():
    """SURVEY.md §4 identity (ii): the AR loop's accumulated f_hat == VectorQuantizer2.embed_to_fhat on its tokens (both on the HIP
    quantizer kernels: bit-identical), and VQVAE.fhat_to_img (HIP decoder via the public API) reproduces the loop's image."""
    z, meta = util.load_case('t_pn12345')
    vae, var = build_models(meta)
    img, tr = hip_run(meta, z)
    idx = np.concatenate(tr['idx'], axis=1)
    ms, cur = [], 0
    for pn in meta['patch_nums']:
        ms.append(torch.from_numpy(idx[:, cur:cur + pn * pn]).cuda()); cur += pn * pn
    with torch.inference_mode():
        hs = [vae.quantize.embedding(i).transpose(1, 2).reshape(len(meta['labels']), vae.Cvae, pn, pn) for i, pn in zip(ms, meta['patch_nums'])]
        f_ref = vae.quantize.embed_to_fhat(hs, all_to_max_scale=True, last_one=True)
        ok, m = util.diff_report('f_hat incremental vs embed_to_fhat (HIP both: exact)', tr['f_hat'][-1], f_ref.cpu().numpy())
        print(m); assert ok, m
        f_list = vae.quantize.embed_to_fhat(hs, all_to_max_scale=True, last_one=False)
        for si in range(len(hs)):
            ok, m = util.diff_report(f'embed_to_fhat list s{si} (exact)', f_list[si].cpu().numpy(), tr['f_hat'][si]); assert ok, m
        im2 = vae.fhat_to_img(torch.from_numpy(tr['f_hat'][-1]).cuda()).add_(1).mul_(0.5)
    ok, m = util.diff_report('fhat_to_img API vs loop image', im2.cpu().numpy(), img, atol=1e-6)
    print(m); assert ok, m


@pytest.mark.parametrize('name', ['inpaint_t_pn12345', 'inpaint_d16_pn123'])
def test_inpainting_vs_reference_and_oracle(name, golden_dir):
    """VAR.inpainting (fork API) on HIP: final tokens identical to the reference's run and to the oracle; kept tokens untouched"""
    import json
    from tests.test_oracle_vs_golden import regen_inpaint_noise
    z = np.load(f'{golden_dir}/{name}.npz')
    meta = json.loads(str(z['meta']))
    vae, var = build_models(meta)
    noise = [torch.from_numpy(n) for n in regen_inpaint_noise(meta, z)]
    gt, mask = torch.from_numpy(z['gt'].astype(np.int64)).cuda(), torch.from_numpy(z['mask']).cuda()
    labels = torch.tensor(meta['labels'], device='cuda')
    img = var.engine().sample(len(meta['labels']), labels, None, meta['cfg'], meta['top_k'], meta['top_p'], noises=noise, trace=True,
                              gt_tokens=gt, keep_mask=mask)
    idx = torch.cat(var.engine().last_trace['idx'], dim=1).cpu().numpy()
    ok, m = util.diff_report(f'{name} tokens vs reference', idx.astype(np.int32), z['idx']); print(m); assert ok, m
    ok, m = util.diff_report(f'{name} image vs reference', img.cpu().numpy(), z['img'], atol=1e-3); print(m); assert ok, m
    util.ensure_oracle_built()
    from oracle.var_oracle import OracleVAR
    var_sd, vae_sd = util.make_weights(meta)
    r = OracleVAR(var_sd, vae_sd, meta['patch_nums'], meta['depth']).run(meta['labels'], regen_inpaint_noise(meta, z), meta['cfg'], meta['top_k'],
                                                                           meta['top_p'], gt_tokens=z['gt'].astype(np.int64), keep_mask=z['mask'])
    ok, m = util.diff_report(f'{name} tokens vs oracle', idx, r['idx']); print(m); assert ok, m
    ok, m = util.diff_report(f'{name} f_hat vs oracle (exact)', var.engine().last_trace['f_hat'][-1].cpu().numpy(), r['f_hat'][-1]); print(m); assert ok, m
    # public API with the device generator: kept tokens are respected, call is deterministic, bad mask shape raises ValueError
    a = var.inpainting(gt, mask, label=labels, g_seed=3, cfg=meta['cfg'], top_k=meta['top_k'], top_p=meta['top_p'])
    b = var.inpainting(gt, mask, label=labels, g_seed=3, cfg=meta['cfg'], top_k=meta['top_k'], top_p=meta['top_p'])
    assert torch.equal(a, b) and a.shape == img.shape
    full = var.inpainting(gt, torch.ones_like(mask), label=labels, g_seed=3)          # everything kept == decode of gt tokens
    ms, cur = [], 0
    for pn in meta['patch_nums']:
        ms.append(gt[:, cur:cur + pn * pn]); cur += pn * pn
    with torch.inference_mode():          # (with autograd enabled the VQVAE methods keep their PyTorch branch: tolerance, not identity)
        ref_img = vae.idxBl_to_img(ms, same_shape=True, last_one=True).add_(1).mul_(0.5)
    ok, m = util.diff_report('all-kept inpainting == idxBl_to_img (exact)', full.cpu().numpy(), ref_img.cpu().numpy()); print(m); assert ok, m
    with pytest.raises(ValueError):
        var.inpainting(gt, mask[:, :-1], label=labels)


@pytest.mark.parametrize('name', ['smooth_t_pn12345_count', 'smooth_t_pn12345_thr', 'smooth_t_pn12345_count_ms', 'smooth_d16_pn123_count'])
def test_smooth_sampling_vs_reference_and_oracle(name, golden_dir):
    """VAR.smooth_sampling (fork API, var.py:367-572) on HIP: chosen tokens identical to the reference's run, both log-likelihood
    sums as in tests/test_oracle_vs_golden.py, and HIP == oracle bit for bit (tokens, per-row values, f_hat)."""
    import json
    z = np.load(f'{golden_dir}/{name}.npz')
    meta = json.loads(str(z['meta']))
    vae, var = build_models(meta)
    B = len(meta['labels'])
    labels = torch.tensor(meta['labels'], device='cuda')
    gt = torch.from_numpy(z['gt'].astype(np.int64)).cuda()
    gum = None
    if meta['more_smooth']:
        g = torch.Generator(); g.manual_seed(meta['seed'])
        gum = [torch.empty(B, pn * pn, meta['V']).exponential_(generator=g).view(-1, meta['V']) for pn in meta['patch_nums']]
    eng = var.engine()
    img = eng.sample(B, labels, None, meta['cfg'], 0, 0.0, trace=True, more_smooth=meta['more_smooth'], gumbel_noises=gum,
                     smooth=dict(gt=gt, n=meta['n'], thr=meta['thr']))
    idx = torch.cat(eng.last_trace['idx'], dim=1).cpu().numpy()
    f_hat = eng.last_trace['f_hat'][-1].cpu().numpy()
    sll, sdl = (float(t) for t in eng.last_smooth)
    nrows = z['gt'].size
    if 'idx' in z.files:
        ok, m = util.diff_report(f'{name} tokens vs reference', idx.astype(np.int32), z['idx']); print(m); assert ok, m
    assert abs(sll - float(z['sum_ll'])) <= max(1.0, nrows / 100), (sll, float(z['sum_ll']))
    assert abs(sdl - float(z['sum_dist_ll'])) <= nrows * 2.0 * float(z['self_dist_max']) + 1e-3, (sdl, float(z['sum_dist_ll']))
    ok, m = util.diff_report(f'{name} image vs reference', img.cpu().numpy(), z['img'], atol=2e-3 if meta['more_smooth'] else 1e-3); print(m); assert ok, m
    util.ensure_oracle_built()
    from oracle.var_oracle import OracleVAR
    var_sd, vae_sd = util.make_weights(meta)
    r = OracleVAR(var_sd, vae_sd, meta['patch_nums'], meta['depth']).run(meta['labels'], None, meta['cfg'], 0, 0.0, more_smooth=meta['more_smooth'],
                                                                           gumbel_noises=None if gum is None else [a.numpy() for a in gum],
                                                                           smooth=dict(gt=z['gt'].astype(np.int64), n=meta['n'], thr=meta['thr']))
    ok, m = util.diff_report(f'{name} tokens vs oracle', idx, r['idx']); print(m); assert ok, m
    ok, m = util.diff_report(f'{name} f_hat vs oracle (exact)', f_hat, r['f_hat'][-1]); print(m); assert ok, m
    assert sll == float(r['sum_ll']), (sll, float(r['sum_ll']))
    assert abs(sdl - float(r['sum_dist_ll'])) <= 1e-4 * nrows        # same addends; torch's sum order is its own
    # public API: deterministic, returns the documented triple
    a = var.smooth_sampling(gt, meta['n'], label=labels, g_seed=3, cfg=meta['cfg'], more_smooth=meta['more_smooth'], neighbor_threshold=meta['thr'])
    b = var.smooth_sampling(gt, meta['n'], label=labels, g_seed=3, cfg=meta['cfg'], more_smooth=meta['more_smooth'], neighbor_threshold=meta['thr'])
    assert torch.equal(a[0], b[0]) and a[0].shape == img.shape and float(a[1]) == float(b[1]) and a[1].dtype == torch.float32
    if not meta['more_smooth']:
        assert torch.equal(a[0], img) and float(a[1]) == sll


def test_more_smooth_vs_oracle_and_reference(golden_dir):
    """more_smooth=True (gumbel-softmax embeddings): HIP == oracle bit for bit (same vm_log/vm_exp, same sums); vs the reference
    fixture within the 1/tau-amplified tolerance of tests/test_oracle_vs_golden.py::test_more_smooth_case"""
    import json
    from tests.test_oracle_vs_golden import regen_smooth_noise
    z = np.load(f'{golden_dir}/more_smooth_t_pn12345.npz')
    meta = json.loads(str(z['meta']))
    vae, var = build_models(meta)
    n1, n2 = regen_smooth_noise(meta, z)
    labels = torch.tensor(meta['labels'], device='cuda')
    img = var.engine().sample(len(meta['labels']), labels, None, meta['cfg'], meta['top_k'], meta['top_p'], noises=[torch.from_numpy(a) for a in n1],
                              trace=True, more_smooth=True, gumbel_noises=[torch.from_numpy(a) for a in n2])
    f_hat = var.engine().last_trace['f_hat'][-1].cpu().numpy()
    util.ensure_oracle_built()
    from oracle.var_oracle import OracleVAR
    var_sd, vae_sd = util.make_weights(meta)
    r = OracleVAR(var_sd, vae_sd, meta['patch_nums'], meta['depth']).run(meta['labels'], n1, meta['cfg'], meta['top_k'], meta['top_p'],
                                                                           more_smooth=True, gumbel_noises=n2)
    ok, m = util.diff_report('more_smooth f_hat vs oracle (exact)', f_hat, r['f_hat'][-1]); print(m); assert ok, m
    ok, m = util.diff_report('more_smooth image vs oracle', img.cpu().numpy(), r['img'], atol=1e-5); print(m); assert ok, m
    ok, m = util.diff_report('more_smooth image vs reference', img.cpu().numpy(), z['img'], atol=2e-3); print(m); assert ok, m
    a = var.autoregressive_infer_cfg(2, labels, g_seed=4, cfg=1.5, top_k=900, top_p=0.96, more_smooth=True)
    b = var.autoregressive_infer_cfg(2, labels, g_seed=4, cfg=1.5, top_k=900, top_p=0.96, more_smooth=True)
    c = var.autoregressive_infer_cfg(2, labels, g_seed=4, cfg=1.5, top_k=900, top_p=0.96, more_smooth=False)
    assert torch.equal(a, b) and not torch.equal(a, c) and torch.isfinite(a).all()


@pytest.mark.parametrize('width,heads,saln,pns', [(1920, 30, False, (1, 2, 3)), (2304, 36, True, (1, 2, 3, 4, 6))])
def test_wide_models_vs_oracle(width, heads, saln, pns):
    """the widths of VAR-d30 (C=1920, 30 heads; BASELINE.json configs[3]) and VAR-d36 (C=2304, 36 heads, shared AdaLN, the 512px
    scale schedule's first scales; configs[4]) at depth 2: every C-dependent kernel (LayerNorm with C % 256 != 0, GEMM N = 1920 /
    5760 / 7680, 30-36 heads in attention and q/k/v prep) bit-exact against the oracle, free-running"""
    from models import VAR, VQVAE
    from var_amd import shapes
    from var_amd.detinit import fill_module_, make_state_dict
    depth, ch, labels, seed = 2, 32, [11, 987], 5
    with contextlib.redirect_stdout(io.StringIO()):
        vae = VQVAE(vocab_size=4096, z_channels=32, ch=ch, test_mode=True, share_quant_resi=4, v_patch_nums=pns).cuda()
        var = VAR(vae_local=vae, depth=depth, embed_dim=width, num_heads=heads, shared_aln=saln, attn_l2_norm=True, patch_nums=pns).cuda().eval()
    fill_module_(var, depth, 0, 'var.'); fill_module_(vae, depth, 0, 'vae.')
    g = torch.Generator().manual_seed(seed)
    noise = [torch.empty(len(labels) * pn * pn, 4096).exponential_(1, generator=g) for pn in pns]
    img = var.engine().sample(len(labels), torch.tensor(labels, device='cuda'), None, 1.5, 900, 0.96, noises=noise, trace=True)
    tr = var.engine().last_trace
    util.ensure_oracle_built()
    from oracle.var_oracle import OracleVAR
    var_sd = make_state_dict(shapes.var_shapes(depth, pns, shared_aln=saln, embed_dim=width, num_heads=heads), depth=depth, seed=0, prefix='var.')
    var_sd['lvl_1L'] = np.concatenate([np.full((p * p,), i, dtype=np.int64) for i, p in enumerate(pns)]).reshape(1, -1)
    vae_sd = make_state_dict(shapes.vae_shapes(ch=ch, patch_nums=pns, include_encoder=False), depth=depth, seed=0, prefix='vae.')
    r = OracleVAR(var_sd, vae_sd, pns, depth, shared_aln=saln).run(labels, [n.numpy() for n in noise], 1.5, 900, 0.96)
    ok, m = util.diff_report(f'C={width} tokens vs oracle', torch.cat(tr['idx'], 1).cpu().numpy(), r['idx']); print(m); assert ok, m
    for si in range(len(pns)):
        ok, m = util.diff_report(f'C={width} logits s{si} vs oracle (exact)', tr['logits'][si].cpu().numpy(), r['logits'][si]); print(m); assert ok, m
    ok, m = util.diff_report(f'C={width} image vs oracle', img.cpu().numpy(), r['img'], atol=1e-5); print(m); assert ok, m


def test_encode_side_and_teacher_forcing_vs_reference(golden_dir):
    """SURVEY.md §8f row 3 on HIP: image -> encoder -> residual quantisation -> tokens -> teacher-forcing input -> VAR.forward logits,
    through the public module API, against the reference's own outputs (tests/golden/encode_t_pn12345.npz)"""
    import json
    z = np.load(f'{golden_dir}/encode_t_pn12345.npz')
    meta = json.loads(str(z['meta']))
    from models import build_vae_var
    from var_amd.detinit import fill_module_
    with contextlib.redirect_stdout(io.StringIO()):
        vae, var = build_vae_var(device='cuda', patch_nums=tuple(meta['patch_nums']), depth=meta['depth'], ch=meta['ch'])
    fill_module_(var, meta['depth'], 0, 'var.'); fill_module_(vae, meta['depth'], 0, 'vae.')
    var.eval(); var.cond_drop_rate = 0.0
    img = torch.from_numpy(z['img']).cuda()
    with torch.inference_mode():
        f = vae.img_to_post(img)
        idx = vae.img_to_idxBl(img)
        fh = vae.img_to_fhat(img)
        x_in = vae.quantize.idxBl_to_var_input(idx)
        logits = var(torch.tensor(meta['labels'], device='cuda'), x_in)
    ok, m = util.diff_report('encoder+quant_conv f vs reference', f.cpu().numpy(), z['f'], atol=5e-5, rtol=1e-4); print(m); assert ok, m
    for si, i in enumerate(idx):
        ok, m = util.diff_report(f'encode tokens s{si} vs reference', i.cpu().numpy().astype(np.int32), z[f'idx_s{si}']); print(m); assert ok, m
    ok, m = util.diff_report('f_hat (last) vs reference', fh[-1].cpu().numpy(), z['f_hat_last'], atol=2e-5, rtol=1e-5); print(m); assert ok, m
    ok, m = util.diff_report('teacher-forcing input vs reference', x_in.cpu().numpy(), z['var_input'], atol=2e-5, rtol=1e-5); print(m); assert ok, m
    ok, m = util.diff_report('teacher-forced logits vs reference', logits.cpu().numpy(), z['logits'], atol=3e-4, rtol=1e-5); print(m); assert ok, m
    # identity (i) of SURVEY.md §4 on the HIP path, bit-exact: teacher-forced logits on the tokens an AR run sampled == that run's conditional logits
    z2, meta2 = util.load_case('t_pn12345')
    img_ar, tr = hip_run(meta2, z2)
    vae2, var2 = build_models(meta2)
    var2.cond_drop_rate = 0.0
    ms, cur = [], 0
    for pn in meta2['patch_nums']:
        ms.append(torch.cat(tr['idx'], 1)[:, cur:cur + pn * pn].cuda() if isinstance(tr['idx'][0], torch.Tensor) else torch.from_numpy(np.concatenate(tr['idx'], 1)[:, cur:cur + pn * pn]).cuda())
        cur += pn * pn
    with torch.inference_mode():
        tf = var2(torch.tensor(meta2['labels'], device='cuda'), vae2.quantize.idxBl_to_var_input(ms)).cpu().numpy()
    B = len(meta2['labels'])
    ar = np.concatenate([lg[:B] for lg in tr['logits']], axis=1)
    ok, m = util.diff_report('teacher-forced == AR conditional logits (exact)', tf, ar); print(m); assert ok, m


def test_d16_batch64_properties():
    """BASELINE.json configs[1] at full size (d16, 10 scales, B=64): determinism and agreement of the first two images with the
    B=2 reference fixture when fed the same noise rows (batch-slice invariance at the headline shape)."""
    z, meta = util.load_case('d16_full')
    vae, var = build_models(meta)
    eng = var.engine()
    V, B = var.V, 64
    g = torch.Generator().manual_seed(99)
    labels = torch.cat([torch.tensor(meta['labels']), (torch.arange(B - 2) * 7) % 1000]).cuda()
    ref_noise = util.regen_noise(meta, z)
    noise = []
    for si, pn in enumerate(meta['patch_nums']):
        n = torch.empty(B * pn * pn, V).exponential_(1, generator=g).view(B, pn * pn, V)
        n[:2] = torch.from_numpy(ref_noise[si]).view(2, pn * pn, V)
        noise.append(n.view(-1, V))
    img = eng.sample(B, labels, None, meta['cfg'], meta['top_k'], meta['top_p'], noises=noise, trace=True)
    idx = torch.cat(eng.last_trace['idx'], dim=1).cpu().numpy()
    ok, m = util.diff_report('B=64: tokens of images 0,1 vs reference fixture', idx[:2].astype(np.int32), z['idx']); print(m); assert ok, m
    ok, m = util.diff_report('B=64: images 0,1 vs reference fixture', img[:2].cpu().numpy(), z['img'], atol=1e-3); print(m); assert ok, m
    assert torch.isfinite(img).all() and float(img.min()) >= 0 and float(img.max()) <= 1
    assert len({tuple(r) for r in idx[:, :30].tolist()}) > 32, 'different labels/noise must give different token maps'


@pytest.mark.parametrize('prec', ['f32', 'f16'])
def test_calls_in_flight_on_two_streams(prec):
    """Calls issued on different HIP streams may be in flight together on ONE model (workspaces are per stream; a call's latency-bound small
    scales then run beside another call's decoder: tools/exp/two_stream_steps.py, +7.5 % images/s in the 16-bit mode): six
    autoregressive_infer_cfg calls (d16, all 10 scales, B=8) alternating between two streams give the images of the same calls issued one
    after the other on one stream, bit for bit."""
    z, meta = util.load_case('d16_full')
    vae, var = build_models(meta)
    B = 8
    labels = ((torch.arange(B) * 37) % 1000).cuda()
    var.rng = torch.Generator(device='cuda')
    var.set_hip_precision(prec)
    try:
        def call(i):
            with torch.inference_mode():
                return var.autoregressive_infer_cfg(B, labels, g_seed=100 + i, cfg=1.5, top_k=900, top_p=0.96)
        serial = [call(i).clone() for i in range(6)]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        outs = []
        for i in range(6):
            with torch.cuda.stream(streams[i % 2]):
                outs.append(call(i))
        torch.cuda.synchronize()
        assert len(var.engine()._ws) >= 2                      # one workspace per stream
        for i in range(6):
            assert torch.equal(outs[i], serial[i]), f'call {i} ({prec}) differs when two calls are in flight: max {float((outs[i] - serial[i]).abs().max()):.3e}'
        assert not torch.equal(serial[0], serial[1])
    finally:
        var.set_hip_precision('f32')
