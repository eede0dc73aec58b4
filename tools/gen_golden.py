#!/usr/bin/env python3
"""Generate golden fixtures under tests/golden/ by running the *reference* implementation on CPU.

Runs only in the build container (the reference at /root/reference never travels to the GPU box).
The reference's `models` package is imported unmodified; the one harness-side shim is
`torch.Optional = typing.Optional` (its annotations at models/var.py:241-242 need torch<=2.1 otherwise).

Weights come from var_amd.detinit (hash of seed/name/index), loaded through the reference's own
`load_state_dict(strict=True)`, so (a) the GPU box can rebuild the identical weights without the
reference and (b) a key/shape mismatch between detinit and the reference fails here.

What is recorded per end-to-end case (reference call chain: models/var.py:126-190):
  idx        (B, L) int32      token ids the reference sampled at every scale (helpers.py:6-19)
  img        (B,3,H,W) f32     final decoded image in [0,1] (var.py:190, vqvae.py:62-63)
  f_hat_s{si}(B,Cvae,P,P) f32  accumulated feature map after scale si (quant.py:187-196)
  pool_s{si} (B,Cvae,p',p')    area-pooled input of the next scale (quant.py:192)
  logits_s{si}                 raw 2B-row logits before CFG (var.py:170): all rows for `full_logits`
                               cases, else rows t in {0, l-1}
  lsum/labs  (S,) f64          sum / abs-sum of each scale's raw logits (checksum for the big cases)
  noise_head (S,8) f32, noise_sum (S,) f64   head/checksum of the Exp(1) fill torch.multinomial draws
                               (one (B*l,V) fill per scale), so tests can regenerate it with a CPU
                               torch.Generator and prove they got the same stream.
"""
import argparse
import json
import os
import sys
import time
import typing

import numpy as np
import torch

torch.Optional = typing.Optional          # shim, see module docstring
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, '/root/reference')
sys.path.insert(1, REPO)

from models import build_vae_var            # noqa: E402  (the REFERENCE package)
from var_amd.detinit import fill_module_    # noqa: E402

GOLD = os.path.join(REPO, 'tests', 'golden')

CASES = {
    # name: dict(depth, ch, patch_nums, attn_l2_norm, shared_aln, B, labels, seed, cfg, top_k, top_p, full_logits)
    't_pn123_base':  dict(depth=2, ch=32, patch_nums=(1, 2, 3), attn_l2_norm=True, shared_aln=False, labels=(3, 7), seed=0, cfg=1.5, top_k=900, top_p=0.96, full_logits=True),
    't_pn12345':     dict(depth=2, ch=32, patch_nums=(1, 2, 3, 4, 5), attn_l2_norm=True, shared_aln=False, labels=(980, 437), seed=1, cfg=1.5, top_k=900, top_p=0.96, full_logits=False),
    't_nol2':        dict(depth=2, ch=32, patch_nums=(1, 2, 3), attn_l2_norm=False, shared_aln=False, labels=(22, 562), seed=2, cfg=1.5, top_k=900, top_p=0.96, full_logits=True),
    't_saln':        dict(depth=2, ch=32, patch_nums=(1, 2, 3), attn_l2_norm=True, shared_aln=True, labels=(1, 999), seed=3, cfg=1.5, top_k=900, top_p=0.96, full_logits=True),
    't_greedy':      dict(depth=2, ch=32, patch_nums=(1, 2, 3), attn_l2_norm=True, shared_aln=False, labels=(5, 6), seed=4, cfg=4.0, top_k=1, top_p=0.0, full_logits=False),
    't_nofilter':    dict(depth=2, ch=32, patch_nums=(1, 2, 3), attn_l2_norm=True, shared_aln=False, labels=(100, 200), seed=5, cfg=0.0, top_k=0, top_p=0.0, full_logits=False),
    't_b3_pn1234':   dict(depth=2, ch=32, patch_nums=(1, 2, 3, 4), attn_l2_norm=True, shared_aln=False, labels=(0, 500, 999), seed=123, cfg=3.0, top_k=600, top_p=0.5, full_logits=False),
    'd16_pn123':     dict(depth=16, ch=160, patch_nums=(1, 2, 3), attn_l2_norm=True, shared_aln=False, labels=(3, 7), seed=0, cfg=1.5, top_k=900, top_p=0.96, full_logits=False),
    # the widths of BASELINE.json configs[3] / configs[4]: VAR-d30 (C=1920, 30 heads, 2.0 B parameters) on the first three scales and
    # VAR-d36 (C=2304, 36 heads, shared AdaLN, 2.3 B parameters) on the first five scales of the 512-pixel schedule (utils/arg_util.py:244-251)
    'd30_pn123':     dict(depth=30, ch=160, patch_nums=(1, 2, 3), attn_l2_norm=True, shared_aln=False, labels=(3, 7), seed=0, cfg=1.5, top_k=900, top_p=0.96, full_logits=False),
    'd36_saln_pn12346': dict(depth=36, ch=160, patch_nums=(1, 2, 3, 4, 6), attn_l2_norm=True, shared_aln=True, labels=(980, 437), seed=1, cfg=1.5, top_k=900, top_p=0.96, full_logits=False),
    'd16_full':      dict(depth=16, ch=160, patch_nums=(1, 2, 3, 4, 5, 6, 8, 10, 13, 16), attn_l2_norm=True, shared_aln=False, labels=(0, 7), seed=0, cfg=1.5, top_k=900, top_p=0.96, full_logits=False),
}


def build_reference(cfg, init_seed=0):
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        vae, var = build_vae_var(device='cpu', patch_nums=tuple(cfg['patch_nums']), depth=cfg['depth'], ch=cfg['ch'],
                                 shared_aln=cfg['shared_aln'], attn_l2_norm=cfg['attn_l2_norm'])
    fill_module_(var, cfg['depth'], init_seed, 'var.')
    fill_module_(vae, cfg['depth'], init_seed, 'vae.')
    # round-trip through the reference's strict loader: proves key/shape compatibility of detinit
    var.load_state_dict({k: v.clone() for k, v in var.state_dict().items()}, strict=True)
    vae.load_state_dict({k: v.clone() for k, v in vae.state_dict().items()}, strict=True)
    return vae.eval(), var.eval()


def run_case(name, cfg, autocast16=False):
    """autocast16: the reference's own 16-bit execution — the call wrapped in torch.autocast(dtype=float16) exactly as demo_sample.py:66-68 does
    (device 'cpu' here: F.linear / conv2d / SDPA run on fp16 operands with fp16 results, LayerNorm / softmax / get_logits' .float() in fp32).
    autocast16='bf16': the same under torch.autocast(dtype=bfloat16), the reference's other 16-bit option (utils/arg_util.py `fp16: int  # 1: using fp16, 2: bf16`).
    These fixtures pin the 16-bit throughput mode (and its CPU twin) against the reference itself, with a tolerance."""
    vae, var = build_reference(cfg)
    pns = tuple(cfg['patch_nums'])
    B = len(cfg['labels'])
    rec = {}
    logits, idxs, fhats, pools = [], [], [], []

    orig_get_logits = var.get_logits
    def get_logits(h, c):
        r = orig_get_logits(h, c)
        logits.append(r.detach().clone())
        return r
    var.get_logits = get_logits

    quant = vae.quantize
    hk = quant.embedding.register_forward_hook(lambda m, inp, out: idxs.append(inp[0].detach().clone()))
    orig_next = quant.get_next_autoregressive_input
    def get_next(si, SN, f_hat, h):
        f, nxt = orig_next(si, SN, f_hat, h)
        fhats.append(f.detach().clone())
        pools.append(nxt.detach().clone())
        return f, nxt
    quant.get_next_autoregressive_input = get_next

    t0 = time.time()
    import contextlib
    with torch.inference_mode(), (torch.autocast('cpu', dtype=torch.bfloat16 if autocast16 == 'bf16' else torch.float16) if autocast16 else contextlib.nullcontext()):
        img = var.autoregressive_infer_cfg(B, torch.tensor(cfg['labels'], dtype=torch.long), g_seed=cfg['seed'], cfg=cfg['cfg'],
                                           top_k=cfg['top_k'], top_p=cfg['top_p'], more_smooth=False)
    dt = time.time() - t0
    hk.remove()
    logits[:] = [x.float() for x in logits]; fhats[:] = [x.float() for x in fhats]; pools[:] = [x.float() for x in pools]

    rec['img'] = img.float().numpy().astype(np.float32)
    rec['idx'] = torch.cat(idxs, dim=1).numpy().astype(np.int32)
    lsum, labs = [], []
    for si, pn in enumerate(pns):
        lg = logits[si].numpy()
        lsum.append(lg.astype(np.float64).sum()); labs.append(np.abs(lg.astype(np.float64)).sum())
        rec[f'logits_s{si}'] = lg if cfg['full_logits'] else lg[:, sorted({0, pn * pn - 1}), :]
        rec[f'f_hat_s{si}'] = fhats[si].numpy().copy()
        if si < len(pns) - 1:
            rec[f'pool_s{si}'] = pools[si].numpy().copy()
    rec['lsum'] = np.array(lsum); rec['labs'] = np.array(labs)
    # the Exp(1) noise stream the reference consumed (torch.multinomial, n=1: one exponential_ fill per scale)
    g = torch.Generator(device='cpu'); g.manual_seed(cfg['seed'])
    V = logits[0].shape[-1]
    nh, ns = [], []
    for pn in pns:
        q = torch.empty(B * pn * pn, V, dtype=torch.float32).exponential_(1, generator=g)
        nh.append(q.view(-1)[:8].numpy().copy()); ns.append(q.double().sum().item())
    rec['noise_head'] = np.stack(nh); rec['noise_sum'] = np.array(ns)
    meta = dict(cfg); meta['B'] = B; meta['V'] = int(V); meta['ref_seconds'] = dt; meta['torch'] = torch.__version__
    meta['threads'] = torch.get_num_threads()
    meta['autocast16'] = bool(autocast16)
    meta['autocast_dtype'] = ('bf16' if autocast16 == 'bf16' else 'f16') if autocast16 else None
    rec['meta'] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(GOLD, f'e2e_{name}.npz'), **rec)
    print(f'[gen_golden] {name}: {dt:.2f}s  img mean {img.mean():.4f}  tokens {rec["idx"].shape}', flush=True)


def run_inpaint():
    """VAR.inpainting fixtures (fork, reference models/var.py:236-364): tokens of an earlier AR run are kept where mask is True.
    Scales 0-1 fully kept (the reference then skips sampling AND the RNG draw), scale 2 mixed, scale 3 fully resampled, scale 4 mixed."""
    for name, base in (('inpaint_t_pn12345', 't_pn12345'), ('inpaint_d16_pn123', 'd16_pn123')):
        cfg = dict(CASES[base])
        vae, var = build_reference(cfg)
        z = np.load(os.path.join(GOLD, f'e2e_{base}.npz'))
        gt = torch.from_numpy(z['idx'].astype(np.int64))
        B, L = gt.shape
        g = torch.Generator(); g.manual_seed(11)
        mask = torch.rand(B, L, generator=g) < 0.5
        pns = cfg['patch_nums']
        ends = np.cumsum([p * p for p in pns])
        begs = np.concatenate([[0], ends[:-1]])
        mask[:, begs[0]:ends[0]] = True
        if len(pns) > 3:
            mask[:, begs[1]:ends[1]] = True
            mask[:, begs[3]:ends[3]] = False
        finals, fhats = [], []
        hk = vae.quantize.embedding.register_forward_hook(lambda m, inp, out: finals.append(inp[0].detach().clone()))
        orig_next = vae.quantize.get_next_autoregressive_input
        def get_next(si, SN, f_hat, h):
            f, nxt = orig_next(si, SN, f_hat, h); fhats.append(f.detach().clone()); return f, nxt
        vae.quantize.get_next_autoregressive_input = get_next
        seed = 21
        with torch.inference_mode():
            img = var.inpainting(gt, mask, label=torch.tensor(cfg['labels']), g_seed=seed, cfg=cfg['cfg'], top_k=cfg['top_k'], top_p=cfg['top_p'])
        hk.remove()
        rec = dict(gt=gt.numpy().astype(np.int32), mask=mask.numpy(), img=img.numpy(), idx=torch.cat(finals, 1).numpy().astype(np.int32),
                   f_hat=fhats[-1].numpy())
        # the Exp(1) fills actually consumed: one per scale that is not fully kept, in order
        gg = torch.Generator(); gg.manual_seed(seed)
        nh, ns, drawn = [], [], []
        for si, pn in enumerate(pns):
            if bool(mask[:, begs[si]:ends[si]].all()): continue
            q = torch.empty(B * pn * pn, 4096).exponential_(1, generator=gg)
            nh.append(q.view(-1)[:8].numpy().copy()); ns.append(q.double().sum().item()); drawn.append(si)
        meta = dict(cfg); meta.update(B=B, V=4096, seed=seed, base=base, drawn_scales=drawn)
        rec.update(noise_head=np.stack(nh), noise_sum=np.array(ns), meta=np.array(json.dumps(meta)))
        np.savez_compressed(os.path.join(GOLD, f'{name}.npz'), **rec)
        print(f'[gen_golden] {name}: kept {int(mask.sum())}/{mask.numel()} tokens, {int((rec["idx"] != rec["gt"]).sum())} resampled differently', flush=True)


def run_inpaint_more_smooth():
    """VAR.inpainting(more_smooth=True) (fork, var.py:332-341) with no fully kept scale: the embeddings come from the gumbel softmax of the
    filtered logits alone (the kept tokens only reach `final_tokens`, which that branch does not read); two Exp(1) fills per scale."""
    cfg = dict(CASES['t_pn12345']); seed = 13
    vae, var = build_reference(cfg)
    z = np.load(os.path.join(GOLD, 'e2e_t_pn12345.npz'))
    gt = torch.from_numpy(z['idx'].astype(np.int64))
    B, L = gt.shape
    g = torch.Generator(); g.manual_seed(12)
    mask = torch.rand(B, L, generator=g) < 0.5
    mask[:, 0] = False                                     # scale 0 has one token per image: never fully kept
    fhats = []
    orig_next = vae.quantize.get_next_autoregressive_input
    def get_next(si, SN, f_hat, h):
        f, nxt = orig_next(si, SN, f_hat, h); fhats.append(f.detach().clone()); return f, nxt
    vae.quantize.get_next_autoregressive_input = get_next
    with torch.inference_mode():
        img = var.inpainting(gt, mask, label=torch.tensor(cfg['labels']), g_seed=seed, cfg=cfg['cfg'], top_k=cfg['top_k'], top_p=cfg['top_p'], more_smooth=True)
    gg = torch.Generator(); gg.manual_seed(seed)
    heads = []
    for pn in cfg['patch_nums']:
        a = torch.empty(B * pn * pn, 4096).exponential_(1, generator=gg); b = torch.empty(B, pn * pn, 4096).exponential_(generator=gg)
        heads.append(np.concatenate([a.view(-1)[:4].numpy(), b.view(-1)[:4].numpy()]))
    meta = dict(cfg); meta.update(B=B, V=4096, seed=seed)
    np.savez_compressed(os.path.join(GOLD, 'inpaint_ms_t_pn12345.npz'), gt=gt.numpy().astype(np.int32), mask=mask.numpy(), img=img.numpy(),
                        f_hat=fhats[-1].numpy(), noise_head=np.stack(heads), meta=np.array(json.dumps(meta)))
    print(f'[gen_golden] inpaint_ms_t_pn12345: img mean {img.mean():.4f}', flush=True)


def run_nearest_code_cos():
    """A17 with using_znorm=True (reference models/quant.py:151-153): cosine arg-max instead of the L2 arg-min"""
    import contextlib, io
    from models import VQVAE
    cfg = CASES['t_pn12345']
    pns = tuple(cfg['patch_nums'])
    with contextlib.redirect_stdout(io.StringIO()):
        vae = VQVAE(vocab_size=4096, z_channels=32, ch=cfg['ch'], using_znorm=True, test_mode=True, share_quant_resi=4, v_patch_nums=pns)
    fill_module_(vae, cfg['depth'], 0, 'vae.')
    vae.eval()
    P = pns[-1]
    g = torch.Generator(); g.manual_seed(78)
    f = torch.randn(3, 32, P, P, generator=g) * 1.5
    with torch.inference_mode():
        idx = vae.quantize.f_to_idxBl_or_fhat(f, to_fhat=False)
        fh = vae.quantize.f_to_idxBl_or_fhat(f, to_fhat=True)
    rec = dict(f=f.numpy(), meta=np.array(json.dumps(dict(cfg))))
    for si, (i, h) in enumerate(zip(idx, fh)):
        rec[f'idx_s{si}'] = i.numpy().astype(np.int32); rec[f'f_hat_s{si}'] = h.numpy()
    # how decisive the arg-max is: margin between the best and second-best cosine of the first scale's queries
    zq = torch.nn.functional.normalize(f.mean(dim=(2, 3)), dim=-1) @ torch.nn.functional.normalize(vae.quantize.embedding.weight.data.T, dim=0)
    top2 = zq.topk(2, dim=1).values
    rec['margin_s0'] = (top2[:, 0] - top2[:, 1]).numpy()
    np.savez_compressed(os.path.join(GOLD, 'nearest_code_cos.npz'), **rec)
    print('[gen_golden] nearest_code_cos done; s0 margins', rec['margin_s0'], flush=True)


def run_more_smooth():
    """autoregressive_infer_cfg(more_smooth=True) (var.py:178-180, helpers.py:22-36): the embeddings come from a gumbel-softmax
    over the FILTERED logits (sample_with_top_k_top_p_ mutates them in place) times the codebook; two Exp(1) fills per scale."""
    cfg = dict(CASES['t_pn12345']); cfg['seed'] = 9
    vae, var = build_reference(cfg)
    B = len(cfg['labels'])
    hs, fhats = [], []
    orig_next = vae.quantize.get_next_autoregressive_input
    def get_next(si, SN, f_hat, h):
        hs.append(h.detach().clone()); f, nxt = orig_next(si, SN, f_hat, h); fhats.append(f.detach().clone()); return f, nxt
    vae.quantize.get_next_autoregressive_input = get_next
    with torch.inference_mode():
        img = var.autoregressive_infer_cfg(B, torch.tensor(cfg['labels']), g_seed=cfg['seed'], cfg=cfg['cfg'], top_k=cfg['top_k'], top_p=cfg['top_p'], more_smooth=True)
    rec = dict(img=img.numpy(), f_hat=fhats[-1].numpy())
    for si, h in enumerate(hs): rec[f'h_s{si}'] = h.numpy()            # (B, Cvae, pn, pn)
    g = torch.Generator(); g.manual_seed(cfg['seed'])
    heads = []
    for pn in cfg['patch_nums']:
        a = torch.empty(B * pn * pn, 4096).exponential_(1, generator=g); b = torch.empty(B, pn * pn, 4096).exponential_(generator=g)
        heads.append(np.concatenate([a.view(-1)[:4].numpy(), b.view(-1)[:4].numpy()]))
    meta = dict(cfg); meta.update(B=B, V=4096)
    rec.update(noise_head=np.stack(heads), meta=np.array(json.dumps(meta)))
    np.savez_compressed(os.path.join(GOLD, 'more_smooth_t_pn12345.npz'), **rec)
    print(f'[gen_golden] more_smooth_t_pn12345: img mean {img.mean():.4f}', flush=True)


def run_smooth_sampling():
    """VAR.smooth_sampling fixtures (fork, reference models/var.py:367-572): every position takes, among the nearest codebook
    neighbours of its ground-truth token, the one with the highest CFG log-probability.  Three variants on the tiny config
    (candidate-count mode, threshold mode, candidate-count + more_smooth) and candidate-count mode on d16 (1,2,3)."""
    for name, base, n, thr_rank, smooth in (('smooth_t_pn12345_count', 't_pn12345', 6, None, False), ('smooth_t_pn12345_thr', 't_pn12345', 8, 4, False),
                                            ('smooth_t_pn12345_count_ms', 't_pn12345', 5, None, True), ('smooth_d16_pn123_count', 'd16_pn123', 6, None, False)):
        cfg = dict(CASES[base])
        vae, var = build_reference(cfg)
        z = np.load(os.path.join(GOLD, f'e2e_{base}.npz'))
        gt = torch.from_numpy(z['idx'].astype(np.int64))
        B, L = gt.shape
        gq = torch.Generator(); gq.manual_seed(77)
        rnd_tok = torch.randint(0, 4096, (B, L), generator=gq)
        gt = torch.where(torch.rand(B, L, generator=gq) < 0.7, rnd_tok, gt)     # mostly random ground truth: the model disagrees with it often
        emb = vae.quantize.embedding.weight.detach()
        d = torch.cdist(emb, emb, p=2)
        srt = torch.sort(d, dim=1).values
        thr = None
        if thr_rank is not None:                       # a threshold that cuts between the thr_rank-th and the next neighbour for a typical token
            thr = float(0.5 * (srt[:, thr_rank].median() + srt[:, thr_rank + 1].median()))
        finals, fhats = [], []
        hk = vae.quantize.embedding.register_forward_hook(lambda m, inp, out: finals.append(inp[0].detach().clone()))
        orig_next = vae.quantize.get_next_autoregressive_input
        def get_next(si, SN, f_hat, h):
            f, nxt = orig_next(si, SN, f_hat, h); fhats.append(f.detach().clone()); return f, nxt
        vae.quantize.get_next_autoregressive_input = get_next
        seed = 31
        with torch.inference_mode():
            img, sll, sdl = var.smooth_sampling(gt, n, label=torch.tensor(cfg['labels']), g_seed=seed, cfg=cfg['cfg'], more_smooth=smooth, neighbor_threshold=thr)
        hk.remove()
        rec = dict(gt=gt.numpy().astype(np.int32), img=img.numpy(), f_hat=fhats[-1].numpy(), sum_ll=np.float64(float(sll)), sum_dist_ll=np.float64(float(sdl)),
                   self_dist_max=np.float64(float(d.diagonal().max())), gap_min=np.float64(float((srt[:, 1:n + 1] - srt[:, 0:n]).min())))
        if not smooth: rec['idx'] = torch.cat(finals, 1).numpy().astype(np.int32)        # with more_smooth the embedding lookup is bypassed
        if smooth:
            g = torch.Generator(); g.manual_seed(seed)
            heads = []
            for pn in cfg['patch_nums']:
                a = torch.empty(B, pn * pn, 4096).exponential_(generator=g); heads.append(a.view(-1)[:4].numpy().copy())
            rec['noise_head'] = np.stack(heads)
        meta = dict(cfg); meta.update(B=B, V=4096, seed=seed, base=base, n=n, thr=thr, more_smooth=smooth, sum_ll_dtype=str(sll.dtype) if torch.is_tensor(sll) else 'float')
        rec['meta'] = np.array(json.dumps(meta))
        np.savez_compressed(os.path.join(GOLD, f'{name}.npz'), **rec)
        moved = int((rec['idx'] != rec['gt']).sum()) if 'idx' in rec else -1
        print(f'[gen_golden] {name}: n={n} thr={thr} sum_ll={float(sll):.3f} sum_dist_ll={float(sdl):.4f} moved {moved}/{gt.numel()} '
              f'self-dist max {float(d.diagonal().max()):.2e} min neighbour gap {float(rec["gap_min"]):.2e}', flush=True)


def run_encode():
    """encode side + teacher forcing: VQVAE.img_to_post / img_to_idxBl (vqvae.py:65-75), quantize.idxBl_to_var_input
    (quant.py:169-184) and VAR.forward (var.py:192-234, cond_drop_rate 0) on a seeded random image"""
    cfg = dict(CASES['t_pn12345'])
    vae, var = build_reference(cfg)
    var.cond_drop_rate = 0.0
    P = cfg['patch_nums'][-1]
    g = torch.Generator(); g.manual_seed(31)
    img = torch.rand(2, 3, 16 * P, 16 * P, generator=g) * 2 - 1
    lab = torch.tensor(cfg['labels'])
    with torch.inference_mode():
        f = vae.img_to_post(img)
        idx = vae.img_to_idxBl(img)
        fh = vae.img_to_fhat(img)
        x_in = vae.quantize.idxBl_to_var_input(idx)
        logits = var(lab, x_in)
    rec = dict(img=img.numpy(), f=f.numpy(), var_input=x_in.numpy(), logits=logits.numpy(), f_hat_last=fh[-1].numpy(),
               meta=np.array(json.dumps(dict(cfg))))
    for si, i in enumerate(idx): rec[f'idx_s{si}'] = i.numpy().astype(np.int32)
    np.savez_compressed(os.path.join(GOLD, 'encode_t_pn12345.npz'), **rec)
    print(f'[gen_golden] encode_t_pn12345: f std {f.std():.3f}, logits std {logits.std():.3f}', flush=True)


def run_nearest_code():
    """A17 fixture: VectorQuantizer2.f_to_idxBl_or_fhat (reference models/quant.py:135-166) on a random feature map."""
    cfg = CASES['t_pn12345']
    vae, _ = build_reference(cfg)
    P = cfg['patch_nums'][-1]
    g = torch.Generator(); g.manual_seed(77)
    f = torch.randn(3, 32, P, P, generator=g) * 1.5
    with torch.inference_mode():
        idx = vae.quantize.f_to_idxBl_or_fhat(f, to_fhat=False)
        fh = vae.quantize.f_to_idxBl_or_fhat(f, to_fhat=True)
        var_in = vae.quantize.idxBl_to_var_input(idx)
    rec = dict(f=f.numpy(), var_input=var_in.numpy(), meta=np.array(json.dumps(dict(cfg))))
    for si, (i, h) in enumerate(zip(idx, fh)):
        rec[f'idx_s{si}'] = i.numpy().astype(np.int32); rec[f'f_hat_s{si}'] = h.numpy()
    np.savez_compressed(os.path.join(GOLD, 'nearest_code.npz'), **rec)
    print('[gen_golden] nearest_code done', flush=True)


def run_sampler_vectors():
    """A8 fixture: sample_with_top_k_top_p_ (reference models/helpers.py:6-19) on synthetic logits incl. ties and edge settings."""
    from models.helpers import sample_with_top_k_top_p_
    g = torch.Generator(); g.manual_seed(2024)
    rec = {}
    cases = []
    for ci, (B, l, V, top_k, top_p, kind) in enumerate([
        (2, 5, 4096, 900, 0.96, 'normal'), (2, 5, 4096, 0, 0.0, 'normal'), (2, 5, 4096, 1, 0.0, 'normal'),
        (2, 5, 4096, 0, 0.5, 'normal'), (2, 5, 4096, 50, 0.999, 'peaked'), (2, 5, 4096, 900, 0.96, 'ties'),
        (1, 3, 4096, 4096, 0.0001, 'normal'), (2, 4, 512, 100, 0.9, 'normal'),
    ]):
        if kind == 'normal': lg = torch.randn(B, l, V, generator=g) * 2.5
        elif kind == 'peaked': lg = torch.randn(B, l, V, generator=g) * 8.0
        else: lg = (torch.randn(B, l, V, generator=g) * 2.0).round()       # many exact ties
        gs = torch.Generator(); gs.manual_seed(1000 + ci)
        noise = torch.empty(B * l, V).exponential_(1, generator=torch.Generator().manual_seed(1000 + ci))
        work = lg.clone()
        idx = sample_with_top_k_top_p_(work, rng=gs, top_k=top_k, top_p=top_p, num_samples=1)[:, :, 0]
        rec[f'logits_{ci}'] = lg.numpy(); rec[f'noise_{ci}'] = noise.numpy(); rec[f'idx_{ci}'] = idx.numpy().astype(np.int32)
        rec[f'kept_{ci}'] = torch.isfinite(work).numpy()
        cases.append(dict(B=B, l=l, V=V, top_k=top_k, top_p=top_p, kind=kind))
    rec['meta'] = np.array(json.dumps(cases))
    np.savez_compressed(os.path.join(GOLD, 'sampler.npz'), **rec)
    print('[gen_golden] sampler done', flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', nargs='*', default=None)
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    for name, cfg in CASES.items():
        if args.only and name not in args.only: continue
        run_case(name, cfg)
    for base in ('t_pn12345', 't_saln', 'd16_pn123'):                 # the reference under its harness' fp16 autocast (demo_sample.py:66-68)
        if not args.only or ('ac16_' + base) in args.only: run_case('ac16_' + base, CASES[base], autocast16=True)
    for base in ('t_pn12345', 'd16_pn123'):                           # ... and under bfloat16 autocast
        if not args.only or ('acbf16_' + base) in args.only: run_case('acbf16_' + base, CASES[base], autocast16='bf16')
    if not args.only or 'inpaint' in args.only: run_inpaint()
    if not args.only or 'more_smooth' in args.only: run_more_smooth()
    if not args.only or 'inpaint_more_smooth' in args.only: run_inpaint_more_smooth()
    if not args.only or 'nearest_code_cos' in args.only: run_nearest_code_cos()
    if not args.only or 'smooth_sampling' in args.only: run_smooth_sampling()
    if not args.only or 'encode' in args.only: run_encode()
    if not args.only or 'nearest_code' in args.only: run_nearest_code()
    if not args.only or 'sampler' in args.only: run_sampler_vectors()


if __name__ == '__main__':
    main()
