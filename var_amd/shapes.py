"""State-dict layout of the reference's VAR / VQVAE as plain data (name -> shape), SURVEY.md §8(b).

Used by (a) the tests and the bench to build deterministic weights without instantiating modules, (b) a test
that our nn.Modules expose exactly these keys (strict checkpoint compatibility with demo_sample.py:33-34).
Key names and shapes follow reference models/var.py:55-116, basic_var.py:59-149, quant.py:17-42, vqvae.py:32-49,
basic_vae.py:99-208.
"""
from collections import OrderedDict
from typing import Sequence, Tuple

CH_MULT = (1, 1, 2, 2, 4)          # vqvae.py:35
NUM_RES_BLOCKS = 2


def var_shapes(depth: int, patch_nums: Sequence[int], shared_aln: bool = False, attn_l2_norm: bool = True,
               num_classes: int = 1000, V: int = 4096, Cvae: int = 32, embed_dim: int = None, num_heads: int = None,
               mlp_ratio: float = 4.0) -> 'OrderedDict[str, Tuple[int, ...]]':
    C = embed_dim or depth * 64
    H = num_heads or depth
    L = sum(p * p for p in patch_nums)
    first_l = patch_nums[0] ** 2
    hid = round(C * mlp_ratio)
    s = OrderedDict()
    s['pos_start'] = (1, first_l, C)
    s['pos_1LC'] = (1, L, C)
    s['lvl_1L'] = (1, L)
    s['attn_bias_for_masking'] = (1, 1, L, L)
    s['word_embed.weight'] = (C, Cvae)
    s['word_embed.bias'] = (C,)
    s['class_emb.weight'] = (num_classes + 1, C)
    s['lvl_embed.weight'] = (len(patch_nums), C)
    if shared_aln:
        s['shared_ada_lin.1.weight'] = (6 * C, C)
        s['shared_ada_lin.1.bias'] = (6 * C,)
    for b in range(depth):
        p = f'blocks.{b}.'
        if shared_aln:
            s[p + 'ada_gss'] = (1, 1, 6, C)
        if attn_l2_norm:
            s[p + 'attn.scale_mul_1H11'] = (1, H, 1, 1)
        s[p + 'attn.q_bias'] = (C,)
        s[p + 'attn.v_bias'] = (C,)
        s[p + 'attn.zero_k_bias'] = (C,)
        s[p + 'attn.mat_qkv.weight'] = (3 * C, C)
        s[p + 'attn.proj.weight'] = (C, C)
        s[p + 'attn.proj.bias'] = (C,)
        s[p + 'ffn.fc1.weight'] = (hid, C)
        s[p + 'ffn.fc1.bias'] = (hid,)
        s[p + 'ffn.fc2.weight'] = (C, hid)
        s[p + 'ffn.fc2.bias'] = (C,)
        if not shared_aln:
            s[p + 'ada_lin.1.weight'] = (6 * C, C)
            s[p + 'ada_lin.1.bias'] = (6 * C,)
    s['head_nm.ada_lin.1.weight'] = (2 * C, C)
    s['head_nm.ada_lin.1.bias'] = (2 * C,)
    s['head.weight'] = (V, C)
    s['head.bias'] = (V,)
    return s


def _conv(s, name, cout, cin, k):
    s[name + '.weight'] = (cout, cin, k, k)
    s[name + '.bias'] = (cout,)


def _norm(s, name, c):
    s[name + '.weight'] = (c,)
    s[name + '.bias'] = (c,)


def _res(s, name, cin, cout):
    _norm(s, name + '.norm1', cin); _conv(s, name + '.conv1', cout, cin, 3)
    _norm(s, name + '.norm2', cout); _conv(s, name + '.conv2', cout, cout, 3)
    if cin != cout:
        _conv(s, name + '.nin_shortcut', cout, cin, 1)


def _attn(s, name, c):
    _norm(s, name + '.norm', c); _conv(s, name + '.qkv', 3 * c, c, 1); _conv(s, name + '.proj_out', c, c, 1)


def vae_shapes(ch: int = 160, patch_nums: Sequence[int] = (1, 2, 3, 4, 5, 6, 8, 10, 13, 16), V: int = 4096, Cvae: int = 32,
               share_quant_resi: int = 4, include_encoder: bool = True) -> 'OrderedDict[str, Tuple[int, ...]]':
    s = OrderedDict()
    nres = len(CH_MULT)
    if include_encoder:                                   # basic_vae.py:99-160
        _conv(s, 'encoder.conv_in', ch, 3, 3)
        in_mult = (1,) + CH_MULT
        cin = ch
        for lev in range(nres):
            cin, cout = ch * in_mult[lev], ch * CH_MULT[lev]
            for ib in range(NUM_RES_BLOCKS):
                _res(s, f'encoder.down.{lev}.block.{ib}', cin, cout)
                cin = cout
                if lev == nres - 1:
                    _attn(s, f'encoder.down.{lev}.attn.{ib}', cin)
            if lev != nres - 1:
                _conv(s, f'encoder.down.{lev}.downsample.conv', cin, cin, 3)
        _res(s, 'encoder.mid.block_1', cin, cin); _attn(s, 'encoder.mid.attn_1', cin); _res(s, 'encoder.mid.block_2', cin, cin)
        _norm(s, 'encoder.norm_out', cin); _conv(s, 'encoder.conv_out', Cvae, cin, 3)
    # decoder, basic_vae.py:163-208
    cin = ch * CH_MULT[-1]
    _conv(s, 'decoder.conv_in', cin, Cvae, 3)
    _res(s, 'decoder.mid.block_1', cin, cin); _attn(s, 'decoder.mid.attn_1', cin); _res(s, 'decoder.mid.block_2', cin, cin)
    ups = {}
    for lev in reversed(range(nres)):
        cout = ch * CH_MULT[lev]
        d = OrderedDict()
        for ib in range(NUM_RES_BLOCKS + 1):
            _res(d, f'decoder.up.{lev}.block.{ib}', cin, cout)
            cin = cout
        if lev == nres - 1:
            for ib in range(NUM_RES_BLOCKS + 1):
                _attn(d, f'decoder.up.{lev}.attn.{ib}', cin)
        if lev != 0:
            _conv(d, f'decoder.up.{lev}.upsample.conv', cin, cin, 3)
        ups[lev] = d
    for lev in range(nres):                               # ModuleList order: up.0 .. up.4 (insert(0, ...), basic_vae.py:204)
        s.update(ups[lev])
    _norm(s, 'decoder.norm_out', cin); _conv(s, 'decoder.conv_out', 3, cin, 3)
    s['quantize.ema_vocab_hit_SV'] = (len(patch_nums), V)
    for k in range(share_quant_resi):
        _conv(s, f'quantize.quant_resi.qresi_ls.{k}', Cvae, Cvae, 3)
    s['quantize.embedding.weight'] = (V, Cvae)
    if include_encoder:
        _conv(s, 'quant_conv', Cvae, Cvae, 3)
    _conv(s, 'post_quant_conv', Cvae, Cvae, 3)
    return s
