"""VAR transformer with the reference's module API (models/var.py:21-190, 577-653).

`autoregressive_infer_cfg` — the hot path — is a thin shell: argument handling as in the reference, then
var_amd.engine.SamplingEngine, which runs the whole loop on HIP kernels for gfx950.  There is deliberately no PyTorch
implementation of that loop in this file: on a machine without the HIP library or a GPU the call fails loudly.
`forward` (teacher forcing, autograd) is plain PyTorch for trainer.py / likelihood scripts."""
import math
from functools import partial
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from .. import dist
from .basic_var import AdaLNBeforeHead, AdaLNSelfAttn
from .helpers import gumbel_softmax_with_rng, sample_with_top_k_top_p_     # noqa: F401  (names the notebooks import from here)
from .vqvae import VQVAE, VectorQuantizer2


class SharedAdaLin(nn.Linear):
    def forward(self, cond_BD):
        return super().forward(cond_BD).view(-1, 1, 6, self.weight.shape[0] // 6)


class VAR(nn.Module):
    def __init__(self, vae_local: VQVAE, num_classes=1000, depth=16, embed_dim=1024, num_heads=16, mlp_ratio=4., drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0., norm_eps=1e-6, shared_aln=False, cond_drop_rate=0.1, attn_l2_norm=False,
                 patch_nums=(1, 2, 3, 4, 5, 6, 8, 10, 13, 16), flash_if_available=True, fused_if_available=True):
        super().__init__()
        assert embed_dim % num_heads == 0
        self.Cvae, self.V = vae_local.Cvae, vae_local.vocab_size
        self.depth, self.C, self.D, self.num_heads = depth, embed_dim, embed_dim, num_heads
        self.cond_drop_rate, self.prog_si, self.norm_eps, self.shared_aln = cond_drop_rate, -1, norm_eps, shared_aln
        self.patch_nums: Tuple[int] = tuple(patch_nums)
        sizes = [pn * pn for pn in self.patch_nums]
        self.L, self.first_l = sum(sizes), sizes[0]
        ends = torch.tensor(sizes).cumsum(0).tolist()
        self.begin_ends = list(zip([0] + ends[:-1], ends))
        self.num_stages_minus_1 = len(self.patch_nums) - 1
        self.rng = torch.Generator(device=dist.get_device())

        self.vae_proxy: Tuple[VQVAE] = (vae_local,)                          # tuples: the VAE is not a sub-module (not in our state-dict)
        self.vae_quant_proxy: Tuple[VectorQuantizer2] = (vae_local.quantize,)
        self.word_embed = nn.Linear(self.Cvae, self.C)

        std = math.sqrt(1 / self.C / 3)
        tn = lambda *shape: nn.init.trunc_normal_(torch.empty(*shape), mean=0, std=std)
        self.num_classes = num_classes
        self.uniform_prob = torch.full((1, num_classes), fill_value=1.0 / num_classes, dtype=torch.float32, device=dist.get_device())
        self.class_emb = nn.Embedding(num_classes + 1, self.C)
        nn.init.trunc_normal_(self.class_emb.weight.data, mean=0, std=std)
        self.pos_start = nn.Parameter(tn(1, self.first_l, self.C))
        self.pos_1LC = nn.Parameter(torch.cat([tn(1, n, self.C) for n in sizes], dim=1))
        self.lvl_embed = nn.Embedding(len(self.patch_nums), self.C)
        nn.init.trunc_normal_(self.lvl_embed.weight.data, mean=0, std=std)

        self.shared_ada_lin = nn.Sequential(nn.SiLU(inplace=False), SharedAdaLin(self.D, 6 * self.C)) if shared_aln else nn.Identity()
        norm_layer = partial(nn.LayerNorm, eps=norm_eps)
        self.drop_path_rate = drop_path_rate
        dpr = torch.linspace(0, drop_path_rate, depth).tolist()
        self.blocks = nn.ModuleList(
            AdaLNSelfAttn(cond_dim=self.D, shared_aln=shared_aln, block_idx=i, embed_dim=self.C, norm_layer=norm_layer, num_heads=num_heads,
                          mlp_ratio=mlp_ratio, drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[i], last_drop_p=0 if i == 0 else dpr[i - 1],
                          attn_l2_norm=attn_l2_norm, flash_if_available=flash_if_available, fused_if_available=fused_if_available)
            for i in range(depth))
        self.using_fused_add_norm_fn = False
        print(f'\n[constructor]  ==== MI355X HIP sampling path (var_amd); VAR config: embed_dim={embed_dim}, num_heads={num_heads}, depth={depth}, '
              f'mlp_ratio={mlp_ratio}, drop_path_rate={drop_path_rate:g}, patch_nums={self.patch_nums} ====\n', flush=True)

        lvl = torch.cat([torch.full((n,), i) for i, n in enumerate(sizes)]).view(1, self.L, 1)
        self.register_buffer('lvl_1L', lvl.transpose(1, 2)[:, 0].contiguous())
        self.register_buffer('attn_bias_for_masking', torch.where(lvl >= lvl.transpose(1, 2), 0., -torch.inf).reshape(1, 1, self.L, self.L).contiguous())

        self.head_nm = AdaLNBeforeHead(self.C, self.D, norm_layer=norm_layer)
        self.head = nn.Linear(self.C, self.V)
        self._engine = None

    # ---- HIP sampling path ------------------------------------------------------------------------------------------
    def engine(self):
        if self._engine is None:
            from ..engine import SamplingEngine
            self._engine = SamplingEngine(self)
        return self._engine

    def set_hip_precision(self, precision: str = 'f32'):
        """'f32' (default: tokens bit-identical to the CPU oracle, pixels within 1e-3 of the reference), 'f16' / 'bf16': 16-bit weights / GEMM
        operands / KV cache with fp32 accumulation — the arithmetic the reference's harness requests through
        torch.autocast('cuda', dtype=torch.float16) (demo_sample.py:66-68) — selected explicitly (an enclosing autocast context does not
        change the result of any of the three), or 'auto': every call follows the caller's autocast state the way the reference does
        (basic_var.py:97 branches on the dtype autocast hands it): inside torch.autocast('cuda', dtype=float16 | bfloat16) the call runs the
        matching 16-bit mode, outside it (or with enabled=False) f32.  With 'auto' demo_sample.py runs its fp16 path unchanged.
        The environment variable VARHIP_FOLLOW_AUTOCAST=1 makes 'auto' the initial policy of every new engine (default: 'f32')."""
        self.engine().set_precision(precision)
        return self

    @torch.no_grad()
    def autoregressive_infer_cfg(self, B: int, label_B: Optional[Union[int, torch.LongTensor]], g_seed: Optional[int] = None, cfg=1.5,
                                 top_k=0, top_p=0.0, more_smooth=False) -> torch.Tensor:
        """Sample B images; returns (B, 3, H, W) in [0, 1].  Same arguments and RNG consumption as the reference
        (var.py:126-190): `g_seed` seeds self.rng, one Exp(1) fill of shape (B*l, V) is drawn per scale."""
        dev = self.lvl_1L.device
        if dev.type != 'cuda':
            raise RuntimeError('VAR.autoregressive_infer_cfg: this build runs the sampling loop on MI355X HIP kernels only; move the model to a '
                               'CUDA/ROCm device (there is no CPU fallback by design)')
        if g_seed is None: rng = None
        else: self.rng.manual_seed(g_seed); rng = self.rng
        if label_B is None:
            label_B = torch.multinomial(self.uniform_prob, num_samples=B, replacement=True, generator=rng).reshape(B)
        elif isinstance(label_B, int):
            label_B = torch.full((B,), fill_value=self.num_classes if label_B < 0 else label_B, device=dev)
        return self.engine().sample(B, label_B.to(dev).long(), rng, cfg, top_k, top_p, more_smooth=bool(more_smooth))

    # ---- teacher-forced forward (PyTorch; reference var.py:118-124,192-234) ---------------------------------------------
    def get_logits(self, h_or_h_and_residual, cond_BD: Optional[torch.Tensor]):
        if not isinstance(h_or_h_and_residual, torch.Tensor):
            h, resi = h_or_h_and_residual
            h_or_h_and_residual = resi + self.blocks[-1].drop_path(h)
        return self.head(self.head_nm(h_or_h_and_residual.float(), cond_BD).float()).float()

    def forward(self, label_B: torch.LongTensor, x_BLCv_wo_first_l: torch.Tensor) -> torch.Tensor:
        """logits (B, L, V) for teacher-forced inputs (B, L-first_l, Cvae); block-causal mask instead of a KV cache"""
        if (not torch.is_grad_enabled() and not self.training and self.prog_si < 0 and self.lvl_1L.is_cuda and self.head.weight.dtype == torch.float32
                and self.C == 64 * self.num_heads and x_BLCv_wo_first_l is not None and x_BLCv_wo_first_l.shape[1] == self.L - self.first_l):
            # (train mode keeps the PyTorch branch below: DropPath / dropout are live there, reference helpers.py:39-59)
            # inference (no autograd): scale-by-scale over the KV cache on the HIP kernels, fp32, same label dropping as below
            label_B = torch.where(torch.rand(label_B.shape[0], device=label_B.device) < self.cond_drop_rate, self.num_classes, label_B)
            return self.engine().teacher_forced_logits(label_B, x_BLCv_wo_first_l)
        bg, ed = self.begin_ends[self.prog_si] if self.prog_si >= 0 else (0, self.L)
        B = x_BLCv_wo_first_l.shape[0]
        with torch.autocast(device_type=x_BLCv_wo_first_l.device.type, enabled=False):
            label_B = torch.where(torch.rand(B, device=label_B.device) < self.cond_drop_rate, self.num_classes, label_B)
            cond_BD = self.class_emb(label_B)
            sos = cond_BD.unsqueeze(1).expand(B, self.first_l, -1) + self.pos_start.expand(B, self.first_l, -1)
            x = sos if self.prog_si == 0 else torch.cat((sos, self.word_embed(x_BLCv_wo_first_l.float())), dim=1)
            x = x + self.lvl_embed(self.lvl_1L[:, :ed].expand(B, -1)) + self.pos_1LC[:, :ed]
        mask = self.attn_bias_for_masking[:, :, :ed, :ed]
        cond_or_gss = self.shared_ada_lin(cond_BD)
        main_type = torch.matmul(x.new_ones(8, 8), x.new_ones(8, 8)).dtype       # follows an enclosing autocast, like the reference
        x, cond_or_gss, mask = x.to(main_type), cond_or_gss.to(main_type), mask.to(main_type)
        for blk in self.blocks:
            x = blk(x=x, cond_BD=cond_or_gss, attn_bias=mask)
        x = self.get_logits(x.float(), cond_BD)
        if self.prog_si == 0:      # keep word_embed in the graph for DDP
            x[0, 0, 0] += self.word_embed.weight[0, 0] * 0 + self.word_embed.bias[0] * 0
        return x

    @torch.no_grad()
    def inpainting(self, gt_tokens: torch.Tensor, mask: torch.Tensor, label: Optional[Union[int, torch.LongTensor]] = None,
                   g_seed: Optional[int] = None, cfg: float = 1.5, top_k: int = 0, top_p: float = 0.0, more_smooth: bool = False) -> torch.Tensor:
        """Fork API (reference var.py:236-364): resample the tokens where `mask` is False, keep `gt_tokens` where it is True;
        gt_tokens/mask are (B, L) as produced by vae.img_to_idxBl (concatenated).  Returns (B, 3, H, W) in [0, 1].
        Runs the same HIP loop as autoregressive_infer_cfg with the token replacement fused in (SamplingEngine.sample)."""
        if mask.shape != gt_tokens.shape:
            raise ValueError('Mask shape must match the latent token shape obtained from vae.img_to_idxBl')
        dev = self.lvl_1L.device
        if dev.type != 'cuda':
            raise RuntimeError('VAR.inpainting: this build runs the sampling loop on MI355X HIP kernels only (no CPU fallback by design)')
        B = gt_tokens.shape[0]
        if label is None:
            label = torch.multinomial(self.uniform_prob, num_samples=B, replacement=True).reshape(B)     # unseeded, as in the reference
        elif isinstance(label, int):
            label = torch.full((B,), fill_value=label, device=dev)
        if g_seed is None: rng = None
        else: self.rng.manual_seed(g_seed); rng = self.rng
        # more_smooth (var.py:332-341): the embeddings then come from the gumbel softmax of the filtered logits alone — the kept tokens
        # only enter through the reference's `final_tokens`, which that branch never reads; defined only when no scale is fully kept
        return self.engine().sample(B, label.to(dev).long(), rng, cfg, top_k, top_p, gt_tokens=gt_tokens, keep_mask=mask, more_smooth=bool(more_smooth))

    def smooth_sampling(self, gt_tokens: torch.Tensor, n: int, label: Optional[Union[int, torch.LongTensor]] = None,
                        g_seed: Optional[int] = None, cfg: float = 1.5, more_smooth: bool = False,
                        neighbor_threshold: Optional[float] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Fork API (reference var.py:367-572): every position takes, among the `n` nearest codebook neighbours of its
        ground-truth token (candidate-count mode: the first 1 + int((n-1)*ratio) of them; threshold mode: those within
        d_min + (neighbor_threshold - d_min)*ratio), the one with the highest CFG log-probability.
        Returns (image (B,3,H,W) in [0,1], sum of the chosen log-probabilities — each truncated to an integer first, as the
        reference's `sampled_tokens.new_tensor(max_vals)` does —, sum of log_softmax(-distance) at the chosen candidates).
        The neighbour table uses the direct-form L2 distance with ties broken by index (var_hip.h), where the reference's
        torch.cdist/argsort pair leaves both to the BLAS and an unstable sort."""
        dev = self.lvl_1L.device
        if dev.type != 'cuda':
            raise RuntimeError('VAR.smooth_sampling: this build runs the sampling loop on MI355X HIP kernels only (no CPU fallback by design)')
        B = gt_tokens.shape[0]
        if label is None:
            label = torch.multinomial(self.uniform_prob, num_samples=B, replacement=True).reshape(B)
        elif isinstance(label, int):
            label = torch.full((B,), fill_value=label, device=dev)
        if g_seed is None: rng = None
        else: self.rng.manual_seed(g_seed); rng = self.rng
        eng = self.engine()
        img = eng.sample(B, label.to(dev).long(), rng, cfg, 0, 0.0, more_smooth=more_smooth,
                         smooth=dict(gt=gt_tokens, n=n, thr=None if neighbor_threshold is None else float(neighbor_threshold)))
        sum_ll, sum_dist_ll = eng.last_smooth
        return img, sum_ll, sum_dist_ll

    # ---- initialisation (reference var.py:577-627) -------------------------------------------------------------------
    def init_weights(self, init_adaln=0.5, init_adaln_gamma=1e-5, init_head=0.02, init_std=0.02, conv_std_or_gain=0.02):
        if init_std < 0: init_std = (1 / self.C / 3) ** 0.5
        print(f'[init_weights] {type(self).__name__} with {init_std=:g}')
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Embedding)):
                nn.init.trunc_normal_(m.weight.data, std=init_std)
                if getattr(m, 'bias', None) is not None: m.bias.data.zero_()
                if isinstance(m, nn.Embedding) and m.padding_idx is not None: m.weight.data[m.padding_idx].zero_()
            elif isinstance(m, (nn.LayerNorm, nn.GroupNorm)):
                if m.weight is not None: m.weight.data.fill_(1.)
                if m.bias is not None: m.bias.data.zero_()
        if init_head >= 0:
            self.head.weight.data.mul_(init_head); self.head.bias.data.zero_()
        self.head_nm.ada_lin[-1].weight.data.mul_(init_adaln); self.head_nm.ada_lin[-1].bias.data.zero_()
        for blk in self.blocks:
            blk.attn.proj.weight.data.div_(math.sqrt(2 * self.depth))
            blk.ffn.fc2.weight.data.div_(math.sqrt(2 * self.depth))
            if hasattr(blk, 'ada_lin'):
                lin = blk.ada_lin[-1]
                lin.weight.data[2 * self.C:].mul_(init_adaln); lin.weight.data[:2 * self.C].mul_(init_adaln_gamma); lin.bias.data.zero_()
            else:
                blk.ada_gss.data[:, :, 2:].mul_(init_adaln); blk.ada_gss.data[:, :, :2].mul_(init_adaln_gamma)
        self.invalidate_engine()           # `.data` edits bump no version counter: the HIP engine must re-read the weights

    def invalidate_engine(self):
        """Tell the HIP engine that parameters changed behind autograd's back (`p.data.copy_()`, EMA swaps, checkpoint surgery)."""
        if self._engine is not None:
            self._engine.invalidate()

    def load_state_dict(self, state_dict, strict=True, assign=False):
        ret = super().load_state_dict(state_dict, strict=strict, assign=assign)
        self.invalidate_engine()
        return ret

    def extra_repr(self):
        return f'drop_path_rate={self.drop_path_rate:g}'


try:
    from huggingface_hub import PyTorchModelHubMixin

    class VARHF(VAR, PyTorchModelHubMixin):
        def __init__(self, vae_kwargs, **kwargs):
            super().__init__(vae_local=VQVAE(**vae_kwargs), **kwargs)
except Exception:       # hub mixin is optional: it only adds from_pretrained / push_to_hub
    VARHF = None
