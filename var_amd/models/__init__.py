"""`models` package of the MI355X build: same import names as the reference (models/__init__.py)."""
from typing import Tuple

import torch.nn as nn

from .quant import VectorQuantizer2
from .var import VAR
from .vqvae import VQVAE

# layers whose default initialisers the reference's factory switches off process-wide (checkpoints, or init_weights, fill them)
_NO_DEFAULT_INIT = (nn.Linear, nn.LayerNorm, nn.BatchNorm2d, nn.SyncBatchNorm, nn.Conv1d, nn.Conv2d, nn.ConvTranspose1d, nn.ConvTranspose2d)


def _skip_default_inits():
    def _noop(module):
        return None
    for layer_type in _NO_DEFAULT_INIT:
        layer_type.reset_parameters = _noop


def build_vae_var(device, patch_nums=(1, 2, 3, 4, 5, 6, 8, 10, 13, 16), V=4096, Cvae=32, ch=160, share_quant_resi=4,
                  num_classes=1000, depth=16, shared_aln=False, attn_l2_norm=True, flash_if_available=True, fused_if_available=True,
                  init_adaln=0.5, init_adaln_gamma=1e-5, init_head=0.02, init_std=-1) -> Tuple[VQVAE, VAR]:
    """Factory with the reference's signature.  Sizes follow from the depth: `depth` heads of 64 channels, stochastic-depth rate
    0.1 * depth / 24 (inactive in eval), LayerNorm eps 1e-6, label dropout 0.1 for training."""
    _skip_default_inits()
    tokenizer = VQVAE(vocab_size=V, z_channels=Cvae, ch=ch, test_mode=True, share_quant_resi=share_quant_resi, v_patch_nums=patch_nums)
    tokenizer = tokenizer.to(device)
    width, heads = 64 * depth, depth
    transformer_cfg = dict(num_classes=num_classes, depth=depth, embed_dim=width, num_heads=heads,
                           drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=depth * 0.1 / 24,
                           norm_eps=1e-6, shared_aln=shared_aln, cond_drop_rate=0.1, attn_l2_norm=attn_l2_norm, patch_nums=patch_nums,
                           flash_if_available=flash_if_available, fused_if_available=fused_if_available)
    model = VAR(vae_local=tokenizer, **transformer_cfg).to(device)
    model.init_weights(init_adaln=init_adaln, init_adaln_gamma=init_adaln_gamma, init_head=init_head, init_std=init_std)
    return tokenizer, model
