"""`models` package of the MI355X build: same import names as the reference (models/__init__.py)."""
from typing import Tuple

import torch.nn as nn

from .quant import VectorQuantizer2
from .var import VAR
from .vqvae import VQVAE


def build_vae_var(device, patch_nums=(1, 2, 3, 4, 5, 6, 8, 10, 13, 16), V=4096, Cvae=32, ch=160, share_quant_resi=4,
                  num_classes=1000, depth=16, shared_aln=False, attn_l2_norm=True, flash_if_available=True, fused_if_available=True,
                  init_adaln=0.5, init_adaln_gamma=1e-5, init_head=0.02, init_std=-1) -> Tuple[VQVAE, VAR]:
    """factory with the reference's signature and derived sizes: heads = depth, width = 64*depth, dpr = 0.1*depth/24"""
    for clz in (nn.Linear, nn.LayerNorm, nn.BatchNorm2d, nn.SyncBatchNorm, nn.Conv1d, nn.Conv2d, nn.ConvTranspose1d, nn.ConvTranspose2d):
        setattr(clz, 'reset_parameters', lambda self: None)        # as the reference: checkpoints overwrite everything anyway
    vae = VQVAE(vocab_size=V, z_channels=Cvae, ch=ch, test_mode=True, share_quant_resi=share_quant_resi, v_patch_nums=patch_nums).to(device)
    var = VAR(vae_local=vae, num_classes=num_classes, depth=depth, embed_dim=depth * 64, num_heads=depth, drop_rate=0., attn_drop_rate=0.,
              drop_path_rate=0.1 * depth / 24, norm_eps=1e-6, shared_aln=shared_aln, cond_drop_rate=0.1, attn_l2_norm=attn_l2_norm,
              patch_nums=patch_nums, flash_if_available=flash_if_available, fused_if_available=fused_if_available).to(device)
    var.init_weights(init_adaln=init_adaln, init_adaln_gamma=init_adaln_gamma, init_head=init_head, init_std=init_std)
    return vae, var
