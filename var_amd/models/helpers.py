"""Sampling helpers with the reference's names and calling conventions (reference models/helpers.py).

`sample_with_top_k_top_p_` is the fused HIP sampler (var_amd/csrc/sampler.hip): it masks `logits_BlV` in place like the
reference does and consumes one Exp(1) fill of the generator, i.e. the very stream torch.multinomial(n=1) would use."""
import torch
from torch import nn

from .. import hip


def sample_with_top_k_top_p_(logits_BlV: torch.Tensor, top_k: int = 0, top_p: float = 0.0, rng=None, num_samples=1) -> torch.Tensor:
    """(B, l, V) logits -> (B, l, num_samples) int64 token ids; logits are filtered IN PLACE (reference helpers.py:6-19)."""
    B, l, V = logits_BlV.shape
    if not logits_BlV.is_cuda or logits_BlV.dtype != torch.float32 or abs(num_samples) != 1 or not logits_BlV.is_contiguous():
        raise NotImplementedError('this build implements the top-k/top-p sampler as a HIP kernel: it needs a contiguous fp32 CUDA tensor and num_samples=1')
    if top_k > V:
        raise RuntimeError(f'top_k={top_k} out of range for V={V}')          # torch.topk raises likewise
    noise = torch.empty(B * l, V, dtype=torch.float32, device=logits_BlV.device).exponential_(1, generator=rng)
    idx = torch.empty(B * l, dtype=torch.int64, device=logits_BlV.device)
    masked = torch.empty_like(logits_BlV)
    # t_cfg = 0 makes the CFG stage the identity on the first B*l rows; the "unconditional" half is never read as such
    two = torch.cat((logits_BlV.view(B * l, V), torch.zeros_like(logits_BlV).view(B * l, V)), dim=0)
    hip.call('cfg_sample_f32', two, noise, idx, masked, B, l, V, 0.0, max(int(top_k), 0), max(float(top_p), 0.0))
    logits_BlV.copy_(masked)
    return idx.view(B, l, 1)


def gumbel_softmax_with_rng(logits: torch.Tensor, tau: float = 1, hard: bool = False, eps: float = 1e-10, dim: int = -1, rng: torch.Generator = None) -> torch.Tensor:
    """Gumbel-softmax drawing its noise from `rng` (reference helpers.py:22-36; only the more_smooth visualisation path uses it)."""
    if rng is None:
        return nn.functional.gumbel_softmax(logits=logits, tau=tau, hard=hard, eps=eps, dim=dim)
    g = -torch.empty_like(logits, memory_format=torch.legacy_contiguous_format).exponential_(generator=rng).log()
    y = ((logits + g) / tau).softmax(dim)
    if not hard:
        return y
    one_hot = torch.zeros_like(logits, memory_format=torch.legacy_contiguous_format).scatter_(dim, y.max(dim, keepdim=True)[1], 1.0)
    return one_hot - y.detach() + y


def drop_path(x, drop_prob: float = 0., training: bool = False, scale_by_keep: bool = True):
    """stochastic depth; identity unless training (reference helpers.py:39-46)"""
    if drop_prob == 0. or not training:
        return x
    keep = 1 - drop_prob
    mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
    if keep > 0.0 and scale_by_keep:
        mask.div_(keep)
    return x * mask


class DropPath(nn.Module):
    def __init__(self, drop_prob: float = 0., scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

    def forward(self, x):
        return drop_path(x, self.drop_prob, self.training, self.scale_by_keep)

    def extra_repr(self):
        return f'drop_prob={self.drop_prob}'
