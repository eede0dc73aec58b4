"""Batch-sharded sampling over the GPUs of one node: one process per GPU, no collective inside the loop, one RCCL
all-gather of the decoded images at the end (shape-equivalent to the reference's unused dist.allgather,
dist.py:109-119).

RNG modes
  'exact'    every rank seeds the same generator and draws the FULL (B_total*l, V) Exp(1) fill per scale, keeping the rows
             of its own images: the token stream is bit-identical to a single-GPU call with batch B_total (the reference
             draws that fill row-major from one generator, helpers.py:19).  Cost: W x the RNG work, ~1 % of a call.
  'per_rank' rank r seeds g_seed + r and draws only its own rows: a different but equally valid stream.
"""
from typing import Callable, Optional

import torch

from . import dist


def shard_range(B_total: int, rank: int, world: int):
    if B_total % world:
        raise ValueError(f'global batch {B_total} is not divisible by {world} ranks')
    per = B_total // world
    return rank * per, (rank + 1) * per


def sample_sharded(var, B_total: int, label_B: torch.Tensor, g_seed: Optional[int], cfg: float = 1.5, top_k: int = 0, top_p: float = 0.0,
                   rng_mode: str = 'exact', gather: bool = True, sample_fn: Optional[Callable] = None,
                   rank: Optional[int] = None, world: Optional[int] = None, gather_events: Optional[list] = None) -> torch.Tensor:
    """Sample `B_total` images split evenly over the ranks; returns all images on every rank (gather=True) or the local shard.

    `label_B`: the GLOBAL int64 label vector (same on every rank).  `sample_fn(B_local, labels_local, noise_fn)` defaults to the
    HIP engine; the CPU tests of the sharding logic substitute a stand-in.  `gather_events`: a list that receives one (start, end) pair of
    CUDA events per call around the all-gather."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_range(B_total, rank, world)
    B_local = hi - lo
    dev = var.lvl_1L.device
    V = var.V
    rng = var.rng
    if rng_mode == 'exact':
        if g_seed is not None: rng.manual_seed(g_seed)

        def noise_fn(si, l):
            full = torch.empty(B_total * l, V, dtype=torch.float32, device=rng.device).exponential_(1, generator=rng if g_seed is not None else None)
            return full.view(B_total, l, V)[lo:hi].reshape(B_local * l, V)
    elif rng_mode == 'per_rank':
        if g_seed is not None: rng.manual_seed(g_seed + rank)

        def noise_fn(si, l):
            return torch.empty(B_local * l, V, dtype=torch.float32, device=rng.device).exponential_(1, generator=rng if g_seed is not None else None)
    else:
        raise ValueError(f'unknown rng_mode {rng_mode!r}')
    labels_local = label_B[lo:hi].to(dev).long()
    if sample_fn is None:
        img = var.engine().sample(B_local, labels_local, None, cfg, top_k, top_p, noises=noise_fn)
    else:
        img = sample_fn(B_local, labels_local, noise_fn)
    if not gather or world == 1:
        return img
    if gather_events is None or not img.is_cuda:
        return dist.allgather(img, cat=True)
    # bench.py: HIP events around the collective on the stream it is enqueued on (diagnosis of a scaling run: a rank that arrives early
    # waits inside the all-gather, so max - min over ranks of this time is the arrival skew)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    out = dist.allgather(img, cat=True)
    b.record()
    gather_events.append((a, b))
    return out


def rank_stats(dt: float, dt_own: float, gather_ms: float, steps: int, gathered_shape, device) -> dict:
    """bench.py, N > 1: one all-gather of (elapsed time to the closing barrier, this rank's own elapsed time, all-gather ms per step) over the
    ranks -> the job's time (max) and the `ranks` object of the bench line: the spread a scaling run needs to be diagnosed (a slow rank shows in
    rank_ms_max; ranks that arrive early wait inside the all-gather, so allgather_ms max - min is the arrival skew)."""
    import torch.distributed as tdist
    world = dist.get_world_size()
    t = torch.tensor([dt, dt_own, gather_ms], device=dist.collective_device(device), dtype=torch.float64)
    allt = [torch.empty_like(t) for _ in range(world)]
    tdist.all_gather(allt, t)
    rows = [[float(v) for v in x.cpu()] for x in allt]
    nbytes = 4
    for d in gathered_shape: nbytes *= int(d)
    return {'dt_max': max(r[0] for r in rows),
            'ranks': {'rank_ms_min': round(min(r[1] for r in rows) / steps * 1e3, 3), 'rank_ms_max': round(max(r[1] for r in rows) / steps * 1e3, 3),
                      'allgather_ms': round(max(r[2] for r in rows), 3), 'allgather_ms_min': round(min(r[2] for r in rows), 3),
                      'allgather_mbytes': round(nbytes / 1e6, 1),
                      'note': "rank_ms_*: each rank's own wall time per step for the timed region (sampling + decode + all-gather, before the closing barrier); "
                              'allgather_ms: HIP-event time of the RCCL all-gather per step (max / min over ranks; a rank that arrives early waits inside it)'}}
