"""Process-global device + thin torch.distributed helpers (counterpart of the reference's top-level dist.py).

Only what the sampling path and its callers touch is provided: get_device()/initialized() (reference models/var.py:49,60;
quant.py:79), initialize() for `torchrun` launches (RCCL is torch's "nccl" backend on ROCm), rank/world queries,
barrier, and allgather — the one collective of the multi-GPU sampling path (decoded images)."""
import datetime
import os
import sys
from typing import List, Union

import torch
import torch.distributed as tdist

_state = dict(rank=0, local_rank=0, world=1, device='cuda' if torch.cuda.is_available() else 'cpu', init=False)


def initialized() -> bool:
    return _state['init']


def initialize(fork=False, backend='nccl', gpu_id_if_not_distibuted=0, timeout=30):
    if not torch.cuda.is_available():
        print('[dist initialize] cuda is not available, use cpu instead', file=sys.stderr)
        return
    if 'RANK' not in os.environ:
        torch.cuda.set_device(gpu_id_if_not_distibuted)
        _state['device'] = torch.empty(1).cuda().device
        return
    rank, ngpu = int(os.environ['RANK']), torch.cuda.device_count()
    local = int(os.environ.get('LOCAL_RANK', rank % max(ngpu, 1)))
    if backend == 'gloo':
        local %= max(ngpu, 1)               # rehearsal of an N-rank job on fewer cards (ranks share a GPU and meet on the host); RCCL wants one GPU per rank
    torch.cuda.set_device(local)
    if not tdist.is_initialized():
        tdist.init_process_group(backend=backend, timeout=datetime.timedelta(seconds=timeout * 60))
    _state.update(rank=tdist.get_rank(), local_rank=local, world=tdist.get_world_size(), device=torch.empty(1).cuda().device, init=True)


def get_rank(): return _state['rank']
def get_local_rank(): return _state['local_rank']
def get_world_size(): return _state['world']
def get_device(): return _state['device']
def is_master(): return _state['rank'] == 0
def is_local_master(): return _state['local_rank'] == 0


def set_gpu_id(gpu_id):
    if gpu_id is None: return
    torch.cuda.set_device(int(gpu_id))
    _state['device'] = torch.empty(1).cuda().device


def barrier():
    if not _state['init']: return
    if tdist.get_backend() == 'nccl': tdist.barrier(device_ids=[_state['local_rank']])      # name the device: no guessing, no warning
    else: tdist.barrier()


def allreduce(t: torch.Tensor, async_op=False):
    if not _state['init']: return None
    if t.is_cuda: return tdist.all_reduce(t, async_op=async_op)
    cu = t.detach().cuda(); ret = tdist.all_reduce(cu, async_op=async_op); t.copy_(cu.cpu()); return ret


def allgather(t: torch.Tensor, cat=True) -> Union[List[torch.Tensor], torch.Tensor]:
    """all ranks' `t` (same shape), concatenated along dim 0: one all_gather_into_tensor (RCCL over xGMI on MI355X)"""
    if not _state['init']:
        return t if cat else [t]
    if not t.is_cuda and tdist.get_backend() == 'nccl': t = t.cuda()       # RCCL moves device memory only
    t = t.contiguous()
    if t.is_cuda and tdist.get_backend() == 'gloo':                        # gloo moves host memory: stage through it (rehearsals only; the product backend is RCCL)
        host = torch.empty((_state['world'] * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype)
        tdist.all_gather_into_tensor(host, t.cpu())
        out = host.to(t.device)
    else:
        out = torch.empty((_state['world'] * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        tdist.all_gather_into_tensor(out, t)
    return out if cat else list(out.chunk(_state['world'], dim=0))


def collective_device(device):
    """where small bookkeeping tensors of a collective must live: the GPU under RCCL, the host under gloo"""
    return torch.device('cpu') if (_state['init'] and tdist.get_backend() == 'gloo') else device


def broadcast(t: torch.Tensor, src_rank) -> None:
    if _state['init']: tdist.broadcast(t, src=src_rank)


def finalize():
    if _state['init']:
        tdist.destroy_process_group(); _state['init'] = False
