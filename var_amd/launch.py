"""Start one rank per GPU on this node (counterpart of launching the reference with `torchrun`, README.md:131-144, whose ranks
then read RANK / WORLD_SIZE in dist.py:20-49).

`python bench.py --gpus N` outside a torchrun environment calls `spawn_ranks`: the parent process never touches a GPU (no HIP call,
no exec of a GPU-initialised process); it starts `python -m torch.distributed.run --nproc-per-node N <script> <argv>` as a CHILD,
lets the children's stdout through (rank 0 prints the one JSON line) and returns the child's exit code."""
import os
import socket
import subprocess
import sys
from typing import Callable, List, Optional


def under_launcher(env=None) -> bool:
    """True inside a torchrun / torch.distributed.run rank (it exports RANK and WORLD_SIZE)"""
    env = os.environ if env is None else env
    return 'RANK' in env and 'WORLD_SIZE' in env


def free_port() -> int:
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


KFD_NODES = '/sys/class/kfd/kfd/topology/nodes'


def _visible_filter(n: int, env) -> int:
    """apply ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES (comma lists; each narrows the previous one) to n GPUs"""
    for k in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = env.get(k)
        if v is not None:
            n = min(n, len([t for t in v.split(',') if t.strip() != '']))
    return n


def _gpu_count(nodes_dir: str = KFD_NODES, env=None, dri_dir: str = '/dev/dri') -> int:
    """GPUs of this node WITHOUT loading the HIP/HSA runtime into this process: the KFD topology lists one node per agent and a GPU
    node is one with simd_count > 0 (CPU nodes have 0) whose render node (drm_render_minor) this process may open — a container that
    was handed one GPU of eight still sees all eight in the topology.  torch.cuda.device_count() can fall through to
    hipGetDeviceCount on ROCm, which would make the launcher a GPU-initialised parent of the ranks."""
    env = os.environ if env is None else env
    n = 0
    try:
        for node in sorted(os.listdir(nodes_dir)):
            try:
                with open(os.path.join(nodes_dir, node, 'properties')) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            except OSError:
                continue
            if int(props.get('simd_count', '0')) > 0:
                minor = props.get('drm_render_minor')
                if minor is None or os.access(os.path.join(dri_dir, f'renderD{minor}'), os.R_OK | os.W_OK):
                    n += 1
    except OSError:
        return 0
    return _visible_filter(n, env)


def spawn_ranks(script: str, argv: List[str], nproc: int, device_count: Optional[Callable[[], int]] = None,
                need_gpus: bool = True, timeout: Optional[float] = None) -> int:
    """Run `script argv` as `nproc` ranks on 127.0.0.1; returns the launcher's exit code (non-zero if any rank failed).
    Refuses (code 2) when the node has fewer GPUs than ranks."""
    if nproc < 1:
        raise ValueError('nproc must be >= 1')
    if need_gpus:
        have = (device_count or _gpu_count)()
        if have < nproc:
            print(f'[launch] {nproc} ranks requested but this node shows {have} GPU(s)', file=sys.stderr, flush=True)
            return 2
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC: RCCL needs it on this driver
    env.setdefault('OMP_NUM_THREADS', '4')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), script, *argv]
    try:
        return subprocess.run(cmd, env=env, timeout=timeout).returncode
    except subprocess.TimeoutExpired:
        print(f'[launch] ranks did not finish within {timeout} s', file=sys.stderr, flush=True)
        return 124
