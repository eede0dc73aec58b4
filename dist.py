"""Top-level `dist` module, as the reference's callers import it (`import dist`): re-exports var_amd.dist."""
from var_amd.dist import *  # noqa: F401,F403
from var_amd.dist import (allgather, allreduce, barrier, broadcast, finalize, get_device, get_local_rank, get_rank,  # noqa: F401
                          get_world_size, initialize, initialized, is_local_master, is_master, set_gpu_id)
