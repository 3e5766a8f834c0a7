"""Rank-0 console output (thetis/log.py:43-72 behaviour: only rank 0 prints)."""
import os
import sys


def _rank():
    """The rank of this process in the run's communicator (thetis_amd/comm.py) once there is one - without creating it: printing
    must never be the first collective of a run - else what the launcher put into the environment."""
    from . import comm
    c = getattr(comm, '_comm', None)
    if c is not None:
        return int(c.rank)
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return int(dist.get_rank())
    except Exception:                                   # noqa: BLE001
        pass
    return int(os.environ.get('RANK', '0'))


def print_output(msg):
    if _rank() == 0:
        print(msg)
        sys.stdout.flush()
