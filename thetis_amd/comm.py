"""
The rank communicator behind the ``FlowSolver2d`` surface: what ``mesh.comm`` is to the reference.

Under ``mpiexec -n N python script.py`` the reference decomposes the mesh and its user script runs unchanged on every rank
(examples/README.md:51-56); the collectives it makes are few: ``comm.allreduce`` of mesh statistics and of the time step
(thetis/solver2d.py:192-193,240), all-reduced diagnostics (thetis/callback.py:478-482), rank-0 printing and file output
(thetis/log.py:43-72, thetis/callback.py:88-92).  Here the launcher is ``python -m torch.distributed.run --nproc-per-node N
script.py`` (one process per GPU) and this module is the control plane: a gloo group over CPU tensors that exists on every
node, independent of the transport that moves halo cells between GPUs (thetis_amd/distributed.py).

The host side of a run stays REPLICATED: every rank builds the same global mesh and holds the same global ``Function``s (a
1 M-triangle mesh is ~150 MB of host arrays), every rank's GPU holds and steps its own partition.  Reading a solution
field on the host after a step gathers the owned cells of all ranks (collective, like ``norm`` / ``assemble`` in the
reference), so that what a script sees is what a single-device run shows.
"""
import datetime
import os

import numpy as np

__all__ = ['Comm', 'get_comm', 'reset_comm']


class Comm(object):
    """rank / size and the handful of collectives the 2D driver makes; every method is a no-op identity on one rank."""

    def __init__(self, rank=0, size=1, group=None, local_rank=0, rccl=False):
        self.rank, self.size, self.group, self.local_rank = int(rank), int(size), group, int(local_rank)
        self.rccl = bool(rccl)                  # the default process group can move device tensors (backend "nccl" = RCCL)

    # ---- reductions of a few doubles
    def _reduce(self, values, op):
        a = np.atleast_1d(np.asarray(values, dtype=np.float64)).copy()
        if self.size > 1:
            import torch
            import torch.distributed as dist
            t = torch.from_numpy(a)
            dist.all_reduce(t, op={'min': dist.ReduceOp.MIN, 'max': dist.ReduceOp.MAX, 'sum': dist.ReduceOp.SUM}[op],
                            group=self.group)
        return a

    def allreduce_min(self, values):
        return self._reduce(values, 'min')

    def allreduce_max(self, values):
        return self._reduce(values, 'max')

    def allreduce_sum(self, values):
        """Floating point sum over the ranks (order fixed by the reduction tree of the backend, not by the partition)."""
        return self._reduce(values, 'sum')

    def allreduce_sum_int(self, values):
        """Exact (integer) sum over the ranks."""
        a = np.atleast_1d(np.asarray(values, dtype=np.int64)).copy()
        if self.size > 1:
            import torch
            import torch.distributed as dist
            dist.all_reduce(torch.from_numpy(a), op=dist.ReduceOp.SUM, group=self.group)
        return a

    def all_agree(self, ok):
        """True when ``ok`` holds on every rank (a local failure is seen by all: nobody is left waiting in a collective)."""
        return bool(self._reduce([1.0 if ok else 0.0], 'min')[0] > 0.5)

    def barrier(self):
        self._reduce([0.0], 'max')

    def allgather_object(self, obj):
        if self.size == 1:
            return [obj]
        import torch.distributed as dist
        out = [None]*self.size
        dist.all_gather_object(out, obj, group=self.group)
        return out

    def gather_rows(self, ids, rows, n_global):
        """Every rank contributes ``rows`` (n_i, ...) of the global rows ``ids`` (n_i,): returns the (n_global, ...) array on every
        rank - a pure copy (bitwise, signs of zeros included).  The pieces must cover every global row exactly once."""
        rows = np.ascontiguousarray(rows, dtype=np.float64)
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        out = np.empty((int(n_global),) + rows.shape[1:], dtype=np.float64)
        if self.size == 1:
            out[ids] = rows
            return out
        import torch
        import torch.distributed as dist
        width = int(np.prod(rows.shape[1:], dtype=np.int64)) if rows.ndim > 1 else 1
        counts = self.allgather_object(int(len(ids)))
        n_max = max(counts)
        send_i = torch.zeros(n_max, dtype=torch.int64)
        send_i[:len(ids)] = torch.from_numpy(ids)
        send_v = torch.zeros(n_max*width, dtype=torch.float64)
        send_v[:rows.size] = torch.from_numpy(rows.reshape(-1))
        recv_i = [torch.empty(n_max, dtype=torch.int64) for _ in range(self.size)]
        recv_v = [torch.empty(n_max*width, dtype=torch.float64) for _ in range(self.size)]
        dist.all_gather(recv_i, send_i, group=self.group)
        dist.all_gather(recv_v, send_v, group=self.group)
        flat = out.reshape(int(n_global), width)
        seen = 0
        for r in range(self.size):
            c = counts[r]
            flat[recv_i[r][:c].numpy()] = recv_v[r][:c*width].numpy().reshape(c, width)
            seen += c
        if seen != int(n_global):
            raise RuntimeError('gather_rows: the ranks own {:d} rows, the global array has {:d}'.format(seen, int(n_global)))
        return out


_comm = None


def reset_comm():
    """Forget the cached communicator (tests that start and stop process groups inside one interpreter)."""
    global _comm
    _comm = None


def get_comm():
    """The communicator of this process: taken from the launcher's environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*)
    on first use.  COLLECTIVE on first use when WORLD_SIZE > 1 (process group creation).

    ``THETIS_AMD_DIST_BACKEND`` = 'nccl' | 'gloo' overrides the backend of the default process group (default: RCCL for
    device tensors + gloo for CPU tensors when every local rank has a GPU of its own, gloo alone otherwise - RCCL refuses two
    ranks on one device).  A process group the caller has already initialised is used as it is."""
    global _comm
    if _comm is not None:
        return _comm
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world <= 1:
        _comm = Comm()
        return _comm
    import torch
    import torch.distributed as dist
    local_rank = int(os.environ.get('LOCAL_RANK', str(rank)))
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', str(world)))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29511')
    timeout = datetime.timedelta(seconds=float(os.environ.get('THETIS_AMD_DIST_TIMEOUT_S', '300')))
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    rccl = False
    if not dist.is_initialized():
        backend = os.environ.get('THETIS_AMD_DIST_BACKEND') or ('nccl' if n_dev >= local_world else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            try:
                dist.init_process_group(backend='cpu:gloo,cuda:nccl', rank=rank, world_size=world, timeout=timeout)
                rccl = True
            except Exception:
                if dist.is_initialized():
                    dist.destroy_process_group()
        if not dist.is_initialized():
            dist.init_process_group(backend='gloo', rank=rank, world_size=world, timeout=timeout)
    else:
        rccl = 'nccl' in str(dist.get_backend())
        rank, world = dist.get_rank(), dist.get_world_size()
    # the control plane: its own gloo group, so that its small CPU collectives never queue behind device work
    group = dist.new_group(backend='gloo', timeout=timeout)
    _comm = Comm(rank, world, group, local_rank=(local_rank % n_dev if n_dev else local_rank), rccl=rccl)
    return _comm
