#!/usr/bin/env python
"""
One process per GPU: the SSPRK33 shallow water step on a strip-partitioned (or RCB-partitioned) mesh with halo exchange
over torch.distributed (RCCL), thetis_amd/distributed.py.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29512 \
        examples/multi_gpu.py --nx 2000 --ny 1000 --steps 400 [--rcb] [--exchange-every 4]

Every rank builds its own partition from the replicated mesh (deterministic, no handshake); results are bitwise those of
a single-device run of the same mesh for every partitioning and exchange schedule.

This script drives the multi-GPU ENGINE directly (arrays in, arrays out: what bench.py times).  A ``FlowSolver2d`` user script needs
none of it: ``python -m torch.distributed.run --nproc-per-node N examples/channel2d.py`` runs the unchanged single-GPU script
partitioned (thetis_amd/spmd.py), as ``mpiexec -n N`` does for the reference.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thetis_amd.mesh import RectangleMesh                                     # noqa: E402
from thetis_amd.distributed import DistributedSwe2d                          # noqa: E402
from thetis_amd.partition import rcb_owner, strip_owner                      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--nx', type=int, default=1000)
    ap.add_argument('--ny', type=int, default=500)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--rcb', action='store_true', help='recursive coordinate bisection instead of strips along x')
    ap.add_argument('--exchange-every', type=int, default=4)
    ap.add_argument('--overlap-stages', type=int, default=0)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', str(rank)))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29512')
    torch.cuda.set_device(local_rank)
    dist.init_process_group(backend='nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))

    lx, ly = 100.0*args.nx, 100.0*args.ny
    mesh = RectangleMesh(args.nx, args.ny, lx, ly)
    bath = np.full(mesh.num_vertices, 20.0)
    xy = mesh.cell_xy()
    eta = 0.5*np.exp(-((xy[:, :, 0] - 0.5*lx)**2 + (xy[:, :, 1] - 0.5*ly)**2)/(0.05*lx)**2)
    uv = np.zeros(xy.shape)
    owner = rcb_owner(mesh, world) if args.rcb else strip_owner(mesh, world)
    solver = DistributedSwe2d(mesh, bath, 0.25, rank, world, local_rank, owner=owner,
                              exchange_every=args.exchange_every, overlap_stages=args.overlap_stages)
    solver.set_state_global(uv, eta)
    d0 = solver.diagnostics()
    solver.advance(args.exchange_every, use_graph=False)         # connections, module loading
    solver.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    solver.advance(args.steps)
    solver.synchronize()
    dist.barrier()
    t = time.perf_counter() - t0
    d1 = solver.diagnostics()
    if rank == 0:
        print('{:d} ranks, {:d} cells: {:.1f} us/step, {:.3e} element-updates/s, volume drift {:.1e}'.format(
            world, mesh.num_cells, 1e6*t/args.steps, 3.0*mesh.num_cells*args.steps/t, abs(d1[2] - d0[2])/d0[2]))
    solver.dev.close()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
