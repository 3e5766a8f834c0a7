#!/usr/bin/env python
"""Chunked stepping on ONE device for meshes beyond the Infinity Cache, measured with what the library
   already has: the mesh is cut into P strips, every strip becomes a range of ONE device mesh together with copies of the three facet
   layers of its neighbours (thetis_amd.partition.build_partition: the ghost layers of a rank), the three stages of a step run strip
   after strip on shrinking ranges (swe2d_solve_stage_cells) - a strip's three buffers then stay in the cache between its stages -
   and one pack + one unpack launch per step refresh the copies (swe2d_halo_pack / _unpack with both lists on the same handle).
   The result is compared with plain stepping of the same mesh (bit for bit on a small mesh: --check).
   python tools/chunkbench.py --nx 2000 --ny 1000 --chunks 6 [--steps 30]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class ChunkedMesh(object):
    pass


def build(mesh, n_chunks):
    from thetis_amd.partition import strip_owner, build_partition
    owner = strip_owner(mesh, n_chunks)
    parts = [build_partition(mesh, owner, r, halo_depth=3) for r in range(n_chunks)]
    cm = ChunkedMesh()
    cbase = np.cumsum([0] + [p.num_cells for p in parts])
    vbase = np.cumsum([0] + [p.num_vertices for p in parts])
    cm.cells = np.concatenate([p.cells.astype(np.int64) + vbase[r] for r, p in enumerate(parts)]).astype(np.int32)
    cm.vertex_xy = np.concatenate([p.vertex_xy for p in parts])
    cm.cell_nbr = np.concatenate([np.where(p.cell_nbr >= 0, p.cell_nbr.astype(np.int64) + cbase[r], p.cell_nbr)
                                  for r, p in enumerate(parts)]).astype(np.int32)
    cm.cell_nbr_facet = np.concatenate([p.cell_nbr_facet for p in parts])
    cm.boundary_len = dict(mesh.boundary_len)
    cm.boundary_markers = mesh.boundary_markers
    cm.num_cells, cm.num_vertices = int(cbase[-1]), int(vbase[-1])
    cm.local_to_global = np.concatenate([p.local_to_global for p in parts])
    cm.vertex_global = np.concatenate([p.vertex_global for p in parts])
    cm.structured_parent = (int(mesh.nx), int(mesh.ny))
    cm.topo_vertex = None
    # ranges a device numbering must not mix: owned cells and every ghost layer of every chunk
    bounds = []
    for r, p in enumerate(parts):
        bounds.append(cbase[r] + p.n_owned)
        for l in range(1, len(p.layer_sizes) + 1):
            bounds.append(cbase[r] + p.layer_end(l))
    cm.ranges = sorted(set(int(b) for b in bounds))
    # the copies' refresh: ghost cell <- the owner's cell of the same global id
    g2owned = np.full(mesh.num_cells, -1, dtype=np.int64)
    for r, p in enumerate(parts):
        g2owned[p.local_to_global[:p.n_owned]] = cbase[r] + np.arange(p.n_owned)
    recv = np.concatenate([cbase[r] + np.arange(p.n_owned, p.num_cells) for r, p in enumerate(parts)])
    send = g2owned[cm.local_to_global[recv]]
    assert (send >= 0).all()
    cm.send, cm.recv = send, recv
    cm.stages = [[(int(cbase[r]), int(cbase[r] + p.stage_range(s))) for s in range(3)] for r, p in enumerate(parts)]
    cm.owned = np.concatenate([cbase[r] + np.arange(p.n_owned) for r, p in enumerate(parts)])
    return cm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--nx', type=int, default=2000)
    ap.add_argument('--ny', type=int, default=1000)
    ap.add_argument('--chunks', type=int, default=6)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--check', action='store_true', help='compare the owned cells with plain stepping (small meshes)')
    args = ap.parse_args()
    import torch
    import bench
    from thetis_amd.device import Swe2dDevice
    os.environ['THETIS_AMD_FLOW'] = '0'
    mesh, bath, uv, eta = bench.build_case(args.nx, args.ny)
    dt = bench.DT*min(1.0, 1000.0/args.nx, 500.0/args.ny)
    t0 = time.perf_counter()
    cm = build(mesh, args.chunks)
    t_build = time.perf_counter() - t0
    dev = Swe2dDevice(cm, np.asarray(bath)[cm.vertex_global], dt, boundary_len=cm.boundary_len, ranges=cm.ranges)
    dev.set_state(uv[cm.local_to_global], eta[cm.local_to_global])
    dev.halo_setup(cm.send, cm.recv)
    buf = torch.empty(len(cm.recv)*9, dtype=torch.float64, device='cuda')

    def step(n):
        for _ in range(n):
            for chunk in cm.stages:
                for s, (a, b) in enumerate(chunk):
                    dev.solve_stage_cells(s, a, b)
            dev.halo_pack(0, buf.data_ptr())
            dev.halo_unpack(0, buf.data_ptr())

    out = {'n_cells': mesh.num_cells, 'chunks': args.chunks, 'cells_with_copies': cm.num_cells, 'build_s': round(t_build, 1)}
    if args.check:
        plain = Swe2dDevice(mesh, bath, dt)
        plain.set_state(uv, eta)
        plain.advance(args.steps)
        pu, pe = plain.get_state()
        step(args.steps)
        dev.synchronize()
        cu, ce = dev.get_state()
        out['equal_bits'] = bool(np.array_equal(cu[cm.owned], pu[cm.local_to_global[cm.owned]])
                                 and np.array_equal(ce[cm.owned], pe[cm.local_to_global[cm.owned]]))
        plain.close()
    else:
        plain = Swe2dDevice(mesh, bath, dt)
        plain.set_state(uv, eta)
        t_end = time.perf_counter() + 0.4
        while time.perf_counter() < t_end:
            plain.advance(20)
            plain.synchronize()
        best = 1e9
        for _ in range(3):
            ms, _k = plain.advance_timed(args.steps, per_launch=False)
            best = min(best, ms/args.steps)
        out['plain_us_per_step'] = round(1e3*best, 1)
        plain.close()
        step(10)
        dev.synchronize()
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            step(args.steps)
            dev.synchronize()
            best = min(best, (time.perf_counter() - t0)/args.steps)
        out['chunked_us_per_step'] = round(1e6*best, 1)
        out['frac_plain'] = round(684.0*mesh.num_cells/(out['plain_us_per_step']*1e-6)/8e12, 3)
        out['frac_chunked'] = round(684.0*mesh.num_cells/(out['chunked_us_per_step']*1e-6)/8e12, 3)
    print(json.dumps(out))
    dev.close()


if __name__ == '__main__':
    main()
