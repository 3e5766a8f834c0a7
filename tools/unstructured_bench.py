#!/usr/bin/env python
"""Stage-kernel throughput on an UNSTRUCTURED triangulation (Delaunay of random points in a rectangle) of about the bench
size: the cell order is the Hilbert curve through the centroids, the vertex order first-touch (thetis_amd/ordering.py).
Measured (MI355X, 1 M triangles, random input numbering): 1442 us/step in the given order, 139-150 us/step (0.57-0.61 of
8 TB/s) along the Hilbert curve; sorting by rows inside 64/256/1024-cell patches of the curve changes nothing (the same order
timed first and second in one process differs by more: the GPU slows a few % under sustained load).
   python tools/unstructured_bench.py [--points 500000]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--points', type=int, default=500000)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--triple', action='store_true', help='the three-stage kernel on patches of the bisection order against the fused pair')
    args = ap.parse_args()
    from scipy.spatial import Delaunay
    from helpers import _rect_marker_fn
    from thetis_amd.device import Swe2dDevice
    from thetis_amd.mesh import Mesh2d
    t0 = time.perf_counter()
    # Delaunay triangulation of a jittered point grid (regular boundary points, no slivers), then a RANDOM numbering of
    # cells and vertices: what a mesh generator may hand over
    lx, ly = 100e3, 50e3
    ny = int(round(np.sqrt(args.points/2.0)))
    nx = 2*ny
    rng = np.random.default_rng(1)
    gx, gy = np.meshgrid(np.linspace(0, lx, nx + 1), np.linspace(0, ly, ny + 1), indexing='ij')
    jit = 0.35*lx/nx
    gx[1:-1, 1:-1] += rng.uniform(-jit, jit, size=gx[1:-1, 1:-1].shape)
    gy[1:-1, 1:-1] += rng.uniform(-jit, jit, size=gy[1:-1, 1:-1].shape)
    pts = np.stack([gx.ravel(), gy.ravel()], axis=1)
    cells = Delaunay(pts).simplices
    vperm = rng.permutation(len(pts))
    vinv = np.empty_like(vperm)
    vinv[vperm] = np.arange(len(pts))
    pts, cells = pts[vperm], vinv[cells][rng.permutation(len(cells))]
    mesh = Mesh2d(pts, cells, marker_fn=_rect_marker_fn(lx, ly))
    bath = 15.0 + 5.0*np.sin(mesh.vertex_xy[:, 0]/lx*3.0)*np.cos(mesh.vertex_xy[:, 1]/ly*2.0)
    uv = rng.normal(size=(mesh.num_cells, 3, 2))
    eta = rng.normal(size=(mesh.num_cells, 3))
    t_mesh = time.perf_counter() - t0
    n = mesh.num_cells
    xy = mesh.cell_xy()
    area = 0.5*np.abs((xy[:, 1, 0] - xy[:, 0, 0])*(xy[:, 2, 1] - xy[:, 0, 1]) - (xy[:, 2, 0] - xy[:, 0, 0])*(xy[:, 1, 1] - xy[:, 0, 1]))
    dt = 0.02*np.sqrt(area.min())/np.sqrt(9.81*20.0)
    out = {}
    from thetis_amd import ordering
    cen = mesh.cell_xy().mean(axis=1)
    hil = ordering.hilbert_cell_order(cen)
    variants = {'warmup(auto)': 'auto', 'auto': 'auto', 'hilbert': hil, 'None': None}
    for patch in (64, 256, 1024):
        variants['hilbert+rows{:d}'.format(patch)] = ordering.patch_row_order(cen, hil, patch=patch)
    tiles = {}
    if args.triple:
        # all three stages in one launch (csrc/swe2d_fuse.h): two-ring tiles as runs of the Hilbert order (what the library cuts by
        # itself), as leaves of a coordinate bisection with exactly K cells (compact boxes) over the Hilbert numbering, and with the
        # leaves as the device numbering itself (a tile's cells consecutive in memory)
        from thetis_amd import _lib
        variants = {'warmup(auto)': 'auto', 'pair (auto)': 'auto', 'triple, runs of the Hilbert order': 'auto'}
        for K in (144, 152, 160, 168):
            leaves = ordering.bisection_block_order(cen, block=K)
            variants['triple, bisection leaves of {:d} over the Hilbert numbering'.format(K)] = 'auto'
            variants['triple, bisection leaves of {:d} as the numbering'.format(K)] = leaves
            tiles['triple, bisection leaves of {:d} over the Hilbert numbering'.format(K)] = (leaves, np.arange(0, n, K))
            tiles['triple, bisection leaves of {:d} as the numbering'.format(K)] = (leaves, np.arange(0, n, K))
        variants['pair (auto) again'] = 'auto'
    for name, reorder in variants.items():
        if name == 'None' and os.environ.get('SKIP_UNORDERED'):
            continue
        dev = Swe2dDevice(mesh, bath, dt, reorder=reorder)
        info = None
        if args.triple:
            dev.set_option(_lib.OPT_FUSED_STAGES, 3 if name.startswith('triple') else 2)
            if name in tiles:
                dev.fused_set_triple_tiles(*tiles[name])
            info = list(dev.fused_triple_info()) if name.startswith('triple') else list(dev.fused_pair_info())
        dev.set_state(0.01*uv, 0.01*eta)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.6:
            dev.advance(50)
            dev.synchronize()
        best = 1e9
        for _ in range(3):
            ms, _k = dev.advance_timed(args.steps, per_launch=False)
            best = min(best, ms/args.steps)
        d = dev.diagnostics()
        out[name] = {'us_per_step': 1e3*best, 'frac_of_8TBs': 684.0*n/(best*1e-3)/8e12, 'finite': bool(np.isfinite(d).all())}
        if info is not None:
            out[name]['tiles'] = info
        dev.close()
    print(json.dumps({'n_cells': int(n), 'mesh_build_s': t_mesh, 'dt': dt, 'order': out}))


if __name__ == '__main__':
    main()
