#!/usr/bin/env python
"""Kernel A/B harness: times the stage kernels of one libswe2d_hip.so build (THETIS_AMD_LIB) on the bench workload.
   python tools/kbench.py [--order natural|tile:BX:BY] [--steps K]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def tile_perm(mesh, bx, by):
    """Cell order by (tile row, tile col, row in tile, col in tile) of bx x by quads."""
    nx, ny = mesh.nx, mesh.ny
    q = np.arange(mesh.num_cells)//2
    i, j = q % nx, q//nx
    key = np.lexsort((np.arange(mesh.num_cells), i % bx, j % by, i//bx, j//by))
    return key


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--order', default='auto', help="device cell numbering: 'auto' (what every other entry point uses), 'natural', "
                    "'hilbert', 'tile:BX:BY', 'htile:BX:BY'")
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--tag', default='')
    ap.add_argument('--nx', type=int, default=0, help='override the mesh: RectangleMesh(nx, ny), same cell size')
    ap.add_argument('--ny', type=int, default=0)
    ap.add_argument('--prewarm', type=float, default=0.3, help='seconds of stepping before the measurement (clock settling)')
    ap.add_argument('--calibrate', action='store_true', help='also run the PMC calibration copy kernel')
    args = ap.parse_args()
    import bench
    from thetis_amd.device import Swe2dDevice
    if args.nx:
        mesh, bath, uv, eta = bench.build_case(args.nx, args.ny)
    else:
        mesh, bath, uv, eta = bench.build_case()
    reorder = None
    if args.order.startswith('tile'):
        _, bx, by = args.order.split(':')
        reorder = tile_perm(mesh, int(bx), int(by))
    elif args.order in ('hilbert', 'auto'):
        reorder = args.order
    elif args.order != 'natural':
        raise SystemExit('unknown --order ' + args.order)
    elif args.order == 'hilbert-rows':
        from thetis_amd import ordering
        cen = mesh.cell_xy().mean(axis=1)
        reorder = ordering.patch_row_order(cen, ordering.hilbert_cell_order(cen))
    elif args.order.startswith('htile'):
        # tiles of bx x by quads visited along a Hilbert curve over the tile grid, row-major inside a tile
        from thetis_amd import ordering
        _, bx, by = args.order.split(':')
        bx, by = int(bx), int(by)
        q = np.arange(mesh.num_cells)//2
        i, j = q % mesh.nx, q//mesh.nx
        d = ordering.hilbert_index(i//bx, j//by, 12)
        reorder = np.lexsort((np.arange(mesh.num_cells), i % bx, j % by, d))
    # keep the CFL number of the bench workload when the mesh is refined
    dt = bench.DT*min(1.0, 1000.0/max(args.nx, 1) if args.nx else 1.0, 500.0/max(args.ny, 1) if args.ny else 1.0)
    dev = Swe2dDevice(mesh, bath, dt, reorder=reorder)
    dev.set_state(uv, eta)
    import time
    t0 = time.perf_counter()
    dev.advance(5)
    dev.synchronize()
    while time.perf_counter() - t0 < args.prewarm:
        dev.advance(100)
        dev.synchronize()
    if args.calibrate:
        dev._ck(dev.lib.swe2d_debug_calibration_copy(dev.h, 5))
    best = 1e9
    for rep in range(3):
        ms_tot, _ = dev.advance_timed(args.steps, per_launch=False)
        best = min(best, ms_tot/args.steps)
    _, ms_k = dev.advance_timed(args.steps, per_launch=True)
    try:
        d = dev.diagnostics()
    except Exception:          # experimental kernel variants may produce garbage
        d = [float('nan')]*4
    n = mesh.num_cells
    fused = {}
    try:
        fused = {'fused_pair': list(dev.fused_pair_info()), 'fused_triple': list(dev.fused_triple_info())}
    except Exception:
        pass
    print(json.dumps({'tag': args.tag or os.environ.get('THETIS_AMD_LIB', 'default'), 'order': args.order, 'n_cells': n, 'fused': fused,
                      'us_per_step': 1e3*best, 'us_per_launch': 1e3*ms_k, 'frac': 684.0*n/(best*1e-3)/8e12,
                      'vol': d[2]}))
    dev.close()


if __name__ == '__main__':
    main()
