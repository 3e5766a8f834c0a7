#!/usr/bin/env python
"""A/B of the one-launch step kernel (csrc/swe2d_step.h) against the stage-by-stage path on the bench workload at several sizes.
   python tools/stepbench.py [--sizes 354x177,500x250,1000x500] [--steps K] [--tiles C,B ...]
Wall time per step around dev.advance(K) (stream synchronised), both paths in the same process on the same box."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(dev, steps, reps=5):
    best = 1e30
    for _ in range(reps):
        dev.synchronize()
        t0 = time.perf_counter()
        dev.advance(steps)
        dev.synchronize()
        best = min(best, (time.perf_counter() - t0)/steps)
    return best*1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--sizes', default='250x125,354x177,500x250,707x354,1000x500')
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--tiles', nargs='*', default=['256,384'])
    ap.add_argument('--prewarm', type=float, default=0.3)
    ap.add_argument('--sources', action='store_true', help='Manning drag + Coriolis + wind stress (the SRC kernel variants)')
    args = ap.parse_args()
    import numpy as np
    import bench
    from thetis_amd.device import Swe2dDevice
    for size in args.sizes.split(','):
        nx, ny = (int(x) for x in size.split('x'))
        mesh, bath, uv, eta = bench.build_case(nx, ny)
        dt = bench.DT*1000.0/nx
        row = {'cells': mesh.num_cells}
        ref = None
        for tile in [None] + list(args.tiles):
            if tile is None:
                os.environ['THETIS_AMD_FUSED_STEP'] = '0'
            else:
                os.environ['THETIS_AMD_FUSED_STEP'] = '1'
                os.environ['THETIS_AMD_STEP_TILE'] = tile
            dev = Swe2dDevice(mesh, bath, dt, boundary_len=mesh.boundary_len)
            if args.sources:
                from thetis_amd import _lib
                dev.set_scalar(_lib.SCALAR_MANNING_DRAG, 0.02)
                dev.set_field(_lib.FIELD_CORIOLIS, 1e-4*np.ones((mesh.num_cells, 3)))
                dev.set_field(_lib.FIELD_WIND_STRESS, 0.1*np.ones((mesh.num_cells, 3, 2)))
            dev.set_state(uv, eta)
            t_end = time.perf_counter() + args.prewarm
            while time.perf_counter() < t_end:
                dev.advance(50)
                dev.synchronize()
            dev.set_state(uv, eta)
            us = timed(dev, args.steps)
            dev.set_state(uv, eta)
            dev.advance(6)
            st = dev.get_state()
            if ref is None:
                ref = st
            same = bool(np.array_equal(ref[0], st[0]) and np.array_equal(ref[1], st[1]))
            row['stages' if tile is None else 'fused ' + tile] = round(us, 2)
            if tile is not None:
                row['same_bits ' + tile] = same
            dev.close()
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
