#!/usr/bin/env python
"""The dataflow kernel on an UNSTRUCTURED triangulation small enough for it (Delaunay of ~62 k jittered points = ~124 k triangles): us per
step of swe2d_advance with the blocks of ordering.flow_block_order (bisection boxes) and, THETIS_AMD_FLOW_BLOCKS=0, with 64 consecutive
cells of the device numbering (Hilbert curve); THETIS_AMD_FLOW=0: stage launches.
   python tools/unstructured_flow.py [--points 62000]"""
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
    ap.add_argument('--points', type=int, default=62000)
    ap.add_argument('--steps', type=int, default=384)
    ap.add_argument('--tag', default='')
    args = ap.parse_args()
    from helpers import delaunay_case
    from thetis_amd.device import Swe2dDevice
    mesh, bath, uv, eta = delaunay_case(n_points=args.points, seed=5)[:4]
    dev = Swe2dDevice(mesh, bath, 0.02, boundary_len=mesh.boundary_len)
    dev.set_state(uv, eta)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        dev.advance(64)
        dev.synchronize()
    best = 1e9
    for _ in range(3):
        dev.set_state(uv, eta)
        dev.synchronize()
        t0 = time.perf_counter()
        dev.advance(args.steps)
        dev.synchronize()
        best = min(best, (time.perf_counter() - t0)/args.steps)
    print(json.dumps({'tag': args.tag, 'n_cells': int(mesh.num_cells), 'flow_supported': dev.flow_supported(), 'us_per_step': 1e6*best,
                      'flow_blocks': os.environ.get('THETIS_AMD_FLOW_BLOCKS', '1'), 'flow': os.environ.get('THETIS_AMD_FLOW', '1')}))
    dev.close()


if __name__ == '__main__':
    main()
