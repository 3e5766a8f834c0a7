#!/usr/bin/env python
"""Times ONE stage kernel launched repeatedly on the same buffers (no A/B/C rotation) next to the full step, to separate
the cost of the access pattern from the cost of the three-buffer working set.   python tools/stagebench.py [--nx --ny]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--nx', type=int, default=1000)
    ap.add_argument('--ny', type=int, default=500)
    ap.add_argument('--reps', type=int, default=300)
    args = ap.parse_args()
    import bench
    from thetis_amd.device import Swe2dDevice
    mesh, bath, uv, eta = bench.build_case(args.nx, args.ny)
    dev = Swe2dDevice(mesh, bath, bench.DT)
    dev.set_state(uv, eta)
    out = {'n_cells': mesh.num_cells}
    for stage in (0, 1, 2):
        for _ in range(10):
            dev.solve_stage(stage)
        dev.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            dev.solve_stage(stage)
        dev.synchronize()
        out['stage{:d}_alone_us'.format(stage)] = 1e6*(time.perf_counter() - t0)/args.reps
        dev.set_state(uv, eta)
    dev.advance(10)
    dev.synchronize()
    t0 = time.perf_counter()
    dev.advance(args.reps//3)
    dev.synchronize()
    out['full_step_us'] = 1e6*(time.perf_counter() - t0)/(args.reps//3)
    print(json.dumps(out))
    dev.close()


if __name__ == '__main__':
    main()
