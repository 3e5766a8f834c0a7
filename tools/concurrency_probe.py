#!/usr/bin/env python
"""How much of a stage launch's ramp and tail does a second, independent launch hide?  Two handles (two meshes of the bench
   geometry) on ONE device, each on its own stream: stepped one after the other, then interleaved (the device runs their stage
   kernels concurrently).  The ratio sequential / concurrent bounds what running two dependent chains of half-launches of ONE
   mesh side by side could gain.
   python tools/concurrency_probe.py --nx 707 --ny 354 --nx2 707 --ny2 354 [--steps 200]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--nx', type=int, default=707)
    ap.add_argument('--ny', type=int, default=354)
    ap.add_argument('--nx2', type=int, default=707)
    ap.add_argument('--ny2', type=int, default=354)
    ap.add_argument('--steps', type=int, default=200)
    args = ap.parse_args()
    import bench
    from thetis_amd.device import Swe2dDevice
    os.environ['THETIS_AMD_FLOW'] = '0'
    devs = []
    for nx, ny in ((args.nx, args.ny), (args.nx2, args.ny2)):
        mesh, bath, uv, eta = bench.build_case(nx, ny)
        dt = bench.DT*min(1.0, 1000.0/nx, 500.0/ny)
        d = Swe2dDevice(mesh, bath, dt, reorder='auto')
        d.set_state(uv, eta)
        devs.append((d, mesh.num_cells))

    def sync():
        for d, _ in devs:
            d.synchronize()

    def run(which, n):
        sync()
        t0 = time.perf_counter()
        if which == 'both':
            for _ in range(n):
                devs[0][0].advance(1)
                devs[1][0].advance(1)
        else:
            devs[which][0].advance(n)
        sync()
        return (time.perf_counter() - t0)/n*1e6

    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:
        run('both', 50)
    out = {'cells': [n for _, n in devs]}
    for rep in range(3):
        a, b, c = run(0, args.steps), run(1, args.steps), run('both', args.steps)
        out.setdefault('us_per_step_first', []).append(round(a, 2))
        out.setdefault('us_per_step_second', []).append(round(b, 2))
        out.setdefault('us_per_step_both', []).append(round(c, 2))
    a, b, c = min(out['us_per_step_first']), min(out['us_per_step_second']), min(out['us_per_step_both'])
    out['sequential_over_concurrent'] = round((a + b)/c, 4)
    print(json.dumps(out))
    for d, _ in devs:
        d.close()


if __name__ == '__main__':
    main()
