#!/usr/bin/env python
"""swe2d_advance as two chains of half-launches against single launches, eager and replayed from a HIP graph (the chains cost
   a dozen more runtime calls per step: events and waits).
   python tools/chainbench.py --nx 1000 --ny 500 [--window 8] [--reps 12]"""
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
    ap.add_argument('--window', type=int, default=8, help='steps per graph')
    ap.add_argument('--reps', type=int, default=12)
    args = ap.parse_args()
    import torch
    import bench
    from thetis_amd.device import Swe2dDevice
    os.environ['THETIS_AMD_FLOW'] = '0'
    mesh, bath, uv, eta = bench.build_case(args.nx, args.ny)
    dt = bench.DT*min(1.0, 1000.0/args.nx, 500.0/args.ny)
    dev = Swe2dDevice(mesh, bath, dt, chains=3*args.window)
    dev.set_state(uv, eta)
    stream = torch.cuda.Stream()
    dev.set_stream(stream.cuda_stream)
    out = {'n_cells': mesh.num_cells, 'window': args.window}

    def timed(fn, n_steps):
        best = 1e9
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(stream):
                a.record(stream)
                fn()
                b.record(stream)
            b.synchronize()
            best = min(best, a.elapsed_time(b)*1e3/n_steps)
        return round(best, 2)

    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:
        dev.advance(100)
        dev.synchronize()
    for chains in ('0', '1'):
        os.environ['THETIS_AMD_CHAINS'] = chains
        n = args.window*args.reps
        out['eager chains=' + chains] = timed(lambda: dev.advance(n), n)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream, capture_error_mode='thread_local'):
            dev.advance(args.window)
        out['graph chains=' + chains] = timed(lambda: [g.replay() for _ in range(args.reps)], n)
        del g
    out['vol'] = dev.diagnostics()[2]
    print(json.dumps(out))
    dev.close()


if __name__ == '__main__':
    main()
