#!/usr/bin/env python
"""Where does a stage of the dataflow launch (csrc/swe2d_flow.h) spend its time?  Needs a -DSWE_WAVE_TIMING build of the
library (THETIS_AMD_LIB): every block records the 100 MHz wall clock at the top of one stage (SWE_FLOW_TS_STAGE, default 7),
after its flag wait, after its gathers have landed, after the arithmetic and after its stores are drained.
   THETIS_AMD_LIB=variants/flow_wt.so python tools/flowtiming.py --nx 354 --ny 177"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--nx', type=int, default=354)
    ap.add_argument('--ny', type=int, default=177)
    ap.add_argument('--stages', type=int, default=12)
    args = ap.parse_args()
    import time
    import bench
    from thetis_amd.device import Swe2dDevice
    mesh, bath, uv, eta = bench.build_case(args.nx, args.ny)
    dev = Swe2dDevice(mesh, bath, bench.DT*1000.0/args.nx)
    dev.set_state(uv, eta)
    ends = [dev.n_cells]*args.stages
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        for _ in range(20):
            dev.solve_flow(ends)
        dev.synchronize()
    fn = dev.lib.swe2d_debug_read_wave_timing
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    nmax = 8192
    nb = (mesh.num_cells + 63)//64
    runs = []
    for rep in range(5):
        dev.solve_flow(ends)
        ts = np.zeros((6, nmax), dtype=np.uint64)
        dev._ck(fn(dev.h, ts.ctypes.data))
        t = ts[:5, :min(nb, nmax)].astype(np.int64)
        us = lambda x: float(np.mean(x))/100.0
        xcc = (ts[5, :min(nb, nmax)] >> np.uint64(32)).astype(np.int64)
        per = ((nb + 7)//8*8)//8
        expect = np.arange(min(nb, nmax))//per
        runs.append({'blocks': int(t.shape[1]), 'wait_us': us(t[1] - t[0]), 'gather_us': us(t[2] - t[1]), 'arith_us': us(t[3] - t[2]),
                     'drain_us': us(t[4] - t[3]), 'stage_us': us(t[4] - t[0]),
                     'stage_p10_p90_us': [float(np.percentile(t[4] - t[0], 10))/100.0, float(np.percentile(t[4] - t[0], 90))/100.0],
                     'front_spread_us': float(t[4].max() - t[4].min())/100.0,
                     'blocks_on_expected_xcd': float((xcc == expected_chunk(expect)).mean())})
    print(json.dumps({'n_cells': mesh.num_cells, 'stages': args.stages, 'runs': runs}, indent=1))
    dev.close()


def expected_chunk(c):
    return c


if __name__ == '__main__':
    main()
