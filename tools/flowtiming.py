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
    # neighbour blocks of every block (device numbering; the default flow order is the device numbering)
    nb_dev = np.asarray(dev._keep[2])
    cell_blk = np.arange(mesh.num_cells)//64
    pairs = set()
    for f in range(3):
        ok = nb_dev[:, f] >= 0
        a, b = cell_blk[ok], nb_dev[ok, f]//64
        d = a != b
        pairs.update(zip(a[d].tolist(), b[d].tolist()))
    nbrs = [[] for _ in range(nb)]
    for a, b in pairs:
        nbrs[a].append(b)
    for rep in range(5):
        dev.solve_flow(ends)
        ts = np.zeros((6, nmax), dtype=np.uint64)
        dev._ck(fn(dev.h, ts.ctypes.data))
        m = min(nb, nmax)
        t = ts[:5, :m].astype(np.int64)
        pub = (ts[5, :m] & np.uint64(0xffffffffff)).astype(np.int64)      # previous stage's granules issued
        xcc = (ts[5, :m] >> np.uint64(56)).astype(np.int64)
        t = t & 0xffffffffff                      # the clock as stored with the previous publish: 40 bits
        t48 = t
        key = ((ts[5, :m] >> np.uint64(40)) & np.uint64(0x3fff)).astype(np.int64)      # simd | cu << 2 | sh << 6 | se << 7 | wave slot << 10
        us = lambda x: float(np.mean(x))/100.0
        # hop latency: from the moment the LAST neighbour issued its granules (or this block started polling, whichever is later)
        # to the moment this block's polling pass saw them all
        hop, hop_same, hop_cross = [], [], []
        for b in range(m):
            ns = [a for a in nbrs[b] if a < m]
            if not ns:
                continue
            last = max(ns, key=lambda a: pub[a])
            lat = t48[1, b] - max(pub[last], t48[0, b])
            hop.append(lat)
            (hop_same if xcc[last] == xcc[b] else hop_cross).append(lat)
        runs.append({'blocks': int(m), 'wait_us': us(t[1] - t[0]), 'arith_us': us(t[3] - t[2]),
                     'publish_us': us(t[4] - t[3]), 'stage_us': us(t[4] - t[0]),
                     'stage_p10_p90_us': [float(np.percentile(t[4] - t[0], 10))/100.0, float(np.percentile(t[4] - t[0], 90))/100.0],
                     'front_spread_us': float(t[4].max() - t[4].min())/100.0,
                     'hop_us': us(hop), 'hop_p10_p90_us': [float(np.percentile(hop, 10))/100.0, float(np.percentile(hop, 90))/100.0],
                     'hop_same_xcd_us': us(hop_same) if hop_same else None, 'hop_cross_xcd_us': us(hop_cross) if hop_cross else None,
                     'last_neighbour_on_same_xcd': float(len(hop_same))/max(1, len(hop)),
                     'waiting_before_last_publish_frac': float(np.mean([t48[0, b] < pub[max([a for a in nbrs[b] if a < m], key=lambda a: pub[a])]
                                                                         for b in range(m) if [a for a in nbrs[b] if a < m]]))})
        # ---- SIMD mates: blocks on the same SIMD (XCC, SE, SH, CU, SIMD).  Do they compute at the same time (each then takes twice
        #      as long: one f64 pipe), one right after the other (in phase: the second waits for the pipe), or apart?
        simd = xcc*1024 + (key & 0x3ff)
        by_simd = {}
        for b in range(m):
            by_simd.setdefault(int(simd[b]), []).append(b)
        period = float(np.median(t[4] - pub))                    # previous publish -> this publish
        sep, overlap_frac, dist = [], [], []
        for blocks in by_simd.values():
            for i in range(len(blocks)):
                for j in range(i + 1, len(blocks)):
                    a, b = blocks[i], blocks[j]
                    sep.append(abs(int(t[2, a]) - int(t[2, b]))/100.0)               # between the starts of their arithmetic
                    lo, hi = max(t[2, a], t[2, b]), min(t[3, a], t[3, b])
                    overlap_frac.append(max(0.0, float(hi - lo))/max(1.0, float(min(t[3, a] - t[2, a], t[3, b] - t[2, b]))))
                    dist.append(abs(a - b))
        sizes = np.bincount([len(v) for v in by_simd.values()])
        runs[-1]['mates'] = {'simds_used': len(by_simd), 'blocks_per_simd_histogram': sizes.tolist(), 'pairs': len(sep),
                             'stage_period_us': period/100.0,
                             'start_separation_us_p10_p50_p90': [float(np.percentile(sep, q)) for q in (10, 50, 90)] if sep else None,
                             'arith_overlap_fraction_p10_p50_p90': [float(np.percentile(overlap_frac, q)) for q in (10, 50, 90)] if sep else None,
                             'pairs_overlapping_more_than_half': float(np.mean(np.asarray(overlap_frac) > 0.5)) if sep else None,
                             'block_index_distance_p10_p50_p90': [float(np.percentile(dist, q)) for q in (10, 50, 90)] if sep else None,
                             'arith_us_alone_vs_overlapping': [float(np.mean((t[3] - t[2])[[b for v in by_simd.values() if len(v) == 1 for b in v]]))/100.0
                                                               if any(len(v) == 1 for v in by_simd.values()) else None,
                                                               float(np.mean((t[3] - t[2])[[b for v in by_simd.values() if len(v) > 1 for b in v]]))/100.0
                                                               if any(len(v) > 1 for v in by_simd.values()) else None]}
    print(json.dumps({'n_cells': mesh.num_cells, 'stages': args.stages, 'runs': runs}, indent=1))
    dev.close()


if __name__ == '__main__':
    main()
