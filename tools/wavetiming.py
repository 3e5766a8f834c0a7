#!/usr/bin/env python
"""Where does a stage launch spend its time?  Needs a -DSWE_WAVE_TIMING build of the library (THETIS_AMD_LIB): every one-wave
workgroup records the 100 MHz wall clock at kernel entry, after its index loads, after all loads, after the arithmetic and
after its stores.   THETIS_AMD_LIB=variants/wt.so python tools/wavetiming.py --nx 125 --ny 500 [--stage 1]"""
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
    ap.add_argument('--nx', type=int, default=125)
    ap.add_argument('--ny', type=int, default=500)
    ap.add_argument('--stage', type=int, default=1)
    args = ap.parse_args()
    import bench
    from thetis_amd.device import Swe2dDevice
    mesh, bath, uv, eta = bench.build_case(args.nx, args.ny)
    dev = Swe2dDevice(mesh, bath, bench.DT)
    dev.set_state(uv, eta)
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        dev.advance(100)
        dev.synchronize()
    fn = dev.lib.swe2d_debug_read_wave_timing
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    nmax = 8192
    res = []
    for rep in range(5):
        dev.advance(3)
        for i in range(args.stage + 1):
            dev.solve_stage(i)
        ts = np.zeros((6, nmax), dtype=np.uint64)
        dev._ck(fn(dev.h, ts.ctypes.data))
        nw = min(nmax, ((mesh.num_cells + 63)//64 + 7)//8*8)
        nreal = (mesh.num_cells + 63)//64
        t = ts[:5, :nw].astype(np.int64)
        ok = (t[4] > 0)
        t = t[:, ok]
        ph = [(t[i + 1] - t[i]) for i in range(4)]
        hw = ts[5, :nw][ok]
        # gfx9 HW_ID: wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13]; XCC id in bits 32..35
        simd_key = ((hw >> np.uint64(4)) & np.uint64(0x3)) | (((hw >> np.uint64(8)) & np.uint64(0xff)) << np.uint64(2)) \
            | ((hw >> np.uint64(32)) << np.uint64(10))
        cu_key = simd_key >> np.uint64(2)
        _, per_simd = np.unique(simd_key, return_counts=True)
        _, per_cu = np.unique(cu_key, return_counts=True)
        _, inv = np.unique(simd_key, return_inverse=True)
        load = per_simd[inv]                       # waves sharing the SIMD of each wave
        tot = (t[4] - t[0])/100.0
        by_load = {int(l): float(tot[load == l].mean()) for l in np.unique(load)}
        # which waves finish last?  phases of the slowest 5 %, and how many of them own boundary cells
        nb_dev = np.asarray(dev._keep[2])                     # (N, k) neighbours in device numbering, < 0: boundary
        bnd_cell = (nb_dev < 0).any(axis=1)
        per = (nw + 7) >> 3
        blocks = np.nonzero(ok)[0]
        lb = (blocks & 7)*per + (blocks >> 3)
        has_bnd = np.array([bnd_cell[64*b:64*b + 64].any() for b in lb])
        slow = tot >= np.percentile(tot, 95)
        tail = {'n': int(slow.sum()), 'boundary_waves_in_tail': int(has_bnd[slow].sum()), 'boundary_waves_total': int(has_bnd.sum()),
                'index_us': float(ph[0][slow].mean())/100.0, 'loads_us': float(ph[1][slow].mean())/100.0,
                'arithmetic_us': float(ph[2][slow].mean())/100.0, 'stores_us': float(ph[3][slow].mean())/100.0,
                'boundary_wave_total_us': float(tot[has_bnd].mean()) if has_bnd.any() else None,
                'interior_wave_total_us': float(tot[~has_bnd].mean())}
        base = t[0].min()
        us = lambda x: float(x)/100.0
        ph = [(t[i + 1] - t[i]) for i in range(4)]
        res.append({'waves': int(ok.sum()), 'kernel_span_us': us(t[4].max() - base),
                    'start_spread_us': {'p50': us(np.percentile(t[0] - base, 50)), 'p90': us(np.percentile(t[0] - base, 90)),
                                        'max': us((t[0] - base).max())},
                    'wave_total_us': {'mean': us((t[4] - t[0]).mean()), 'p90': us(np.percentile(t[4] - t[0], 90))},
                    'simds_used': int(len(per_simd)), 'cus_used': int(len(per_cu)),
                    'waves_per_simd_hist': {int(k): int(v) for k, v in zip(*np.unique(per_simd, return_counts=True))},
                    'waves_per_cu_hist': {int(k): int(v) for k, v in zip(*np.unique(per_cu, return_counts=True))},
                    'wave_total_us_by_simd_load': by_load, 'slowest_5pct': tail, 'wave_total_max_us': float(tot.max()),
                    'index_loads_us': us(ph[0].mean()), 'gathers_and_own_loads_us': us(ph[1].mean()),
                    'arithmetic_us': us(ph[2].mean()), 'stores_us': us(ph[3].mean())})
    print(json.dumps({'n_cells': mesh.num_cells, 'stage': args.stage, 'runs': res}, indent=1))
    dev.close()


if __name__ == '__main__':
    main()
