#!/usr/bin/env python
"""Host cost of the stage-by-stage path: FlowSolver2d on the cfg 5 geometry (wetting-drying + Manning, tide on the deep boundary) stepped
(a) in batches without forcing updates and (b) with ``update_forcings`` before every stage (rungekutta.py:933-934), where the host
has to stay ahead of 30 us stage kernels.   python tools/forcingbench.py [--nx 707 --ny 354 --steps 300]"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from thetis_amd import Constant, Function, RectangleMesh, get_functionspace, solver2d       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--nx', type=int, default=707)
    ap.add_argument('--ny', type=int, default=354)
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--plain', action='store_true', help='no wetting-drying / Manning (the headline kernel)')
    args = ap.parse_args()
    lx, ly = 13800.0, 7200.0
    mesh2d = RectangleMesh(args.nx, args.ny, lx, ly)
    bathymetry = Function(get_functionspace(mesh2d, 'CG', 1)).interpolate(lambda x, y: x/2760.0 + (3.0 if args.plain else 0.0))
    s = solver2d.FlowSolver2d(mesh2d, bathymetry)
    o = s.options
    o.swe_timestepper_type = 'SSPRK33'
    o.swe_timestepper_options.use_automatic_timestep = False
    o.timestep = 0.15*707.0/args.nx
    o.simulation_end_time = 1e9
    o.simulation_export_time = 1e9
    o.no_exports = True
    if not args.plain:
        o.use_wetting_and_drying = True
        o.wetting_and_drying_alpha = Constant(0.4)
        o.manning_drag_coefficient = Constant(0.02)
    bnd_elev = Constant(0.0)
    s.bnd_functions['shallow_water'] = {2: {'elev': bnd_elev}}
    s.assign_initial_conditions(elev=Constant(0.0))
    ts = s.timestepper
    dev = ts.device
    ts.advance_steps(0.0, 200)
    dev.synchronize()
    t0 = time.perf_counter()
    ts.advance_steps(0.0, args.steps)
    dev.synchronize()
    batch = (time.perf_counter() - t0)/args.steps

    def forcing(t):
        bnd_elev.assign(-2.0*math.sin(2*math.pi*t/43200.0))
    for i in range(20):
        ts.advance(i*s.dt, forcing)
    dev.synchronize()
    t0 = time.perf_counter()
    marks = []
    for i in range(args.steps):
        ts.advance(i*s.dt, forcing)
        if i % 100 == 99:
            marks.append(time.perf_counter())
    t_enq = time.perf_counter() - t0
    seg = [round(1e4*(b - a), 1) for a, b in zip([t0] + marks[:-1], marks)]           # us per step in every 100 steps
    dev.synchronize()
    staged = (time.perf_counter() - t0)/args.steps
    print(json.dumps({'n_cells': mesh2d.num_cells, 'plain': args.plain, 'us_per_step_batched': 1e6*batch, 'us_per_step_with_update_forcings': 1e6*staged,
                      'host_us_per_step_enqueue': 1e6*t_enq/args.steps, 'host_us_per_step_by_100_steps': seg}))


if __name__ == '__main__':
    main()
