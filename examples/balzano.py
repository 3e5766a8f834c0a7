#!/usr/bin/env python
"""
Tidal flat with wetting and drying: the scenario of the reference's examples/balzano/balzano.py (:32-104; Balzano 1998,
test 1) written against thetis_amd with the EXPLICIT stepper - a linearly sloping beach h = x/2760 on 13800 x 7200 m,
Manning friction 0.02, wetting-drying parameter alpha = 0.4, a 12 h tide of 2 m amplitude on the deep boundary.

The reference cannot combine use_wetting_and_drying with SSPRK33; what runs here is this build's own explicit nodal
formulation, which needs a time step well below the gravity-wave limit in the thin film.

    python examples/balzano.py [--nx 12 --ny 6 --dt 10 --hours 12]
"""
import argparse
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thetis_amd import Constant, Function, RectangleMesh, get_functionspace, solver2d       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--nx', type=int, default=12)
    ap.add_argument('--ny', type=int, default=6)
    ap.add_argument('--dt', type=float, default=10.0)
    ap.add_argument('--hours', type=float, default=12.0)
    ap.add_argument('--export', action='store_true', help='write VTK files to outputs/')
    args = ap.parse_args()
    lx, ly = 13800.0, 7200.0
    mesh2d = RectangleMesh(args.nx, args.ny, lx, ly)
    bathymetry = Function(get_functionspace(mesh2d, 'CG', 1), name='bathymetry').interpolate(lambda x, y: x/2760.0)
    solver_obj = solver2d.FlowSolver2d(mesh2d, bathymetry)
    options = solver_obj.options
    options.swe_timestepper_type = 'SSPRK33'
    options.swe_timestepper_options.use_automatic_timestep = False
    options.timestep = args.dt
    options.simulation_end_time = 3600.0*args.hours
    options.simulation_export_time = 1800.0
    options.use_wetting_and_drying = True
    options.wetting_and_drying_alpha = Constant(0.4)
    options.manning_drag_coefficient = Constant(0.02)
    options.check_volume_conservation_2d = True
    options.fields_to_export = ['uv_2d', 'elev_2d']
    options.no_exports = not args.export
    bnd_elev = Constant(0.0)
    solver_obj.bnd_functions['shallow_water'] = {2: {'elev': bnd_elev}}
    solver_obj.assign_initial_conditions(elev=Constant(0.0))
    solver_obj.iterate(update_forcings=lambda t: bnd_elev.assign(-2.0*math.sin(2*math.pi*t/43200.0)))
    eta = solver_obj.fields.elev_2d.cell_node_values()
    uv = solver_obj.fields.uv_2d.cell_node_values()
    depth = bathymetry.dat.data_ro[mesh2d.cells] + eta
    print('finite {:}  min(h + eta) {:.3f} m  max |u| {:.3f} m/s'.format(
        bool(np.isfinite(eta).all() and np.isfinite(uv).all()), depth.min(), np.abs(uv).max()))


if __name__ == '__main__':
    main()
