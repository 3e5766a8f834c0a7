#!/usr/bin/env python
"""
Dam-break in a closed channel with a sloping bed: the scenario of the reference's examples/channel2d/channel2d.py
(:21-63) written against thetis_amd.  With the reference installed, the same script runs there after replacing the
import by ``from thetis import *`` and the three callables by UFL expressions of ``SpatialCoordinate(mesh2d)``.

    python examples/channel2d.py [--nx 80 --ny 3 --t-end 500]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thetis_amd import Constant, Function, RectangleMesh, get_functionspace, solver2d       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--nx', type=int, default=80)
    ap.add_argument('--ny', type=int, default=3)
    ap.add_argument('--t-end', type=float, default=500.0)
    ap.add_argument('--export', action='store_true', help='write VTK files to outputs/')
    args = ap.parse_args()
    lx, ly = 100e3, 3750.0
    mesh2d = RectangleMesh(args.nx, args.ny, lx, ly)
    P1_2d = get_functionspace(mesh2d, 'CG', 1)
    bathymetry_2d = Function(P1_2d, name='Bathymetry').interpolate(lambda x, y: 20.0 + (5.0 - 20.0)*x/lx)

    solver_obj = solver2d.FlowSolver2d(mesh2d, bathymetry_2d)
    options = solver_obj.options
    options.simulation_export_time = 100.0
    options.simulation_end_time = args.t_end
    options.horizontal_velocity_scale = Constant(6.0)        # used by the automatic CFL time step
    options.check_volume_conservation_2d = True
    options.fields_to_export = ['uv_2d', 'elev_2d']
    options.no_exports = not args.export
    options.swe_timestepper_type = 'SSPRK33'                 # the explicit, device-resident stepper

    elev_init = Function(P1_2d).interpolate(lambda x, y: np.where(x < 30e3, 6.0*(1 - x/30e3), 0.0))
    solver_obj.assign_initial_conditions(elev=elev_init)
    solver_obj.iterate()
    vol, rel = solver_obj.callbacks['export']['volume2d']()
    print('volume {:.6e}  relative change {:.2e}'.format(vol, rel))


if __name__ == '__main__':
    main()
