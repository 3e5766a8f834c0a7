#!/usr/bin/env python
"""
Solid-body rotation of the LeVeque tracer field (bell + cone + slotted cylinder): the scenario of the reference's
demos/demo_2d_tracer.py (:19-137) written against thetis_amd - a quadrilateral mesh, tracer-only mode, SSPRK33,
optionally with the vertex-based limiter.

    python examples/tracer2d.py [--n 40 --limiter --revolutions 1]
"""
import argparse
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thetis_amd import Constant, Function, UnitSquareMesh, get_functionspace, solver2d       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=40)
    ap.add_argument('--limiter', action='store_true')
    ap.add_argument('--revolutions', type=float, default=1.0)
    ap.add_argument('--export', action='store_true', help='write VTK files to outputs/')
    args = ap.parse_args()
    mesh2d = UnitSquareMesh(args.n, args.n, quadrilateral=True)
    P1_2d = get_functionspace(mesh2d, 'CG', 1)
    bathymetry2d = Function(P1_2d).assign(1.0)
    solver_obj = solver2d.FlowSolver2d(mesh2d, bathymetry2d)
    options = solver_obj.options
    options.add_tracer_2d('tracer_2d', 'Depth averaged tracer', 'Tracer2d', source=None, diffusivity=None)
    options.tracer_only = True
    options.fields_to_export = ['tracer_2d']
    options.no_exports = not args.export
    options.tracer_timestepper_type = 'SSPRK33'
    options.timestep = math.pi/300.0*40.0/args.n
    options.simulation_end_time = 2*math.pi*args.revolutions
    options.simulation_export_time = math.pi/15.0
    options.tracer_timestepper_options.use_automatic_timestep = False
    options.use_lax_friedrichs_tracer = False
    options.use_limiter_for_tracers = args.limiter
    solver_obj.bnd_functions['tracer_2d'] = {'on_boundary': {'value': Constant(1.0)}}

    def q0(x, y):
        bell = 0.25*(1 + np.cos(np.pi*np.minimum(np.sqrt((x - 0.25)**2 + (y - 0.5)**2)/0.15, 1.0)))
        cone = 1.0 - np.minimum(np.sqrt((x - 0.5)**2 + (y - 0.25)**2)/0.15, 1.0)
        cyl = np.where(np.sqrt((x - 0.5)**2 + (y - 0.75)**2) < 0.15,
                       np.where((x > 0.475) & (x < 0.525) & (y < 0.85), 0.0, 1.0), 0.0)
        return 1.0 + bell + cone + cyl
    q_init = Function(P1_2d).interpolate(q0)
    solver_obj.assign_initial_conditions(uv=lambda x, y: (0.5 - y, x - 0.5), tracer_2d=q_init)
    solver_obj.iterate()
    q = solver_obj.fields.tracer_2d.cell_node_values()
    q_i = q_init.cell_node_values()
    # relative L2 error after the rotation(s) (demo_2d_tracer.py:131-137); nodal root-mean-square on the uniform mesh
    err = math.sqrt(((q - q_i)**2).mean())/math.sqrt((q_i**2).mean())
    print('relative L2 error {:.4f}  min {:.4f} max {:.4f}'.format(err, q.min(), q.max()))
    return err, q.min(), q.max()


if __name__ == '__main__':
    main()
