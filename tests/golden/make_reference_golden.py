#!/usr/bin/env python
"""
Residual-level golden vectors FROM THE REFERENCE: run this where ``from thetis import *`` works (a Firedrake installation with
thetisproject/thetis on the path - e.g. the docker image the reference's CI uses, firedrakeproject/firedrake-vanilla-default).

    python tests/golden/make_reference_golden.py            # writes tests/golden/reference_vectors.json

IT CANNOT BE RUN IN THE CONTAINER THIS REPOSITORY IS BUILT IN (no Firedrake, no network: SURVEY.md section 8c), and it has never
been run: every Firedrake / Thetis call below is written from the reference's sources - the call sites are cited - and is
[FD-assumed] until someone does.  It is committed because it is the only route from "parity unpinned" to pinned: the file it
writes is consumed by ``tests/test_reference_golden.py`` (``-m gpu``), which compares the HIP path with these vectors at 1e-12
(tendency, one step) / 1e-11 (ten steps) and SKIPS LOUDLY while the file is absent.  The consumer's own plumbing - the file
format, the Firedrake-shaped topology tables through ``thetis_amd/firedrake_adapter.py``, the DG dof permutation - is tested
without Firedrake by vectors the oracle writes in the same format (``test_the_consumer_reads_what_the_generator_writes``).

What one run settles (the [FD-assumed] list of oracle/swe2d_oracle.py's header and DESIGN.md section 3):
  * facet quadrature (2-point Gauss-Legendre for degree 3) and cell quadrature - case 'manning' has a non-polynomial cell
    integrand, where the 6-point degree-4 rule this build chose and whatever FIAT gives Firedrake for degree 3 differ;
  * node order of DG-P1 / DQ-1 dofs, FIAT local facet numbering, orientation handling (thetis_amd/firedrake_adapter.py);
  * the 'left' diagonal and the markers 1-4 of RectangleMesh;
  * signs and '+'/'-' conventions of every term (cases 'channel', 'optional_terms', 'open_boundaries').

For every case: the mesh's topology tables, bathymetry, options, the initial state, then
  tendency   = ERKGenericShuOsher.tendency after ONE ``solver.solve()``          (thetis/rungekutta.py:919-924, :938-939)
  uv_1 ...   = the solution after ``advance`` once / ten times                    (thetis/rungekutta.py:930-952)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from reference_vector_format import write_cases      # noqa: E402


def mesh_tables(mesh2d, P1DG):
    """the arguments of thetis_amd.firedrake_adapter.swe2d_mesh_arrays, as INTEGRATION.md section 2 takes them from a mesh"""
    return {'coords': np.array(mesh2d.coordinates.dat.data_ro),
            'cell_vertices': np.array(mesh2d.coordinates.cell_node_map().values),
            'int_facet_cell': np.array(mesh2d.interior_facets.facet_cell),
            'int_local_facet': np.array(mesh2d.interior_facets.local_facet_dat.data_ro),
            'ext_facet_cell': np.array(mesh2d.exterior_facets.facet_cell).reshape(-1),
            'ext_local_facet': np.array(mesh2d.exterior_facets.local_facet_dat.data_ro).reshape(-1),
            'ext_markers': np.array(mesh2d.exterior_facets.markers),
            'dg_cell_nodes': np.array(P1DG.cell_node_map().values)}


def run_case(name, mesh2d, bathymetry_expr, elev_expr, uv_expr, dt, options=None, scalars=None, fields_dg=None, bnd=None):
    """one FlowSolver2d with SSPRK33 on dg-dg P1; returns the case dict of reference_vector_format"""
    from thetis import (Constant, Function, as_vector, get_functionspace, solver2d)          # noqa: F401
    options = dict(options or {})
    P1_2d = get_functionspace(mesh2d, 'CG', 1)
    bathymetry_2d = Function(P1_2d, name='Bathymetry')
    bathymetry_2d.interpolate(bathymetry_expr)
    solver_obj = solver2d.FlowSolver2d(mesh2d, bathymetry_2d)
    o = solver_obj.options
    o.element_family = 'dg-dg'
    o.polynomial_degree = 1
    o.swe_timestepper_type = 'SSPRK33'
    o.no_exports = True
    o.simulation_export_time = 1e9
    o.simulation_end_time = 20*dt
    o.horizontal_velocity_scale = Constant(1.0)
    if hasattr(o.swe_timestepper_options, 'use_automatic_timestep'):
        o.swe_timestepper_options.use_automatic_timestep = False
    o.timestep = dt
    o.use_nonlinear_equations = bool(options.get('use_nonlinear_equations', True))
    o.use_lax_friedrichs_velocity = bool(options.get('use_lax_friedrichs_velocity', True))
    o.lax_friedrichs_velocity_scaling_factor = Constant(float(options.get('lax_friedrichs_velocity_scaling_factor', 1.0)))
    for key, value in (scalars or {}).items():                     # Constants: linear / quadratic / Manning drag
        setattr(o, key, Constant(float(value)))
    solver_obj.create_function_spaces()
    P1DG = solver_obj.function_spaces.P1DG_2d
    # optional DG-P1 fields (thetis/solver2d.py:547-557: the dict handed to the equation)
    dg_fields = {}
    for key, (attr, expr) in (fields_dg or {}).items():
        space = solver_obj.function_spaces.P1DGv_2d if key in ('momentum_source', 'wind_stress') else P1DG
        f = Function(space, name=key)
        f.interpolate(expr)
        setattr(o, attr, f)
        dg_fields[key] = np.array(f.dat.data_ro)
    # constant boundary values (thetis/shallowwater_eq.py:232-272)
    bnd_out = {}
    if bnd:
        solver_obj.bnd_functions['shallow_water'] = {}
        for marker, funcs in bnd.items():
            solver_obj.bnd_functions['shallow_water'][int(marker)] = {
                k: (Constant(tuple(v)) if k == 'uv' else Constant(float(v))) for k, v in funcs.items()}
            bnd_out[int(marker)] = dict(funcs)
    solver_obj.create_equations()
    solver_obj.assign_initial_conditions(elev=elev_expr, uv=uv_expr)        # L2 projections into DG-P1, :763-766
    ts = solver_obj.timestepper                                            # rungekutta.SSPRK33 (:699)
    uv_2d, elev_2d = solver_obj.fields.solution_2d.subfunctions
    case = {'name': name, 'mesh': mesh_tables(mesh2d, P1DG), 'bathymetry': np.array(bathymetry_2d.dat.data_ro), 'dt': dt,
            'options': {'use_nonlinear_equations': o.use_nonlinear_equations, 'use_lax_friedrichs_velocity': o.use_lax_friedrichs_velocity,
                        'lax_friedrichs_velocity_scaling_factor': float(options.get('lax_friedrichs_velocity_scaling_factor', 1.0))},
            'scalars': dict(scalars or {}), 'fields_dg': dg_fields, 'bnd': bnd_out,
            'uv0': np.array(uv_2d.dat.data_ro), 'elev0': np.array(elev_2d.dat.data_ro)}
    # ---- the tendency of the initial state: one solve of  M k = dt R(U0)  (rungekutta.py:919-924, 938-939)
    ts.solver.solve()
    k_uv, k_elev = ts.tendency.subfunctions
    case['tendency_uv'], case['tendency_elev'] = np.array(k_uv.dat.data_ro), np.array(k_elev.dat.data_ro)
    # ---- one and ten steps
    t = 0.0
    for i in range(10):
        ts.advance(t)
        t += dt
        if i == 0:
            case['uv_1'], case['elev_1'] = np.array(uv_2d.dat.data_ro), np.array(elev_2d.dat.data_ro)
    case['uv_10'], case['elev_10'] = np.array(uv_2d.dat.data_ro), np.array(elev_2d.dat.data_ro)
    return case


def main():
    from thetis import RectangleMesh, SpatialCoordinate, as_vector, cos, exp, pi, sin
    import firedrake
    cases = []
    # ---- 'channel': examples/channel2d geometry (80 x 3 x 2 triangles, sloping bathymetry, closed walls), smooth initial state
    lx, ly = 100e3, 3750.0
    m = RectangleMesh(80, 3, lx, ly)
    x, y = SpatialCoordinate(m)
    cases.append(run_case('channel', m, 20.0 - 15.0*x/lx, 0.5*exp(-((x - 0.3*lx)/8e3)**2), as_vector((0.05*sin(pi*x/lx), 0.0)), 2.0))
    cases.append(run_case('channel_linear_nolf', m, 20.0 - 15.0*x/lx, 0.5*exp(-((x - 0.3*lx)/8e3)**2), as_vector((0.05*sin(pi*x/lx), 0.0)), 2.0,
                          options={'use_nonlinear_equations': False, 'use_lax_friedrichs_velocity': False}))
    # ---- 'quad': BASELINE cfg 1 as written (40 x 25 quadrilaterals on the unit rectangle is the size; 10 x 4 here), DQ-1
    m = RectangleMesh(10, 4, 1.0, 0.4, quadrilateral=True)
    x, y = SpatialCoordinate(m)
    cases.append(run_case('quad', m, 1.0 + 0.0*x, 0.01*cos(pi*x)*cos(2.5*pi*y), as_vector((0.01*sin(pi*x), 0.01*sin(2.5*pi*y))), 1e-3))
    # ---- 'optional_terms': Coriolis, atmospheric pressure, wind stress, momentum and volume sources, linear drag (all polynomial
    #      cell integrands), on a small basin with varying bathymetry
    lx, ly = 10e3, 7.5e3
    m = RectangleMesh(8, 6, lx, ly)
    x, y = SpatialCoordinate(m)
    cases.append(run_case('optional_terms', m, 10.0 + 3.0*x/lx + 2.0*y/ly, 0.1*cos(pi*x/lx)*cos(pi*y/ly),
                          as_vector((0.1*sin(pi*x/lx), -0.05*sin(pi*y/ly))), 5.0,
                          scalars={'linear_drag_coefficient': 1e-3},
                          fields_dg={'coriolis': ('coriolis_frequency', 1e-4 + 2e-8*y),
                                     'atmospheric_pressure': ('atmospheric_pressure', 50.0*x/lx),
                                     'wind_stress': ('wind_stress', as_vector((0.1*y/ly, 0.05*x/lx))),
                                     'momentum_source': ('momentum_source_2d', as_vector((1e-5*x/lx, -2e-5*y/ly))),
                                     'volume_source': ('volume_source_2d', 1e-6*x/lx)}))
    # ---- 'manning': the non-polynomial cell integrand (C_D = g mu^2 / H^(1/3), shallowwater_eq.py:685-700): pins the cell rule
    cases.append(run_case('manning', m, 4.0 + 3.0*x/lx + 2.0*y/ly, 0.1*cos(pi*x/lx)*cos(pi*y/ly),
                          as_vector((0.4*sin(pi*x/lx) + 0.1, -0.3*sin(pi*y/ly))), 5.0, scalars={'manning_drag_coefficient': 0.03}))
    cases.append(run_case('quadratic_drag', m, 4.0 + 3.0*x/lx + 2.0*y/ly, 0.1*cos(pi*x/lx)*cos(pi*y/ly),
                          as_vector((0.4*sin(pi*x/lx) + 0.1, -0.3*sin(pi*y/ly))), 5.0, scalars={'quadratic_drag_coefficient': 2.5e-3}))
    # ---- 'open_boundaries': the four Riemann forms with constant data (shallowwater_eq.py:367-375, :431-442, :498-509)
    cases.append(run_case('open_boundaries', m, 10.0 + 3.0*x/lx, 0.1*cos(pi*x/lx),
                          as_vector((0.1 + 0.05*sin(pi*y/ly), 0.02*cos(pi*x/lx))), 5.0,
                          bnd={1: {'elev': 0.05, 'un': -0.1}, 2: {'elev': -0.02}, 3: {'flux': 500.0}, 4: {'uv': [0.05, -0.02]}}))
    import thetis
    meta = {'generator': 'tests/golden/make_reference_golden.py', 'firedrake': getattr(firedrake, '__version__', '?'),
            'thetis': getattr(thetis, '__version__', '?'), 'numpy': np.__version__}
    out = os.path.join(HERE, 'reference_vectors.json')
    write_cases(out, cases, meta)
    print('wrote {:} ({:d} cases)'.format(out, len(cases)))


if __name__ == '__main__':
    main()
