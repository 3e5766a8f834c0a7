"""FlowSolver2d surface on the GPU: examples/channel2d (regression length) and forcing/boundary plumbing."""
import math

import numpy as np
import pytest

from helpers import make_ref, rel_linf
from thetis_amd import Constant, Function, RectangleMesh, get_functionspace, solver2d

pytestmark = pytest.mark.gpu


def _channel2d_solver(dt=2.0, t_end=500.0):
    # examples/channel2d/channel2d.py:21-61 with THETIS_REGRESSION_TEST (t_end = 5 exports)
    lx, ly, nx, ny = 100e3, 3750.0, 80, 3
    mesh2d = RectangleMesh(nx, ny, lx, ly)
    P1_2d = get_functionspace(mesh2d, 'CG', 1)
    bathymetry_2d = Function(P1_2d, name='Bathymetry')
    bathymetry_2d.interpolate(lambda x, y: 20.0 + (5.0 - 20.0)*x/lx)
    solver_obj = solver2d.FlowSolver2d(mesh2d, bathymetry_2d)
    options = solver_obj.options
    options.simulation_export_time = 100.0
    options.simulation_end_time = t_end
    options.horizontal_velocity_scale = Constant(6.0)
    options.check_volume_conservation_2d = True
    options.fields_to_export = ['uv_2d', 'elev_2d']
    options.swe_timestepper_type = 'SSPRK33'
    options.swe_timestepper_options.use_automatic_timestep = False
    options.timestep = dt
    elev_init = Function(P1_2d)
    elev_init.interpolate(lambda x, y: np.where(x < 30e3, 6.0*(1 - x/30e3), 0.0))
    return solver_obj, mesh2d, bathymetry_2d, elev_init


def test_channel2d_example_matches_cpu_restatement(hip_lib, ref_so, capsys):
    solver_obj, mesh, bath, elev_init = _channel2d_solver()
    solver_obj.assign_initial_conditions(elev=elev_init)
    times = [t for t in solver_obj.create_iterator()]
    # generator yields the time before it is incremented (solver2d.py:1122-1127)
    assert times[0] == 0 and math.isclose(times[-1], 498.0) and len(times) == 250
    assert solver_obj.iteration == 250 and solver_obj.i_export == 5 and math.isclose(solver_obj.simulation_time, 500.0)
    uv, eta = solver_obj.fields.solution_2d.subfunctions
    ref = make_ref(mesh, bath.dat.data_ro)
    eta0 = elev_init.dat.data_ro[mesh.cells]
    u_r, e_r = ref.advance(np.zeros((mesh.num_cells, 3, 2)), eta0, 2.0, 250)
    assert rel_linf(eta.dat.data_ro.reshape(-1, 3), e_r) < 1e-10
    assert rel_linf(uv.dat.data_ro.reshape(-1, 3, 2), u_r) < 1e-10
    out = capsys.readouterr().out
    assert 'Using time integrator: SSPRK33' in out and 'eta norm' in out
    rel = [float(l.split()[-1]) for l in out.splitlines() if l.startswith('volume2d rel. error')]
    assert len(rel) == 6 and max(abs(r) for r in rel) < 1e-12      # reference bar: test_closed_channel.py:77-78
    # print_state line format (solver2d.py:931-970)
    line = [l for l in out.splitlines() if l.strip().startswith('5   250')][0]
    assert line.split()[2] == '500.00'


def test_automatic_timestep_runs(hip_lib):
    solver_obj, mesh, bath, elev_init = _channel2d_solver(t_end=100.0)
    solver_obj.options.swe_timestepper_options.use_automatic_timestep = True
    solver_obj.assign_initial_conditions(elev=elev_init)
    # CFL dt with alpha = 0.05 (solver2d.py:214): min over nodes of dx/(sqrt(g h) + 6)
    assert math.isclose(solver_obj.dt, 0.05*math.sqrt(1250.0*1250.0/2)/(math.sqrt(9.81*20.0) + 6.0), rel_tol=1e-3)  # L2 projection of a non-polynomial integrand
    solver_obj.iterate()
    d = solver_obj.timestepper.diagnostics()
    assert np.isfinite(d).all()


def test_update_forcings_and_open_boundary(hip_lib, ref_so):
    """Time-dependent elevation on marker 2 through ``update_forcings`` at t + c_i dt (rungekutta.py:933-934)."""
    lx, ly = 13800.0, 7200.0
    mesh2d = RectangleMesh(12, 6, lx, ly)
    bath = Function(get_functionspace(mesh2d, 'CG', 1)).interpolate(lambda x, y: 5.0 + x/2760.0)
    s = solver2d.FlowSolver2d(mesh2d, bath)
    o = s.options
    o.swe_timestepper_type = 'SSPRK33'
    o.swe_timestepper_options.use_automatic_timestep = False
    o.timestep = 5.0
    o.simulation_end_time = 100.0
    o.simulation_export_time = 50.0
    o.manning_drag_coefficient = Constant(0.02)
    elev_bc = Constant(0.0)
    s.bnd_functions['shallow_water'] = {2: {'elev': elev_bc}}
    calls = []

    def update_forcings(t):
        calls.append(t)
        elev_bc.assign(-0.5*math.sin(2*math.pi*t/1000.0))
    s.assign_initial_conditions(elev=Constant(0.0))
    s.iterate(update_forcings=update_forcings)
    assert calls[:3] == [0.0, 5.0, 2.5]                      # c = (0, 1, 1/2)
    uv, eta = s.fields.solution_2d.subfunctions
    # same sequence with the numpy oracle (literal forms incl. Manning quadrature and the Riemann boundary)
    from helpers import make_oracle
    val = {'v': 0.0}
    orc = make_oracle(mesh2d, bath.dat.data_ro, manning_drag_coefficient=0.02,
                      bnd_conditions={2: {'elev': lambda t: val['v']}})

    def uf(t):
        val['v'] = -0.5*math.sin(2*math.pi*t/1000.0)
    u_o = np.zeros((mesh2d.num_cells, 3, 2))
    e_o = np.zeros((mesh2d.num_cells, 3))
    for k in range(20):
        u_o, e_o = orc.ssprk33_step(u_o, e_o, 5.0, t=5.0*k, update_forcings=uf)
    assert rel_linf(eta.dat.data_ro.reshape(-1, 3), e_o) < 1e-10
    assert rel_linf(uv.dat.data_ro.reshape(-1, 3, 2), u_o) < 1e-10


def test_unsupported_configurations_fail_loudly(hip_lib):
    solver_obj, mesh, bath, elev_init = _channel2d_solver()
    solver_obj.options.swe_timestepper_type = 'CrankNicolson'
    with pytest.raises(NotImplementedError):
        solver_obj.assign_initial_conditions(elev=elev_init)
    s2, *_ = _channel2d_solver()
    s2.options.use_wetting_and_drying = True        # needs the nonlinear equations
    s2.options.use_nonlinear_equations = False
    with pytest.raises(Exception):
        s2.assign_initial_conditions(elev=elev_init)
    s3, *_ = _channel2d_solver()
    s3.options.horizontal_viscosity = Function(get_functionspace(mesh, 'DG', 1)).assign(10.0)    # discontinuous viscosity
    with pytest.raises(NotImplementedError):
        s3.assign_initial_conditions(elev=elev_init)
    s5, *_ = _channel2d_solver()
    s5.options.horizontal_viscosity = Constant(10.0)           # supported since the SIPG pass kernel exists
    s5.assign_initial_conditions(elev=elev_init)
    s4, *_ = _channel2d_solver()
    s4.bnd_functions['shallow_water'] = {1: {'temperature': Constant(1.0)}}
    with pytest.raises(Exception):
        s4.assign_initial_conditions(elev=elev_init)


def test_rossby_soliton_reference_criterion(hip_lib):
    """test/swe2d/test_rossby_wave.py::test_convergence[SSPRK33-dg-dg] through FlowSolver2d on the device."""
    import rossby
    import thetis_amd
    g_saved = float(thetis_amd.physical_constants['g_grav'])
    thetis_amd.physical_constants['g_grav'].assign(1.0)
    try:
        res = []
        for level in (24, 48):
            mesh2d = rossby.rossby_mesh(level)
            P1_2d = get_functionspace(mesh2d, 'CG', 1)
            bathymetry2d = Function(P1_2d).assign(1.0)
            solver_obj = solver2d.FlowSolver2d(mesh2d, bathymetry2d)
            options = solver_obj.options
            options.swe_timestepper_type = 'SSPRK33'
            options.element_family = 'dg-dg'
            options.swe_timestepper_options.use_automatic_timestep = False
            options.timestep = 0.96/level
            options.simulation_export_time = 5.0
            options.simulation_end_time = 30.0
            options.horizontal_viscosity = None
            solver_obj.create_function_spaces()
            options.coriolis_frequency = Function(solver_obj.function_spaces.P1_2d).interpolate(lambda x, y: y)
            options.no_exports = True
            solver_obj.create_equations()
            for tag in mesh2d.boundary_markers:
                solver_obj.bnd_functions['shallow_water'][tag] = {'uv': Constant((0., 0.))}
            uv_a = Function(solver_obj.function_spaces.U_2d).interpolate(rossby.asymptotic_uv)
            elev_a = Function(solver_obj.function_spaces.H_2d).interpolate(rossby.asymptotic_elev)
            solver_obj.assign_initial_conditions(uv=uv_a, elev=elev_a)
            solver_obj.iterate()
            eta = solver_obj.fields.elev_2d.cell_node_values()
            res.append(rossby.metrics(mesh2d.cell_xy(), eta))
        rossby.check_convergence(res[0], res[1])
    finally:
        thetis_amd.physical_constants['g_grav'].assign(g_saved)


def test_export_and_restart(hip_lib, tmp_path):
    """VTK export + checkpoint restart (solver2d.py:704-730,820-921): a run continued from export 2 equals the uninterrupted run."""
    def make(outdir):
        s, mesh, bath, elev_init = _channel2d_solver(t_end=400.0)
        s.options.no_exports = False
        s.options.output_directory = str(outdir)
        s.options.fields_to_export = ['elev_2d', 'uv_2d']
        s.options.fields_to_export_hdf5 = ['elev_2d', 'uv_2d']
        return s, elev_init
    s1, elev_init = make(tmp_path/'a')
    s1.assign_initial_conditions(elev=elev_init)
    s1.iterate()
    e_full = s1.fields.elev_2d.cell_node_values().copy()
    import os
    assert os.path.exists(tmp_path/'a'/'Elevation2d'/'Elevation2d_4.vtu') and os.path.exists(tmp_path/'a'/'Velocity2d'/'Velocity2d.pvd')
    assert os.path.exists(tmp_path/'a'/'hdf5'/'Elevation2d_00002.npz')
    s2, _ = make(tmp_path/'b')
    s2.load_state(2, outputdir=str(tmp_path/'a'))
    assert s2.iteration == 100 and math.isclose(s2.simulation_time, 200.0) and s2.i_export == 2
    s2.iterate()
    assert s2.iteration == 200 and s2.i_export == 4
    assert np.array_equal(s2.fields.elev_2d.cell_node_values(), e_full)          # deterministic kernel: bitwise restart


def test_atmospheric_pressure_reference_test(hip_lib):
    """test/swe2d/test_atmospheric_pressure.py::test_pressure_forcing[SSPRK33-dg-dg] through FlowSolver2d on the device."""
    from test_oracle_known_answers import _pressure_forcing_errors, check_pressure_forcing_orders

    def run(mesh2d, dt, t_end, patm_fn, eta_fn):
        P1 = get_functionspace(mesh2d, 'DG', 1)
        bathymetry = Function(P1, name='bathymetry').interpolate(Constant(5.0))
        atmospheric_pressure = Function(P1, name='atmospheric_pressure').interpolate(patm_fn)
        solverObj = solver2d.FlowSolver2d(mesh2d, bathymetry)
        o = solverObj.options
        o.polynomial_degree = 1
        o.swe_timestepper_type = 'SSPRK33'
        o.swe_timestepper_options.use_automatic_timestep = False
        o.element_family = 'dg-dg'
        o.check_volume_conservation_2d = False
        o.timestep = dt
        o.simulation_export_time = 3600.0
        o.simulation_end_time = t_end
        o.no_exports = True
        o.manning_drag_coefficient = Constant(1.0)
        o.atmospheric_pressure = atmospheric_pressure
        solverObj.assign_initial_conditions(uv=Constant((1e-7, 0.)))
        solverObj.iterate()
        eta = solverObj.fields.elev_2d.cell_node_values()
        eta_ana = Function(solverObj.function_spaces.H_2d).project(eta_fn).cell_node_values()
        d = eta - eta_ana
        area = mesh2d.cell_areas()
        return math.sqrt(float(np.sum(area/12.0*(d.sum(axis=1)**2 + (d**2).sum(axis=1)))))
    check_pressure_forcing_orders(_pressure_forcing_errors(run))


def test_function_valued_tidal_boundary(hip_lib):
    """A tidal elevation FIELD on an open boundary, re-assigned by update_forcings every stage (the set-up of
    demos/demo_2d_north_sea: bnd_functions['shallow_water'] = {marker: {'elev': tidal_elev}}), against the numpy oracle."""
    from helpers import make_oracle
    lx, ly = 13800.0, 7200.0
    mesh2d = RectangleMesh(12, 6, lx, ly)
    P1_2d = get_functionspace(mesh2d, 'CG', 1)
    bath = Function(P1_2d).interpolate(lambda x, y: 5.0 + x/2760.0)
    s = solver2d.FlowSolver2d(mesh2d, bath)
    o = s.options
    o.swe_timestepper_type = 'SSPRK33'
    o.swe_timestepper_options.use_automatic_timestep = False
    o.timestep = 5.0
    o.simulation_end_time = 50.0
    o.simulation_export_time = 25.0
    tidal_elev = Function(P1_2d)
    s.bnd_functions['shallow_water'] = {2: {'elev': tidal_elev}}
    shape = lambda x, y: 1.0 + 0.3*np.sin(2*np.pi*y/ly)
    tide = lambda t: -0.5*math.sin(2*math.pi*t/1000.0)
    s.assign_initial_conditions(elev=Constant(0.0))
    s.iterate(update_forcings=lambda t: tidal_elev.interpolate(lambda x, y: tide(t)*shape(x, y)))
    eta = s.fields.elev_2d.cell_node_values()
    uv = s.fields.uv_2d.cell_node_values()
    cxy = mesh2d.cell_xy()
    val = {'t': 0.0}
    orc = make_oracle(mesh2d, bath.dat.data_ro,
                      bnd_conditions={2: {'elev': lambda t: tide(val['t'])*shape(cxy[:, :, 0], cxy[:, :, 1])}})
    u_o, e_o = np.zeros_like(uv), np.zeros_like(eta)
    for kstep in range(10):
        u_o, e_o = orc.ssprk33_step(u_o, e_o, 5.0, t=5.0*kstep, update_forcings=lambda t: val.__setitem__('t', t))
    assert rel_linf(eta, e_o) < 1e-10 and rel_linf(uv, u_o) < 1e-10


@pytest.mark.parametrize('name', ['setup7', 'setup8', 'setup9'])
def test_steady_state_basin_mms_convergence(hip_lib, name):
    """test/swe2d/test_steady_state_basin_mms.py::test_steady_state_basin_convergence[dg-dg] with SSPRK33 on the device:
    refinements [1, 2, 4, 6], second order for elevation and velocity within the reference's 20 % slope tolerance."""
    import mms_basin
    refs = [1, 2, 4, 6]
    errs = [mms_basin.run_device(name, r) for r in refs]
    slope_e, slope_u = mms_basin.convergence_rates(errs, refs)
    assert abs(slope_e - 2.0)/2.0 < 0.2, (errs, slope_e)
    assert abs(slope_u - 2.0)/2.0 < 0.2, (errs, slope_u)
    # and the device result is the oracle's (same scenario, refinement 1)
    eo, uo = mms_basin.run_oracle(name, 1)
    assert abs(errs[0][0] - eo) < 1e-3*eo and abs(errs[0][1] - uo) < 1e-3*uo


@pytest.mark.parametrize('dt,t_end', [(2.0, 500.0), (3.0, 470.0)])
def test_iterate_batches_steps_between_exports_with_identical_results(hip_lib, capsys, dt, t_end):
    """iterate() issues all steps up to the next export in one library call (no forcings, no per-step callbacks): same
    step count, times, export instants, printed state lines and bitwise the same state as the step-by-step generator."""
    a, _, _, elev_init = _channel2d_solver(dt=dt, t_end=t_end)
    a.options.no_exports = True
    a.assign_initial_conditions(elev=elev_init)
    for _ in a.create_iterator():
        pass
    out_a = capsys.readouterr().out
    b, _, _, elev_init = _channel2d_solver(dt=dt, t_end=t_end)
    b.options.no_exports = True
    b.assign_initial_conditions(elev=elev_init)
    b.iterate()
    out_b = capsys.readouterr().out
    assert (a.iteration, a.i_export, a.simulation_time) == (b.iteration, b.i_export, b.simulation_time)
    assert np.array_equal(a.fields.uv_2d.dat.data_ro, b.fields.uv_2d.dat.data_ro)
    assert np.array_equal(a.fields.elev_2d.dat.data_ro, b.fields.elev_2d.dat.data_ro)
    strip = lambda s: [' '.join(l.split()[:-1]) for l in s.splitlines() if l.strip() and l.split()[0].isdigit()]   # drop Tcpu
    assert strip(out_a) == strip(out_b) and len(strip(out_a)) >= 5


def test_steady_state_channel_reference_scenario(hip_lib):
    """test/swe2d/test_steady_state_channel.py through FlowSolver2d with SSPRK33 marching (the reference uses an implicit
    solve): Function-valued 'un' inflow and 'elev' outflow, linear drag; L2 error of eta against 1 - x/lx below 1e-2."""
    from mms_basin import l2_error
    lx, ly = 5e3, 1e3
    mesh2d = RectangleMesh(10, 1, lx, ly)
    p1_2d = get_functionspace(mesh2d, 'CG', 1)
    bathymetry_2d = Function(p1_2d, name='bathymetry').assign(100.0)
    g = 9.81
    so = solver2d.FlowSolver2d(mesh2d, bathymetry_2d)
    o = so.options
    o.use_nonlinear_equations = False
    o.no_exports = True
    o.swe_timestepper_type = 'SSPRK33'
    o.swe_timestepper_options.use_automatic_timestep = False
    o.timestep = 2.0
    o.simulation_export_time = 1000.0
    o.simulation_end_time = 6000.0
    o.linear_drag_coefficient = Constant(g/lx)
    inflow_func = Function(p1_2d).assign(-1.0)                   # NOTE negative into domain
    outflow_func = Function(p1_2d).assign(0.0)
    so.bnd_functions['shallow_water'] = {1: {'un': inflow_func}, 2: {'elev': outflow_func}}
    so.create_equations()
    so.assign_initial_conditions(uv=Constant((1.0, 0.0)))
    so.iterate()
    eta = so.fields.elev_2d.cell_node_values()
    assert l2_error(mesh2d, eta, lambda x, y: 1.0 - x/lx) < 1e-2


def test_steady_state_channel_mms_reference_scenario(hip_lib):
    """test/swe2d/test_steady_state_channel_mms.py[dg-dg]: nonlinear equations, quadratic drag C_D = 0.0025, manufactured
    eta = cos(kx), u = Q/H with the momentum source of :42, Function-valued 'un' inflow / 'elev' outflow, meshes
    48*2^i x 1.  The reference finds the steady state with a Newton solve; here SSPRK33 marches to it (1e5 s, up to 1e6
    steps of the 768-cell mesh, ~1 min of GPU time) from the reference's initial guess u = (1, 0).  Criteria :113-127:
    every refinement divides the eta and u errors by more than 4*0.75, the total by more than 4^3*0.75."""
    from mms_basin import l2_error
    lx, ly = 5e3, 1e3
    g, H0, Q, eta0, C_D = 9.81, 10.0, 10.0, 1.0, 0.0025
    k = 4.0*math.pi/lx
    eta_f = lambda x, y: eta0*np.cos(k*x)
    H_f = lambda x: H0 + eta0*np.cos(k*x)
    u_f = lambda x, y: Q/H_f(x)
    src_f = lambda x, y: (k*eta0*(Q**2/H_f(x)**3 - g)*np.sin(k*x) + C_D*np.abs(u_f(x, y))*u_f(x, y)/H_f(x), 0.0*x)
    eta_errs, u_errs = [], []
    for i in range(4):
        n = 48*2**i
        mesh2d = RectangleMesh(n, 1, lx, ly)
        p1_2d = get_functionspace(mesh2d, 'CG', 1)
        bathymetry_2d = Function(p1_2d, name='bathymetry').assign(H0)
        so = solver2d.FlowSolver2d(mesh2d, bathymetry_2d)
        o = so.options
        o.element_family = 'dg-dg'
        o.polynomial_degree = 1
        o.use_nonlinear_equations = True
        o.quadratic_drag_coefficient = Constant(C_D)
        o.no_exports = True
        o.swe_timestepper_type = 'SSPRK33'
        o.swe_timestepper_options.use_automatic_timestep = False
        o.timestep = 0.08*(lx/n)/math.sqrt(g*(H0 + eta0))
        o.simulation_end_time = 1.0e5
        o.simulation_export_time = 1.0e5
        so.create_function_spaces()
        o.momentum_source_2d = Function(so.function_spaces.U_2d, name='Source').project(src_f)
        inflow_func = Function(p1_2d).interpolate(lambda x, y: -u_f(x, y))
        outflow_func = Function(p1_2d).assign(eta0)
        so.bnd_functions['shallow_water'] = {1: {'un': inflow_func}, 2: {'elev': outflow_func}}
        so.create_equations()
        so.assign_initial_conditions(uv=Constant((1.0, 0.0)))
        so.iterate()
        eta_errs.append(l2_error(mesh2d, so.fields.elev_2d.cell_node_values(), eta_f))
        u_errs.append(l2_error(mesh2d, so.fields.uv_2d.cell_node_values()[:, :, 0], u_f))
    for errs in (np.array(eta_errs), np.array(u_errs)):
        assert all(errs[:-1]/errs[1:] > 2.0**2*0.75), errs
        assert errs[0]/errs[-1] > (2.0**2)**3*0.75, errs


def test_spatially_varying_drag_coefficients_through_solver(hip_lib):
    """options.manning_drag_coefficient / linear_drag_coefficient given as Functions (roughness maps): nodal fields on the
    device; the same run with the oracle."""
    from helpers import make_oracle
    solver_obj, mesh, bath, elev_init = _channel2d_solver(dt=2.0, t_end=20.0)
    P1_2d = get_functionspace(mesh, 'CG', 1)
    mann = Function(P1_2d).interpolate(lambda x, y: 0.02*(1.0 + x/100e3))
    lin = Function(P1_2d).interpolate(lambda x, y: 1e-4*(1.0 + y/3750.0))
    o = solver_obj.options
    o.manning_drag_coefficient = mann
    o.linear_drag_coefficient = lin
    o.no_exports = True
    o.simulation_export_time = 20.0
    solver_obj.assign_initial_conditions(elev=elev_init)
    uv0 = solver_obj.fields.uv_2d.cell_node_values().copy()
    e0 = solver_obj.fields.elev_2d.cell_node_values().copy()
    solver_obj.iterate()
    orc = make_oracle(mesh, bath.dat.data_ro, manning_drag_coefficient=mann.dat.data_ro, linear_drag_coefficient=lin.dat.data_ro)
    u, e = uv0, e0
    for _ in range(solver_obj.iteration):
        u, e = orc.ssprk33_step(u, e, 2.0)
    assert solver_obj.iteration == 10
    assert rel_linf(solver_obj.fields.uv_2d.cell_node_values(), u) < 1e-11
    assert rel_linf(solver_obj.fields.elev_2d.cell_node_values(), e) < 1e-11


def test_time_dependent_forcing_fields_follow_update_forcings(hip_lib):
    """A wind-stress Function (and an atmospheric pressure Function) rewritten by ``update_forcings`` at every stage time
    t + c_i dt (rungekutta.py:933-934): the device copies are refreshed when the Functions change."""
    from helpers import make_oracle
    solver_obj, mesh, bath, elev_init = _channel2d_solver(dt=2.0, t_end=12.0)
    o = solver_obj.options
    o.no_exports = True
    o.simulation_export_time = 12.0
    solver_obj.create_function_spaces()
    wind = Function(solver_obj.function_spaces.U_2d, name='wind stress')
    patm = Function(solver_obj.function_spaces.H_2d, name='atmospheric pressure')
    o.wind_stress = wind
    o.atmospheric_pressure = patm
    xy = mesh.cell_xy()

    def tau(t):
        return np.stack([0.3*np.sin(0.4*t)*(1.0 + xy[:, :, 0]/100e3), 0.1*np.cos(0.3*t) + 0.0*xy[:, :, 0]], axis=2)

    def pa(t):
        return 1.0e5 + 500.0*np.sin(0.2*t)*np.cos(xy[:, :, 0]/2e4)
    calls = []

    def update_forcings(t):
        calls.append(t)
        wind.dat.data[...] = tau(t).reshape(wind.dat.data_ro.shape)
        patm.dat.data[...] = pa(t).reshape(patm.dat.data_ro.shape)
    update_forcings(0.0)
    solver_obj.assign_initial_conditions(elev=elev_init)
    u, e = solver_obj.fields.uv_2d.cell_node_values().copy(), solver_obj.fields.elev_2d.cell_node_values().copy()
    solver_obj.iterate(update_forcings=update_forcings)
    orc = make_oracle(mesh, bath.dat.data_ro, wind_stress=tau(0.0), atmospheric_pressure=pa(0.0))

    def orc_forcings(t):
        orc.wind, orc.patm = tau(t), pa(t)
    for k in range(solver_obj.iteration):
        u, e = orc.ssprk33_step(u, e, 2.0, t=2.0*k, update_forcings=orc_forcings)
    assert solver_obj.iteration == 6 and calls[1:4] == [0.0, 2.0, 1.0]          # c = (0, 1, 1/2)
    assert rel_linf(solver_obj.fields.uv_2d.cell_node_values(), u) < 1e-11
    assert rel_linf(solver_obj.fields.elev_2d.cell_node_values(), e) < 1e-11


def test_geostrophic_gyre_example_stays_in_balance(hip_lib):
    """examples/geostrophicGyre/geoGyre2d.py (f-plane, linear equations: a Gaussian elevation bell with the velocity of
    geostrophic balance is a steady state) with the explicit stepper instead of CrankNicolson.  The discrete state must
    stay at the analytical one up to the spatial error, and volume is conserved."""
    lx, nx, depth, elev_amp = 1.0e6, 20, 1000.0, 3.0
    f0, sigma, g = 1.0e-4, 160.0e3, 9.81
    x0 = y0 = lx/2

    def elev_expr(x, y):
        return elev_amp*np.exp(-((x - x0)**2 + (y - y0)**2)/sigma**2)

    def run(n):
        mesh2d = RectangleMesh(n, n, lx, lx)
        P1_2d = get_functionspace(mesh2d, 'CG', 1)
        bathymetry_2d = Function(P1_2d, name='Bathymetry').assign(depth)
        solver_obj = solver2d.FlowSolver2d(mesh2d, bathymetry_2d)
        options = solver_obj.options
        options.use_nonlinear_equations = False
        options.coriolis_frequency = Function(P1_2d).assign(f0)
        options.simulation_export_time = 3600.0
        options.simulation_end_time = 4*3600.0
        options.swe_timestepper_type = 'SSPRK33'
        options.swe_timestepper_options.use_automatic_timestep = False
        dt = 0.05*(lx/n)/math.sqrt(2.0)/math.sqrt(g*depth)           # the reference's explicit rule, solver2d.py:213-214
        options.timestep = 3600.0/math.ceil(3600.0/dt)
        options.check_volume_conservation_2d = True
        options.no_exports = True
        solver_obj.create_equations()
        elev_init = Function(solver_obj.function_spaces.H_2d).project(elev_expr)
        uv_init = Function(solver_obj.function_spaces.U_2d).project(
            lambda x, y: (g/f0*2*(y - y0)/sigma**2*elev_expr(x, y), -g/f0*2*(x - x0)/sigma**2*elev_expr(x, y)))
        solver_obj.assign_initial_conditions(elev=elev_init, uv=uv_init)
        e0 = solver_obj.fields.elev_2d.cell_node_values().copy()
        u0 = solver_obj.fields.uv_2d.cell_node_values().copy()
        solver_obj.iterate()
        e1 = solver_obj.fields.elev_2d.cell_node_values()
        u1 = solver_obj.fields.uv_2d.cell_node_values()
        vol = solver_obj.callbacks['export']['volume2d']
        assert abs(vol.rel_diff if hasattr(vol, 'rel_diff') else 0.0) < 1e-10
        return np.abs(e1 - e0).max()/elev_amp, np.abs(u1 - u0).max()/np.abs(u0).max()

    de20, du20 = run(20)
    de40, du40 = run(40)
    # 4 h = 2.3 inertial periods: the drift away from the balanced state is the spatial truncation error
    assert de20 < 0.05 and du20 < 0.15, (de20, du20)
    assert de40 < 0.5*de20 and du40 < 0.6*du20, (de20, de40, du20, du40)     # and it converges


def test_stommel_gyre_example_spins_up(hip_lib):
    """examples/stommel2d/stommel2d.py (beta-plane Coriolis, wind stress, linear drag, linear equations) for its
    regression length (10 h) with the explicit stepper: the wind does work at the analytical rate at early times -
    d/dt int u dx = int tau_x/(rho0 H) dx - and the basin keeps its volume."""
    lx, nx, depth = 1.0e6, 20, 1000.0
    mesh2d = RectangleMesh(nx, nx, lx, lx)
    P1_2d = get_functionspace(mesh2d, 'CG', 1)
    P1v_2d = get_functionspace(mesh2d, 'CG', 1, vector=True)
    bathymetry_2d = Function(P1_2d, name='Bathymetry').assign(depth)
    f0, beta, tau_max = 1.0e-4, 2.0e-11, 0.1
    coriolis_2d = Function(P1_2d).interpolate(lambda x, y: f0 + beta*y)
    wind_stress_2d = Function(P1v_2d, name='wind stress').interpolate(
        lambda x, y: (tau_max*np.sin(np.pi*(y/lx - 0.5)), 0.0*x))
    solver_obj = solver2d.FlowSolver2d(mesh2d, bathymetry_2d)
    options = solver_obj.options
    options.use_nonlinear_equations = False
    options.coriolis_frequency = coriolis_2d
    options.wind_stress = wind_stress_2d
    options.linear_drag_coefficient = Constant(1e-6)
    options.simulation_export_time = 3600.0*2
    options.simulation_end_time = 5*3600.0*2
    options.swe_timestepper_type = 'SSPRK33'
    options.swe_timestepper_options.use_automatic_timestep = False
    options.timestep = 15.0
    options.horizontal_velocity_scale = Constant(0.01)
    options.check_volume_conservation_2d = True
    options.no_exports = True
    solver_obj.assign_initial_conditions()
    solver_obj.iterate()
    uv = solver_obj.fields.uv_2d.cell_node_values()
    eta = solver_obj.fields.elev_2d.cell_node_values()
    assert np.isfinite(uv).all() and np.isfinite(eta).all()
    # the wind blows westward in the south, eastward in the north: after 10 h (2 inertial periods) the basin-mean zonal
    # velocity of the southern / northern half has the sign of the stress, its size is bounded by tau t/(rho0 H)
    xy = mesh2d.cell_xy()
    south = xy[:, :, 1].mean(axis=1) < 0.5*lx
    u_s, u_n = uv[south, :, 0].mean(), uv[~south, :, 0].mean()
    bound = tau_max*36000.0/(1000.0*depth)
    assert u_s < 0 < u_n and abs(u_s) < bound and abs(u_n) < bound, (u_s, u_n, bound)
    assert abs(u_s + u_n) < 0.05*(abs(u_s) + abs(u_n))                   # antisymmetric forcing
    assert 1e-4 < np.abs(eta).max() < 1.0


def test_demo_2d_channel_bnd_time_dependent_flux(hip_lib):
    """demos/demo_2d_channel_bnd.py: 'elev' + 'flux' on the right boundary, a tidal 'flux' Constant on the left updated by
    ``update_forcings``; SSPRK33 in place of CrankNicolson, the first 20 minutes, against the numpy oracle step by step."""
    lx, ly, nx, ny, depth = 40e3, 2e3, 25, 2, 20.0
    mesh2d = RectangleMesh(nx, ny, lx, ly)
    P1_2d = get_functionspace(mesh2d, 'CG', 1)
    bathymetry_2d = Function(P1_2d, name='Bathymetry').assign(depth)
    solver_obj = solver2d.FlowSolver2d(mesh2d, bathymetry_2d)
    options = solver_obj.options
    dt, t_end = 4.0, 1200.0
    options.simulation_export_time = 300.0
    options.simulation_end_time = t_end
    options.swe_timestepper_type = 'SSPRK33'
    options.swe_timestepper_options.use_automatic_timestep = False
    options.timestep = dt
    options.no_exports = True
    left_bnd_id, right_bnd_id, in_flux = 1, 2, 1e3

    def timedep_flux(simulation_time):
        return -2e3*math.sin(2*math.pi*simulation_time/(12*3600.0)) + in_flux
    tide_flux_const = Constant(timedep_flux(0))
    solver_obj.bnd_functions['shallow_water'] = {right_bnd_id: {'elev': Constant(0.0), 'flux': Constant(-in_flux)},
                                                 left_bnd_id: {'flux': tide_flux_const}}

    def update_forcings(t_new):
        tide_flux_const.assign(timedep_flux(t_new))
    solver_obj.assign_initial_conditions()
    solver_obj.iterate(update_forcings=update_forcings)
    uv, eta = solver_obj.fields.solution_2d.subfunctions
    e_d = eta.dat.data_ro.reshape(-1, 3)
    u_d = uv.dat.data_ro.reshape(-1, 3, 2)

    from helpers import make_oracle
    val = {'f': timedep_flux(0)}
    orc = make_oracle(mesh2d, bathymetry_2d.dat.data_ro,
                      bnd_conditions={right_bnd_id: {'elev': 0.0, 'flux': -in_flux}, left_bnd_id: {'flux': lambda t: val['f']}})

    def uf(t):
        val['f'] = timedep_flux(t)
    u_o = np.zeros((mesh2d.num_cells, 3, 2))
    e_o = np.zeros((mesh2d.num_cells, 3))
    for k in range(int(round(t_end/dt))):
        u_o, e_o = orc.ssprk33_step(u_o, e_o, dt, t=dt*k, update_forcings=uf)
    assert rel_linf(e_d, e_o) < 1e-10 and rel_linf(u_d, u_o) < 1e-10
    assert np.abs(e_d).max() < 0.5 and np.abs(u_d).max() < 0.2


def test_wave_eq_2d_example_returns_to_initial_state(hip_lib):
    """examples/waveEq2d/channel2d_waveEq.py with its explicit-scheme time step (dt/40): linear standing wave whose
    initial condition repeats every 20 exports; one full cycle with SSPRK33."""
    lx, ly, nx, ny, depth, elev_amp = 44294.46, 3000.0, 25, 2, 50.0, 1.0
    mesh2d = RectangleMesh(nx, ny, lx, ly)
    P1_2d = get_functionspace(mesh2d, 'CG', 1)
    bathymetry_2d = Function(P1_2d, name='Bathymetry').assign(depth)
    c_wave = math.sqrt(9.81*depth)
    t_cycle = lx/c_wave
    dt = round(t_cycle/20)
    solver_obj = solver2d.FlowSolver2d(mesh2d, bathymetry_2d)
    options = solver_obj.options
    options.use_nonlinear_equations = False
    options.simulation_export_time = dt
    options.simulation_end_time = 20*dt
    options.horizontal_velocity_scale = Constant(0.5)
    options.check_volume_conservation_2d = True
    options.swe_timestepper_type = 'SSPRK33'
    options.swe_timestepper_options.use_automatic_timestep = False
    options.timestep = dt/40.0
    options.no_exports = True
    solver_obj.create_equations()
    elev_init = Function(solver_obj.function_spaces.H_2d).interpolate(lambda x, y: -elev_amp*np.cos(2*np.pi*x/lx))
    solver_obj.assign_initial_conditions(elev=elev_init)
    e0 = solver_obj.fields.elev_2d.cell_node_values().copy()
    solver_obj.iterate()
    e1 = solver_obj.fields.elev_2d.cell_node_values()
    u1 = solver_obj.fields.uv_2d.cell_node_values()
    # 20 exports of round(T/20) s are 0.2 % longer than the true period; the rest is the (small) dispersion error
    err = np.sqrt(((e1 - e0)**2).mean())/np.sqrt((e0**2).mean())
    assert err < 0.03, err
    # velocity amplitude of the wave is a c/H = 0.44 m/s at mid-cycle; at a full cycle it is back at (nearly) rest
    assert np.abs(u1[:, :, 0]).max() < 0.02 and np.abs(u1[:, :, 1]).max() < 0.01
    # mid-cycle the wave is inverted: run half a cycle from the same start
    solver2 = solver2d.FlowSolver2d(mesh2d, bathymetry_2d)
    o2 = solver2.options
    o2.use_nonlinear_equations = False
    o2.simulation_export_time = dt
    o2.simulation_end_time = 10*dt
    o2.swe_timestepper_type = 'SSPRK33'
    o2.swe_timestepper_options.use_automatic_timestep = False
    o2.timestep = dt/40.0
    o2.no_exports = True
    solver2.create_equations()
    solver2.assign_initial_conditions(elev=elev_init)
    solver2.iterate()
    eh = solver2.fields.elev_2d.cell_node_values()
    assert np.sqrt(((eh + e0)**2).mean())/np.sqrt((e0**2).mean()) < 0.03
