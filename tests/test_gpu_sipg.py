"""GPU: the SIPG pass kernels (swe2d_sipg.h) against the oracle - HorizontalViscosityTerm inside the shallow-water
stage, tracer HorizontalDiffusionTerm inside the tracer stage - through the C ABI."""
import numpy as np
import pytest

from helpers import channel_case, delaunay_case, make_oracle, rel_linf

pytestmark = pytest.mark.gpu
TOL = 1e-12


def _dev(mesh, bath, dt, **kw):
    from thetis_amd.device import Swe2dDevice
    return Swe2dDevice(mesh, bath, dt, **kw)


@pytest.fixture(params=['fused', 'separate_pass'], autouse=True)
def viscosity_path(request, monkeypatch):
    """Every test of this file runs on both implementations of the triangle viscosity: fused into the stage kernel
    (default) and as the separate SIPG pass (THETIS_AMD_NO_VISC_FUSION, read by swe2d_create)."""
    if request.param == 'separate_pass':
        monkeypatch.setenv('THETIS_AMD_NO_VISC_FUSION', '1')
    else:
        monkeypatch.delenv('THETIS_AMD_NO_VISC_FUSION', raising=False)
    return request.param


VISC_CASES = {
    'const': dict(nu='const'),
    'vertex_field': dict(nu='field', sipg_factor=2.5),
    'grad_div': dict(nu='field', use_grad_div_viscosity_term=True),
    'no_grad_depth': dict(nu='const', use_grad_depth_viscosity_term=False),
    'grad_div_linear': dict(nu='field', use_grad_div_viscosity_term=True, use_nonlinear_equations=False),
}


@pytest.mark.parametrize('mesh_kind', ['channel', 'delaunay'])
@pytest.mark.parametrize('case', sorted(VISC_CASES))
def test_viscosity_tendency_and_step_match_oracle(hip_lib, case, mesh_kind):
    cfg = dict(VISC_CASES[case])
    mesh, bath, uv, eta = channel_case(seed=21) if mesh_kind == 'channel' else delaunay_case(n_points=300, seed=5)
    rng = np.random.default_rng(7)
    nu = 40.0 if cfg.pop('nu') == 'const' else 20.0 + 30.0*rng.uniform(size=mesh.num_vertices)
    nonlin = cfg.pop('use_nonlinear_equations', True)
    dt = 2.0 if mesh_kind == 'channel' else 0.2
    orc = make_oracle(mesh, bath, horizontal_viscosity=nu, use_nonlinear_equations=nonlin, **cfg)
    dev = _dev(mesh, bath, dt, use_nonlinear_equations=nonlin)
    dev.set_viscosity(nu, sipg_factor=cfg.get('sipg_factor', 1.0),
                      use_grad_div_viscosity_term=cfg.get('use_grad_div_viscosity_term', False),
                      use_grad_depth_viscosity_term=cfg.get('use_grad_depth_viscosity_term', True))
    dev.set_state(uv, eta)
    ku, ke = dev.tendency()
    ku_o, ke_o = orc.tendency(uv, eta, dt)
    # the viscous part must be visible in the comparison
    orc0 = make_oracle(mesh, bath, use_nonlinear_equations=nonlin)
    assert rel_linf(orc0.tendency(uv, eta, dt)[0], ku_o) > 1e-4
    assert rel_linf(ku, ku_o) < TOL and rel_linf(ke, ke_o) < TOL
    dev.advance(2)
    u1, e1 = dev.get_state()
    uo, eo = uv, eta
    for _ in range(2):
        uo, eo = orc.ssprk33_step(uo, eo, dt)
    assert rel_linf(u1, uo) < TOL and rel_linf(e1, eo) < TOL
    # switching the term off restores the inviscid result
    dev.set_viscosity(None)
    dev.set_state(uv, eta)
    assert rel_linf(dev.tendency()[0], orc0.tendency(uv, eta, dt)[0]) < TOL
    dev.close()


@pytest.mark.parametrize('bcs', [
    {1: {'un': 0.3}, 2: {'elev': 0.2}},
    {1: {'uv': (0.4, -0.1)}, 2: {'elev': 0.1, 'un': -0.2}},
    {1: {'flux': 2.0e4}, 2: {'elev': 0.3, 'flux': -1.5e4}, 3: {'elev': 0.1, 'uv': (0.1, 0.2)}},
], ids=['un', 'uv', 'flux'])
@pytest.mark.parametrize('grad_div', [False, True])
def test_viscosity_dirichlet_boundary_terms_match_oracle(hip_lib, bcs, grad_div):
    mesh, bath, uv, eta = channel_case(seed=23)
    dt = 2.0
    orc = make_oracle(mesh, bath, bnd_conditions=bcs, horizontal_viscosity=60.0, use_grad_div_viscosity_term=grad_div)
    dev = _dev(mesh, bath, dt)
    for marker, funcs in bcs.items():
        dev.set_bc(marker, funcs)
    dev.set_viscosity(60.0, use_grad_div_viscosity_term=grad_div)
    dev.set_state(uv, eta)
    ku, ke = dev.tendency()
    ku_o, ke_o = orc.tendency(uv, eta, dt)
    assert rel_linf(ku, ku_o) < TOL and rel_linf(ke, ke_o) < TOL
    dev.close()


def test_viscosity_with_function_valued_boundary_velocity(hip_lib):
    mesh, bath, uv, eta = channel_case(seed=24)
    rng = np.random.default_rng(1)
    uvf = 0.3*rng.normal(size=(mesh.num_cells, 3, 2))
    unf = 0.2*rng.normal(size=(mesh.num_cells, 3))
    bcs = {1: {'uv': uvf}, 2: {'un': unf}}
    dt = 2.0
    orc = make_oracle(mesh, bath, bnd_conditions=bcs, horizontal_viscosity=35.0)
    dev = _dev(mesh, bath, dt)
    for marker, funcs in bcs.items():
        dev.set_bc(marker, funcs)
    dev.set_viscosity(35.0)
    dev.set_state(uv, eta)
    ku, ke = dev.tendency()
    ku_o, ke_o = orc.tendency(uv, eta, dt)
    assert rel_linf(ku, ku_o) < TOL and rel_linf(ke, ke_o) < TOL
    dev.close()


@pytest.mark.parametrize('case', ['const', 'vertex_field', 'value_bc', 'diff_flux_bc', 'reordered'])
def test_tracer_diffusion_matches_oracle(hip_lib, case):
    mesh, bath, uv, eta = delaunay_case(n_points=300, seed=9) if case == 'reordered' else channel_case(seed=25)
    rng = np.random.default_rng(11)
    T = rng.normal(size=(mesh.num_cells, 3))
    dt = 0.2 if case == 'reordered' else 2.0
    mu = 15.0 + 10.0*rng.uniform(size=mesh.num_vertices) if case in ('vertex_field', 'reordered') else 25.0
    sipg = 1.7 if case == 'vertex_field' else 1.0
    orc = make_oracle(mesh, bath)
    dev = _dev(mesh, bath, dt, reorder='hilbert' if case == 'reordered' else 'auto')
    tid = dev.add_tracer()
    kw = dict(diffusivity=mu, sipg_factor_tracer=sipg)
    dev.tracer_set_diffusivity(tid, mu, sipg)
    if case == 'value_bc':
        kw['bnd_conditions'] = {1: {'value': 2.0}, 3: {'value': -1.0}}
        for m, v in ((1, 2.0), (3, -1.0)):
            dev.tracer_set_bc(tid, m, v)
            dev.tracer_set_diffusion_bc(tid, m, 2)
    if case == 'diff_flux_bc':
        kw['bnd_conditions'] = {2: {'diff_flux': 0.05}, 4: {'diff_flux': -0.02}}
        dev.tracer_set_diffusion_bc(tid, 2, 1, 0.05)
        dev.tracer_set_diffusion_bc(tid, 4, 1, -0.02)
    dev.set_state(uv, eta)
    dev.tracer_set_state(tid, T)
    k_o = orc.tracer_tendency(T, uv, eta, dt, **kw)
    kw0 = {k: v for k, v in kw.items() if k not in ('diffusivity', 'sipg_factor_tracer')}
    assert rel_linf(orc.tracer_tendency(T, uv, eta, dt, **kw0), k_o) > 1e-4
    assert rel_linf(dev.tracer_tendency(tid), k_o) < TOL
    for s in range(3):
        dev.tracer_solve_stage(tid, s)
    assert rel_linf(dev.tracer_get_state(tid), orc.tracer_ssprk33_step(T, uv, eta, dt, **kw)) < TOL
    dev.tracer_set_diffusivity(tid, None)
    dev.tracer_set_state(tid, T)
    assert rel_linf(dev.tracer_tendency(tid), orc.tracer_tendency(T, uv, eta, dt, **kw0)) < TOL
    dev.close()


@pytest.mark.parametrize('mesh_kind', ['channel', 'quads'])
@pytest.mark.parametrize('grad_depth', [True, False])
def test_viscosity_with_wetting_and_drying_matches_oracle(hip_lib, mesh_kind, grad_depth):
    """HorizontalViscosityTerm together with the explicit wetting-drying formulation (shallowwater_eq.py:554-616 with
    total_h = the displaced depth): the depth of the grad-depth term and of 'flux' boundaries is D, and the dry-ground
    relaxation of a stage acts on the whole new velocity, viscous share included (the oracle's order of operations)."""
    from helpers import make_oracle_generic, quad_case
    if mesh_kind == 'quads':
        mesh, bath, uv, eta = quad_case(nx=9, ny=6, skew=0.2, seed=4)
        mk = make_oracle_generic
    else:
        mesh, bath, uv, eta = channel_case(nx=9, ny=6, seed=4)
        mk = make_oracle
    bath = bath - 12.0                                              # partly dry
    rng = np.random.default_rng(11)
    nu = 20.0 + 30.0*rng.uniform(size=mesh.num_vertices)
    alpha = 0.4 + 0.3*rng.uniform(size=mesh.num_vertices)
    dt = 0.5
    bcs = {1: {'flux': 2.0e3}, 2: {'elev': 0.3, 'un': 0.1}}
    orc = mk(mesh, bath, horizontal_viscosity=nu, use_grad_depth_viscosity_term=grad_depth, use_grad_div_viscosity_term=True,
             use_wetting_and_drying=True, wetting_and_drying_alpha=alpha, wd_mode='nodal', bnd_conditions=bcs)
    dev = _dev(mesh, bath, dt)
    dev.set_wetting_and_drying(alpha)
    dev.set_viscosity(nu, use_grad_div_viscosity_term=True, use_grad_depth_viscosity_term=grad_depth)
    for m, f in bcs.items():
        dev.set_bc(m, f)
    eta0 = orc.wd_clip_state(eta)
    dev.set_state(uv, eta)
    ku, ke = dev.tendency()
    ku_o, ke_o = orc.tendency(uv, eta0, dt)
    orc0 = mk(mesh, bath, use_wetting_and_drying=True, wetting_and_drying_alpha=alpha, wd_mode='nodal', bnd_conditions=bcs)
    assert rel_linf(orc0.tendency(uv, eta0, dt)[0], ku_o) > 1e-4   # the viscous part is visible
    assert rel_linf(ku, ku_o) < 1e-9 and rel_linf(ke, ke_o) < 1e-9
    dev.advance(3)
    u1, e1 = dev.get_state()
    uo, eo = uv, eta0
    for _ in range(3):
        uo, eo = orc.ssprk33_step(uo, eo, dt)
    assert np.isfinite(u1).all() and rel_linf(u1, uo) < 1e-9 and rel_linf(e1, eo) < 1e-9
    dev.close()


def _run_h_diffusion_solver(refinement):
    """test/tracerEq/test_h-diffusion_mes_2d.py:9-103 through FlowSolver2d (custom time loop on the tracer stepper)."""
    import math
    from scipy.special import erf
    from thetis_amd import Constant, Function, RectangleMesh, get_functionspace, solver2d
    lx, ly = 20.0e3, 5.0e3/refinement
    depth = 30.0
    horizontal_diffusivity = Constant(1.0e3)
    mesh2d = RectangleMesh(8*refinement + 1, 1, lx, ly)
    t_end, t_init = 3000.0, 1000.0
    p1_2d = get_functionspace(mesh2d, 'CG', 1)
    bathymetry_2d = Function(p1_2d, name='Bathymetry')
    bathymetry_2d.assign(depth)
    solverobj = solver2d.FlowSolver2d(mesh2d, bathymetry_2d)
    options = solverobj.options
    options.use_nonlinear_equations = False
    options.horizontal_velocity_scale = Constant(1.0)
    options.no_exports = True
    options.simulation_end_time = t_end
    options.simulation_export_time = (t_end - t_init)/8.0
    options.add_tracer_2d('tracer_2d', 'Depth averaged tracer', 'Tracer2d', diffusivity=horizontal_diffusivity)
    options.use_limiter_for_tracers = True
    options.horizontal_viscosity_scale = horizontal_diffusivity
    options.swe_timestepper_type = 'SSPRK33'
    options.tracer_timestepper_type = 'SSPRK33'
    solverobj.create_equations()
    mu = float(horizontal_diffusivity)
    x0 = lx/2.0

    def tracer_expr(t):
        return lambda x, y: -erf((x - x0)/np.sqrt(4*mu*t))
    elev_init = Function(solverobj.function_spaces.H_2d, name='elev init')
    solverobj.assign_initial_conditions(elev=elev_init, tracer=tracer_expr(t_init))
    ti = solverobj.timestepper.timesteppers.tracer_2d
    t = t_init
    while t < t_end - 1e-8:
        ti.advance(t)                      # the tracer stepper alone: no limiter in this loop (as in the reference)
        t += solverobj.dt
    T = solverobj.fields.tracer_2d.cell_node_values()
    # L2 error (degree-4 cell quadrature) / sqrt(area)
    from thetis_amd.function import triangle_quadrature
    xy = mesh2d.cell_xy()
    err2 = 0.0
    for bary, w in zip(*triangle_quadrature()):
        xq = xy[:, :, 0] @ bary
        err2 += np.sum(w*mesh2d.cell_areas()*((T @ bary) - tracer_expr(t)(xq, 0.0))**2)
    return math.sqrt(err2)/math.sqrt(lx*ly)


def test_horizontal_diffusion_convergence_through_solver(hip_lib):
    """The reference's test_horizontal_diffusion[SSPRK33-1-1.8] on the device path."""
    from scipy import stats
    refs = [1, 2, 3]
    errs = [_run_h_diffusion_solver(r) for r in refs]
    slope = stats.linregress(np.log10(np.array(refs, dtype=float)**-1), np.log10(errs)).slope
    assert slope > 1.8, (errs, slope)


def test_decaying_shear_flow_through_solver(hip_lib):
    """options.horizontal_viscosity through FlowSolver2d: u = sin(k y) e^{-nu k^2 t} (linear equations, periodic box);
    same scenario as tests/test_oracle_sipg.py::test_decaying_shear_flow_with_viscosity."""
    import math
    from thetis_amd import Constant, Function, PeriodicRectangleMesh, get_functionspace, solver2d
    errs = []
    for n in (8, 16):
        lx = ly = 1000.0
        nu = 20.0
        mesh2d = PeriodicRectangleMesh(n, n, lx, ly, direction='both')
        bath = Function(get_functionspace(mesh2d, 'CG', 1), name='Bathymetry').assign(10.0)
        so = solver2d.FlowSolver2d(mesh2d, bath)
        o = so.options
        o.use_nonlinear_equations = False
        o.horizontal_viscosity = Constant(nu)
        o.swe_timestepper_type = 'SSPRK33'
        o.swe_timestepper_options.use_automatic_timestep = False
        k = 2*math.pi/ly
        t_end = 0.1/(nu*k*k)
        dt = min(0.02*(lx/n)**2/nu, 0.05*(lx/n)/math.sqrt(9.81*10.0))
        nsteps = int(math.ceil(t_end/dt))
        o.timestep = t_end/nsteps
        o.simulation_end_time = t_end
        o.simulation_export_time = t_end
        o.no_exports = True
        so.assign_initial_conditions(uv=lambda x, y: (np.sin(k*y), 0.0*x))
        so.iterate()
        uv = so.fields.uv_2d.cell_node_values()
        xy = mesh2d.cell_xy()
        # the initial condition is L2-projected (not interpolated): compare with the projected exact solution's decay
        exact = np.sin(k*xy[:, :, 1])*math.exp(-nu*k*k*t_end)
        errs.append(np.sqrt(np.mean((uv[:, :, 0] - exact)**2)))
        assert abs(so.simulation_time - t_end) < 1e-9*t_end
    assert errs[0] < 0.05
    assert math.log2(errs[0]/errs[1]) > 1.7, errs


def _fourier_series_solution(xs, lx, diff_flux, nu, time):
    """test/tracerEq/test_bcs_2d.py:5-88 in closed form: c_t = nu c_xx, c_x(0) = D, c_x(l) = 0, c(x, 0) = 0, written as two
    homogeneous-Neumann problems (initial condition -I and source S = -nu D / l) minus I(x) = D (l - x)^2 / (2 l)."""
    ic = lambda x: diff_flux*0.5*(lx - x)*(lx - x)/lx
    src = -nu*diff_flux/lx
    xq = (np.arange(20000) + 0.5)*lx/20000                      # midpoint rule for the cosine coefficients
    coeff = lambda n: 2.0/lx*np.sum(ic(xq)*np.cos(n*np.pi*xq/lx))*(lx/20000)
    expr = 0.5*(2.0*src)*time + 0.5*coeff(0) + 0.0*xs
    for k in range(1, 100):
        expr = expr + coeff(k)*np.exp(-nu*(k*np.pi/lx)**2*time)*np.cos(k*np.pi*xs/lx)
    return -(expr - ic(xs))


@pytest.mark.parametrize('stepper', ['SSPRK33', 'ForwardEuler'])
def test_diffusive_flux_boundary_convergence(hip_lib, stepper):
    """test/tracerEq/test_bcs_2d.py::test_horizontal_advection[dg, SSPRK33 | ForwardEuler]: pure diffusion driven by a
    prescribed 'diff_flux' boundary, tracer_only, limiter on; the L2 error against the Fourier-series solution must more
    than halve per refinement (1, 2, 4)."""
    from thetis_amd import Constant, Function, RectangleMesh, get_functionspace, solver2d
    from mms_basin import l2_error
    errors = []
    for refinement in (1, 2, 4):
        lx, ly = 10.0, 1.0
        mesh2d = RectangleMesh(40*refinement, 4, lx, ly)
        dt = 0.1/refinement
        t_end = 1.0
        nu = Constant(0.1)
        diff_flux = 0.2
        bathy_2d = Function(get_functionspace(mesh2d, 'CG', 1), name='Bathymetry').assign(40.0)
        so = solver2d.FlowSolver2d(mesh2d, bathy_2d)
        o = so.options
        o.no_exports = True
        o.timestep = dt
        o.simulation_export_time = 0.1
        o.add_tracer_2d('tracer_2d', 'Depth averaged tracer', 'Tracer2d', diffusivity=nu)
        o.tracer_only = True
        o.horizontal_diffusivity_scale = nu
        o.horizontal_velocity_scale = Constant(0.0)
        o.tracer_timestepper_type = stepper
        o.tracer_element_family = 'dg'
        o.use_limiter_for_tracers = True
        o.simulation_end_time = t_end - 0.5*dt
        so.bnd_functions['tracer_2d'] = {1: {'diff_flux': diff_flux*float(nu)}}
        so.assign_initial_conditions()
        so.iterate()
        sol = so.fields.tracer_2d.cell_node_values()
        t_reached = so.simulation_time
        exact = lambda x, y: _fourier_series_solution(x, lx, diff_flux, float(nu), t_reached)
        errors.append(l2_error(mesh2d, sol, exact)*np.sqrt(lx*ly))
    assert errors[0]/errors[1] > 2 and errors[1]/errors[2] > 2, errors


QUAD_VISC_CASES = {
    'const': dict(nu='const'),
    'field_grad_div': dict(nu='field', sipg_factor=2.5, use_grad_div_viscosity_term=True),
    'no_grad_depth_linear': dict(nu='field', use_grad_depth_viscosity_term=False, use_nonlinear_equations=False),
}


@pytest.mark.parametrize('case', sorted(QUAD_VISC_CASES))
@pytest.mark.parametrize('geometry', ['parallelograms', 'general'])
def test_quad_viscosity_matches_oracle(hip_lib, case, geometry):
    """swe_sipg_kernel_quad<2> on skewed parallelograms, with Dirichlet terms of every velocity-type boundary kind; ``general``:
    warped convex cells (swe_sipg_kernel_quad<2, false>: gradients through the Jacobian at every quadrature point on both sides of
    a facet, true cell areas in the penalty, 4 x 4 mass solve)."""
    from helpers import make_oracle_generic, quad_case
    cfg = dict(QUAD_VISC_CASES[case])
    mesh, bath, uv, eta = quad_case(nx=8, ny=6, skew=0.3, seed=33, warp=(0.25 if geometry == 'general' else 0.0))
    assert mesh.affine == (geometry == 'parallelograms')
    rng = np.random.default_rng(17)
    nu = 40.0 if cfg.pop('nu') == 'const' else 20.0 + 30.0*rng.uniform(size=mesh.num_vertices)
    nonlin = cfg.pop('use_nonlinear_equations', True)
    n = mesh.num_cells
    bcs = {1: {'un': 0.3}, 2: {'elev': 0.2, 'flux': -1.5e4}, 3: {'uv': 0.3*rng.normal(size=(n, 4, 2))}, 4: {'elev': 0.1}}
    dt = 2.0
    orc = make_oracle_generic(mesh, bath, horizontal_viscosity=nu, use_nonlinear_equations=nonlin, bnd_conditions=bcs, **cfg)
    dev = _dev(mesh, bath, dt, use_nonlinear_equations=nonlin, boundary_len=mesh.boundary_len)
    for marker, funcs in bcs.items():
        dev.set_bc(marker, funcs)
    dev.set_viscosity(nu, sipg_factor=cfg.get('sipg_factor', 1.0),
                      use_grad_div_viscosity_term=cfg.get('use_grad_div_viscosity_term', False),
                      use_grad_depth_viscosity_term=cfg.get('use_grad_depth_viscosity_term', True))
    dev.set_state(uv, eta)
    ku, ke = dev.tendency()
    ku_o, ke_o = orc.tendency(uv, eta, dt)
    orc0 = make_oracle_generic(mesh, bath, use_nonlinear_equations=nonlin, bnd_conditions=bcs)
    assert rel_linf(orc0.tendency(uv, eta, dt)[0], ku_o) > 1e-4
    assert rel_linf(ku, ku_o) < TOL and rel_linf(ke, ke_o) < TOL
    dev.advance(2)
    u1, e1 = dev.get_state()
    uo, eo = uv, eta
    for _ in range(2):
        uo, eo = orc.ssprk33_step(uo, eo, dt)
    assert rel_linf(u1, uo) < TOL and rel_linf(e1, eo) < TOL
    dev.close()


@pytest.mark.parametrize('case', ['const', 'field_and_bcs'])
@pytest.mark.parametrize('geometry', ['parallelograms', 'general'])
def test_quad_tracer_diffusion_matches_oracle(hip_lib, case, geometry):
    from helpers import make_oracle_generic, quad_case
    mesh, bath, uv, eta = quad_case(nx=8, ny=6, skew=0.3, seed=35, warp=(0.25 if geometry == 'general' else 0.0))
    rng = np.random.default_rng(19)
    n = mesh.num_cells
    T = rng.normal(size=(n, 4))
    dt = 2.0
    orc = make_oracle_generic(mesh, bath)
    dev = _dev(mesh, bath, dt, boundary_len=mesh.boundary_len)
    tid = dev.add_tracer()
    if case == 'const':
        mu, sipg, bcs = 25.0, 1.0, {}
    else:
        mu, sipg = 15.0 + 10.0*rng.uniform(size=mesh.num_vertices), 1.7
        bcs = {1: {'value': rng.normal(size=(n, 4))}, 2: {'diff_flux': 0.05}, 3: {'value': 0.7, 'uv': np.array([0.3, -0.2])},
               4: {'elev': 0.1}}
        dev.tracer_set_bc(tid, 1, bcs[1]['value'])
        dev.tracer_set_bc(tid, 3, 0.7)
        dev.tracer_set_bc_velocity(tid, 3, uv=(0.3, -0.2))
        for m_, kind, fl in ((1, 4, 0.0), (2, 1, 0.05), (3, 2, 0.0), (4, 3, 0.0)):
            dev.tracer_set_diffusion_bc(tid, m_, kind, fl)
    kw = dict(diffusivity=mu, sipg_factor_tracer=sipg, bnd_conditions=bcs)
    dev.tracer_set_diffusivity(tid, mu, sipg)
    dev.set_state(uv, eta)
    dev.tracer_set_state(tid, T)
    k_o = orc.tracer_tendency(T, uv, eta, dt, **kw)
    assert rel_linf(orc.tracer_tendency(T, uv, eta, dt, bnd_conditions=bcs), k_o) > 1e-4
    assert rel_linf(dev.tracer_tendency(tid), k_o) < TOL
    for s in range(3):
        dev.tracer_solve_stage(tid, s)
    assert rel_linf(dev.tracer_get_state(tid), orc.tracer_ssprk33_step(T, uv, eta, dt, **kw)) < TOL
    dev.close()


def test_fused_and_separate_viscosity_agree_to_roundoff(hip_lib, monkeypatch, viscosity_path):
    """Same physics, different summation order: the two paths differ by round-off only (open and closed boundaries,
    sources, a vertex field, several steps)."""
    if viscosity_path != 'fused':
        pytest.skip('one comparison is enough')
    from thetis_amd import _lib
    mesh, bath, uv, eta = channel_case(nx=14, ny=6, seed=11)
    rng = np.random.default_rng(3)
    nu = 20.0 + 30.0*rng.uniform(size=mesh.num_vertices)
    out = {}
    for mode in ('fused', 'separate'):
        if mode == 'separate':
            monkeypatch.setenv('THETIS_AMD_NO_VISC_FUSION', '1')
        else:
            monkeypatch.delenv('THETIS_AMD_NO_VISC_FUSION', raising=False)
        dev = _dev(mesh, bath, 1.0)
        dev.set_viscosity(nu, sipg_factor=1.5, use_grad_div_viscosity_term=True)
        dev.set_scalar(_lib.SCALAR_MANNING_DRAG, 0.02)
        for marker in mesh.boundary_markers:
            dev.set_bc(marker, {'un': 0.05})
        dev.set_state(uv, eta)
        dev.advance(5)
        out[mode] = dev.get_state()
        dev.close()
    assert rel_linf(out['fused'][0], out['separate'][0]) < 1e-13 and rel_linf(out['fused'][1], out['separate'][1]) < 1e-13
    assert np.isfinite(out['fused'][0]).all()
