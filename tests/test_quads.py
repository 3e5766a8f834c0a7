"""DQ-1 on quadrilaterals (parallelograms: affine kernels; general convex cells: bilinear map, true mass matrix): oracle pins on
CPU, HIP parity on GPU (BASELINE cfg 1(ii), demo_2d_tracer mesh)."""
import math

import numpy as np
import pytest

from helpers import make_oracle_generic, make_ref, quad_case, rel_linf
from thetis_amd import Constant, Function, RectangleMesh, get_functionspace, solver2d

_BCS = {1: {'elev': 0.3}, 2: {'un': 0.2}, 3: {'flux': 1e4, 'elev': 0.1}, 4: {'uv': (0.1, -0.2)}}


def _cases(mesh, n):
    x, y = mesh.vertex_xy.T
    rng = np.random.default_rng(9)
    return {
        'default': {}, 'linear': dict(use_nonlinear_equations=False), 'no_lf': dict(use_lax_friedrichs_velocity=False),
        'sources': dict(coriolis=1e-4*(1 + y/30e3), linear_drag_coefficient=1e-3,
                        atmospheric_pressure=1e5 + 300*np.sin(x/2e4), momentum_source=1e-3*rng.normal(size=(n, 4, 2)),
                        volume_source=1e-3*rng.normal(size=(n, 4))),
        'manning': dict(manning_drag_coefficient=0.02),
        'quad_drag': dict(quadratic_drag_coefficient=0.0025, norm_smoother=0.1),
        'drag_fields': dict(manning_drag_coefficient=0.02*(1 + x/x.max()), linear_drag_coefficient=1e-3*(1 + y/y.max())),
        'nikuradse_field': dict(nikuradse_bed_roughness=0.05*(1 + x/x.max())),
        'bcs': dict(bnd_conditions=_BCS),
        'wind_bdrag': dict(wind_stress=0.1*rng.normal(size=(n, 4, 2)),
                           bnd_conditions={3: {'drag': 0.0025}, 1: {'drag': 0.01, 'elev': 0.1}}),
    }


def test_quad_mesh_connectivity():
    m = RectangleMesh(7, 4, 14.0, 4.0, quadrilateral=True)
    assert m.num_cells == 28 and m.nodes_per_cell == 4 and m.boundary_len == {1: 4.0, 2: 4.0, 3: 14.0, 4: 14.0}
    assert np.allclose(m.cell_areas(), 2.0)
    for c in range(m.num_cells):
        for f in range(4):
            nb = m.cell_nbr[c, f]
            if nb >= 0:
                f2 = m.cell_nbr_facet[c, f]
                assert m.cell_nbr[nb, f2] == c
                assert m.cells[c, f] == m.cells[nb, (f2 + 1) % 4] and m.cells[c, (f + 1) % 4] == m.cells[nb, f2]
    from thetis_amd.mesh import Mesh2d
    assert m.affine
    g = Mesh2d(np.array([[0, 0], [1, 0], [1.2, 1.1], [0, 1.0]]), np.array([[0, 1, 2, 3]]))      # not a parallelogram: general kernels
    assert not g.affine and math.isclose(g.cell_areas()[0], 0.5*(1.2*1.0 + 1.1*1.0 - 0.0))     # shoelace: 1.15
    with pytest.raises(ValueError):
        Mesh2d(np.array([[0, 0], [1, 0], [0.2, 0.2], [0, 1.0]]), np.array([[0, 1, 2, 3]]))      # not convex


def test_general_quadrilaterals_oracle_invariants():
    """The numpy oracle on warped (non-affine) cells - what the HIP kernels are compared with under -m gpu: lake at rest, closed
    domain conserves volume, 2 x 2 and 3 x 3 Gauss rules agree on the polynomial integrands (the mass matrix: det J is linear),
    a constant tracer stays constant, the mass-weighted cell mean is the P0 projection, the limiter conserves the tracer integral,
    and the closed form of the mass matrix the kernels use (swe_quad_mass) equals the quadrature."""
    mesh, bath, uv, eta = quad_case(skew=0.2, warp=0.3)
    assert not mesh.affine
    orc = make_oracle_generic(mesh, bath)
    assert not orc.affine and np.allclose(orc.mean_w.sum(axis=1), 1.0) and np.abs(orc.mean_w - 0.25).max() > 1e-3
    ru, re = orc.residual(np.zeros_like(uv), np.full_like(eta, 0.3))
    assert np.abs(ru).max() < 1e-8 and np.abs(re).max() == 0.0
    ru, re = orc.residual(uv, eta)
    assert abs(re.sum()) < 1e-12*np.abs(re).sum()
    from oracle.swe2d_oracle import SWEOracle
    o3 = SWEOracle(mesh.vertex_xy, mesh.cells, bath, quad_rule_points=3)
    assert rel_linf(o3.mass_matrix(), orc.mass_matrix()) < 1e-13
    ru3, re3 = o3.residual(uv, eta)
    # (on a warped cell adj(J) is linear in (xi, zeta): the integrands H u . adj(J)^T grad_ref(phi) have degree 4 per direction,
    #  beyond the 2-point rule - the two rules agree to discretisation accuracy only; on parallelograms they are equal)
    assert rel_linf(re3, re) < 5e-2
    T = np.full((mesh.num_cells, 4), 4.5)
    assert np.abs(orc.tracer_residual(T, uv, eta)).max() < 1e-9
    u1, e1 = orc.ssprk33_step(uv, eta, 2.0)
    assert abs(orc.volume(e1) - orc.volume(eta))/orc.volume(eta) < 1e-13
    # closed form of the mass matrix
    p = mesh.cell_xy()
    a, b, c = p[:, 1] - p[:, 0], p[:, 3] - p[:, 0], p[:, 0] - p[:, 1] + p[:, 2] - p[:, 3]
    cr = lambda u, v: u[:, 0]*v[:, 1] - u[:, 1]*v[:, 0]
    d0, d1, d2 = cr(a, b), cr(a, c), cr(c, b)
    m_, m1 = np.array([[1/3, 1/6], [1/6, 1/3]]), np.array([[1/12, 1/12], [1/12, 1/4]])
    ia, ib = [0, 1, 1, 0], [0, 0, 1, 1]
    M = np.zeros((mesh.num_cells, 4, 4))
    for i in range(4):
        for j in range(4):
            M[:, i, j] = (d0*m_[ia[i], ia[j]]*m_[ib[i], ib[j]] + d1*m1[ia[i], ia[j]]*m_[ib[i], ib[j]]
                          + d2*m_[ia[i], ia[j]]*m1[ib[i], ib[j]])
    assert rel_linf(M, orc.mass_matrix()) < 1e-14
    assert np.allclose(d0 + 0.5*(d1 + d2), mesh.cell_areas(), rtol=1e-14)
    # limiter: P0 projection with the mass weights, tracer integral conserved
    rng = np.random.default_rng(5)
    T = rng.normal(size=(mesh.num_cells, 4))
    Tl = orc.limit(T)
    integral = lambda q: float((mesh.cell_areas()*(orc.mean_w*q).sum(axis=1)).sum())
    assert abs(integral(Tl) - integral(T)) < 1e-12*abs(mesh.cell_areas().sum())


def test_quad_oracle_invariants():
    mesh, bath, uv, eta = quad_case(skew=0.3)
    orc = make_oracle_generic(mesh, bath)
    ru, re = orc.residual(np.zeros_like(uv), np.full_like(eta, 0.3))
    assert np.abs(ru).max() < 1e-9 and np.abs(re).max() == 0.0                 # lake at rest
    ru, re = orc.residual(uv, eta)
    assert abs(re.sum()) < 1e-12*np.abs(re).sum()                              # closed domain conserves volume
    # exact integration of the polynomial integrands: 2x2 and 3x3 tensor Gauss rules give the same residual
    from oracle.swe2d_oracle import SWEOracle
    o3 = SWEOracle(mesh.vertex_xy, mesh.cells, bath, quad_rule_points=3)
    ru3, re3 = o3.residual(uv, eta)
    assert rel_linf(ru3, ru) < 1e-13 and rel_linf(re3, re) < 1e-13
    # constant tracer stays constant; limiter keeps x-linear fields and conserves mass
    T = np.full((mesh.num_cells, 4), 4.5)
    assert np.abs(orc.tracer_residual(T, uv, eta)).max() < 1e-9
    u1, e1 = orc.ssprk33_step(uv, eta, 2.0)
    assert abs(orc.volume(e1) - orc.volume(eta))/orc.volume(eta) < 1e-13


@pytest.mark.parametrize('case', ['default', 'linear', 'no_lf', 'sources', 'manning', 'quad_drag', 'bcs', 'wind_bdrag'])
@pytest.mark.parametrize('skew', [0.0, 0.3])
def test_quad_numpy_and_c_restatements_agree(ref_so, case, skew):
    mesh, bath, uv, eta = quad_case(skew=skew, seed=1)
    kw = _cases(mesh, mesh.num_cells)[case]
    if case in ('manning', 'quad_drag', 'wind_bdrag'):
        eta = np.abs(eta)
    orc = make_oracle_generic(mesh, bath, **kw)
    ref = make_ref(mesh, bath, **kw)
    ku, ke = orc.tendency(uv, eta, 3.0)
    ku2, ke2 = ref.tendency(uv, eta, 3.0)
    assert rel_linf(ku2, ku) < 1e-13 and rel_linf(ke2, ke) < 1e-13
    u1, e1 = orc.ssprk33_step(uv, eta, 3.0)
    u2, e2 = ref.advance(uv, eta, 3.0, 1)
    assert rel_linf(u2, u1) < 1e-13 and rel_linf(e2, e1) < 1e-13


def test_quad_standing_wave_second_order(ref_so):
    """Linear standing wave of test/swe2d/test_standing_wave.py on quadrilaterals: order > 2*0.8 in space."""
    errs = []
    for nx in (20, 40, 80):
        lx, ly, depth = 5e3, 1e3, 100.0
        mesh = RectangleMesh(nx, 1, lx, ly, quadrilateral=True)
        bath = np.full(mesh.num_vertices, depth)
        period = 2*lx/math.sqrt(9.81*depth)
        n_steps = 40*nx
        eta0 = np.cos(np.pi*mesh.cell_xy()[:, :, 0]/lx)
        ref = make_ref(mesh, bath, use_nonlinear_equations=False)
        uv, eta = ref.advance(np.zeros((mesh.num_cells, 4, 2)), eta0, period/n_steps, n_steps)
        orc = make_oracle_generic(mesh, bath)
        errs.append(orc.l2_norm(eta - eta0)/math.sqrt(lx*ly))
    rates = [math.log(errs[i]/errs[i + 1], 2) for i in range(2)]
    assert all(r > 1.6 for r in rates), (errs, rates)
    assert errs[-1] < 1.25e-3


# ------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize('case', ['default', 'linear', 'no_lf', 'sources', 'manning', 'quad_drag', 'drag_fields',
                                  'nikuradse_field', 'bcs', 'wind_bdrag'])
@pytest.mark.parametrize('geometry', ['parallelograms', 'general'])
def test_quad_gpu_tendency_matches_oracle(hip_lib, case, geometry):
    """``general``: warped convex cells (bilinear map with a varying Jacobian, 4 x 4 mass solve per cell: the AFFINE = false
    kernels; thetis/solver2d.py:340-345 accepts any quadrilateral mesh)."""
    from thetis_amd import _lib
    from thetis_amd.device import Swe2dDevice
    mesh, bath, uv, eta = quad_case(skew=0.3, seed=1, warp=(0.3 if geometry == 'general' else 0.0))
    assert mesh.affine == (geometry == 'parallelograms')
    kw = _cases(mesh, mesh.num_cells)[case]
    if case in ('manning', 'quad_drag', 'wind_bdrag', 'drag_fields', 'nikuradse_field'):
        eta = np.abs(eta)
    dt = 3.0
    orc = make_oracle_generic(mesh, bath, **kw)
    dev_kw = {k: kw[k] for k in ('use_nonlinear_equations', 'use_lax_friedrichs_velocity') if k in kw}
    dev = Swe2dDevice(mesh, bath, dt, boundary_len=mesh.boundary_len, **dev_kw)
    if case == 'sources':
        dev.set_field(_lib.FIELD_CORIOLIS, kw['coriolis'][mesh.cells])
        dev.set_field(_lib.FIELD_ATMOSPHERIC_PRESSURE, kw['atmospheric_pressure'][mesh.cells])
        dev.set_field(_lib.FIELD_MOMENTUM_SOURCE, kw['momentum_source'])
        dev.set_field(_lib.FIELD_VOLUME_SOURCE, kw['volume_source'])
        dev.set_scalar(_lib.SCALAR_LINEAR_DRAG, 1e-3)
    if case == 'manning':
        dev.set_scalar(_lib.SCALAR_MANNING_DRAG, 0.02)
    if case == 'quad_drag':
        dev.set_scalar(_lib.SCALAR_QUADRATIC_DRAG, 0.0025)
        dev.set_scalar(_lib.SCALAR_NORM_SMOOTHER, 0.1)
    if case == 'drag_fields':
        dev.set_field(_lib.FIELD_MANNING_DRAG, kw['manning_drag_coefficient'][mesh.cells])
        dev.set_field(_lib.FIELD_LINEAR_DRAG, kw['linear_drag_coefficient'][mesh.cells])
    if case == 'nikuradse_field':
        dev.set_field(_lib.FIELD_NIKURADSE, kw['nikuradse_bed_roughness'][mesh.cells])
    if case == 'bcs':
        for m, funcs in _BCS.items():
            dev.set_bc(m, funcs)
    if case == 'wind_bdrag':
        dev.set_field(_lib.FIELD_WIND_STRESS, kw['wind_stress'])
        for m, funcs in kw['bnd_conditions'].items():
            dev.set_bc(m, funcs)
    dev.set_state(uv, eta)
    ku, ke = dev.tendency()
    ku_o, ke_o = orc.tendency(uv, eta, dt)
    assert rel_linf(ku, ku_o) < 1e-12 and rel_linf(ke, ke_o) < 1e-12
    uv2, eta2 = dev.get_state()
    assert np.array_equal(uv2, uv) and np.array_equal(eta2, eta)
    dev.advance(2)
    ud, ed = dev.get_state()
    uo, eo = orc.ssprk33_step(*orc.ssprk33_step(uv, eta, dt), dt)
    assert rel_linf(ud, uo) < 1e-11 and rel_linf(ed, eo) < 1e-11
    d = dev.diagnostics()
    assert math.isclose(d[2], orc.volume(ed), rel_tol=1e-12)
    assert math.isclose(math.sqrt(d[0]), orc.l2_norm(ed), rel_tol=1e-12)
    assert math.isclose(math.sqrt(d[1]), orc.l2_norm(ud), rel_tol=1e-12)
    dev.close()


@pytest.mark.gpu
def test_baseline_cfg1_quad_unit_rectangle(hip_lib, ref_so):
    """BASELINE.json configs[0] as literally written: 40x25-quad unit rectangle, DG(DQ)-P1, SSPRK33, h = 1, g = 9.81,
    eta0 = 0.01 cos(pi x), closed, dt = 1e-3, 100 steps (SURVEY.md 8d cfg 1(ii)) through FlowSolver2d."""
    mesh2d = RectangleMesh(40, 25, 1.0, 1.0, quadrilateral=True)
    bath = Function(get_functionspace(mesh2d, 'CG', 1)).assign(1.0)
    s = solver2d.FlowSolver2d(mesh2d, bath)
    o = s.options
    o.swe_timestepper_type = 'SSPRK33'
    o.swe_timestepper_options.use_automatic_timestep = False
    o.timestep = 1e-3
    o.simulation_end_time = 0.1
    o.simulation_export_time = 0.05
    o.check_volume_conservation_2d = True
    s.assign_initial_conditions(elev=lambda x, y: 0.01*np.cos(np.pi*x))
    eta0 = s.fields.elev_2d.cell_node_values().copy()
    s.iterate()
    assert s.iteration == 100
    ref = make_ref(mesh2d, bath.dat.data_ro)
    u_r, e_r = ref.advance(np.zeros((1000, 4, 2)), eta0, 1e-3, 100)
    assert rel_linf(s.fields.elev_2d.cell_node_values(), e_r) < 1e-10
    assert rel_linf(s.fields.uv_2d.cell_node_values(), u_r) < 1e-10
    vol, rerr = s.callbacks['export']['volume2d']()
    assert abs(rerr) < 1e-12


def test_project_on_general_quadrilaterals_is_the_l2_projection():
    """Function.project on warped cells: right-hand side weighted with det J at the Gauss points, true 4 x 4 mass matrix per cell -
    a bilinear function of the reference coordinates is reproduced, and the projection of x y conserves its integral."""
    mesh, _, _, _ = quad_case(nx=6, ny=5, lx=3.0, ly=2.0, skew=0.1, warp=0.3)
    assert not mesh.affine
    P1DG = get_functionspace(mesh, 'DG', 1)
    f = Function(P1DG).project(lambda x, y: 2.0 + 0.5*x - 0.25*y)           # affine in x: exactly representable on every cell
    v = f.cell_node_values()
    p = mesh.cell_xy()
    assert np.abs(v - (2.0 + 0.5*p[:, :, 0] - 0.25*p[:, :, 1])).max() < 1e-12
    g = Function(P1DG).project(lambda x, y: x*y).cell_node_values()
    orc = make_oracle_generic(mesh, np.ones(mesh.num_vertices))
    integral = sum(float(np.sum(w*(g @ phi))) for phi, _, w in orc.cell_quad)
    exact = sum(float(np.sum(w*((p[:, :, 0] @ phi)*(p[:, :, 1] @ phi)))) for phi, _, w in
                make_oracle_generic(mesh, np.ones(mesh.num_vertices), quad_rule_points=4).cell_quad)
    assert abs(integral - exact) < 1e-3*abs(exact)         # (the right-hand side is integrated with the 2 x 2 rule)


@pytest.mark.gpu
def test_general_quadrilaterals_through_flowsolver(hip_lib):
    """FlowSolver2d on a mesh of warped convex quadrilaterals (thetis/solver2d.py:340-345 accepts any quadrilateral mesh): a seiche
    with bottom friction and an open boundary, 60 steps against the numpy oracle, volume conservation callback on a closed basin."""
    mesh2d, _, _, _ = quad_case(nx=16, ny=8, lx=8e3, ly=4e3, skew=0.0, warp=0.3)
    assert not mesh2d.affine
    x, y = mesh2d.vertex_xy.T
    bath_v = 10.0 + 2.0*np.sin(x/2e3)
    for closed in (True, False):
        bath = Function(get_functionspace(mesh2d, 'CG', 1)).assign(bath_v)
        s = solver2d.FlowSolver2d(mesh2d, bath)
        o = s.options
        o.swe_timestepper_type = 'SSPRK33'
        o.swe_timestepper_options.use_automatic_timestep = False
        o.timestep = 2.0
        o.simulation_end_time = 120.0
        o.simulation_export_time = 60.0
        o.no_exports = True
        o.check_volume_conservation_2d = True
        kw = {}
        if not closed:
            o.manning_drag_coefficient = Constant(0.02)
            s.bnd_functions['shallow_water'] = {2: {'elev': Constant(0.05)}}
            kw = dict(manning_drag_coefficient=0.02, bnd_conditions={2: {'elev': 0.05}})
        s.assign_initial_conditions(elev=lambda x, y: 0.1*np.cos(np.pi*x/8e3))
        eta0 = s.fields.elev_2d.cell_node_values().copy()
        s.iterate()
        assert s.iteration == 60
        orc = make_oracle_generic(mesh2d, bath_v, **kw)
        u_o, e_o = np.zeros((mesh2d.num_cells, 4, 2)), eta0
        for _ in range(60):
            u_o, e_o = orc.ssprk33_step(u_o, e_o, 2.0)
        assert rel_linf(s.fields.elev_2d.cell_node_values(), e_o) < 1e-10
        assert rel_linf(s.fields.uv_2d.cell_node_values(), u_o) < 1e-10
        if closed:
            vol, rerr = s.callbacks['export']['volume2d']()
            assert abs(rerr) < 1e-12 and math.isclose(vol, orc.volume(e_o), rel_tol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['quad', 'quadgen'])
def test_quad_two_ranks_on_one_gpu(tmp_path, hip_lib, kind):
    """Strip partitions of a quadrilateral mesh (``quadgen``: general convex cells), two ranks sharing one GPU == one device, bitwise."""
    import dist_worker
    from thetis_amd.device import Swe2dDevice
    dist_worker.CASE = kind
    try:
        mesh, bath, uv, eta = dist_worker._case()
        assert mesh.affine == (kind == 'quad')
        dist_worker.run_workers(dist_worker.gpu_worker, 2, 3, str(tmp_path), axis=0, case=kind)
        u_p, e_p, _ = dist_worker.gather(str(tmp_path), 2, mesh.num_cells)
    finally:
        dist_worker.CASE = 'channel'
    dev = Swe2dDevice(mesh, bath, 2.0)
    dev.set_state(uv, eta)
    dev.advance(3)
    u_s, e_s = dev.get_state()
    assert np.array_equal(u_p, u_s) and np.array_equal(e_p, e_s)
    dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['default', 'lf', 'value_bc'])
@pytest.mark.parametrize('geometry', ['parallelograms', 'general'])
def test_quad_tracer_and_limiter_match_oracle(hip_lib, case, geometry):
    from thetis_amd.device import Swe2dDevice
    mesh, bath, uv, eta = quad_case(skew=0.3, seed=2, warp=(0.3 if geometry == 'general' else 0.0))
    rng = np.random.default_rng(3)
    T = rng.normal(size=(mesh.num_cells, 4))
    src = 1e-3*rng.normal(size=T.shape)
    dt = 3.0
    orc = make_oracle_generic(mesh, bath)
    dev = Swe2dDevice(mesh, bath, dt, boundary_len=mesh.boundary_len)
    tid = dev.add_tracer()
    kw = {}
    if case == 'lf':
        kw = dict(use_lax_friedrichs_tracer=True, lax_friedrichs_tracer_scaling_factor=0.7,
                  tracer_advective_velocity_factor=0.9, source=src)
        dev.tracer_set_options(True, 0.7, 0.9)
        dev.tracer_set_source(tid, src)
    if case == 'value_bc':
        kw = dict(bnd_conditions={1: {'value': 2.0}, 3: {'value': -1.0}})
        dev.tracer_set_bc(tid, 1, 2.0)
        dev.tracer_set_bc(tid, 3, -1.0)
    dev.set_state(uv, eta)
    dev.tracer_set_state(tid, T)
    assert np.array_equal(dev.tracer_get_state(tid), T)
    assert rel_linf(dev.tracer_tendency(tid), orc.tracer_tendency(T, uv, eta, dt, **kw)) < 1e-12
    for s in range(3):
        dev.tracer_solve_stage(tid, s)
    T1 = dev.tracer_get_state(tid)
    assert rel_linf(T1, orc.tracer_ssprk33_step(T, uv, eta, dt, **kw)) < 1e-12
    d = dev.tracer_diagnostics(tid)
    assert math.isclose(d[0], orc.tracer_mass(T1, eta), rel_tol=1e-12)
    assert d[2] == T1.min() and d[3] == T1.max()
    dev.tracer_set_state(tid, T)
    dev.tracer_limit(tid)
    assert rel_linf(dev.tracer_get_state(tid), orc.limit(T)) < 1e-14
    dev.close()


@pytest.mark.gpu
def test_general_quadrilaterals_coupled_steps(hip_lib):
    """Warped cells through the coupled step (shallow water, tracer with the updated velocity, limiter from the means the last
    tracer stage writes): five steps against the numpy oracle, volume and tracer integral conserved."""
    from thetis_amd import _lib
    from thetis_amd.device import Swe2dDevice
    mesh, bath, uv, eta = quad_case(nx=12, ny=8, skew=0.1, seed=4, warp=0.35, amp_eta=0.2, amp_u=0.2)
    assert not mesh.affine
    rng = np.random.default_rng(8)
    T = 1.0 + 0.5*rng.normal(size=(mesh.num_cells, 4))
    dt = 2.0
    orc = make_oracle_generic(mesh, bath)
    dev = Swe2dDevice(mesh, bath, dt, boundary_len=mesh.boundary_len)
    tid = dev.add_tracer()
    dev.set_state(uv, eta)
    dev.tracer_set_state(tid, T)
    v0, m0 = dev.diagnostics()[2], dev.tracer_diagnostics(tid)[0]
    dev.advance_coupled(5, tracer_only=False, use_limiter=True)
    u_o, e_o, T_o = uv, eta, T
    for _ in range(5):
        u_o, e_o = orc.ssprk33_step(u_o, e_o, dt)
        T_o = orc.limit(orc.tracer_ssprk33_step(T_o, u_o, e_o, dt))
    u_d, e_d = dev.get_state()
    assert rel_linf(u_d, u_o) < 1e-11 and rel_linf(e_d, e_o) < 1e-11
    assert rel_linf(dev.tracer_get_state(tid), T_o) < 1e-10
    assert math.isclose(dev.diagnostics()[2], v0, rel_tol=1e-13)
    assert math.isclose(dev.tracer_diagnostics(tid)[0], orc.tracer_mass(T_o, e_o), rel_tol=1e-10)
    dev.close()


@pytest.mark.gpu
def test_demo_2d_tracer_solid_body_rotation(hip_lib, ref_so):
    """demos/demo_2d_tracer.py:19-137: LeVeque bell + cone + slotted cylinder on UnitSquareMesh(40, 40, quadrilateral),
    tracer_only, SSPRK33, dt = pi/300, one rotation; checked step by step against the CPU restatement."""
    from oracle.ref_lib import RefTracer
    from thetis_amd import UnitSquareMesh
    mesh2d = UnitSquareMesh(40, 40, quadrilateral=True)
    P1_2d = get_functionspace(mesh2d, 'CG', 1)
    bathymetry2d = Function(P1_2d).assign(1.0)
    solver_obj = solver2d.FlowSolver2d(mesh2d, bathymetry2d)
    options = solver_obj.options
    options.add_tracer_2d('tracer_2d', 'Depth averaged tracer', 'Tracer2d', source=None, diffusivity=None)
    options.tracer_only = True
    t_end, timestep = 2*math.pi, math.pi/300.0
    options.tracer_timestepper_type = 'SSPRK33'
    options.timestep = timestep
    options.simulation_end_time = t_end
    options.simulation_export_time = math.pi/15.0
    options.tracer_timestepper_options.use_automatic_timestep = False
    options.use_lax_friedrichs_tracer = False
    options.use_limiter_for_tracers = False
    solver_obj.bnd_functions['tracer_2d'] = {'on_boundary': {'value': Constant(1.0)}}

    def q0(x, y):
        bell = 0.25*(1 + np.cos(np.pi*np.minimum(np.sqrt((x - 0.25)**2 + (y - 0.5)**2)/0.15, 1.0)))
        cone = 1.0 - np.minimum(np.sqrt((x - 0.5)**2 + (y - 0.25)**2)/0.15, 1.0)
        cyl = np.where(np.sqrt((x - 0.5)**2 + (y - 0.75)**2) < 0.15,
                       np.where((x > 0.475) & (x < 0.525) & (y < 0.85), 0.0, 1.0), 0.0)
        return 1.0 + bell + cone + cyl
    q_init = Function(P1_2d).interpolate(q0)
    solver_obj.assign_initial_conditions(uv=lambda x, y: (0.5 - y, x - 0.5), tracer_2d=q_init)
    it = solver_obj.create_iterator()
    t = 0
    while t < t_end - timestep:
        t = next(it)
    q = solver_obj.fields.tracer_2d.cell_node_values()
    q_i = q_init.cell_node_values()
    orc = make_oracle_generic(mesh2d, np.ones(mesh2d.num_vertices))
    l2 = orc.l2_norm(q - q_i)/orc.l2_norm(q_i)
    assert l2 < 0.1                                         # dispersion of the unlimited scheme on a 40x40 mesh
    # same number of steps with the CPU restatement
    ref = make_ref(mesh2d, np.ones(mesh2d.num_vertices))
    rt = RefTracer(ref, cell_topo_vertices=mesh2d.cells)
    uv = solver_obj.fields.uv_2d.cell_node_values()
    T = q_i.copy()
    # the generator yields after advance() and before the counter is incremented (solver2d.py:1116-1125)
    for _ in range(solver_obj.iteration + 1):
        T = rt.step(T, uv, timestep)
    assert rel_linf(q, T) < 1e-10


@pytest.mark.gpu
def test_demo_2d_multiple_tracers(hip_lib, ref_so):
    """demos/demo_2d_multiple_tracers.py: three labelled tracers (bell, cone, slotted cylinder) advected together in
    tracer-only mode, each with its own boundary dict; a quarter rotation, every tracer against the CPU restatement."""
    from oracle.ref_lib import RefTracer
    from thetis_amd import UnitSquareMesh
    mesh2d = UnitSquareMesh(40, 40, quadrilateral=True)
    P1_2d = get_functionspace(mesh2d, 'CG', 1)
    bathymetry2d = Function(P1_2d).assign(1.0)
    solver_obj = solver2d.FlowSolver2d(mesh2d, bathymetry2d)
    labels = ['bell_2d', 'cone_2d', 'slot_cyl_2d']
    names = ['Gaussian bell', 'Cone', 'Slotted cylinder']
    filenames = ['GaussianBell2d', 'Cone2d', 'SlottedCylinder2d']
    options = solver_obj.options
    options.tracer_only = True
    options.fields_to_export = labels
    for label, name, filename in zip(labels, names, filenames):
        options.add_tracer_2d(label, name, filename, source=None, diffusivity=None)
        solver_obj.bnd_functions[label] = {'on_boundary': {'value': Constant(1.0)}}
    timestep = math.pi/300.0
    options.tracer_timestepper_type = 'SSPRK33'
    options.timestep = timestep
    options.simulation_end_time = math.pi/2
    options.simulation_export_time = math.pi/15.0
    options.tracer_timestepper_options.use_automatic_timestep = False
    options.use_lax_friedrichs_tracer = False
    options.use_limiter_for_tracers = False
    options.no_exports = True

    def r(x, y, cx, cy):
        return np.sqrt((x - cx)**2 + (y - cy)**2)
    inits = {
        'bell_2d': lambda x, y: 1.0 + 0.25*(1 + np.cos(np.pi*np.minimum(r(x, y, 0.25, 0.5)/0.15, 1.0))),
        'cone_2d': lambda x, y: 1.0 + 1.0 - np.minimum(r(x, y, 0.5, 0.25)/0.15, 1.0),
        'slot_cyl_2d': lambda x, y: 1.0 + np.where(r(x, y, 0.5, 0.75) < 0.15,
                                                   np.where((x > 0.475) & (x < 0.525) & (y < 0.85), 0.0, 1.0), 0.0),
    }
    funcs = {label: Function(P1_2d).interpolate(f) for label, f in inits.items()}
    solver_obj.assign_initial_conditions(uv=lambda x, y: (0.5 - y, x - 0.5), **funcs)
    solver_obj.iterate()
    ref = make_ref(mesh2d, np.ones(mesh2d.num_vertices))
    rt = RefTracer(ref, cell_topo_vertices=mesh2d.cells)
    uv = solver_obj.fields.uv_2d.cell_node_values()
    assert solver_obj.iteration == 150
    for label in labels:
        T = funcs[label].cell_node_values().copy()
        for _ in range(solver_obj.iteration):
            T = rt.step(T, uv, timestep)
        q = solver_obj.fields[label].cell_node_values()
        assert rel_linf(q, T) < 1e-10, label
    # the three fields are different tracers, not copies
    assert np.abs(solver_obj.fields['bell_2d'].cell_node_values() - solver_obj.fields['cone_2d'].cell_node_values()).max() > 0.1



@pytest.mark.gpu
@pytest.mark.parametrize('case', ['parallelograms_open', 'skewed_sources', 'general', 'general_sources', 'ragged_random_numbering', 'linear_no_lf',
                                  'by_the_rule_900k', 'coupled_tracer_270k'])
def test_fused_stage_pair_on_quadrilaterals_gives_the_bits_of_the_stage_launches(hip_lib, ref_so, monkeypatch, case):
    """csrc/swe2d_fuse.h, swe_fuse12_quad_kernel (round 6): stages 1 and 2 of a step in one launch on tiles of up to 192 quadrilaterals
    + their ring, U(1) in LDS; the arithmetic is swe_quad_stage_cell, the function the stage launches call.  Bit for bit
    SWE2D_OPT_FUSED_STAGES = 0: parallelograms with open boundaries and walls, a skewed mesh with Coriolis + Manning + wind, general
    (warped) cells with and without source terms, partial tiles in a random numbering, linear equations without Lax-Friedrichs, a
    900 k-cell mesh where the library takes it by itself (beyond the Infinity Cache), the shallow-water half of swe2d_advance_coupled; and one step
    against the oracle's C restatement."""
    from thetis_amd import _lib
    from thetis_amd.device import Swe2dDevice
    monkeypatch.setenv('THETIS_AMD_FLOW', '0')
    kw, reorder, forced = {}, 'auto', '1'
    if case == 'by_the_rule_900k':                    # (the rule: whole meshes of >= 850 k quadrilaterals)
        mesh, bath, uv, eta = quad_case(nx=1200, ny=750, lx=100e3, ly=62.5e3, seed=5, amp_eta=0.3, amp_u=0.2)
        forced = None
    elif case == 'coupled_tracer_270k':
        mesh, bath, uv, eta = quad_case(nx=600, ny=450, lx=100e3, ly=75e3, seed=5, amp_eta=0.3, amp_u=0.2)
    elif case == 'skewed_sources':
        mesh, bath, uv, eta = quad_case(nx=90, ny=70, lx=100e3, ly=70e3, seed=6, amp_eta=0.05, amp_u=0.05, skew=0.2)
    elif case.startswith('general'):
        mesh, bath, uv, eta = quad_case(nx=90, ny=70, lx=100e3, ly=70e3, seed=7, amp_eta=0.05, amp_u=0.05, warp=0.25)
    elif case == 'ragged_random_numbering':
        m0, bath, uv0, eta0 = quad_case(nx=53, ny=31, lx=100e3, ly=60e3, seed=6, amp_eta=0.3, amp_u=0.2)
        cperm = np.random.default_rng(11).permutation(m0.num_cells)
        mesh = m0.renumbered(cperm)
        uv, eta, reorder = uv0[cperm], eta0[cperm], None
    else:
        mesh, bath, uv, eta = quad_case(nx=120, ny=90, lx=100e3, ly=75e3, seed=8, amp_eta=0.3, amp_u=0.2)
        if case == 'linear_no_lf':
            kw = dict(use_nonlinear_equations=False, use_lax_friedrichs_velocity=False)
    cxy = mesh.cell_xy()
    src = case.endswith('sources')
    out = []
    for fuse in ('0', forced):
        if fuse is None:
            monkeypatch.delenv('THETIS_AMD_FUSE12', raising=False)
        else:
            monkeypatch.setenv('THETIS_AMD_FUSE12', fuse)
        dev = Swe2dDevice(mesh, bath, 0.5, reorder=reorder, boundary_len=mesh.boundary_len, **kw)
        m = mesh.boundary_markers
        dev.set_bc(m[1], {'elev': 0.1})
        dev.set_bc(m[2], {'un': 0.05})
        if src:
            dev.set_field(_lib.FIELD_CORIOLIS, 1e-4*(1.0 + cxy[:, :, 1]/50e3))
            dev.set_scalar(_lib.SCALAR_MANNING_DRAG, 0.02)
            dev.set_field(_lib.FIELD_WIND_STRESS, np.stack([0.1*np.sin(cxy[:, :, 0]/2e4), 0.05*np.cos(cxy[:, :, 1]/1e4)], axis=2))
        dev.set_state(uv, np.abs(eta) if src else eta)
        assert dev.fused_pair_info()[0] == (fuse != '0'), (case, fuse, dev.fused_pair_info())
        if case == 'coupled_tracer_270k':
            tid = dev.add_tracer()
            dev.tracer_set_state(tid, np.where(cxy[:, :, 0] < 40e3, 0.0, 30.0))
            dev.advance_coupled(3, tracer_only=False, use_limiter=True)
            out.append(dev.get_state() + (dev.tracer_get_state(tid),))
        else:
            dev.advance(1)
            first = dev.get_state()
            dev.advance(2)
            dev.advance(2)
            out.append(first + dev.get_state())
        dev.close()
    assert np.isfinite(out[0][1]).all()
    for a_, b_ in zip(out[0], out[1]):
        assert np.array_equal(a_, b_)
    if case == 'parallelograms_open':                 # ... and the fused launch against the oracle directly
        ref = make_ref(mesh, bath, bnd_conditions={mesh.boundary_markers[1]: {'elev': 0.1}, mesh.boundary_markers[2]: {'un': 0.05}})
        ur, er = ref.advance(uv, eta, 0.5, 1)
        assert rel_linf(out[1][0], ur) < 1e-11 and rel_linf(out[1][1], er) < 1e-11
