"""
CPU tests that pin the oracle (numpy restatement + C restatement) with the reference's own known-answer tests
(SURVEY.md 8c): lake at rest, closed-domain volume conservation, the two formulations against each other for every
term, linear standing wave (test/swe2d/test_standing_wave.py:21-35,96-97), SSPRK33 order 3
(test/time_integration/test_convergence_ode.py:152-186), second-order spatial convergence.
"""
import json
import math
import os

import numpy as np
import pytest

from helpers import channel_case, make_oracle, make_ref, rel_linf
from thetis_amd.mesh import PeriodicRectangleMesh, RectangleMesh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lake_at_rest_is_steady_over_variable_bathymetry():
    mesh, bath, uv, eta = channel_case(nx=10, ny=4)
    orc = make_oracle(mesh, bath)
    ru, re = orc.residual(np.zeros_like(uv), np.full_like(eta, 0.3))
    # R = 0 up to round-off of g*eta*|F|  (shallowwater_eq.py:361-366,422-427)
    scale = 9.81*0.3*np.hypot(mesh.lx/10, mesh.ly/4)
    assert np.abs(ru).max() < 1e-13*scale*10
    assert np.abs(re).max() == 0.0


@pytest.mark.parametrize('nonlin', [True, False])
def test_closed_domain_conserves_volume(nonlin):
    mesh, bath, uv, eta = channel_case(nx=10, ny=4, seed=2)
    orc = make_oracle(mesh, bath, use_nonlinear_equations=nonlin)
    _, re = orc.residual(uv, eta)
    # sum over all test functions of R_eta = d/dt int eta dx = 0 on a closed domain
    assert abs(re.sum()) < 1e-12*np.abs(re).sum()
    u1, e1 = orc.ssprk33_step(uv, eta, 2.0)
    v0, v1 = orc.volume(eta), orc.volume(e1)
    assert abs(v1 - v0)/v0 < 1e-13


_BCS1 = {1: {'elev': 0.3}, 2: {'un': 0.2}, 3: {'flux': 1e4, 'elev': 0.1}}
_BCS2 = {4: {'uv': (0.1, -0.2)}, 1: {'flux': -3e3}, 2: {'elev': 0.1, 'uv': (0.3, 0.1)}, 3: {'elev': -0.1, 'un': 0.05}}


@pytest.mark.parametrize('case', ['default', 'linear', 'no_lf', 'half_lf', 'sources', 'manning', 'quad_drag', 'bcs1', 'bcs2',
                                  'wind_bdrag'])
def test_numpy_and_c_restatements_agree(ref_so, case):
    """Literal UFL restatement (quadrature everywhere, facet loop) vs element-centric closed forms in C."""
    mesh, bath, uv, eta = channel_case(seed=1)
    x, y = mesh.vertex_xy.T
    rng = np.random.default_rng(9)
    n = mesh.num_cells
    kw = {
        'default': {}, 'linear': dict(use_nonlinear_equations=False), 'no_lf': dict(use_lax_friedrichs_velocity=False),
        'half_lf': dict(lax_friedrichs_velocity_scaling_factor=0.5),
        'sources': dict(coriolis=1e-4*(1 + y/30e3), linear_drag_coefficient=1e-3,
                        atmospheric_pressure=1e5 + 300*np.sin(x/2e4), momentum_source=1e-3*rng.normal(size=(n, 3, 2)),
                        volume_source=1e-3*rng.normal(size=(n, 3))),
        'manning': dict(manning_drag_coefficient=0.02),
        'quad_drag': dict(quadratic_drag_coefficient=0.0025, norm_smoother=0.1),
        'bcs1': dict(bnd_conditions=_BCS1), 'bcs2': dict(bnd_conditions=_BCS2),
        'wind_bdrag': dict(wind_stress=0.1*rng.normal(size=(n, 3, 2)),
                           bnd_conditions={3: {'drag': 0.0025}, 1: {'drag': 0.01, 'elev': 0.1}}),
    }[case]
    if case in ('manning', 'quad_drag', 'wind_bdrag'):
        eta = np.abs(eta)
    orc = make_oracle(mesh, bath, **kw)
    ref = make_ref(mesh, bath, **kw)
    ku, ke = orc.tendency(uv, eta, 3.0)
    ku2, ke2 = ref.tendency(uv, eta, 3.0)
    assert rel_linf(ku2, ku) < 1e-13
    assert rel_linf(ke2, ke) < 1e-13
    u1, e1 = orc.ssprk33_step(uv, eta, 3.0)
    u2, e2 = ref.advance(uv, eta, 3.0, 1)
    assert rel_linf(u2, u1) < 1e-13 and rel_linf(e2, e1) < 1e-13


def test_periodic_mesh_translation_invariance():
    """On an x-periodic mesh a state shifted by one column gives the shifted tendency (pins periodic connectivity)."""
    nx, ny = 8, 4
    mesh = PeriodicRectangleMesh(nx, ny, 8.0, 4.0, direction='x')
    from oracle.swe2d_oracle import SWEOracle
    orc = SWEOracle(mesh.vertex_xy, mesh.cells, np.full(mesh.num_vertices, 2.0), topo_vertex=mesh.topo_vertex)
    assert len(orc.ext_facets) == 2*nx                      # only y = 0 and y = Ly are boundaries
    rng = np.random.default_rng(4)
    n = mesh.num_cells
    uv = rng.normal(size=(n, 3, 2))*0.1
    eta = rng.normal(size=(n, 3))*0.1
    ku, ke = orc.tendency(uv, eta, 0.01)
    # cells are ordered row by row (2*nx per row): shift every row by one quad (2 triangles)
    def shift(a):
        b = a.reshape((ny, 2*nx) + a.shape[1:])
        return np.roll(b, 2, axis=1).reshape(a.shape)
    ku_s, ke_s = orc.tendency(shift(uv), shift(eta), 0.01)
    assert rel_linf(ku_s, shift(ku)) < 1e-12 and rel_linf(ke_s, shift(ke)) < 1e-12


def _standing_wave(ref_so, nx, n_steps_per_period, nonlin=False, periods=1.0):
    """Linear standing wave eta = cos(pi x/L) in a closed channel, h = 100, L = 5 km (test_standing_wave.py:21-35)."""
    lx, ly, depth = 5e3, 1e3, 100.0
    mesh = RectangleMesh(nx, 1, lx, ly)
    bath = np.full(mesh.num_vertices, depth)
    c = math.sqrt(9.81*depth)
    period = 2*lx/c
    dt = period/n_steps_per_period
    cxy = mesh.cell_xy()
    eta0 = np.cos(np.pi*cxy[:, :, 0]/lx)
    uv0 = np.zeros((mesh.num_cells, 3, 2))
    ref = make_ref(mesh, bath, use_nonlinear_equations=nonlin)
    n_steps = int(round(periods*n_steps_per_period))
    uv, eta = ref.advance(uv0, eta0, dt, n_steps)
    orc = make_oracle(mesh, bath)
    err = orc.l2_norm(eta - eta0)/math.sqrt(lx*ly)
    return err, (mesh, uv, eta, eta0)


def test_linear_standing_wave_returns_after_one_period(ref_so):
    # reference bar for the 2nd-order CrankNicolson stepper with 40 steps: rel_err < 1.25e-3 (test_standing_wave.py:12-13)
    err, _ = _standing_wave(ref_so, nx=100, n_steps_per_period=2000)
    assert err < 1.25e-3
    # and the explicit 3rd-order stepper on a DG-P1 mesh is far below that
    assert err < 2e-4


def test_standing_wave_second_order_in_space(ref_so):
    errs = [_standing_wave(ref_so, nx=nx, n_steps_per_period=40*nx)[0] for nx in (20, 40, 80)]
    rates = [math.log(errs[i]/errs[i + 1], 2) for i in range(2)]
    assert all(r > 2*0.8 for r in rates), (errs, rates)        # reference criterion: order > 2*0.8


def test_ssprk33_third_order_in_time_on_pde(ref_so):
    """Same mesh, dt halved: difference to a fine-dt solution on the same mesh decays with order 3."""
    nx = 20
    sols = {}
    for nsp in (400, 800, 1600, 12800):
        _, (mesh, uv, eta, _) = _standing_wave(ref_so, nx=nx, n_steps_per_period=nsp, nonlin=True, periods=0.5)
        sols[nsp] = eta
    errs = [np.abs(sols[n] - sols[12800]).max() for n in (400, 800, 1600)]
    rates = [math.log(errs[i]/errs[i + 1], 2) for i in range(2)]
    assert all(abs(r - 3.0) < 0.15 for r in rates), (errs, rates)


def test_ssprk33_shu_osher_order_on_ode():
    """a' = alpha b, b' = -alpha a integrated with the golden Shu-Osher coefficients: slope 3.0 +- 5 %
    (test/time_integration/test_convergence_ode.py:152-166,186)."""
    with open(os.path.join(ROOT, 'tests', 'golden', 'shuosher_ssprk33.json')) as f:
        g = json.load(f)
    al, be = np.array(g['alpha']), np.array(g['beta'])
    alpha_ode = 2*np.pi
    f = lambda y: np.array([alpha_ode*y[1], -alpha_ode*y[0]])
    errs, refs = [], [1, 2, 3, 4]
    for r in refs:
        n = 20*2**r
        dt = 1.0/n
        y = np.array([0.0, 1.0])
        vals = [y.copy()]
        for _ in range(n):
            stage = [y]
            for i in range(3):
                k = dt*f(stage[i])
                stage.append(be[i + 1][i]*k + sum(al[i + 1][j]*stage[j] for j in range(i + 1)))
            y = stage[3]
            vals.append(y.copy())
        t = np.arange(n + 1)*dt
        exact = np.vstack((np.sin(alpha_ode*t), np.cos(alpha_ode*t))).T
        errs.append(np.sqrt(np.mean((np.array(vals) - exact)**2)))
    slope = np.polyfit(np.log10(1.0/(20*2.0**np.array(refs))), np.log10(errs), 1)[0]
    assert abs(slope - 3.0)/slope < 0.05


def test_rossby_soliton_metrics_do_not_diverge(ref_so):
    """test/swe2d/test_rossby_wave.py::test_convergence[SSPRK33-dg-dg]: levels 24 -> 48, t_end = 30, g = 1, h = 1,
    f = y, uv = 0 on the channel walls, periodic in x; peak height / phase speed metrics vs FVCOM (0.1567020, 47.18)."""
    import rossby
    from oracle.ref_lib import RefSWE
    res = []
    for level in (24, 48):
        mesh = rossby.rossby_mesh(level)
        cxy = mesh.cell_xy()
        u, v = rossby.asymptotic_uv(cxy[:, :, 0], cxy[:, :, 1])
        uv0 = np.stack([u, v], axis=2)
        eta0 = rossby.asymptotic_elev(cxy[:, :, 0], cxy[:, :, 1])
        ref = RefSWE(cxy, mesh.cell_nbr, mesh.cell_nbr_facet, np.ones((mesh.num_cells, 3)), g=1.0,
                     coriolis=cxy[:, :, 1], bnd_conditions={3: {'uv': (0.0, 0.0)}, 4: {'uv': (0.0, 0.0)}},
                     boundary_len=mesh.boundary_len)
        dt = 0.96/level
        n_steps = int(round(30.0/dt))
        uv, eta = ref.advance(uv0, eta0, dt, n_steps)
        assert np.isfinite(eta).all()
        res.append(rossby.metrics(cxy, eta))
    rossby.check_convergence(res[0], res[1])
    # the soliton keeps most of its height on the finer mesh (FVCOM at dx = 0.25: 0.85 / 0.818, test/swe2d/data/FVCOM.json);
    # the phase metric is normalised for T = 120 and only has to be non-divergent at t_end = 30
    assert all(0.8 < m < 1.1 for m in res[1][:2]), res


def _pressure_forcing_errors(run):
    """test/swe2d/test_atmospheric_pressure.py: steady balance eta = A cos(pi x/L) cos(pi y/L) between the atmospheric
    pressure gradient and the elevation gradient (Manning mu = 1 damps the transient), closed 10 km box, h = 5,
    SSPRK33 on n = 2, 4, 8; returns the three L2 errors / sqrt(area).
    Time step: the reference uses dt = 20/2^i; at n = 2 that is the edge of SSPRK33's real-axis stability interval for the
    stiff drag term (2 C_D |u|/H dt ~ 2.8 with this build's 6-point cell rule vs the limit 2.51), so whether it survives
    depends on Firedrake's degree-3 cell quadrature, which is unknowable here ([FD-assumed], parity unpinned).  The test
    measures a STEADY state, so dt = 10/2^i is used: same criterion, same answer."""
    lx = ly = 10000.0
    A = 2.0
    errs = []
    for i in range(3):
        n, dt = 2**(i + 1), 10.0/2**i
        mesh = RectangleMesh(n, n, lx, ly)
        errs.append(run(mesh, dt, 43200.0,
                        lambda x, y: -1000.0*9.81*A*np.cos(np.pi*x/lx)*np.cos(np.pi*y/ly),
                        lambda x, y: A*np.cos(np.pi*x/lx)*np.cos(np.pi*y/ly))/math.sqrt(lx*ly))
    return np.array(errs)


def check_pressure_forcing_orders(errs):
    expected_order = 2
    assert all(errs[:-1]/errs[1:] > 2.**expected_order*0.75), errs           # test_atmospheric_pressure.py:93
    assert errs[0]/errs[-1] > (2.**expected_order)**(len(errs) - 1)*0.75, errs   # :94


def test_atmospheric_pressure_balance_second_order(ref_so):
    def run(mesh, dt, t_end, patm_fn, eta_fn):
        cxy = mesh.cell_xy()
        patm = patm_fn(cxy[:, :, 0], cxy[:, :, 1])                          # DG-P1 interpolation (:60-61)
        bath = np.full(mesh.num_vertices, 5.0)
        ref = make_ref(mesh, bath, manning_drag_coefficient=1.0, atmospheric_pressure=patm)
        uv0 = np.zeros((mesh.num_cells, 3, 2))
        uv0[:, :, 0] = 1e-7
        uv, eta = ref.advance(uv0, np.zeros((mesh.num_cells, 3)), dt, int(round(t_end/dt)))
        orc = make_oracle(mesh, bath)
        return orc.l2_norm(eta - orc.project(eta_fn))
    check_pressure_forcing_orders(_pressure_forcing_errors(run))


def test_steady_state_channel_linear_drag():
    """test/swe2d/test_steady_state_channel.py with explicit marching instead of the implicit solve: linear equations,
    linear drag f = g/lx, inflow un = -1 on marker 1, elev = 0 on marker 2, initial u = (1, 0); the steady free surface is
    eta = 1 - x/lx (the reference's criterion: L2 error / sqrt(area) < 1e-2)."""
    lx, ly = 5e3, 1e3
    mesh = RectangleMesh(10, 1, lx, ly)
    bath = np.full(mesh.num_vertices, 100.0)
    g = 9.81
    orc = make_oracle(mesh, bath, use_nonlinear_equations=False, linear_drag_coefficient=g/lx,
                      bnd_conditions={1: {'un': -1.0}, 2: {'elev': 0.0}})
    n = mesh.num_cells
    uv = np.zeros((n, 3, 2))
    uv[:, :, 0] = 1.0
    eta = np.zeros((n, 3))
    dt = 2.0
    for _ in range(int(6000.0/dt)):                  # ~12 drag time scales (1/f = 510 s)
        uv, eta = orc.ssprk33_step(uv, eta, dt)
    x = mesh.cell_xy()[:, :, 0]
    err2 = 0.0
    for bary, _, wA in orc.cell_quad:
        err2 += np.sum(wA*((eta @ bary) - (1.0 - (x @ bary)/lx))**2)
    assert math.sqrt(err2/(lx*ly)) < 1e-2
    assert np.abs(uv[:, :, 0] - 1.0).max() < 2e-2 and np.abs(uv[:, :, 1]).max() < 1e-2


def test_blocked_baseline_stepper_gives_the_bits_of_the_plain_one(ref_so):
    """bench.py's cpu_baseline times swe2d_ref_advance_blocked (one OpenMP region, owner-touched cell blocks, fused stage
    update on three rotating buffers): same operator, same expression order, bit for bit the stepper the parity tests use."""
    from helpers import channel_case, make_ref, quad_case
    for mesh, bath, uv, eta in (channel_case(nx=30, ny=14, seed=3), quad_case(nx=14, ny=9, seed=4)):
        ref = make_ref(mesh, bath)
        u_a, e_a = ref.advance(uv, eta, 1.5, 4)
        u_b, e_b, seconds = ref.advance_blocked(uv, eta, 1.5, 4)
        assert seconds >= 0.0 and np.array_equal(u_a, u_b) and np.array_equal(e_a, e_b)
