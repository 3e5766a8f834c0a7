"""CPU: tracer advection + vertex limiter of the oracle, pinned by the reference's own criteria
(test/slopelimiter/test_slopelimiter.py:50-57, test/tracerEq/test_consistency_2d.py:98-128)."""
import numpy as np
import pytest

from helpers import channel_case, make_oracle, make_ref, rel_linf
from thetis_amd.mesh import PeriodicRectangleMesh, UnitSquareMesh


def _tracer(ref, mesh, **kw):
    from oracle.ref_lib import RefTracer
    return RefTracer(ref, cell_topo_vertices=mesh.topo_vertex[mesh.cells], **kw)


@pytest.mark.parametrize('direction', ['x', 'y'])
def test_limiter_keeps_linear_field(direction):
    """type == 'linear': a linear field in x|y is not altered (l2_err < 1e-12); 'xy' is an expected failure upstream."""
    from oracle.swe2d_oracle import SWEOracle
    m = UnitSquareMesh(5, 5)
    orc = SWEOracle(m.vertex_xy, m.cells, np.ones(m.num_vertices))
    f = m.cell_xy()[:, :, 0 if direction == 'x' else 1].copy()
    assert orc.l2_norm(orc.limit(f) - f) < 1e-12


@pytest.mark.parametrize('direction', ['x', 'y'])
def test_limiter_jump_conserves_mass_and_removes_overshoots(direction):
    from oracle.swe2d_oracle import SWEOracle
    m = UnitSquareMesh(5, 5)
    orc = SWEOracle(m.vertex_xy, m.cells, np.ones(m.num_vertices))
    ax = 0 if direction == 'x' else 1
    jump = orc.project(lambda x, y: 0.5 + 0.5*np.tanh(20*((x, y)[ax] - 0.5)))
    lim = orc.limit(jump)
    mass = lambda t: float(np.sum(orc.area[:, None]/3.0*t))
    assert abs(mass(lim) - mass(jump)) < 1e-12
    assert jump.min() < -1e-3 and lim.min() > -2e-5


def test_limiter_c_restatement_equals_numpy(ref_so):
    mesh, bath, uv, eta = channel_case(seed=4)
    T = np.random.default_rng(0).normal(size=(mesh.num_cells, 3))
    assert np.array_equal(make_oracle(mesh, bath).limit(T), _tracer(make_ref(mesh, bath), mesh).limit(T))
    # periodic mesh: vertices on the two sides of the seam are the same mesh vertex
    pm = PeriodicRectangleMesh(6, 4, 3.0, 2.0, direction='x')
    from oracle.ref_lib import RefSWE
    from oracle.swe2d_oracle import SWEOracle
    orc = SWEOracle(pm.vertex_xy, pm.cells, np.ones(pm.num_vertices), topo_vertex=pm.topo_vertex)
    ref = RefSWE(pm.cell_xy(), pm.cell_nbr, pm.cell_nbr_facet, np.ones((pm.num_cells, 3)))
    T = np.random.default_rng(1).normal(size=(pm.num_cells, 3))
    topo = pm.topo_vertex[pm.cells]
    assert np.array_equal(orc.limit(T, topo_cells=topo), _tracer(ref, pm).limit(T))


def test_constant_tracer_stays_constant():
    """test_const_tracer: R_T(const) = 0 for any (even divergent, discontinuous) velocity field."""
    mesh, bath, uv, eta = channel_case(seed=5)
    orc = make_oracle(mesh, bath)
    T = np.full((mesh.num_cells, 3), 4.5)
    r = orc.tracer_residual(T, uv, eta)
    scale = np.abs(orc.tracer_residual(np.random.default_rng(2).normal(size=T.shape), uv, eta)).max()
    assert np.abs(r).max() < 1e-13*scale
    T1 = orc.tracer_ssprk33_step(T, uv, eta, 3.0)
    assert np.abs(T1 - 4.5).max() < 1e-12


@pytest.mark.parametrize('case', ['default', 'lf', 'value_bc'])
def test_tracer_numpy_and_c_restatements_agree(ref_so, case):
    mesh, bath, uv, eta = channel_case(seed=6)
    rng = np.random.default_rng(3)
    T = rng.normal(size=(mesh.num_cells, 3))
    src = 1e-3*rng.normal(size=T.shape)
    kw_np, kw_c = {
        'default': ({}, {}),
        'lf': (dict(use_lax_friedrichs_tracer=True, lax_friedrichs_tracer_scaling_factor=0.7,
                    tracer_advective_velocity_factor=0.9, source=src),)*2,
        'value_bc': (dict(bnd_conditions={1: {'value': 2.0}, 3: {'value': -1.0}}), dict(bnd_values={1: 2.0, 3: -1.0})),
    }[case]
    orc, ref = make_oracle(mesh, bath), make_ref(mesh, bath)
    rt = _tracer(ref, mesh, **kw_c)
    assert rel_linf(rt.tendency(T, uv, 3.0), orc.tracer_tendency(T, uv, eta, 3.0, **kw_np)) < 1e-13
    assert rel_linf(rt.step(T, uv, 3.0), orc.tracer_ssprk33_step(T, uv, eta, 3.0, **kw_np)) < 1e-13


def test_conservative_form_conserves_the_depth_integrated_tracer_on_a_periodic_mesh():
    """ConservativeHorizontalAdvectionTerm (tracer_eq_2d.py:341-395): with test function 1 the cell term vanishes and the
    upwind fluxes cancel pairwise, so int q dx is constant to round-off when there is no boundary."""
    from thetis_amd.mesh import PeriodicRectangleMesh
    mesh = PeriodicRectangleMesh(7, 5, 70.0, 40.0, direction='both')
    rng = np.random.default_rng(4)
    bath = 10.0 + rng.uniform(size=mesh.num_vertices)
    orc = make_oracle(mesh, bath)
    n = mesh.num_cells
    uv = rng.normal(size=(n, 3, 2))
    eta = 0.1*rng.normal(size=(n, 3))
    q = 5.0 + rng.normal(size=(n, 3))
    for lf in (False, True):
        r = orc.tracer_residual(q, uv, eta, conservative=True, use_lax_friedrichs_tracer=lf)
        assert abs(r.sum()) < 1e-12*np.abs(r).sum()
    # the non-conservative form does not have this property for a divergent velocity field
    r = orc.tracer_residual(q, uv, eta)
    assert abs(r.sum()) > 1e-6*np.abs(r).sum()


def test_conservative_and_nonconservative_forms_agree_for_divergence_free_constant_velocity():
    """u = const: div u = 0 and both sides carry the same velocity, so the two weak forms coincide."""
    mesh, bath, _, eta = channel_case(seed=3)
    orc = make_oracle(mesh, bath)
    n = mesh.num_cells
    uv = np.broadcast_to(np.array([0.7, -0.3]), (n, 3, 2)).copy()
    T = np.random.default_rng(1).normal(size=(n, 3))
    r0 = orc.tracer_residual(T, uv, eta, bnd_conditions={m: {'value': 1.0} for m in (1, 2, 3, 4)})
    r1 = orc.tracer_residual(T, uv, eta, bnd_conditions={m: {'value': 1.0} for m in (1, 2, 3, 4)}, conservative=True)
    assert np.abs(r0 - r1).max() < 1e-12*np.abs(r0).max()
