"""
GPU parity tests proper: the HIP path, called through the C ABI, against the oracle on the same seeded inputs.

Tolerance (FP64): rel. Linf <= 1e-12 per tendency evaluation, <= 1e-10 after 100 steps.
The device kernel uses FMA contraction and the algebraically equal sqrt(g*H) for g*sqrt(H/g) and H*sqrt(g/H), so
bitwise identity with the CPU is not expected; 1e-12 leaves ~4 decimal digits of slack over observed round-off.
"""
import numpy as np
import pytest

from helpers import channel_case, make_oracle, make_ref, rel_linf

pytestmark = pytest.mark.gpu

TOL_RHS = 1e-12
TOL_100 = 1e-10


def _device(mesh, bath, dt, **kw):
    from thetis_amd.device import Swe2dDevice
    return Swe2dDevice(mesh, bath, dt, **kw)


@pytest.mark.parametrize('opts', [
    dict(),
    dict(use_nonlinear_equations=False),
    dict(use_lax_friedrichs_velocity=False),
    dict(lax_friedrichs_velocity_scaling_factor=0.5),
])
def test_tendency_matches_numpy_oracle(hip_lib, opts):
    mesh, bath, uv, eta = channel_case()
    dt = 3.0
    orc = make_oracle(mesh, bath, **opts)
    ku_o, ke_o = orc.tendency(uv, eta, dt)
    dev = _device(mesh, bath, dt, **opts)
    dev.set_state(uv, eta)
    ku, ke = dev.tendency()
    assert rel_linf(ku, ku_o) < TOL_RHS
    assert rel_linf(ke, ke_o) < TOL_RHS
    # set/get round trip is exact
    uv2, eta2 = dev.get_state()
    assert np.array_equal(uv2, uv) and np.array_equal(eta2, eta)
    dev.close()


def test_step_matches_numpy_oracle(hip_lib):
    mesh, bath, uv, eta = channel_case(amp_eta=0.3, amp_u=0.2)
    dt = 2.0
    orc = make_oracle(mesh, bath)
    uo, eo = uv, eta
    dev = _device(mesh, bath, dt)
    dev.set_state(uv, eta)
    for _ in range(3):
        uo, eo = orc.ssprk33_step(uo, eo, dt)
    dev.advance(3)
    ud, ed = dev.get_state()
    assert rel_linf(ud, uo) < 10*TOL_RHS
    assert rel_linf(ed, eo) < 10*TOL_RHS
    # stage-by-stage entry point gives the same result as advance()
    dev.set_state(uv, eta)
    for _ in range(3):
        for s in range(3):
            dev.solve_stage(s)
    ud2, ed2 = dev.get_state()
    assert np.array_equal(ud, ud2) and np.array_equal(ed, ed2)
    dev.close()


def test_source_terms_match_oracle(hip_lib):
    from thetis_amd import _lib
    mesh, bath, uv, eta = channel_case(seed=3)
    x, y = mesh.vertex_xy.T
    rng = np.random.default_rng(5)
    n = mesh.num_cells
    cor = 1e-4*(1 + y/30e3)
    patm = 1e5 + 300*np.sin(x/2e4)
    msrc = 1e-3*rng.normal(size=(n, 3, 2))
    vsrc = 1e-3*rng.normal(size=(n, 3))
    dt = 3.0
    cases = [
        dict(coriolis=cor, linear_drag_coefficient=1e-3, atmospheric_pressure=patm, momentum_source=msrc, volume_source=vsrc),
        dict(manning_drag_coefficient=0.02),
        dict(quadratic_drag_coefficient=0.0025, norm_smoother=0.1),
        dict(nikuradse_bed_roughness=0.05, norm_smoother=0.01),
        dict(nikuradse_bed_roughness=18.0),             # k_s above part of the depth range: C_D = 0 there (:696-697)
        # spatially varying coefficients (Functions in the reference): per-vertex P1 fields
        dict(manning_drag_coefficient=0.02*(1 + x/100e3), linear_drag_coefficient=1e-3*(1 + y/30e3)),
        dict(quadratic_drag_coefficient=0.0025*(1 + y/30e3), norm_smoother=0.1),
        dict(nikuradse_bed_roughness=0.05*(1 + x/100e3)),
        dict(wind_stress=0.1*rng.normal(size=(n, 3, 2)), bnd_conditions={3: {'drag': 0.0025}, 1: {'drag': 0.01, 'elev': 0.1}}),
    ]
    for kw in cases:
        orc = make_oracle(mesh, bath, **kw)
        ku_o, ke_o = orc.tendency(uv, np.abs(eta), dt)
        dev = _device(mesh, bath, dt)
        if 'coriolis' in kw:
            dev.set_field(_lib.FIELD_CORIOLIS, cor[mesh.cells])
            dev.set_field(_lib.FIELD_ATMOSPHERIC_PRESSURE, patm[mesh.cells])
            dev.set_field(_lib.FIELD_MOMENTUM_SOURCE, msrc)
            dev.set_field(_lib.FIELD_VOLUME_SOURCE, vsrc)
            dev.set_scalar(_lib.SCALAR_LINEAR_DRAG, 1e-3)
        is_field = lambda v: isinstance(v, np.ndarray)
        for key, sid, fid in (('manning_drag_coefficient', _lib.SCALAR_MANNING_DRAG, _lib.FIELD_MANNING_DRAG),
                              ('quadratic_drag_coefficient', _lib.SCALAR_QUADRATIC_DRAG, _lib.FIELD_QUADRATIC_DRAG),
                              ('nikuradse_bed_roughness', _lib.SCALAR_NIKURADSE, _lib.FIELD_NIKURADSE),
                              ('linear_drag_coefficient', _lib.SCALAR_LINEAR_DRAG, _lib.FIELD_LINEAR_DRAG)):
            if key in kw and 'coriolis' not in kw:
                if is_field(kw[key]):
                    dev.set_field(fid, kw[key][mesh.cells])
                else:
                    dev.set_scalar(sid, kw[key])
        if 'norm_smoother' in kw:
            dev.set_scalar(_lib.SCALAR_NORM_SMOOTHER, kw['norm_smoother'])
        if 'wind_stress' in kw:
            dev.set_field(_lib.FIELD_WIND_STRESS, kw['wind_stress'])
            for marker, funcs in kw['bnd_conditions'].items():
                dev.set_bc(marker, funcs)
        dev.set_state(uv, np.abs(eta))
        ku, ke = dev.tendency()
        assert rel_linf(ku, ku_o) < TOL_RHS, kw.keys()
        assert rel_linf(ke, ke_o) < TOL_RHS, kw.keys()
        dev.close()


@pytest.mark.parametrize('bcs', [
    {1: {'elev': 0.3}, 2: {'un': 0.2}, 3: {'flux': 1e4, 'elev': 0.1}},
    {4: {'uv': (0.1, -0.2)}, 1: {'flux': -3e3}, 2: {'elev': 0.1, 'uv': (0.3, 0.1)}, 3: {'elev': -0.1, 'un': 0.05}},
])
def test_open_boundaries_match_oracle(hip_lib, bcs):
    mesh, bath, uv, eta = channel_case(seed=11)
    dt = 3.0
    orc = make_oracle(mesh, bath, bnd_conditions=bcs)
    ku_o, ke_o = orc.tendency(uv, eta, dt)
    dev = _device(mesh, bath, dt)
    for marker, funcs in bcs.items():
        dev.set_bc(marker, funcs)
    dev.set_state(uv, eta)
    ku, ke = dev.tendency()
    assert rel_linf(ku, ku_o) < TOL_RHS
    assert rel_linf(ke, ke_o) < TOL_RHS
    dev.close()


def test_100_steps_vs_c_restatement_medium_mesh(hip_lib, ref_so):
    """40k triangles, 100 SSPRK33 steps, against the C restatement (itself pinned to the numpy oracle)."""
    mesh, bath, _, _ = channel_case(nx=200, ny=100, lx=100e3, ly=50e3)
    n = mesh.num_cells
    rng = np.random.default_rng(1234)
    cx, cy = mesh.cell_xy()[:, :, 0], mesh.cell_xy()[:, :, 1]
    eta = 0.5*np.exp(-((cx - 50e3)**2 + (cy - 25e3)**2)/(5e3)**2) + 1e-3*rng.uniform(-1, 1, size=(n, 3))
    uv = 1e-3*rng.uniform(-1, 1, size=(n, 3, 2))
    dt = 1.0
    ref = make_ref(mesh, bath)
    dev = _device(mesh, bath, dt)
    dev.set_state(uv, eta)
    ku, ke = dev.tendency()
    ku_r, ke_r = ref.tendency(uv, eta, dt)
    assert rel_linf(ku, ku_r) < TOL_RHS and rel_linf(ke, ke_r) < TOL_RHS
    dev.advance(100)
    ud, ed = dev.get_state()
    ur, er = ref.advance(uv, eta, dt, 100)
    assert rel_linf(ud, ur) < TOL_100
    assert rel_linf(ed, er) < TOL_100
    # closed domain: volume is conserved to round-off (test/barotropicChannel/test_closed_channel.py:77-78 bar: 1e-12)
    d1 = dev.diagnostics()
    dev.set_state(uv, eta)
    d0 = dev.diagnostics()
    assert abs(d1[2] - d0[2])/d0[2] < 1e-12
    dev.close()


def test_full_size_properties(hip_lib):
    """BASELINE cfg 2 size (1M triangles): size-independent properties - lake at rest, volume conservation."""
    from thetis_amd.mesh import RectangleMesh
    mesh = RectangleMesh(1000, 500, 100e3, 50e3)
    x, y = mesh.vertex_xy.T
    bath = 20.0 - 10.0*x/100e3 + 2.0*np.sin(y/5000.0)
    n = mesh.num_cells
    dt = 0.25
    dev = _device(mesh, bath, dt)
    # lake at rest over variable bathymetry: tendency is zero to round-off
    dev.set_state(np.zeros((n, 3, 2)), np.full((n, 3), 0.37))
    ku, ke = dev.tendency()
    assert np.abs(ku).max() < 1e-13*9.81*20 and np.abs(ke).max() == 0.0
    rng = np.random.default_rng(1234)
    cx, cy = mesh.cell_xy()[:, :, 0], mesh.cell_xy()[:, :, 1]
    eta = 0.5*np.exp(-((cx - 50e3)**2 + (cy - 25e3)**2)/(5e3)**2) + 1e-3*rng.uniform(-1, 1, size=(n, 3))
    uv = 1e-3*rng.uniform(-1, 1, size=(n, 3, 2))
    dev.set_state(uv, eta)
    d0 = dev.diagnostics()
    dev.advance(20)
    d1 = dev.diagnostics()
    assert abs(d1[2] - d0[2])/d0[2] < 1e-12
    assert np.isfinite(d1).all()
    dev.close()


@pytest.mark.parametrize('cells', ['triangles', 'quadrilaterals'])
@pytest.mark.parametrize('wd', [False, True])
def test_lake_at_rest_with_manning_friction_has_no_tendency(hip_lib, cells, wd):
    """|u| = 0 exactly at every quadrature point of the Manning term (shallowwater_eq.py:685-700): the square root of the velocity
    magnitude is taken through a clamped reciprocal square root, H^(-1/3) through a float seed - the drag must still vanish
    identically and the lake stay at rest (no NaN from 0 * rsq(0)), with and without the displaced depth of wetting-drying."""
    from helpers import quad_case
    from thetis_amd import _lib
    from thetis_amd.device import Swe2dDevice
    if cells == 'quadrilaterals':
        mesh, bath = quad_case(nx=24, ny=11, seed=3, skew=0.2)[:2]
    else:
        mesh, bath = channel_case(nx=31, ny=13, seed=3)[:2]
    n, k = mesh.cells.shape
    dev = Swe2dDevice(mesh, bath, 2.0, boundary_len=mesh.boundary_len)
    dev.set_scalar(_lib.SCALAR_MANNING_DRAG, 0.02)
    if wd:
        dev.set_wetting_and_drying(0.4)
    dev.set_state(np.zeros((n, k, 2)), np.full((n, k), 0.25))
    ku, ke = dev.tendency()
    assert np.isfinite(ku).all() and np.isfinite(ke).all()
    assert np.abs(ku).max() < 1e-13*9.81*20 and np.abs(ke).max() == 0.0
    dev.advance(3)
    uv, eta = dev.get_state()
    assert np.abs(uv).max() < 1e-12 and np.abs(eta - 0.25).max() < 1e-12
    dev.close()


@pytest.mark.parametrize('quad', [False, True])
def test_function_valued_boundary_data(hip_lib, quad):
    """Spatially varying external elevation / velocity / normal velocity (Function-valued bnd_functions entries)."""
    from helpers import make_oracle_generic, quad_case
    from thetis_amd.device import Swe2dDevice
    if quad:
        mesh, bath, uv, eta = quad_case(seed=5)
    else:
        mesh, bath, uv, eta = channel_case(seed=5)
    n, k = mesh.cells.shape
    cxy = mesh.cell_xy()
    elev_f = 0.3*np.sin(cxy[:, :, 1]/4000.0)                                   # varies along the boundary x = 0
    un_f = 0.1*np.cos(cxy[:, :, 1]/6000.0)
    uv_f = np.stack([0.2*np.sin(cxy[:, :, 0]/2e4), -0.1*np.cos(cxy[:, :, 0]/3e4)], axis=2)
    bcs = {1: {'elev': elev_f}, 2: {'un': un_f, 'elev': 0.1}, 3: {'uv': uv_f}, 4: {'elev': elev_f, 'uv': uv_f}}
    dt = 3.0
    orc = make_oracle_generic(mesh, bath, bnd_conditions=bcs)
    dev = Swe2dDevice(mesh, bath, dt, boundary_len=mesh.boundary_len)
    for m, funcs in bcs.items():
        dev.set_bc(m, funcs)
    dev.set_state(uv, eta)
    ku, ke = dev.tendency()
    ku_o, ke_o = orc.tendency(uv, eta, dt)
    assert rel_linf(ku, ku_o) < TOL_RHS and rel_linf(ke, ke_o) < TOL_RHS
    dev.close()


@pytest.mark.parametrize('cells', ['triangles', 'quadrilaterals'])
def test_forward_euler_steps_match_oracle(hip_lib, cells):
    """timeintegrator.ForwardEuler (thetis/timeintegrator.py:115-165), the other explicit stepper of solver2d.py:662-672."""
    from helpers import make_oracle_generic, quad_case
    if cells == 'triangles':
        mesh, bath, uv, eta = channel_case(seed=31)
        orc = make_oracle(mesh, bath, bnd_conditions={2: {'elev': 0.1}})
    else:
        mesh, bath, uv, eta = quad_case(seed=31)
        orc = make_oracle_generic(mesh, bath, bnd_conditions={2: {'elev': 0.1}})
    dt = 1.5
    dev = _device(mesh, bath, dt, boundary_len=mesh.boundary_len)
    dev.set_bc(2, {'elev': 0.1})
    dev.set_state(uv, eta)
    dev.advance_forward_euler(3)
    u, e = dev.get_state()
    uo, eo = uv, eta
    for _ in range(3):
        uo, eo = orc.forward_euler_step(uo, eo, dt)
    assert rel_linf(u, uo) < TOL_RHS and rel_linf(e, eo) < TOL_RHS
    # mixing steppers on one handle keeps working (buffer swap is transparent)
    dev.advance(1)
    u, e = dev.get_state()
    uo, eo = orc.ssprk33_step(uo, eo, dt)
    assert rel_linf(u, uo) < TOL_RHS and rel_linf(e, eo) < TOL_RHS
    dev.close()


def test_per_marker_boundary_fields_with_corner_cells(hip_lib):
    """Different Functions of the same kind on different markers (the MMS scenarios prescribe -flux_x, +flux_x, -flux_y):
    stored per facet, so the corner cells (two boundary facets sharing a node) keep both."""
    mesh, bath, uv, eta = channel_case(nx=6, ny=4, seed=41)
    rng = np.random.default_rng(9)
    n = mesh.num_cells
    f = lambda s=1.0: s*rng.normal(size=(n, 3))
    bcs = {1: {'elev': f(0.1), 'flux': f(2e4)}, 2: {'flux': f(2e4)}, 3: {'elev': f(0.1), 'un': f(0.3)},
           4: {'uv': 0.3*rng.normal(size=(n, 3, 2))}}
    corner = np.sum(mesh.cell_nbr < 0, axis=1) == 2
    assert corner.any()
    dt = 2.0
    orc = make_oracle(mesh, bath, bnd_conditions=bcs, horizontal_viscosity=30.0)
    dev = _device(mesh, bath, dt)
    for marker, funcs in bcs.items():
        dev.set_bc(marker, funcs)
    dev.set_viscosity(30.0)
    dev.set_state(uv, eta)
    ku, ke = dev.tendency()
    ku_o, ke_o = orc.tendency(uv, eta, dt)
    assert rel_linf(ku, ku_o) < TOL_RHS and rel_linf(ke, ke_o) < TOL_RHS
    assert rel_linf(ku[corner], ku_o[corner]) < TOL_RHS
    dev.close()


def test_edge_cases_of_the_c_abi(hip_lib):
    """Smallest meshes (2 cells; 1 cell with walls all round), zero steps, ragged block tails (n not a multiple of 256 or
    of the 8-block XCD grid), invalid arguments returning error codes instead of crashing."""
    from thetis_amd import _lib
    from thetis_amd.mesh import Mesh2d, RectangleMesh
    # one triangle, three wall facets
    xy = np.array([[0.0, 0.0], [100.0, 0.0], [0.0, 80.0]])
    mesh = Mesh2d(xy, np.array([[0, 1, 2]]))
    bath = np.full(3, 5.0)
    uv = np.array([[[0.1, -0.2], [0.05, 0.1], [-0.1, 0.0]]])
    eta = np.array([[0.01, -0.02, 0.03]])
    from oracle.swe2d_oracle import SWEOracle
    orc = SWEOracle(mesh.vertex_xy, mesh.cells, bath)        # every exterior facet: marker 1, closed
    dev = _device(mesh, bath, 0.5)
    dev.set_state(uv, eta)
    dev.advance(0)                                           # no-op
    u0, e0 = dev.get_state()
    assert np.array_equal(u0, uv) and np.array_equal(e0, eta)
    dev.advance(2)
    u, e = dev.get_state()
    uo, eo = uv, eta
    for _ in range(2):
        uo, eo = orc.ssprk33_step(uo, eo, 0.5)
    assert rel_linf(u, uo) < TOL_RHS and rel_linf(e, eo) < TOL_RHS
    # invalid arguments: error codes + message, handle stays usable
    assert hip_lib.swe2d_solve_stage(dev.h, 3) != 0
    assert b'i_stage' in hip_lib.swe2d_last_error(dev.h)
    assert hip_lib.swe2d_advance(dev.h, -1) != 0
    assert hip_lib.swe2d_set_dt(dev.h, -1.0) != 0
    assert hip_lib.swe2d_set_bc(dev.h, 99, 1, None) != 0
    assert hip_lib.swe2d_tracer_solve_stage(dev.h, 0, 0) != 0          # no such tracer
    dev.advance(1)
    dev.close()
    # ragged sizes: 2, 254, 258 and 2050 cells (block tail, XCD-grid padding)
    for nx, ny in ((1, 1), (127, 1), (129, 1), (41, 25)):
        mesh = RectangleMesh(nx, ny, 1000.0*nx, 900.0*ny)
        rng = np.random.default_rng(nx)
        bath = 10.0 + rng.uniform(size=mesh.num_vertices)
        uv = 0.1*rng.normal(size=(mesh.num_cells, 3, 2))
        eta = 0.05*rng.normal(size=(mesh.num_cells, 3))
        dev = _device(mesh, bath, 1.0)
        dev.set_state(uv, eta)
        ku, ke = dev.tendency()
        ku_o, ke_o = make_oracle(mesh, bath).tendency(uv, eta, 1.0)
        assert rel_linf(ku, ku_o) < TOL_RHS and rel_linf(ke, ke_o) < TOL_RHS
        dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize('reorder', ['auto', None])
def test_compact_boundary_facet_upload_equals_nodal_field_upload(hip_lib, reorder):
    """swe2d_set_bc_facets / swe2d_tracer_set_bc_facets (what update_forcings costs per call: the boundary facets' values)
    against swe2d_set_bc_field / swe2d_tracer_set_bc_field (the whole nodal field): identical tendencies."""
    from thetis_amd import _lib
    from thetis_amd.device import FacetValues, Swe2dDevice
    mesh, bath, uv, eta = channel_case(nx=9, ny=5, seed=5)
    rng = np.random.default_rng(8)
    n, k = mesh.num_cells, 3
    f_elev, f_uv, f_T = 0.2*rng.normal(size=(n, k)), 0.2*rng.normal(size=(n, k, 2)), rng.normal(size=(n, k))
    res = {}
    for mode in ('nodal', 'compact'):
        dev = Swe2dDevice(mesh, bath, 2.0, reorder=reorder)
        tid = dev.add_tracer()
        dev.tracer_set_diffusivity(tid, 30.0, 1.0)
        if mode == 'nodal':
            dev.set_bc(1, {'elev': f_elev})
            dev.set_bc(2, {'uv': f_uv})
            dev.tracer_set_bc(tid, 2, f_T)
        else:
            dev.set_bc(1, {'elev': dev.facet_node_values(1, f_elev)})
            dev.set_bc(2, {'uv': dev.facet_node_values(2, f_uv)})
            assert isinstance(dev.facet_node_values(2, f_uv), FacetValues)
            cells, _ = dev.boundary_facets(dev._slot(2))
            dev.tracer_set_bc_facets(tid, dev._slot(2), f_T[cells])
        dev.tracer_set_diffusion_bc(tid, 2, 4, 0.0)
        dev.set_state(uv, eta)
        dev.tracer_set_state(tid, 3.0 + rng.normal(size=(n, k))*0 + np.arange(n)[:, None]*1e-3)
        res[mode] = (dev.tendency(), dev.tracer_tendency(tid))
        dev.close()
    (ku_a, ke_a), kt_a = res['nodal']
    (ku_b, ke_b), kt_b = res['compact']
    assert np.array_equal(ku_a, ku_b) and np.array_equal(ke_a, ke_b) and np.array_equal(kt_a, kt_b)
    # the boundary values matter: a different field gives a different tendency
    assert np.abs(ku_a).max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize('cells', ['tri', 'quad'])
def test_vertex_field_upload_equals_nodal_upload(hip_lib, cells):
    """swe2d_set_field_vertex (continuous P1 coefficient by vertex values, injected into the DG nodes on the device)
    against swe2d_set_field with the host-side injection."""
    from thetis_amd import _lib
    from thetis_amd.device import Swe2dDevice
    if cells == 'quad':
        from helpers import quad_case
        mesh, bath, uv, eta = quad_case(nx=7, ny=5, skew=0.2, seed=2)
    else:
        mesh, bath, uv, eta = channel_case(nx=9, ny=5, seed=5)
    rng = np.random.default_rng(9)
    f_cor = 1e-4*(1.0 + rng.uniform(size=mesh.num_vertices))
    tau = 0.1*rng.normal(size=(mesh.num_vertices, 2))
    res = []
    for mode in ('nodal', 'vertex'):
        dev = Swe2dDevice(mesh, bath, 1.0)
        if mode == 'nodal':
            dev.set_field(_lib.FIELD_CORIOLIS, f_cor[mesh.cells])
            dev.set_field(_lib.FIELD_WIND_STRESS, tau[mesh.cells])
        else:
            dev.set_field_vertex(_lib.FIELD_CORIOLIS, f_cor)
            dev.set_field_vertex(_lib.FIELD_WIND_STRESS, tau)
        dev.set_state(uv, eta)
        res.append(dev.tendency())
        dev.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize('setup', ['walls', 'open', 'fields+sources+drag', 'wetting-drying'])
def test_boundary_inline_variant_gives_the_bits_of_the_epilogue_variant(hip_lib, setup):
    """Small launches (<= two waves per SIMD) evaluate boundary fluxes inside the facet loop (template parameter BINL), large
    ones after the cell's outputs are finished from reloaded values: a partition and the whole mesh take different variants,
    so their results must not differ in a single bit (tests/test_distributed.py compares them bitwise)."""
    import os
    from thetis_amd import _lib
    from thetis_amd.device import Swe2dDevice
    mesh, bath, uv, eta = channel_case(nx=24, ny=9, seed=17, amp_eta=0.3, amp_u=0.2)      # 432 cells: 6.75 blocks
    cxy = mesh.cell_xy()
    out = {}
    for force in ('0', '1', '1+ldsx'):            # epilogue variant, boundary-inline variant, the same with the LDS exchange
        os.environ['THETIS_AMD_BND_INLINE'] = force[0]
        os.environ['THETIS_AMD_LDSX'] = '1' if force.endswith('ldsx') else '0'
        try:
            dev = Swe2dDevice(mesh, bath, 2.0, boundary_len=mesh.boundary_len)
            if setup == 'open':
                for m, funcs in {1: {'elev': 0.3}, 2: {'un': 0.2}, 3: {'flux': 1e4, 'elev': 0.1}, 4: {'uv': (0.1, -0.2)}}.items():
                    dev.set_bc(m, funcs)
            elif setup == 'fields+sources+drag':
                elev_f = 0.3*np.sin(cxy[:, :, 1]/4000.0)
                uv_f = np.stack([0.2*np.sin(cxy[:, :, 0]/2e4), -0.1*np.cos(cxy[:, :, 0]/3e4)], axis=2)
                for m, funcs in {1: {'elev': elev_f}, 2: {'un': 0.1*np.cos(cxy[:, :, 1]/6000.0), 'elev': 0.1},
                                 3: {'uv': uv_f, 'drag': 0.01}, 4: {'flux': 0.5e4*np.ones_like(elev_f)}}.items():
                    dev.set_bc(m, funcs)
                dev.set_scalar(_lib.SCALAR_MANNING_DRAG, 0.02)
                dev.set_field(_lib.FIELD_CORIOLIS, 1e-4*np.ones_like(eta))
            elif setup == 'wetting-drying':
                dev.set_wetting_and_drying(0.5)
                dev.set_bc(2, {'elev': 0.2})
            dev.set_state(uv, eta)
            dev.advance(3)
            out[force] = dev.get_state()
            dev.close()
        finally:
            os.environ.pop('THETIS_AMD_BND_INLINE', None)
            os.environ.pop('THETIS_AMD_LDSX', None)
    for force in ('1', '1+ldsx'):
        assert np.array_equal(out['0'][0], out[force][0]) and np.array_equal(out['0'][1], out[force][1]), force



def test_stage_kernel_variants_agree_bitwise_on_a_large_launch(hip_lib):
    """The host picks the stage-kernel variant by launch size (LDS exchange once the state exceeds the Infinity Cache, 1.24 M triangles), so a partition and the
    whole mesh of the bench run different variants: 200 k cells, 5 steps, every variant forced in turn."""
    import os
    from thetis_amd.device import Swe2dDevice
    mesh, bath, uv, eta = channel_case(nx=400, ny=250, lx=100e3, ly=50e3, seed=5, amp_eta=0.3, amp_u=0.2)
    out = {}
    for binl, ldsx in (('0', '0'), ('1', '0'), ('1', '1')):
        os.environ['THETIS_AMD_BND_INLINE'], os.environ['THETIS_AMD_LDSX'] = binl, ldsx
        try:
            dev = Swe2dDevice(mesh, bath, 0.5)
            dev.set_bc(2, {'elev': 0.1})
            dev.set_state(uv, eta)
            dev.advance(5)
            out[binl + ldsx] = dev.get_state()
            dev.close()
        finally:
            os.environ.pop('THETIS_AMD_BND_INLINE', None)
            os.environ.pop('THETIS_AMD_LDSX', None)
    assert np.isfinite(out['00'][1]).all()
    for key in ('10', '11'):
        assert np.array_equal(out['00'][0], out[key][0]) and np.array_equal(out['00'][1], out[key][1]), key


@pytest.mark.parametrize('numbering', ['device', 'device_wetting_drying', 'caller_random'])
def test_compact_connectivity_gives_the_bits_of_the_wide_records(hip_lib, monkeypatch, numbering):
    """The triangle kernels read the connectivity as 16-B records of differences (csrc/swe2d_conn.h; the wide
    24-B records where a difference does not fit).  Same bits as with THETIS_AMD_COMPACT_IDX=0: SWE stages with open boundaries,
    a tracer with its limiter; in the device's numbering (few escapes) and in a RANDOM numbering of cells and vertices that is
    kept as it is (a third of the cells of a 400 k-cell mesh escape, the rest decode differences of every size and sign); with
    wetting-drying and Manning friction (kernels that take the records only when forced: they are bound by their arithmetic)."""
    from thetis_amd.device import Swe2dDevice
    from thetis_amd import _lib
    from thetis_amd.mesh import Mesh2d, _rect_marker_fn
    monkeypatch.setenv('THETIS_AMD_FLOW', '0')
    if numbering.startswith('device'):
        mesh, bath, uv, eta = channel_case(nx=450, ny=300, lx=100e3, ly=50e3, seed=5, amp_eta=0.3, amp_u=0.2)      # 270 k cells
        reorder = 'auto'
    else:
        m0, bath0, uv0, eta0 = channel_case(nx=500, ny=400, lx=100e3, ly=80e3, seed=6, amp_eta=0.3, amp_u=0.2)
        rng = np.random.default_rng(11)
        cperm, vperm = rng.permutation(m0.num_cells), rng.permutation(m0.num_vertices)
        vinv = np.empty_like(vperm)
        vinv[vperm] = np.arange(len(vperm))
        mesh = Mesh2d(m0.vertex_xy[vperm], vinv[m0.cells[cperm]], marker_fn=_rect_marker_fn(100e3, 80e3))
        bath, uv, eta = bath0[vperm], uv0[cperm], eta0[cperm]
        reorder = None
    tr0 = np.where(mesh.cell_xy()[:, :, 0] < 40e3, 0.0, 30.0)
    out, info = [], []
    for compact in ('0', '2', '1'):                             # never / in every launch / where it pays (conn_pays: >= 250 k cells)
        monkeypatch.setenv('THETIS_AMD_COMPACT_IDX', compact)
        dev = Swe2dDevice(mesh, bath, 0.5, reorder=reorder)
        info.append(dev.connectivity_info())
        if numbering == 'device_wetting_drying':
            dev.set_wetting_and_drying(0.5)
            dev.set_scalar(_lib.SCALAR_MANNING_DRAG, 0.02)
        dev.set_bc(2, {'elev': 0.1})
        dev.set_state(0.1*uv, eta)
        tid = dev.add_tracer()
        dev.tracer_set_state(tid, tr0)
        dev.advance_coupled(3, tracer_only=False, use_limiter=True)
        dev.advance(2)
        out.append(dev.get_state() + (dev.tracer_get_state(tid),))
        dev.close()
    assert info[0] == (0, 0) and info[1][0] == 1 and info[2] == info[1]
    n = mesh.num_cells
    if numbering.startswith('device'):
        assert 0 <= info[1][1] < 0.02*n, info
    else:
        assert 0.2*n < info[1][1] < n, info
    assert np.isfinite(out[0][1]).all()
    for other in out[1:]:
        for a, b in zip(out[0], other):
            assert np.array_equal(a, b)


@pytest.mark.parametrize('case', ['structured', 'linear_no_lf', 'ragged_small', 'random_numbering', 'by_the_rule_270k', 'sources', 'sources_by_the_rule_270k',
                                  'coupled_by_the_rule_270k', 'tile_order'])
def test_fused_stage_pair_gives_the_bits_of_the_stage_launches(hip_lib, monkeypatch, case):
    """csrc/swe2d_fuse.h: stages 1 and 2 of a step in one launch by overlapped tiles (192 interior cells + their ring per 256-lane
    workgroup, U(1) never leaves the chip), stage 3 as a stage launch - what swe2d_advance takes from 250 k cells where the kernel
    covers the configuration.  Bit for bit THETIS_AMD_FUSE12=0 (three stage launches): open boundaries and walls, linear equations
    without the Lax-Friedrichs term, partial tiles on a small ragged mesh, a random numbering (tiles of a few cells with rings that
    fill the workgroup; forced), and a 270 k-cell mesh where the library decides by itself."""
    from thetis_amd.device import Swe2dDevice
    from thetis_amd.mesh import Mesh2d, _rect_marker_fn
    monkeypatch.setenv('THETIS_AMD_FLOW', '0')
    kw, reorder, forced = {}, 'auto', '1'
    if case == 'structured':
        mesh, bath, uv, eta = channel_case(nx=200, ny=120, lx=100e3, ly=50e3, seed=5, amp_eta=0.3, amp_u=0.2)
    elif case == 'linear_no_lf':
        mesh, bath, uv, eta = channel_case(nx=120, ny=90, lx=100e3, ly=50e3, seed=6, amp_eta=0.3, amp_u=0.2)
        kw = dict(use_nonlinear_equations=False, use_lax_friedrichs_velocity=False)
    elif case == 'ragged_small':
        mesh, bath, uv, eta = channel_case(nx=53, ny=31, seed=5)
    elif case == 'random_numbering':
        m0, bath0, uv0, eta0 = channel_case(nx=90, ny=70, lx=100e3, ly=80e3, seed=6, amp_eta=0.3, amp_u=0.2)
        rng = np.random.default_rng(11)
        cperm, vperm = rng.permutation(m0.num_cells), rng.permutation(m0.num_vertices)
        vinv = np.empty_like(vperm)
        vinv[vperm] = np.arange(len(vperm))
        mesh = Mesh2d(m0.vertex_xy[vperm], vinv[m0.cells[cperm]], marker_fn=_rect_marker_fn(100e3, 80e3))
        bath, uv, eta, reorder = bath0[vperm], uv0[cperm], eta0[cperm], None
    elif case in ('sources', 'tile_order'):
        mesh, bath, uv, eta = channel_case(nx=120, ny=90, lx=100e3, ly=50e3, seed=8, amp_eta=0.3, amp_u=0.2)
    else:
        mesh, bath, uv, eta = channel_case(nx=450, ny=300, lx=100e3, ly=50e3, seed=5, amp_eta=0.3, amp_u=0.2)
        forced = '2'              # the pair by the library's rule (250 k cells), never the three-stage launch the mesh would take by itself
    from thetis_amd import _lib
    cxy = mesh.cell_xy()
    out = []
    for fuse in ('0', forced):
        if fuse is None:
            monkeypatch.delenv('THETIS_AMD_FUSE12', raising=False)
        else:
            monkeypatch.setenv('THETIS_AMD_FUSE12', fuse)
        dev = Swe2dDevice(mesh, bath, 0.5, reorder=reorder, **kw)
        dev.set_bc(2, {'elev': 0.1})
        dev.set_bc(3, {'un': 0.05})
        if case.startswith('sources'):
            # round 6: the SRC instances - a Coriolis field, Manning friction, wind stress, a linear drag field, atmospheric pressure
            dev.set_field(_lib.FIELD_CORIOLIS, 1e-4*(1.0 + cxy[:, :, 1]/50e3))
            dev.set_scalar(_lib.SCALAR_MANNING_DRAG, 0.02)
            dev.set_field(_lib.FIELD_WIND_STRESS, np.stack([0.1*np.sin(cxy[:, :, 0]/2e4), 0.05*np.cos(cxy[:, :, 1]/1e4)], axis=2))
            dev.set_field(_lib.FIELD_LINEAR_DRAG, 1e-4*(1.0 + np.cos(cxy[:, :, 0]/3e4)))
            dev.set_field(_lib.FIELD_ATMOSPHERIC_PRESSURE, 1e5 + 200.0*np.sin(cxy[:, :, 0]/2.5e4))
        if case == 'tile_order' and fuse != '0':
            # tiles cut from another order than the numbering (what a partition passes): same bits
            dev.fused_set_order(np.random.default_rng(3).permutation(mesh.num_cells)[np.argsort(
                np.random.default_rng(3).permutation(mesh.num_cells)//5000, kind='stable')])
        dev.set_state(uv if not case.startswith('sources') else 0.1*uv, eta if not case.startswith('sources') else 0.1*np.abs(eta))
        if fuse != '0':
            assert dev.fused_pair_info()[0], case
        if case.startswith('coupled'):
            # round 6: swe2d_advance_coupled takes the fused pair for its shallow-water half (cfg 4 on triangles)
            tid = dev.add_tracer()
            dev.tracer_set_state(tid, np.where(cxy[:, :, 0] < 40e3, 0.0, 30.0))
            dev.advance_coupled(3, tracer_only=False, use_limiter=True)
            dev.advance_coupled(2, tracer_only=False, use_limiter=True)
            out.append(dev.get_state() + (dev.tracer_get_state(tid),))
        else:
            dev.advance(3)
            dev.advance(2)
            out.append(dev.get_state())
        dev.close()
    assert np.isfinite(out[0][1]).all()
    for a_, b_ in zip(out[0], out[1]):
        assert np.array_equal(a_, b_)


@pytest.mark.parametrize('case', ['structured', 'linear_no_lf', 'ragged_small', 'random_numbering', 'sources', 'patches_12x7', 'patches_5x3', 'by_the_rule_270k', 'coupled_by_the_rule_270k',
                                  'bench_size_vs_oracle'])
def test_fused_stage_triple_gives_the_bits_of_the_stage_launches(hip_lib, ref_so, monkeypatch, case):
    """csrc/swe2d_fuse.h, swe_fuse123_kernel (round 6; SWE2D_OPT_FUSED_STAGES = 3): ALL three stages of a step in one launch, tiles of
    interior + two rings in a 256-lane workgroup, U(1) and U(2) never leave the chip, U(3) into the second state buffer and the two
    swap.  Bit for bit the three stage launches (open boundaries and walls, linear equations without Lax-Friedrichs, partial tiles,
    a random numbering with tiny tiles, source terms; tiles cut as patches of 12 x 7 quads - what a RectangleMesh of >= 500 k triangles
    gets by itself - and of 5 x 3 (swe2d_fused_set_triple_tiles; a mesh whose sides are no multiples of the patch); odd and even numbers of
    steps: the buffers change places every step), and at 1 M cells 20 steps against oracle/swe2d_ref.c at 1e-11."""
    from thetis_amd import _lib
    from thetis_amd.device import Swe2dDevice
    from thetis_amd.mesh import Mesh2d, RectangleMesh, _rect_marker_fn
    monkeypatch.setenv('THETIS_AMD_FLOW', '0')
    kw, reorder = {}, 'auto'
    if case == 'bench_size_vs_oracle':
        mesh = RectangleMesh(1000, 500, 100e3, 50e3)
        bath = np.full(mesh.num_vertices, 20.0)
        uv, eta = _bump_state(mesh)
        monkeypatch.setenv('THETIS_AMD_FUSE12', '3')
        dev = Swe2dDevice(mesh, bath, 0.25)
        dev.set_state(uv, eta)
        dev.advance(20)
        ud, ed = dev.get_state()
        dev.close()
        ur, er = make_ref(mesh, bath).advance(uv, eta, 0.25, 20)
        assert rel_linf(ud, ur) < 1e-11 and rel_linf(ed, er) < 1e-11, (rel_linf(ud, ur), rel_linf(ed, er))
        return
    if case == 'structured':
        mesh, bath, uv, eta = channel_case(nx=200, ny=120, lx=100e3, ly=50e3, seed=5, amp_eta=0.3, amp_u=0.2)
    elif case.startswith('patches'):
        mesh, bath, uv, eta = channel_case(nx=131, ny=75, lx=100e3, ly=50e3, seed=9, amp_eta=0.3, amp_u=0.2)
    elif case.endswith('by_the_rule_270k'):
        mesh, bath, uv, eta = channel_case(nx=450, ny=300, lx=100e3, ly=50e3, seed=5, amp_eta=0.3, amp_u=0.2)
    elif case == 'linear_no_lf':
        mesh, bath, uv, eta = channel_case(nx=120, ny=90, lx=100e3, ly=50e3, seed=6, amp_eta=0.3, amp_u=0.2)
        kw = dict(use_nonlinear_equations=False, use_lax_friedrichs_velocity=False)
    elif case == 'ragged_small':
        mesh, bath, uv, eta = channel_case(nx=53, ny=31, seed=5)
    elif case == 'random_numbering':
        m0, bath0, uv0, eta0 = channel_case(nx=90, ny=70, lx=100e3, ly=80e3, seed=6, amp_eta=0.3, amp_u=0.2)
        rng = np.random.default_rng(11)
        cperm, vperm = rng.permutation(m0.num_cells), rng.permutation(m0.num_vertices)
        vinv = np.empty_like(vperm)
        vinv[vperm] = np.arange(len(vperm))
        mesh = Mesh2d(m0.vertex_xy[vperm], vinv[m0.cells[cperm]], marker_fn=_rect_marker_fn(100e3, 80e3))
        bath, uv, eta, reorder = bath0[vperm], uv0[cperm], eta0[cperm], None
    else:
        mesh, bath, uv, eta = channel_case(nx=120, ny=90, lx=100e3, ly=50e3, seed=8, amp_eta=0.3, amp_u=0.2)
        uv, eta = 0.1*uv, 0.1*np.abs(eta)
    cxy = mesh.cell_xy()
    out = []
    for fuse in ('0', '3'):
        monkeypatch.setenv('THETIS_AMD_FUSE12', fuse)
        if fuse == '3' and case.endswith('by_the_rule_270k'):
            monkeypatch.delenv('THETIS_AMD_FUSE12')          # what the library takes by itself (11 x 8-quad patches from Swe2dDevice)
        dev = Swe2dDevice(mesh, bath, 0.5, reorder=reorder, **kw)
        dev.set_bc(2, {'elev': 0.1})
        dev.set_bc(3, {'un': 0.05})
        if case == 'sources':
            dev.set_field(_lib.FIELD_CORIOLIS, 1e-4*(1.0 + cxy[:, :, 1]/50e3))
            dev.set_scalar(_lib.SCALAR_MANNING_DRAG, 0.02)
            dev.set_field(_lib.FIELD_WIND_STRESS, np.stack([0.1*np.sin(cxy[:, :, 0]/2e4), 0.05*np.cos(cxy[:, :, 1]/1e4)], axis=2))
        if case.startswith('patches') and fuse == '3':
            from thetis_amd import ordering
            bx, by = (12, 7) if case == 'patches_12x7' else (5, 3)
            order, starts = ordering.triple_tile_order(mesh, bx, by)
            dev.fused_set_triple_tiles(order, starts)
            on, tiles, ring1, ring2 = dev.fused_triple_info()
            assert on and tiles >= len(starts) and (tiles == len(starts) or case == 'patches_12x7'), (tiles, len(starts))
        dev.set_state(uv, eta)
        if fuse == '3':
            assert dev.fused_triple_info()[0], case
        if case.startswith('coupled'):
            # swe2d_advance_coupled: the shallow-water half in one launch, the tracer stages read the velocity from the buffer it ends in
            tid = dev.add_tracer()
            dev.tracer_set_state(tid, np.where(cxy[:, :, 0] < 40e3, 0.0, 30.0))
            dev.advance_coupled(3, tracer_only=False, use_limiter=True)
            dev.advance_coupled(2, tracer_only=False, use_limiter=True)
            out.append(dev.get_state() + (dev.tracer_get_state(tid),))
            dev.close()
            continue
        dev.advance(3)
        dev.advance(2)
        a = dev.get_state()
        dev.solve_stage(0); dev.solve_stage(1); dev.solve_stage(2)     # stage launches after an odd number of buffer swaps
        dev.advance(1)
        out.append(a + dev.get_state())
        dev.close()
    assert np.isfinite(out[0][1]).all()
    for a_, b_ in zip(out[0], out[1]):
        assert np.array_equal(a_, b_)


def test_step_launches_inside_a_stream_capture_must_come_in_pairs(hip_lib, monkeypatch):
    """swe2d_solve_step_cells leaves its result in the second state buffer and swaps the two on the host: a captured sequence with an
    EVEN number of them replays (twice here: four steps, the bits of four eager steps by stage launches); one with an odd number ends
    on the other buffer than it began on - the library says so at the next call outside the capture (SWE2D_ERR_UNSUPPORTED), it does not
    let a second replay read a stale buffer silently."""
    import torch
    from thetis_amd import _lib
    from thetis_amd.device import Swe2dDevice
    monkeypatch.setenv('THETIS_AMD_FLOW', '0')
    monkeypatch.setenv('THETIS_AMD_FUSE12', '0')
    mesh, bath, uv, eta = channel_case(nx=60, ny=40, seed=3, amp_eta=0.3, amp_u=0.2)
    dev = Swe2dDevice(mesh, bath, 0.5)
    dev.set_state(uv, eta)
    dev.advance(4)
    ref_state = dev.get_state()
    dev.set_option(_lib.OPT_FUSED_STAGES, 3)
    assert dev.fused_step_info()[0]                      # (builds the tile tables: never inside a capture)
    s = torch.cuda.Stream()
    dev.set_stream(s.cuda_stream)
    n = mesh.num_cells
    with torch.cuda.stream(s):
        dev.set_state(uv, eta)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode='thread_local'):
            dev.solve_step_cells(n)
            dev.solve_step_cells(n)
        g.replay()
        g.replay()
        s.synchronize()
        dev.synchronize()
        u, e = dev.get_state()
        assert np.array_equal(u, ref_state[0]) and np.array_equal(e, ref_state[1])
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, stream=s, capture_error_mode='thread_local'):
            dev.solve_step_cells(n)
        with pytest.raises(_lib.Swe2dError) as err:
            dev.synchronize()
        assert err.value.code == _lib.ERR_UNSUPPORTED and 'odd number' in str(err.value)
        dev.synchronize()                                # reported once
    dev.set_stream(None)
    dev.close()


def _bump_state(mesh, seed=1234):
    n = mesh.num_cells
    rng = np.random.default_rng(seed)
    cx, cy = mesh.cell_xy()[:, :, 0], mesh.cell_xy()[:, :, 1]
    eta = 0.5*np.exp(-((cx - 50e3)**2 + (cy - 25e3)**2)/(5e3)**2) + 1e-3*rng.uniform(-1, 1, size=(n, 3))
    uv = 1e-3*rng.uniform(-1, 1, size=(n, 3, 2))
    return uv, eta


@pytest.mark.parametrize('bcs', [
    {},
    {1: {'elev': 0.3}, 2: {'un': 0.2}, 3: {'flux': 1e4, 'elev': 0.1}},
    {4: {'uv': (0.1, -0.2)}, 1: {'flux': -3e3}, 2: {'elev': 0.1, 'uv': (0.3, 0.1)}, 3: {'elev': -0.1, 'un': 0.05}},
], ids=['walls', 'open_a', 'open_b'])
@pytest.mark.parametrize('kernel', ['pair', 'triple'])
def test_fused_stage_pair_matches_the_c_restatement(hip_lib, ref_so, monkeypatch, bcs, kernel):
    """The headline kernel against the ORACLE, not against the stage launches: 270 k triangles - the size from
    which swe2d_advance takes the fused stage pair by itself, asserted - 1, 10 and 100 SSPRK33 steps against oracle/swe2d_ref.c at
    1e-12 / 1e-11 / 1e-10, closed walls and both open-boundary sets of test_open_boundaries_match_oracle.  'triple': what that mesh takes
    by itself since the end of round 6 - all three stages in one launch on 11 x 8-quad patches, asserted; 'pair': SWE2D_OPT_FUSED_STAGES
    = 2, the fused pair + stage 3 (what a mesh with source terms, a partition or a stream capture takes)."""
    for v in ('THETIS_AMD_FUSE12', 'THETIS_AMD_FLOW', 'THETIS_AMD_BND_INLINE', 'THETIS_AMD_TRIPLE_TILE'):
        monkeypatch.delenv(v, raising=False)
    if kernel == 'pair':
        monkeypatch.setenv('THETIS_AMD_FUSE12', '2')
    mesh, bath, _, _ = channel_case(nx=450, ny=300, lx=100e3, ly=50e3)
    uv, eta = _bump_state(mesh)
    dt = 0.5
    ref = make_ref(mesh, bath, bnd_conditions=bcs)
    dev = _device(mesh, bath, dt)
    for marker, funcs in bcs.items():
        dev.set_bc(marker, funcs)
    assert dev.fused_pair_info()[0], 'the library did not take the fused stage pair on a 270 k-cell plain mesh'
    assert bool(dev.fused_triple_info()[0]) == (kernel == 'triple'), (kernel, dev.fused_triple_info())
    ur, er = uv, eta
    done = 0
    for n_steps, tol in ((1, 1e-12), (10, 1e-11), (100, 1e-10)):
        dev.set_state(uv, eta)
        dev.advance(n_steps)
        ud, ed = dev.get_state()
        ur, er = ref.advance(ur, er, dt, n_steps - done)          # the restatement steps on from where it was
        done = n_steps
        assert np.isfinite(ed).all()
        assert rel_linf(ud, ur) < tol and rel_linf(ed, er) < tol, (n_steps, rel_linf(ud, ur), rel_linf(ed, er))
    assert dev.fused_pair_info()[0]
    dev.close()


@pytest.mark.parametrize('kernel', ['pair', 'triple'])
def test_fused_stage_pair_matches_the_c_restatement_at_bench_size(hip_lib, ref_so, monkeypatch, kernel):
    """BASELINE cfg 2 itself (1 M triangles, the bench's state and time step): 20 steps of the path bench.py times - 'triple': all
    three stages in one launch, what the library takes there, asserted; 'pair': fused stage pair + stage 3 (SWE2D_OPT_FUSED_STAGES = 2) -
    against oracle/swe2d_ref.c at 1e-11."""
    from thetis_amd.mesh import RectangleMesh
    for v in ('THETIS_AMD_FUSE12', 'THETIS_AMD_FLOW', 'THETIS_AMD_BND_INLINE', 'THETIS_AMD_TRIPLE_TILE'):
        monkeypatch.delenv(v, raising=False)
    if kernel == 'pair':
        monkeypatch.setenv('THETIS_AMD_FUSE12', '2')
    mesh = RectangleMesh(1000, 500, 100e3, 50e3)
    bath = np.full(mesh.num_vertices, 20.0)
    uv, eta = _bump_state(mesh)
    dt = 0.25
    dev = _device(mesh, bath, dt)
    assert dev.fused_pair_info()[0] and bool(dev.fused_triple_info()[0]) == (kernel == 'triple')
    dev.set_state(uv, eta)
    dev.advance(20)
    ud, ed = dev.get_state()
    dev.close()
    ur, er = make_ref(mesh, bath).advance(uv, eta, dt, 20)
    assert rel_linf(ud, ur) < 1e-11 and rel_linf(ed, er) < 1e-11, (rel_linf(ud, ur), rel_linf(ed, er))


def test_stage_solutions_the_step_did_not_leave_in_memory_are_refused(hip_lib, monkeypatch):
    """swe2d_get_stage_state(h, i) hands out stage_sol[i] of the reference (rungekutta.py:930-946) - or SWE2D_ERR_UNSUPPORTED when
    the kernels that made the last step kept it on chip (fused stage pair: U(1); dataflow kernel: U(1) and U(2)), never the stale
    contents of the buffer."""
    from thetis_amd import _lib
    for v in ('THETIS_AMD_FUSE12', 'THETIS_AMD_FLOW'):
        monkeypatch.delenv(v, raising=False)
    mesh, bath, uv, eta = channel_case(nx=40, ny=20, seed=3)
    dev = _device(mesh, bath, 0.5)
    dev.set_state(uv, eta)
    for i in (0, 1):                                   # nothing has run yet
        with pytest.raises(_lib.Swe2dError) as e:
            dev.get_state(i)
        assert e.value.code == _lib.ERR_UNSUPPORTED
    dev.solve_stage(0)
    u1 = dev.get_state(0)                              # a stage launch leaves its solution
    with pytest.raises(_lib.Swe2dError):
        dev.get_state(1)
    dev.solve_stage(1)
    u2 = dev.get_state(1)
    dev.solve_stage(2)
    u3 = dev.get_state()
    assert not np.array_equal(u1[1], u2[1]) and not np.array_equal(u2[1], u3[1])
    dev.advance(1)                                     # small mesh: the dataflow kernel - neither stage solution in memory
    for i in (0, 1):
        with pytest.raises(_lib.Swe2dError) as e:
            dev.get_state(i)
        assert e.value.code == _lib.ERR_UNSUPPORTED
    dev.get_state()
    dev.set_option(_lib.OPT_FLOW, 0)
    dev.set_option(_lib.OPT_FUSED_STAGES, 1)           # forced on the small mesh: U(1) on chip, U(2) in buffer C
    dev.set_state(uv, eta)
    assert dev.fused_pair_info()[0]
    dev.advance(1)
    with pytest.raises(_lib.Swe2dError):
        dev.get_state(0)
    assert np.array_equal(dev.get_state(1)[1], u2[1]) and np.array_equal(dev.get_state()[1], u3[1])
    dev.set_option(_lib.OPT_FUSED_STAGES, 0)           # three stage launches: everything is there
    dev.set_state(uv, eta)
    dev.advance(1)
    assert np.array_equal(dev.get_state(0)[1], u1[1]) and np.array_equal(dev.get_state(1)[1], u2[1])
    dev.close()


def test_options_are_the_handles_not_the_process_environments(hip_lib, monkeypatch):
    """include/swe2d.h swe2d_set_option: the library reads no environment variable; the binding hands THETIS_AMD_* over once, at
    construction (thetis_amd/_lib.py OPTION_ENV); changing the environment afterwards changes nothing, swe2d_set_option does."""
    from thetis_amd import _lib
    monkeypatch.setenv('THETIS_AMD_FUSE12', '1')
    monkeypatch.setenv('THETIS_AMD_FLOW', '0')
    monkeypatch.setenv('THETIS_AMD_FLOW_TIMEOUT_S', '0.25')
    mesh, bath, uv, eta = channel_case(nx=40, ny=20, seed=3)
    dev = _device(mesh, bath, 0.5)
    assert dev.get_option(_lib.OPT_FUSED_STAGES) == 1 and dev.get_option(_lib.OPT_FLOW) == 0
    assert dev.get_option(_lib.OPT_FLOW_TIMEOUT_MS) == 250 and dev.get_option(_lib.OPT_LDSX) == -1
    assert dev.fused_pair_info()[0]
    monkeypatch.setenv('THETIS_AMD_FUSE12', '0')       # too late for this handle
    assert dev.fused_pair_info()[0]
    dev.set_option(_lib.OPT_FUSED_STAGES, 0)
    assert not dev.fused_pair_info()[0]
    dev.set_option(_lib.OPT_FUSED_STAGES, None)        # the library's rule: 1600 cells are far below 250 k
    assert not dev.fused_pair_info()[0]
    with pytest.raises(_lib.Swe2dError):
        dev.set_option(99, 1)
    with pytest.raises(_lib.Swe2dError):
        dev.set_option(_lib.OPT_FLOW, -7)
    dev.close()


def test_fused_stage_pair_is_what_a_large_plain_mesh_takes(hip_lib, monkeypatch):
    """swe2d_fused_pair_info: by itself from 250 k triangles on a whole mesh in the device numbering, not below, not with source
    terms or wetting-drying, not with THETIS_AMD_FUSE12=0 - so that a silent return to three stage launches shows up here."""
    from thetis_amd import _lib
    from thetis_amd.device import Swe2dDevice
    monkeypatch.delenv('THETIS_AMD_FUSE12', raising=False)
    mesh, bath, uv, eta = channel_case(nx=450, ny=300, lx=100e3, ly=50e3, seed=5)                     # 270 k cells
    dev = Swe2dDevice(mesh, bath, 0.5)
    on, tiles, ring, cells = dev.fused_pair_info()
    assert on and cells == mesh.num_cells and cells/192.0 <= tiles < cells/176.0 and 0.15*cells < ring < 0.34*cells, (on, tiles, ring, cells)
    dev.set_scalar(_lib.SCALAR_MANNING_DRAG, 0.02)
    assert dev.fused_pair_info()[0]                    # round 6: source terms are covered
    dev.set_viscosity(1.0)
    assert not dev.fused_pair_info()[0]                # viscosity is not
    dev.set_viscosity(None)
    assert dev.fused_pair_info()[0]
    dev.set_option(_lib.OPT_FUSED_STAGES, 0)
    assert not dev.fused_pair_info()[0]
    # all three stages in one launch (swe2d_fused_triple_info): what a RectangleMesh beyond the dataflow kernel takes by itself - its
    # two-ring tiles are the 11 x 8-quad patches Swe2dDevice hands over (every patch one tile) -, not with source terms, not when the
    # pair is asked for (2), not without the patches below 2.5 M cells
    dev.set_option(_lib.OPT_FUSED_STAGES, None)
    dev.set_scalar(_lib.SCALAR_MANNING_DRAG, None)
    on, tiles, ring1, ring2 = dev.fused_triple_info()
    assert on and tiles == (-(-450//11))*(-(-300//8)) and ring1 < 0.25*cells and ring2 < 0.28*cells, (on, tiles, ring1, ring2)
    dev.set_scalar(_lib.SCALAR_MANNING_DRAG, 0.02)
    assert not dev.fused_triple_info()[0] and dev.fused_pair_info()[0]
    dev.set_scalar(_lib.SCALAR_MANNING_DRAG, None)
    assert dev.fused_triple_info()[0]
    dev.set_option(_lib.OPT_FUSED_STAGES, 2)
    assert not dev.fused_triple_info()[0] and dev.fused_pair_info()[0]
    dev.set_option(_lib.OPT_FUSED_STAGES, None)
    dev.fused_set_triple_tiles(None)
    assert not dev.fused_triple_info()[0] and dev.fused_pair_info()[0]
    dev.close()
    small, bath_s, _, _ = channel_case(nx=300, ny=200, lx=100e3, ly=50e3, seed=5)                     # 120 k cells: the dataflow kernel's
    dev = Swe2dDevice(small, bath_s, 0.5)
    assert not dev.fused_pair_info()[0] and not dev.fused_triple_info()[0]
    dev.close()


@pytest.mark.parametrize('quad', [False, True])
def test_alternating_launch_direction_gives_the_same_bits(hip_lib, monkeypatch, quad):
    """Launches whose state does not fit the Infinity Cache walk the cell range alternately forwards and backwards
    (SweStageArgs::reverse): forced on a small mesh, odd and even numbers of launches, a ragged last block, sub-ranges."""
    from helpers import quad_case
    from thetis_amd.device import Swe2dDevice
    monkeypatch.setenv('THETIS_AMD_FLOW', '0')                  # the stage launches themselves
    mesh, bath, uv, eta = quad_case(nx=53, ny=31, seed=5) if quad else channel_case(nx=53, ny=31, seed=5)
    out = []
    for alt in ('0', '1'):
        monkeypatch.setenv('THETIS_AMD_ALTERNATE', alt)
        dev = Swe2dDevice(mesh, bath, 0.5)
        dev.set_bc(2, {'elev': 0.1})
        dev.set_state(uv, eta)
        dev.advance(3)
        dev.solve_stage(0)                                      # an odd number of launches so far
        n = dev.n_cells
        dev.solve_stage_cells(1, 0, n//3 + 5)
        dev.solve_stage_cells(1, n//3 + 5, n)
        dev.solve_stage(2)
        out.append(dev.get_state())
        dev.close()
    assert np.isfinite(out[0][0]).all()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


@pytest.mark.gpu
@pytest.mark.parametrize('config', ['triangles', 'triangles_flow', 'triangles_linear', 'triangles_wd_manning', 'quads', 'quads_wd_manning',
                                    'quads_general'])
def test_closed_wall_path_gives_the_bits_of_the_general_boundary_path(hip_lib, monkeypatch, config):
    """swe_boundary_facet starts with the closed wall without boundary drag as a path of its own (the blocks that own boundary
    cells set the pace of a dataflow launch); THETIS_AMD_WALL_FAST=0 sends walls through the general path: the same bits, for every
    kernel family that inlines the boundary code, next to open boundaries and a wall WITH drag (which stay on the general path)."""
    from helpers import channel_case, quad_case
    from thetis_amd import _lib
    from thetis_amd.device import Swe2dDevice
    quad = config.startswith('quads')
    if quad:
        mesh, bath, uv, eta = quad_case(nx=17, ny=9, seed=2, warp=(0.3 if config == 'quads_general' else 0.0))
    else:
        mesh, bath, uv, eta = channel_case(nx=23, ny=11, seed=1)
    wd = config.endswith('wd_manning')
    out = []
    for fast in ('1', '0'):
        monkeypatch.setenv('THETIS_AMD_WALL_FAST', fast)
        monkeypatch.setenv('THETIS_AMD_FLOW', '1' if config == 'triangles_flow' else '0')
        dev = Swe2dDevice(mesh, bath - 0.6*bath.max() if wd else bath, 0.05, boundary_len=mesh.boundary_len,
                          **({'use_nonlinear_equations': False} if config == 'triangles_linear' else {}))
        m = mesh.boundary_markers
        dev.set_bc(m[0], {'elev': 0.1})
        dev.set_bc(m[-1], {'drag': 0.01})
        if wd:
            dev.set_wetting_and_drying(0.5)
            dev.set_scalar(_lib.SCALAR_MANNING_DRAG, 0.02)
        dev.set_state(0.1*uv, 0.1*np.abs(eta))
        dev.advance(7)
        out.append(dev.get_state() + (dev.tendency(),))
        dev.close()
    assert np.isfinite(out[0][0]).all() and np.isfinite(out[0][1]).all()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][2][0], out[1][2][0]) and np.array_equal(out[0][2][1], out[1][2][1])


@pytest.mark.parametrize('cells', ['triangles', 'quadrilaterals'])
def test_diagnostics_of_small_fields_on_small_cells_keep_their_digits(hip_lib, cells):
    """the order-independent limb sums had a floor of 2^-74 ~ 5e-23 per cell term - a lake-at-rest residual of 1e-14 on a
    unit-square mesh with 1e-4 cells (eta^2 A ~ 1e-32) summed to exactly 0 where the reference's floating-point all-reduce keeps
    the value (thetis/solver2d.py:955-956, callback.py:478-482).  With six limbs (floor 2^-150) the integrals of such fields agree
    with the plain sums to rounding."""
    from thetis_amd.mesh import RectangleMesh
    mesh = RectangleMesh(100, 100, 1.0, 1.0, quadrilateral=(cells == 'quadrilaterals'))
    n, k = mesh.cells.shape
    rng = np.random.default_rng(3)
    eta = 1e-14*rng.uniform(-1, 1, size=(n, k))
    uv = 1e-12*rng.uniform(-1, 1, size=(n, k, 2))
    dev = _device(mesh, np.ones(mesh.num_vertices), 1e-3)
    dev.set_state(uv, eta)
    d = dev.diagnostics()
    dev.close()
    xy = mesh.vertex_xy[mesh.cells]
    if k == 3:
        area = 0.5*np.abs((xy[:, 1, 0] - xy[:, 0, 0])*(xy[:, 2, 1] - xy[:, 0, 1]) - (xy[:, 2, 0] - xy[:, 0, 0])*(xy[:, 1, 1] - xy[:, 0, 1]))
        m = (np.ones((3, 3)) + np.eye(3))/12.0
    else:
        area = np.abs((xy[:, 1, 0] - xy[:, 0, 0])*(xy[:, 3, 1] - xy[:, 0, 1]) - (xy[:, 1, 1] - xy[:, 0, 1])*(xy[:, 3, 0] - xy[:, 0, 0]))
        m1 = np.array([[2.0, 1.0], [1.0, 2.0]])/6.0
        # cyclic node order 0 (0,0), 1 (1,0), 2 (1,1), 3 (0,1): tensor mass matrix in that order
        pos = [(0, 0), (1, 0), (1, 1), (0, 1)]
        m = np.array([[m1[a[0], b[0]]*m1[a[1], b[1]] for b in pos] for a in pos])
    e2 = float(np.sum(area*np.einsum('ci,ij,cj->c', eta, m, eta)))
    u2 = float(np.sum(area*(np.einsum('ci,ij,cj->c', uv[..., 0], m, uv[..., 0]) + np.einsum('ci,ij,cj->c', uv[..., 1], m, uv[..., 1]))))
    assert e2 > 0 and abs(d[0] - e2) <= 1e-12*e2, (d[0], e2)
    assert abs(d[1] - u2) <= 1e-12*u2, (d[1], u2)
    assert abs(d[2] - 1.0) < 1e-13           # volume: int (eta + h) over the unit square
