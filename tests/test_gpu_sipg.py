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


def test_sipg_rejects_unsupported_configurations(hip_lib):
    from helpers import quad_case
    mesh, bath, uv, eta = quad_case()
    dev = _dev(mesh, bath, 1.0)
    with pytest.raises(RuntimeError, match='triangles only'):
        dev.set_viscosity(1.0)
    dev.close()
    mesh, bath, uv, eta = channel_case()
    dev = _dev(mesh, bath, 1.0)
    dev.set_wetting_and_drying(0.5)
    with pytest.raises(RuntimeError, match='wetting'):
        dev.set_viscosity(1.0)
    dev.close()
