"""GPU: randomised differential test - the HIP path against the numpy oracle over random combinations of every option of
the shallow-water stage (cell type, linear/nonlinear, Lax-Friedrichs, source terms as constants or fields, drag kinds,
viscosity options, wetting-drying, boundary kinds as constants or Functions, boundary drag), tendency and one SSPRK33 step.
Deterministic seeds; the single-option tests elsewhere localise a failure, this one looks for interactions."""
import numpy as np
import pytest

from helpers import channel_case, make_oracle, make_oracle_generic, quad_case, rel_linf
from thetis_amd import _lib

pytestmark = pytest.mark.gpu
TOL = 2e-12


def _random_config(rng, mesh, quad):
    n, k = mesh.num_cells, mesh.cells.shape[1]
    x, y = mesh.vertex_xy.T
    o, dev_ops = {}, []                       # oracle kwargs, device calls (name, args)
    nonlin = bool(rng.integers(0, 2))
    wd = nonlin and rng.random() < 0.25
    o['use_nonlinear_equations'] = nonlin
    o['use_lax_friedrichs_velocity'] = bool(rng.integers(0, 2))
    o['lax_friedrichs_velocity_scaling_factor'] = float(rng.choice([1.0, 0.6]))
    if wd:
        o.update(use_wetting_and_drying=True, wetting_and_drying_alpha=0.5 + 0.3*rng.random(), wd_mode='nodal')
        dev_ops.append(('set_wetting_and_drying', (o['wetting_and_drying_alpha'],)))
    if rng.random() < 0.5:
        cor = 1e-4*(1 + y/(abs(y).max() + 1.0))
        o['coriolis'] = cor
        dev_ops.append(('set_field', (_lib.FIELD_CORIOLIS, cor[mesh.cells])))
    if rng.random() < 0.4:
        pa = 1e5 + 300*np.sin(x/2e4)
        o['atmospheric_pressure'] = pa
        dev_ops.append(('set_field', (_lib.FIELD_ATMOSPHERIC_PRESSURE, pa[mesh.cells])))
    if rng.random() < 0.4:
        ms = 1e-3*rng.normal(size=(n, k, 2))
        o['momentum_source'] = ms
        dev_ops.append(('set_field', (_lib.FIELD_MOMENTUM_SOURCE, ms)))
    if rng.random() < 0.4:
        vs = 1e-3*rng.normal(size=(n, k))
        o['volume_source'] = vs
        dev_ops.append(('set_field', (_lib.FIELD_VOLUME_SOURCE, vs)))
    if rng.random() < 0.3:
        ws = 0.1*rng.normal(size=(n, k, 2))
        o['wind_stress'] = ws
        dev_ops.append(('set_field', (_lib.FIELD_WIND_STRESS, ws)))
    lin = rng.random()
    if lin < 0.25:
        o['linear_drag_coefficient'] = 1e-3
        dev_ops.append(('set_scalar', (_lib.SCALAR_LINEAR_DRAG, 1e-3)))
    elif lin < 0.45:
        c = 1e-3*(1 + x/(abs(x).max() + 1.0))
        o['linear_drag_coefficient'] = c
        dev_ops.append(('set_field', (_lib.FIELD_LINEAR_DRAG, c[mesh.cells])))
    drag = rng.integers(0, 7)
    field = 1.0 + 0.5*x/(abs(x).max() + 1.0)
    if drag == 1:
        o['quadratic_drag_coefficient'] = 0.0025
        dev_ops.append(('set_scalar', (_lib.SCALAR_QUADRATIC_DRAG, 0.0025)))
    elif drag == 2:
        o['manning_drag_coefficient'] = 0.02
        dev_ops.append(('set_scalar', (_lib.SCALAR_MANNING_DRAG, 0.02)))
    elif drag == 3:
        o['nikuradse_bed_roughness'] = 0.05
        dev_ops.append(('set_scalar', (_lib.SCALAR_NIKURADSE, 0.05)))
    elif drag == 4:
        o['manning_drag_coefficient'] = 0.02*field
        dev_ops.append(('set_field', (_lib.FIELD_MANNING_DRAG, (0.02*field)[mesh.cells])))
    elif drag == 5:
        o['quadratic_drag_coefficient'] = 0.0025*field
        dev_ops.append(('set_field', (_lib.FIELD_QUADRATIC_DRAG, (0.0025*field)[mesh.cells])))
    if drag and rng.random() < 0.5:
        o['norm_smoother'] = 0.05
        dev_ops.append(('set_scalar', (_lib.SCALAR_NORM_SMOOTHER, 0.05)))
    visc = None
    if rng.random() < 0.4:
        nu = 30.0 if rng.random() < 0.5 else 20.0 + 20.0*rng.uniform(size=mesh.num_vertices)
        visc = dict(sipg_factor=float(rng.choice([1.0, 2.0])), use_grad_div_viscosity_term=bool(rng.integers(0, 2)),
                    use_grad_depth_viscosity_term=bool(rng.integers(0, 2)))
        o.update(horizontal_viscosity=nu, **visc)
        dev_ops.append(('set_viscosity', (nu,), visc))
    # boundaries
    bcs = {}
    kinds = [None, {'elev': 1}, {'uv': 1}, {'un': 1}, {'flux': 1}, {'elev': 1, 'uv': 1}, {'elev': 1, 'un': 1}, {'elev': 1, 'flux': 1}]
    for marker in (1, 2, 3, 4):
        kind = kinds[int(rng.integers(0, len(kinds)))]
        funcs = {}
        for key in (kind or {}):
            as_field = rng.random() < 0.4
            if key == 'elev':
                funcs[key] = 0.1*rng.normal(size=(n, k)) if as_field else 0.1*rng.normal()
            elif key == 'uv':
                funcs[key] = 0.2*rng.normal(size=(n, k, 2)) if as_field else tuple(0.2*rng.normal(size=2))
            elif key == 'un':
                funcs[key] = 0.2*rng.normal(size=(n, k)) if as_field else 0.2*rng.normal()
            else:
                funcs[key] = 2e4*rng.normal(size=(n, k)) if as_field else 2e4*rng.normal()
        if rng.random() < 0.2:
            funcs['drag'] = 0.01
        if funcs:
            bcs[marker] = funcs
    o['bnd_conditions'] = bcs
    return o, dev_ops, bcs, wd


@pytest.mark.parametrize('seed', range(192))
def test_random_option_combinations_match_oracle(hip_lib, seed):
    from thetis_amd.device import Swe2dDevice
    rng = np.random.default_rng(1000 + seed)
    quad = seed % 3 == 2
    if quad:
        mesh, bath, uv, eta = quad_case(nx=7, ny=5, skew=0.25, seed=seed)
        mk = make_oracle_generic
    else:
        mesh, bath, uv, eta = channel_case(nx=7, ny=5, seed=seed)
        mk = make_oracle
    o, dev_ops, bcs, wd = _random_config(rng, mesh, quad)
    if wd:
        bath = bath - 12.0                   # partly dry
    else:
        eta = np.abs(eta)                    # keep the depth positive for the drag terms
    dt = 0.5 if wd else 2.0
    orc = mk(mesh, bath, **o)
    dev = Swe2dDevice(mesh, bath, dt, use_nonlinear_equations=o['use_nonlinear_equations'],
                      use_lax_friedrichs_velocity=o['use_lax_friedrichs_velocity'],
                      lax_friedrichs_velocity_scaling_factor=o['lax_friedrichs_velocity_scaling_factor'],
                      boundary_len=mesh.boundary_len, reorder=('hilbert' if seed % 2 else 'auto'))
    for op in dev_ops:
        getattr(dev, op[0])(*op[1], **(op[2] if len(op) > 2 else {}))
    for marker, funcs in bcs.items():
        dev.set_bc(marker, funcs)
    dev.set_state(uv, eta)
    if wd:
        eta = orc.wd_clip_state(eta)         # what set_state does on the device: nodal depths through the positivity limiter
    ku, ke = dev.tendency()
    ku_o, ke_o = orc.tendency(uv, eta, dt)
    desc = {kk: (vv if np.isscalar(vv) or isinstance(vv, (bool, str)) else type(vv).__name__) for kk, vv in o.items()}
    assert rel_linf(ku, ku_o) < TOL, desc
    scale = max(np.abs(ke_o).max(), 1e-12*np.abs(ku_o).max())      # (with wetting-drying: the tendency of zeta = D - h)
    assert np.abs(ke - ke_o).max() < TOL*max(scale, 1e-300), desc
    dev.advance(1)
    u1, e1 = dev.get_state()
    uo, eo = orc.ssprk33_step(uv, eta, dt)
    assert rel_linf(u1, uo) < 10*TOL and rel_linf(e1, eo) < 10*TOL, desc
    if dev.flow_supported():
        # two steps in one dataflow launch (csrc/swe2d_flow.h) and six stage launches: the same bits for any option combination
        dev.set_state(uv, eta)
        for i in range(6):
            dev.solve_stage(i % 3)
        us, es = dev.get_state()
        dev.set_state(uv, eta)
        dev.solve_flow([mesh.num_cells]*6)
        uf, ef = dev.get_state()
        assert np.array_equal(us, uf) and np.array_equal(es, ef), desc
    dev.close()


@pytest.mark.parametrize('seed', range(120))
def test_random_tracer_option_combinations_match_oracle(hip_lib, seed):
    """The tracer stage (+ SIPG pass): conservative / non-conservative, Lax-Friedrichs, velocity factor, source, diffusivity
    (constant or field), boundary dicts with constant / Function values, velocity keys and prescribed diffusive fluxes."""
    from thetis_amd.device import Swe2dDevice
    rng = np.random.default_rng(5000 + seed)
    quad = seed % 3 == 1
    if quad:
        mesh, bath, uv, eta = quad_case(nx=7, ny=5, skew=0.25, seed=seed)
        orc = make_oracle_generic(mesh, bath)
    else:
        mesh, bath, uv, eta = channel_case(nx=7, ny=5, seed=seed)
        orc = make_oracle(mesh, bath)
    n, k = mesh.num_cells, mesh.cells.shape[1]
    T = 3.0 + rng.normal(size=(n, k))
    dt = 2.0
    dev = Swe2dDevice(mesh, bath, dt, boundary_len=mesh.boundary_len, reorder=('hilbert' if seed % 2 else 'auto'))
    tid = dev.add_tracer()
    kw = {}
    cons = bool(rng.integers(0, 2))
    kw['conservative'] = cons
    dev.tracer_set_conservative(tid, cons)
    lf = bool(rng.integers(0, 2))
    lf_fac = float(rng.choice([1.0, 0.7]))
    vf = float(rng.choice([1.0, 0.9]))
    kw.update(use_lax_friedrichs_tracer=lf, lax_friedrichs_tracer_scaling_factor=lf_fac, tracer_advective_velocity_factor=vf)
    dev.tracer_set_options(lf, lf_fac, vf)
    if rng.random() < 0.5:
        src = 1e-3*rng.normal(size=(n, k))
        kw['source'] = src
        dev.tracer_set_source(tid, src)
    diff = rng.random() < 0.6
    if diff:
        mu = 25.0 if rng.random() < 0.5 else 15.0 + 10.0*rng.uniform(size=mesh.num_vertices)
        sipg = float(rng.choice([1.0, 1.7]))
        kw.update(diffusivity=mu, sipg_factor_tracer=sipg)
        dev.tracer_set_diffusivity(tid, mu, sipg)
    bcs = {}
    for marker in (1, 2, 3, 4):
        r = rng.integers(0, 9)
        funcs = None
        if r == 1:
            funcs = {'value': float(rng.normal())}
        elif r == 2:
            funcs = {'value': rng.normal(size=(n, k))}
        elif r == 3:
            funcs = {'value': float(rng.normal()), 'uv': 0.3*rng.normal(size=2)}
        elif r == 4:
            funcs = {'un': float(0.3*rng.normal())}
        elif r == 5 and diff:
            funcs = {'diff_flux': float(0.05*rng.normal())}
        elif r == 6:
            funcs = {'elev': 0.1}
        elif r == 7:                    # 'flux' (volume flux out of the domain) with the interior elevation
            funcs = {'value': float(rng.normal()), 'flux': float(2e3*rng.normal())}
        elif r == 8:                    # 'flux' with an external elevation
            funcs = {'flux': float(2e3*rng.normal()), 'elev': float(0.2*rng.normal())}
        if funcs is None:
            continue
        bcs[marker] = funcs
        v = funcs.get('value')
        if v is not None:
            dev.tracer_set_bc(tid, marker, v)
        if 'uv' in funcs:
            dev.tracer_set_bc_velocity(tid, marker, uv=funcs['uv'])
        elif 'flux' in funcs:
            dev.tracer_set_bc_velocity(tid, marker, flux=funcs['flux'], elev=funcs.get('elev'))
        elif 'un' in funcs:
            dev.tracer_set_bc_velocity(tid, marker, un=funcs['un'])
        if diff:
            kind = 1 if 'diff_flux' in funcs else (3 if v is None else (4 if isinstance(v, np.ndarray) else 2))
            dev.tracer_set_diffusion_bc(tid, marker, kind, funcs.get('diff_flux', 0.0))
    kw['bnd_conditions'] = bcs
    dev.set_state(uv, eta)
    dev.tracer_set_state(tid, T)
    desc = {kk: (vv if np.isscalar(vv) or isinstance(vv, (bool, str)) else type(vv).__name__) for kk, vv in kw.items()}
    assert rel_linf(dev.tracer_tendency(tid), orc.tracer_tendency(T, uv, eta, dt, **kw)) < TOL, (desc, bcs.keys())
    for s_ in range(3):
        dev.tracer_solve_stage(tid, s_)
    assert rel_linf(dev.tracer_get_state(tid), orc.tracer_ssprk33_step(T, uv, eta, dt, **kw)) < 10*TOL, desc
    dev.close()
