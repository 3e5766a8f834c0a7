"""GPU: the one-launch SSPRK33 step (csrc/swe2d_step.h) gives the bits of three stage launches."""
import os

import numpy as np
import pytest

from helpers import channel_case, delaunay_case

pytestmark = pytest.mark.gpu


def _device(mesh, bath, dt, **kw):
    from thetis_amd.device import Swe2dDevice
    return Swe2dDevice(mesh, bath, dt, boundary_len=mesh.boundary_len, **kw)


def _steps_by_stage(dev, n):
    for _ in range(n):
        for i in range(3):
            dev.solve_stage_cells(i, 0, dev.n_cells)


def _steps_fused(dev, n):
    for _ in range(n):
        dev.solve_step_cells(0, dev.n_cells)
        dev.swap_state_buffers()


@pytest.mark.parametrize('case', ['channel', 'channel_open', 'unstructured', 'linear', 'no_lf', 'sources', 'sources_large',
                                  'ragged_ranges', 'periodic'])
def test_fused_step_gives_the_bits_of_three_stage_launches(hip_lib, case):
    from thetis_amd import _lib
    if case == 'unstructured':
        mesh, bath, uv, eta = delaunay_case(n_points=3000, seed=5)[:4]
    elif case == 'periodic':
        from thetis_amd.mesh import PeriodicRectangleMesh
        mesh = PeriodicRectangleMesh(61, 29, 100e3, 30e3, direction='x')       # tiles wrap around the periodic direction
        rng = np.random.default_rng(17)
        bath = 20.0 + 2.0*np.sin(mesh.vertex_xy[:, 1]/5000.0)
        uv, eta = 0.5*rng.normal(size=(mesh.num_cells, 3, 2)), 0.5*rng.normal(size=(mesh.num_cells, 3))
    else:
        nx, ny = (250, 125) if case == 'sources_large' else (67, 31)      # large: 62 k cells (one-ulp differences need cells to show)
        mesh, bath, uv, eta = channel_case(nx=nx, ny=ny, seed=11)
    kw = {}
    if case == 'linear':
        kw['use_nonlinear_equations'] = False
    if case == 'no_lf':
        kw['use_lax_friedrichs_velocity'] = False
    out = []
    for fused in (False, True):
        dev = _device(mesh, bath, 0.05 if case != 'unstructured' else 0.02, **kw)
        assert dev.fused_step_supported()
        k = mesh.cells.shape[1]
        cxy = mesh.cell_xy()
        if case in ('channel_open', 'sources', 'sources_large'):
            m = mesh.boundary_markers
            dev.set_bc(m[0], {'elev': 0.2*np.sin(cxy[:, :, 1]/3e3)})
            dev.set_bc(m[-1], {'un': 0.05, 'drag': 0.01})
            if len(m) > 2:
                dev.set_bc(m[1], {'flux': 30.0})
        if case in ('sources', 'sources_large'):
            dev.set_scalar(_lib.SCALAR_MANNING_DRAG, 0.02)
            dev.set_field(_lib.FIELD_CORIOLIS, 1e-4*np.ones((mesh.num_cells, k)))
            dev.set_field(_lib.FIELD_WIND_STRESS, 0.1*np.ones((mesh.num_cells, k, 2)))
        dev.set_state(uv, eta)
        if fused and case == 'ragged_ranges':
            n = dev.n_cells
            cuts = [0, 1, 65, n//3 + 7, n - 3, n]
            for _ in range(4):
                for a, b in zip(cuts[:-1], cuts[1:]):
                    dev.solve_step_cells(a, b)
                dev.swap_state_buffers()
        else:
            (_steps_fused if fused else _steps_by_stage)(dev, 4)
        out.append(dev.get_state())
        dev.close()
    assert np.isfinite(out[0][0]).all() and np.abs(out[0][0]).max() > 0
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_advance_takes_the_fused_step_and_matches_the_stage_by_stage_path(hip_lib, monkeypatch):
    mesh, bath, uv, eta = channel_case(nx=41, ny=23, seed=3)
    res = []
    for flag in ('0', '1'):
        monkeypatch.setenv('THETIS_AMD_FUSED_STEP', flag)
        dev = _device(mesh, bath, 0.05)
        dev.set_state(uv, eta)
        dev.advance(5)              # odd: the state ends in the other buffer
        res.append(dev.get_state() + (dev.diagnostics(),))
        dev.advance(2)
        res.append(dev.get_state())
        dev.close()
    assert np.array_equal(res[0][0], res[2][0]) and np.array_equal(res[0][1], res[2][1]) and np.array_equal(res[0][2], res[2][2])
    assert np.array_equal(res[1][0], res[3][0]) and np.array_equal(res[1][1], res[3][1])


def test_fused_step_is_refused_where_it_does_not_apply(hip_lib):
    from thetis_amd._lib import Swe2dError
    mesh, bath, uv, eta = channel_case(nx=9, ny=5, seed=1)
    dev = _device(mesh, bath - 0.6*bath.max(), 0.05)
    dev.set_wetting_and_drying(0.5)
    assert not dev.fused_step_supported()
    with pytest.raises(Swe2dError):
        dev.solve_step_cells(0, dev.n_cells)
    dev.close()
