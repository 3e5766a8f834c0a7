"""GPU: ``swe2d_advance`` as two chains of half-launches on two streams (``swe2d_set_chains``, include/swe2d.h; numbering by
``thetis_amd.ordering.chain_order``) gives the bits of the single launches, in every kernel family that goes through
``swe2d_advance`` and over several joins."""
import numpy as np
import pytest

from helpers import channel_case, delaunay_case, quad_case

pytestmark = pytest.mark.gpu


def _case(name):
    kw, cfg = {}, {}
    if name == 'unstructured':
        mesh, bath, uv, eta = delaunay_case(n_points=3000, seed=5)[:4]
        dt = 0.02
    elif name in ('quads', 'general_quads'):
        mesh, bath, uv, eta = quad_case(nx=60, ny=31, seed=2, skew=0.1, warp=0.2 if name == 'general_quads' else 0.0)
        dt = 0.05
    else:
        mesh, bath, uv, eta = channel_case(nx=67, ny=31, seed=11)
        dt = 0.05
    return mesh, bath, uv, eta, dt, kw


def _configure(dev, mesh, name):
    from thetis_amd import _lib
    k = mesh.cells.shape[1]
    cxy = mesh.cell_xy()
    if name in ('open_sources', 'wetting_drying', 'viscosity'):
        m = mesh.boundary_markers
        dev.set_bc(m[0], {'elev': 0.2*np.sin(cxy[:, :, 1]/3e3)})
        dev.set_bc(m[-1], {'un': 0.05, 'drag': 0.01})
    if name in ('open_sources', 'wetting_drying'):
        dev.set_scalar(_lib.SCALAR_MANNING_DRAG, 0.02)
        dev.set_field(_lib.FIELD_CORIOLIS, 1e-4*np.ones((mesh.num_cells, k)))
    if name == 'wetting_drying':
        dev.set_wetting_and_drying(0.4)
    if name == 'viscosity':
        dev.set_viscosity(50.0)


@pytest.mark.parametrize('lead', ['1', '2', '50'])
@pytest.mark.parametrize('name', ['channel', 'open_sources', 'unstructured', 'wetting_drying', 'viscosity', 'quads',
                                  'general_quads'])
def test_two_chains_give_the_bits_of_single_launches(hip_lib, monkeypatch, name, lead):
    from thetis_amd.device import Swe2dDevice
    monkeypatch.setenv('THETIS_AMD_FLOW', '0')                 # meshes of test size would take the dataflow kernel
    monkeypatch.setenv('THETIS_AMD_CHAIN_LEAD', lead)
    mesh, bath, uv, eta, dt, kw = _case(name)
    n_steps = 8                                               # joins after steps 3 and 6 (nine stages between joins), then two more
    out = []
    for variant in ('chains', 'same numbering, single launches', 'plain numbering'):
        monkeypatch.setenv('THETIS_AMD_CHAINS', '1' if variant == 'chains' else '0')
        dev = Swe2dDevice(mesh, bath, dt, boundary_len=mesh.boundary_len, chains=None if variant == 'plain numbering' else 9, **kw)
        if variant != 'plain numbering':
            fe = dev._chain_front_end
            assert len(fe) == 9 and fe[0] == round(mesh.num_cells/2) and (np.diff(fe) < 0).all()
        _configure(dev, mesh, name)
        dev.set_state(uv, eta)
        dev.advance(n_steps)
        dev.synchronize()
        out.append(dev.get_state())
        dev.close()
    assert np.isfinite(out[0][0]).all() and np.abs(out[0][0]).max() > 0
    for other in out[1:]:
        assert np.array_equal(out[0][0], other[0]) and np.array_equal(out[0][1], other[1])


def test_chains_are_the_default_from_bench_size_down_to_the_flow_capacity(hip_lib):
    from thetis_amd.device import Swe2dDevice, CHAIN_MIN_CELLS, CHAIN_STAGES
    mesh, bath, uv, eta = channel_case(nx=300, ny=250, seed=3, flat=True)
    assert mesh.num_cells >= CHAIN_MIN_CELLS
    dev = Swe2dDevice(mesh, bath, 0.05)
    assert dev._chain_front_end is not None and len(dev._chain_front_end) == CHAIN_STAGES
    dev.close()
    small = channel_case(nx=100, ny=50, seed=3, flat=True)
    dev = Swe2dDevice(small[0], small[1], 0.05)
    assert dev._chain_front_end is None
    dev.close()
