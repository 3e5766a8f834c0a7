"""CPU tests: the C-ABI library loads and exports every symbol include/swe2d.h declares; golden Shu-Osher vectors."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'swe2d.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(swe2d_[a-z0-9_]+)\s*\(', text)))


def test_every_declared_symbol_is_exported(hip_lib):
    from thetis_amd import _lib
    names = _declared_symbols()
    assert len(names) >= 20
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in names:
        assert hasattr(raw, name), 'libswe2d_hip.so does not export {:}'.format(name)
    # and the ctypes binding knows every one of them (and nothing else)
    assert sorted(_lib.SYMBOLS) == names


def test_abi_version_and_loud_failure_without_device(hip_lib):
    from thetis_amd import _lib
    from thetis_amd.device import Swe2dDevice
    from thetis_amd.mesh import RectangleMesh
    assert hip_lib.swe2d_abi_version() == _lib.ABI_VERSION == 12
    if hip_lib.swe2d_device_count() > 0:
        pytest.skip('a GPU is present')
    mesh = RectangleMesh(4, 3, 1.0, 1.0)
    with pytest.raises(_lib.Swe2dError) as err:
        Swe2dDevice(mesh, np.ones(mesh.num_vertices), 0.1)
    assert err.value.code == _lib.ERR_NO_DEVICE          # no CPU fallback


def test_create_rejects_meshes_beyond_the_32bit_plane_offsets(hip_lib):
    """The size check runs before any device work (include/swe2d.h, swe2d_create): no compute call, CPU-safe."""
    import ctypes
    from thetis_amd import _lib
    dummy_i = (ctypes.c_int32*4)()
    dummy_d = (ctypes.c_double*4)()
    dummy_b = (ctypes.c_uint8*4)()
    for npc, n_cells, expect_unsupported in ((3, 178956971, True), (4, 134217729, True), (3, 1000, False)):
        mesh = _lib.Swe2dMesh(n_cells=n_cells, n_owned=n_cells, n_vertices=10, nodes_per_cell=npc,
                              cell_vertices=ctypes.cast(dummy_i, type(_lib.Swe2dMesh().cell_vertices)),
                              vertex_xy=ctypes.cast(dummy_d, type(_lib.Swe2dMesh().vertex_xy)),
                              cell_neighbours=ctypes.cast(dummy_i, type(_lib.Swe2dMesh().cell_neighbours)),
                              cell_neighbour_facets=ctypes.cast(dummy_b, type(_lib.Swe2dMesh().cell_neighbour_facets)),
                              bathymetry=ctypes.cast(dummy_d, type(_lib.Swe2dMesh().bathymetry)),
                              boundary_len=ctypes.cast(dummy_d, type(_lib.Swe2dMesh().boundary_len)))
        par = _lib.Swe2dParams(g_grav=9.81, dt=1.0, use_nonlinear_equations=1, use_lax_friedrichs_velocity=1,
                               lax_friedrichs_velocity_scaling_factor=1.0, device_id=0)
        out = ctypes.c_void_p()
        if not expect_unsupported and hip_lib.swe2d_device_count() > 0:
            continue                                    # would really build a handle from the dummy arrays
        rc = hip_lib.swe2d_create(ctypes.byref(mesh), ctypes.byref(par), ctypes.byref(out))
        assert rc == (_lib.ERR_UNSUPPORTED if expect_unsupported else _lib.ERR_NO_DEVICE), rc
        if expect_unsupported:
            assert b'4 GiB' in hip_lib.swe2d_last_error(None)


def test_struct_layout_matches_header(hip_lib):
    from thetis_amd import _lib
    # swe2d_mesh: 4 x int32 + 6 pointers; swe2d_params: see include/swe2d.h
    assert ctypes.sizeof(_lib.Swe2dMesh) == 16 + 6*8
    assert ctypes.sizeof(_lib.Swe2dParams) == 8 + 8 + 4 + 4 + 8 + 4 + 4


def _golden():
    with open(os.path.join(ROOT, 'tests', 'golden', 'shuosher_ssprk33.json')) as f:
        return json.load(f)


def test_shuosher_golden_pins_python_classes_and_oracle():
    g = _golden()
    alpha = np.array([[float.fromhex(x) for x in row] for row in g['alpha_hex']])
    beta = np.array([[float.fromhex(x) for x in row] for row in g['beta_hex']])
    assert np.array_equal(alpha, np.array(g['alpha'])) and np.array_equal(beta, np.array(g['beta']))
    from thetis_amd.rungekutta import SSPRK33Abstract
    assert np.array_equal(SSPRK33Abstract.alpha, alpha)        # bit-exact
    assert np.array_equal(SSPRK33Abstract.beta, beta)
    assert np.array_equal(SSPRK33Abstract.a, np.array(g['a']))
    assert np.array_equal(SSPRK33Abstract.b, np.array(g['b']))
    assert list(SSPRK33Abstract.c) == list(g['c']) and SSPRK33Abstract.cfl_coeff == g['cfl_coeff']
    from oracle import swe2d_oracle as orc
    assert np.array_equal(np.array(orc.SSPRK33_ALPHA), alpha)
    assert np.array_equal(np.array(orc.SSPRK33_BETA), beta)
    assert list(orc.SSPRK33_C) == list(g['c'])
    # consistency of a Shu-Osher form (rungekutta.py:80-85)
    assert np.allclose(alpha.sum(axis=1), 1.0)


def test_shuosher_golden_pins_device_coefficients(hip_lib):
    g = _golden()
    a0 = (ctypes.c_double*3)()
    ai = (ctypes.c_double*3)()
    be = (ctypes.c_double*3)()
    hip_lib.swe2d_ssprk33_coefficients(a0, ai, be)
    alpha = np.array(g['alpha'])
    beta = np.array(g['beta'])
    for i in range(3):
        assert be[i] == beta[i + 1][i]
        # stage i: sum_j alpha[i+1][j] U_j, of which only j = 0 and j = i are non-zero for SSPRK33
        if i == 0:
            assert ai[0] == alpha[1][0] and a0[0] == 0.0
        else:
            assert a0[i] == alpha[i + 1][0] and ai[i] == alpha[i + 1][i]
            assert all(alpha[i + 1][j] == 0.0 for j in range(1, i))


def test_general_butcher_to_shuosher_conversion_matches_the_reference_output():
    """thetis_amd.rungekutta.butcher_to_shuosher_form (own back substitution) against tests/golden/shuosher_explicit.json = the
    reference's function executed on every explicit tableau of its file (tests/golden/make_shuosher_golden.py): bit for bit for
    SSPRK33 (what the device coefficients are), to round-off for the others (the reference inverts with LAPACK)."""
    import json
    import os
    from thetis_amd.rungekutta import SSPRK33Abstract, butcher_to_shuosher_form
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'shuosher_explicit.json')) as f:
        gold = json.load(f)['schemes']
    assert set(gold) >= {'SSPRK33Abstract', 'ForwardEulerAbstract', 'ERKMidpointAbstract', 'ERKLSPUM2Abstract', 'ERKLPUM2Abstract'}
    for name, g in gold.items():
        al, be = butcher_to_shuosher_form(np.array(g['a']), np.array(g['b']))
        al_g = np.array([[float.fromhex(x) for x in r] for r in g['alpha_hex']])
        be_g = np.array([[float.fromhex(x) for x in r] for r in g['beta_hex']])
        assert np.allclose(al, al_g, rtol=0, atol=4e-16) and np.allclose(be, be_g, rtol=0, atol=4e-16), name
        if name == 'SSPRK33Abstract':
            assert np.array_equal(al, al_g) and np.array_equal(be, be_g)
            assert np.array_equal(al, SSPRK33Abstract.alpha) and np.array_equal(be, SSPRK33Abstract.beta)
    with pytest.raises(NotImplementedError):
        butcher_to_shuosher_form(np.array([[0.5]]), np.array([1.0]))              # implicit midpoint: not this path


def test_limb_sums_round_to_the_nearest_double():
    """swe2d_sum_limbs_to_double (host code of the library: runs without a GPU) against exact rational arithmetic: random limb
    totals incl. negative ones, ties, carries between limbs, values below one unit of the top limb."""
    import ctypes
    from fractions import Fraction
    import numpy as np
    from thetis_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)

    units = (40, 2, -36, -74, -112, -150)

    def exact(limbs):
        return float(sum(Fraction(int(l))*Fraction(2)**s for l, s in zip(limbs, units)))

    four = [[0, 0, 0, 0], [0, 0, 0, 1], [0, 0, 0, -1], [1, 0, 0, 0], [-1, 0, 0, 1], [0, 0, 1 << 37, 1], [0, (1 << 53) + 1, 0, 0],
            [1 << 13, 0, 0, 1], [1 << 13, 0, 0, -1], [-(1 << 20), 3, -5, 7], [0, 1 << 52, 1 << 36, 0], [0, 1 << 52, 1 << 36, 1],
            [0, (1 << 52) + 1, 1 << 36, 0], [(1 << 62) - 1, (1 << 62) - 1, (1 << 62) - 1, (1 << 62) - 1],
            [-(1 << 62), -(1 << 62), -(1 << 62), -(1 << 62)]]
    # the round-4 cases in the upper four limbs, in the lower four, and with a lone unit far below them (a sticky bit that decides a tie)
    cases = [c + [0, 0] for c in four] + [[0, 0] + c for c in four] + [c + [0, 1] for c in four] + [c + [0, -1] for c in four]
    cases += [[0, 0, 0, 0, 0, 1], [0, 0, 0, 0, 0, -1], [0, 0, 0, 0, 1, 0], [-1, 0, 0, 0, 0, 1], [1, 0, 0, 0, 0, -1], [0, -1, 0, 0, 0, 1],
              [(1 << 62) - 1]*6, [-(1 << 62)]*6, [1 << 52, 0, 0, 0, 0, 1], [(1 << 53) + 1, 0, 0, 0, 0, 0], [0, 0, 0, (1 << 53) + 1, 0, 0]]
    for _ in range(3000):
        mag = rng.integers(1, 62, size=6)
        cases.append([int(rng.integers(-(1 << m), 1 << m)) if rng.random() > 0.35 else 0 for m in mag])
    for c in cases:
        a = np.array(c, dtype=np.int64)
        got = lib.swe2d_sum_limbs_to_double(a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
        assert got == exact(c), (c, got, exact(c))
