"""
The HIP path against residual-level vectors produced BY THE REFERENCE (tests/golden/reference_vectors.json, written by
tests/golden/make_reference_golden.py in a Firedrake container) - the one test that would pin the oracle (SURVEY.md section 8c,
VERDICT r04 "missing 2").  The file cannot be produced in this repository's build container (no Firedrake); while it is absent the
test SKIPS, loudly.  What IS tested here without it: the consumer itself - file format, Firedrake-shaped topology tables through
thetis_amd/firedrake_adapter.py, DG dof permutation, options / fields / boundary plumbing - on vectors the oracle writes in the
same format, with a scrambled DG numbering and clockwise cells, so that the day the generator runs nothing else has to change.
"""
import os
import sys
import types

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
from reference_vector_format import read_cases, write_cases      # noqa: E402

VECTORS = os.path.join(HERE, 'golden', 'reference_vectors.json')
FIELD_IDS = {'coriolis': 0, 'atmospheric_pressure': 1, 'momentum_source': 2, 'volume_source': 3, 'wind_stress': 4}
SCALAR_IDS = {'linear_drag_coefficient': 0, 'quadratic_drag_coefficient': 1, 'manning_drag_coefficient': 2}


def device_for(case):
    """(device, dg_perm) for a case in the reference's array shapes: what INTEGRATION.md section 2 does inside SSPRK33HIP.__init__"""
    from thetis_amd.device import Swe2dDevice
    from thetis_amd.firedrake_adapter import swe2d_mesh_arrays
    m = case['mesh']
    a = swe2d_mesh_arrays(m['coords'], m['cell_vertices'], m['int_facet_cell'], m['int_local_facet'], m['ext_facet_cell'],
                          m['ext_local_facet'], m['ext_markers'], dg_cell_nodes=m['dg_cell_nodes'])
    cells, xy, nbr = a['cell_vertices'], a['vertex_xy'], a['cell_neighbours']
    k = cells.shape[1]
    # boundary lengths per marker (shallowwater_eq.py:214-223): the 'flux' key divides by them
    blen = {}
    for c, f in zip(*np.nonzero(nbr < 0)):
        p, q = xy[cells[c, f]], xy[cells[c, (f + 1) % k]]
        blen[int(-nbr[c, f])] = blen.get(int(-nbr[c, f]), 0.0) + float(np.hypot(*(q - p)))
    mesh = types.SimpleNamespace(cells=cells, vertex_xy=xy, cell_nbr=nbr, cell_nbr_facet=a['cell_neighbour_facets'], topo_vertex=None,
                                 num_cells=len(cells), structured=False)
    o = case['options']
    dev = Swe2dDevice(mesh, case['bathymetry'], case['dt'], use_nonlinear_equations=o.get('use_nonlinear_equations', True),
                      use_lax_friedrichs_velocity=o.get('use_lax_friedrichs_velocity', True),
                      lax_friedrichs_velocity_scaling_factor=o.get('lax_friedrichs_velocity_scaling_factor', 1.0), boundary_len=blen,
                      reorder='hilbert')
    perm = a['dg_perm']                                       # (N, k): DG dof of node i of cell c
    for key, value in case['scalars'].items():
        dev.set_scalar(SCALAR_IDS[key], value)
    for key, value in case['fields_dg'].items():
        dev.set_field(FIELD_IDS[key], np.asarray(value)[perm])
    for marker, funcs in case['bnd'].items():
        dev.set_bc(marker, funcs)
    return dev, perm


def run_case_on_device(case):
    dev, perm = device_for(case)
    out = {}
    try:
        dev.set_state(case['uv0'][perm], case['elev0'][perm])
        out['tendency_uv'], out['tendency_elev'] = dev.tendency()
        dev.advance(1)
        out['uv_1'], out['elev_1'] = dev.get_state()
        dev.advance(9)
        out['uv_10'], out['elev_10'] = dev.get_state()
    finally:
        dev.close()
    return out, perm


def compare(case, out, perm, tol_1, tol_10):
    worst = {}
    for key, tol in (('tendency_uv', tol_1), ('tendency_elev', tol_1), ('uv_1', tol_1), ('elev_1', tol_1), ('uv_10', tol_10), ('elev_10', tol_10)):
        if key not in case:
            continue
        ref = np.asarray(case[key])[perm]
        scale = max(np.abs(ref).max(), 1e-300)
        worst[key] = float(np.abs(out[key] - ref).max()/scale)
        assert worst[key] <= tol, '{:} / {:}: rel. L_inf {:.3e} > {:.1e}'.format(case['name'], key, worst[key], tol)
    return worst


@pytest.mark.gpu
def test_hip_path_matches_the_vectors_of_the_reference(hip_lib):
    """tendency and one step at 1e-12, ten steps at 1e-11 (relative L_inf), every case of the reference's file"""
    if not os.path.exists(VECTORS):
        pytest.skip('PARITY UNPINNED: tests/golden/reference_vectors.json is absent - run tests/golden/make_reference_golden.py where '
                    '`import thetis` works (a Firedrake container) and commit the file it writes')
    meta, cases = read_cases(VECTORS)
    assert cases, 'empty vector file'
    report = {}
    for case in cases:
        out, perm = run_case_on_device(case)
        report[case['name']] = compare(case, out, perm, 1e-12, 1e-11)
    print('reference vectors ({:}): {:}'.format(meta, report))


# ---- the consumer's self-test: the same file, written by the oracle in the reference's array shapes
def firedrake_shaped(mesh, rng):
    """Mesh2d -> the tables a Firedrake mesh hands out (FIAT local numbering), with every second triangle stored CLOCKWISE and the
    DG dofs numbered at random: the inverse of thetis_amd/firedrake_adapter.py, written independently of it"""
    cells = np.asarray(mesh.cells)
    n, k = cells.shape
    nbr, nbf = np.asarray(mesh.cell_nbr), np.asarray(mesh.cell_nbr_facet)
    if k == 3:
        # local vertex order handed out: counter-clockwise as it is, or clockwise (local vertices 1 and 2 swapped)
        flip = (np.arange(n) % 2) == 1
        order = np.where(flip[:, None], np.array([0, 2, 1]), np.array([0, 1, 2]))          # handed-out local j = our local order[j]
        # FIAT facet i = the edge opposite local vertex i
        def fiat_facet(c, f):          # our facet f of cell c joins our local vertices f, f + 1: opposite our local vertex f + 2
            ours = (f + 2) % 3
            return int(np.nonzero(order[c] == ours)[0][0])
    else:
        flip = np.zeros(n, dtype=bool)
        order = np.tile(np.array([0, 3, 1, 2]), (n, 1))                                     # lexicographic j = our cyclic order[j]
        lex_facets = {frozenset((0, 1)): 0, frozenset((2, 3)): 1, frozenset((0, 2)): 2, frozenset((1, 3)): 3}

        def fiat_facet(c, f):
            a = int(np.nonzero(order[c] == f)[0][0])
            b = int(np.nonzero(order[c] == (f + 1) % 4)[0][0])
            return lex_facets[frozenset((a, b))]
    cv = np.take_along_axis(cells, order, axis=1)
    ifc, ilf, efc, elf, emk = [], [], [], [], []
    for c in range(n):
        for f in range(k):
            if nbr[c, f] < 0:
                efc.append(c); elf.append(fiat_facet(c, f)); emk.append(-int(nbr[c, f]))
            elif nbr[c, f] > c:
                c2, f2 = int(nbr[c, f]), int(nbf[c, f])
                ifc.append([c, c2]); ilf.append([fiat_facet(c, f), fiat_facet(c2, f2)])
    dof = rng.permutation(n*k).reshape(n, k)                                                 # dof of handed-out local node j of cell c
    return {'coords': np.asarray(mesh.vertex_xy), 'cell_vertices': cv, 'int_facet_cell': np.array(ifc).reshape(-1, 2),
            'int_local_facet': np.array(ilf).reshape(-1, 2), 'ext_facet_cell': np.array(efc), 'ext_local_facet': np.array(elf),
            'ext_markers': np.array(emk), 'dg_cell_nodes': dof}, order


def oracle_case(name, mesh, bath, uv, eta, dt, rng, **kw):
    """a case of the file format whose vectors come from the numpy oracle; arrays in DG dof order like uv_2d.dat.data"""
    from oracle.swe2d_oracle import SWEOracle
    tables, order = firedrake_shaped(mesh, rng)
    n, k = mesh.cells.shape
    dof = tables['dg_cell_nodes']

    def to_dofs(a):              # (N, k, ...) in our node order -> (ndof, ...) in DG dof order
        a = np.asarray(a)
        out = np.empty((n*k,) + a.shape[2:])
        out[dof.ravel()] = np.take_along_axis(a, order.reshape(order.shape + (1,)*(a.ndim - 2)), axis=1).reshape((n*k,) + a.shape[2:])
        return out
    orc = SWEOracle(mesh.vertex_xy, mesh.cells, bath, **kw)
    k_uv, k_eta = orc.tendency(uv, eta, dt)
    u1, e1 = orc.ssprk33_step(uv, eta, dt)
    u10, e10 = u1, e1
    for _ in range(9):
        u10, e10 = orc.ssprk33_step(u10, e10, dt)
    return {'name': name, 'mesh': tables, 'bathymetry': bath, 'dt': dt,
            'options': {'use_nonlinear_equations': kw.get('nonlinear', True), 'use_lax_friedrichs_velocity': kw.get('use_lf', True),
                        'lax_friedrichs_velocity_scaling_factor': kw.get('lf_factor', 1.0)},
            'scalars': {}, 'fields_dg': {}, 'bnd': {},
            'uv0': to_dofs(uv), 'elev0': to_dofs(eta), 'tendency_uv': to_dofs(k_uv), 'tendency_elev': to_dofs(k_eta),
            'uv_1': to_dofs(u1), 'elev_1': to_dofs(e1), 'uv_10': to_dofs(u10), 'elev_10': to_dofs(e10)}


@pytest.mark.gpu
def test_the_consumer_reads_what_the_generator_writes(hip_lib, tmp_path):
    """Vectors from the oracle, written with the generator's writer in the reference's shapes (clockwise cells, FIAT facet numbers,
    random DG numbering; triangles and quadrilaterals), read back and compared like the reference's file will be."""
    import inspect
    from oracle.swe2d_oracle import SWEOracle
    from thetis_amd.mesh import RectangleMesh
    rng = np.random.default_rng(5)
    cases = []
    for quad in (False, True):
        mesh = RectangleMesh(9, 5, 9e3, 5e3, quadrilateral=quad)
        x, y = mesh.vertex_xy.T
        bath = 10.0 + 3.0*x/9e3 + 2.0*y/5e3
        cxy = mesh.cell_xy()
        eta = 0.1*np.cos(np.pi*cxy[:, :, 0]/9e3)*np.cos(np.pi*cxy[:, :, 1]/5e3)
        uv = np.stack([0.1*np.sin(np.pi*cxy[:, :, 0]/9e3), -0.05*np.sin(np.pi*cxy[:, :, 1]/5e3)], axis=-1)
        cases.append(oracle_case('quad' if quad else 'tri', mesh, bath, uv, eta, 2.0, rng))
    assert 'tendency' in dict(inspect.getmembers(SWEOracle))
    path = str(tmp_path/'vectors.json')
    write_cases(path, cases, {'generator': 'oracle (self-test of the consumer)'})
    _, back = read_cases(path)
    assert [c['name'] for c in back] == ['tri', 'quad']
    assert np.array_equal(back[0]['uv0'], cases[0]['uv0']) and np.array_equal(back[1]['mesh']['dg_cell_nodes'], cases[1]['mesh']['dg_cell_nodes'])
    for case in back:
        out, perm = run_case_on_device(case)
        compare(case, out, perm, 1e-11, 1e-10)


def test_vector_file_round_trip_is_exact(tmp_path):
    """no GPU: hex floats and integer tables survive the file bit for bit"""
    rng = np.random.default_rng(0)
    case = {'name': 'x', 'dt': 0.1, 'options': {'use_nonlinear_equations': True}, 'scalars': {'manning_drag_coefficient': 0.03},
            'bnd': {2: {'elev': 0.25, 'uv': [0.1, -0.2]}}, 'fields_dg': {'coriolis': rng.standard_normal(6)},
            'mesh': {'coords': rng.standard_normal((4, 2)), 'cell_vertices': np.array([[0, 1, 2], [2, 1, 3]]), 'int_facet_cell': np.array([[0, 1]]),
                     'int_local_facet': np.array([[0, 2]]), 'ext_facet_cell': np.array([0, 0, 1, 1]), 'ext_local_facet': np.array([1, 2, 0, 1]),
                     'ext_markers': np.array([1, 3, 2, 4]), 'dg_cell_nodes': np.array([[0, 1, 2], [3, 4, 5]])},
            'bathymetry': rng.standard_normal(4), 'uv0': rng.standard_normal((6, 2)), 'elev0': rng.standard_normal(6)*1e-300}
    p = str(tmp_path/'v.json')
    write_cases(p, [case])
    _, (back,) = read_cases(p)
    for key in ('bathymetry', 'uv0', 'elev0'):
        assert np.array_equal(back[key], case[key])
    assert back['bnd'] == {2: {'elev': 0.25, 'uv': [0.1, -0.2]}} and back['scalars'] == case['scalars'] and back['dt'] == 0.1
    assert all(np.array_equal(back['mesh'][k], case['mesh'][k]) for k in case['mesh'])
    assert np.array_equal(back['fields_dg']['coriolis'], case['fields_dg']['coriolis'])
