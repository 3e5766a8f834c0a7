"""
The file format of ``tests/golden/reference_vectors.json``: residual-level vectors of the REFERENCE (Thetis on Firedrake) for the
hot path - what would pin the oracle (SURVEY.md section 8c says there are none in the reference's tree, VERDICT r04 "missing 2").

Written by ``tests/golden/make_reference_golden.py`` where ``import thetis`` works (a Firedrake container; not this one), read by
``tests/test_reference_golden.py`` on the GPU box.  Pure numpy / json: this module is imported by both sides and is also what
the consumer's self-test (vectors produced by the oracle in the reference's array shapes) writes with.

A case is a dict:

    name              str
    mesh              Firedrake-shaped topology tables, the arguments of thetis_amd.firedrake_adapter.swe2d_mesh_arrays:
                      coords (V,2), cell_vertices (N,k), int_facet_cell (Fi,2), int_local_facet (Fi,2), ext_facet_cell (Fe,),
                      ext_local_facet (Fe,), ext_markers (Fe,), dg_cell_nodes (N,k)
    bathymetry        (V,)   CG-P1 dofs, the coordinate numbering
    dt                float
    options           {use_nonlinear_equations, use_lax_friedrichs_velocity, lax_friedrichs_velocity_scaling_factor}
    scalars           {linear_drag_coefficient | quadratic_drag_coefficient | manning_drag_coefficient: float}   (Constants)
    fields_dg         {coriolis | atmospheric_pressure | volume_source: (ndof,), momentum_source | wind_stress: (ndof,2)}  DG-P1 dof order
    bnd               {marker: {elev | un | flux: float, uv: [u, v]}}            constant boundary values
    uv0, elev0        (ndof,2), (ndof,)   the state the vectors start from, DG dof order (uv_2d.dat.data_ro, elev_2d.dat.data_ro)
    tendency_uv, tendency_elev            ERKGenericShuOsher.tendency after ONE solver.solve() from (uv0, elev0): M^-1 dt R(U0)
    uv_1, elev_1      the solution after advance() once; uv_10, elev_10 after ten times (optional keys)

Floats travel as C99 hex strings (float.hex): exact.
"""
import json

import numpy as np

FLOAT_KEYS = ('bathymetry', 'uv0', 'elev0', 'tendency_uv', 'tendency_elev', 'uv_1', 'elev_1', 'uv_10', 'elev_10')
MESH_KEYS = ('coords', 'cell_vertices', 'int_facet_cell', 'int_local_facet', 'ext_facet_cell', 'ext_local_facet', 'ext_markers',
             'dg_cell_nodes')


def _hex(a):
    a = np.asarray(a, dtype=np.float64)
    return {'shape': list(a.shape), 'hex': [float(x).hex() for x in a.ravel()]}


def _unhex(d):
    return np.array([float.fromhex(x) for x in d['hex']], dtype=np.float64).reshape(d['shape'])


def encode_case(case):
    out = {'name': case['name'], 'dt': float(case['dt']).hex(), 'options': dict(case.get('options', {})),
           'scalars': {k: float(v).hex() for k, v in case.get('scalars', {}).items()},
           'bnd': {str(m): {k: ([float(x).hex() for x in np.atleast_1d(v)]) for k, v in f.items()} for m, f in case.get('bnd', {}).items()},
           'fields_dg': {k: _hex(v) for k, v in case.get('fields_dg', {}).items()}, 'mesh': {}}
    for k in MESH_KEYS:
        a = np.asarray(case['mesh'][k])
        out['mesh'][k] = _hex(a) if k == 'coords' else {'shape': list(a.shape), 'int': [int(x) for x in a.ravel()]}
    for k in FLOAT_KEYS:
        if k in case:
            out[k] = _hex(case[k])
    return out


def decode_case(d):
    case = {'name': d['name'], 'dt': float.fromhex(d['dt']), 'options': dict(d.get('options', {})),
            'scalars': {k: float.fromhex(v) for k, v in d.get('scalars', {}).items()},
            'bnd': {int(m): {k: (float.fromhex(v[0]) if len(v) == 1 else [float.fromhex(x) for x in v]) for k, v in f.items()}
                    for m, f in d.get('bnd', {}).items()},
            'fields_dg': {k: _unhex(v) for k, v in d.get('fields_dg', {}).items()}, 'mesh': {}}
    for k in MESH_KEYS:
        m = d['mesh'][k]
        case['mesh'][k] = _unhex(m) if k == 'coords' else np.array(m['int'], dtype=np.int64).reshape(m['shape'])
    for k in FLOAT_KEYS:
        if k in d:
            case[k] = _unhex(d[k])
    return case


def write_cases(path, cases, meta=None):
    with open(path, 'w') as f:
        json.dump({'_meta': meta or {}, 'cases': [encode_case(c) for c in cases]}, f)


def read_cases(path):
    with open(path) as f:
        d = json.load(f)
    return d.get('_meta', {}), [decode_case(c) for c in d['cases']]
