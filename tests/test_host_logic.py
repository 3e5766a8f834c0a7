"""CPU tests of the host-side mirror of the reference interface: options, functions/projection, meshes, CFL dt."""
import math

import numpy as np
import pytest

from helpers import make_oracle
from thetis_amd import (Constant, Function, ModelOptions2d, PeriodicRectangleMesh, RectangleMesh, get_functionspace)


def test_options_defaults_match_reference():
    o = ModelOptions2d()
    # thetis/options.py:583-733,866-945
    assert o.element_family == 'dg-dg' and o.polynomial_degree == 1
    assert o.use_nonlinear_equations is True and o.use_lax_friedrichs_velocity is True
    assert float(o.lax_friedrichs_velocity_scaling_factor) == 1.0
    assert o.timestep == 10.0 and o.simulation_export_time == 100.0 and o.simulation_end_time is None
    assert float(o.horizontal_velocity_scale) == 0.1 and o.cfl_2d == 1.0
    assert o.swe_timestepper_type == 'CrankNicolson' and o.use_limiter_for_tracers is True
    assert o.use_wetting_and_drying is False and float(o.wetting_and_drying_alpha) == 0.5
    assert o.manning_drag_coefficient is None and o.coriolis_frequency is None
    assert float(o.norm_smoother) == 0.0 and o.check_volume_conservation_2d is False


def test_options_are_frozen_validated_and_paired():
    o = ModelOptions2d()
    with pytest.raises(TypeError):
        o.not_an_option = 1                                # configuration.py:294-331
    with pytest.raises(AssertionError):
        o.timestep = -1.0
    with pytest.raises(ValueError):
        o.swe_timestepper_type = 'RK4'
    o.swe_timestepper_type = 'SSPRK33'                     # options.py:838-852 paired options
    assert o.swe_timestepper_options.use_automatic_timestep is True
    assert o.swe_timestepper_options.solver_parameters['pc_type'] == 'bjacobi'   # options.py:145-152
    assert not hasattr(ModelOptions2d().swe_timestepper_options, 'use_automatic_timestep')  # CrankNicolson default
    o.set_timestepper_type('SSPRK33', use_automatic_timestep=False)
    assert o.tracer_timestepper_type == 'SSPRK33' and o.swe_timestepper_options.use_automatic_timestep is False
    o.update({'timestep': 2.0, 'horizontal_velocity_scale': 6.0})
    assert o.timestep == 2.0 and isinstance(o.horizontal_velocity_scale, Constant)
    o.add_tracer_2d('tracer_2d', 'Depth averaged tracer', 'Tracer2d')
    with pytest.raises(AssertionError):
        o.add_tracer_2d('tracer_2d', 'again', 'Tracer2d')


def test_rectangle_mesh_conventions():
    m = RectangleMesh(80, 3, 100e3, 3750.0)                # examples/channel2d/channel2d.py:21-25
    assert m.num_cells == 480 and m.num_vertices == 81*4
    assert m.boundary_markers == [1, 2, 3, 4]
    assert m.boundary_len == {1: 3750.0, 2: 3750.0, 3: 100e3, 4: 100e3}
    assert np.all(m.cell_areas() > 0) and math.isclose(m.cell_areas().sum(), 100e3*3750.0)
    # neighbour symmetry and shared vertices
    for k in range(m.num_cells):
        for f in range(3):
            nb = m.cell_nbr[k, f]
            if nb >= 0:
                f2 = m.cell_nbr_facet[k, f]
                assert m.cell_nbr[nb, f2] == k and m.cell_nbr_facet[nb, f2] == f
                assert m.cells[k, f] == m.cells[nb, (f2 + 1) % 3] and m.cells[k, (f + 1) % 3] == m.cells[nb, f2]
    # marker geometry: 1: x=0, 2: x=Lx, 3: y=0, 4: y=Ly
    c, f = np.nonzero(m.cell_nbr < 0)
    mid = 0.5*(m.vertex_xy[m.cells[c, f]] + m.vertex_xy[m.cells[c, (f + 1) % 3]])
    mk = -m.cell_nbr[c, f]
    assert np.allclose(mid[mk == 1, 0], 0) and np.allclose(mid[mk == 2, 0], 100e3)
    assert np.allclose(mid[mk == 3, 1], 0) and np.allclose(mid[mk == 4, 1], 3750.0)


def test_periodic_mesh_and_renumbering():
    m = PeriodicRectangleMesh(6, 4, 3.0, 2.0, direction='x')
    assert m.boundary_markers == [3, 4] and (m.cell_nbr >= 0).sum() == 3*m.num_cells - 2*6
    perm = np.random.default_rng(0).permutation(m.num_cells)
    r = m.renumbered(perm)
    for k in range(r.num_cells):
        for f in range(3):
            nb = r.cell_nbr[k, f]
            if nb >= 0:
                assert perm[nb] == m.cell_nbr[perm[k], f]
            else:
                assert nb == m.cell_nbr[perm[k], f]


def test_projection_and_interpolation():
    m = RectangleMesh(7, 5, 3.0, 2.0)
    P1 = get_functionspace(m, 'CG', 1)
    P1DG = get_functionspace(m, 'DG', 1)
    P1DGv = get_functionspace(m, 'DG', 1, vector=True)
    lin = lambda x, y: 1.0 + 2.0*x - 3.0*y
    f = Function(P1DG).project(lin)
    assert np.allclose(f.dat.data_ro, lin(*P1DG.node_xy().T), atol=1e-13)    # P1 functions are reproduced
    g = Function(P1).interpolate(lin)
    h = Function(P1DG).project(g)                                           # CG-P1 -> DG-P1 is injection
    assert np.allclose(h.dat.data_ro, f.dat.data_ro, atol=1e-13)
    # non-polynomial: equals the oracle's projection (same rule)
    orc = make_oracle(m, np.ones(m.num_vertices))
    fn = lambda x, y: np.cos(x)*np.exp(-y)
    assert np.allclose(Function(P1DG).project(fn).cell_node_values(), orc.project(fn), atol=1e-14)
    fv = lambda x, y: (np.sin(x), y*x)
    assert np.allclose(Function(P1DGv).project(fv).cell_node_values(), orc.project(fv, vector=True), atol=1e-14)
    assert Function(P1DG).assign(Constant(2.5)).dat.data_ro.min() == 2.5


def test_automatic_time_step_formula():
    """solver2d.py:149-177,213-248 on a uniform mesh: dt = cfl_2d * 0.05 * sqrt(cell area)/(sqrt(g h) + U)."""
    from thetis_amd import solver2d
    from thetis_amd.cgproject import elem_size_p1
    m = RectangleMesh(10, 4, 1000.0, 400.0)
    bath = Function(get_functionspace(m, 'CG', 1)).assign(20.0)
    s = solver2d.FlowSolver2d(m, bath)
    s.options.swe_timestepper_type = 'SSPRK33'
    s.options.horizontal_velocity_scale = Constant(6.0)
    s.create_function_spaces()
    s.create_fields()
    s.fields.h_elem_size_2d = Function(s.function_spaces.P1_2d).assign(elem_size_p1(m))
    s.set_time_step()
    expect = 1.0*0.05*math.sqrt(100.0*100.0/2)/(math.sqrt(9.81*20.0) + 6.0)
    assert math.isclose(s.dt, expect, rel_tol=1e-12)
    s.options.swe_timestepper_options.use_automatic_timestep = False
    s.options.timestep = 2.0
    s.set_time_step()
    assert s.dt == 2.0


def test_viscosity_and_diffusivity_configuration_checks():
    """Unsupported SIPG configurations raise instead of silently dropping the term."""
    from thetis_amd import Constant, Function, RectangleMesh, get_functionspace
    from thetis_amd.options import ModelOptions2d
    from thetis_amd.shallowwater_eq import DepthExpression, ShallowWaterEquations
    mesh = RectangleMesh(4, 3, 10.0, 10.0)
    bath = Function(get_functionspace(mesh, 'CG', 1)).assign(5.0)
    opts = ModelOptions2d()
    eq = ShallowWaterEquations(get_functionspace(mesh, 'DG', 1), DepthExpression(bath), opts)
    eq.check_fields({'viscosity_h': Constant(1.0)})
    eq.check_fields({'viscosity_h': Function(get_functionspace(mesh, 'CG', 1)).assign(2.0)})
    with pytest.raises(NotImplementedError, match='continuous'):
        eq.check_fields({'viscosity_h': Function(get_functionspace(mesh, 'DG', 1)).assign(2.0)})
    qmesh = RectangleMesh(4, 3, 10.0, 10.0, quadrilateral=True)
    qbath = Function(get_functionspace(qmesh, 'CG', 1)).assign(5.0)
    qeq = ShallowWaterEquations(get_functionspace(qmesh, 'DG', 1), DepthExpression(qbath), opts)
    qeq.check_fields({'viscosity_h': Constant(1.0)})          # quadrilaterals: swe_sipg_kernel_quad


def test_drag_parameter_combinations_raise_like_the_reference():
    """shallowwater_eq.py:686-696: at most one of quadratic / Manning / Nikuradse."""
    from thetis_amd import Constant, Function, RectangleMesh, get_functionspace
    from thetis_amd.options import ModelOptions2d
    from thetis_amd.shallowwater_eq import DepthExpression, ShallowWaterEquations
    mesh = RectangleMesh(4, 3, 10.0, 10.0)
    bath = Function(get_functionspace(mesh, 'CG', 1)).assign(5.0)
    eq = ShallowWaterEquations(get_functionspace(mesh, 'DG', 1), DepthExpression(bath), ModelOptions2d())
    eq.check_fields({'nikuradse_bed_roughness': Constant(0.05)})
    with pytest.raises(Exception, match='Nikuradse drag and Manning'):
        eq.check_fields({'nikuradse_bed_roughness': Constant(0.05), 'manning_drag_coefficient': Constant(0.02)})
    with pytest.raises(Exception, match='dimensionless and Nikuradse'):
        eq.check_fields({'nikuradse_bed_roughness': Constant(0.05), 'quadratic_drag_coefficient': Constant(0.002)})
    with pytest.raises(Exception, match='dimensionless and Manning'):
        eq.check_fields({'manning_drag_coefficient': Constant(0.02), 'quadratic_drag_coefficient': Constant(0.002)})


def test_vtu_export_is_readable_binary(tmp_path):
    """The .vtu writer (XML header + raw appended blocks, UInt64 byte counts): parse it back and compare."""
    import re
    from thetis_amd import Function, RectangleMesh, get_functionspace
    from thetis_amd.exporter import VTKExporter
    mesh = RectangleMesh(5, 3, 10.0, 6.0)
    H = get_functionspace(mesh, 'DG', 1)
    U = get_functionspace(mesh, 'DG', 1, vector=True)
    elev = Function(H, name='elev_2d').interpolate(lambda x, y: 0.1*x - 0.2*y)
    uv = Function(U, name='uv_2d').interpolate(lambda x, y: (x, -y))
    for func, name in ((elev, 'Elevation2d'), (uv, 'Velocity2d')):
        ex = VTKExporter(func.name(), str(tmp_path), name)
        ex.export(func, time=1.5)
        raw = open(tmp_path/name/(name + '_0.vtu'), 'rb').read()
        head, tail = raw.split(b'<AppendedData encoding="raw">\n_')
        offs = [int(o) for o in re.findall(rb'offset="(\d+)"', head)]
        assert len(offs) == 5 and b'header_type="UInt64"' in head

        def block(i, dtype):
            nbytes = int(np.frombuffer(tail, dtype='<u8', count=1, offset=offs[i])[0])
            return np.frombuffer(tail, dtype=dtype, count=nbytes//np.dtype(dtype).itemsize, offset=offs[i] + 8)
        pts = block(0, '<f8').reshape(-1, 3)
        assert np.array_equal(pts[:, :2], mesh.cell_xy().reshape(-1, 2)) and not pts[:, 2].any()
        assert np.array_equal(block(1, '<i4'), np.arange(3*mesh.num_cells))
        assert np.array_equal(block(2, '<i4'), 3*(np.arange(mesh.num_cells) + 1)) and (block(3, 'u1') == 5).all()
        vals = func.cell_node_values()
        data = block(4, '<f8')
        if vals.ndim == 3:
            assert np.array_equal(data.reshape(-1, 3)[:, :2], vals.reshape(-1, 2))
        else:
            assert np.array_equal(data, vals.reshape(-1))
        assert tail.rstrip().endswith(b'</VTKFile>')
        assert 'timestep="1.5"' in open(tmp_path/name/(name + '.pvd')).read()


def test_firedrake_shaped_arrays_give_the_swe2d_mesh_of_mesh2d():
    """The array contract of the reference-side binding (thetis_amd/firedrake_adapter.py): a
    hand-written fixture in the shapes a Firedrake mesh hands out (cell_node_map, interior / exterior facet tables with FIAT
    local facet numbers, markers, a clockwise cell) is flattened to the swe2d_mesh arrays; they must be the arrays Mesh2d
    derives independently from vertices and cells (edge matching) - what every device test of this repository runs on."""
    import json
    import os
    from thetis_amd.firedrake_adapter import swe2d_mesh_arrays
    from thetis_amd.mesh import Mesh2d, _rect_marker_fn
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'firedrake_like_mesh.json')) as f:
        fx = json.load(f)
    out = swe2d_mesh_arrays(fx['coords'], fx['cell_vertices'], fx['int_facet_cell'], fx['int_local_facet'], fx['ext_facet_cell'],
                            fx['ext_local_facet'], fx['ext_markers'], dg_cell_nodes=fx['dg_cell_nodes'])
    assert np.array_equal(out['cell_vertices'], fx['expected_cell_vertices_ccw'])
    assert np.array_equal(out['dg_perm'], fx['expected_dg_perm'])
    mesh = Mesh2d(np.array(fx['coords']), np.array(fx['cell_vertices']), marker_fn=_rect_marker_fn(2.0, 1.0))
    assert np.array_equal(mesh.cells, out['cell_vertices'])
    assert np.array_equal(mesh.cell_nbr, out['cell_neighbours'])
    assert np.array_equal(mesh.cell_nbr_facet, out['cell_neighbour_facets'])
    # node i of cell c in the C ABI's cell-major layout is the vertex cell_vertices[c, i]: the DG permutation follows the flip
    dg_vertex = np.empty(12, dtype=int)
    dg_vertex[np.array(fx['dg_cell_nodes']).ravel()] = np.array(fx['cell_vertices']).ravel()      # DG dof -> vertex it sits on
    assert np.array_equal(dg_vertex[out['dg_perm']], out['cell_vertices'])
    # a facet that is in neither table, or an unmarked exterior facet, is an error, not a silent wall
    with pytest.raises(ValueError):
        swe2d_mesh_arrays(fx['coords'], fx['cell_vertices'], fx['int_facet_cell'][:2], fx['int_local_facet'][:2],
                          fx['ext_facet_cell'], fx['ext_local_facet'], fx['ext_markers'])
    bad = list(fx['ext_markers'])
    bad[0] = 0
    with pytest.raises(ValueError):
        swe2d_mesh_arrays(fx['coords'], fx['cell_vertices'], fx['int_facet_cell'], fx['int_local_facet'], fx['ext_facet_cell'],
                          fx['ext_local_facet'], bad)
    assert (swe2d_mesh_arrays(fx['coords'], fx['cell_vertices'], fx['int_facet_cell'], fx['int_local_facet'],
                              fx['ext_facet_cell'], fx['ext_local_facet'], bad, halo_marker=9)['cell_neighbours'] == -9).sum() == 1


def test_firedrake_shaped_quadrilateral_arrays_give_the_swe2d_mesh_of_mesh2d():
    """... and for FIAT tensor-product quadrilaterals (lexicographic, non-cyclic local vertices; facets x=0, x=1, y=0, y=1; one
    mirrored cell; tests/golden/make_firedrake_like_quad_mesh.py): cyclic counter-clockwise cells, the facet tables and the DQ-1
    permutation must be what Mesh2d derives from the cyclic cells by edge matching."""
    import json
    import os
    from thetis_amd.firedrake_adapter import swe2d_mesh_arrays
    from thetis_amd.mesh import Mesh2d, _rect_marker_fn
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'firedrake_like_quad_mesh.json')) as f:
        fx = json.load(f)
    out = swe2d_mesh_arrays(fx['coords'], fx['cell_vertices'], fx['int_facet_cell'], fx['int_local_facet'], fx['ext_facet_cell'],
                            fx['ext_local_facet'], fx['ext_markers'], dg_cell_nodes=fx['dg_cell_nodes'])
    assert np.array_equal(out['cell_vertices'], fx['expected_cell_vertices_ccw'])
    assert np.array_equal(out['dg_perm'], fx['expected_dg_perm'])
    mesh = Mesh2d(np.array(fx['coords'], dtype=float), np.array(fx['expected_cell_vertices_ccw']), marker_fn=_rect_marker_fn(2.0, 2.0))
    assert np.array_equal(mesh.cells, out['cell_vertices'])
    assert np.array_equal(mesh.cell_nbr, out['cell_neighbours'])
    assert np.array_equal(mesh.cell_nbr_facet, out['cell_neighbour_facets'])
    dg_vertex = np.empty(16, dtype=int)
    dg_vertex[np.array(fx['dg_cell_nodes']).ravel()] = np.array(fx['cell_vertices']).ravel()
    assert np.array_equal(dg_vertex[out['dg_perm']], out['cell_vertices'])
    with pytest.raises(NotImplementedError):
        swe2d_mesh_arrays(fx['coords'], [c + [0] for c in fx['cell_vertices']], [], [], [], [], [])


def _reference_surface():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_surface.json')) as f:
        return json.load(f)


def test_option_names_and_defaults_are_the_references():
    """Every option of this build's ``ModelOptions2d`` / explicit time stepper option classes exists in the reference under the
    same name with the same default (tests/golden/reference_surface.json: trait defaults read from thetis/options.py by AST,
    tests/golden/make_surface_golden.py) - except the ones this build adds, listed here."""
    from thetis_amd import options as o
    ref = _reference_surface()

    def ref_options(cls_name):
        out = {}
        c = ref['options'][cls_name]
        for b in c['bases']:
            if b in ref['options']:
                out.update(ref_options(b))
        out.update(c['options'])
        return out

    def value(v):
        if isinstance(v, o.Constant):
            vals = v.values()
            return vals[0] if len(vals) == 1 else list(vals)
        return v
    own = {'ModelOptions2d': {'wetting_and_drying_cfl_factor'}}
    checked = 0
    for cls_name, cls in (('ModelOptions2d', o.ModelOptions2d), ('ExplicitSWETimeStepperOptions2d', o.ExplicitSWETimeStepperOptions2d),
                          ('ExplicitTracerTimeStepperOptions2d', o.ExplicitTracerTimeStepperOptions2d)):
        theirs = ref_options(cls_name)
        obj = cls()
        for key in cls._all_spec():
            if key in own.get(cls_name, ()) or key in ref['paired']:
                continue
            assert key in theirs, '{:}.{:} is not an option of the reference'.format(cls_name, key)
            d = theirs[key]['default']
            if isinstance(d, dict) and 'expr' in d:
                continue                                   # not a literal in the reference (e.g. PETSc parameter dictionaries)
            assert value(getattr(obj, key)) == d, '{:}.{:}: {!r} != reference default {!r}'.format(cls_name, key, value(getattr(obj, key)), d)
            checked += 1
    assert checked >= 50
    # the steppers tables and their defaults (attach_paired_options of ModelOptions2d)
    m = o.ModelOptions2d()
    for name, table in (('swe_timestepper_type', o._SWE_STEPPERS), ('tracer_timestepper_type', o._TRACER_STEPPERS)):
        p = ref['paired'][name]
        assert [c[0] for c in p['choices']] == list(table) and getattr(m, name) == p['default']
        explicit = [c[0] for c in p['choices'] if c[1].startswith('Explicit')]
        assert explicit == ['SSPRK33', 'ForwardEuler'] and all(table[e].__name__ == dict(p['choices'])[e] for e in explicit)


def test_field_metadata_and_physical_constants_are_the_references():
    """thetis/field_defs.py executed, thetis/physical_constants.py read (tests/golden/make_surface_golden.py): export file names,
    units, g and rho_0 of this build are theirs"""
    from thetis_amd import exporter, shallowwater_eq
    ref = _reference_surface()
    for key, meta in exporter.field_metadata.items():
        assert ref['field_metadata'][key] == meta, key
    assert float(shallowwater_eq.g_grav) == ref['physical_constants']['g_grav'] == 9.81
    assert float(shallowwater_eq.rho_0) == ref['physical_constants']['rho0']


def test_flow_block_order_makes_compact_blocks():
    """ordering.flow_block_order: the 64-cell blocks of the dataflow kernel as 8 x 4-quad tiles aligned with the LOCAL extent of the
    cells - a permutation; 24 rim facets per interior block of a RectangleMesh (the kernel then polls with four granule loads per
    lane), a quarter fewer on the strips of a partitioned one, where the device numbering's two-row blocks have 36 and up to 68."""
    import numpy as np
    from thetis_amd import ordering, partition
    from thetis_amd.mesh import RectangleMesh

    def rims(cell_nbr, order):
        order = np.asarray(order)
        n = len(order)
        assert sorted(order.tolist()) == list(range(n))
        pos = np.empty(n, dtype=np.int64)
        pos[order] = np.arange(n)
        blk = pos//64
        rim = np.zeros((n + 63)//64, dtype=np.int64)
        nbr = np.asarray(cell_nbr)
        for f in range(nbr.shape[1]):
            ok = nbr[:, f] >= 0
            a, b = blk[np.nonzero(ok)[0]], blk[nbr[ok, f]]
            np.add.at(rim, a[a != b], 1)
        return rim

    mesh = RectangleMesh(96, 64, 96e3, 64e3)
    r_new, r_old = rims(mesh.cell_nbr, ordering.flow_block_order(mesh)), rims(mesh.cell_nbr, ordering.auto_cell_order(mesh))
    assert r_new.max() <= 24 and r_old.max() >= 36
    mesh = RectangleMesh(200, 60, 200e3, 60e3)
    owner = partition.strip_owner(mesh, 4)
    for rank in (0, 1, 3):
        p = partition.build_partition(mesh, owner, rank, halo_depth=6)
        r_new = rims(p.cell_nbr, ordering.flow_block_order(p, 0, p.num_cells))
        r_old = rims(p.cell_nbr, ordering.auto_cell_order(p, 0, p.num_cells))
        # (53 / 56 local quad columns: the partial tile column at the far edge lets the blocks behind it straddle two tiles)
        assert r_new.max() <= 40 < r_old.max() and r_new.mean() < 0.75*r_old.mean(), (rank, r_new.max(), r_old.max())
    # quadrilaterals: the device order (no flow kernel)
    mq = RectangleMesh(20, 10, 20e3, 10e3, quadrilateral=True)
    assert np.array_equal(ordering.flow_block_order(mq), ordering.auto_cell_order(mq))
    # any other triangulation: bisection boxes of exactly 64 cells
    from helpers import delaunay_case
    md = delaunay_case(n_points=6000, seed=5)[0]
    r_new, r_old = rims(md.cell_nbr, ordering.flow_block_order(md)), rims(md.cell_nbr, ordering.auto_cell_order(md))
    assert r_new.max() <= 42 and r_new.max() < r_old.max() and r_new.mean() < 0.9*r_old.mean(), (r_new.max(), r_old.max(), r_new.mean(), r_old.mean())


def test_triple_tile_order_gives_patches_that_fill_a_workgroup():
    """ordering.triple_tile_order: the two-ring tiles of the three-stage kernel (csrc/swe2d_fuse.h swe_fuse123_kernel) as patches of
    11 x 8 quads of a RectangleMesh - a permutation, every patch its own run of the order, interior + facet neighbours + their facet
    neighbours = 176 + 38 + 42 = 256 lanes for a patch inside the mesh and never more (what fuse123_build counts, restated here), also
    over the owned and ghost cells of a partition; ``None`` for quadrilaterals and for meshes that are no RectangleMesh."""
    import numpy as np
    from thetis_amd import ordering, partition
    from thetis_amd.mesh import RectangleMesh
    from helpers import delaunay_case

    def lanes(cell_nbr, cells):
        cells = np.asarray(cells)
        inside = set(cells.tolist())
        ring1 = set(int(c) for c in np.asarray(cell_nbr)[cells].ravel() if c >= 0) - inside
        ring2 = set(int(c) for c in np.asarray(cell_nbr)[np.array(sorted(ring1), dtype=np.int64)].ravel() if c >= 0) - inside - ring1
        return len(inside), len(ring1), len(ring2)

    mesh = RectangleMesh(60, 37, 60e3, 37e3)
    order, starts = ordering.triple_tile_order(mesh)
    assert sorted(order.tolist()) == list(range(mesh.num_cells)) and starts[0] == 0 and (np.diff(starts) > 0).all()
    assert len(starts) == (-(-60//11))*(-(-37//8))
    full = 0
    for a, b in zip(starts, list(starts[1:]) + [mesh.num_cells]):
        q = order[a:b]//2
        assert (q % 60).max() - (q % 60).min() < 11 and (q//60).max() - (q//60).min() < 8
        n = lanes(mesh.cell_nbr, order[a:b])
        assert sum(n) <= 256, n
        if n[0] == 176 and n[1] == 38:
            full += 1
            assert n == (176, 38, 42)
    assert full >= 6
    # a partition: patches of the PARENT mesh over all local cells
    owner = partition.strip_owner(mesh, 3)
    p = partition.build_partition(mesh, owner, 1, halo_depth=6)
    order, starts = ordering.triple_tile_order(p)
    assert sorted(order.tolist()) == list(range(p.num_cells))
    g = np.asarray(p.local_to_global)
    for a, b in zip(starts, list(starts[1:]) + [p.num_cells]):
        q = g[order[a:b]]//2
        assert (q % 60).max() - (q % 60).min() < 11 and (q//60).max() - (q//60).min() < 8
        assert sum(lanes(p.cell_nbr, order[a:b])) <= 256
    assert ordering.triple_tile_order(RectangleMesh(20, 10, 20e3, 10e3, quadrilateral=True)) is None
    assert ordering.triple_tile_order(delaunay_case(n_points=400, seed=5)[0]) is None


def test_compact_connectivity_records_round_trip(tmp_path):
    """thetis_amd/csrc/swe2d_conn.h: the 16-B connectivity records the triangle kernels read.  Host packing against the kernels'
    unpacking (the same header compiled with g++): random records of every span, markers, the edges of the 19-bit range, and that
    exactly the records that do not fit come back as escapes."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path/'conn.cpp'
    src.write_text(r'''
#include <cstdio>
#include <cstdlib>
#include "swe2d_conn.h"
static int check(int k, const int nb[3], const int vid[3], int want_escape) {
    int nb2[3], vid2[3];
    const bool e = swe_conn_unpack(swe_conn_pack(k, nb, vid), k, nb2, vid2);
    if (want_escape >= 0 && (int)e != want_escape) { printf("escape %d, expected %d at k %d\n", (int)e, want_escape, k); return -1; }
    if (!e) for (int i = 0; i < 3; i++) if (nb2[i] != nb[i] || vid2[i] != vid[i]) { printf("mismatch at k %d\n", k); return -1; }
    return e ? 1 : 0;
}
int main() {
    srand(1);
    long esc = 0, n = 0;
    for (int it = 0; it < 2000000; it++) {
        const int span = (it & 1) ? 300000 : 1 << (rand() % 24);
        const int k = rand() % (1 << 24);
        int nb[3], vid[3];
        vid[0] = rand() % (1 << 26);
        for (int i = 1; i < 3; i++) vid[i] = vid[0] + rand() % (2*span + 1) - span;
        if (vid[1] < 0 || vid[2] < 0) continue;
        bool fits = swe_conn_fits((long long)vid[1] - vid[0]) && swe_conn_fits((long long)vid[2] - vid[0]);
        for (int f = 0; f < 3; f++) {
            if (rand() % 8 == 0) nb[f] = -(1 + rand() % 15);
            else {
                long long kn = (long long)k + rand() % (2*span + 1) - span;
                if (kn < 0) kn = 0;
                nb[f] = (int)((kn << 2) | (rand() % 3));
                fits = fits && swe_conn_fits(kn - k);
            }
        }
        const int r = check(k, nb, vid, fits ? 0 : 1);
        if (r < 0) return 1;
        esc += r; n++;
    }
    const int lim = 1 << (SWE_CONN_DBITS - 1);
    for (int d = -lim - 2; d <= lim + 2; d++) {
        if (d > -lim + 2 && d < lim - 2 && d != 0 && d != 1 && d != -1) continue;
        const int k = 1000000;
        const int nb[3] = {((k + d) << 2) | 2, -15, ((k - d) << 2) | 0}, vid[3] = {5000000, 5000000 + d, 5000000 - d};
        if (check(k, nb, vid, (d >= -lim && d < lim && -d >= -lim && -d < lim) ? 0 : 1) < 0) return 1;
    }
    { const int nb[3] = {-1, -2, -3}, vid[3] = {(1 << 26), 5, 6}; if (check(0, nb, vid, 1) < 0) return 1; }      // vertex id too large
    { const int nb[3] = {-1, -2, 4}, vid[3] = {(1 << 26) - 1, (1 << 26) - 2, (1 << 26) - 3}; if (check(0, nb, vid, 0) < 0) return 1; }
    printf("ok %ld %ld\n", n, esc);
    return 0;
}
''')
    exe = tmp_path/'conn'
    subprocess.run(['g++', '-O1', '-w', '-I', os.path.join(root, 'thetis_amd', 'csrc'), str(src), '-o', str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith('ok '), out.stdout + out.stderr
    n, esc = (int(x) for x in out.stdout.split()[1:3])
    assert n > 1500000 and 0.1*n < esc < 0.6*n
