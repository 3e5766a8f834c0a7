"""Gmsh MSH 2.2 reader/writer; real unstructured mesh of the reference when it is available (build container only)."""
import os

import numpy as np
import pytest

from helpers import delaunay_case, make_oracle_generic, make_ref, rel_linf
from thetis_amd.meshio import read_gmsh, write_gmsh

NORTH_SEA = '/root/reference/demos/north_sea.msh'


def test_gmsh_roundtrip(tmp_path):
    mesh, bath, uv, eta = delaunay_case(n_points=150)
    path = str(tmp_path/'m.msh')
    write_gmsh(mesh, path)
    m2 = read_gmsh(path)
    assert np.array_equal(m2.cells, mesh.cells) and np.allclose(m2.vertex_xy, mesh.vertex_xy, rtol=0, atol=0)
    assert np.array_equal(m2.cell_nbr, mesh.cell_nbr) and m2.boundary_len == pytest.approx(mesh.boundary_len)
    from thetis_amd import RectangleMesh
    q = RectangleMesh(5, 4, 2.0, 1.0, quadrilateral=True)
    write_gmsh(q, path)
    q2 = read_gmsh(path)
    assert q2.nodes_per_cell == 4 and np.array_equal(q2.cell_nbr, q.cell_nbr)
    with open(path, 'w') as f:
        f.write('$MeshFormat\n4.1 0 8\n$EndMeshFormat\n$Nodes\n0\n$EndNodes\n$Elements\n0\n$EndElements\n')
    with pytest.raises(NotImplementedError):
        read_gmsh(path)


@pytest.mark.skipif(not os.path.exists(NORTH_SEA), reason='reference mesh only exists in the build container')
def test_reference_north_sea_mesh_oracles_agree(ref_so):
    """demos/north_sea.msh (10,920 triangles, physical ids 100 and 200): the two CPU restatements on a real coastal mesh."""
    mesh = read_gmsh(NORTH_SEA)
    assert mesh.num_cells == 10920 and mesh.boundary_markers == [100, 200]
    x, y = mesh.vertex_xy.T
    bath = 40.0 + 20.0*np.sin(x/3e5)*np.cos(y/2e5)
    rng = np.random.default_rng(0)
    uv = 0.3*rng.normal(size=(mesh.num_cells, 3, 2))
    eta = 0.3*rng.normal(size=(mesh.num_cells, 3))
    bcs = {100: {'elev': 0.2}}
    orc = make_oracle_generic(mesh, bath, bnd_conditions=bcs, manning_drag_coefficient=0.02)
    ref = make_ref(mesh, bath, bnd_conditions=bcs, manning_drag_coefficient=0.02)
    ku, ke = orc.tendency(uv, eta, 1.0)
    ku2, ke2 = ref.tendency(uv, eta, 1.0)
    assert rel_linf(ku2, ku) < 1e-12 and rel_linf(ke2, ke) < 1e-12


@pytest.mark.gpu
def test_gmsh_mesh_with_large_marker_ids_on_gpu(hip_lib, tmp_path):
    """Arbitrary physical ids (as in the reference's .msh files) are mapped to the C ABI's marker slots transparently."""
    from thetis_amd.device import Swe2dDevice
    mesh, bath, uv, eta = delaunay_case(n_points=800, seed=4)
    mesh.cell_nbr[mesh.cell_nbr == -1] = -100
    mesh.cell_nbr[mesh.cell_nbr == -2] = -200
    mesh.cell_nbr[mesh.cell_nbr == -3] = -1000
    mesh.boundary_len = mesh._boundary_length()
    path = str(tmp_path/'m.msh')
    write_gmsh(mesh, path)
    m2 = read_gmsh(path)
    assert m2.boundary_markers == [4, 100, 200, 1000]
    bcs = {100: {'elev': 0.2}, 200: {'un': 0.1}, 1000: {'flux': 50.0}}
    orc = make_oracle_generic(m2, bath, bnd_conditions=bcs)
    dev = Swe2dDevice(m2, bath, 0.1, boundary_len=m2.boundary_len)
    for m, funcs in bcs.items():
        dev.set_bc(m, funcs)
    dev.set_state(uv, eta)
    ku, ke = dev.tendency()
    ku_o, ke_o = orc.tendency(uv, eta, 0.1)
    assert rel_linf(ku, ku_o) < 1e-12 and rel_linf(ke, ke_o) < 1e-12
    with pytest.raises(KeyError):
        dev.set_bc(7, {'elev': 0.0})
    dev.close()
