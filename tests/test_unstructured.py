"""Unstructured (Delaunay) meshes: the two CPU restatements against each other, and the HIP path against them."""
import numpy as np
import pytest

from helpers import delaunay_case, make_oracle, make_ref, rel_linf


def test_unstructured_numpy_vs_c(ref_so):
    mesh, bath, uv, eta = delaunay_case()
    assert set(mesh.boundary_markers) == {1, 2, 3, 4}
    bcs = {1: {'elev': 0.2}, 2: {'un': 0.1}}
    orc = make_oracle(mesh, bath, bnd_conditions=bcs, manning_drag_coefficient=0.02)
    ref = make_ref(mesh, bath, bnd_conditions=bcs, manning_drag_coefficient=0.02)
    ku, ke = orc.tendency(uv, np.abs(eta), 0.5)
    ku2, ke2 = ref.tendency(uv, np.abs(eta), 0.5)
    assert rel_linf(ku2, ku) < 1e-13 and rel_linf(ke2, ke) < 1e-13


@pytest.mark.gpu
@pytest.mark.parametrize('reorder', ['auto', None])
def test_unstructured_gpu_parity(hip_lib, ref_so, reorder):
    from oracle.ref_lib import RefTracer
    from thetis_amd import _lib
    from thetis_amd.device import Swe2dDevice
    mesh, bath, uv, eta = delaunay_case(n_points=3000, seed=2)
    eta = np.abs(eta)
    bcs = {1: {'elev': 0.2}, 2: {'un': 0.1}, 4: {'flux': 50.0}}
    # CFL-limited step: Delaunay meshes contain small and thin cells (inradius ~ 2A/perimeter)
    p = mesh.cell_xy()
    per = sum(np.hypot(*(p[:, (i + 1) % 3] - p[:, i]).T) for i in range(3))
    dt = 0.05*float((2*mesh.cell_areas()/per).min())/(np.sqrt(9.81*20.0) + 1.0)
    orc = make_oracle(mesh, bath, bnd_conditions=bcs, manning_drag_coefficient=0.02)
    ref = make_ref(mesh, bath, bnd_conditions=bcs, manning_drag_coefficient=0.02)
    dev = Swe2dDevice(mesh, bath, dt, reorder=reorder)
    for m, funcs in bcs.items():
        dev.set_bc(m, funcs)
    dev.set_scalar(_lib.SCALAR_MANNING_DRAG, 0.02)
    dev.set_state(uv, eta)
    ku, ke = dev.tendency()
    ku_o, ke_o = orc.tendency(uv, eta, dt)
    assert rel_linf(ku, ku_o) < 1e-12 and rel_linf(ke, ke_o) < 1e-12
    dev.advance(20)
    u_d, e_d = dev.get_state()
    u_r, e_r = ref.advance(uv, eta, dt, 20)
    assert np.isfinite(u_r).all()
    assert rel_linf(u_d, u_r) < 1e-11 and rel_linf(e_d, e_r) < 1e-11
    # tracer + limiter on the unstructured mesh
    T = np.random.default_rng(5).normal(size=(mesh.num_cells, 3))
    tid = dev.add_tracer()
    dev.tracer_set_state(tid, T)
    rt = RefTracer(ref, cell_topo_vertices=mesh.topo_vertex[mesh.cells])
    assert rel_linf(dev.tracer_tendency(tid), rt.tendency(T, u_d, dt)) < 1e-12
    dev.tracer_limit(tid)
    assert rel_linf(dev.tracer_get_state(tid), rt.limit(T)) < 1e-14
    dev.close()


@pytest.mark.gpu
def test_unstructured_partition_two_ranks_on_one_gpu(tmp_path, hip_lib):
    """Strips of an unstructured mesh (irregular cut, more than one ghost per boundary cell)."""
    import dist_worker
    from thetis_amd.device import Swe2dDevice
    dist_worker.CASE = 'delaunay'
    try:
        mesh, bath, uv, eta = dist_worker._case()
        dist_worker.run_workers(dist_worker.gpu_worker, 2, 3, str(tmp_path), axis=0, case='delaunay')
        u_p, e_p, _ = dist_worker.gather(str(tmp_path), 2, mesh.num_cells)
    finally:
        dist_worker.CASE = 'channel'
    dev = Swe2dDevice(mesh, bath, 2.0)
    dev.set_state(uv, eta)
    dev.advance(3)
    u_s, e_s = dev.get_state()
    assert np.array_equal(u_p, u_s) and np.array_equal(e_p, e_s)
    dev.close()
