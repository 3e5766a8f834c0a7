"""Multi-rank path: partition, halo lists and exchange.  CPU: world_size 2-3 gloo with the oracle as compute;
GPU: the real DistributedSwe2d with two ranks sharing the one GPU of the test box."""
import numpy as np
import pytest

from dist_worker import _case, cpu_worker, gather, gpu_worker, run_workers
from helpers import make_ref, rel_linf
from thetis_amd.partition import build_partition, strip_owner


@pytest.mark.parametrize('world,axis', [(2, 0), (3, 0), (2, 1)])
def test_partition_invariants(world, axis):
    mesh, bath, uv, eta = _case()
    owner = strip_owner(mesh, world, axis=axis)
    counts = np.bincount(owner, minlength=world)
    assert counts.max() - counts.min() <= 1
    parts = [build_partition(mesh, owner, r) for r in range(world)]
    seen = np.concatenate([p.local_to_global[:p.n_owned] for p in parts])
    assert sorted(seen) == list(range(mesh.num_cells))           # every cell owned exactly once
    for p in parts:
        g = p.local_to_global
        assert len(p.layer_sizes) == 3 and sum(p.layer_sizes) == p.n_ghost
        assert [p.stage_range(i) for i in range(3)] == [p.n_owned + p.layer_sizes[0] + p.layer_sizes[1],
                                                        p.n_owned + p.layer_sizes[0], p.n_owned]
        # a cell updated by stage i only reads cells that stage i-1 updated (or the exchanged step input)
        for i in range(3):
            end = p.stage_range(i)
            valid_in = p.num_cells if i == 0 else p.stage_range(i - 1)
            nb = p.cell_nbr[:end]
            assert nb.max() < valid_in
        # interior cells are in nobody's halo; send cells are
        sent = np.zeros(p.num_cells, dtype=bool)
        sent[p.send_cells] = True
        assert not sent[:p.n_interior].any() and sent[p.n_interior:p.n_owned].all() and not sent[p.n_owned:].any()
        # local connectivity is the global one wherever a cell is updated
        for k in range(p.stage_range(0)):
            for f in range(3):
                gn = mesh.cell_nbr[g[k], f]
                assert (gn < 0 and p.cell_nbr[k, f] == gn) or g[p.cell_nbr[k, f]] == gn
        # what peer q sends is exactly what I expect to receive from q, in the same order
        for q, (off, cnt) in p.recv.items():
            soff, scnt = parts[q].send[p.rank]
            assert scnt == cnt
            sent_global = parts[q].local_to_global[parts[q].send_cells[soff:soff + scnt]]
            assert np.array_equal(sent_global, g[p.recv_cells[off:off + cnt]])
        assert sorted(p.recv_cells) == list(range(p.n_owned, p.num_cells))    # every ghost is received exactly once
        assert len(p.peers) <= 2                                  # strips: one xGMI link per side


@pytest.mark.parametrize('world,axis', [(2, 0), (3, 1)])
def test_gloo_partitioned_step_equals_global(tmp_path, ref_so, world, axis):
    mesh, bath, uv, eta = _case()
    run_workers(cpu_worker, world, 3, str(tmp_path), axis=axis)
    u_p, e_p, _ = gather(str(tmp_path), world, mesh.num_cells)
    u_g, e_g = make_ref(mesh, bath).advance(uv, eta, 2.0, 3)
    # same arithmetic per cell, ghost traces delivered exactly: bitwise equal
    assert np.array_equal(u_p, u_g) and np.array_equal(e_p, e_g)


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_match_single_device(tmp_path, hip_lib):
    from thetis_amd.device import Swe2dDevice
    mesh, bath, uv, eta = _case()
    run_workers(gpu_worker, 2, 3, str(tmp_path), axis=0)
    u_p, e_p, extra = gather(str(tmp_path), 2, mesh.num_cells)
    dev = Swe2dDevice(mesh, bath, 2.0)
    dev.set_state(uv, eta)
    dev.advance(3)
    u_s, e_s = dev.get_state()
    assert np.array_equal(u_p, u_s) and np.array_equal(e_p, e_s)       # deterministic kernel: bitwise
    d = dev.diagnostics()
    assert np.allclose(extra[0]['d1'][:3], d[:3], rtol=1e-13) and extra[0]['d1'][3] == d[3]
    dev.close()


@pytest.mark.gpu
def test_hip_graph_capture_single_rank(hip_lib):
    """The graph-captured step loop (stream plumbing through torch) gives the same state as eager launches."""
    import torch
    from thetis_amd.device import Swe2dDevice
    mesh, bath, uv, eta = _case()
    dev = Swe2dDevice(mesh, bath, 2.0)
    dev.set_state(uv, eta)
    dev.advance(4)
    ref_state = dev.get_state()
    s = torch.cuda.Stream()
    dev.set_stream(s.cuda_stream)
    dev.set_state(uv, eta)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        dev.advance(1)                       # warm-up on the stream
        s.synchronize()
        dev.set_state(uv, eta)
        with torch.cuda.graph(g, stream=s, capture_error_mode='thread_local'):
            dev.advance(2)
        g.replay()
        g.replay()
    s.synchronize()
    u, e = dev.get_state()
    assert np.array_equal(u, ref_state[0]) and np.array_equal(e, ref_state[1])
    dev.set_stream(None)
    dev.close()


def test_gloo_unstructured_partition_equals_global(tmp_path, ref_so):
    import dist_worker
    dist_worker.CASE = 'delaunay'
    try:
        mesh, bath, uv, eta = dist_worker._case()
        run_workers(cpu_worker, 3, 2, str(tmp_path), axis=0, case='delaunay')
        u_p, e_p, _ = gather(str(tmp_path), 3, mesh.num_cells)
    finally:
        dist_worker.CASE = 'channel'
    u_g, e_g = make_ref(mesh, bath).advance(uv, eta, 2.0, 2)
    assert np.array_equal(u_p, u_g) and np.array_equal(e_p, e_g)
