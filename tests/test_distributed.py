"""Multi-rank path: partition, halo lists and exchange.  CPU: world_size 2-3 gloo with the oracle as compute;
GPU: the real DistributedSwe2d with two ranks sharing the one GPU of the test box."""
import os

import numpy as np
import pytest

from dist_worker import (_case, coupled_step_reference, cpu_coupled_cycles_worker, cpu_coupled_worker, cpu_worker, gather,
                         gpu_coupled_worker,
                         gpu_worker, run_workers, tracer_initial)
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


@pytest.mark.parametrize('world,axis', [(2, 0), (3, 1), (8, 0)])
def test_gloo_partitioned_step_equals_global(tmp_path, ref_so, world, axis):
    mesh, bath, uv, eta = _case()
    run_workers(cpu_worker, world, 3, str(tmp_path), axis=axis)
    u_p, e_p, _ = gather(str(tmp_path), world, mesh.num_cells)
    u_g, e_g = make_ref(mesh, bath).advance(uv, eta, 2.0, 3)
    # same arithmetic per cell, ghost traces delivered exactly: bitwise equal
    assert np.array_equal(u_p, u_g) and np.array_equal(e_p, e_g)


@pytest.mark.parametrize('world,axis,every,n_steps', [(2, 0, 2, 5), (3, 1, 3, 4), (2, 1, 4, 8), (8, 0, 2, 5)])
def test_gloo_exchange_every_m_steps_equals_global(tmp_path, ref_so, world, axis, every, n_steps):
    """3m ghost layers, one exchange per m steps (the last cycle may be shorter): stale layers are NaN-poisoned in the
    worker, the result is bitwise the global one."""
    mesh, bath, uv, eta = _case()
    run_workers(cpu_worker, world, n_steps, str(tmp_path), axis=axis, case='channel+every{:d}'.format(every))
    u_p, e_p, extra = gather(str(tmp_path), world, mesh.num_cells)
    u_g, e_g = make_ref(mesh, bath).advance(uv, eta, 2.0, n_steps)
    assert np.array_equal(u_p, u_g) and np.array_equal(e_p, e_g)


@pytest.mark.parametrize('world,axis,every,overlap,n_steps', [(2, 0, 1, 2, 4), (2, 0, 2, 3, 7), (3, 1, 2, 5, 6), (2, 1, 4, 3, 9),
                                                              (3, 0, 1, 1, 3)])
def test_gloo_overlapped_exchange_equals_global(tmp_path, ref_so, world, axis, every, overlap, n_steps):
    """The exchange stays in flight while the next cycle's first ``overlap`` stages run on the cells that cannot see ghost
    data yet; the worker mirrors the launch sequence on three rotating buffers with NaN for everything stale."""
    mesh, bath, uv, eta = _case()
    run_workers(cpu_worker, world, n_steps, str(tmp_path), axis=axis,
                case='channel+every{:d}+overlap{:d}'.format(every, overlap))
    u_p, e_p, extra = gather(str(tmp_path), world, mesh.num_cells)
    u_g, e_g = make_ref(mesh, bath).advance(uv, eta, 2.0, n_steps)
    assert np.array_equal(u_p, u_g) and np.array_equal(e_p, e_g)


@pytest.mark.parametrize('world,axis,every,n_steps', [(2, 0, 1, 4), (3, 1, 3, 7), (2, 1, 4, 9)])
def test_gloo_forward_euler_on_partitions_equals_global(tmp_path, ref_so, world, axis, every, n_steps):
    """ForwardEuler (timeintegrator.py:115-165) on partitions: one ghost layer per step, one exchange per m steps."""
    mesh, bath, uv, eta = _case()
    run_workers(cpu_worker, world, n_steps, str(tmp_path), axis=axis, case='channel+every{:d}+fe'.format(every))
    u_p, e_p, extra = gather(str(tmp_path), world, mesh.num_cells)
    ref = make_ref(mesh, bath)
    u_g, e_g = uv.copy(), eta.copy()
    for _ in range(n_steps):
        ku, ke = ref.tendency(u_g, e_g, 2.0)
        u_g, e_g = u_g + ku, e_g + ke
    assert np.array_equal(u_p, u_g) and np.array_equal(e_p, e_g)


@pytest.mark.gpu
@pytest.mark.parametrize('every,n_steps', [(1, 3), (3, 7)])
def test_two_ranks_forward_euler_on_one_gpu(tmp_path, hip_lib, every, n_steps):
    from thetis_amd.device import Swe2dDevice
    mesh, bath, uv, eta = _case()
    run_workers(gpu_worker, 2, n_steps, str(tmp_path), axis=0, case='channel+every{:d}+fe'.format(every))
    u_p, e_p, extra = gather(str(tmp_path), 2, mesh.num_cells)
    dev = Swe2dDevice(mesh, bath, 2.0)
    dev.set_state(uv, eta)
    dev.advance_forward_euler(n_steps)
    u_s, e_s = dev.get_state()
    assert np.array_equal(u_p, u_s) and np.array_equal(e_p, e_s)
    dev.close()


def test_owned_cells_are_sorted_by_distance_from_the_cut():
    from thetis_amd.partition import build_partition, rcb_owner
    mesh, bath, uv, eta = _case()
    for owner, depth in ((strip_owner(mesh, 2), 6), (rcb_owner(mesh, 4), 3)):
        for rank in range(int(owner.max()) + 1):
            p = build_partition(mesh, owner, rank, halo_depth=depth)
            g = p.local_to_global
            mine = np.zeros(mesh.num_cells, dtype=bool)
            mine[g[:p.n_owned]] = True
            # brute-force facet distance of every owned cell from the non-owned cells
            dist = np.where(mine, -1, 0)
            frontier = np.nonzero(~mine)[0]
            d = 0
            while len(frontier):
                d += 1
                nb = mesh.cell_nbr[frontier].ravel()
                nb = nb[nb >= 0]
                new = np.unique(nb[dist[nb] < 0])
                dist[new] = d
                frontier = new
            dl = dist[g[:p.n_owned]]
            dl = np.where(dl < 0, 10**6, dl)                      # unreachable (single part): interior
            for dd in range(0, depth + 2):
                n = p.owned_prefix(dd)
                assert (dl[:n] >= dd).all() and (dl[n:] < dd).all(), (rank, dd)
            assert p.owned_prefix(1) == p.n_owned and p.owned_prefix(depth + 1) == p.n_interior


def test_state_digest_does_not_depend_on_the_halo_depth():
    """``state_digest`` compares runs of DIFFERENT exchange schedules (first contact: a dataflow candidate with one exchange per step
    against the host-staged reference with one per two steps): a rank numbers its send cells last and which cells those are depends on
    the halo depth, so the fingerprint must take the owned cells in the order of their global ids.  (Found by ``tools.first_contact``
    at the end of round 6: the candidate was dropped as "does not reproduce the host-staged exchange" with identical states.)"""
    from thetis_amd.distributed import state_digest
    from helpers import channel_case
    mesh, bath, uv, eta = channel_case(nx=24, ny=6, seed=3, amp_eta=0.3, amp_u=0.2)
    owner = strip_owner(mesh, 2)

    class Stub(object):
        def __init__(self, part):
            self.part = part

        def get_state_owned(self):
            g = self.part.local_to_global[:self.part.n_owned]
            return g, uv[g], eta[g]
    parts = [build_partition(mesh, owner, 0, halo_depth=d) for d in (3, 6, 12)]
    assert any(not np.array_equal(parts[0].local_to_global[:parts[0].n_owned], q.local_to_global[:q.n_owned]) for q in parts[1:])
    assert len({state_digest(Stub(q)) for q in parts}) == 1
    eta2 = eta.copy()
    eta2[parts[0].local_to_global[0], 0] += 1e-13
    one = state_digest(Stub(parts[0]))
    eta[:] = eta2
    assert state_digest(Stub(parts[0])) != one


@pytest.mark.gpu
@pytest.mark.parametrize('every,overlap,n_steps,graph', [(1, 2, 4, ''), (2, 3, 7, ''), (4, 3, 8, ''), (2, 3, 8, '+graph'),
                                                         (4, 0, 16, '+graph'), (1, 0, 6, '+graph')])
def test_two_ranks_overlapped_exchange_on_one_gpu(tmp_path, hip_lib, every, overlap, n_steps, graph):
    """``+graph``: the kernel sequences before / during an exchange run from HIP graphs (graph_mode 'cycle'), the exchange
    (gloo + host staging here, RCCL on a multi-GPU node) stays eager between two graph launches."""
    from thetis_amd.device import Swe2dDevice
    mesh, bath, uv, eta = _case()
    run_workers(gpu_worker, 2, n_steps, str(tmp_path), axis=0,
                case='channel+every{:d}+overlap{:d}{:}'.format(every, overlap, graph))
    u_p, e_p, extra = gather(str(tmp_path), 2, mesh.num_cells)
    dev = Swe2dDevice(mesh, bath, 2.0)
    dev.set_state(uv, eta)
    dev.advance(n_steps)
    u_s, e_s = dev.get_state()
    assert np.array_equal(u_p, u_s) and np.array_equal(e_p, e_s)
    dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize('world,case,n_steps', [(4, 'channel+every2+overlap3+graph', 9), (3, 'channel+every4+overlap0+graph', 8)])
def test_three_and_four_ranks_on_one_gpu(tmp_path, hip_lib, world, case, n_steps):
    """Strips with two peers per interior rank (what an 8-GPU run looks like to a rank), per-cycle graphs, overlap."""
    from thetis_amd.device import Swe2dDevice
    mesh, bath, uv, eta = _case()
    run_workers(gpu_worker, world, n_steps, str(tmp_path), axis=0, case=case)
    u_p, e_p, extra = gather(str(tmp_path), world, mesh.num_cells)
    dev = Swe2dDevice(mesh, bath, 2.0)
    dev.set_state(uv, eta)
    dev.advance(n_steps)
    u_s, e_s = dev.get_state()
    assert np.array_equal(u_p, u_s) and np.array_equal(e_p, e_s)
    dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize('every,n_steps', [(2, 5), (4, 4)])
def test_two_ranks_exchange_every_m_steps_on_one_gpu(tmp_path, hip_lib, every, n_steps):
    from thetis_amd.device import Swe2dDevice
    mesh, bath, uv, eta = _case()
    run_workers(gpu_worker, 2, n_steps, str(tmp_path), axis=0, case='channel+every{:d}'.format(every))
    u_p, e_p, extra = gather(str(tmp_path), 2, mesh.num_cells)
    dev = Swe2dDevice(mesh, bath, 2.0)
    dev.set_state(uv, eta)
    dev.advance(n_steps)
    u_s, e_s = dev.get_state()
    assert np.array_equal(u_p, u_s) and np.array_equal(e_p, e_s)
    dev.close()


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


@pytest.mark.parametrize('world,axis', [(2, 0), (3, 1)])
def test_vertex_halo_partition_invariants(world, axis):
    """halo_depth=4, adjacency='vertex' (coupled runs with the vertex-based limiter): every cell around a vertex of a cell
    in owned + layers 1-3 is local, and the facet-stencil invariants of the 3-layer ranges still hold."""
    mesh, bath, uv, eta = _case()
    owner = strip_owner(mesh, world, axis=axis)
    parts = [build_partition(mesh, owner, r, halo_depth=4, adjacency='vertex') for r in range(world)]
    v2c = [[] for _ in range(mesh.num_vertices)]
    for c, vs in enumerate(mesh.cells):
        for v in vs:
            v2c[v].append(c)
    for p in parts:
        g = p.local_to_global
        assert len(p.layer_sizes) == 4 and sum(p.layer_sizes) == p.n_ghost
        local = set(g.tolist())
        for k in range(p.layer_end(3)):
            for v in mesh.cells[g[k]]:
                assert set(v2c[v]) <= local
        for i in range(3):
            end = p.stage_range(i)
            valid_in = p.layer_end(3) if i == 0 else p.stage_range(i - 1)
            assert p.cell_nbr[:end].max() < valid_in
        for q, (off, cnt) in p.recv.items():
            soff, scnt = parts[q].send[p.rank]
            assert scnt == cnt
            assert np.array_equal(parts[q].local_to_global[parts[q].send_cells[soff:soff + scnt]], g[p.recv_cells[off:off + cnt]])
        assert sorted(p.recv_cells) == list(range(p.n_owned, p.num_cells))


@pytest.mark.parametrize('world,axis', [(2, 0), (3, 1)])
def test_gloo_partitioned_coupled_step_equals_global(tmp_path, ref_so, world, axis):
    """SWE + tracer + vertex limiter on partitions == the same algorithm on the whole mesh, bitwise."""
    from oracle.ref_lib import RefTracer
    mesh, bath, uv, eta = _case()
    n_steps = 3
    run_workers(cpu_coupled_worker, world, n_steps, str(tmp_path), axis=axis)
    u_p, e_p, extra = gather(str(tmp_path), world, mesh.num_cells)
    T_p = extra[-1]
    ref = make_ref(mesh, bath)
    rt = RefTracer(ref, cell_topo_vertices=mesh.topo_vertex[mesh.cells])
    n = mesh.num_cells
    u, e, T = uv.copy(), eta.copy(), tracer_initial(mesh)
    for _ in range(n_steps):
        u, e, T = coupled_step_reference(ref, rt, u, e, T, 2.0, (n, n, n), n)
    assert np.array_equal(u_p, u) and np.array_equal(e_p, e) and np.array_equal(T_p, T)
    # the limiter was active (otherwise the test would not see a wrong limiter halo)
    rt_step = rt.step(tracer_initial(mesh), uv, 2.0)
    assert np.abs(rt.limit(rt_step) - rt_step).max() > 1e-3


@pytest.mark.parametrize('world,axis,case,n_steps', [(2, 0, 'channel+every2', 5), (3, 1, 'channel+every1', 3),
                                                      (2, 1, 'channel+every3+nolim', 4), (3, 0, 'delaunay+every2', 3),
                                                      (2, 0, 'channel+every3+fe', 7), (3, 1, 'channel+every2+fe+nolim', 5)])
def test_gloo_coupled_cycles_with_one_exchange_equal_global(tmp_path, ref_so, world, axis, case, n_steps):
    """m coupled steps between two exchanges of all fields on 4m + 3 vertex layers (3m + 3 facet layers without the limiter),
    every launch on the range coupled_cycle_schedule gives it (stale cells poisoned with NaN) == the whole-mesh algorithm,
    bitwise; odd step counts end with a shorter cycle."""
    import dist_worker
    from oracle.ref_lib import RefTracer
    dist_worker.CASE = case.split('+')[0]
    try:
        mesh, bath, uv, eta = _case()
    finally:
        dist_worker.CASE = 'channel'
    run_workers(cpu_coupled_cycles_worker, world, n_steps, str(tmp_path), axis=axis, case=case)
    u_p, e_p, extra = gather(str(tmp_path), world, mesh.num_cells)
    T_p = extra[-1]
    ref = make_ref(mesh, bath)
    rt = RefTracer(ref, cell_topo_vertices=mesh.topo_vertex[mesh.cells])
    n = mesh.num_cells
    u, e, T = uv.copy(), eta.copy(), tracer_initial(mesh)
    for _ in range(n_steps):
        if '+fe' in case:                                   # ForwardEuler: U + dt M^-1 R(U), tracer with the updated velocity
            ku, ke = ref.tendency(u, e, 2.0)
            u, e = u + ku, e + ke
            T = T + rt.tendency(T, u, 2.0)
            if '+nolim' not in case:
                T = rt.limit(T)
        else:
            u, e, T = coupled_step_reference(ref, rt, u, e, T, 2.0, (n, n, n), 0 if '+nolim' in case else n)
    assert np.array_equal(u_p, u) and np.array_equal(e_p, e) and np.array_equal(T_p, T)


def test_coupled_cycle_schedule_counts_layers():
    from thetis_amd.distributed import coupled_cycle_schedule, coupled_halo_depth
    from thetis_amd.partition import build_partition, strip_owner
    mesh, bath, uv, eta = _case()
    assert coupled_halo_depth(1, True) == 7 and coupled_halo_depth(2, True) == 11 and coupled_halo_depth(2, False) == 9
    assert coupled_halo_depth(3, True, 1) == 7 and coupled_halo_depth(3, False, 1) == 4
    part = build_partition(mesh, strip_owner(mesh, 2), 0, halo_depth=7, adjacency='vertex')
    ops = coupled_cycle_schedule(part, 1, 2, True)
    assert [o[0] for o in ops] == ['swe']*3 + ['swe_done'] + ['tracer']*6 + ['limit']*2
    assert [o[-1] for o in ops[:3]] == [part.layer_end(6), part.layer_end(5), part.layer_end(4)]
    assert [o[-1] for o in ops[4:7]] == [part.layer_end(3), part.layer_end(2), part.layer_end(1)]
    assert ops[-1] == ('limit', 1, part.n_owned)
    with pytest.raises(ValueError):
        coupled_cycle_schedule(part, 2, 1, True)


@pytest.mark.gpu
@pytest.mark.parametrize('case,n_steps', [('channel+every2', 5), ('channel+combined', 3), ('channel+every2+p2p', 4),
                                          ('channel+every3+nolim+p2p', 4), ('channel+every2+fe', 5), ('channel+every1+fe+p2p', 3),
                                          ('channel+every2+overlap3', 7), ('channel+every1+overlap2+combined+p2p', 4),
                                          ('channel+every3+overlap3+nolim+p2p', 8), ('channel+every2+p2p+step3', 5), ('channel+every4+step3', 9),
                                          ('channel+every3+nolim+p2p+step3', 7), ('channel+every2+p2p+step3+graph', 36)])
def test_two_ranks_coupled_cycles_with_one_exchange_match_single_device(tmp_path, hip_lib, monkeypatch, case, n_steps):
    """DistributedSwe2d(n_tracers=1, exchange_every=m | combined_exchange): one exchange of all fields per m coupled steps,
    host-staged and peer-to-peer, == the single-device coupled stepping, bitwise.  ``+overlapJ``: the first J shallow water
    stages of the next cycle run while the tracer's exchange is in flight (overlap_stages on coupled runs).  ``+step3``: the shallow-water
    steps of a cycle as one launch each, in pairs (swe2d_solve_step_cells; the state buffers change places under the tracer stages), also
    replayed from HIP graphs."""
    from thetis_amd.device import Swe2dDevice
    mesh, bath, uv, eta = _case()
    if '+step3' in case:
        monkeypatch.setenv('THETIS_AMD_FUSE12', '3')
    run_workers(gpu_coupled_worker, 2, n_steps, str(tmp_path), axis=0, case=case)
    monkeypatch.setenv('THETIS_AMD_FUSE12', '0')
    u_p, e_p, extra = gather(str(tmp_path), 2, mesh.num_cells)
    T_p = extra[-1]
    dev = Swe2dDevice(mesh, bath, 2.0)
    tid = dev.add_tracer()
    dev.set_state(uv, eta)
    dev.tracer_set_state(tid, tracer_initial(mesh))
    if '+fe' in case:
        for _ in range(n_steps):
            dev.advance_forward_euler(1)
            dev.tracer_forward_euler(tid)
            if '+nolim' not in case:
                dev.tracer_limit(tid)
    else:
        dev.advance_coupled(n_steps, tracer_only=False, use_limiter='+nolim' not in case)
    u_s, e_s = dev.get_state()
    assert np.array_equal(u_p, u_s) and np.array_equal(e_p, e_s) and np.array_equal(T_p, dev.tracer_get_state(tid))
    dev.close()


@pytest.mark.gpu
def test_two_ranks_coupled_on_one_gpu_match_single_device(tmp_path, hip_lib):
    from thetis_amd.device import Swe2dDevice
    mesh, bath, uv, eta = _case()
    run_workers(gpu_coupled_worker, 2, 3, str(tmp_path), axis=0)
    u_p, e_p, extra = gather(str(tmp_path), 2, mesh.num_cells)
    T_p = extra[-1]
    dev = Swe2dDevice(mesh, bath, 2.0)
    tid = dev.add_tracer()
    dev.set_state(uv, eta)
    dev.tracer_set_state(tid, tracer_initial(mesh))
    dev.advance_coupled(3, tracer_only=False, use_limiter=True)
    u_s, e_s = dev.get_state()
    T_s = dev.tracer_get_state(tid)
    assert np.array_equal(u_p, u_s) and np.array_equal(e_p, e_s)
    assert np.array_equal(T_p, T_s)                                  # deterministic kernels, exact halo data: bitwise
    d = dev.tracer_diagnostics(tid)
    assert np.allclose(extra[0]['td'][:2], d[:2], rtol=1e-13) and extra[0]['td'][2] == d[2] and extra[0]['td'][3] == d[3]
    dev.close()


def test_rcb_partition_invariants_and_halo_consistency():
    """Recursive coordinate bisection for general meshes: balanced, compact parts; more than two peers per
    rank (corner neighbours); the 3-layer halo bookkeeping holds for any owner array."""
    from thetis_amd.partition import rcb_owner
    import dist_worker
    dist_worker.CASE = 'delaunay'
    try:
        mesh, bath, uv, eta = dist_worker._case()
    finally:
        dist_worker.CASE = 'channel'
    for world in (4, 6):
        owner = rcb_owner(mesh, world)
        counts = np.bincount(owner, minlength=world)
        assert counts.min() > 0 and counts.max() - counts.min() <= world
        parts = [build_partition(mesh, owner, r) for r in range(world)]
        assert sorted(np.concatenate([p.local_to_global[:p.n_owned] for p in parts])) == list(range(mesh.num_cells))
        assert max(len(p.peers) for p in parts) > 2
        for p in parts:
            g = p.local_to_global
            for i in range(3):
                end = p.stage_range(i)
                valid_in = p.num_cells if i == 0 else p.stage_range(i - 1)
                assert p.cell_nbr[:end].max() < valid_in
            for q, (off, cnt) in p.recv.items():
                soff, scnt = parts[q].send[p.rank]
                assert scnt == cnt
                assert np.array_equal(parts[q].local_to_global[parts[q].send_cells[soff:soff + scnt]],
                                      g[p.recv_cells[off:off + cnt]])
            assert sorted(p.recv_cells) == list(range(p.n_owned, p.num_cells))


def test_gloo_rcb_partition_equals_global(tmp_path, ref_so):
    """4 ranks, RCB owner array on the unstructured mesh: bitwise the global result."""
    import dist_worker
    dist_worker.CASE = 'delaunay'
    try:
        mesh, bath, uv, eta = dist_worker._case()
        run_workers(cpu_worker, 4, 2, str(tmp_path), axis=-1, case='delaunay')
        u_p, e_p, _ = gather(str(tmp_path), 4, mesh.num_cells)
    finally:
        dist_worker.CASE = 'channel'
    u_g, e_g = make_ref(mesh, bath).advance(uv, eta, 2.0, 2)
    assert np.array_equal(u_p, u_g) and np.array_equal(e_p, e_g)


@pytest.mark.gpu
def test_two_ranks_with_viscosity_match_single_device(tmp_path, hip_lib):
    """The SIPG viscosity pass on partitions (range launches, neighbour gradients across the halo): bitwise the single-
    device result."""
    from dist_worker import viscosity_field
    from thetis_amd.device import Swe2dDevice
    mesh, bath, uv, eta = _case()
    run_workers(gpu_worker, 2, 3, str(tmp_path), axis=0, case='channel+visc')
    u_p, e_p, _ = gather(str(tmp_path), 2, mesh.num_cells)
    dev = Swe2dDevice(mesh, bath, 2.0)
    dev.set_viscosity(viscosity_field(mesh), use_grad_div_viscosity_term=True)
    dev.set_state(uv, eta)
    dev.advance(3)
    u_s, e_s = dev.get_state()
    assert np.array_equal(u_p, u_s) and np.array_equal(e_p, e_s)
    dev.set_viscosity(None)
    dev.set_state(uv, eta)
    dev.advance(3)
    assert not np.array_equal(dev.get_state()[0], u_s)           # the viscous term was active
    dev.close()


def _single_device(n_steps):
    from thetis_amd.device import Swe2dDevice
    mesh, bath, uv, eta = _case()
    dev = Swe2dDevice(mesh, bath, 2.0)
    dev.set_state(uv, eta)
    dev.advance(n_steps)
    out = dev.get_state()
    dev.close()
    return mesh, out


@pytest.mark.gpu
@pytest.mark.parametrize('world,case,n_steps', [
    (2, 'channel+p2p', 3), (2, 'channel+every2+p2p', 5), (2, 'channel+every4+overlap3+p2p+graph', 16),
    (2, 'channel+every2+p2p+nosplit+graph', 9), (3, 'channel+every4+p2p+graph', 8), (4, 'channel+every2+overlap3+p2p+graph', 9),
    (2, 'quad+p2p', 3), (3, 'delaunay+p2p+graph', 2), (2, 'channel+every3+fe+p2p', 7), (2, 'channel+every2+p2p+verify', 20)])
def test_ranks_on_one_gpu_with_peer_to_peer_halos(tmp_path, hip_lib, world, case, n_steps):
    """The exchange as two kernels writing into / polling IPC-mapped landing zones (csrc/swe2d_p2p.h): separate processes on
    one GPU map each other's zones with hipIpcOpenMemHandle exactly as ranks on different GPUs do; with '+graph' the whole
    cycle incl. the exchange kernels is replayed from one HIP graph.  Bitwise the single-device result."""
    from thetis_amd.device import Swe2dDevice
    import dist_worker
    base = case.split('+')[0]
    dist_worker.CASE = base
    mesh, bath, uv, eta = dist_worker._case()
    run_workers(gpu_worker, world, n_steps, str(tmp_path), axis=0, case=case)
    u_p, e_p, extra = gather(str(tmp_path), world, mesh.num_cells)
    dev = Swe2dDevice(mesh, bath, 2.0)
    dev.set_state(uv, eta)
    if '+fe' in case:
        dev.advance_forward_euler(n_steps)
    else:
        dev.advance(n_steps)
    u_s, e_s = dev.get_state()
    dev.close()
    dist_worker.CASE = 'channel'
    assert np.array_equal(u_p, u_s) and np.array_equal(e_p, e_s)


@pytest.mark.gpu
@pytest.mark.parametrize('world,case,n_steps', [
    (2, 'channel64+every2+p2p', 7), (2, 'channel64+every4', 9), (3, 'delaunay+p2p+graph', 2), (4, 'channel64+every2+overlap3+p2p+graph', 9),
    (2, 'channel64+every2+p2p+capture', 8)])
def test_ranks_on_one_gpu_with_the_fused_stage_pair(tmp_path, hip_lib, monkeypatch, world, case, n_steps):
    """Stages 1 and 2 of every step of an exchange cycle as ONE launch by overlapped tiles on a partition (csrc/swe2d_fuse.h,
    swe2d_solve_stage_pair_cells: tiles cut from an order in which the ghost layers sit next to the owned cells they touch, stage 2
    on the shrinking range of its stage; forced here - ranks of the bench mesh take it from 250 k cells by themselves): bitwise the
    single-device run, with host-staged and peer-to-peer halos, per-cycle graphs, overlap (the early stages stay stage launches) and
    a capture outside advance()."""
    from thetis_amd.device import Swe2dDevice
    import dist_worker
    monkeypatch.setenv('THETIS_AMD_FUSE12', '1')
    base = case.split('+')[0]
    dist_worker.CASE = base
    mesh, bath, uv, eta = dist_worker._case()
    run_workers(gpu_worker, world, n_steps, str(tmp_path), axis=0, case=case)
    u_p, e_p, extra = gather(str(tmp_path), world, mesh.num_cells)
    assert all(int(d['fused']) == 1 for d in extra), 'a rank did not take the fused stage pair'
    monkeypatch.setenv('THETIS_AMD_FUSE12', '0')
    monkeypatch.setenv('THETIS_AMD_FLOW', '0')
    dev = Swe2dDevice(mesh, bath, 2.0)
    dev.set_state(uv, eta)
    dev.advance(n_steps)
    u_s, e_s = dev.get_state()
    dev.close()
    dist_worker.CASE = 'channel'
    assert np.array_equal(u_p, u_s) and np.array_equal(e_p, e_s)


@pytest.mark.gpu
@pytest.mark.parametrize('world,case,n_steps', [
    (2, 'channel64+every2+p2p+nosplit', 7), (2, 'channel64+every4+nosplit', 9), (3, 'delaunay+every2+p2p+nosplit', 2),
    (4, 'channel64+every3+p2p+nosplit+graph', 9), (2, 'channel64+every2+p2p+nosplit+capture', 8), (2, 'channel256+every4+p2p+nosplit+capture', 8),
    (4, 'channel256+every4+p2p+nosplit+graph', 11), (2, 'channel360k+every4+p2p+nosplit+capture+byrule', 8)])
def test_ranks_on_one_gpu_with_whole_steps_in_one_launch(tmp_path, hip_lib, monkeypatch, world, case, n_steps):
    """A partition's steps as ONE launch each (csrc/swe2d_fuse.h swe_fuse123_kernel through swe2d_solve_step_cells: two-ring tiles over
    owned and ghost cells - the 11 x 8-quad patches of the parent mesh, or runs of the Hilbert order on a Delaunay partition -, stage 3
    on the step's last shrinking range, the state buffers change places after every launch): DistributedSwe2d takes the steps of a cycle
    in pairs, so that every cycle - eager, per-cycle graph or one graph for the whole advance - ends on the buffer it began on; an odd
    step of a cycle (m = 3, and the trailing cycles of these step counts) goes by the fused pair + stage 3.  Forced here - ranks take it by
    themselves beyond 131 k cells: the last case, two ranks of 180 k cells.  Bitwise the single-device run by stage launches."""
    from thetis_amd.device import Swe2dDevice
    import dist_worker
    if '+byrule' in case:                 # 180 k cells per rank: taken without being asked for
        monkeypatch.delenv('THETIS_AMD_FUSE12', raising=False)
        case = case.replace('+byrule', '')
    else:
        monkeypatch.setenv('THETIS_AMD_FUSE12', '3')
    base = case.split('+')[0]
    dist_worker.CASE = base
    mesh, bath, uv, eta = dist_worker._case()
    run_workers(gpu_worker, world, n_steps, str(tmp_path), axis=0, case=case)
    u_p, e_p, extra = gather(str(tmp_path), world, mesh.num_cells)
    assert all(int(d['step3']) == 1 for d in extra), 'a rank did not take the one-launch steps'
    monkeypatch.setenv('THETIS_AMD_FUSE12', '0')
    monkeypatch.setenv('THETIS_AMD_FLOW', '0')
    dev = Swe2dDevice(mesh, bath, dist_worker._dt())
    dev.set_state(uv, eta)
    dev.advance(n_steps)
    u_s, e_s = dev.get_state()
    dev.close()
    dist_worker.CASE = 'channel'
    assert np.array_equal(u_p, u_s) and np.array_equal(e_p, e_s)


@pytest.mark.gpu
@pytest.mark.parametrize('world,case,n_steps', [
    (2, 'channel+flow', 3), (2, 'channel+every2+flow', 5), (2, 'channel+p2p+flow', 3), (3, 'channel+every3+p2p+flow+graph', 11),
    (2, 'channel+every2+p2p+flow+nosplit+graph', 9), (2, 'channel+every4+p2p+flow+capture', 8), (3, 'delaunay+p2p+flow+graph', 2),
    (2, 'channel+p2p+flowx', 3), (2, 'channel+every2+p2p+flowx', 5), (3, 'channel+every3+p2p+flowx+graph', 11), (4, 'channel+every2+p2p+flowx', 24),
    (2, 'channel+every4+p2p+flowx+capture', 8), (3, 'delaunay+p2p+flowx+graph', 2), (2, 'delaunay+every2+p2p+flowx', 3), (2, 'channel+every1+p2p+flowx', 40),
    (2, 'channel+every1+p2p+flowx', 150), (3, 'channel+every2+p2p+flowx+mix', 11), (2, 'channel+every2+p2p+flow+mix+graph', 9),
    (2, 'channel+every4+p2p+mix+graph', 10), (4, 'channel64+every1+p2p+flowx', 9), (4, 'channel256+every1+p2p+flowx', 9), (4, 'channel256+every2+p2p+flowx', 9)])
def test_ranks_on_one_gpu_with_one_launch_per_cycle(tmp_path, hip_lib, monkeypatch, world, case, n_steps):
    """flow: the 3m stages of a cycle in ONE dataflow launch on the shrinking ranges (csrc/swe2d_flow.h; the blocks of the
    launch follow a locality order over owned and ghost cells), then the exchange - eager, host-staged and peer-to-peer, and
    replayed from per-cycle HIP graphs, with a shorter trailing cycle.  '+flowx': the exchange inside the launches (FX kernels: up
    to 64 cycles per launch, ghost lanes read the landing zone, send lanes store into the peers' zones; the runs of 24 and 40 steps
    are cut into launches of at most five cycles, so that they take several launches and a trailing partial cycle; the 150-step run
    takes launches of 64, 64 and 22 cycles).  '+mix': batches alternate with time steps driven stage by stage from the host
    (``run_stage``: exchange kernels on channel 0 between flow launches whose granules travel on the last channel).  Bitwise the
    single-device result."""
    from thetis_amd.device import Swe2dDevice
    import dist_worker
    if '+flowx' in case and n_steps in (24, 40):
        monkeypatch.setenv('THETIS_AMD_FLOWX_CYCLES', '5')
    dist_worker.CASE = case.split('+')[0]
    try:
        mesh, bath, uv, eta = dist_worker._case()
        run_workers(gpu_worker, world, n_steps, str(tmp_path), axis=0, case=case)
    finally:
        dist_worker.CASE = 'channel'
    u_p, e_p, extra = gather(str(tmp_path), world, mesh.num_cells)
    dev = Swe2dDevice(mesh, bath, 2.0)
    dev.set_state(uv, eta)
    dev.advance(n_steps)
    u_s, e_s = dev.get_state()
    dev.close()
    assert np.array_equal(u_p, u_s) and np.array_equal(e_p, e_s)


@pytest.mark.gpu
@pytest.mark.parametrize('case,n_steps,min_peers', [('channel64+every2+p2p+flowx', 12, 2), ('channel+every2+p2p+flow+graph', 7, 4),
                                                    ('channel+every2+p2p+graph', 7, 4), ('channel+every1+overlap2', 4, 2)])
def test_eight_ranks_on_one_gpu_match_single_device(tmp_path, hip_lib, case, n_steps, min_peers):
    """BASELINE cfg 3's rank count with the bits checked: eight processes share the test GPU (IPC handles exchanged among eight
    processes, SWE_P2P_MAX_PEERS = 8).  'channel64': strips of eight cell columns, wider than the six-layer halo - the in-launch
    exchange of the flow kernel ('+flowx', middle ranks with two peers).  'channel': strips of two columns, narrower than the
    halo of ``every2``, so a rank's ghost layers reach its second and third neighbours (a cell is then sent to more than two peers:
    flow launches followed by the exchange kernels, and exchange kernels inside per-cycle HIP graphs); host-staged exchange with
    overlap.  Bitwise the single-device result."""
    from thetis_amd.device import Swe2dDevice
    import dist_worker
    dist_worker.CASE = case.split('+')[0]
    try:
        mesh, bath, uv, eta = dist_worker._case()
        run_workers(gpu_worker, 8, n_steps, str(tmp_path), axis=0, case=case)
    finally:
        dist_worker.CASE = 'channel'
    u_p, e_p, extra = gather(str(tmp_path), 8, mesh.num_cells)
    dev = Swe2dDevice(mesh, bath, 2.0)
    dev.set_state(uv, eta)
    dev.advance(n_steps)
    u_s, e_s = dev.get_state()
    dev.close()
    assert np.array_equal(u_p, u_s) and np.array_equal(e_p, e_s)
    peers = [len(set(int(q) for q in np.atleast_1d(d['peers']))) for d in extra if 'peers' in d]
    assert max(peers) >= min_peers


@pytest.mark.gpu
@pytest.mark.parametrize('world,case,n_steps,where', [(2, 'channel+every2+p2p+flowx+delay', 12, 15), (3, 'channel+every1+p2p+flowx+delay', 9, 12),
                                                      (4, 'channel64+every2+p2p+flowx+delay', 10, 3)])
def test_lagging_blocks_do_not_change_the_in_launch_exchange(tmp_path, hip_lib, monkeypatch, world, case, n_steps, where):
    """Adversary for the in-launch exchange (-DSWE_FLOW_DELAY build only; skipped with the product library): on every rank one block
    sleeps 15 us - two to three stage periods - before in-launch receives (4) and pushes (8), before polling passes (1) and
    publishes (2), while its peers run ahead as far as "push n + 2 only after receive n + 1" lets them.  Bitwise the single device."""
    import ctypes
    from thetis_amd import _lib
    from thetis_amd.device import Swe2dDevice
    import dist_worker
    dist_worker.CASE = case.split('+')[0]
    try:
        mesh, bath, uv, eta = dist_worker._case()
        dev = Swe2dDevice(mesh, bath, 2.0)
        if dev.lib.swe2d_debug_flow_delay(dev.h, -1, 0, 0, 1) != _lib.OK:
            dev.close()
            pytest.skip('needs the -DSWE_FLOW_DELAY build (tools/range_check.sh)')
        monkeypatch.setenv('FLOW_DELAY_WHERE', str(where))
        run_workers(gpu_worker, world, n_steps, str(tmp_path), axis=0, case=case)
    finally:
        dist_worker.CASE = 'channel'
    u_p, e_p, extra = gather(str(tmp_path), world, mesh.num_cells)
    dev.set_state(uv, eta)
    dev.advance(n_steps)
    u_s, e_s = dev.get_state()
    dev.close()
    assert np.array_equal(u_p, u_s) and np.array_equal(e_p, e_s)


@pytest.mark.gpu
@pytest.mark.parametrize('world,case,n_steps', [(2, 'channel+every2+p2p+flowx+tear', 12), (3, 'channel+every1+p2p+flowx+tear', 9)])
def test_torn_granules_do_not_change_the_in_launch_exchange(tmp_path, hip_lib, world, case, n_steps):
    """Adversary for the granules that cross ranks (-DSWE_FLOW_TEAR build only; skipped with the product library): every rim publish
    and every push into a peer's landing zone stores the half with the new tag 3 us before the value it belongs to.  With the check
    word the receiving side re-polls: bitwise the single device.  (Negative control, -DSWE_FLOW_NOCHECK: must differ.)"""
    from thetis_amd.device import Swe2dDevice
    import dist_worker
    dist_worker.CASE = 'channel'
    mesh, bath, uv, eta = dist_worker._case()
    dev = Swe2dDevice(mesh, bath, 2.0)
    rc = dev.lib.swe2d_debug_flow_tear(dev.h, -3, 0, 1, 0)
    if rc < 0:
        dev.close()
        pytest.skip('needs the -DSWE_FLOW_TEAR build (tools/range_check.sh)')
    run_workers(gpu_worker, world, n_steps, str(tmp_path), axis=0, case=case)
    u_p, e_p, extra = gather(str(tmp_path), world, mesh.num_cells)
    dev.set_state(uv, eta)
    dev.lib.swe2d_debug_flow_tear(dev.h, -1, 0, 1, 0)
    for _ in range(n_steps):
        for i in range(3):
            dev.solve_stage(i)
    u_s, e_s = dev.get_state()
    dev.close()
    same = np.array_equal(u_p, u_s) and np.array_equal(e_p, e_s)
    assert same if rc == 1 else not same


@pytest.mark.gpu
@pytest.mark.parametrize('world,case,n_steps', [(2, 'channel+every4+overlap3+p2p+graph', 16), (3, 'channel+every2+p2p', 7),
                                                (2, 'channel+every2+p2p+flow+graph', 9)])
def test_exchange_kernels_on_a_side_stream(tmp_path, hip_lib, monkeypatch, world, case, n_steps):
    """THETIS_AMD_P2P_SIDE_STREAM=1 (opt-in; measured slower, DESIGN_ANNEX.md A5): push and wait-and-unpack on a stream of their own,
    forked and joined by events - eagerly and inside per-cycle HIP graphs (cross-stream capture).  Bitwise the single device."""
    from thetis_amd.device import Swe2dDevice
    import dist_worker
    monkeypatch.setenv('THETIS_AMD_P2P_SIDE_STREAM', '1')
    dist_worker.CASE = 'channel'
    mesh, bath, uv, eta = dist_worker._case()
    run_workers(gpu_worker, world, n_steps, str(tmp_path), axis=0, case=case)
    u_p, e_p, extra = gather(str(tmp_path), world, mesh.num_cells)
    dev = Swe2dDevice(mesh, bath, 2.0)
    dev.set_state(uv, eta)
    dev.advance(n_steps)
    u_s, e_s = dev.get_state()
    dev.close()
    assert np.array_equal(u_p, u_s) and np.array_equal(e_p, e_s)


@pytest.mark.gpu
@pytest.mark.parametrize('case,n_steps', [('quadhalf', 4), ('quadhalf+every2+p2p', 5)])
def test_a_rank_of_parallelograms_takes_the_general_kernels_of_the_mesh(tmp_path, hip_lib, case, n_steps):
    """A quadrilateral mesh whose general (non-parallelogram) cells all lie in the right third, two strips: rank 0's own cells and
    ghost layers are parallelograms only.  Left to itself its handle would take the constant-Jacobian kernels and its cells next to
    the cut would differ from the single-device run (general kernels everywhere) in the last bits; the partition carries the global
    mesh's flag (``LocalPartition.affine`` -> ``swe2d_set_general_quadrilaterals``).  Bitwise the single device."""
    from thetis_amd.device import Swe2dDevice
    from thetis_amd.partition import build_partition, strip_owner
    import dist_worker
    dist_worker.CASE = 'quadhalf'
    try:
        mesh, bath, uv, eta = dist_worker._case()
        assert not mesh.affine
        part0 = build_partition(mesh, strip_owner(mesh, 2), 0, halo_depth=6)
        from thetis_amd.mesh import Mesh2d
        local = Mesh2d(part0.vertex_xy, part0.cells)
        assert local.affine and not part0.affine                 # the premise: rank 0 alone would decide otherwise
        run_workers(gpu_worker, 2, n_steps, str(tmp_path), axis=0, case=case)
    finally:
        dist_worker.CASE = 'channel'
    u_p, e_p, extra = gather(str(tmp_path), 2, mesh.num_cells)
    dev = Swe2dDevice(mesh, bath, 2.0, boundary_len=mesh.boundary_len)
    dev.set_state(uv, eta)
    dev.advance(n_steps)
    u_s, e_s = dev.get_state()
    dev.close()
    assert np.array_equal(u_p, u_s) and np.array_equal(e_p, e_s)


@pytest.mark.gpu
def test_in_launch_exchange_is_refused_by_all_ranks_together(tmp_path, hip_lib):
    """``flow_exchange=True`` where a cell goes to more than two peers (strips narrower than the halo): every rank raises the same
    ValueError at its first ``advance`` - none is left waiting for granules that will never come."""
    import dist_worker
    with pytest.raises(AssertionError, match='exit code'):
        run_workers(gpu_worker, 8, 4, str(tmp_path), axis=0, case='channel+every2+p2p+flowx')
    log = (tmp_path/'refused.txt')
    assert sorted(log.read_text().split()) == [str(r) for r in range(8)]


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['channel+every2+overlap3+capture', 'channel+every4+p2p+capture'])
def test_capture_outside_advance_as_the_bench_does(tmp_path, hip_lib, case):
    """bench.py calls DistributedSwe2d._capture directly (not through advance): the capture run must use the solver's own
    stream for the exchange ordering and restore the state only after everything it enqueued has finished."""
    n_steps = 8
    mesh, (u_s, e_s) = _single_device(n_steps)
    run_workers(gpu_worker, 2, n_steps, str(tmp_path), axis=0, case=case)
    u_p, e_p, extra = gather(str(tmp_path), 2, mesh.num_cells)
    assert np.array_equal(u_p, u_s) and np.array_equal(e_p, e_s)


@pytest.mark.gpu
def test_two_ranks_coupled_with_peer_to_peer_halos(tmp_path, hip_lib):
    """tracer channels of the landing zone: SWE state on channel 0, the tracer on channel 1"""
    from thetis_amd.device import Swe2dDevice
    from dist_worker import tracer_initial
    mesh, bath, uv, eta = _case()
    run_workers(gpu_coupled_worker, 2, 3, str(tmp_path), axis=0, case='channel+p2p')
    u_p, e_p, extra = gather(str(tmp_path), 2, mesh.num_cells)
    dev = Swe2dDevice(mesh, bath, 2.0)
    dev.set_state(uv, eta)
    tid = dev.add_tracer()
    dev.tracer_set_state(tid, tracer_initial(mesh))
    dev.advance_coupled(3)
    u_s, e_s = dev.get_state()
    T_s = dev.tracer_get_state(tid)
    dev.close()
    assert np.array_equal(u_p, u_s) and np.array_equal(e_p, e_s) and np.array_equal(extra[-1], T_s)


@pytest.mark.gpu
def test_a_lost_peer_costs_one_bounded_wait_and_never_hangs_the_device(hip_lib):
    """The wait of the peer-to-peer unpack is bounded (THETIS_AMD_P2P_TIMEOUT_S) and sticky: a peer that never pushes costs one
    timeout, the next waits of the handle do not spin at all, the state stays finite and the device answers.  (Loopback zone:
    the rank is its own peer, and nobody pushes.)"""
    import time
    from thetis_amd.device import Swe2dDevice
    from thetis_amd.partition import build_partition, strip_owner
    mesh, bath, uv, eta = _case()
    part = build_partition(mesh, strip_owner(mesh, 2), 0)
    os.environ['THETIS_AMD_P2P_TIMEOUT_S'] = '0.3'
    try:
        dev = Swe2dDevice(part, bath[part.vertex_global], 2.0, n_owned=part.n_owned, boundary_len=part.boundary_len,
                          ranges=part.reorder_ranges())
        dev.halo_setup(part.send_cells, part.recv_cells)
        dev.set_state(uv[part.local_to_global], eta[part.local_to_global])
        dev.p2p_create([9])
        _, base, _ = dev.p2p_export()
        peers = sorted(part.send)
        dev.p2p_connect([base]*len(peers), [part.send[q][0] for q in peers], [part.send[q][1] for q in peers],
                        [part.recv[q][0] for q in peers], list(range(len(peers))), [len(part.recv_cells)]*len(peers), n_from=len(peers))
        t0 = time.perf_counter()
        dev.p2p_wait_unpack(0, 0)                      # nobody pushed: runs into the timeout
        sent, received, timeouts = dev.p2p_status()
        t1 = time.perf_counter()
        assert timeouts == 1 and 0.25 < t1 - t0 < 5.0
        dev.p2p_wait_unpack(0, 0)                      # sticky: no second spin
        dev.p2p_wait_unpack(0, 0)
        _, _, timeouts = dev.p2p_status()
        assert timeouts >= 1 and time.perf_counter() - t1 < 0.25
        dev.p2p_push(0, 0)                             # the device still works
        for i in range(3):
            dev.solve_stage_cells(i, 0, part.n_owned)
        u, e = dev.get_state()
        assert np.isfinite(u[:part.n_owned]).all() and np.isfinite(e[:part.n_owned]).all()
        dev.close()
    finally:
        os.environ.pop('THETIS_AMD_P2P_TIMEOUT_S', None)


@pytest.mark.parametrize('world,depth', [(4, 6), (3, 12)])
def test_strip_submesh_partition_equals_the_partition_of_the_global_mesh(world, depth):
    """bench.py's large-mesh line builds every rank's strip from a sub-rectangle of the channel instead of the 8 M-cell mesh:
    the same cells (centroids), the same ghost layers, and send lists that are the peers' receive lists."""
    from thetis_amd.distributed import strip_submesh_case
    from thetis_amd.mesh import RectangleMesh
    from thetis_amd.partition import build_partition, strip_owner
    nx, ny, lx, ly = 96, 10, 100e3, 50e3
    glob = RectangleMesh(nx, ny, lx, ly)
    owner = strip_owner(glob, world)
    parts = [strip_submesh_case(r, world, nx, ny, lx, ly, depth)[0] for r in range(world)]

    def cen(p, cells=None):
        c = p.vertex_xy[p.cells].mean(axis=1)
        return c if cells is None else c[cells]
    for r in range(world):
        ref = build_partition(glob, owner, r, halo_depth=depth)
        p = parts[r]
        assert p.n_owned == ref.n_owned and p.layer_sizes == ref.layer_sizes and sorted(p.send) == sorted(ref.send)
        # the same owned cells and the same cells layer by layer (as sets of centroids)
        key = lambda a: np.round(a, 3)[np.lexsort(np.round(a, 3).T)]
        assert np.array_equal(key(cen(p)[:p.n_owned]), key(cen(ref)[:ref.n_owned]))
        a = p.n_owned
        for size in p.layer_sizes:
            assert np.array_equal(key(cen(p)[a:a + size]), key(cen(ref)[a:a + size]))
            a += size
        assert p.boundary_len == {1: ly, 2: ly, 3: lx, 4: lx}
        # channel ends are walls, strip ends are not: the owned cells carry markers only where the global partition does
        assert sorted(np.unique(-p.cell_nbr[:p.n_owned][p.cell_nbr[:p.n_owned] < 0])) == \
            sorted(np.unique(-ref.cell_nbr[:ref.n_owned][ref.cell_nbr[:ref.n_owned] < 0]))
    for r in range(world - 1):
        a, c = parts[r], parts[r + 1]
        for s, d in ((a, c), (c, a)):
            o, n = s.send[d.rank]
            o2, n2 = d.recv[s.rank]
            assert n == n2 and np.abs(cen(s, s.send_cells[o:o + n]) - cen(d, d.recv_cells[o2:o2 + n2])).max() < 1e-9
