"""Worker bodies for the 2-rank tests (spawned processes; gloo rendezvous on 127.0.0.1)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


CASE = 'channel'


def _case():
    from helpers import channel_case, delaunay_case, quad_case
    if CASE == 'quad':
        return quad_case(nx=16, ny=6, seed=21, amp_eta=0.3, amp_u=0.2)
    if CASE == 'quadgen':                        # general (non-parallelogram) convex cells
        return quad_case(nx=16, ny=6, seed=21, amp_eta=0.3, amp_u=0.2, warp=0.3)
    if CASE == 'quadhalf':                       # general cells only in the rightmost columns: rank 0 of two holds parallelograms only
        return quad_case(nx=24, ny=6, seed=21, amp_eta=0.3, amp_u=0.2, warp=0.3, warp_from=0.85)
    if CASE == 'delaunay':
        mesh, bath, uv, eta = delaunay_case(n_points=600, lx=100e3, ly=60e3, seed=7)
        return mesh, bath, 0.1*uv, 0.1*eta
    if CASE == 'channel64':                      # eight cell columns per rank of eight: wider than a six-layer halo
        return channel_case(nx=64, ny=6, seed=21, amp_eta=0.3, amp_u=0.2)
    if CASE == 'channel360k':                    # two ranks of 180 k cells each: beyond the dataflow kernel, where a partition takes its steps as one launch each by itself
        return channel_case(nx=600, ny=300, lx=100e3, ly=50e3, seed=21, amp_eta=0.3, amp_u=0.2)
    if CASE == 'channel256':                     # the mesh of the first-contact test (tests/test_gpu_bench_contract.py): 64 x 64 quads per rank of four
        return channel_case(nx=256, ny=64, seed=21, amp_eta=0.3, amp_u=0.2)
    return channel_case(nx=16, ny=6, seed=21, amp_eta=0.3, amp_u=0.2)


def _dt():
    """time step of the GPU workers: 2 s on the small cases, 0.5 s on the 167 m cells of 'channel360k' (2 s is beyond its CFL limit)"""
    return 0.5 if CASE == 'channel360k' else 2.0


def _split_every(case):
    """'channel+every2+overlap3' -> ('channel', 2, 3): exchange_every and overlap_stages of the distributed stepper"""
    overlap = 0
    if '+overlap' in case:
        case, j = case.split('+overlap')
        overlap = int(j)
    if '+every' in case:
        case, m = case.split('+every')
        return case, int(m), overlap
    return case, 1, overlap


def _init(rank, world, port):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['RANK'] = str(rank)
    os.environ['WORLD_SIZE'] = str(world)
    dist.init_process_group(backend='gloo', rank=rank, world_size=world)
    return dist


def cpu_worker(rank, world, port, n_steps, out_dir, axis, case='channel'):
    """Partition + halo exchange logic of the product (3m-layer halo, one exchange per m steps, optional overlap of the
    exchange with the ghost-independent part of the next stages), with the oracle's C restatement as the (CPU) compute.
    Mirrors DistributedSwe2d._steps_eager / _cycle_swe launch by launch on three rotating buffers; everything a launch
    must not read is NaN: stale ghost layers, ghosts between pack and unpack, cells a stage has not written yet."""
    global CASE
    fe = case.endswith('+fe')              # ForwardEuler: one stage and one ghost layer per step
    case = case.replace('+fe', '')
    case, every, overlap = _split_every(case)
    CASE = case
    import torch
    from oracle.ref_lib import RefSWE
    from thetis_amd.distributed import HaloExchanger
    from thetis_amd.partition import build_partition, strip_owner
    dist = _init(rank, world, port)
    mesh, bath, uv, eta = _case()
    if axis < 0:                       # recursive coordinate bisection instead of strips
        from thetis_amd.partition import rcb_owner
        owner = rcb_owner(mesh, world)
    else:
        owner = strip_owner(mesh, world, axis=axis)
    part = build_partition(mesh, owner, rank, halo_depth=(every if fe else 3*every))
    ref = RefSWE(part.cell_xy(), part.cell_nbr, part.cell_nbr_facet, bath[part.vertex_global][part.cells],
                 boundary_len=part.boundary_len)
    g = part.local_to_global
    halo = HaloExchanger(part, torch.device('cpu'))
    dt = 2.0
    no = part.n_owned
    k = part.cells.shape[1]
    # buffers 0..2 as the device rotates them: stage 0 reads 0 writes 1, stage 1 reads 1 writes 2, stage 2 reads 2 writes 0
    U = [uv[g].copy(), np.full_like(uv[g], np.nan), np.full_like(uv[g], np.nan)]
    E = [eta[g].copy(), np.full_like(eta[g], np.nan), np.full_like(eta[g], np.nan)]

    def stage(i, begin, end, keep_rest=False):
        """one launch; afterwards everything of the output buffer beyond ``end`` is stale (NaN) unless ``keep_rest``
        (launches that write a part of buffer 0 while the rest still holds the step result)"""
        src, dst = i, (i + 1) % 3
        ku, ke = ref.tendency(U[src], E[src], dt)          # computed everywhere; only [begin, end) is used
        nu = BE[i]*ku[begin:end] + AL0[i]*U[0][begin:end] + ALI[i]*U[src][begin:end]
        ne = BE[i]*ke[begin:end] + AL0[i]*E[0][begin:end] + ALI[i]*E[src][begin:end]
        assert not np.isnan(nu).any() and not np.isnan(ne).any(), 'stage {:d} on [{:d}, {:d}) read stale data'.format(i, begin, end)
        U[dst][begin:end], E[dst][begin:end] = nu, ne
        if not keep_rest:
            U[dst][end:], E[dst][end:] = np.nan, np.nan

    def start_exchange():
        sc = part.send_cells
        packed = np.concatenate([U[0][sc, :, 0], U[0][sc, :, 1], E[0][sc]], axis=1)      # [n][3k] = u.. v.. e..
        halo.send_buf[:packed.size] = torch.from_numpy(packed.reshape(-1))
        U[0][no:], E[0][no:] = np.nan, np.nan               # ghosts are stale until the unpack
        return halo.start()

    def finish_exchange(reqs):
        rc = part.recv_cells
        halo.finish(reqs)
        r = halo.recv_buf[:3*k*len(rc)].numpy().reshape(-1, 3*k)
        U[0][rc, :, 0], U[0][rc, :, 1], E[0][rc] = r[:, 0:k], r[:, k:2*k], r[:, 2*k:3*k]

    cycles = [every]*(n_steps//every) + ([n_steps % every] if n_steps % every else [])
    early = 0
    if fe:
        # DistributedSwe2d._cycle_forward_euler: the step goes from buffer 0 into buffer 1, then the buffers trade places
        def swap():
            U[0], U[1] = U[1], U[0]
            E[0], E[1] = E[1], E[0]
        for r in cycles:
            for gs in range(r - 1):
                stage(0, 0, part.stage_range(gs, depth=r))
                swap()
            stage(0, part.n_interior, no, keep_rest=True)
            sc = part.send_cells
            packed = np.concatenate([U[1][sc, :, 0], U[1][sc, :, 1], E[1][sc]], axis=1)
            halo.send_buf[:packed.size] = torch.from_numpy(packed.reshape(-1))
            reqs = halo.start()
            stage(0, 0, part.n_interior, keep_rest=True)
            swap()
            U[0][no:], E[0][no:] = np.nan, np.nan
            finish_exchange(reqs)
        cycles = []
    for ic, r in enumerate(cycles):
        nxt = min(overlap, 3*cycles[ic + 1] - 1) if ic + 1 < len(cycles) else 0
        n = 3*r
        for gs in range(n - 1):
            begin = part.owned_prefix(gs + 2) if gs < early else 0
            end = part.stage_range(gs, depth=n)
            stage(gs % 3, begin, end)
        stage(2, part.n_interior, no, keep_rest=True)
        reqs = start_exchange()
        stage(2, 0, part.n_interior, keep_rest=True)
        for gs in range(nxt):
            # the first early launch into buffer 1 / 2 finds nothing there that anyone still needs; buffer 0 holds the step
            # result, and from the second round on the rest of a buffer holds the previous round's early output
            stage(gs % 3, 0, part.owned_prefix(gs + 2), keep_rest=(gs >= 2))
        finish_exchange(reqs)
        early = nxt
    u, e = U[0], E[0]
    assert not np.isnan(u).any() and not np.isnan(e).any()
    np.savez(os.path.join(out_dir, 'rank{:d}.npz'.format(rank)), ids=g[:no], uv=u[:no], eta=e[:no],
             n_interior=part.n_interior, n_ghost=part.n_ghost)
    dist.barrier()
    dist.destroy_process_group()


AL0 = [0.0, 0.75, 0.33333333333333337]
ALI = [1.0, 0.25, 0.6666666666666666]
BE = [1.0, 0.25, 0.6666666666666666]


def tracer_initial(mesh):
    xy = mesh.cell_xy()
    rng = np.random.default_rng(77)
    # a front (the limiter has work to do) plus noise
    return 10.0*(xy[:, :, 0] > 0.4*xy[:, :, 0].max()) + rng.normal(size=xy.shape[:2])


def coupled_step_reference(ref, rt, u, e, T, dt, ranges, n_limit, exchange=None):
    """One coupled step (SWE, tracer with the updated velocity, limiter) written stage by stage on cell ranges; with
    ranges = (n, n, n) and n_limit = n it is the plain global algorithm, with a partition's ranges it is the
    distributed one.  Stale cells are poisoned with NaN so that a wrong range shows up."""
    u0, e0 = u.copy(), e.copy()
    cur_u, cur_e = u, e
    for i in range(3):
        end = ranges[i]
        ku, ke = ref.tendency(cur_u, cur_e, dt)
        new_u, new_e = np.full_like(cur_u, np.nan), np.full_like(cur_e, np.nan)
        new_u[:end] = BE[i]*ku[:end] + AL0[i]*u0[:end] + ALI[i]*cur_u[:end]
        new_e[:end] = BE[i]*ke[:end] + AL0[i]*e0[:end] + ALI[i]*cur_e[:end]
        cur_u, cur_e = new_u, new_e
    u, e = cur_u, cur_e
    if exchange is not None:
        exchange(u=u, e=e)
    T0, cur = T.copy(), T
    for i in range(3):
        end = ranges[i]
        k = rt.tendency(np.nan_to_num(cur), np.nan_to_num(u), dt)      # garbage in = garbage out beyond `end`, never used
        new = np.full_like(cur, np.nan)
        new[:end] = BE[i]*k[:end] + AL0[i]*T0[:end] + ALI[i]*cur[:end]
        cur = new
    T = cur
    if exchange is not None:
        exchange(T=T)
    assert not np.isnan(T).any()
    lim = rt.limit(T)
    T = T.copy()
    T[:n_limit] = lim[:n_limit]
    return u, e, T


def cpu_coupled_worker(rank, world, port, n_steps, out_dir, axis, case='channel'):
    """Coupled SWE + tracer + vertex limiter on a 4-layer vertex-adjacent halo with the C restatement as compute."""
    global CASE
    CASE = case
    import torch
    from oracle.ref_lib import RefSWE, RefTracer
    from thetis_amd.distributed import HaloExchanger
    from thetis_amd.partition import build_partition, strip_owner
    dist = _init(rank, world, port)
    mesh, bath, uv, eta = _case()
    owner = strip_owner(mesh, world, axis=axis)
    part = build_partition(mesh, owner, rank, halo_depth=4, adjacency='vertex')
    ref = RefSWE(part.cell_xy(), part.cell_nbr, part.cell_nbr_facet, bath[part.vertex_global][part.cells],
                 boundary_len=part.boundary_len)
    rt = RefTracer(ref, cell_topo_vertices=part.topo_vertex[part.cells])
    g = part.local_to_global
    u, e, T = uv[g].copy(), eta[g].copy(), tracer_initial(mesh)[g].copy()
    k = part.cells.shape[1]
    halo = HaloExchanger(part, torch.device('cpu'))
    thalo = HaloExchanger(part, torch.device('cpu'), width=k)
    sc, rc = part.send_cells, part.recv_cells

    def exchange(u=None, e=None, T=None):
        if T is None:
            packed = np.concatenate([u[sc, :, 0], u[sc, :, 1], e[sc]], axis=1)
            halo.send_buf[:packed.size] = torch.from_numpy(packed.reshape(-1))
            halo.finish(halo.start())
            r = halo.recv_buf[:3*k*len(rc)].numpy().reshape(-1, 3*k)
            u[rc, :, 0], u[rc, :, 1], e[rc] = r[:, 0:k], r[:, k:2*k], r[:, 2*k:3*k]
        else:
            thalo.send_buf[:k*len(sc)] = torch.from_numpy(np.ascontiguousarray(T[sc]).reshape(-1))
            thalo.finish(thalo.start())
            T[rc] = thalo.recv_buf[:k*len(rc)].numpy().reshape(-1, k)
    ranges = [part.stage_range(i) for i in range(3)]
    for _ in range(n_steps):
        u, e, T = coupled_step_reference(ref, rt, u, e, T, 2.0, ranges, part.layer_end(3), exchange)
        assert not np.isnan(u).any() and not np.isnan(e).any()
    no = part.n_owned
    np.savez(os.path.join(out_dir, 'rank{:d}.npz'.format(rank)), ids=g[:no], uv=u[:no], eta=e[:no], T=T[:no])
    dist.barrier()
    dist.destroy_process_group()


def cpu_coupled_cycles_worker(rank, world, port, n_steps, out_dir, axis, case='channel+every2'):
    """The product's coupled cycles (m coupled steps between two exchanges of all fields, coupled_cycle_schedule) launch by
    launch with the C restatement as compute; everything a launch must not read is NaN."""
    global CASE
    no_lim, fe = '+nolim' in case, '+fe' in case
    case = case.replace('+nolim', '').replace('+fe', '')
    case, every, _ = _split_every(case)
    CASE = case
    sps = 1 if fe else 3
    import torch
    from oracle.ref_lib import RefSWE, RefTracer
    from thetis_amd.distributed import HaloExchanger, coupled_cycle_schedule, coupled_halo_depth
    from thetis_amd.partition import build_partition, strip_owner
    dist = _init(rank, world, port)
    mesh, bath, uv, eta = _case()
    owner = strip_owner(mesh, world, axis=axis)
    part = build_partition(mesh, owner, rank, halo_depth=coupled_halo_depth(every, not no_lim, sps),
                           adjacency='facet' if no_lim else 'vertex')
    ref = RefSWE(part.cell_xy(), part.cell_nbr, part.cell_nbr_facet, bath[part.vertex_global][part.cells],
                 boundary_len=part.boundary_len)
    rt = RefTracer(ref, cell_topo_vertices=part.topo_vertex[part.cells])
    g = part.local_to_global
    u, e, T = uv[g].copy(), eta[g].copy(), tracer_initial(mesh)[g].copy()
    k = part.cells.shape[1]
    halo = HaloExchanger(part, torch.device('cpu'))
    thalo = HaloExchanger(part, torch.device('cpu'), width=k)
    sc, rc = part.send_cells, part.recv_cells
    dt = 2.0
    left = n_steps
    while left > 0:
        r = min(every, left)
        left -= r
        for op in coupled_cycle_schedule(part, r, 1, not no_lim, stages_per_step=sps):
            if op[0] == 'swe':
                _, i, end = op
                if i == 0:
                    u0, e0 = u.copy(), e.copy()
                ku, ke = ref.tendency(u, e, dt)
                nu, ne = np.full_like(u, np.nan), np.full_like(e, np.nan)
                nu[:end] = BE[i]*ku[:end] + AL0[i]*u0[:end] + ALI[i]*u[:end]
                ne[:end] = BE[i]*ke[:end] + AL0[i]*e0[:end] + ALI[i]*e[:end]
                assert not np.isnan(nu[:end]).any() and not np.isnan(ne[:end]).any()
                u, e = nu, ne
            elif op[0] == 'tracer':
                _, _, i, end = op
                if i == 0:
                    T0 = T.copy()
                # NaN in -> NaN out for the cells that read it: the C restatement propagates them like the arithmetic does
                kt = rt.tendency(T, u, dt)
                nt = np.full_like(T, np.nan)
                nt[:end] = BE[i]*kt[:end] + AL0[i]*T0[:end] + ALI[i]*T[:end]
                assert not np.isnan(nt[:end]).any()
                T = nt
            elif op[0] == 'limit':
                end = op[2]
                lim = rt.limit(T)
                nt = np.full_like(T, np.nan)
                nt[:end] = lim[:end]
                assert not np.isnan(nt[:end]).any()
                T = nt
        packed = np.concatenate([u[sc, :, 0], u[sc, :, 1], e[sc]], axis=1)
        assert not np.isnan(packed).any() and not np.isnan(T[sc]).any()
        halo.send_buf[:packed.size] = torch.from_numpy(packed.reshape(-1))
        halo.finish(halo.start())
        rr = halo.recv_buf[:3*k*len(rc)].numpy().reshape(-1, 3*k)
        u[rc, :, 0], u[rc, :, 1], e[rc] = rr[:, 0:k], rr[:, k:2*k], rr[:, 2*k:3*k]
        thalo.send_buf[:k*len(sc)] = torch.from_numpy(np.ascontiguousarray(T[sc]).reshape(-1))
        thalo.finish(thalo.start())
        T[rc] = thalo.recv_buf[:k*len(rc)].numpy().reshape(-1, k)
        assert not np.isnan(u).any() and not np.isnan(e).any() and not np.isnan(T).any()
    no = part.n_owned
    np.savez(os.path.join(out_dir, 'rank{:d}.npz'.format(rank)), ids=g[:no], uv=u[:no], eta=e[:no], T=T[:no])
    dist.barrier()
    dist.destroy_process_group()


def gpu_coupled_worker(rank, world, port, n_steps, out_dir, axis, case='channel'):
    """DistributedSwe2d with one tracer + limiter, two ranks sharing ONE GPU (gloo + host staging stands in for RCCL)."""
    global CASE
    p2p, combined, no_lim, fe, step3, graph = '+p2p' in case, '+combined' in case, '+nolim' in case, '+fe' in case, '+step3' in case, '+graph' in case
    case = case.replace('+p2p', '').replace('+combined', '').replace('+nolim', '').replace('+fe', '').replace('+step3', '').replace('+graph', '')
    case, every, overlap = _split_every(case)
    CASE = case
    from thetis_amd.distributed import DistributedSwe2d
    from thetis_amd.partition import strip_owner
    dist = _init(rank, world, port)
    mesh, bath, uv, eta = _case()
    owner = strip_owner(mesh, world, axis=axis)
    solver = DistributedSwe2d(mesh, bath, 2.0, rank, world, 0, owner=owner, n_tracers=1, exchange=('p2p' if p2p else 'host'),
                              exchange_every=every, combined_exchange=combined, use_limiter=not no_lim,
                              stepper=('ForwardEuler' if fe else 'SSPRK33'), overlap_stages=overlap)
    solver.set_state_global(uv, eta)
    solver.set_tracer_global(0, tracer_initial(mesh))
    if step3:            # whole shallow-water steps in one launch each (forced by the test: THETIS_AMD_FUSE12=3)
        assert solver.dev.fused_step_info()[0]
    if graph:            # twice the same advance: the second call replays what the first one captured
        solver.advance(n_steps - n_steps//2, use_graph=True)
        solver.advance(n_steps//2, use_graph=True)
    else:
        solver.advance(n_steps, use_graph=False)
    solver.synchronize()
    ids, u, e = solver.get_state_owned()
    _, T = solver.get_tracer_owned(0)
    td = solver.tracer_diagnostics(0)
    np.savez(os.path.join(out_dir, 'rank{:d}.npz'.format(rank)), ids=ids, uv=u, eta=e, T=T, td=td)
    dist.barrier()
    dist.destroy_process_group()


def viscosity_field(mesh):
    x, y = mesh.vertex_xy.T
    return 20.0 + 30.0*(x - x.min())/(x.max() - x.min()) + 10.0*np.sin(y/(y.max() + 1.0)*3.0)


def gpu_worker(rank, world, port, n_steps, out_dir, axis, case='channel'):
    global CASE
    flags = {}
    for f in ('+capture', '+graph', '+p2p', '+nosplit', '+flowx', '+flow', '+delay', '+mix', '+tear', '+verify'):   # order-independent suffix flags
        flags[f] = f in case
        case = case.replace(f, '')
    graphed = flags['+graph']              # per-cycle HIP graphs (around the eager host-staged exchange, or incl. the p2p kernels)
    fe = case.endswith('+fe')
    case = case.replace('+fe', '')
    case, every, overlap = _split_every(case)
    viscous = case.endswith('+visc')
    case = case.replace('+visc', '')
    CASE = case
    """The real DistributedSwe2d on ONE GPU shared by both ranks (gloo + host staging stands in for RCCL)."""
    from thetis_amd.distributed import DistributedSwe2d
    from thetis_amd.partition import strip_owner
    dist = _init(rank, world, port)
    mesh, bath, uv, eta = _case()
    owner = strip_owner(mesh, world, axis=axis)
    solver = DistributedSwe2d(mesh, bath, _dt(), rank, world, 0, owner=owner, exchange=('p2p' if flags['+p2p'] else 'host'),
                              exchange_every=every, overlap_stages=overlap, split_last_stage=not flags['+nosplit'],
                              stepper=('ForwardEuler' if fe else 'SSPRK33'), flow=(True if (flags['+flow'] or flags['+flowx']) else False), flow_exchange=flags['+flowx'],
                              **(dict(verify_every=5, graph_mode='full') if flags['+verify'] else {}))
    if viscous:     # SIPG pass on the partition: same cell ranges as the stage kernels, per-vertex viscosity of the local vertices
        solver.dev.set_viscosity(viscosity_field(mesh)[solver.part.vertex_global], use_grad_div_viscosity_term=True)
    solver.set_state_global(uv, eta)
    d0 = solver.diagnostics()
    if flags['+delay']:
        # -DSWE_FLOW_DELAY build: on every rank one block (rank r: block 2 + r) sleeps 15 us before every in-launch receive and push
        # and before every second polling pass / publish
        from thetis_amd import _lib
        rc = solver.dev.lib.swe2d_debug_flow_delay(solver.dev.h, 2 + rank, int(os.environ.get('FLOW_DELAY_WHERE', '15')), 15, 2)
        if rc != _lib.OK:
            raise RuntimeError('the loaded library is not the -DSWE_FLOW_DELAY build')
    if flags['+tear']:
        # -DSWE_FLOW_TEAR build: every block makes the granule stores of every second publish - and every push into a peer's landing
        # zone - in two halves, the new tag 3 us ahead of the value it belongs to
        rc = solver.dev.lib.swe2d_debug_flow_tear(solver.dev.h, -2, 3, 2, 1)
        if rc != 0:
            raise RuntimeError('the loaded library is not the -DSWE_FLOW_TEAR build')
    if flags['+mix']:
        # batches (flow launches with the exchange inside, or stage launches in graphs) alternating with time steps driven stage by
        # stage from the host, as FlowSolver2d does when some steps have forcing updates and others do not
        a = n_steps//3
        solver.advance(a, use_graph=graphed)
        for i in range(3):
            solver.run_stage('swe', i)
        solver.advance(n_steps - a - 2, use_graph=graphed)
        for i in range(3):
            solver.run_stage('swe', i)
    elif flags['+capture']:
        # what bench.py does: capture outside advance() (state restored), one untimed replay, state reset, the run
        solver._capture(n_steps)
        assert solver.graphed
        solver.advance(n_steps, use_graph=True)
        solver.synchronize()
        solver.set_state_global(uv, eta)
        solver.advance(n_steps, use_graph=True)
    elif flags['+verify']:
        # verification windows of five steps that the advance sizes do not divide, whole advances captured as ONE graph: the second
        # and third call capture (warm-up steps, state restored) in the MIDDLE of a window - the window's start must survive that
        # (one snapshot buffer served both; every rank then replayed from the wrong state and "found" a mismatch)
        a = n_steps//3
        solver.advance(a, use_graph=True)
        solver.advance(a + 1, use_graph=True)
        solver.advance(n_steps - 2*a - 1, use_graph=True)
        assert solver.graph_mode == 'full' and solver.verify_report['windows'] >= 2, solver.verify_report
        assert solver.verify_report['mismatches'] == 0, solver.verify_report
    elif graphed:
        # twice the same advance: the second call replays the graphs the first one captured
        solver.advance(n_steps - n_steps//2, use_graph=True)
        solver.advance(n_steps//2, use_graph=True)
        assert solver.graphed and solver.graph_mode == 'cycle'
    else:
        try:
            solver.advance(n_steps, use_graph=False)
        except ValueError as e:
            if 'flow_exchange=True' in str(e):         # refused collectively (test_in_launch_exchange_is_refused_by_all_ranks_together)
                with open(os.path.join(out_dir, 'refused.txt'), 'a') as f:
                    f.write('{:d}\n'.format(rank))
            raise
    solver.synchronize()
    if solver.p2p is not None:
        # channel 0 carries the exchange kernels' halo, the last channel the flow kernel's granules ('+flowx')
        sent, received, timeouts = solver.dev.p2p_status(solver.p2p.n_channels)
        ch = solver.p2p.n_channels - 1 if solver.flow_exchange else 0
        assert timeouts == 0 and sent == received and sent[ch] > 0 and solver.dev.flow_timeouts() == 0, (sent, received, timeouts)
    d1 = solver.diagnostics()
    ids, u, e = solver.get_state_owned()
    np.savez(os.path.join(out_dir, 'rank{:d}.npz'.format(rank)), ids=ids, uv=u, eta=e, d0=d0, d1=d1,
             peers=np.array(solver.part.peers, dtype=np.int64), fused=np.array(int(solver.dev.fused_pair_info()[0])), step3=np.array(int(solver.dev.fused_step_info()[0])))
    dist.barrier()
    dist.destroy_process_group()


def run_workers(target, world, n_steps, out_dir, axis=0, case='channel'):
    import multiprocessing as mp
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=target, args=(r, world, port, n_steps, out_dir, axis, case)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    for p in procs:
        if p.is_alive():
            p.terminate()
            raise RuntimeError('distributed worker timed out')
        assert p.exitcode == 0, 'distributed worker failed with exit code {:}'.format(p.exitcode)


def gather(out_dir, world, n_cells):
    k = np.load(os.path.join(out_dir, 'rank0.npz'))['eta'].shape[1]
    uv = np.full((n_cells, k, 2), np.nan)
    eta = np.full((n_cells, k), np.nan)
    extra = []
    for r in range(world):
        d = np.load(os.path.join(out_dir, 'rank{:d}.npz'.format(r)))
        uv[d['ids']] = d['uv']
        eta[d['ids']] = d['eta']
        extra.append(d)
    if 'T' in extra[0]:
        T = np.full((n_cells, k), np.nan)
        for d in extra:
            T[d['ids']] = d['T']
        assert not np.isnan(T).any()
        extra.append(T)
    assert not np.isnan(uv).any() and not np.isnan(eta).any(), 'some cell is owned by no rank'
    return uv, eta, extra


def spmd_worker(rank, world, port, out_dir, name, cpu=True, env=None):
    """One rank of an UNCHANGED FlowSolver2d user script (tests/spmd_cases.py) under ``world`` ranks: what
    ``python -m torch.distributed.run --nproc-per-node world script.py`` starts.  ``cpu``: the host stand-in device
    (tests/cpu_device.py) instead of the HIP library; on the GPU all ranks share device 0 (gloo control plane, p2p / host halos)."""
    import pickle
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'RANK': str(rank), 'WORLD_SIZE': str(world),
                       'LOCAL_RANK': str(rank), 'LOCAL_WORLD_SIZE': str(world), 'THETIS_AMD_DIST_BACKEND': 'gloo'})
    os.environ.update(env or {})
    from thetis_amd import solver2d
    if cpu:
        from cpu_device import CpuSwe2dDevice
        solver2d.FlowSolver2d._device_cls = CpuSwe2dDevice
    import spmd_cases
    res = spmd_cases.run(name, os.path.join(out_dir, 'out_w{:d}'.format(world)))
    with open(os.path.join(out_dir, 'res_w{:d}_r{:d}.pkl'.format(world, rank)), 'wb') as f:
        pickle.dump(res, f)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def run_spmd(world, out_dir, name, cpu=True, env=None, timeout=600):
    """spawn ``world`` ranks of ``spmd_worker``; returns the per-rank result dictionaries"""
    import multiprocessing as mp
    import pickle
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=spmd_worker, args=(r, world, port, out_dir, name, cpu, env)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout)
    for p in procs:
        if p.is_alive():
            for q in procs:
                if q.is_alive():
                    q.terminate()
            raise RuntimeError('spmd worker timed out')
        assert p.exitcode == 0, 'spmd worker failed with exit code {:}'.format(p.exitcode)
    out = []
    for r in range(world):
        with open(os.path.join(out_dir, 'res_w{:d}_r{:d}.pkl'.format(world, r)), 'rb') as f:
            out.append(pickle.load(f))
    return out
