"""
bench.py body for N > 1 (``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N``): first contact with a multi-GPU
node - transport verification against the host-staged exchange, schedule tuning inside a set-up budget, the timed region, the
8 M-triangle second region, the ONE JSON line.  A harness around the product (thetis_amd/distributed.py: DistributedSwe2d), not part
of it; moved out of the package in round 5.
"""
import os
import time

import numpy as np

from thetis_amd.distributed import DistributedSwe2d, state_digest, strip_submesh_case
from thetis_amd.partition import build_partition, strip_owner

LARGE_NX, LARGE_NY = 4000, 1000          # 8 M triangles of the bench channel (100 km x 50 km): 1 M per rank at N = 8


class _Agree(object):
    """Rank-collective decisions of the bench through one small CPU (gloo) all-reduce each: every rank takes the same
    branch even when only one of them saw an exception."""

    def __init__(self, group, world):
        self.group, self.world = group, world

    def _reduce(self, x, op):
        import torch
        import torch.distributed as dist
        t = torch.tensor([float(x)], dtype=torch.float64)
        if self.world > 1:
            dist.all_reduce(t, op=op, group=self.group)
        return float(t.item())

    def all_ok(self, ok):
        import torch.distributed as dist
        return self._reduce(1.0 if ok else 0.0, dist.ReduceOp.MIN) > 0.5

    def max(self, x):
        import torch.distributed as dist
        return self._reduce(x, dist.ReduceOp.MAX)

    def barrier(self):
        import torch.distributed as dist
        self._reduce(0.0, dist.ReduceOp.MAX)


def run_distributed_bench(args, build_case, dt, bytes_per_update, hbm_peak):
    """bench.py body for N > 1: strong scaling of the same 1M-triangle mesh, strips along x.

    First contact with a multi-GPU node must never end without the JSON line: every transport / schedule candidate is
    built, verified and timed inside try/except, the outcome is agreed on by all ranks over a gloo side channel, failures
    are listed in ``config.failures`` and the run falls back transport by transport (p2p -> rccl -> host)."""
    import datetime
    import json
    import traceback
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', str(rank)))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29511')
    # THETIS_AMD_DIST_BACKEND=gloo: test hook - several ranks share the visible GPU(s) (RCCL refuses two ranks on one
    # device), so 'rccl' is not a candidate; everything else of this function runs as on a multi-GPU node
    backend = os.environ.get('THETIS_AMD_DIST_BACKEND', 'nccl')
    have_rccl = backend == 'nccl'
    if not have_rccl:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    failures = []
    t_start = time.perf_counter()

    def progress(msg):
        """the running log of first contact (rank 0, stderr, flushed at once: a run that dies half way leaves what it got to)"""
        if rank == 0:
            import sys
            print('[thetis_amd {:7.1f} s] {:}'.format(time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    def note(what, exc=None):
        msg = what if exc is None else '{:}: {:}'.format(what, (str(exc).strip().splitlines() or [type(exc).__name__])[0][:300])
        failures.append(msg)
        progress('FAILED / dropped: ' + msg)

    # control plane = gloo on CPU tensors (always available); RCCL only carries device-side exchanges and is created
    # lazily by its first use, so a node whose RCCL is broken still produces a number through 'p2p' or 'host'
    timeout = datetime.timedelta(seconds=float(os.environ.get('THETIS_AMD_DIST_TIMEOUT_S', '300')))
    if have_rccl:
        try:
            dist.init_process_group(backend='cpu:gloo,cuda:nccl', rank=rank, world_size=world, timeout=timeout)
        except Exception as e:
            note('init_process_group(cpu:gloo,cuda:nccl) failed, continuing without RCCL', e)
            have_rccl = False
            if dist.is_initialized():
                dist.destroy_process_group()
    if not have_rccl:
        dist.init_process_group(backend='gloo', rank=rank, world_size=world, timeout=timeout)
    ctrl = dist.new_group(backend='gloo', timeout=timeout) if world > 1 else None
    agree = _Agree(ctrl, world)
    progress('{:d} ranks, control plane gloo, device collectives {:}; visible devices {:d}'.format(
        world, 'RCCL (created by its first use)' if have_rccl else 'none (gloo backend)', torch.cuda.device_count()))
    mesh, bath, uv, eta = build_case()
    n_total = mesh.num_cells
    use_graph = not os.environ.get('THETIS_AMD_NO_GRAPH')
    mode0 = os.environ.get('THETIS_AMD_GRAPH_MODE', 'cycle')      # 'full' only on request

    parts = {}
    owner = strip_owner(mesh, world)

    def make(exchange, every, overlap, split, mode, flow=False):
        if every not in parts:
            parts[every] = build_partition(mesh, owner, rank, halo_depth=3*every)
        s = DistributedSwe2d(mesh, bath, dt, rank, world, local_rank, exchange_every=every, overlap_stages=overlap,
                             graph_mode=mode, exchange=exchange, split_last_stage=split, partition=parts[every],
                             group=(ctrl if exchange != 'rccl' else None), flow=flow)
        if flow:                                       # the kernel must cover every rank's partition (all its blocks resident at once)
            try:
                covered = bool(s.flow)
            except ValueError:                         # this rank's answer is "no": the others must still meet it in the all-reduce
                covered = False
            if not agree.all_ok(covered):
                s.close()
                raise RuntimeError('the flow kernel does not cover the partition of every rank')
        s.set_state_global(uv, eta)
        return s

    def attempt(label, fn):
        """run fn() on every rank; True when it succeeded everywhere (a local exception is recorded, not raised)"""
        ok, out = True, None
        try:
            out = fn()
        except Exception as e:
            ok = False
            note(label, e)
            if os.environ.get('THETIS_AMD_DEBUG'):
                traceback.print_exc()
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
        everywhere = agree.all_ok(ok)
        if ok and not everywhere:
            note(label + ': failed on another rank')
        return everywhere, out

    # ---- which transports work on this node?  'host' (gloo through host memory) is the yardstick: the others must reproduce
    #      its result bit for bit on a short run (an exchange is a pure copy)
    n_check = 9
    transports = []
    # everything from here to the timed region is set-up: transport checks and schedule tuning stop adding candidates once
    # THETIS_AMD_SETUP_BUDGET_S (default 60 s) is spent, so that first contact with a node always reaches the timed region
    t_setup0 = time.perf_counter()
    setup_budget = float(os.environ.get('THETIS_AMD_SETUP_BUDGET_S', '60'))
    skipped = []

    def budget_left():
        return agree.max(time.perf_counter() - t_setup0) < setup_budget
    forced = os.environ.get('THETIS_AMD_EXCHANGE')
    wanted = [forced] if forced else ['p2p'] + (['rccl'] if have_rccl else []) + ['host']
    digest_ref = None

    # ---- soak (round 5): nine steps catch a transport that is broken, not one that delivers a wrong word once in 10^6 messages.
    #      Every verified transport (and every flow candidate, which moves the halo itself) therefore also steps for >= soak_s
    #      seconds (THETIS_AMD_SOAK_S, default 2; inside the set-up budget) and must end on the bits of a run that has NO exchange at
    #      all: the whole mesh stepped by this rank's own GPU alone, the rank's owned rows hashed.  One reference serves all.
    soak_s = float(os.environ.get('THETIS_AMD_SOAK_S', '2'))
    soak = {'steps': 0, 'digest': None, 'seconds': soak_s, 'verified': [], 'reference_s': None}

    def soak_reference(n_steps, solver):
        from thetis_amd.device import Swe2dDevice
        import hashlib
        t0 = time.perf_counter()
        one = Swe2dDevice(mesh, bath, dt, device_id=local_rank)
        try:
            one.set_state(uv, eta)
            one.advance(n_steps)
            u1, e1 = one.get_state()
        finally:
            one.close()
        ids = np.sort(solver.part.local_to_global[:solver.part.n_owned])      # (the order state_digest hashes in)
        soak['reference_s'] = time.perf_counter() - t0
        return hashlib.blake2b(np.ascontiguousarray(u1[ids]).tobytes() + np.ascontiguousarray(e1[ids]).tobytes(), digest_size=16).hexdigest()

    def soak_run(s, label):
        """steps `s` (already past its short check) from the initial state for the soak's step count and compares; collective"""
        if soak_s <= 0 or world == 1:
            return
        if soak['digest'] is None:
            # the step count: what the first soaked schedule needs for soak_s seconds (timed on 240 steps, max over ranks), a multiple of 24
            s.set_state_global(uv, eta)
            s.advance(240, use_graph=False)
            s.synchronize()
            agree.barrier()
            t0 = time.perf_counter()
            s.advance(240, use_graph=False)
            s.synchronize()
            t_step = agree.max(time.perf_counter() - t0)/240
            soak['steps'] = int(min(400000, max(240, 24*int(np.ceil(soak_s/t_step/24)))))
            soak['digest'] = soak_reference(soak['steps'], s)
        s.set_state_global(uv, eta)
        t0 = time.perf_counter()
        s.advance(soak['steps'], use_graph=False)
        s.synchronize()
        took = agree.max(time.perf_counter() - t0)
        if s.p2p is not None and s.p2p.timeouts():
            raise RuntimeError('{:d} peer-to-peer waits timed out during the soak'.format(s.p2p.timeouts()))
        if not agree.all_ok(state_digest(s) == soak['digest']):
            raise RuntimeError('{:}: after {:d} steps ({:.1f} s) the state differs from the single-device run of the same mesh'.format(
                label, soak['steps'], took))
        soak['verified'].append({'what': label, 'steps': soak['steps'], 'seconds': took})
        progress('soak: {:} stepped {:d} steps in {:.1f} s and ended on the bits of the single-device run'.format(label, soak['steps'], took))

    def short_run(exchange):
        s = make(exchange, 2, 0, True, 'none')
        try:
            s.advance(n_check, use_graph=False)
            s.synchronize()
            if s.p2p is not None and s.p2p.timeouts():
                raise RuntimeError('{:d} peer-to-peer waits timed out'.format(s.p2p.timeouts()))
            dg = state_digest(s)
            if exchange != 'host' and digest_ref is not None and agree.all_ok(dg == digest_ref):
                soak_run(s, "transport '{:}'".format(exchange))
            return dg
        finally:
            s.close()

    if world > 1 and not forced:
        ok, digest_ref = attempt("transport 'host' (reference run)", lambda: short_run('host'))
        if not ok:
            digest_ref = None
    for ex in wanted:
        if ex == 'host':
            if digest_ref is not None or forced or world == 1:
                transports.append(ex)
            continue
        if transports and not budget_left():
            skipped.append("transport '{:}'".format(ex))
            continue
        ok, dg = attempt("transport '{:}'".format(ex), lambda ex=ex: short_run(ex))
        if ok and digest_ref is not None:
            same = agree.all_ok(dg == digest_ref)
            if not same:
                note("transport '{:}' does not reproduce the host-staged exchange bit for bit: not used".format(ex))
            ok = same
        if ok:
            transports.append(ex)
            progress("transport '{:}': set up on every rank, {:d} steps reproduce the host-staged exchange bit for bit".format(ex, n_check))
    if not transports:
        transports = ['host']

    # ---- exchange schedule: one exchange per `every` time steps on 3*every ghost layers, optionally overlapped with the
    #      first `overlap` stages of the next cycle, last stage split or not (bitwise the same result for every choice; see
    #      DistributedSwe2d).  The best choice depends on the node, so a few candidates are timed during set-up (not in the
    #      timed region; every rank takes the max over ranks and therefore the same decision).
    if os.environ.get('THETIS_AMD_EXCHANGE_EVERY'):
        sched = [(max(1, int(os.environ['THETIS_AMD_EXCHANGE_EVERY'])), int(os.environ.get('THETIS_AMD_OVERLAP_STAGES', '0')),
                  not os.environ.get('THETIS_AMD_NO_SPLIT'), mode0, os.environ.get('THETIS_AMD_FLOW') == '1')]
        candidates = [(transports[0],) + sched[0]]
    elif world == 1 and not os.environ.get('THETIS_AMD_TUNE_SCHEDULE'):
        candidates = [(transports[0], 4, 0, True, mode0, False)]
    else:
        candidates = []
        for ex in transports:
            # most promising first: the set-up budget may cut the list short.  flow = the 3m stages of a cycle as one dataflow
            # launch (csrc/swe2d_flow.h; needs every 64-cell block of the partition resident at once: <= 131 k cells, i.e.
            # m <= 2 for an eighth of the bench mesh)
            if ex == 'p2p':
                candidates += [(ex, 4, 0, False, mode0, False), (ex, 2, 0, False, mode0, True), (ex, 8, 0, False, mode0, False)]
                # one graph per cycle costs a graph launch per cycle; the whole timed loop in ONE graph, or no graph at all, were
                # both faster for a 125 k-cell rank (us/step, m = 4: cycle 30.8, none 29.7, full 29.2): let the node decide
                if mode0 == 'cycle' and use_graph:
                    candidates += [(ex, 2, 0, False, 'full', True), (ex, 4, 0, False, 'full', False), (ex, 8, 0, False, 'full', False),
                                   (ex, 4, 0, False, 'none', False)]
                candidates += [(ex, 4, 0, True, mode0, False), (ex, 8, 3, True, mode0, False), (ex, 2, 0, False, mode0, False),
                               (ex, 1, 0, False, mode0, True)]
            elif ex == 'rccl':
                # graphs take the per-launch CPU cost off the critical path (it matters once an RCCL enqueue sits in every
                # cycle); when the CPU keeps up anyway eager launches are a little faster: time both
                candidates += [(ex, 4, 0, True, mode0, False), (ex, 8, 0, True, mode0, False), (ex, 2, 0, False, mode0, True),
                               (ex, 8, 3, True, mode0, False), (ex, 8, 0, True, 'none', False)]
            elif len(transports) == 1:
                candidates += [(ex, 8, 0, True, 'none', False)]
    solver, chosen, best_us, tuning = None, None, float('inf'), []
    n_tune = 96
    first = True
    for cand in candidates:
        ex, every_c, overlap_c, split_c, mode_c, flow_c = cand
        if solver is not None and not budget_left():
            skipped.append(str(cand))
            continue
        if flow_c and len(candidates) > 1:
            # the flow kernel needs every 64-cell block of the partition resident at once (2 one-wave workgroups per SIMD x 1024
            # SIMDs on an MI355X): not a candidate for larger partitions - decided on every rank the same way, no failure
            if every_c not in parts:
                parts[every_c] = build_partition(mesh, owner, rank, halo_depth=3*every_c)
            if not agree.all_ok(parts[every_c].num_cells <= 2048*64):
                skipped.append(str(cand) + ': partition too large for the flow kernel')
                continue

        def time_candidate():
            s = make(*cand)
            try:
                if flow_c and digest_ref is not None:
                    # the flow kernels move the halo themselves (FX): what was verified for the transport's exchange kernels is
                    # verified again for this path - the same short run must reproduce the host-staged result bit for bit
                    s.advance(n_check, use_graph=False)
                    s.synchronize()
                    if not agree.all_ok(state_digest(s) == digest_ref):
                        raise RuntimeError('the flow path does not reproduce the host-staged exchange bit for bit')
                    if bool(s.flow_exchange) and not any(v['what'].startswith('in-launch') for v in soak['verified']):
                        soak_run(s, 'in-launch exchange of the flow kernel (m = {:d})'.format(every_c))
                    s.set_state_global(uv, eta)
                s.advance(2000 if first else n_tune, use_graph=False)          # connections; clocks (first candidate)
                s.synchronize()
                graph_c = use_graph and mode_c != 'none'
                if graph_c:
                    s._capture(n_tune)
                t_best = float('inf')
                for _ in range(4):
                    agree.barrier()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    s.advance(n_tune, use_graph=graph_c)
                    s.synchronize()
                    t_best = min(t_best, time.perf_counter() - t0)
                if s.p2p is not None and s.p2p.timeouts():
                    raise RuntimeError('peer-to-peer waits timed out')
                return s, t_best
            except Exception:
                s.close()
                raise
        if len(candidates) == 1:
            ok, s = attempt('candidate {:}'.format(cand), lambda: make(*cand))
            if ok:
                solver, chosen = s, cand
            break
        ok, res = attempt('candidate {:}'.format(cand), time_candidate)
        first = False
        if not ok:
            if res is not None:
                res[0].close()
            continue
        s, t_best = res
        us = 1e6*agree.max(t_best)/n_tune
        tuning.append({'exchange': ex, 'exchange_every': every_c, 'overlap_stages': overlap_c, 'split_last_stage': split_c,
                       'graph_mode': mode_c, 'flow': bool(flow_c), 'us_per_step': us})
        progress('schedule {:}: exchange every {:d} steps, overlap {:d}, split last stage {:}, graphs {:}, dataflow launches {:}: '
                 '{:.2f} us per step (max over ranks, best of 4 x {:d} steps){:}'.format(
                     ex, every_c, overlap_c, split_c, mode_c, bool(flow_c), us, n_tune, '  <- best so far' if us < best_us else ''))
        if us < best_us:
            if solver is not None:
                solver.close()
            solver, chosen, best_us = s, cand, us
        else:
            s.close()
    if solver is None:
        # last resort: host-staged exchange, eager launches
        chosen = ('host', 4, 0, True, 'none', False)
        ok, solver = attempt('fallback {:}'.format(chosen), lambda: make(*chosen))
        if not ok:
            solver = None
    setup_s = agree.max(time.perf_counter() - t_setup0)
    progress('set-up took {:.1f} s (budget {:.0f} s); transports verified: {:}; chosen schedule: {:}{:}'.format(
        setup_s, setup_budget, transports, chosen, '; skipped for the budget: {:}'.format(skipped) if skipped else ''))
    out = None
    if solver is not None:
        ok, out = attempt('timed region', lambda: _timed_region(args, solver, chosen, agree, uv, eta, use_graph, n_total, world,
                                                                bytes_per_update, hbm_peak, tuning, transports))
        if not ok:
            out = None
    if out is not None:
        progress('timed region: {:d} steps, {:.4f} ms per step, {:.3e} element-updates/s, volume conserved: {:}, p2p time-outs: {:}'.format(
            out['steps'], out['ms_per_step'], out['value'], out['config'].get('volume_conserved'), out['config'].get('p2p_timeouts')))
        out['config'].update({'setup_s': setup_s, 'setup_budget_s': setup_budget, 'setup_skipped': skipped,
                              'soak': {'seconds_asked': soak['seconds'], 'steps': soak['steps'], 'reference_s': soak['reference_s'],
                                       'verified': soak['verified'],
                                       'note': 'every entry stepped that long from the initial state and ended on the bits of the whole '
                                               'mesh stepped by one GPU alone (no exchange of any kind)'}})
        # second, untuned timed region: an 8x larger mesh of the same channel (1 M triangles per rank at N = 8), where a rank is
        # bandwidth-bound like the single-GPU headline - it shows whether partitions, halo transport and graphs scale once the
        # latency floor of a 125 k-cell rank is out of the picture
        if world > 1 and not os.environ.get('THETIS_AMD_NO_LARGE_MESH'):
            if solver is not None:
                solver.close()
                solver = None
            ok, large = attempt('large-mesh region', lambda: _large_mesh_region(args, chosen, agree, rank, world, local_rank, ctrl, dt,
                                                                               use_graph, bytes_per_update, hbm_peak))
            out['config']['large_mesh'] = large if ok else None
    if out is None:
        out = {'metric': 'DG element-updates/sec, 2D SWE DG-P1 SSPRK33', 'value': 0.0, 'unit': 'element-updates/s', 'n_gpus': int(world),
               'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': None, 'higher_is_better': True, 'scaling': 'strong',
               'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
               'config': {'workload': 'BASELINE cfg3 (1M-triangle channel over {:d} GPUs): NO transport worked'.format(world)},
               'error': 'every transport / schedule failed, see config.failures'}
    out['config']['failures'] = failures
    # RCCL writes its version banner to stdout: tear the communicator down first so that the JSON is the LAST stdout line
    try:
        if solver is not None:
            solver.close()
        dist.destroy_process_group()
    except Exception as e:
        note('teardown', e)
    if rank == 0:
        import ctypes
        import sys
        # RCCL's banner sits in the C stdio buffer (flushed at exit when stdout is a pipe or a file): flush it now so that
        # the JSON line is the last thing on stdout
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


def _large_mesh_region(args, chosen, agree, rank, world, local_rank, ctrl, dt, use_graph, bytes_per_update, hbm_peak):
    """config.large_mesh: K timed steps on RectangleMesh(4000, 1000) of the same channel with the chosen transport, one exchange
    per 4 steps, stage launches (untuned)."""
    import torch
    ex, mode = chosen[0], chosen[4]
    every = 4
    lx, ly = 100e3, 50e3
    nx_l, ny_l = LARGE_NX, LARGE_NY
    if os.environ.get('THETIS_AMD_LARGE_MESH'):                # tests: "nx,ny"
        nx_l, ny_l = (int(v) for v in os.environ['THETIS_AMD_LARGE_MESH'].split(','))
    part, bath, uv, eta = strip_submesh_case(rank, world, nx_l, ny_l, lx, ly, 3*every)
    n_total = 2*nx_l*ny_l
    s = DistributedSwe2d(None, bath, dt*1000.0/nx_l, rank, world, local_rank, exchange_every=every, graph_mode=mode, exchange=ex,
                         split_last_stage=True, partition=part, group=(ctrl if ex != 'rccl' else None), flow=False)
    try:
        s.dev.set_state(uv[part.local_to_global], eta[part.local_to_global])
        d0 = s.diagnostics()
        s.advance(200, use_graph=False)                        # connections, clocks
        s.synchronize()
        graph = use_graph and s.graph_mode != 'none'
        if graph:
            s._capture(args.steps)
            if s.graphed:
                s.advance(args.steps, use_graph=True)
                s.synchronize()
        agree.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.advance(args.steps, use_graph=graph)
        s.synchronize()
        agree.barrier()
        torch.cuda.synchronize()
        t = agree.max(time.perf_counter() - t0)
        d1 = s.diagnostics()
        ok = bool(np.isfinite(d1).all() and abs(d1[2] - d0[2])/d0[2] < 1e-10)
        # model of the same mesh on ONE GPU: the stage kernels beyond the Infinity Cache run at 0.58 of the 8 TB/s roofline
        # (roofline.frac_beyond_cache of the N = 1 line; 8 M cells measured 1115 us/step in round 2)
        t1_model = 3.0*bytes_per_update*n_total/(0.58*hbm_peak*1e9)
        return {'workload': 'RectangleMesh({:d},{:d},100e3,50e3) = {:d} triangles, strips along x, {:d} cells per rank'.format(
                    nx_l, ny_l, n_total, n_total//world),
                'n_cells': n_total, 'ms_per_step': float(1e3*t/args.steps), 'value': float(n_total*3.0*args.steps/t),
                'unit': 'element-updates/s', 'exchange': ex, 'exchange_every': every, 'graph_mode': s.graph_mode, 'hip_graph': bool(s.graphed),
                'frac_of_hbm_roofline_per_gpu': float(bytes_per_update*n_total/world*3*args.steps/t/1e9/hbm_peak),
                'speedup_model': float(t1_model/(t/args.steps)),
                'speedup_model_note': 'against a MODEL of one GPU on the same mesh (stage kernels at 0.58 of 8 TB/s beyond the '
                                      'Infinity Cache = {:.0f} us/step); not a measured single-GPU run'.format(1e6*t1_model),
                'volume_conserved': ok}
    finally:
        s.close()


def _timed_region(args, solver, chosen, agree, uv, eta, use_graph, n_total, world, bytes_per_update, hbm_peak, tuning, transports):
    import torch
    ex, every, overlap, split, _, flow_on = chosen
    solver.graph = None
    solver.set_state_global(uv, eta)
    d0 = solver.diagnostics()
    prewarm = float(getattr(args, 'prewarm', 0.0) or 0.0)
    if prewarm > 0:
        # clock settling (bench.py docstring): a FIXED number of steps so that every rank posts the same exchanges
        solver.advance(int(prewarm/100e-6), use_graph=False)
        solver.synchronize()
    if args.warmup > 0:
        solver.advance(args.warmup, use_graph=False)
    solver.synchronize()
    use_graph = use_graph and solver.graph_mode != 'none'
    if use_graph:
        # build the graphs for the timed step count before the timed region (capture is set-up, not stepping);
        # _capture restores the state it perturbs
        solver._capture(args.steps)
        if solver.graphed:
            # the first launch of an instantiated graph uploads it to the device (~1 ms for a few thousand nodes): spend
            # it on K more untimed warm-up steps instead of inside the timed region
            solver.advance(args.steps, use_graph=True)
            solver.synchronize()
    agree.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    solver.advance(args.steps, use_graph=use_graph)
    solver.synchronize()
    agree.barrier()
    torch.cuda.synchronize()
    t = agree.max(time.perf_counter() - t0)
    d1 = solver.diagnostics()
    ok = bool(np.isfinite(d1).all() and abs(d1[2] - d0[2])/d0[2] < 1e-10)
    timeouts = solver.p2p.timeouts() if solver.p2p is not None else 0
    hip_graph = bool(solver.graphed)
    was_flowx = bool(solver.flow_exchange)
    # SURVEY 8(d) cfg 3 "exchange time fraction": the same K steps twice more, eagerly - once as they are, once with every send and
    # receive left out (the halo goes stale: timing only, after everything that is reported has been read) - never fatal
    exchange_fraction, exchange_note = None, None
    try:
        times = []
        for stub in (False, True):
            solver._no_exchange = stub
            solver.advance(min(args.steps, 4*every), use_graph=False)          # the launches of this mode once, untimed
            solver.synchronize()
            agree.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            solver.advance(args.steps, use_graph=False)
            if solver.stream is not None:
                solver.stream.synchronize()
            torch.cuda.synchronize()
            times.append(agree.max(time.perf_counter() - t1))
        solver._no_exchange = False
        if times[0] > 0:
            exchange_fraction = float(max(0.0, 1.0 - times[1]/times[0]))
            exchange_note = ('eager launches, max over ranks: {:.1f} us per step with the exchange, {:.1f} without '
                             '(the in-launch exchange of the flow kernel is replaced by flow launches without an exchange)'.format(
                                 1e6*times[0]/args.steps, 1e6*times[1]/args.steps))
    except Exception as e:                                                     # noqa: BLE001
        solver._no_exchange = False
        exchange_note = 'not measured: {:}'.format((str(e).strip().splitlines() or [type(e).__name__])[0][:200])
    value = n_total*3.0*args.steps/t
    per_gpu_bytes = bytes_per_update*n_total/world
    transport = {'p2p': 'peer-to-peer stores into IPC-mapped landing zones ({:} memory) + epoch flags, {:}'.format(
                        solver.p2p.zone_kind if solver.p2p is not None else '',
                        'made by the flow kernel itself (FX: up to 64 exchange cycles per launch)' if was_flowx
                        else 'exchange kernels inside the per-cycle HIP graph'),
                 'rccl': 'RCCL batch_isend_irecv between two graph launches',
                 'host': 'gloo through host memory (fallback)'}[ex]
    return {
        'metric': 'DG element-updates/sec, 2D SWE DG-P1 SSPRK33',
        'value': float(value), 'unit': 'element-updates/s', 'n_gpus': int(world), 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': float(1e3*t/args.steps), 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
        'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': ('BASELINE cfg3: the cfg2 1M-triangle channel' if n_total == 1000000 else
                                'NOT the BASELINE workload (THETIS_AMD_BENCH_MESH): a {:d}-triangle channel'.format(n_total))
                               + ' strip-partitioned along x over {:d} GPUs, {:d}-layer halo, one exchange per {:d} time steps'.format(
                                   world, 3*every, every),
                   'n_cells': int(n_total),
                   'parallelism': 'dd{:d} (domain decomposition, {:d}-cell halo, 1 exchange per {:d} steps)'.format(
                       world, 3*every, every),
                   'exchange': ex, 'exchange_transport': transport, 'transports_verified': transports,
                   'exchange_every': every, 'overlap_stages': overlap, 'split_last_stage': split, 'flow': bool(solver.flow), 'flow_exchange': was_flowx,
                   'exchange_time_fraction': exchange_fraction, 'exchange_time_fraction_note': exchange_note,
                   'flow_timeouts': int(solver.dev.flow_timeouts()), 'schedule_tuning': tuning,
                   'hip_graph': hip_graph, 'graph_mode': solver.graph_mode, 'graph_warm_replays': int(hip_graph),
                   'volume_conserved': ok, 'p2p_timeouts': int(timeouts), 'prewarm_s': prewarm},
        'roofline': {'bound': 'hbm', 'achieved': float(per_gpu_bytes*3*args.steps/t/1e9), 'peak': hbm_peak, 'unit': 'GB/s',
                     'frac': float(per_gpu_bytes*3*args.steps/t/1e9/hbm_peak), 'traffic': None,
                     'note': 'per GPU, algorithmic bytes over wall time per stage (includes halo exchange); '
                             'kernel-only figure is measured at N=1'},
    }
