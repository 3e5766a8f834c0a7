"""
One process per GPU: halo exchange around the per-rank C-ABI handle.

Per cycle of m = ``exchange_every`` time steps (thetis_amd/partition.py: 3m ghost layers, ONE exchange per cycle
instead of one per stage):

    stage g = 0 .. 3m-2 on owned + the first 3m-1-g ghost layers  ->  last stage on the send cells  -> send
    -> exchange in flight  ||  last stage on the interior cells
       (+ optionally the ghost-independent part of the next cycle's first stages)  -> receive into the ghosts

The exchange is 72 B (96 B for quadrilaterals) per halo cell, i.e. pure latency.  Three transports (``exchange``):

``'p2p'``   peer-to-peer stores into IPC-mapped landing zones + epoch flags (csrc/swe2d_p2p.h): push and wait+unpack are
            two kernels on the handle's stream, the whole cycle is ONE HIP graph, no host or RCCL call in the step loop;
``'rccl'``  pack kernel -> ``batch_isend_irecv`` (backend "nccl" = RCCL over xGMI) -> unpack kernel; the kernel sequences
            before and during an exchange run from HIP graphs, the RCCL calls stay eager stream work in between;
``'host'``  the same through host memory and gloo (tests on one GPU / CPU, and the last-resort fallback of the bench).
"""
import os

import numpy as np

from .partition import build_partition, strip_owner

__all__ = ['HaloExchanger', 'P2PHalo', 'DistributedSwe2d', 'state_digest']
_TEST_FAILED_ONCE = []


class HaloExchanger(object):
    """Neighbour exchange of [n][9] cell states between ranks; tensors may live on the CPU (gloo) or the GPU (RCCL)."""

    def __init__(self, part, device, host_staged=False, width=None, group=None):
        import torch
        self.part = part
        self.group = group
        # doubles per cell: u, v, eta at every node (state) or one value per node (a tracer)
        self.w = w = 3*int(part.cells.shape[1]) if width is None else int(width)
        self.send_buf = torch.zeros(max(1, len(part.send_cells))*w, dtype=torch.float64, device=device)
        self.recv_buf = torch.zeros(max(1, len(part.recv_cells))*w, dtype=torch.float64, device=device)
        # gloo cannot move device memory: stage through the host (test path only; RCCL sends device buffers directly)
        self.host_staged = host_staged
        if host_staged:
            self._send_h = torch.zeros_like(self.send_buf, device='cpu')
            self._recv_h = torch.zeros_like(self.recv_buf, device='cpu')

    def start(self):
        """Post the sends/receives for the packed send buffer; returns request handles."""
        import torch
        import torch.distributed as dist
        sbuf, rbuf = self.send_buf, self.recv_buf
        if self.host_staged:
            torch.cuda.current_stream().synchronize()
            self._send_h.copy_(self.send_buf)
            sbuf, rbuf = self._send_h, self._recv_h
        ops = []
        w = self.w
        for q in self.part.peers:
            if q in self.part.recv:
                off, cnt = self.part.recv[q]
                ops.append(dist.P2POp(dist.irecv, rbuf[w*off:w*(off + cnt)], q, group=self.group))
            if q in self.part.send:
                off, cnt = self.part.send[q]
                ops.append(dist.P2POp(dist.isend, sbuf[w*off:w*(off + cnt)], q, group=self.group))
        return dist.batch_isend_irecv(ops) if ops else []

    def finish(self, reqs):
        for r in reqs:
            r.wait()
        if self.host_staged:
            self.recv_buf.copy_(self._recv_h)


class P2PHalo(object):
    """Set-up of the peer-to-peer exchange of one handle: landing zone, IPC handles swapped through ``group`` (any backend;
    an object all-gather at set-up only), peers' zones mapped, segments connected.  Channel 0 = SWE state, 1 + t = tracer t;
    on triangles a last channel of 18 doubles per cell carries the granules of the flow kernel's in-launch exchange
    (csrc/swe2d_flow.h, FX kernels: nine 16-byte {value, push number} pairs per cell)."""

    def __init__(self, dev, part, rank, world, n_tracers=0, group=None, local_error=None):
        """``local_error``: what already went wrong on this rank before the handshake (the handle could not be built, ...): it is
        carried into the first all-gather, so that every rank leaves the handshake together with the same exception instead of
        this rank going on alone while its peers wait in ``all_gather_object``."""
        import torch.distributed as dist
        k = int(part.cells.shape[1])
        self.dev, self.n_channels = dev, 1 + int(n_tracers) + (1 if k == 3 else 0)

        def gather(obj):
            out = [None]*world
            if world > 1:
                dist.all_gather_object(out, obj, group=group)
            else:
                out[0] = obj
            return out

        def raise_if_any(errors, what):
            bad = ['rank {:d}: {:}'.format(r, e) for r, e in enumerate(errors) if e]
            if bad:
                raise RuntimeError('peer-to-peer halo {:} failed ({:})'.format(what, '; '.join(bad)))
        # every step that can fail locally is followed by an all-gather of the error strings, so that all ranks give up
        # together instead of leaving the others waiting in a collective
        mine = {'pid': os.getpid(), 'error': None}
        try:
            if local_error is not None:
                raise RuntimeError('before the handshake: {:}'.format(local_error))
            dev.p2p_create([3*k] + [k]*int(n_tracers) + ([18] if k == 3 else []))
            handle, base, kind = dev.p2p_export()
            self.zone_kind = {1: 'uncached', 2: 'fine-grained', 3: 'device'}.get(kind, '?')
            mine.update(handle=handle, base=base, n_recv=int(len(part.recv_cells)),
                        recv={int(q): (int(o), int(c)) for q, (o, c) in part.recv.items()})
        except Exception as e:
            mine['error'] = str(e)
        info = gather(mine)
        raise_if_any([i['error'] for i in info], 'set-up')
        err = None
        try:
            bases, s_off, s_cnt, r_off, r_flag, r_n = [], [], [], [], [], []
            for q in sorted(part.send):
                off, cnt = part.send[q]
                theirs = info[q]
                if rank not in theirs['recv'] or theirs['recv'][rank][1] != cnt:
                    raise RuntimeError('halo lists of ranks {:d} and {:d} do not match'.format(rank, q))
                # a peer inside this process (single-process tests) is addressed directly, another process through IPC
                bases.append(theirs['base'] if theirs['pid'] == mine['pid'] else dev.p2p_open(theirs['handle']))
                s_off.append(off)
                s_cnt.append(cnt)
                r_off.append(theirs['recv'][rank][0])
                r_flag.append(sorted(theirs['recv']).index(rank))
                r_n.append(theirs['n_recv'])
            dev.p2p_connect(bases, s_off, s_cnt, r_off, r_flag, r_n, n_from=len(part.recv))
        except Exception as e:
            err = str(e)
        # also the barrier: nobody pushes before every zone is mapped and zeroed
        raise_if_any(gather(err), 'connection')

    def timeouts(self):
        return self.dev.p2p_status(self.n_channels)[2]


def coupled_halo_depth(exchange_every, use_limiter, stages_per_step=3):
    """Ghost layers a coupled cycle of ``exchange_every`` steps with ONE exchange at its end needs (vertex-adjacent layers with
    the limiter, facet-adjacent ones without): a step costs the shallow water state one layer per stage (three for SSPRK33, one
    for ForwardEuler); the tracer stages read the new velocity one layer further out than they write, so the tracer ends a step
    ``stages_per_step`` (with the limiter: one more) layers inside the shallow water state it started from."""
    sps = int(stages_per_step)
    return (sps + 1 if use_limiter else sps)*int(exchange_every) + sps


def coupled_cycle_schedule(part, n_steps, n_tracers, use_limiter, tracer_only=False, stages_per_step=3):
    """Launch list of ``n_steps`` coupled steps (GeneralCoupledTimeIntegrator2D.advance, coupled_timeintegrator_2d.py:93-113:
    shallow water step, every tracer with the updated velocity, limiter) between two exchanges of ALL fields, on the shrinking
    cell ranges that stay exact: ('swe', stage, cell_end) | ('tracer', i_tracer, stage, cell_end) | ('limit', i_tracer, cell_end)
    | ('swe_done',) after the last shallow water stage of the cycle (its exchange can start there).  Validity is counted in
    ghost layers: a stage is exact on one layer less than its input, the tracer stages also need the velocity on the layer
    they read, the limiter needs the cell means one (vertex) layer further out.  ``stages_per_step``: 3 SSPRK33, 1 ForwardEuler."""
    sps = int(stages_per_step)
    depth = coupled_halo_depth(n_steps, use_limiter, sps)
    if depth > len(part.layer_sizes):
        raise ValueError('a coupled cycle of {:d} steps needs {:d} ghost layers, the partition has {:d}'.format(
            n_steps, depth, len(part.layer_sizes)))
    v_swe = v_t = depth
    ops = []
    for k in range(n_steps):
        if not tracer_only:
            for g in range(sps):
                v_swe -= 1
                ops.append(('swe', g, part.layer_end(v_swe)))
            if k == n_steps - 1:
                ops.append(('swe_done',))
        v_in = min(v_t, v_swe)
        for i in range(n_tracers):
            for g in range(sps):
                ops.append(('tracer', i, g, part.layer_end(v_in - 1 - g)))
        v_t = v_in - sps
        if use_limiter:
            v_t -= 1
            for i in range(n_tracers):
                ops.append(('limit', i, part.layer_end(v_t)))
    assert v_t >= 0 and v_swe >= 0
    return ops


class DistributedSwe2d(object):
    """SSPRK33 on a strip-partitioned mesh, one rank per GPU."""

    def __init__(self, mesh, bathymetry_vertex, dt, rank, world_size, device_id, owner=None, host_staged=False,
                 n_tracers=0, use_limiter=True, tracer_only=False, exchange_every=1, overlap_stages=0,
                 graph_mode=None, stepper='SSPRK33', exchange=None, split_last_stage=True, group=None, partition=None,
                 combined_exchange=False, flow=None, flow_exchange=None, device_cls=None, **opts):
        """``n_tracers`` > 0: the coupled step of GeneralCoupledTimeIntegrator2D.advance (coupled_timeintegrator_2d.py:
        93-113) on the partition - shallow water step, then every tracer with the updated velocity, then the limiter.
        The vertex-based limiter needs every cell around a vertex, so coupled runs with the limiter use four ghost layers
        built by VERTEX distance: the ghosts receive the neighbours' unlimited values once per step and every rank limits
        its owned cells and layers 1-3 redundantly (same means, same bounds => bitwise the owner's result); layer 4 is
        only ever read by the limiter.

        Coupled runs with ``exchange_every`` = m > 1 (or ``combined_exchange``): m coupled steps between two exchanges of ALL
        fields (shallow water state and every tracer, one exchange instead of 1 + n_tracers per step) on 4m + 3 vertex-adjacent
        ghost layers (3m + 3 facet-adjacent ones without the limiter), every launch on the largest cell range that is still
        exact (coupled_cycle_schedule); the shallow water exchange starts after the cycle's last shallow water stage and
        overlaps the tracer work.  Bitwise the results of the per-step exchanges.

        ``exchange_every`` = m > 1 (shallow water only): 3m facet-adjacent ghost layers and ONE exchange every m time
        steps - stage g = 0..3m-1 of a cycle updates the owned cells and the first 3m-1-g layers, so the redundant work
        shrinks by one layer per stage (strips of the 1 M-triangle bench mesh at 8 ranks, m = 4: 2 x 12 layers of ~500
        cells, on average +5 % cell updates) while the latency of the exchange is paid once per m steps.  Results are
        bitwise those of m = 1.

        ``overlap_stages`` = j > 0 (shallow water only): while an exchange is in flight the next cycle already runs its
        first j stages on the owned cells that cannot see ghost data yet - stage g of a cycle reaches ghost cells only
        through cells at distance <= g + 1 from the cut, and the owned cells at distance >= d are a prefix of the local
        numbering (LocalPartition.owned_prefix) - and completes those stages on the remaining wedge (cells at distance
        <= g + 1 and the ghost layers) after the unpack.  Disjoint read / write sets (a late stage g reads distance
        <= g + 2, an early stage g' > g writes distance >= g' + 2), bitwise the same result.
        On coupled runs with the combined exchange: the first j shallow water stages of the NEXT cycle (at most those of its
        first step) run while the last tracer's exchange of this cycle is in flight - the shallow water state has already
        been exchanged by then (it travels during the tracer stages) and does not depend on the tracers, so the stages run on
        their full ranges and nothing is left to complete afterwards.  Bitwise the same result.

        ``flow``: the 3m stages of a cycle as ONE launch without grid-wide barriers (csrc/swe2d_flow.h: a 64-cell block starts
        its next stage as soon as the blocks around it have finished the previous one; bit for bit the stage launches),
        followed by the exchange: None = where the kernel covers the partition (every block resident at once; shallow water
        only, triangles without wetting-drying / viscosity, no ``overlap_stages``), True = required, False = never.

        ``flow_exchange`` (with ``flow`` and the peer-to-peer transport): the exchange INSIDE the flow launch - up to 64 cycles per
        launch, a cycle starts by reading the ghost cells from the landing zone and ends by pushing the send cells into the
        peers' zones, cell by cell as tagged granules (csrc/swe2d_flow.h, FX kernels: no flag per rank, a ghost cell is ready as
        soon as the peer's block that owns it has finished); the push of an advance's last cycle is received by one unpack kernel
        at its end.  None = where it applies, False = flow launch + push + unpack kernels per cycle.

        ``exchange``: 'p2p' | 'rccl' | 'host' (module docstring); default 'host' if ``host_staged`` else 'rccl'.
        ``partition``: a LocalPartition already built for this rank with the halo depth the other arguments imply.
        ``split_last_stage`` = False: the last stage of a cycle is ONE launch over the owned cells followed by the send
        (one launch fewer per cycle, the exchange latency is exposed) instead of send cells first / interior during the
        exchange.  ``group``: process group of the exchange and the reductions (default: the world group).
        ``device_cls``: the class of the per-rank handle (default ``Swe2dDevice`` = the HIP library; tests/cpu_device.py passes a
        host stand-in with ``is_host = True`` to run this class's launch schedule and exchange logic without a GPU)."""
        import torch
        from .device import Swe2dDevice
        device_cls = Swe2dDevice if device_cls is None else device_cls
        self._on_gpu = not getattr(device_cls, 'is_host', False)
        self.rank, self.world = rank, world_size
        self.group = group
        # default: strips (<= 2 peers = one xGMI link each); pass owner=rcb_owner(mesh, n) for compact parts of a general mesh
        if partition is None:
            owner = strip_owner(mesh, world_size) if owner is None else owner
        self.use_limiter = bool(use_limiter) and n_tracers > 0
        self.tracer_only = bool(tracer_only)
        if stepper not in ('SSPRK33', 'ForwardEuler'):
            raise ValueError("stepper must be 'SSPRK33' or 'ForwardEuler'")
        # ForwardEuler (the other explicit entry of the steppers table): one stage per step, one ghost layer per step
        self.stages_per_step = 3 if stepper == 'SSPRK33' else 1
        if self.stages_per_step == 1 and int(overlap_stages) > 0:
            raise ValueError('ForwardEuler on partitions: no overlap_stages')
        self.exchange_every = m = int(exchange_every)
        self.overlap_stages = int(overlap_stages)
        if m < 1:
            raise ValueError('exchange_every must be >= 1')
        # coupled runs: exchange_every = 1 exchanges after the shallow water step and after every tracer step (four ghost
        # layers); exchange_every = m > 1 (or combined_exchange) runs m coupled steps between two exchanges of ALL fields
        self.coupled_cycles = n_tracers > 0 and (m > 1 or bool(combined_exchange) or self.stages_per_step == 1)
        if self.overlap_stages > 0 and n_tracers > 0 and (not self.coupled_cycles or tracer_only):
            raise ValueError('overlap_stages on coupled runs needs the combined exchange (exchange_every > 1 or '
                             'combined_exchange=True) and a shallow water step to overlap with (not tracer_only)')
        if not 0 <= self.overlap_stages <= 3*m - 1:
            raise ValueError('overlap_stages must be in 0 .. 3*exchange_every - 1')
        self.exchange = exchange or ('host' if host_staged else 'rccl')
        if self.exchange not in ('p2p', 'rccl', 'host'):
            raise ValueError("exchange must be 'p2p', 'rccl' or 'host'")
        self.split_last_stage = bool(split_last_stage)
        if not self.split_last_stage and self.overlap_stages:
            raise ValueError('overlap_stages needs split_last_stage')
        if partition is not None:
            self.part = partition                   # built by the caller with the matching halo depth (bench: reused)
        elif self.coupled_cycles:
            self.part = build_partition(mesh, owner, rank, halo_depth=coupled_halo_depth(m, self.use_limiter, self.stages_per_step),
                                        adjacency='vertex' if self.use_limiter else 'facet')
        elif self.stages_per_step == 1:
            self.part = build_partition(mesh, owner, rank, halo_depth=m)
        elif m > 1:
            self.part = build_partition(mesh, owner, rank, halo_depth=3*m)
        elif self.use_limiter:
            self.part = build_partition(mesh, owner, rank, halo_depth=4, adjacency='vertex')
        else:
            self.part = build_partition(mesh, owner, rank)
        p = self.part
        if self._on_gpu:
            torch.cuda.set_device(device_id)
            self.torch_device = torch.device('cuda', device_id)
        else:
            self.torch_device = torch.device('cpu')
            if self.exchange != 'host':
                raise ValueError("a host stand-in device exchanges through host memory: exchange='host'")
        self._flow_request = flow
        # opt-in periodic check of the fast path (THETIS_AMD_VERIFY_EVERY = n, or verify_every=n): see _advance_verified
        ve = opts.pop('verify_every', None)
        self._verify_every = int(ve if ve is not None else os.environ.get('THETIS_AMD_VERIFY_EVERY', '0') or 0)
        self._v_snapshot, self._v_steps = None, 0
        self._replaying = False                  # a verification replay runs: stage launches, host-staged exchange, no graphs
        self._distrust = 0                       # 0 as configured | 1 after a mismatch: no dataflow launches | 2: and the exchange through the host
        self.verify_report = {'windows': 0, 'mismatches': 0, 'bad_ranks': []}
        self._host_halos = {}
        self._flowx_request = flow_exchange
        self._flow_now = None                    # the rank-collective decision of the current advance() (see _decide_flow)
        self._flowx_now = False
        self._flow_epoch, self._decided_epoch = 0, None
        self._no_exchange = False                # measurement only (bench: config.exchange_time_fraction): the schedule without its exchanges
        self._shared_device = None               # do several ranks step on this GPU?  (found out at the first automatic decision)
        self.dev = None
        self.tids = []
        self.halo = self.thalo = self.p2p = None
        try:
            self._build_handle(device_cls, p, bathymetry_vertex, dt, device_id, n_tracers, opts)
            local_error = None
        except Exception as e:
            # the peer-to-peer handshake below is collective: a rank that failed here must still enter it (and leave it with
            # everybody else) - its peers would wait in all_gather_object while it went on to the caller's next collective
            if self.exchange != 'p2p' or world_size == 1:
                raise
            local_error = '{:}: {:}'.format(type(e).__name__, (str(e).strip().splitlines() or [''])[0])
        if self.exchange == 'p2p':
            try:
                self.p2p = P2PHalo(self.dev, p, rank, world_size, n_tracers=n_tracers, group=group, local_error=local_error)
            except Exception:
                if self.dev is not None:
                    self.dev.close()
                raise
        else:
            staged = self.exchange == 'host' and self._on_gpu       # a host stand-in's buffers are CPU tensors already
            self.halo = HaloExchanger(p, self.torch_device, host_staged=staged, group=group)
            self.thalo = HaloExchanger(p, self.torch_device, host_staged=staged, width=p.cells.shape[1], group=group) if n_tracers else None
        self._finish_init(graph_mode)

    def _build_handle(self, device_cls, p, bathymetry_vertex, dt, device_id, n_tracers, opts):
        """Everything of the constructor that is local to this rank: the handle, its halo lists, flow order, tracers."""
        flow = self._flow_request
        if os.environ.get('THETIS_AMD_TEST_FAIL_HANDLE_RANK') == str(self.rank) and not _TEST_FAILED_ONCE:
            _TEST_FAILED_ONCE.append(1)                  # tests: this rank's FIRST handle of the process cannot be built
            raise RuntimeError('THETIS_AMD_TEST_FAIL_HANDLE_RANK')
        self.dev = device_cls(p, np.asarray(bathymetry_vertex)[p.vertex_global], dt, device_id=device_id,
                              n_owned=p.n_owned, boundary_len=p.boundary_len, ranges=p.reorder_ranges(), **opts)
        self.dev.halo_setup(p.send_cells, p.recv_cells)
        if os.environ.get('THETIS_AMD_TEST_TEAR'):         # tests, -DSWE_FLOW_TEAR builds: "microseconds:every" - granule stores in two halves
            us, ev = (int(v) for v in os.environ['THETIS_AMD_TEST_TEAR'].split(':'))
            if self.dev.lib.swe2d_debug_flow_tear(self.dev.h, -2, us, ev, 1) != 0:
                raise RuntimeError('THETIS_AMD_TEST_TEAR needs the -DSWE_FLOW_TEAR build of the library')
        if flow is not False and self.dev.npc == 3 and self._on_gpu:
            # the flow kernel's blocks: all local cells (owned + ghost layers) in one locality order, so that a ghost cell
            # shares its block with the cells it touches (in the device numbering - ghost layers appended layer by layer - a
            # block of ghost cells has more rim facets than the kernel's staging area holds)
            from . import ordering
            blocks = os.environ.get('THETIS_AMD_FLOW_BLOCKS', '1') != '0'        # (0: the tile order of the device numbering, A/B)
            self.dev.flow_set_order((ordering.flow_block_order if blocks else ordering.auto_cell_order)(p, 0, p.num_cells))
        if self.dev.npc == 3 and self._on_gpu:
            # the tiles of the fused stage pair (stages 1 + 2 of a step in one launch where the kernel covers the partition:
            # _cycle_before_exchange): cut from an order in which the ghost cells sit in the tiles of the owned cells they touch;
            # built here, never inside a graph capture
            from . import ordering
            order = ordering.fused_tile_order(p)
            if order is not None:
                self.dev.fused_set_order(order)
            self.dev.fused_pair_info()
            # ... and the two-ring tiles of a whole step in one launch (swe2d_solve_step_cells: pairs of steps of a cycle, see
            # _cycle_before_exchange): the 11 x 8-quad patches of the parent mesh, over owned and ghost cells alike
            tt = os.environ.get('THETIS_AMD_TRIPLE_TILE', '11,8')
            tiles = ordering.triple_tile_order(p, *(int(v) for v in tt.split(','))) if tt != '0' else None
            if tiles is not None:
                self.dev.fused_set_triple_tiles(*tiles)
            self.dev.fused_step_info()
        self._ranges = [p.stage_range(i) for i in range(3)]
        self.tids = [self.dev.add_tracer() for _ in range(n_tracers)]

    def _finish_init(self, graph_mode):
        import torch
        self.xstream = None
        if self._on_gpu:
            self.stream = torch.cuda.Stream(device=self.torch_device)
            self.dev.set_stream(self.stream.cuda_stream)
            if self.p2p is not None and os.environ.get('THETIS_AMD_P2P_SIDE_STREAM', '0') == '1':
                # OPT-IN (measured slower, DESIGN_ANNEX.md A5): the exchange kernels (push: 5-6 us, wait + unpack: 5-6 us per cycle,
                # a few thousand cells each) on a stream of their own - forked off after the send cells' stage, joined before the
                # next reader of the ghost cells, they run beside the stage kernels of the interior instead of between them (same
                # disjoint read / write sets as an exchange in flight on the other transports; bitwise the same results)
                self.xstream = torch.cuda.Stream(device=self.torch_device)
                self.dev.set_exchange_stream(self.xstream.cuda_stream)
                self._ev_fork = torch.cuda.Event()
                self._ev_join = torch.cuda.Event()
                self._ev_pushed = torch.cuda.Event()
        else:
            self.stream = None
        self.graph = None
        self.graph_steps = 0
        # 'cycle' (default): HIP graphs of the kernel sequences of a cycle - with 'p2p' the whole cycle incl. the exchange
        # kernels, otherwise the sequences before / during the exchange with the RCCL calls launched eagerly in between
        # (nothing of RCCL inside a capture); 'full': the whole K-step loop (with 'rccl': incl. the RCCL calls, which needs
        # RCCL point-to-point capture to work on the node) in ONE graph; 'none': eager launches
        self.graph_mode = graph_mode or os.environ.get('THETIS_AMD_GRAPH_MODE', 'cycle')
        if self.stages_per_step == 1:
            self.graph_mode = 'none'        # the step swaps the state buffers: kernel arguments change from replay to replay
        if self.graph_mode not in ('cycle', 'full', 'none'):
            raise ValueError("graph_mode / THETIS_AMD_GRAPH_MODE must be 'cycle', 'full' or 'none'")
        if self.exchange == 'host' and self.graph_mode == 'full':
            self.graph_mode = 'cycle'       # a host-staged exchange synchronises the stream: never inside a capture
        if not self._on_gpu:
            self.graph_mode = 'none'
        self._cycle_graphs = {}

    def close(self):
        self.dev.close()

    def _flow_local(self):
        """This rank's own answer to "does a cycle run as one dataflow launch?" (see ``flow``); evaluated per call: the device
        configuration (source terms, viscosity, wetting-drying) may be set after construction."""
        if self._flow_request is False or os.environ.get('THETIS_AMD_FLOW') == '0' or not self._on_gpu or self._distrust:
            return False
        plain = (self.stages_per_step == 3 and not self.tids and not self.tracer_only and self.overlap_stages == 0
                 and 3*self.exchange_every <= 384)
        if not plain or not self.dev.flow_supported():
            if self._flow_request is True:
                raise ValueError('flow=True: the flow kernel covers SSPRK33 shallow-water-only runs on triangles without wetting-drying, '
                                 'viscosity and overlap_stages, on partitions whose 64-cell blocks are all resident at once')
            return False
        return True

    def _ranks_share_a_device(self):
        """Do two ranks of the group step on the same GPU?  (one all-gather, once.)  The flow kernel needs every block of a launch
        resident at once; its capacity check assumes the device is this rank's alone."""
        if self._shared_device is None:
            import socket
            import torch
            import torch.distributed as dist
            prop = torch.cuda.get_device_properties(self.torch_device)
            ident = (socket.gethostname(), str(getattr(prop, 'uuid', '')), getattr(prop, 'pci_bus_id', -1),
                     getattr(prop, 'pci_device_id', -1), getattr(prop, 'pci_domain_id', -1),
                     os.environ.get('HIP_VISIBLE_DEVICES', os.environ.get('ROCR_VISIBLE_DEVICES', '')), self.torch_device.index)
            everyone = [None]*self.world
            dist.all_gather_object(everyone, ident, group=self.group)
            self._shared_device = len(set(everyone)) < len(everyone)
        return self._shared_device

    def _flowx_local(self):
        """Can the exchange run INSIDE this rank's flow launches?  The FX kernels push a cell to at most two peers (strips, RCB
        corners); a halo deeper than a part is wide sends cells to more - then the flow launch is followed by the exchange kernels."""
        if self.p2p is None or self._flowx_request is False or os.environ.get('THETIS_AMD_FLOWX') == '0':
            return False
        if self.dev.flow_supported() not in (1, 2):
            return False
        sc = np.asarray(self.part.send_cells)
        return not len(sc) or int(np.bincount(sc).max()) <= 2

    def _decide_flow(self):
        """The decision all ranks take TOGETHER at the start of an ``advance`` (collective when world > 1 and the choice is not
        ``flow=False``): a rank whose partition the kernel does not cover (one ghost side more than its neighbour, just over the
        resident capacity) would otherwise wait for stage-launch halos while its peers push flow granules, and every wait would
        run into its timeout.  Automatic choice (``flow=None``): only where every rank is covered and no two ranks share a GPU,
        agreed anew at the first ``advance`` after a configuration change that matters (``config_changed``); ``flow=True``:
        agreed once, a rank that is not covered makes all ranks raise.  The exchange inside the launches (``flow_exchange``) is
        part of the same agreement."""
        import torch.distributed as dist
        if self._flow_request is False or self.world == 1:
            self._flow_now = self._flow_local()
            self._flowx_now = self._flow_now and self._flowx_local()
            return self._flow_now
        if self._flow_request is True:
            if self._flow_now is None:
                try:
                    ok = self._flow_local()
                except ValueError:
                    ok = False
                okx = ok and self._flowx_local()
                both = self._all_reduce([1.0 if ok else 0.0, 1.0 if okx else 0.0], dist.ReduceOp.MIN)
                if both[0] < 0.5:
                    raise ValueError('flow=True: the flow kernel does not cover the partition of every rank (rank {:d}: {:})'.format(
                        self.rank, 'covered' if ok else 'not covered'))
                if self._flowx_request is True and both[1] < 0.5:
                    raise ValueError('flow_exchange=True: the exchange inside the flow launches needs the peer-to-peer transport and '
                                     'cells that go to at most two peers, on every rank (rank {:d}: {:})'.format(
                                         self.rank, 'possible' if okx else 'not possible'))
                self._flow_now, self._flowx_now = True, bool(both[1] > 0.5)
            return self._flow_now
        # (every rank takes part in the all-gather behind _ranks_share_a_device, whatever its own answer: a rank that skipped it
        #  because its partition is not covered would leave the others waiting in it - seen with the eight strips of the 1 M-triangle
        #  mesh, whose end ranks are covered and whose middle ranks, with two ghost sides, are not)
        if self._decided_epoch == self._flow_epoch and self._flow_now is not None:
            return self._flow_now               # nothing that decides coverage has changed since the ranks last agreed
        shared = self._ranks_share_a_device()
        ok = self._flow_local() and not shared
        okx = ok and self._flowx_local()
        both = self._all_reduce([1.0 if ok else 0.0, 1.0 if okx else 0.0], dist.ReduceOp.MIN)
        self._flow_now, self._flowx_now = bool(both[0] > 0.5), bool(both[1] > 0.5)
        self._decided_epoch = self._flow_epoch
        return self._flow_now

    def config_changed(self):
        """To be called (by every rank alike) after a change of the handle's configuration that decides whether the flow kernel covers it
        - wetting-drying, viscosity: the next ``advance`` lets the ranks agree anew.  ``PartitionedDevice`` does; a caller that
        configures ``self.dev`` directly after the first ``advance`` must."""
        self._flow_epoch += 1

    @property
    def flow(self):
        """True when a cycle runs as one dataflow launch (see ``flow``): the ranks' common decision of the last ``advance``, before
        the first one this rank's own answer."""
        if self._replaying or self._distrust:
            return False
        return self._flow_local() if self._flow_now is None else self._flow_now

    @property
    def flow_exchange(self):
        """True when the exchange runs inside the flow launches (see ``flow_exchange``): part of the ranks' common decision."""
        if self._no_exchange or self._replaying or self._distrust:
            return False
        if self._flow_now is None:
            return self._flow_local() and self._flowx_local()
        return bool(self._flow_now and self._flowx_now)

    def _steps_flow_exchange(self, n_steps, graphed):
        """``n_steps`` time steps as flow launches with the exchange inside: up to 64 cycles (384 stages) per launch, a shorter
        trailing cycle in a launch of its own, and one unpack kernel for the last push."""
        dev, p, m = self.dev, self.part, self.exchange_every
        if graphed:
            dev.flow_prepare_exchange()            # tables: never inside the capture of the first launch
        full, rem = divmod(n_steps, m)
        per_launch = max(1, min(64, 384//(3*m), int(os.environ.get("THETIS_AMD_FLOWX_CYCLES", "64"))))
        ends = [p.stage_range(g, depth=3*m) for g in range(3*m)]
        while full > 0:
            nc = min(per_launch, full)
            self._launch(('X', nc, m), lambda nc=nc: dev.solve_flow_exchange(nc, ends), graphed)
            full -= nc
        if rem:
            ends_r = [p.stage_range(g, depth=3*rem) for g in range(3*rem)]
            self._launch(('X', 1, rem), lambda: dev.solve_flow_exchange(1, ends_r), graphed)
        self._launch(('XU',), lambda: dev.flow_unpack_pending(), graphed)

    def _cycle_swe_flow(self, n_steps, graphed):
        """``n_steps`` time steps = 3 n_steps stages on the shrinking ranges in ONE launch, then the exchange."""
        dev, p = self.dev, self.part
        n = 3*n_steps
        ends = [p.stage_range(g, depth=n) for g in range(n)]
        if self.p2p is not None:
            def whole_cycle():
                dev.solve_flow(ends)
                self._receive(0, 0, self._send(0, 0))
            return self._launch(('W', n_steps), whole_cycle, graphed)
        self._launch(('WA', n_steps), lambda: dev.solve_flow(ends), graphed)
        self._receive(0, 0, self._send(0, 0))

    def set_tracer_global(self, i_tracer, nodal):
        self.dev.tracer_set_state(self.tids[i_tracer], np.asarray(nodal)[self.part.local_to_global])

    def get_tracer_owned(self, i_tracer):
        n = self.part.n_owned
        return self.part.local_to_global[:n], self.dev.tracer_get_state(self.tids[i_tracer])[:n]

    def _all_reduce(self, values, op):
        """Reduce a few doubles over the ranks: on the device with RCCL, on the host when the exchange avoids RCCL."""
        import torch
        import torch.distributed as dist
        on_host = self.exchange != 'rccl'
        t = torch.tensor(list(values), dtype=torch.float64, device='cpu' if on_host else self.torch_device)
        if self.world > 1:
            dist.all_reduce(t, op=op, group=self.group)
        return t.cpu().numpy()

    def _all_reduce_int(self, values):
        """Exact sum of int64 values over the ranks."""
        import torch
        import torch.distributed as dist
        on_host = self.exchange != 'rccl'
        t = torch.tensor(np.asarray(values, dtype=np.int64).ravel(), dtype=torch.int64, device='cpu' if on_host else self.torch_device)
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def tracer_diagnostics(self, i_tracer):
        """Global {int T*H dx, int T dx, min, max} of tracer ``i_tracer``: the integrals as order-independent limb sums added over
        the ranks as integers (include/swe2d.h, swe2d_diagnostics_limbs) - the doubles of the single-device run, bit for bit."""
        import torch.distributed as dist
        limbs, mm = self.dev.tracer_diagnostics_limbs(self.tids[i_tracer])
        total = self._all_reduce_int(limbs).reshape(2, -1)
        m = self._all_reduce([mm[0], -mm[1]], dist.ReduceOp.MIN)
        return np.array([self.dev.limbs_to_double(total[0]), self.dev.limbs_to_double(total[1]), m[0], -m[1]])

    def set_state_global(self, uv, eta):
        g = self.part.local_to_global
        self.dev.set_state(uv[g], eta[g])

    def get_state_owned(self):
        """(global ids, uv, eta) of the owned cells."""
        self._check_exchange()
        uv, eta = self.dev.get_state()
        n = self.part.n_owned
        return self.part.local_to_global[:n], uv[:n], eta[:n]

    # ---- the exchange: send = pack + post (or push), receive = wait + unpack
    def _through_host(self):
        return self._replaying or self._distrust >= 2

    def _host_halo(self, channel):
        """The host-staged exchanger of a channel, built on first use (verification replays of a run whose transport is not 'host')"""
        w = 3*self.dev.npc if channel == 0 else self.dev.npc
        if w not in self._host_halos:
            if self.exchange == 'host':
                self._host_halos[w] = self.halo if channel == 0 else self.thalo
            else:
                self._host_halos[w] = HaloExchanger(self.part, self.torch_device, host_staged=self._on_gpu, width=w, group=self._host_group())
        return self._host_halos[w]

    def _host_group(self):
        """a group that moves CPU tensors: the run's own for 'p2p' / 'host' (gloo), the world group for 'rccl' (cpu:gloo,cuda:nccl)"""
        return self.group

    def _send(self, channel, i_buffer):
        dev = self.dev
        if self._no_exchange:
            return None
        if self._through_host():
            halo = self._host_halo(channel)
            if channel == 0:
                dev.halo_pack(i_buffer, halo.send_buf.data_ptr())
            else:
                dev.tracer_halo_pack(self.tids[channel - 1], i_buffer, halo.send_buf.data_ptr())
            return halo.start()
        if self.p2p is not None:
            if self.xstream is not None:                 # fork: the push follows everything enqueued so far
                self._ev_fork.record(self.stream)
                self.xstream.wait_event(self._ev_fork)
            dev.p2p_push(channel, i_buffer)
            if self.xstream is not None:
                self._ev_pushed.record(self.xstream)
            return None
        if channel == 0:
            dev.halo_pack(i_buffer, self.halo.send_buf.data_ptr())
            return self.halo.start()
        dev.tracer_halo_pack(self.tids[channel - 1], i_buffer, self.thalo.send_buf.data_ptr())
        return self.thalo.start()

    def _receive(self, channel, i_buffer, reqs):
        dev = self.dev
        if self._no_exchange:
            return
        if self._through_host():
            halo = self._host_halo(channel)
            halo.finish(reqs)
            if channel == 0:
                dev.halo_unpack(i_buffer, halo.recv_buf.data_ptr())
            else:
                dev.tracer_halo_unpack(self.tids[channel - 1], i_buffer, halo.recv_buf.data_ptr())
            return
        if self.p2p is not None:
            dev.p2p_wait_unpack(channel, i_buffer)
            if self.xstream is not None:                 # join: whatever comes next sees the ghost cells
                self._ev_join.record(self.xstream)
                self.stream.wait_event(self._ev_join)
        elif channel == 0:
            self.halo.finish(reqs)
            dev.halo_unpack(i_buffer, self.halo.recv_buf.data_ptr())
        else:
            self.thalo.finish(reqs)
            dev.tracer_halo_unpack(self.tids[channel - 1], i_buffer, self.thalo.recv_buf.data_ptr())

    def _step(self):
        if not self.tracer_only:
            self._cycle_swe(1)
        for i in range(len(self.tids)):
            self._step_tracer(i)

    def _step_tracer(self, i):
        """One tracer SSPRK33 step with the (already exchanged) updated velocity, same ranges and overlap as the shallow
        water step, then the limiter on owned cells + ghost layers 1-3."""
        dev, p, tid = self.dev, self.part, self.tids[i]
        dev.tracer_solve_stage_cells(tid, 0, 0, self._ranges[0])
        dev.tracer_solve_stage_cells(tid, 1, 0, self._ranges[1])
        dev.tracer_solve_stage_cells(tid, 2, p.n_interior, p.n_owned)
        reqs = self._send(1 + i, 0)
        dev.tracer_solve_stage_cells(tid, 2, 0, p.n_interior)
        self._receive(1 + i, 0, reqs)
        if self.use_limiter:
            dev.tracer_limit_cells(tid, p.layer_end(3))

    def _cycle_swe(self, n_steps, early_done=0, early_next=0, graphed=False):
        """``n_steps`` (<= exchange_every) time steps on shrinking cell ranges, then one exchange.  One step:
        stage 1 on owned + ghost layers 1, 2; stage 2 on owned + layer 1; stage 3 on the owned cells.
        ``early_done``: stages of this cycle whose ghost-independent part ran during the previous exchange;
        ``early_next``: stages of the next cycle to run (ghost-independent part only) during this cycle's exchange."""
        if self.stages_per_step == 1:
            return self._cycle_forward_euler(n_steps)
        if self.flow:
            return self._cycle_swe_flow(n_steps, graphed)
        if self.p2p is not None and not self._through_host():
            # the exchange is two kernels of this library: the whole cycle is one capturable launch sequence
            def whole_cycle():
                self._cycle_before_exchange(n_steps, early_done)
                reqs = self._send(0, 0)
                self._cycle_during_exchange(early_next)
                self._receive(0, 0, reqs)
            return self._launch(('P', n_steps, early_done, early_next), whole_cycle, graphed)
        self._launch(('A', n_steps, early_done), lambda: self._cycle_before_exchange(n_steps, early_done), graphed)
        reqs = self._send(0, 0)                                      # stage 3 leaves the step result in buffer 0
        self._launch(('B', early_next), lambda: self._cycle_during_exchange(early_next), graphed)
        self._receive(0, 0, reqs)

    def _cycle_forward_euler(self, n_steps):
        """``n_steps`` ForwardEuler steps on shrinking ranges (one ghost layer per step), then one exchange."""
        dev, p = self.dev, self.part
        for g in range(n_steps - 1):
            dev.forward_euler_cells(0, p.stage_range(g, depth=n_steps))
            dev.swap_state_buffers()
        dev.forward_euler_cells(p.n_interior, p.n_owned)            # the cells the peers are waiting for (into buffer 1)
        reqs = self._send(0, 1)
        dev.forward_euler_cells(0, p.n_interior)
        dev.swap_state_buffers()
        self._receive(0, 0, reqs)

    def _cycle_before_exchange(self, n_steps, early_done):
        dev, p = self.dev, self.part
        n = 3*n_steps
        assert early_done <= n - 1
        g = 0
        if (self._on_gpu and early_done == 0 and not self.split_last_stage and n >= 6 and self.stages_per_step == 3
                and dev.fused_step_info()[0]):
            # whole steps in one launch each (csrc/swe2d_fuse.h, swe_fuse123_kernel on the partition's two-ring tiles): stage 3 on the
            # step's last range, stages 1 and 2 on the tiles' supersets of theirs, never leaving the chip.  The launch leaves its result in
            # the other state buffer and the two change places - so steps are taken in PAIRS: every cycle, eager or captured, ends on the
            # buffer it started from (a replayed graph uses the pointers of its capture); an odd step goes by the launches below
            while n - g >= 6:
                for _ in range(2):
                    dev.solve_step_cells(p.n_owned if g + 3 == n else p.stage_range(g + 2, depth=n))
                    g += 3
            if g == n:
                return
        while g < n - 1:
            begin = p.owned_prefix(g + 2) if g < early_done else 0
            if g % 3 == 0 and g >= early_done:             # (n is a multiple of 3: stage g + 1 <= n - 2 is in this loop's range too)
                # stages 1 and 2 of a step on their full ranges: one launch by overlapped tiles where the kernel covers the partition
                # (csrc/swe2d_fuse.h; the two stage launches otherwise - swe2d_solve_stage_pair_cells decides, same bits)
                dev.solve_stage_pair_cells(p.stage_range(g, depth=n), p.stage_range(g + 1, depth=n))
                g += 2
                continue
            dev.solve_stage_cells(g % 3, begin, p.stage_range(g, depth=n))
            g += 1
        if self.split_last_stage:
            dev.solve_stage_cells(2, p.n_interior, p.n_owned)       # the cells the peers are waiting for
        else:
            dev.solve_stage_cells(2, 0, p.n_owned)

    def _cycle_during_exchange(self, early_next):
        dev, p = self.dev, self.part
        if self.split_last_stage:
            dev.solve_stage_cells(2, 0, p.n_interior)               # interior cells overlap the exchange
        if early_next and self.xstream is not None:
            # the third early stage writes buffer 0 on cells at distance >= 4 from the cut - send cells among them - which the push on
            # the side stream may still be reading (on one stream it has finished; the other transports pack before this point)
            self.stream.wait_event(self._ev_pushed)
        for g in range(early_next):                                 # ... and so does the ghost-independent part of the next stages
            dev.solve_stage_cells(g % 3, 0, p.owned_prefix(g + 2))

    def _launch(self, key, fn, graphed):
        """Run the kernel sequence ``fn`` now, or replay its HIP graph (captured on first use).  Only kernels of this
        library are captured: RCCL send/recv stay ordinary stream work between two graph launches.  Returns True when
        ``fn`` itself ran in this call (eagerly, or for the capture): its host-side effects have then happened."""
        import torch
        if not graphed:
            fn()
            return True
        ran = False
        g = self._cycle_graphs.get(key)
        if g is None and self.graph_mode == 'cycle':
            try:
                g = torch.cuda.CUDAGraph()
                self.stream.synchronize()
                with torch.cuda.graph(g, stream=self.stream, capture_error_mode='thread_local'):
                    fn()
                self._cycle_graphs[key] = g
                ran = True
            except Exception as e:       # capture refused: eager from now on (same results; a capture executes nothing)
                if self.rank == 0:
                    print('[thetis_amd] HIP graph capture unavailable ({:}); running eagerly'.format(str(e).splitlines()[0]))
                self.graph_mode, g = 'none', None
                torch.cuda.synchronize()
        if g is not None:
            g.replay()
        else:
            fn()
            ran = True
        return ran

    def _coupled_ops(self, n_steps):
        return coupled_cycle_schedule(self.part, n_steps, len(self.tids), self.use_limiter, self.tracer_only, self.stages_per_step)

    def _coupled_early_ops(self, n_steps_next):
        """The shallow water stages of the next cycle that may run during this cycle's last tracer exchange (overlap_stages)"""
        early = []
        for op in self._coupled_ops(n_steps_next):
            if op[0] != 'swe' or len(early) >= self.overlap_stages:
                break
            early.append(op)
        return early

    def _cycle_coupled(self, n_steps, early_done=0, early_next=()):
        """``n_steps`` (<= exchange_every) coupled steps, then one exchange of the shallow water state and every tracer.
        ``early_done``: leading launches of this cycle that already ran during the previous cycle's tracer exchange;
        ``early_next``: launches of the next cycle to run during this cycle's last tracer exchange."""
        dev = self.dev
        reqs = None
        sent = False
        fe = self.stages_per_step == 1         # ForwardEuler: a "stage" is the whole step, buffer 0 -> 1, then the buffers swap
        ops = self._coupled_ops(n_steps)[early_done:]
        nt = len(self.tids)
        # peer-to-peer halos: the shallow water state and the tracers leave together at the end of the cycle, ONE push and ONE
        # wait-and-unpack launch for all channels (swe2d_p2p_push_multi: the channels' kernels side by side) instead of one pair per
        # channel - on a rank of eight of cfg 4 the four exchange kernels of a cycle were 13 of 77 us per step
        # (profiles/r06c_cfg4_rank8_kernel_stats.csv); the shallow water state has nothing to overlap with that the tracers' exchange,
        # which cannot start earlier, does not wait for anyway
        # (not with overlap_stages: the early shallow water stages of the next cycle need this cycle's shallow water ghosts, which the
        #  per-channel exchange has received by then)
        merged = (self.p2p is not None and not self._through_host() and self.xstream is None and not self._no_exchange
                  and self.overlap_stages == 0 and not early_next and 1 + nt <= 4 and hasattr(dev, 'p2p_push_multi'))
        # whole shallow-water steps in one launch each (swe2d_solve_step_cells, see _cycle_before_exchange): in pairs, so that the cycle
        # ends on the state buffer it began on; the tracer stages read the velocity from whichever buffer holds it when they are launched
        n3 = 0
        if (self._on_gpu and not fe and early_done == 0 and not early_next and not self.tracer_only
                and hasattr(dev, 'fused_step_info') and dev.fused_step_info()[0]):
            n3 = 2*(sum(1 for op in ops if op[0] == 'swe' and op[1] == 0)//2)
        i_step = 0
        skip = 0
        for i_op, op in enumerate(ops):
            if skip:                           # the later stages of a step (or pair) that went out as one launch
                skip -= 1
                continue
            if op[0] == 'swe':
                if fe:
                    dev.forward_euler_cells(0, op[2])
                    dev.swap_state_buffers()
                elif (op[1] == 0 and i_step < n3 and i_op + 2 < len(ops) and ops[i_op + 1][:2] == ('swe', 1)
                      and ops[i_op + 2][:2] == ('swe', 2)):
                    dev.solve_step_cells(ops[i_op + 2][2])
                    i_step += 1
                    skip = 2
                elif op[1] == 0 and i_op + 1 < len(ops) and ops[i_op + 1][0] == 'swe' and ops[i_op + 1][1] == 1:
                    # stages 1 and 2 of a step: one launch by overlapped tiles where the kernel covers the partition (csrc/swe2d_fuse.h)
                    dev.solve_stage_pair_cells(op[2], ops[i_op + 1][2])
                    skip = 1
                else:
                    dev.solve_stage_cells(op[1], 0, op[2])
            elif op[0] == 'swe_done':
                if not merged:
                    reqs = self._send(0, 0)           # travels while the tracers step
                sent = True
            elif op[0] == 'tracer':
                dev.tracer_solve_stage_cells(self.tids[op[1]], op[2], 0, op[3])
                if fe:
                    dev.tracer_swap_buffers(self.tids[op[1]])
            else:
                dev.tracer_limit_cells(self.tids[op[1]], op[2])
        if merged:
            channels = ([0] if sent else []) + [1 + i for i in range(nt)]
            dev.p2p_push_multi(channels, [0]*len(channels))
            dev.p2p_wait_unpack_multi(channels, [0]*len(channels))
            return
        if sent:
            self._receive(0, 0, reqs)
        for i in range(nt):
            reqs = self._send(1 + i, 0)
            if i == nt - 1:
                for op in early_next:               # shallow water stages of the next cycle: they read no tracer
                    dev.solve_stage_cells(op[1], 0, op[2])
            self._receive(1 + i, 0, reqs)

    def _steps_eager(self, n_steps, graphed=False):
        m = self.exchange_every
        if self.coupled_cycles:
            cycles = [m]*(n_steps//m) + ([n_steps % m] if n_steps % m else [])
            done = 0
            for i, r in enumerate(cycles):
                # never across advance() calls: after the last cycle buffer 0 holds the step result and nothing is half done
                nxt = self._coupled_early_ops(cycles[i + 1]) if self.overlap_stages and i + 1 < len(cycles) else []
                self._cycle_coupled(r, early_done=done, early_next=nxt)
                done = len(nxt)
            return
        if self.tids or self.tracer_only:
            for _ in range(n_steps):
                self._step()
            return
        if self.flow_exchange and self.stages_per_step == 3:
            return self._steps_flow_exchange(n_steps, graphed)
        cycles = [m]*(n_steps//m) + ([n_steps % m] if n_steps % m else [])
        early = 0
        for i, r in enumerate(cycles):
            # never across advance() calls: after the last cycle buffer 0 holds the result and nothing is half done
            nxt = min(self.overlap_stages, 3*cycles[i + 1] - 1) if i + 1 < len(cycles) else 0
            self._cycle_swe(r, early_done=early, early_next=nxt, graphed=graphed)
            early = nxt

    def _stream_ctx(self):
        import contextlib
        import torch
        return torch.cuda.stream(self.stream) if self._on_gpu else contextlib.nullcontext()

    def advance(self, n_steps, use_graph=True):
        """``n_steps`` SSPRK33 steps (enqueued; call ``synchronize``).  COLLECTIVE: every rank calls it with the same count."""
        if self._verify_every > 0 and self.world > 1 and not self._no_exchange:
            return self._advance_verified(int(n_steps), use_graph)
        return self._advance(n_steps, use_graph)

    # ---- opt-in periodic verification of the fast path (THETIS_AMD_VERIFY_EVERY = n)
    def _local_digest(self):
        import hashlib
        n = self.part.n_owned
        h = hashlib.blake2b(digest_size=16)
        uv, eta = self.dev.get_state()
        h.update(np.ascontiguousarray(uv[:n]).tobytes())
        h.update(np.ascontiguousarray(eta[:n]).tobytes())
        for tid in self.tids:
            h.update(np.ascontiguousarray(self.dev.tracer_get_state(tid)[:n]).tobytes())
        return h.hexdigest()

    def _advance_verified(self, n_steps, use_graph):
        """``advance`` in windows of ``verify_every`` steps (windows run on across calls).  At the end of a window every rank takes the
        blake2b of its owned state, goes back to the window's start and REPLAYS it the conservative way - stage launches, eager, the
        exchange staged through host memory and gloo: nothing of the fast path's machinery (dataflow launches, tagged granules,
        peer-to-peer stores into mapped zones, HIP graphs) - and compares.  The verdicts are all-gathered over the control plane:
        on a mismatch anywhere every rank keeps the replayed state, the ranks that differed are reported
        (``verify_report``, a line on rank 0) and the run goes on one level more conservative (first without dataflow launches,
        then with the exchange through the host as well).  Costs the window twice plus two state copies: a soak / commissioning
        tool for a new node, not a production setting.  COLLECTIVE like ``advance``."""
        ve = self._verify_every
        while n_steps > 0:
            if self._v_snapshot is None:
                self.synchronize()
                # on the device: exact also with wetting-drying (the device carries D).  Slot 1 belongs to the window: graph captures
                # that happen inside it (an advance size that does not divide verify_every, the first chunk of the coupled p2p path)
                # save and restore around their warm-up steps in slot 0 and must not overwrite the window's start
                self.dev.snapshot(slot=1)
                self._v_snapshot = True
                self._v_steps = 0
            r = min(n_steps, ve - self._v_steps)
            self._advance(r, use_graph)
            self._v_steps += r
            n_steps -= r
            if self._v_steps >= ve:
                fault = os.environ.get('THETIS_AMD_TEST_VERIFY_FAULT')       # tests: "rank:window" - one wrong bit in that window's fast result
                if fault and [int(v) for v in fault.split(':')] == [self.rank, self.verify_report['windows']]:
                    self.synchronize()
                    uv, eta = self.dev.get_state()
                    eta[0, 0] = np.nextafter(eta[0, 0], np.inf)
                    self.dev.set_state(uv, eta)
                self._verify_window()

    def _verify_window(self):
        import torch.distributed as dist
        try:
            self.synchronize()
            fast = self._local_digest()
        except RuntimeError as e:                        # a wait of the fast path timed out: a mismatch by definition
            fast = 'timeout: {:}'.format(e)
        n = self._v_steps
        self._v_snapshot, self._v_steps = None, 0
        self.dev.restore(slot=1)
        self._replaying = True
        try:
            with self._stream_ctx():
                self._steps_eager(n)
            if self.stream is not None:
                self.stream.synchronize()
            slow = self._local_digest()
        finally:
            self._replaying = False
        bad = np.zeros(self.world, dtype=np.int64)
        bad[self.rank] = 0 if fast == slow else 1
        bad = self._all_reduce_int(bad)
        self.verify_report['windows'] += 1
        if bad.any():
            ranks = [int(r) for r in np.nonzero(bad)[0]]
            self.verify_report['mismatches'] += 1
            self.verify_report['bad_ranks'].append(ranks)
            # (a wait of the peer-to-peer kernels that timed out is sticky in the handle: that transport is not used again)
            timed_out = self._all_reduce([1.0 if fast.startswith('timeout') else 0.0], dist.ReduceOp.MAX)[0] > 0.5
            self._distrust = 2 if timed_out else min(2, self._distrust + 1)
            self.graph, self._cycle_graphs = None, {}
            self.config_changed()
            if self.rank == 0:
                print('[thetis_amd] VERIFY: the fast path and its stage-launch replay through host memory differ on rank(s) {:} after a '
                      'window of {:d} steps: the replayed state is kept, the run goes on {:}'.format(
                          ranks, n, 'without dataflow launches' if self._distrust == 1 else 'with stage launches and the exchange through the host'),
                      flush=True)
        # either way the device now holds the replayed state (bitwise the fast one when they agreed), ghosts exchanged, nothing pending

    def _advance(self, n_steps, use_graph=True):
        self._decide_flow()
        with self._stream_ctx():
            if not use_graph or self.graph_mode == 'none' or os.environ.get('THETIS_AMD_NO_GRAPH'):
                self._steps_eager(n_steps)
                return
            coupled = bool(self.tids or self.tracer_only)
            if self.graph_mode == 'cycle' and not coupled:
                self._steps_eager(n_steps, graphed=True)
                return
            if coupled:
                # coupled steps: the launch sequence of a chunk of steps as ONE graph, replayed chunk after chunk (a graph of a
                # whole export interval would have tens of thousands of nodes), the remainder eagerly.  Only with the peer-to-peer
                # exchange, whose sends and receives are kernels of this library: nothing else belongs inside a capture.
                m = self.exchange_every
                chunk = m*max(1, 16//m)
                if self.p2p is None or n_steps < chunk:
                    self._steps_eager(n_steps)
                    return
                if self.graph is None or self.graph_steps != chunk:
                    self._capture(chunk)
                while self.graph is not None and n_steps >= chunk:
                    self.graph.replay()
                    n_steps -= chunk
                if n_steps:
                    self._steps_eager(n_steps)
                return
            if self.graph is None or self.graph_steps != n_steps:
                self._capture(n_steps)
            if self.graph is not None:
                self.graph.replay()
            else:
                self._steps_eager(n_steps)

    # ---- one time step stage by stage (the host runs ``update_forcings`` between the stages, rungekutta.py:933-934)
    def _one_step_ops(self):
        """The launches of ONE time step keyed by what the host asks for: ('swe', i) | ('tracer', t, i) | ('limit', t) -> callable.
        Every field is exchanged right after its last launch of the step (the state after stage 3, a tracer after its limiter),
        so that between two host calls nothing is in flight and the ghost layers a later launch reads are valid: with per-step
        exchanges the ranges are those of ``_cycle_swe(1)`` / ``_step_tracer``, on the deep halos of combined cycles those of a
        one-step ``coupled_cycle_schedule``.  Bitwise the batched ``advance`` (same launches on the same ranges; a received ghost
        value is bitwise the redundantly computed one it replaces)."""
        if getattr(self, '_step_ops', None) is not None:
            return self._step_ops
        dev, p, sps = self.dev, self.part, self.stages_per_step
        fe = sps == 1
        ops = {}

        def exchange(channel, i_buffer=0):
            self._receive(channel, i_buffer, self._send(channel, i_buffer))
        if self.coupled_cycles:
            last_tracer_op = {}
            for op in self._coupled_ops(1):
                if op[0] == 'swe':
                    _, i, end = op
                    if fe:
                        def run(end=end):
                            dev.forward_euler_cells(0, end)
                            dev.swap_state_buffers()
                            exchange(0)
                    elif i == sps - 1:
                        def run(i=i, end=end):
                            dev.solve_stage_cells(i, 0, end)
                            exchange(0)
                    else:
                        def run(i=i, end=end):
                            dev.solve_stage_cells(i, 0, end)
                    ops[('swe', i)] = run
                elif op[0] == 'tracer':
                    _, t, i, end = op

                    def run(t=t, i=i, end=end):
                        dev.tracer_solve_stage_cells(self.tids[t], i, 0, end)
                        if fe:
                            dev.tracer_swap_buffers(self.tids[t])
                        if not self.use_limiter and i == sps - 1:
                            exchange(1 + t)
                    ops[('tracer', t, i)] = run
                elif op[0] == 'limit':
                    _, t, end = op

                    def run(t=t, end=end):
                        dev.tracer_limit_cells(self.tids[t], end)
                        exchange(1 + t)
                    ops[('limit', t)] = run
        else:
            if fe:
                def run():
                    self._cycle_forward_euler(1)
                ops[('swe', 0)] = run
            else:
                r = [p.stage_range(g, depth=3) for g in range(3)]
                ops[('swe', 0)] = lambda: dev.solve_stage_cells(0, 0, r[0])
                ops[('swe', 1)] = lambda: dev.solve_stage_cells(1, 0, r[1])

                def last():
                    dev.solve_stage_cells(2, p.n_interior, p.n_owned)          # the cells the peers are waiting for
                    reqs = self._send(0, 0)
                    dev.solve_stage_cells(2, 0, p.n_interior)
                    self._receive(0, 0, reqs)
                ops[('swe', 2)] = last
            for t in range(len(self.tids)):
                tid = self.tids[t]
                ops[('tracer', t, 0)] = lambda tid=tid: dev.tracer_solve_stage_cells(tid, 0, 0, self._ranges[0])
                ops[('tracer', t, 1)] = lambda tid=tid: dev.tracer_solve_stage_cells(tid, 1, 0, self._ranges[1])

                def tlast(t=t, tid=tid):
                    dev.tracer_solve_stage_cells(tid, 2, p.n_interior, p.n_owned)
                    reqs = self._send(1 + t, 0)
                    dev.tracer_solve_stage_cells(tid, 2, 0, p.n_interior)
                    self._receive(1 + t, 0, reqs)
                ops[('tracer', t, 2)] = tlast
                ops[('limit', t)] = lambda tid=tid: dev.tracer_limit_cells(tid, p.layer_end(3))
        self._step_ops = ops
        return ops

    def run_stage(self, *key):
        """One host-visible piece of a time step: ``run_stage('swe', i)``, ``run_stage('tracer', t, i)``, ``run_stage('limit', t)``
        in the order of the coupled step (coupled_timeintegrator_2d.py:93-113).  COLLECTIVE (the last piece of a field exchanges it)."""
        with self._stream_ctx():
            self._one_step_ops()[tuple(key)]()

    def get_stage_state_owned(self, i_stage=2):
        """(global ids, uv, eta) of the owned cells after stage ``i_stage`` of the current step (2: the step result)."""
        self._check_exchange()
        uv, eta = self.dev.get_state(i_stage)
        n = self.part.n_owned
        return self.part.local_to_global[:n], uv[:n], eta[:n]

    def _capture(self, n_steps):
        """Build the graphs for an ``n_steps`` advance (set-up, not stepping: the state it perturbs is restored).
        COLLECTIVE when it steps: every rank must call it with the same arguments."""
        import torch
        self.graph, self.graph_steps = None, n_steps
        if os.environ.get('THETIS_AMD_NO_GRAPH') or self.graph_mode == 'none':
            return
        self._decide_flow()
        # everything below must run on self.stream: the exchange orders its sends / receives against torch's CURRENT
        # stream, and the kernels of the handle are bound to self.stream
        with torch.cuda.stream(self.stream):
            if self.graph_mode == 'cycle' and not (self.tids or self.tracer_only):
                # build the per-cycle graphs of this step count's schedule by running it once (state restored)
                self.dev.snapshot()
                self._steps_eager(1)                     # RCCL connections, module loading: never inside a capture
                self._steps_eager(n_steps, graphed=True)
                torch.cuda.synchronize()
                self.dev.restore()
                return
            try:
                g = torch.cuda.CUDAGraph()
                # warm-up outside capture (RCCL connection set-up must not happen inside a capture); everything it steps is put back
                self.dev.snapshot()                      # (state and tracers, on the device: exact also with wetting-drying)
                self._steps_eager(1)
                torch.cuda.synchronize()
                self.dev.restore()
                with torch.cuda.graph(g, stream=self.stream, capture_error_mode='thread_local'):
                    self._steps_eager(n_steps)
                self.graph = g
            except Exception as e:   # fall back to eager launches: slower, same results
                print('[thetis_amd] rank {:d}: HIP graph capture unavailable ({:}); running eagerly'.format(
                    self.rank, (str(e).splitlines() or [type(e).__name__])[0]))
                self.graph = None
                torch.cuda.synchronize()
            # The outcome is COLLECTIVE and final: either every rank replays graphs or every rank launches eagerly, and a
            # capture that failed is not tried again.  A rank that retried on its own would run the warm-up step above alone at the
            # next advance - one peer-to-peer push its peers do not match: its epochs run one ahead, and with exchange_every >= 2 the
            # peers would take a one-step halo for the end-of-cycle halo.
            import torch.distributed as dist
            if self.world > 1:
                ok = self._all_reduce([1.0 if self.graph is not None else 0.0], dist.ReduceOp.MIN)[0] > 0.5
            else:
                ok = self.graph is not None
            if not ok:
                self.graph = None
                self.graph_mode = 'none'                 # advance() launches eagerly from now on and never comes back here

    @property
    def graphed(self):
        """True when the step loop runs from HIP graphs (either mode)."""
        return self.graph is not None or bool(self._cycle_graphs)

    def _check_exchange(self):
        """A peer-to-peer wait that timed out (a lost or very late peer: bounded, counted, sticky - csrc/swe2d_p2p.h) leaves stale
        ghost values behind: every point where results leave the device raises instead of returning them."""
        if self.p2p is not None and self._distrust < 2:
            n = self.p2p.timeouts()
            if n:
                raise RuntimeError('{:d} peer-to-peer halo waits timed out on rank {:d} (THETIS_AMD_P2P_TIMEOUT_S): the ghost cells '
                                   'are stale, the state is invalid'.format(n, self.rank))

    def synchronize(self):
        if self.stream is not None:
            self.stream.synchronize()
        if self.xstream is not None:
            self.xstream.synchronize()
        self._check_exchange()

    def diagnostics(self):
        """Global {int eta^2, int |u|^2, int (eta+h), min(h+eta)}: the all-reduced diagnostics of the reference
        (thetis/callback.py:478-482) as limb sums added over the ranks as integers - the doubles of the single-device run."""
        import torch.distributed as dist
        self._check_exchange()
        limbs, lo = self.dev.diagnostics_limbs()
        total = self._all_reduce_int(limbs).reshape(3, -1)
        return np.array([self.dev.limbs_to_double(total[q]) for q in range(3)] + [self._all_reduce([lo], dist.ReduceOp.MIN)[0]])


def state_digest(solver):
    """bitwise fingerprint of the owned state (blake2b of the raw doubles, cells in the order of their GLOBAL ids: a rank's local
    numbering puts its send cells last, and which cells those are depends on the halo depth - two schedules with different
    ``exchange_every`` hold the same owned cells in different orders)"""
    import hashlib
    ids, u, e = solver.get_state_owned()
    o = np.argsort(ids, kind='stable')
    u, e = u[o], e[o]
    return hashlib.blake2b(np.ascontiguousarray(u).tobytes() + np.ascontiguousarray(e).tobytes(), digest_size=16).hexdigest()


def strip_submesh_case(rank, world, nx, ny, lx, ly, halo_depth):
    """Rank ``rank``'s strip of RectangleMesh(nx, ny, lx, ly) WITHOUT building the global mesh: the sub-rectangle of its own
    columns plus ``halo_depth + 2`` columns either side (two triangles per quad: more than ``halo_depth`` facet layers), as a
    RectangleMesh of its own, partitioned with the neighbours' columns owned by rank -+ 1.  Cells of the overlap have the same
    relative order in both ranks' sub-meshes (row by row, column by column), so the send list of one is the receive list of
    the other, as with the global mesh.  Returns (LocalPartition, bathymetry per sub-mesh vertex, uv, eta per sub-mesh cell):
    flat bathymetry, the bench's Gaussian hump plus a deterministic ripple (a function of position, so that a ghost cell
    starts from its owner's values)."""
    from .mesh import RectangleMesh
    i0, i1 = rank*nx//world, (rank + 1)*nx//world
    a, b = max(0, i0 - (halo_depth + 2)), min(nx, i1 + halo_depth + 2)
    dx = lx/nx
    sub = RectangleMesh(b - a, ny, dx*(b - a), ly)
    sub.vertex_xy = sub.vertex_xy + np.array([a*dx, 0.0])
    # the strip's ends are walls only where they are the channel's ends; boundary lengths are those of the whole channel
    sub.boundary_len = {1: ly, 2: ly, 3: lx, 4: lx}
    col = (np.arange(sub.num_cells)//2) % (b - a) + a
    owner = np.where(col < i0, rank - 1, np.where(col < i1, rank, rank + 1))
    part = build_partition(sub, owner, rank, halo_depth=halo_depth)
    cxy = sub.cell_xy()
    x, y = cxy[:, :, 0], cxy[:, :, 1]
    eta = 0.5*np.exp(-((x - 0.5*lx)**2 + (y - 0.5*ly)**2)/(5e3)**2) + 1e-3*np.sin(x/731.0)*np.cos(y/517.0)
    uv = 1e-3*np.stack([np.sin(x/613.0 + y/389.0), np.cos(x/457.0 - y/823.0)], axis=-1)
    return part, np.full(sub.num_vertices, 20.0), uv, eta
