"""
``PartitionedDevice``: the multi-GPU engine (thetis_amd/distributed.py) behind the interface the time integrators drive
(thetis_amd/device.py), so that the SAME ``FlowSolver2d`` user script that steps on one GPU is domain-decomposed when it is
launched with one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 examples/channel2d.py

- the counterpart of ``mpiexec -n 4 python channel2d.py`` in the reference (examples/README.md:51-56).  Every rank runs the
script; host-side objects (mesh, ``Function``s, options, boundary dictionaries) are replicated and GLOBAL (thetis_amd/comm.py);
this class cuts them down to the rank's partition on their way to the GPU:

    nodal / cell fields (N, k[, 2])      ->  rows ``part.local_to_global``          (owned cells + ghost layers)
    vertex fields (V[, 2])               ->  rows ``part.vertex_global``
    per-marker boundary data             ->  the partition's own boundary facets of that marker (``boundary_facets`` hands the
                                             integrators GLOBAL cell ids in the LOCAL facet order, so that they pick the values
                                             from the global ``Function`` and this class passes them on unchanged); a marker the
                                             partition does not touch is skipped
    state / tracers read back            ->  owned cells of every rank gathered into the global array (collective)
    diagnostics                          ->  per-rank partial reductions, all-reduced (thetis/callback.py:478-482)

Time stepping: ``advance(n)`` (no forcing updates: the batch between two exports) runs ``DistributedSwe2d.advance`` - one
exchange per ``exchange_every`` steps on deep halos, HIP graphs, the in-launch exchange of the flow kernel where it applies;
``solve_stage`` / ``tracer_solve_stage`` / ``tracer_limit`` (the host runs ``update_forcings`` between stages,
thetis/rungekutta.py:933-934) run the same launches one host call at a time with an exchange after each field's last stage.
Results are bitwise those of the single-device run (tests/test_spmd.py, tests/test_gpu_spmd.py).

Environment: ``THETIS_AMD_EXCHANGE`` = p2p | rccl | host (default: the first that sets up on every rank),
``THETIS_AMD_EXCHANGE_EVERY`` (default: 2 or 1 where every rank then fits the dataflow kernel with its ghost layers, else 4; 2 for
coupled / ForwardEuler / quadrilateral runs), ``THETIS_AMD_OVERLAP_STAGES`` (default 0), ``THETIS_AMD_PARTITION`` = strip | strip_y |
rcb (default: strips along the longer side of a structured mesh, recursive coordinate bisection otherwise),
``THETIS_AMD_SPMD_FLOW`` = 1 | 0 (require / forbid the dataflow launches; default: the ranks' common automatic choice).
"""
import os

import numpy as np

from .device import FacetValues, Swe2dDevice
from .partition import build_partition, rcb_owner, strip_owner

__all__ = ['PartitionedDevice', 'make_device', 'default_owner']

FLOW_RESIDENT_CELLS = 2048*64          # cells of the 64-cell blocks an MI355X holds resident for the dataflow kernel (2 per SIMD x 1024)


def default_owner(mesh, n_parts):
    """The partition a run takes when the user does not choose: strips (<= 2 peers = one xGMI link each) along the longer
    side of a structured rectangle, compact RCB parts of an unstructured mesh."""
    how = os.environ.get('THETIS_AMD_PARTITION', '')
    if how == 'rcb' or (not how and not getattr(mesh, 'structured', False)):
        return rcb_owner(mesh, n_parts)
    if how == 'strip_y':
        return strip_owner(mesh, n_parts, axis=1)
    if how == 'strip':
        return strip_owner(mesh, n_parts, axis=0)
    if how:
        raise ValueError("THETIS_AMD_PARTITION must be 'strip', 'strip_y' or 'rcb'")
    return strip_owner(mesh, n_parts, axis=0 if int(mesh.nx) >= int(mesh.ny) else 1)


def make_device(mesh, bathymetry_vertex, dt, comm=None, spmd=None, device_cls=None, **kwargs):
    """The handle a time integrator steps: ``Swe2dDevice`` on one rank, ``PartitionedDevice`` on several."""
    if comm is None or comm.size == 1:
        return (device_cls or Swe2dDevice)(mesh, bathymetry_vertex, dt, **kwargs)
    return PartitionedDevice(mesh, bathymetry_vertex, dt, comm, device_cls=device_cls, **dict(spmd or {}, **kwargs))


class PartitionedDevice(object):
    def __init__(self, mesh, bathymetry_vertex, dt, comm, n_tracers=0, use_limiter=True, tracer_only=False, stepper='SSPRK33',
                 device_id=None, boundary_len=None, device_cls=None, owner=None, **opts):
        from .distributed import DistributedSwe2d
        self.comm, self.mesh = comm, mesh
        self.n_cells = int(mesh.num_cells)
        self.npc = int(mesh.cells.shape[1])
        self.n_tracers, self.tracer_only = int(n_tracers), bool(tracer_only)
        self.use_limiter = bool(use_limiter) and self.n_tracers > 0
        self._tracers_handed_out = 0
        on_gpu = not getattr(device_cls, 'is_host', False)
        owner = default_owner(mesh, comm.size) if owner is None else np.asarray(owner)
        if len(np.unique(owner)) != comm.size:
            raise ValueError('the mesh has fewer cells than the run has ranks')
        # time steps between two exchanges (3m ghost layers).  Ranks that fit the dataflow kernel WITH their ghost layers (2048
        # resident 64-cell blocks on an MI355X: an eighth of a 1 M-triangle mesh fits with the six layers of m = 2, not with the twelve
        # of m = 4) exchange every 2 steps (or every step) inside its launches; larger ranks run stage launches, where the two
        # exchange kernels of a cycle are spread over 4 steps (measured: a rank of two 69-70 us per step at m = 4 / 8 against 72 at
        # m = 2, a rank of four 42.5-44 against 45).  Decided by all ranks together from the partitions themselves.
        partition = None
        if os.environ.get('THETIS_AMD_EXCHANGE_EVERY'):
            every = max(1, int(os.environ['THETIS_AMD_EXCHANGE_EVERY']))
        elif self.n_tracers or stepper != 'SSPRK33' or self.npc != 3 or not on_gpu:
            every = 2
        else:
            every = 4
            if self.n_cells <= FLOW_RESIDENT_CELLS*comm.size:
                for m in (2, 1):
                    part = build_partition(mesh, owner, comm.rank, halo_depth=3*m)
                    if comm.all_agree(part.num_cells <= FLOW_RESIDENT_CELLS):
                        every, partition = m, part
                        break
        overlap = int(os.environ.get('THETIS_AMD_OVERLAP_STAGES', '0'))
        if stepper == 'ForwardEuler' or (self.n_tracers and overlap and (every == 1 or tracer_only)):
            overlap = 0
        forced = os.environ.get('THETIS_AMD_EXCHANGE')
        if not on_gpu:
            wanted = ['host']
        elif forced:
            wanted = [forced]
        else:
            wanted = ['p2p'] + (['rccl'] if comm.rccl else []) + ['host']
        device_id = comm.local_rank if device_id is None else device_id
        # the dataflow launch (csrc/swe2d_flow.h): by default where every rank's partition is covered and no two ranks share a GPU
        flow = {'1': True, '0': False}.get(os.environ.get('THETIS_AMD_SPMD_FLOW', ''), None)
        self.dist, errors = None, []
        for ex in wanted:
            d, err = None, None
            try:
                d = DistributedSwe2d(mesh, bathymetry_vertex, dt, comm.rank, comm.size, device_id, owner=owner,
                                     n_tracers=self.n_tracers, use_limiter=use_limiter, tracer_only=tracer_only,
                                     exchange_every=every, overlap_stages=overlap, stepper=stepper, exchange=ex,
                                     group=(None if ex == 'rccl' else comm.group), device_cls=device_cls,
                                     flow=(flow if on_gpu else False), partition=partition,
                                     # the last stage of a cycle in two launches (send cells first) lets a host-staged or RCCL exchange
                                     # travel while the interior finishes; peer-to-peer pushes are kernels on the same stream - one
                                     # launch less per cycle, and the steps of a large partition can go as one launch each
                                     # (DistributedSwe2d._cycle_before_exchange)
                                     split_last_stage=(ex != 'p2p' or overlap > 0), **opts)
            except Exception as e:                                    # e.g. IPC mapping refused: every rank moves on together
                err = '{:}: {:}'.format(ex, (str(e).strip().splitlines() or [type(e).__name__])[0])
            if comm.all_agree(err is None):
                self.dist = d
                break
            if d is not None:
                d.close()
            errors.append(err or '{:}: failed on another rank'.format(ex))
        if self.dist is None:
            raise RuntimeError('no halo transport could be set up ({:})'.format('; '.join(errors)))
        self.exchange = self.dist.exchange
        self.part = self.dist.part
        self.dev = self.dist.dev
        self._g = self.part.local_to_global
        self._vg = self.part.vertex_global
        self._bnd = {}

    # ---- lifetime
    def close(self):
        if getattr(self, 'dist', None) is not None:
            self.dist.close()
            self.dist = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- global -> local
    def _cells(self, a, vector=False):
        """a global cell field (N, k[, 2]), or anything that broadcasts to it  ->  the local rows"""
        shape = (self.n_cells, self.npc, 2) if vector else (self.n_cells, self.npc)
        a = np.asarray(a, dtype=np.float64)
        a = a.reshape(shape) if a.size == int(np.prod(shape)) else np.broadcast_to(a, shape)
        return np.ascontiguousarray(a[self._g])

    def _vertices(self, a):
        """a constant stays a constant; a global per-vertex array (V[, 2]) -> the local rows"""
        a = np.asarray(a, dtype=np.float64)
        if a.ndim == 0:
            return float(a)
        if a.shape[0] != self.mesh.num_vertices:
            raise ValueError('expected one value per mesh vertex')
        return np.ascontiguousarray(a[self._vg])

    def _has(self, marker):
        return int(marker) in self.dev._marker_slot

    def _slot(self, marker):
        """At this level a 'slot' is the marker itself: the local handle's slot numbers differ from rank to rank."""
        if int(marker) not in [int(m) for m in self.mesh.boundary_markers]:
            raise KeyError('the mesh has no boundary with marker {:}'.format(marker))
        return int(marker)

    def boundary_facets(self, marker):
        """(cells, facets) of this PARTITION's boundary facets with ``marker`` - cells as GLOBAL ids, in the order of the local
        handle's facet list (owned and ghost cells: a ghost cell's boundary term is evaluated redundantly); empty where the
        partition does not touch the marker."""
        marker = int(marker)
        if marker not in self._bnd:
            if self._has(marker):
                cells, facets = self.dev.boundary_facets(self.dev._slot(marker))
                self._bnd[marker] = (np.ascontiguousarray(self._g[cells]), facets)
            else:
                self._bnd[marker] = (np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int32))
        return self._bnd[marker]

    def facet_node_values(self, marker, function_values, cells_of_vertices=None):
        """as ``Swe2dDevice.facet_node_values`` with GLOBAL function values: the values at the end nodes of the partition's
        boundary facets of ``marker``"""
        cells, facets = self.boundary_facets(marker)
        nxt = (facets + 1) % self.npc
        d = np.asarray(function_values)
        if cells_of_vertices is None:
            return FacetValues(np.stack([d[cells, facets], d[cells, nxt]], axis=1))
        cv = np.asarray(cells_of_vertices)
        return FacetValues(np.stack([d[cv[cells, facets]], d[cv[cells, nxt]]], axis=1))

    def _bc_entry(self, v, vector=False):
        if isinstance(v, np.ndarray) and v.ndim >= 2:
            return self._cells(v, vector=vector)
        return v

    # ---- configuration (every call is made by every rank with the same global arguments)
    def set_dt(self, dt):
        self.dev.set_dt(dt)

    def set_scalar(self, which, value):
        self.dev.set_scalar(which, value)

    def set_field(self, field, nodal):
        from . import _lib
        if nodal is None:
            return self.dev.set_field(field, None)
        self.dev.set_field(field, self._cells(nodal, vector=field in (_lib.FIELD_MOMENTUM_SOURCE, _lib.FIELD_WIND_STRESS)))

    def set_field_vertex(self, field, vertex_values):
        self.dev.set_field_vertex(field, self._vertices(vertex_values))

    def set_wetting_and_drying(self, alpha):
        self.dev.set_wetting_and_drying(None if alpha is None else self._vertices(alpha))
        self.dist.config_changed()

    def set_viscosity(self, nu, **kwargs):
        self.dev.set_viscosity(None if nu is None else self._vertices(nu), **kwargs)
        self.dist.config_changed()

    def set_bc(self, marker, funcs):
        self._slot(marker)
        if not self._has(marker):
            return
        if funcs:
            funcs = {k: self._bc_entry(v, vector=(k == 'uv')) for k, v in funcs.items()}
        self.dev.set_bc(marker, funcs)

    # ---- state
    def set_state(self, uv, eta):
        self.dist.set_state_global(np.asarray(uv, dtype=np.float64).reshape(self.n_cells, self.npc, 2),
                                   np.asarray(eta, dtype=np.float64).reshape(self.n_cells, self.npc))

    def get_state(self, i_stage=2):
        """The global (uv, eta) on every rank: owned cells of all ranks gathered.  COLLECTIVE."""
        ids, uv, eta = self.dist.get_stage_state_owned(i_stage)
        both = np.concatenate([uv.reshape(len(ids), -1), eta.reshape(len(ids), -1)], axis=1)
        out = self.comm.gather_rows(ids, both, self.n_cells)
        k = self.npc
        return out[:, :2*k].reshape(self.n_cells, k, 2), np.ascontiguousarray(out[:, 2*k:])

    def diagnostics(self):
        return self.dist.diagnostics()

    def synchronize(self):
        self.dist.synchronize()

    # ---- time stepping
    def advance(self, n_steps=1):
        self.dist.advance(int(n_steps))

    def solve_stage(self, i_stage):
        self.dist.run_stage('swe', int(i_stage))

    def advance_forward_euler(self, n_steps=1):
        if self.n_tracers:                       # inside a coupled step driven from the host: one step, exchanged
            for _ in range(int(n_steps)):
                self.dist.run_stage('swe', 0)
        else:
            self.dist.advance(int(n_steps))

    def advance_coupled(self, n_steps=1, tracer_only=False, use_limiter=True):
        if bool(tracer_only) != self.tracer_only or (bool(use_limiter) and self.n_tracers > 0) != self.use_limiter:
            raise ValueError('tracer_only / use_limiter differ from what the partition was built for')
        self.dist.advance(int(n_steps))

    # ---- tracers (handles are indices into the tracers the partition was built for)
    def add_tracer(self):
        if self._tracers_handed_out >= self.n_tracers:
            raise RuntimeError('the partitioned handle was built for {:d} tracer(s)'.format(self.n_tracers))
        self._tracers_handed_out += 1
        return self._tracers_handed_out - 1

    def _tid(self, t):
        return self.dist.tids[int(t)]

    def tracer_set_options(self, *args, **kwargs):
        self.dev.tracer_set_options(*args, **kwargs)

    def tracer_set_conservative(self, t, use_conservative_form=True):
        self.dev.tracer_set_conservative(self._tid(t), use_conservative_form)

    def tracer_set_diffusivity(self, t, mu, sipg_factor_tracer=1.0):
        self.dev.tracer_set_diffusivity(self._tid(t), None if mu is None else self._vertices(mu), sipg_factor_tracer)

    def tracer_set_diffusion_bc(self, t, marker, kind, diff_flux=0.0):
        self._slot(marker)
        if self._has(marker):
            self.dev.tracer_set_diffusion_bc(self._tid(t), marker, kind, diff_flux)

    def tracer_set_source(self, t, nodal):
        self.dev.tracer_set_source(self._tid(t), None if nodal is None else self._cells(nodal))

    def tracer_set_state(self, t, nodal):
        self.dev.tracer_set_state(self._tid(t), self._cells(nodal))

    def tracer_get_state(self, t):
        """The global tracer field on every rank.  COLLECTIVE."""
        ids, T = self.dist.get_tracer_owned(int(t))
        return self.comm.gather_rows(ids, T, self.n_cells)

    def tracer_set_bc(self, t, marker, value):
        self._slot(marker)
        if self._has(marker):
            self.dev.tracer_set_bc(self._tid(t), marker, self._bc_entry(value))

    def tracer_set_bc_facets(self, t, marker, values):
        self._slot(marker)
        if self._has(marker):
            self.dev.tracer_set_bc_facets(self._tid(t), self.dev._slot(marker), values)

    def tracer_set_bc_velocity(self, t, marker, uv=None, un=None, flux=None, elev=None):
        self._slot(marker)
        if self._has(marker):
            self.dev.tracer_set_bc_velocity(self._tid(t), marker, uv=self._bc_entry(uv, vector=True), un=self._bc_entry(un),
                                            flux=self._bc_entry(flux), elev=elev)

    def tracer_solve_stage(self, t, i_stage):
        self.dist.run_stage('tracer', int(t), int(i_stage))

    def tracer_forward_euler(self, t):
        self.dist.run_stage('tracer', int(t), 0)

    def tracer_limit(self, t):
        self.dist.run_stage('limit', int(t))

    def tracer_diagnostics(self, t):
        return self.dist.tracer_diagnostics(int(t))
