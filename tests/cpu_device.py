"""
TEST INFRASTRUCTURE: a host stand-in for ``thetis_amd.device.Swe2dDevice`` whose arithmetic is the oracle's C restatement
(oracle/ref_lib.py).  It exists so that the rank-parallel HOST logic of the product - ``FlowSolver2d`` under several ranks,
``PartitionedDevice``'s global -> local maps, ``DistributedSwe2d``'s launch schedule and exchanges, the collectives of
thetis_amd/comm.py - runs in the CPU test suite (gloo, no GPU): the tests hand this class to the product through the
``device_cls`` seam (``FlowSolver2d._device_cls``), once as the single "device" of a one-rank run and once per rank of a
partitioned run, and compare the two results bit for bit.  The product never imports this file (or anything under oracle/).

Covered: triangles and quadrilaterals, constant boundary conditions, drag / Coriolis / source coefficients, tracers with the
vertex limiter, SSPRK33 and ForwardEuler - the same buffer rotation as the HIP library (stage i reads buffer i, writes buffer
(i + 1) % 3, buffer 0 = U0 and the step result).  Not covered (raises): wetting-drying, viscosity / diffusion, Function-valued
boundary data, conservative tracers.
"""
import ctypes

import numpy as np

from thetis_amd import _lib
from thetis_amd.device import FacetValues

AL0 = [0.0, 0.75, 0.33333333333333337]
ALI = [1.0, 0.25, 0.6666666666666666]
BE = [1.0, 0.25, 0.6666666666666666]


def _as_array(ptr, n):
    """float64 view of ``n`` doubles at raw address ``ptr`` (a CPU torch tensor's data_ptr)"""
    return np.ctypeslib.as_array(ctypes.cast(ctypes.c_void_p(int(ptr)), ctypes.POINTER(ctypes.c_double)), shape=(int(n),))


class CpuSwe2dDevice(object):
    is_host = True

    def __init__(self, mesh, bathymetry_vertex, dt, g_grav=9.81, use_nonlinear_equations=True,
                 use_lax_friedrichs_velocity=True, lax_friedrichs_velocity_scaling_factor=1.0,
                 device_id=0, n_owned=None, boundary_len=None, reorder='auto', ranges=None):
        self.mesh = mesh
        self.cells = np.asarray(mesh.cells)
        self.n_cells, self.npc = int(self.cells.shape[0]), int(self.cells.shape[1])
        self.n_owned = self.n_cells if n_owned is None else int(n_owned)
        self.xy = np.asarray(mesh.vertex_xy, dtype=np.float64)[self.cells]
        self.nbr = np.asarray(mesh.cell_nbr)
        self.nbf = np.asarray(mesh.cell_nbr_facet)
        self.h = np.asarray(bathymetry_vertex, dtype=np.float64)[self.cells]
        self.dt = float(dt)
        self.boundary_len = dict(boundary_len if boundary_len is not None else getattr(mesh, 'boundary_len', {}))
        markers = sorted(int(m) for m in np.unique(-self.nbr[self.nbr < 0])) if (self.nbr < 0).any() else []
        self._marker_slot = {m: m for m in markers}
        self.perm = None
        self._opts = dict(g=g_grav, use_nonlinear_equations=bool(use_nonlinear_equations),
                          use_lax_friedrichs_velocity=bool(use_lax_friedrichs_velocity),
                          lax_friedrichs_velocity_scaling_factor=float(lax_friedrichs_velocity_scaling_factor))
        self._scalars, self._fields, self._bcs = {}, {}, {}
        self._ref = None
        k = self.npc
        self.U = [np.zeros((self.n_cells, k, 2)) for _ in range(3)]
        self.E = [np.zeros((self.n_cells, k)) for _ in range(3)]
        self.tracers = []                      # dict(T=[3 buffers], bc={}, source=None, rt=None)
        self._topt = dict(use_lax_friedrichs_tracer=False, lax_friedrichs_tracer_scaling_factor=1.0,
                          tracer_advective_velocity_factor=1.0)
        topo = getattr(mesh, 'topo_vertex', None)
        tc = self.cells if topo is None else np.asarray(topo)[self.cells]
        _, tc = np.unique(tc, return_inverse=True)
        self._topo_cells = tc.reshape(-1, k).astype(np.int32)
        p = self.xy
        d = p[:, 1:] - p[:, :1]
        self.area = 0.5*np.abs(np.sum(d[:, :-1, 0]*d[:, 1:, 1] - d[:, 1:, 0]*d[:, :-1, 1], axis=1))

    # ---- the C restatement, rebuilt when the configuration changed
    def _dirty(self):
        self._ref = None
        for t in self.tracers:
            t['rt'] = None

    def ref(self):
        if self._ref is None:
            from oracle.ref_lib import RefSWE
            sc, f = self._scalars, self._fields
            bcs = {m: v for m, v in self._bcs.items() if v}
            self._ref = RefSWE(self.xy, self.nbr, self.nbf, self.h, boundary_len=self.boundary_len,
                               coriolis=f.get(_lib.FIELD_CORIOLIS), atmospheric_pressure=f.get(_lib.FIELD_ATMOSPHERIC_PRESSURE),
                               momentum_source=f.get(_lib.FIELD_MOMENTUM_SOURCE), volume_source=f.get(_lib.FIELD_VOLUME_SOURCE),
                               wind_stress=f.get(_lib.FIELD_WIND_STRESS),
                               linear_drag_coefficient=sc.get(_lib.SCALAR_LINEAR_DRAG),
                               quadratic_drag_coefficient=sc.get(_lib.SCALAR_QUADRATIC_DRAG),
                               manning_drag_coefficient=sc.get(_lib.SCALAR_MANNING_DRAG),
                               norm_smoother=sc.get(_lib.SCALAR_NORM_SMOOTHER) or 0.0, bnd_conditions=bcs, **self._opts)
        return self._ref

    def _rt(self, t):
        tr = self.tracers[t]
        if tr['rt'] is None:
            from oracle.ref_lib import RefTracer
            tr['rt'] = RefTracer(self.ref(), source=tr['source'], bnd_values={m: v for m, v in tr['bc'].items() if v is not None},
                                 cell_topo_vertices=self._topo_cells, **self._topt)
        return tr['rt']

    def close(self):
        pass

    # ---- configuration
    def set_dt(self, dt):
        self.dt = float(dt)

    def _slot(self, marker):
        try:
            return self._marker_slot[int(marker)]
        except KeyError:
            raise KeyError('the mesh has no boundary with marker {:}'.format(marker))

    def set_bc(self, marker, funcs):
        marker = self._slot(marker)
        vals = {}
        for key, v in (funcs or {}).items():
            if isinstance(v, FacetValues) or (isinstance(v, np.ndarray) and v.ndim >= 2):
                raise NotImplementedError('the host stand-in takes constant boundary values only')
            vals[key] = tuple(float(x) for x in v) if key == 'uv' else float(v)
        self._bcs[marker] = vals
        self._dirty()

    def set_scalar(self, which, value):
        if which == _lib.SCALAR_NIKURADSE and value is not None:
            raise NotImplementedError('nikuradse_bed_roughness: not in the host stand-in')
        self._scalars[which] = None if value is None else float(value)
        self._dirty()

    def set_field(self, field, nodal):
        if field in (_lib.FIELD_LINEAR_DRAG, _lib.FIELD_QUADRATIC_DRAG, _lib.FIELD_MANNING_DRAG, _lib.FIELD_NIKURADSE):
            if nodal is None:
                return
            raise NotImplementedError('spatially varying drag coefficients: not in the host stand-in')
        vec = field in (_lib.FIELD_MOMENTUM_SOURCE, _lib.FIELD_WIND_STRESS)
        shape = (self.n_cells, self.npc, 2) if vec else (self.n_cells, self.npc)
        self._fields[field] = None if nodal is None else np.ascontiguousarray(np.broadcast_to(np.asarray(nodal, dtype=np.float64), shape))
        self._dirty()

    def set_field_vertex(self, field, vertex_values):
        self.set_field(field, np.asarray(vertex_values, dtype=np.float64)[self.cells])

    def set_wetting_and_drying(self, alpha):
        if alpha is not None:
            raise NotImplementedError('wetting-drying: not in the host stand-in')

    def set_viscosity(self, nu, **kwargs):
        if nu is not None:
            raise NotImplementedError('viscosity: not in the host stand-in')

    def boundary_facets(self, slot):
        c, f = np.nonzero(self.nbr == -int(slot))
        return np.ascontiguousarray(c.astype(np.int32)), np.ascontiguousarray(f.astype(np.int32))

    def facet_node_values(self, marker, function_values, cells_of_vertices=None):
        raise NotImplementedError('the host stand-in takes constant boundary values only')

    # ---- state
    def set_state(self, uv, eta):
        self.U[0][...] = np.asarray(uv, dtype=np.float64).reshape(self.n_cells, self.npc, 2)
        self.E[0][...] = np.asarray(eta, dtype=np.float64).reshape(self.n_cells, self.npc)

    def get_state(self, i_stage=2):
        b = (int(i_stage) + 1) % 3
        return self.U[b].copy(), self.E[b].copy()

    def snapshot(self, slot=0):
        if not hasattr(self, '_snaps'):
            self._snaps = {}
        self._snaps[slot] = (self.U[0].copy(), self.E[0].copy(), [t['T'][0].copy() for t in self.tracers])

    def restore(self, slot=0):
        snap = self._snaps[slot]
        self.U[0][...], self.E[0][...] = snap[0], snap[1]
        for t, T in zip(self.tracers, snap[2]):
            t['T'][0][...] = T

    def synchronize(self):
        pass

    def set_stream(self, stream_ptr):
        pass

    # ---- shallow water stages
    def solve_stage_cells(self, i, begin, end):
        src, dst = i, (i + 1) % 3
        ku, ke = self.ref().tendency(self.U[src], self.E[src], self.dt)
        s = slice(int(begin), int(end))
        nu = BE[i]*ku[s] + AL0[i]*self.U[0][s] + ALI[i]*self.U[src][s]
        ne = BE[i]*ke[s] + AL0[i]*self.E[0][s] + ALI[i]*self.E[src][s]
        self.U[dst][s], self.E[dst][s] = nu, ne

    def solve_stage_pair_cells(self, end_0, end_1):
        self.solve_stage_cells(0, 0, end_0)
        self.solve_stage_cells(1, 0, end_1)

    def solve_stage(self, i):
        self.solve_stage_cells(i, 0, self.n_cells)

    def advance(self, n_steps=1):
        for _ in range(int(n_steps)):
            for i in range(3):
                self.solve_stage(i)

    def forward_euler_cells(self, begin, end):
        ku, ke = self.ref().tendency(self.U[0], self.E[0], self.dt)
        s = slice(int(begin), int(end))
        self.U[1][s], self.E[1][s] = self.U[0][s] + ku[s], self.E[0][s] + ke[s]

    def swap_state_buffers(self):
        self.U[0], self.U[1] = self.U[1], self.U[0]
        self.E[0], self.E[1] = self.E[1], self.E[0]

    def advance_forward_euler(self, n_steps=1):
        for _ in range(int(n_steps)):
            self.forward_euler_cells(0, self.n_cells)
            self.swap_state_buffers()

    def flow_supported(self):
        return 0

    def flow_timeouts(self):
        return 0

    def _cell_integrals(self, a, b=None):
        """per owned cell: int a [b] dx for P1 / Q1 nodal values on affine cells"""
        n = self.n_owned
        a = a[:n]
        if b is None:
            return self.area[:n]*a.mean(axis=1)
        b = b[:n]
        if self.npc == 3:
            return self.area[:n]/12.0*(a.sum(axis=1)*b.sum(axis=1) + (a*b).sum(axis=1))
        kb = 4*b + 2*np.roll(b, -1, axis=1) + 2*np.roll(b, 1, axis=1) + np.roll(b, 2, axis=1)
        return self.area[:n]/36.0*(a*kb).sum(axis=1)

    @staticmethod
    def _limbs(x):
        """the limb sums of include/swe2d.h (swe2d_diagnostics_limbs) of the terms ``x``: units 2^40, 2^2, 2^-36, 2^-74, 2^-112, 2^-150"""
        x = np.asarray(x, dtype=np.float64).copy()
        assert (np.abs(x) < 2.0**78).all()
        out = np.zeros(6, dtype=np.int64)
        for j, s in enumerate((40, 2, -36, -74, -112, -150)):
            t = np.trunc(np.ldexp(x, -s))
            out[j] = int(t.astype(np.int64).sum())
            x -= np.ldexp(t, s)
        return out

    @staticmethod
    def limbs_to_double(limbs):
        """exact total of limb sums, rounded once (Fraction -> float is correctly rounded)"""
        from fractions import Fraction
        v = sum(Fraction(int(l))*Fraction(2)**s for l, s in zip(np.asarray(limbs).reshape(6), (40, 2, -36, -74, -112, -150)))
        return float(v)

    def diagnostics_limbs(self):
        u, e = self.U[0], self.E[0]
        n = self.n_owned
        terms = [self._cell_integrals(e, e), self._cell_integrals(u[..., 0], u[..., 0]) + self._cell_integrals(u[..., 1], u[..., 1]),
                 self._cell_integrals(e + self.h)]
        return np.stack([self._limbs(t) for t in terms]), float((self.h + e)[:n].min())

    def diagnostics(self):
        limbs, lo = self.diagnostics_limbs()
        return np.array([self.limbs_to_double(l) for l in limbs] + [lo])

    # ---- halo plumbing (the exchange buffers are CPU torch tensors)
    def halo_setup(self, send_cells, recv_cells):
        self._send = np.asarray(send_cells, dtype=np.int64)
        self._recv = np.asarray(recv_cells, dtype=np.int64)

    def halo_pack(self, i_buffer, send_buf_ptr):
        sc, k = self._send, self.npc
        if len(sc):
            U, E = self.U[i_buffer], self.E[i_buffer]
            _as_array(send_buf_ptr, 3*k*len(sc))[:] = np.concatenate([U[sc, :, 0], U[sc, :, 1], E[sc]], axis=1).reshape(-1)

    def halo_unpack(self, i_buffer, recv_buf_ptr):
        rc, k = self._recv, self.npc
        if len(rc):
            r = _as_array(recv_buf_ptr, 3*k*len(rc)).reshape(-1, 3*k)
            U, E = self.U[i_buffer], self.E[i_buffer]
            U[rc, :, 0], U[rc, :, 1], E[rc] = r[:, 0:k], r[:, k:2*k], r[:, 2*k:3*k]

    # ---- tracers
    def add_tracer(self):
        self.tracers.append(dict(T=[np.zeros((self.n_cells, self.npc)) for _ in range(3)], bc={}, source=None, rt=None))
        return len(self.tracers) - 1

    def tracer_set_options(self, use_lax_friedrichs_tracer=False, lax_friedrichs_tracer_scaling_factor=1.0,
                           tracer_advective_velocity_factor=1.0):
        self._topt = dict(use_lax_friedrichs_tracer=bool(use_lax_friedrichs_tracer),
                          lax_friedrichs_tracer_scaling_factor=float(lax_friedrichs_tracer_scaling_factor),
                          tracer_advective_velocity_factor=float(tracer_advective_velocity_factor))
        self._dirty()

    def tracer_set_conservative(self, tid, use_conservative_form=True):
        if use_conservative_form:
            raise NotImplementedError('conservative tracers: not in the host stand-in')

    def tracer_set_diffusivity(self, tid, mu, sipg_factor_tracer=1.0):
        if mu is not None:
            raise NotImplementedError('tracer diffusion: not in the host stand-in')

    def tracer_set_diffusion_bc(self, *args, **kwargs):
        pass

    def tracer_set_state(self, tid, nodal):
        self.tracers[tid]['T'][0][...] = np.asarray(nodal, dtype=np.float64).reshape(self.n_cells, self.npc)

    def tracer_get_state(self, tid):
        return self.tracers[tid]['T'][0].copy()

    def tracer_set_bc(self, tid, marker, value):
        if isinstance(value, np.ndarray) and value.ndim >= 2:
            raise NotImplementedError('the host stand-in takes constant boundary values only')
        self.tracers[tid]['bc'][self._slot(marker)] = None if value is None else float(value)
        self.tracers[tid]['rt'] = None

    def tracer_set_bc_facets(self, *args, **kwargs):
        raise NotImplementedError('the host stand-in takes constant boundary values only')

    def tracer_set_bc_velocity(self, tid, marker, uv=None, un=None, flux=None, elev=None):
        if uv is not None or un is not None or flux is not None:
            raise NotImplementedError('external velocities of tracer boundaries: not in the host stand-in')

    def tracer_set_source(self, tid, nodal):
        self.tracers[tid]['source'] = None if nodal is None else np.ascontiguousarray(
            np.broadcast_to(np.asarray(nodal, dtype=np.float64), (self.n_cells, self.npc)))
        self.tracers[tid]['rt'] = None

    def tracer_solve_stage_cells(self, tid, i, begin, end):
        T = self.tracers[tid]['T']
        src, dst = i, (i + 1) % 3
        # the UPDATED velocity (coupled_timeintegrator_2d.py:99-101): the step result in buffer 0
        k = self._rt(tid).tendency(np.nan_to_num(T[src]), np.nan_to_num(self.U[0]), self.dt)
        s = slice(int(begin), int(end))
        T[dst][s] = BE[i]*k[s] + AL0[i]*T[0][s] + ALI[i]*T[src][s]

    def tracer_solve_stage(self, tid, i):
        self.tracer_solve_stage_cells(tid, i, 0, self.n_cells)

    def tracer_swap_buffers(self, tid):
        T = self.tracers[tid]['T']
        T[0], T[1] = T[1], T[0]

    def tracer_forward_euler(self, tid):
        self.tracer_solve_stage_cells(tid, 0, 0, self.n_cells)
        self.tracer_swap_buffers(tid)

    def tracer_limit_cells(self, tid, cell_end):
        T = self.tracers[tid]['T'][0]
        lim = self._rt(tid).limit(T)
        T[:int(cell_end)] = lim[:int(cell_end)]

    def tracer_limit(self, tid):
        self.tracer_limit_cells(tid, self.n_cells)

    def tracer_halo_pack(self, tid, i_buffer, send_buf_ptr):
        sc = self._send
        if len(sc):
            _as_array(send_buf_ptr, self.npc*len(sc))[:] = self.tracers[tid]['T'][i_buffer][sc].reshape(-1)

    def tracer_halo_unpack(self, tid, i_buffer, recv_buf_ptr):
        rc = self._recv
        if len(rc):
            self.tracers[tid]['T'][i_buffer][rc] = _as_array(recv_buf_ptr, self.npc*len(rc)).reshape(-1, self.npc)

    def tracer_diagnostics_limbs(self, tid):
        T = self.tracers[tid]['T'][0]
        n = self.n_owned
        terms = [self._cell_integrals(T, self.h + self.E[0]), self._cell_integrals(T)]
        return np.stack([self._limbs(t) for t in terms]), np.array([float(T[:n].min()), float(T[:n].max())])

    def tracer_diagnostics(self, tid):
        limbs, mm = self.tracer_diagnostics_limbs(tid)
        return np.array([self.limbs_to_double(limbs[0]), self.limbs_to_double(limbs[1]), mm[0], mm[1]])

    def advance_coupled(self, n_steps=1, tracer_only=False, use_limiter=True):
        for _ in range(int(n_steps)):
            if not tracer_only:
                self.advance(1)
            for tid in range(len(self.tracers)):
                for i in range(3):
                    self.tracer_solve_stage(tid, i)
                if use_limiter:
                    self.tracer_limit(tid)
