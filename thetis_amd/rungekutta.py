"""
SSPRK(3,3) in Shu-Osher form, stepping on the GPU.

Mirrors ``thetis/rungekutta.py``: ``SSPRK33Abstract`` (:326-347) carries the tableau, ``ERKGenericShuOsher``
(:870-952) the stage loop

    for i in 0..2:  update_forcings(t + c_i dt);  k = M^-1 dt R(U);  U <- beta_{i+1,i} k + sum_j alpha_{i+1,j} U_j

where the reference's ``self.solver.solve()`` + ``solution.assign(sol_expressions[i])`` + stage copies is ONE fused HIP
kernel launch per stage here (swe2d_solve_stage in include/swe2d.h).
"""
import numpy as np

from . import _lib
from .spmd import make_device
from .function import Function
from .options import Constant
from .shallowwater_eq import g_grav
from .timeintegrator import TimeIntegrator

__all__ = ['ForwardEuler', 'SSPRK33Abstract', 'ERKGenericShuOsher', 'SSPRK33', 'butcher_to_shuosher_form']


def butcher_to_shuosher_form(a, b):
    """Shu-Osher arrays (alpha, beta), both (s+1, s+1), of an EXPLICIT s-stage Butcher tableau (a, b) - the form the reference
    builds in rungekutta.py:13-87: beta holds the sub-diagonal of the stacked tableau K = [a; b], alpha follows from

        K[i, :] = sum_j alpha[i, j] K[j, :] + beta[i, :]          (row i of the stage u_i = u_0 + dt sum_m K[i, m] F_m)

    solved row by row by back substitution (the reference inverts the lower-triangular block with numpy.linalg instead), and
    alpha[i, 0] = 1 - sum_j alpha[i, j] by consistency.  Entries below 1e-13 are rounded to zero as the reference does.
    tests/golden/shuosher_explicit.json holds the reference function's own output for every explicit tableau of its file."""
    a = np.atleast_2d(np.asarray(a, dtype=float))
    b = np.asarray(b, dtype=float).reshape(-1)
    s_ = a.shape[0]
    if a.shape != (s_, s_) or b.shape != (s_,):
        raise ValueError('a must be (s, s) and b (s,)')
    if np.diag(a).any() or np.triu(a, 1).any():
        raise NotImplementedError('implicit tableaus are outside the explicit path of this build')
    K = np.vstack([a, b])                                  # row i: weights of F_0..F_{s-1} in stage i (row s: the step)
    alpha = np.zeros((s_ + 1, s_ + 1))
    beta = np.zeros((s_ + 1, s_ + 1))
    alpha[0, 0] = 1.0
    for i in range(1, s_ + 1):
        beta[i, i - 1] = K[i, i - 1]
        for m in range(i - 2, -1, -1):                     # column m fixes alpha[i, m + 1]
            if K[m + 1, m] == 0.0:
                raise NotImplementedError('zero sub-diagonal entry: this Shu-Osher form does not exist for the tableau')
            rest = sum(alpha[i, j]*K[j, m] for j in range(m + 2, i))
            alpha[i, m + 1] = (K[i, m] - rest)/K[m + 1, m]
        alpha[i, 0] = 1.0 - alpha[i, 1:].sum()
    alpha[np.abs(alpha) < 1e-13] = 0.0
    beta[np.abs(beta) < 1e-13] = 0.0
    assert np.allclose(alpha.sum(axis=1), 1.0)
    assert np.allclose(beta[:, :-1] + alpha[:, :-1] @ a, K)
    return alpha, beta


class SSPRK33Abstract(object):
    """3rd order Strong Stability Preserving Runge-Kutta scheme, SSP(3,3) (rungekutta.py:326-347)."""
    a = np.array([[0, 0, 0], [1.0, 0, 0], [0.25, 0.25, 0]])
    b = np.array([1.0/6.0, 1.0/6.0, 2.0/3.0])
    c = np.array([0, 1.0, 0.5])
    cfl_coeff = 1.0
    n_stages = 3
    is_implicit = False
    is_dirk = False
    # Shu-Osher form = output of the reference's butcher_to_shuosher_form(a, b) (rungekutta.py:13-87);
    # bit-exact values pinned by tests/golden/shuosher_ssprk33.json
    alpha = np.array([[1.0, 0.0, 0.0, 0.0],
                      [1.0, 0.0, 0.0, 0.0],
                      [0.75, 0.25, 0.0, 0.0],
                      [0.33333333333333337, 0.0, 0.6666666666666666, 0.0]])
    beta = np.array([[0.0, 0.0, 0.0, 0.0],
                     [1.0, 0.0, 0.0, 0.0],
                     [0.0, 0.25, 0.0, 0.0],
                     [0.0, 0.0, 0.6666666666666666, 0.0]])


def _const_value(v):
    if v is None:
        return None
    if isinstance(v, Constant):
        vals = v.values()
        return vals[0] if len(vals) == 1 else tuple(vals)
    if isinstance(v, (tuple, list, np.ndarray)):
        return tuple(float(x) for x in v)
    return float(v)


class ERKGenericShuOsher(TimeIntegrator):
    """Generic explicit Runge-Kutta time integrator in Shu-Osher form, device resident (rungekutta.py:870-952)."""

    def __init__(self, equation, solution, fields, dt, options, bnd_conditions, terms_to_add='all',
                 device_id=0, comm=None, spmd=None, device_cls=None):
        """``comm`` (thetis_amd/comm.py) with more than one rank: the mesh is partitioned over the ranks' GPUs and ``self.device`` is
        a ``PartitionedDevice`` (thetis_amd/spmd.py; ``spmd``: what the partition must be built for - n_tracers, use_limiter,
        tracer_only, stepper); otherwise one ``Swe2dDevice`` on ``device_id``."""
        super(ERKGenericShuOsher, self).__init__(equation, solution, fields, dt, options)
        if terms_to_add != 'all':
            raise NotImplementedError("the fused stage kernel evaluates all terms; terms_to_add must be 'all'")
        equation.check_fields(fields)
        self.bnd_conditions = bnd_conditions
        mesh = equation.mesh
        opts = equation.options
        bath = equation.depth.bathymetry_2d
        if not isinstance(bath, Function):
            raise NotImplementedError('bathymetry_2d must be a P1 Function')
        if bath.function_space().family == 'CG':
            bath_vertex = bath.dat.data_ro
        else:
            # a DG-P1 bathymetry (test/swe2d/test_atmospheric_pressure.py:55-57) is accepted when it is continuous: the
            # kernel takes the bathymetry from the mesh vertices (avg(total_h) = h + avg(eta) on facets)
            vals = bath.cell_node_values()
            cells = mesh.cells
            bath_vertex = np.zeros(mesh.num_vertices)
            bath_vertex[cells.ravel()] = vals.ravel()
            if np.abs(bath_vertex[cells] - vals).max() > 1e-12*max(1.0, np.abs(vals).max()):
                raise NotImplementedError('discontinuous (DG) bathymetry is not supported on the device path')
        self.comm = comm
        self.device = make_device(
            mesh, bath_vertex, dt, comm=comm, spmd=spmd, device_cls=device_cls, g_grav=float(g_grav),
            use_nonlinear_equations=opts.use_nonlinear_equations,
            use_lax_friedrichs_velocity=opts.use_lax_friedrichs_velocity,
            lax_friedrichs_velocity_scaling_factor=float(fields.get('lax_friedrichs_velocity_scaling_factor') or 1.0),
            device_id=device_id, boundary_len=getattr(mesh, 'boundary_len', None))
        self._uploaded_version = None
        self._device_ahead = False
        if equation.depth.use_wetting_and_drying:
            alpha = equation.depth.wetting_and_drying_alpha
            if isinstance(alpha, Function):
                if alpha.function_space().family != 'CG':
                    raise NotImplementedError('wetting_and_drying_alpha must be a Constant or a CG-P1 Function')
                alpha = alpha.dat.data_ro
            else:
                alpha = float(alpha)
            self.device.set_wetting_and_drying(alpha)
        self._push_fields()
        self._push_bcs()
        uv, eta = self.solution.subfunctions
        uv._pull_hook = self._pull_solution
        eta._pull_hook = self._pull_solution

    # ---- coefficient / boundary upload
    def _nodal(self, value, vector=False):
        """Constant | callable | Function  ->  (N,3[,2]) nodal values."""
        mesh = self.equation.mesh
        if isinstance(value, Function):
            return value.cell_node_values()
        p = mesh.cell_xy()
        if callable(value):
            val = value(p[:, :, 0], p[:, :, 1])
            if vector:
                return np.stack([np.asarray(val[0])*np.ones(p.shape[:2]), np.asarray(val[1])*np.ones(p.shape[:2])], axis=2)
            return np.asarray(val)*np.ones(p.shape[:2])
        c = _const_value(value)
        if vector:
            return np.broadcast_to(np.asarray(c, dtype=float), p.shape).copy()
        return np.full(p.shape[:2], float(c))

    @staticmethod
    def _signature(v):
        """Changes when a coefficient may have changed: Functions carry a host version, Constants their value."""
        if isinstance(v, Function):
            return ('f', id(v), v._host_version)
        if v is None or callable(v):
            return ('o', id(v))
        return ('c', _const_value(v))

    def _push_fields(self, only_changed=False):
        """Upload the coefficient fields.  ``only_changed``: after an ``update_forcings`` call - the reference's forms see
        updated Functions / Constants automatically (rungekutta.py:933-934), here the ones whose signature changed are
        uploaded again (e.g. a time-dependent wind stress or atmospheric pressure field)."""
        f = self.fields
        dev = self.device
        seen = getattr(self, '_field_signatures', {}) if only_changed else {}
        new = {}

        def changed(key):
            new[key] = self._signature(f.get(key))
            return seen.get(key) != new[key]
        if not only_changed:
            dev.set_scalar(_lib.SCALAR_NORM_SMOOTHER, float(getattr(self.equation.options, 'norm_smoother', 0.0) or 0.0))
        # drag coefficients: Constants go to the scalar slots, Functions (spatially varying) to nodal fields
        for key, sid, fid in (('linear_drag_coefficient', _lib.SCALAR_LINEAR_DRAG, _lib.FIELD_LINEAR_DRAG),
                              ('quadratic_drag_coefficient', _lib.SCALAR_QUADRATIC_DRAG, _lib.FIELD_QUADRATIC_DRAG),
                              ('manning_drag_coefficient', _lib.SCALAR_MANNING_DRAG, _lib.FIELD_MANNING_DRAG),
                              ('nikuradse_bed_roughness', _lib.SCALAR_NIKURADSE, _lib.FIELD_NIKURADSE)):
            if not changed(key):
                continue
            v = f.get(key)
            if isinstance(v, Function) or callable(v):
                dev.set_scalar(sid, None)
                self._set_field(fid, v)
            else:
                dev.set_field(fid, None)
                dev.set_scalar(sid, _const_value(v))
        for key, fid, vec in (('coriolis', _lib.FIELD_CORIOLIS, False),
                              ('atmospheric_pressure', _lib.FIELD_ATMOSPHERIC_PRESSURE, False),
                              ('momentum_source', _lib.FIELD_MOMENTUM_SOURCE, True),
                              ('volume_source', _lib.FIELD_VOLUME_SOURCE, False),
                              ('wind_stress', _lib.FIELD_WIND_STRESS, True)):
            if not changed(key):
                continue
            v = f.get(key)
            self._set_field(fid, v, vector=vec)
        if changed('viscosity_h'):
            nu = f.get('viscosity_h')
            if nu is not None:                   # HorizontalViscosityTerm, shallowwater_eq.py:554-616
                opts = self.equation.options
                dev.set_viscosity(self._vertex_coefficient(nu), sipg_factor=float(_const_value(opts.sipg_factor)),
                                  use_grad_div_viscosity_term=opts.use_grad_div_viscosity_term,
                                  use_grad_depth_viscosity_term=opts.use_grad_depth_viscosity_term)
            elif only_changed:
                dev.set_viscosity(None)
        self._field_signatures = new

    def _set_field(self, fid, v, vector=False):
        """Upload one coefficient field; a continuous (CG-P1) Function goes as one value per vertex and is injected into the
        DG nodes on the device (the cheap path for forcing fields that ``update_forcings`` changes every step)."""
        if v is None:
            self.device.set_field(fid, None)
        elif isinstance(v, Function) and v.function_space().family == 'CG' and v.function_space().vector == vector:
            self.device.set_field_vertex(fid, v.dat.data_ro)
        else:
            self.device.set_field(fid, self._nodal(v, vector=vector))

    @staticmethod
    def _vertex_coefficient(value):
        """Constant -> float;  continuous P1 Function -> per-vertex array (the SIPG kernels take either)."""
        if isinstance(value, Function):
            if value.function_space().family != 'CG':
                raise NotImplementedError('diffusion coefficients must be Constants or continuous (CG-P1) Functions')
            return np.ascontiguousarray(value.dat.data_ro, dtype=np.float64)
        return float(_const_value(value))

    def _push_bcs(self):
        """Upload the boundary conditions; called again after every ``update_forcings`` - markers whose Constants / Functions
        did not change since the last upload are skipped (a Function-valued boundary costs a nodal field copy)."""
        mesh = self.equation.mesh
        cache = self.__dict__.setdefault('_bc_signatures', {})
        for marker in mesh.boundary_markers:
            funcs = self.bnd_conditions.get(marker)
            sig = None if funcs is None else tuple(sorted((k, self._signature(v)) for k, v in funcs.items()))
            if marker in cache and cache[marker] == sig:
                continue
            cache[marker] = sig
            if funcs is None:
                self.device.set_bc(marker, None)
                continue
            vals = {}
            for key, v in funcs.items():
                if key not in ('elev', 'uv', 'un', 'flux', 'drag'):
                    raise Exception('Invalid boundary tag "{:}" specified on boundary {:}'.format(key, marker))
                if isinstance(v, Function):
                    # Function-valued boundary data (e.g. a tidal elevation field): nodal values at the DG nodes
                    if key not in ('elev', 'uv', 'un', 'flux'):
                        raise NotImplementedError("'{:}' must be a constant on the device path".format(key))
                    # only the values on this marker's boundary facets travel to the device (a few KB per update_forcings)
                    fs = v.function_space()
                    if fs.family == 'DG' and fs.degree == 1:
                        d = v.dat.data_ro
                        vals[key] = self.device.facet_node_values(marker, d.reshape((mesh.num_cells, fs.npc) + d.shape[1:]))
                    elif fs.family == 'CG':
                        vals[key] = self.device.facet_node_values(marker, v.dat.data_ro, cells_of_vertices=mesh.cells)
                    else:
                        vals[key] = np.ascontiguousarray(v.cell_node_values())
                elif callable(v):
                    raise NotImplementedError('boundary values must be Constants or Functions on the device path')
                else:
                    vals[key] = _const_value(v)
            self.device.set_bc(marker, vals)

    # ---- host <-> device state
    def _host_version(self):
        uv, eta = self.solution.subfunctions
        return (uv._host_version, eta._host_version)

    def _push_solution(self):
        uv, eta = self.solution.subfunctions
        self.device.set_state(uv._data, eta._data)
        self._uploaded_version = self._host_version()
        self._device_ahead = False

    def _pull_solution(self):
        """Refresh the host copy of ``solution`` (called lazily when someone reads ``.dat.data``)."""
        if self._device_ahead:
            uv, eta = self.solution.subfunctions
            # the reference assigns every stage solution to `solution` (rungekutta.py:930-946): between stages a reader (e.g.
            # update_forcings) sees the last completed stage, not the old step
            u, e = self.device.get_state(getattr(self, '_last_stage', 2))
            uv._data[...] = u.reshape(uv._data.shape)
            eta._data[...] = e.reshape(eta._data.shape)
            self._device_ahead = False

    def _sync_to_device(self):
        if self._uploaded_version != self._host_version():
            self._pull_solution()           # no-op unless the device is ahead (then host edits win on top of it)
            self._push_solution()

    def initialize(self, solution):
        """rungekutta.py:926-927 is a no-op; here the initial state goes to HBM."""
        self._push_solution()

    def set_dt(self, dt):
        super(ERKGenericShuOsher, self).set_dt(dt)
        self.device.set_dt(dt)

    def solve_stage(self, i_stage, t, update_forcings=None):
        """Solve i-th stage and assign solution to :attr:`self.solution` (rungekutta.py:930-946)."""
        if update_forcings is not None:
            update_forcings(t + self.c[i_stage]*self.dt)
            self._push_bcs()
            self._push_fields(only_changed=True)
        if i_stage == 0:
            self._sync_to_device()
        # (host WRITES to `solution` between the stages of a step are not supported: the device keeps U0 and the stage solutions
        #  in separate buffers; reads see the last completed stage, see _pull_solution)
        self.device.solve_stage(i_stage)
        self._last_stage = i_stage
        self._device_ahead = True

    def advance(self, t, update_forcings=None):
        """Advances equations for one time step (rungekutta.py:949-952)."""
        if update_forcings is None:
            self._sync_to_device()
            self.device.advance(1)
            self._last_stage = 2
            self._device_ahead = True
        else:
            for i in range(self.n_stages):
                self.solve_stage(i, t, update_forcings)

    def advance_steps(self, t, n_steps):
        """``n_steps`` time steps without forcing updates in ONE call into the library (FlowSolver2d.iterate batches the
        steps between exports: no Python between the launches)."""
        self._sync_to_device()
        self.device.advance(int(n_steps))
        self._last_stage = 2
        self._device_ahead = True

    def diagnostics(self):
        """{int eta^2, int |u|^2, int (eta+h), min(h+eta)} of the device-resident state."""
        self._sync_to_device()
        return self.device.diagnostics()


class SSPRK33(ERKGenericShuOsher, SSPRK33Abstract):
    pass


class ForwardEuler(ERKGenericShuOsher):
    """Standard forward Euler time integration scheme (thetis/timeintegrator.py:115-165), the other explicit entry of the
    steppers table (solver2d.py:664), on the same device state and kernels: ``U <- U + dt M^-1 R(U)``."""
    cfl_coeff = 1.0
    n_stages = 1
    c = (0.0,)

    def solve_stage(self, i_stage, t, update_forcings=None):
        assert i_stage == 0
        self.advance(t, update_forcings)

    def advance(self, t, update_forcings=None):
        if update_forcings is not None:
            update_forcings(t + self.dt)            # the reference evaluates the forcings at the NEW time (:161-162)
            self._push_bcs()
            self._push_fields(only_changed=True)
        self._sync_to_device()
        self.device.advance_forward_euler(1)
        self._device_ahead = True

    def advance_steps(self, t, n_steps):
        self._sync_to_device()
        self.device.advance_forward_euler(int(n_steps))
        self._device_ahead = True
