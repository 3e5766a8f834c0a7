"""
Coupled 2D time integrator: shallow water step, then every tracer with the UPDATED velocity, then the limiter once per
time step (thetis/coupled_timeintegrator_2d.py:93-113).  All of it stays on the device: the tracers are extra DG-P1
fields of the same C-ABI handle that steps the shallow water equations.
"""
import numpy as np

from .function import Function
from .log import print_output
from .options import Constant
from .timeintegrator import TimeIntegratorBase

__all__ = ['DeviceTracerSSPRK33', 'DeviceTracerForwardEuler', 'GeneralCoupledTimeIntegrator2D']


def _velocity_factor(f):
    """options.tracer_advective_velocity_factor: a Constant, or a Function that is spatially constant (as in
    test/tracerEq/test_h-advection_mes_2d.py:52-53); a varying Function is not on the device path."""
    if isinstance(f, Function):
        d = f.dat.data_ro
        if d.size and np.abs(d - d.flat[0]).max() > 1e-14*max(1.0, abs(float(d.flat[0]))):
            raise NotImplementedError('a spatially varying tracer_advective_velocity_factor is not on the device path')
        return float(d.flat[0]) if d.size else 1.0
    return float(f)


class _AttrDict(dict):
    def __init__(self, *args, **kwargs):
        super(_AttrDict, self).__init__(*args, **kwargs)
        self.__dict__ = self


def _cval(v):
    return None if v is None else float(v)


def _vec(v):
    """constant 2-vector given as a tuple / array / vector Constant"""
    a = np.asarray(v.values() if hasattr(v, 'values') and callable(v.values) else v, dtype=float).ravel()
    assert a.shape == (2,), "tracer boundary 'uv' must be a constant 2-vector"
    return a


class DeviceTracerSSPRK33(object):
    """SSPRK33 of one tracer equation on the device handle of the shallow water stepper (rungekutta.py:870-952)."""
    n_stages = 3
    c = (0.0, 1.0, 0.5)

    def __init__(self, equation, solution, fields, dt, options, bnd_conditions, swe_stepper):
        equation.check_bnd_conditions(bnd_conditions)
        self.equation, self.solution, self.fields, self.dt = equation, solution, fields, dt
        self.bnd_conditions = bnd_conditions or {}
        self.swe = swe_stepper
        self.device = swe_stepper.device
        self.tid = self.device.add_tracer()
        self._uploaded = None
        self._device_ahead = False
        self.solution._pull_hook = self._pull
        self._src_signature = None
        self._push_source()
        if getattr(equation, 'conservative', False):
            self.device.tracer_set_conservative(self.tid, True)
        mu = fields.get('diffusivity_h-{:}'.format(equation.label))
        self.diffusive = mu is not None
        if self.diffusive:                      # HorizontalDiffusionTerm, tracer_eq_2d.py:226-278
            self.device.tracer_set_diffusivity(self.tid, swe_stepper._vertex_coefficient(mu),
                                               float(equation.options.sipg_factor_tracer))
        self._push_bcs()

    def _push_source(self):
        """(re-)upload the tracer source when it changed (a Function updated by update_forcings, or a Constant)"""
        src = self.fields.get('source-{:}'.format(self.equation.label))
        sig = self.swe._signature(src)
        if sig != self._src_signature:
            self.device.tracer_set_source(self.tid, None if src is None else self.swe._nodal(src))
            self._src_signature = sig

    def _push_bcs(self):
        for marker in self.equation.mesh.boundary_markers:
            funcs = self.bnd_conditions.get(marker)
            v = None if funcs is None else funcs.get('value')
            is_field = isinstance(v, Function)
            sig = None if funcs is None else tuple(sorted((k, self.swe._signature(x)) for k, x in funcs.items()))
            cache = self.__dict__.setdefault('_bc_signatures', {})
            if marker in cache and cache[marker] == sig:
                continue                # nothing on this marker changed since the last upload
            cache[marker] = sig
            if is_field:            # Function-valued 'value': the nodal values of the boundary cells only (compact upload)
                mesh = self.equation.mesh
                cells, _ = self.device.boundary_facets(self.device._slot(marker))
                fs = v.function_space()
                if fs.family == 'CG':
                    vals = v.dat.data_ro[np.asarray(mesh.cells)[cells]]
                else:
                    vals = v.cell_node_values()[cells]
                self.device.tracer_set_bc_facets(self.tid, self.device._slot(marker), vals)
            else:
                self.device.tracer_set_bc(self.tid, marker, _cval(v))
            uv_b = None if funcs is None else funcs.get('uv')
            un_b = None if funcs is None else funcs.get('un')
            fl_b = None if funcs is None else funcs.get('flux')
            el_b = None if funcs is None else funcs.get('elev')
            def entry(x, vector=False):
                """a Constant / number, or a Function as the values at the end nodes of the marker's boundary facets"""
                if x is None:
                    return None
                if isinstance(x, Function):
                    fs = x.function_space()
                    if fs.family == 'CG':
                        return self.device.facet_node_values(marker, x.dat.data_ro, cells_of_vertices=self.equation.mesh.cells)
                    return self.device.facet_node_values(marker, x.cell_node_values())
                return _vec(x) if vector else _cval(x)
            self.device.tracer_set_bc_velocity(self.tid, marker, uv=entry(uv_b, vector=True), un=entry(un_b), flux=entry(fl_b),
                                               elev=None if el_b is None else _cval(el_b))
            if self.diffusive:                  # boundary term of the diffusion operator, tracer_eq_2d.py:264-277
                if funcs is None:
                    kind, dfl = 0, 0.0
                elif 'diff_flux' in funcs:
                    kind, dfl = 1, _cval(funcs['diff_flux'])
                elif 'value' not in funcs:
                    kind, dfl = 3, 0.0
                else:
                    kind, dfl = (4 if is_field else 2), 0.0
                self.device.tracer_set_diffusion_bc(self.tid, marker, kind, dfl)

    def _pull(self):
        if self._device_ahead:
            self.solution._data[...] = self.device.tracer_get_state(self.tid).reshape(self.solution._data.shape)
            self._device_ahead = False

    def _sync_to_device(self):
        if self._uploaded != self.solution._host_version:
            self._pull()
            self.device.tracer_set_state(self.tid, self.solution._data.reshape(-1, self.device.npc))
            self._uploaded = self.solution._host_version
            self._device_ahead = False

    def initialize(self, solution):
        self._uploaded = None
        self._sync_to_device()

    def set_dt(self, dt):
        self.dt = dt            # the handle's dt is set by the shallow water stepper

    def solve_stage(self, i_stage, t, update_forcings=None):
        if update_forcings is not None:
            update_forcings(t + self.c[i_stage]*self.dt)
            self._push_bcs()
            self._push_source()
        if i_stage == 0:
            self._sync_to_device()
        self.device.tracer_solve_stage(self.tid, i_stage)
        self._device_ahead = True

    def advance(self, t, update_forcings=None):
        for i in range(self.n_stages):
            self.solve_stage(i, t, update_forcings)


class DeviceTracerForwardEuler(DeviceTracerSSPRK33):
    """timeintegrator.ForwardEuler of one tracer equation (thetis/timeintegrator.py:115-165)."""
    n_stages = 1
    c = (0.0,)

    def solve_stage(self, i_stage, t, update_forcings=None):
        assert i_stage == 0
        self.advance(t, update_forcings)

    def advance(self, t, update_forcings=None):
        if update_forcings is not None:
            update_forcings(t + self.dt)
            self._push_bcs()
            self._push_source()
        self._sync_to_device()
        self.device.tracer_forward_euler(self.tid)
        self._device_ahead = True


class GeneralCoupledTimeIntegrator2D(TimeIntegratorBase):
    def __init__(self, solver, swe_stepper, tracer_steppers):
        self.solver = solver
        self.options = solver.options
        self.fields = solver.fields
        self.timesteppers = _AttrDict(swe2d=swe_stepper)      # AttrDict in the reference (coupled_timeintegrator_2d.py:33)
        self.timesteppers.update(tracer_steppers)
        self.swe = swe_stepper
        self.tracers = tracer_steppers
        self.device = swe_stepper.device
        print_output('Coupled time integrator: {:}'.format(self.__class__.__name__))
        if not self.options.tracer_only:
            print_output('  Shallow Water time integrator: {:}'.format(swe_stepper.__class__.__name__))
        print_output('  Tracer time integrator: {:}'.format(self.options.tracer_timestepper_type))
        o = self.options
        self.device.tracer_set_options(o.use_lax_friedrichs_tracer, float(o.lax_friedrichs_tracer_scaling_factor),
                                       _velocity_factor(o.tracer_advective_velocity_factor))

    def set_dt(self, dt):
        for stepper in sorted(self.timesteppers):
            self.timesteppers[stepper].set_dt(dt)

    def initialize(self, solution2d):
        assert solution2d == self.fields.solution_2d
        self.swe.initialize(self.fields.solution_2d)
        for label, ts in self.tracers.items():
            ts.initialize(self.fields[label])

    def advance(self, t, update_forcings=None):
        """coupled_timeintegrator_2d.py:93-113"""
        use_limiter = self.options.use_limiter_for_tracers and self.options.polynomial_degree > 0
        fused = all(ts.n_stages == 3 for ts in self.tracers.values()) and self.swe.n_stages == 3
        if update_forcings is None and fused:        # all SSPRK33: one C call per time step
            self.swe._sync_to_device()
            for ts in self.tracers.values():
                ts._sync_to_device()
            self.device.advance_coupled(1, tracer_only=self.options.tracer_only, use_limiter=use_limiter)
            self.swe._device_ahead = True
            for ts in self.tracers.values():
                ts._device_ahead = True
            return
        if not self.options.tracer_only:
            self.swe.advance(t, update_forcings=update_forcings)
        else:
            self.swe._sync_to_device()
        for label, ts in self.tracers.items():
            ts.advance(t, update_forcings=update_forcings)
            if use_limiter:
                self.device.tracer_limit(ts.tid)

    def advance_steps(self, t, n_steps):
        """``n_steps`` coupled steps without forcing updates; one library call when every stepper is SSPRK33."""
        use_limiter = self.options.use_limiter_for_tracers and self.options.polynomial_degree > 0
        fused = all(ts.n_stages == 3 for ts in self.tracers.values()) and self.swe.n_stages == 3
        if not fused:
            for i in range(int(n_steps)):
                self.advance(t + i*self.swe.dt)
            return
        self.swe._sync_to_device()
        for ts in self.tracers.values():
            ts._sync_to_device()
        self.device.advance_coupled(int(n_steps), tracer_only=self.options.tracer_only, use_limiter=use_limiter)
        self.swe._device_ahead = True
        for ts in self.tracers.values():
            ts._device_ahead = True

    def diagnostics(self):
        return self.swe.diagnostics()
