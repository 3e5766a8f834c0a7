"""
Run-time checks hooked into the time loop.  The reference has a general diagnostics framework (thetis/callback.py); the 2D
explicit path only ever registers three checks from ``FlowSolver2d.iterate`` (solver2d.py:1034-1059) - volume, tracer mass,
tracer over/undershoot - and all three are reductions the device already provides (``swe2d_diagnostics``,
``swe2d_tracer_diagnostics``).  So this module is one small class built around a device reduction, plus the registry the
solver iterates over; the reference's class names are kept as constructors so that user scripts written against
``thetis.callback`` (``VolumeConservation2DCallback(solver_obj, ...)``, ``solver_obj.add_callback(cb, 'export')``,
``solver_obj.callbacks['export']['volume2d']()``) keep working.  HDF5 sinks are I/O and out of scope.
"""
from .log import print_output

__all__ = ['CallbackManager', 'DeviceCheck', 'VolumeConservation2DCallback', 'TracerMassConservation2DCallback',
           'ConservativeTracerMassConservation2DCallback', 'TracerOvershootCallBack']


class CallbackManager(dict):
    """``manager[mode][name] -> check``; modes are 'export' and 'timestep'."""

    def __missing__(self, mode):
        self[mode] = {}
        return self[mode]

    def add(self, check, mode):
        self[mode][check.name] = check

    def evaluate(self, mode, index=None):
        for name in sorted(self[mode]):
            self[mode][name].evaluate(index=index)


class DeviceCheck(object):
    """A named reduction of the device-resident state compared with its value at the first evaluation.

    ``kind='conserved'``: ``reduce()`` returns a scalar; a call returns (value, relative drift).
    ``kind='bounds'``:    ``reduce()`` returns (min, max); a call returns (min, max, undershoot <= 0, overshoot >= 0)."""

    def __init__(self, name, solver_obj, reduce, kind='conserved', append_to_log=True, start_time=None, end_time=None,
                 **ignored):                       # export_to_hdf5, outputdir, ...: no file sinks on this path
        assert kind in ('conserved', 'bounds')
        self.name, self.solver_obj, self.kind = name, solver_obj, kind
        self._reduce = reduce
        self._log = append_to_log
        self._window = (-float('inf') if start_time is None else start_time, float('inf') if end_time is None else end_time)
        self.reference_value = None
        self.history = []

    def __call__(self):
        now = self._reduce()
        if self.reference_value is None:
            self.reference_value = now
        ref = self.reference_value
        if self.kind == 'conserved':
            return now, (now - ref)/ref
        lo, hi = now
        return lo, hi, min(lo - ref[0], 0.0), max(hi - ref[1], 0.0)

    def message_str(self, *values):
        if self.kind == 'conserved':
            return '{0:s} rel. error {1:11.4e}'.format(self.name, values[1])
        return '{0:s} {1:g} {2:g}'.format(self.name, values[2], values[3])

    def evaluate(self, index=None):
        t = self.solver_obj.simulation_time
        if not self._window[0] <= t <= self._window[1]:
            return
        values = self()
        self.history.append((t,) + tuple(values))
        if self._log:
            print_output(self.message_str(*values))


def _tracer_reduction(solver_obj, tracer_name, pick):
    def reduce():
        ts = solver_obj.timestepper.tracers[tracer_name]
        ts._sync_to_device()
        return pick(ts.device.tracer_diagnostics(ts.tid))        # {int T*H dx, int T dx, min, max}
    return reduce


def VolumeConservation2DCallback(solver_obj, **kwargs):
    """int (eta + h) dx (callback.py:350-364, utility.py:421-425)"""
    return DeviceCheck('volume2d', solver_obj, lambda: float(solver_obj.timestepper.diagnostics()[2]), **kwargs)


def TracerMassConservation2DCallback(tracer_name, solver_obj, **kwargs):
    """int T*H dx of a depth-averaged tracer (callback.py:366-389)"""
    return DeviceCheck(tracer_name + ' mass', solver_obj, _tracer_reduction(solver_obj, tracer_name, lambda d: float(d[0])), **kwargs)


def ConservativeTracerMassConservation2DCallback(tracer_name, solver_obj, **kwargs):
    """int q dx of a depth-integrated tracer (callback.py:392-412)"""
    return DeviceCheck(tracer_name + ' mass', solver_obj, _tracer_reduction(solver_obj, tracer_name, lambda d: float(d[1])), **kwargs)


def TracerOvershootCallBack(tracer_name, solver_obj, **kwargs):
    """nodal min/max of a tracer against their initial values (callback.py:463-483)"""
    return DeviceCheck(tracer_name + ' overshoot', solver_obj,
                       _tracer_reduction(solver_obj, tracer_name, lambda d: (float(d[2]), float(d[3]))), kind='bounds', **kwargs)
