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

__all__ = ['CallbackManager', 'DiagnosticCallback', 'ScalarConservationCallback', 'MinMaxConservationCallback', 'DeviceCheck',
           'VolumeConservation2DCallback', 'TracerMassConservation2DCallback',
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


class DiagnosticCallback(object):
    """The extension point user scripts subclass (thetis/callback.py:162-301): a subclass provides ``name``, ``__call__()``
    (returns the tuple of diagnostic values; any reduction over ranks happens in there) and ``message_str(*values)``;
    ``evaluate`` is what the time loop calls - inside the optional [start_time, end_time] window it evaluates, keeps the
    values in ``history`` and prints the message.  ``variable_names`` is accepted for compatibility; there is no HDF5 sink on
    this path (``export_to_hdf5``, ``outputdir``, ``attrs``, ``array_dim``, ``hdf5_dtype``, ``include_time`` are taken and ignored)."""
    name = 'diagnostic'
    variable_names = ()

    def __init__(self, solver_obj, append_to_log=True, start_time=None, end_time=None, **ignored):
        self.solver_obj = solver_obj
        self.append_to_log = append_to_log
        self.start_time = -float('inf') if start_time is None else start_time
        self.end_time = float('inf') if end_time is None else end_time
        self.history = []

    def __call__(self):
        raise NotImplementedError('a DiagnosticCallback subclass must implement __call__')

    def message_str(self, *values):
        return '{:} diagnostic'.format(self.name)

    def push_to_log(self, time, values):
        print_output(self.message_str(*values))

    def evaluate(self, index=None):
        t = self.solver_obj.simulation_time
        if t < self.start_time or t > self.end_time:
            return
        values = self()
        values = tuple(values) if isinstance(values, (tuple, list)) else (values,)
        self.history.append((t,) + values)
        if self.append_to_log:
            self.push_to_log(t, values)


class ScalarConservationCallback(DiagnosticCallback):
    """``scalar_callback()`` against its first value: returns (value, relative difference) (thetis/callback.py:304-332)."""
    variable_names = ['integral', 'relative_difference']

    def __init__(self, scalar_callback, solver_obj, **kwargs):
        super(ScalarConservationCallback, self).__init__(solver_obj, **kwargs)
        self.scalar_callback = scalar_callback
        self.initial_value = None

    def __call__(self):
        now = self.scalar_callback()
        if self.initial_value is None:
            self.initial_value = now
        return now, (now - self.initial_value)/self.initial_value

    def message_str(self, *values):
        return '{0:s} rel. error {1:11.4e}'.format(self.name, values[1])


class MinMaxConservationCallback(DiagnosticCallback):
    """``minmax_callback()`` -> (min, max) against the first pair: returns (min, max, undershoot <= 0, overshoot >= 0)
    (thetis/callback.py:415-460)."""
    variable_names = ['min_value', 'max_value', 'undershoot', 'overshoot']

    def __init__(self, minmax_callback, solver_obj, **kwargs):
        super(MinMaxConservationCallback, self).__init__(solver_obj, **kwargs)
        self.minmax_callback = minmax_callback
        self.initial_value = None

    def __call__(self):
        lo, hi = self.minmax_callback()
        if self.initial_value is None:
            self.initial_value = (lo, hi)
        return lo, hi, min(lo - self.initial_value[0], 0.0), max(hi - self.initial_value[1], 0.0)

    def message_str(self, *values):
        return '{0:s} {1:g} {2:g}'.format(self.name, values[2], values[3])


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

    @property
    def initial_value(self):                   # the reference's attribute name (callback.py:320)
        return self.reference_value

    @initial_value.setter
    def initial_value(self, value):
        self.reference_value = value

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
