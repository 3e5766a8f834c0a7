"""
Callbacks evaluated from the time loop (thetis/callback.py).  Only the pieces the 2D explicit path uses:
``CallbackManager`` (:14-58), ``DiagnosticCallback`` without the HDF5 sink (:62-239; h5py is absent and I/O is out of
scope), ``ScalarConservationCallback`` (:299-330) and ``VolumeConservation2DCallback`` (:350-364), whose integral
``int (eta + h) dx`` (utility.py:421-425) is reduced on the device (swe2d_diagnostics).
"""
from abc import ABC, abstractmethod
from collections import OrderedDict, defaultdict

import numpy

from .log import print_output

__all__ = ['CallbackManager', 'DiagnosticCallback', 'ScalarConservationCallback', 'VolumeConservation2DCallback',
           'TracerMassConservation2DCallback', 'ConservativeTracerMassConservation2DCallback', 'MinMaxConservationCallback', 'TracerOvershootCallBack']


class CallbackManager(defaultdict):
    def __init__(self):
        super(CallbackManager, self).__init__(OrderedDict)

    def add(self, callback, mode):
        self[mode][callback.name] = callback

    def evaluate(self, mode, index=None):
        for key in sorted(self[mode]):
            self[mode][key].evaluate(index=index)


class DiagnosticCallback(ABC):
    def __init__(self, solver_obj, array_dim=1, attrs=None, outputdir=None, export_to_hdf5=False,
                 append_to_log=True, include_time=True, hdf5_dtype='d', start_time=None, end_time=None):
        self.solver_obj = solver_obj
        self.append_to_log = append_to_log
        self.append_to_hdf5 = False          # no HDF5 sink on this path
        self.start_time = start_time or -numpy.inf
        self.end_time = end_time or numpy.inf
        self.history = []

    @property
    @abstractmethod
    def name(self):
        pass

    @abstractmethod
    def __call__(self):
        pass

    @abstractmethod
    def message_str(self, *args):
        return '{} diagnostic'.format(self.name)

    def push_to_log(self, time, args):
        print_output(self.message_str(*args))

    def evaluate(self, index=None):
        time = self.solver_obj.simulation_time
        if time < self.start_time or time > self.end_time:
            return
        values = self.__call__()
        self.history.append((time,) + tuple(values))
        if self.append_to_log:
            self.push_to_log(time, values)


class ScalarConservationCallback(DiagnosticCallback):
    """Base class for callbacks that check conservation of a scalar quantity (callback.py:299-330)"""
    variable_names = ['integral', 'relative_difference']

    def __init__(self, scalar_callback, solver_obj, **kwargs):
        super(ScalarConservationCallback, self).__init__(solver_obj, **kwargs)
        self.scalar_callback = scalar_callback
        self.initial_value = None

    def __call__(self):
        value = self.scalar_callback()
        if self.initial_value is None:
            self.initial_value = value
        rel_diff = (value - self.initial_value)/self.initial_value
        return value, rel_diff

    def message_str(self, *args):
        return '{0:s} rel. error {1:11.4e}'.format(self.name, args[1])


class VolumeConservation2DCallback(ScalarConservationCallback):
    """Checks conservation of 2D volume (integral of water elevation field) (callback.py:350-364)"""
    name = 'volume2d'

    def __init__(self, solver_obj, **kwargs):
        def vol2d():
            return float(self.solver_obj.timestepper.diagnostics()[2])
        super(VolumeConservation2DCallback, self).__init__(vol2d, solver_obj, **kwargs)


class TracerMassConservation2DCallback(ScalarConservationCallback):
    """Checks conservation of depth-averaged tracer mass = int T*H dx (callback.py:366-389), reduced on the device."""
    name = 'tracer mass'

    def __init__(self, tracer_name, solver_obj, **kwargs):
        self.name = tracer_name + ' mass'

        def mass():
            ts = solver_obj.timestepper.tracers[tracer_name]
            ts._sync_to_device()
            return float(ts.device.tracer_diagnostics(ts.tid)[0])
        super(TracerMassConservation2DCallback, self).__init__(mass, solver_obj, **kwargs)


class ConservativeTracerMassConservation2DCallback(ScalarConservationCallback):
    """Conservative (depth-integrated) tracer: mass = int q dx (callback.py:392-412), reduced on the device."""
    name = 'tracer mass'

    def __init__(self, tracer_name, solver_obj, **kwargs):
        self.name = tracer_name + ' mass'

        def mass():
            ts = solver_obj.timestepper.tracers[tracer_name]
            ts._sync_to_device()
            return float(ts.device.tracer_diagnostics(ts.tid)[1])
        super(ConservativeTracerMassConservation2DCallback, self).__init__(mass, solver_obj, **kwargs)


class MinMaxConservationCallback(DiagnosticCallback):
    """Base class for callbacks that check conservation of a minimum/maximum (callback.py:433-460)"""
    variable_names = ['min_value', 'max_value', 'undershoot', 'overshoot']

    def __init__(self, minmax_callback, solver_obj, **kwargs):
        super(MinMaxConservationCallback, self).__init__(solver_obj, **kwargs)
        self.minmax_callback = minmax_callback
        self.initial_value = None

    def __call__(self):
        value = self.minmax_callback()
        if self.initial_value is None:
            self.initial_value = value
        overshoot = max(value[1] - self.initial_value[1], 0.0)
        undershoot = min(value[0] - self.initial_value[0], 0.0)
        return value[0], value[1], undershoot, overshoot

    def message_str(self, *args):
        return '{0:s} {1:g} {2:g}'.format(self.name, args[2], args[3])


class TracerOvershootCallBack(MinMaxConservationCallback):
    """Checks overshoots of the given tracer field (callback.py:463-483): nodal min/max, reduced on the device."""
    name = 'tracer overshoot'

    def __init__(self, tracer_name, solver_obj, **kwargs):
        self.name = tracer_name + ' overshoot'

        def minmax():
            ts = solver_obj.timestepper.tracers[tracer_name]
            ts._sync_to_device()
            d = ts.device.tracer_diagnostics(ts.tid)
            return float(d[2]), float(d[3])
        super(TracerOvershootCallBack, self).__init__(minmax, solver_obj, **kwargs)
