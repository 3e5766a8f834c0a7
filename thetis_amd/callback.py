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

__all__ = ['CallbackManager', 'DiagnosticCallback', 'ScalarConservationCallback', 'VolumeConservation2DCallback']


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
