"""
The contract between ``FlowSolver2d`` and a time stepper.

The reference's driver touches a stepper in four places only (thetis/solver2d.py:572-573, 785, 1116 and
coupled_timeintegrator_2d.py:61-68): it constructs it with ``(equation, solution, fields, dt, options, bnd_conditions)``, calls
``initialize(solution)`` once, ``advance(t, update_forcings)`` per step and ``set_dt(dt)`` when the step changes.  The two classes
below carry exactly that surface (names as in thetis/timeintegrator.py:13-73, so that reference scripts and subclasses keep
working); everything a stepper actually does lives in thetis_amd/rungekutta.py and on the device.
"""
import abc

import numpy

CFL_UNCONDITIONALLY_STABLE = numpy.inf          # what implicit steppers report as their CFL coefficient


class TimeIntegratorBase(abc.ABC):
    """What the time loop needs from any stepper: ``initialize`` and ``advance``."""

    @abc.abstractmethod
    def initialize(self, init_solution):
        """Take over the initial state (device steppers upload it here)."""

    @abc.abstractmethod
    def advance(self, t, update_forcings=None):
        """One time step starting at time ``t``; ``update_forcings(t_stage)`` is called before every stage evaluation."""


class TimeIntegrator(TimeIntegratorBase):
    """A stepper bound to ONE equation: remembers what it marches, with which step, and under which name it logs."""

    def __init__(self, equation, solution, fields, dt, options):
        self.equation, self.solution, self.fields = equation, solution, fields
        self.dt = self.dt_const = dt              # (not through set_dt: subclasses forward that to a device they do not have yet)
        self.name = '{:}-{:}'.format(type(self).__name__, type(equation).__name__)
        self.solver_parameters = dict(getattr(options, 'solver_parameters', None) or {})
        self.ad_block_tag = getattr(options, 'ad_block_tag', None) or self.name

    def set_dt(self, dt):
        """Change the time step (subclasses forward it to the device)."""
        self.dt = self.dt_const = dt
