"""Time-integrator API of the reference (thetis/timeintegrator.py:13-73)."""
from abc import ABC, abstractmethod

import numpy

CFL_UNCONDITIONALLY_STABLE = numpy.inf


class TimeIntegratorBase(ABC):
    """Abstract class that defines the API for all time integrators (timeintegrator.py:13-39)"""

    @abstractmethod
    def advance(self, t, update_forcings=None):
        """Advances equations for one time step"""
        pass

    @abstractmethod
    def initialize(self, init_solution):
        """Initialize the time integrator"""
        pass


class TimeIntegrator(TimeIntegratorBase):
    """Base class for all time integrator objects that march a single equation (timeintegrator.py:42-73)"""

    def __init__(self, equation, solution, fields, dt, options):
        super(TimeIntegrator, self).__init__()
        self.equation = equation
        self.solution = solution
        self.fields = fields
        self.dt = dt
        self.dt_const = dt
        self.name = '-'.join([self.__class__.__name__, self.equation.__class__.__name__])
        self.ad_block_tag = getattr(options, 'ad_block_tag', None) or self.name
        self.solver_parameters = getattr(options, 'solver_parameters', {})

    def set_dt(self, dt):
        """Update time step"""
        self.dt = dt
        self.dt_const = dt
