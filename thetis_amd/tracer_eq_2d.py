"""
Descriptor of the 2D tracer equation for the device path (thetis/tracer_eq_2d.py).

Implemented by ``swe_tracer_stage_kernel`` (csrc/swe2d_kernels.h): non-conservative ``HorizontalAdvectionTerm``
(:124-193; upwind DG, optional Lax-Friedrichs) and ``SourceTerm`` (:281-298), boundaries without a condition or with a
constant ``'value'``.  Everything else (SIPG diffusion :196-278, conservative form :325-445, SUPG :490-501, CG tracers,
velocity-type boundary keys) raises instead of silently changing the physics.
"""
from .options import Constant

__all__ = ['TracerEquation2D']


class TracerEquation2D(object):
    def __init__(self, label, function_space, depth, options, velocity=None):
        self.label = label
        self.function_space = function_space
        self.mesh = function_space.mesh()
        self.depth = depth
        self.options = options
        topts = options.tracer[label]
        if options.tracer_element_family != 'dg':
            raise NotImplementedError("tracer_element_family='cg' is not on the device path")
        if topts.use_conservative_form or options.use_tracer_conservative_form:
            raise NotImplementedError('the conservative tracer form is not on the device path yet')
        if topts.diffusivity is not None:
            raise NotImplementedError('horizontal tracer diffusion (SIPG) is not on the device path yet')
        if options.use_supg_tracer:
            raise NotImplementedError('SUPG stabilisation applies to CG tracers only')

    @staticmethod
    def check_bnd_conditions(bnd_conditions):
        for marker, funcs in (bnd_conditions or {}).items():
            for key, v in funcs.items():
                if key not in ('value', 'elev'):
                    raise NotImplementedError('tracer boundary key {!r} is not on the device path (only "value")'.format(key))
                if key == 'value' and not isinstance(v, (int, float, Constant)):
                    raise NotImplementedError('tracer boundary values must be constants on the device path')
