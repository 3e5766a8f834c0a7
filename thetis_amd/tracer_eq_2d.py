"""
Descriptor of the 2D tracer equation for the device path (thetis/tracer_eq_2d.py).

Implemented by ``swe_tracer_stage_kernel`` (csrc/swe2d_kernels.h): non-conservative ``HorizontalAdvectionTerm``
(:124-193; upwind DG, optional Lax-Friedrichs) and ``SourceTerm`` (:281-298), boundaries without a condition or with a
constant ``'value'``; ``HorizontalDiffusionTerm`` (SIPG, :196-278) by the pass kernel ``swe_sipg_kernel<1>``
(csrc/swe2d_sipg.h; Constant or CG-P1 diffusivity, ``'diff_flux'`` boundaries); the conservative form
(:325-437) is a flag of the same kernels.  Everything else (SUPG :490-501, CG tracers, Function-valued velocity boundary keys) raises instead of silently changing the physics.
"""
from .function import Function
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
        # conservative form (q = H*T; ConservativeHorizontalAdvectionTerm :341-395, ConservativeSourceTerm :424-437):
        # a flag of the same stage kernels
        self.conservative = bool(topts.use_conservative_form)
        if options.use_supg_tracer:
            raise NotImplementedError('SUPG stabilisation applies to CG tracers only')

    @staticmethod
    def check_bnd_conditions(bnd_conditions):
        for marker, funcs in (bnd_conditions or {}).items():
            for key, v in funcs.items():
                if key not in ('value', 'elev', 'diff_flux', 'uv', 'un', 'flux'):
                    raise NotImplementedError('tracer boundary key {!r} is not on the device path '
                                              '("value", "uv", "un", "flux", "elev", "diff_flux")'.format(key))
                if key == 'elev' and isinstance(v, Function):
                    raise NotImplementedError("tracer boundary 'elev' must be a constant on the device path")
                if key == 'diff_flux' and not isinstance(v, (int, float, Constant)):
                    raise NotImplementedError("'diff_flux' must be a constant on the device path")
                if key == 'value' and not isinstance(v, (int, float, Constant, Function)):
                    raise NotImplementedError('tracer boundary values must be Constants or Functions on the device path')
