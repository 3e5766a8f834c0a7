"""
Model options with the reference's names, defaults and error behaviour - the part of
``thetis/options.py`` + ``thetis/configuration.py`` that the 2D SWE SSPRK33 path reads.

The reference builds these on ``traitlets`` (not installed here, and control-plane rather than hot path), so this
is a small plain-Python restatement of the *behaviour*: typed/validated attributes, frozen attribute sets
(configuration.py:294-331: adding an unknown attribute raises ``TypeError``), ``update(dict)``
(configuration.py:262-275) and paired options - assigning ``swe_timestepper_type`` re-instantiates
``swe_timestepper_options`` (configuration.py:333-368, options.py:838-852).
"""
from collections import OrderedDict

__all__ = ['ModelOptions2d', 'ExplicitSWETimeStepperOptions2d', 'ExplicitTracerTimeStepperOptions2d',
           'TimeStepperOptions', 'Constant']


class Constant(object):
    """Stand-in for ``firedrake.Constant``: a mutable scalar/vector that forcing callbacks can ``assign`` to."""

    def __init__(self, value):
        self.assign(value)

    def assign(self, value):
        if isinstance(value, Constant):
            value = value._value
        try:
            self._value = float(value)
        except TypeError:
            self._value = tuple(float(v) for v in value)
        return self

    def values(self):
        return (self._value,) if isinstance(self._value, float) else self._value

    def __float__(self):
        if not isinstance(self._value, float):
            raise TypeError('vector Constant')
        return self._value

    def __iter__(self):
        return iter(self.values())

    def __getitem__(self, i):
        return self.values()[i]

    def __len__(self):
        return len(self.values())

    def __repr__(self):
        return 'Constant({:})'.format(self._value)


# ---- validators ----------------------------------------------------------------------------------------------------
def _positive_float(name, v):
    v = float(v)
    assert v > 0.0, "The '{:}' trait expected a positive float, not {:}".format(name, v)
    return v


def _positive_float_or_none(name, v):
    return None if v is None else _positive_float(name, v)


def _nonneg_int(name, v):
    assert int(v) == v and v >= 0, "The '{:}' trait expected a non-negative integer, not {:}".format(name, v)
    return int(v)


def _bool(name, v):
    if not isinstance(v, (bool,)) and v not in (0, 1):
        raise TypeError("The '{:}' trait expected a bool, not {!r}".format(name, v))
    return bool(v)


def _enum(*values):
    def check(name, v):
        if v not in values:
            raise ValueError("The '{:}' trait expected any of {:}, not {!r}".format(name, list(values), v))
        return v
    return check


def _constant(name, v):
    """FiredrakeConstantTraitlet: a Constant (floats are promoted)."""
    return v if isinstance(v, Constant) else Constant(v)


def _scalar_expr_or_none(name, v):
    """FiredrakeScalarExpression(allow_none=True): None, a number/Constant, a callable f(x, y) or a nodal array."""
    return v


def _any(name, v):
    return v


def _str(name, v):
    if not isinstance(v, str):
        raise TypeError("The '{:}' trait expected a unicode string, not {!r}".format(name, v))
    return v


def _str_list(name, v):
    v = list(v)
    for s in v:
        _str(name, s)
    return v


class FrozenOptions(object):
    """Validated, frozen attribute container (FrozenHasTraits / FrozenConfigurable behaviour)."""
    name = 'Options'
    _spec = OrderedDict()       # attribute -> (default factory or value, validator)
    _paired = {}                # enum attribute -> (slave attribute, {value: class})

    def __init__(self):
        object.__setattr__(self, '_isfrozen', False)
        for key, (default, _) in self._all_spec().items():
            value = default() if callable(default) and not isinstance(default, Constant) else default
            setattr(self, key, value)
        object.__setattr__(self, '_isfrozen', True)

    @classmethod
    def _all_spec(cls):
        spec = OrderedDict()
        for klass in reversed(cls.__mro__):
            spec.update(getattr(klass, '_spec', {}))
        return spec

    @classmethod
    def _all_paired(cls):
        paired = {}
        for klass in reversed(cls.__mro__):
            paired.update(getattr(klass, '_paired', {}))
        return paired

    def __setattr__(self, key, value):
        spec = self._all_spec()
        if key not in spec:
            if self._isfrozen and not hasattr(self, key):
                raise TypeError('Adding new attribute "{:}" to {:} class is forbidden'.format(key, self.__class__.__name__))
            object.__setattr__(self, key, value)
            return
        value = spec[key][1](key, value)
        object.__setattr__(self, key, value)
        paired = self._all_paired()
        if key in paired:
            slave, table = paired[key]
            object.__setattr__(self, slave, table[value]())

    def update(self, options):
        """Assign options from a dict or another options object (configuration.py:262-275)."""
        if isinstance(options, dict):
            params = options
        else:
            assert isinstance(options, FrozenOptions), 'options must be a dict or an options object'
            params = {k: getattr(options, k) for k in options._all_spec()}
        for key in params:
            setattr(self, key, params[key])

    def __str__(self):
        out = '{:} parameters\n'.format(self.name)
        for k in sorted(self._all_spec()):
            out += '  {:16s} : {:}\n'.format(k, getattr(self, k))
        return out


# ---- time stepper options (options.py:13-163) ----------------------------------------------------------------------
class TimeStepperOptions(FrozenOptions):
    """Base class for all time stepper options (options.py:13-21)"""
    name = 'Time stepper'
    _spec = OrderedDict([
        ('solver_parameters', (dict, _any)),
        ('ad_block_tag', (None, _any)),
    ])


class ExplicitTimeStepperOptions(TimeStepperOptions):
    """Options for explicit time integrator (options.py:24-26)"""
    _spec = OrderedDict([('use_automatic_timestep', (True, _bool))])


class ExplicitSWETimeStepperOptions2d(ExplicitTimeStepperOptions):
    """options.py:140-152.  The PETSc parameters are kept for API fidelity only: with DG the mass matrix is
    block diagonal, cg + bjacobi/ilu is an exact solve, and the device path applies the closed-form 3x3 inverse."""
    _spec = OrderedDict([('solver_parameters', (lambda: {
        'snes_type': 'ksponly', 'ksp_type': 'cg', 'pc_type': 'bjacobi', 'sub_ksp_type': 'preonly',
        'sub_pc_type': 'ilu', 'mat_type': 'aij'}, _any))])


class ExplicitTracerTimeStepperOptions2d(ExplicitTimeStepperOptions):
    """options.py:155-163"""
    _spec = OrderedDict([('solver_parameters', (lambda: {'ksp_type': 'gmres', 'pc_type': 'sor'}, _any))])


class _ImplicitPlaceholderOptions(TimeStepperOptions):
    """Options object of the implicit steppers: selectable (the enum is the reference's), not runnable on this path."""
    _spec = OrderedDict([
        ('implicitness_theta', (0.5, _any)),
        ('use_semi_implicit_linearization', (False, _bool)),
    ])


_SWE_STEPPERS = OrderedDict([
    ('SSPRK33', ExplicitSWETimeStepperOptions2d), ('ForwardEuler', ExplicitSWETimeStepperOptions2d),
    ('BackwardEuler', _ImplicitPlaceholderOptions), ('CrankNicolson', _ImplicitPlaceholderOptions),
    ('DIRK22', _ImplicitPlaceholderOptions), ('DIRK33', _ImplicitPlaceholderOptions),
    ('SteadyState', _ImplicitPlaceholderOptions), ('PressureProjectionPicard', _ImplicitPlaceholderOptions),
    ('SSPIMEX', _ImplicitPlaceholderOptions)])
_TRACER_STEPPERS = OrderedDict([
    ('SSPRK33', ExplicitTracerTimeStepperOptions2d), ('ForwardEuler', ExplicitTracerTimeStepperOptions2d),
    ('BackwardEuler', _ImplicitPlaceholderOptions), ('CrankNicolson', _ImplicitPlaceholderOptions),
    ('DIRK22', _ImplicitPlaceholderOptions), ('DIRK33', _ImplicitPlaceholderOptions),
    ('SteadyState', _ImplicitPlaceholderOptions)])


class TracerFieldOptions(object):
    """options.py:544-572 (metadata, function, source, diffusivity, use_conservative_form)."""

    def __init__(self):
        self.metadata = {}
        self.function = None
        self.source = None
        self.diffusivity = None
        self.use_conservative_form = False


class CommonModelOptions(FrozenOptions):
    """Options that are common for both 2d and 3d models (options.py:583-733), hot-path subset + API names."""
    name = 'Model options'
    _spec = OrderedDict([
        ('polynomial_degree', (1, _nonneg_int)),
        ('element_family', ('dg-dg', _enum('dg-dg', 'rt-dg', 'bdm-dg', 'dg-cg'))),
        ('use_nonlinear_equations', (True, _bool)),
        ('use_grad_div_viscosity_term', (False, _bool)),
        ('use_grad_depth_viscosity_term', (True, _bool)),
        ('use_lax_friedrichs_velocity', (True, _bool)),
        ('lax_friedrichs_velocity_scaling_factor', (lambda: Constant(1.0), _constant)),
        ('use_lax_friedrichs_tracer', (False, _bool)),
        ('lax_friedrichs_tracer_scaling_factor', (lambda: Constant(1.0), _constant)),
        ('use_limiter_for_tracers', (True, _bool)),
        ('check_volume_conservation_2d', (False, _bool)),
        ('log_output', (True, _bool)),
        ('timestep', (10.0, _positive_float)),
        ('cfl_2d', (1.0, _positive_float)),
        ('simulation_initial_date', (None, _any)),
        ('simulation_end_date', (None, _any)),
        ('simulation_export_time', (100.0, _positive_float)),
        ('simulation_end_time', (None, _positive_float_or_none)),
        ('horizontal_velocity_scale', (lambda: Constant(0.1), _constant)),
        ('horizontal_viscosity_scale', (lambda: Constant(1.0), _constant)),
        ('horizontal_diffusivity_scale', (lambda: Constant(1.0), _constant)),
        ('output_directory', ('outputs', _str)),
        ('no_exports', (False, _bool)),
        ('export_diagnostics', (True, _bool)),
        ('fields_to_export', (lambda: ['elev_2d', 'uv_2d', 'uv_3d', 'w_3d'], _str_list)),
        ('fields_to_export_hdf5', (list, _str_list)),
        ('verbose', (0, _any)),
        ('linear_drag_coefficient', (None, _scalar_expr_or_none)),
        ('quadratic_drag_coefficient', (None, _scalar_expr_or_none)),
        ('manning_drag_coefficient', (None, _scalar_expr_or_none)),
        ('nikuradse_bed_roughness', (None, _scalar_expr_or_none)),
        ('norm_smoother', (lambda: Constant(0.0), _constant)),
        ('horizontal_viscosity', (None, _scalar_expr_or_none)),
        ('coriolis_frequency', (None, _scalar_expr_or_none)),
        ('wind_stress', (None, _scalar_expr_or_none)),
        ('atmospheric_pressure', (None, _scalar_expr_or_none)),
        ('momentum_source_2d', (None, _scalar_expr_or_none)),
        ('volume_source_2d', (None, _scalar_expr_or_none)),
        ('sipg_factor', (lambda: Constant(1.0), _constant)),
        ('sipg_factor_tracer', (lambda: Constant(1.0), _constant)),
    ])


class ModelOptions2d(CommonModelOptions):
    """Options for 2D depth-averaged shallow water model (options.py:866-1041)"""
    name = 'Depth-averaged 2D model'
    _spec = OrderedDict([
        ('use_tracer_conservative_form', (False, _bool)),
        ('use_wetting_and_drying', (False, _bool)),
        ('wetting_and_drying_alpha', (lambda: Constant(0.5), _any)),
        ('use_automatic_wetting_and_drying_alpha', (False, _bool)),
        ('wetting_and_drying_alpha_min', (None, _any)),
        ('wetting_and_drying_alpha_max', (lambda: Constant(2.0), _any)),
        # this build's explicit wetting-drying: factor applied to the automatic time step (not a reference option)
        ('wetting_and_drying_cfl_factor', (0.4, _positive_float)),
        ('check_tracer_conservation', (False, _bool)),
        ('tracer_advective_velocity_factor', (lambda: Constant(1.0), _any)),
        ('check_tracer_overshoot', (False, _bool)),
        ('tracer_only', (False, _bool)),
        ('tracer_element_family', ('dg', _enum('dg', 'cg'))),
        ('use_supg_tracer', (False, _bool)),
        ('tracer_picard_iterations', (1, _nonneg_int)),
        # paired enums last so that their slaves exist after construction
        ('swe_timestepper_type', ('CrankNicolson', _enum(*_SWE_STEPPERS))),
        ('tracer_timestepper_type', ('CrankNicolson', _enum(*_TRACER_STEPPERS))),
    ])
    _paired = {
        'swe_timestepper_type': ('swe_timestepper_options', _SWE_STEPPERS),
        'tracer_timestepper_type': ('tracer_timestepper_options', _TRACER_STEPPERS),
    }

    def __init__(self):
        object.__setattr__(self, 'tracer', OrderedDict())
        object.__setattr__(self, 'tracer_fields', OrderedDict())
        object.__setattr__(self, 'swe_timestepper_options', None)
        object.__setattr__(self, 'tracer_timestepper_options', None)
        super().__init__()

    def add_tracer_2d(self, label, name, filename, shortname=None, unit='-', **kwargs):
        """Add a 2D tracer field to :attr:`tracer` (options.py:945-983)."""
        assert isinstance(label, str)
        assert isinstance(name, str)
        assert isinstance(filename, str)
        assert shortname is None or isinstance(shortname, str)
        assert isinstance(unit, str)
        assert label not in self.tracer, "Field '{:}' already exists.".format(label)
        assert ' ' not in label, "Labels cannot contain spaces"
        assert ',' not in label, "Labels cannot contain commas"
        assert ' ' not in filename, "Filenames cannot contain spaces"
        self.tracer[label] = TracerFieldOptions()
        self.tracer[label].metadata = {'name': name, 'shortname': shortname or name, 'unit': unit, 'filename': filename}
        self.tracer[label].function = kwargs.get('function')
        self.tracer[label].source = kwargs.get('source')
        self.tracer[label].diffusivity = kwargs.get('diffusivity')
        self.tracer[label].use_conservative_form = kwargs.get('use_conservative_form', False)
        if not kwargs.get('mixed', False):
            self.tracer_fields[label] = self.tracer[label].function

    def set_timestepper_type(self, timestepper_type, **kwargs):
        """Set the same timestepper type for all components (options.py:1017-1041)."""
        self.swe_timestepper_type = timestepper_type
        self.tracer_timestepper_type = timestepper_type
        for key, value in kwargs.items():
            for option in (self.swe_timestepper_options, self.tracer_timestepper_options):
                setattr(option, key, value)
