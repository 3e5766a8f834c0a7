"""
User-script shaped scenarios run once on ONE rank and once partitioned over N ranks (tests/test_spmd.py on the CPU with the host
stand-in device, tests/test_gpu_spmd.py on the GPU with the HIP library): the same ``FlowSolver2d`` code in both runs, as an
unchanged script under ``mpiexec -n N`` in the reference (examples/README.md:51-56).  ``run(name, outdir)`` returns everything a
user could observe: fields, iteration / time / export counters, callback histories, the files written.
"""
import hashlib
import math
import os

import numpy as np

from thetis_amd import Constant, Function, RectangleMesh, UnitSquareMesh, get_functionspace, solver2d
from thetis_amd import callback as cb_mod


def _channel(outdir, nx=24, ny=3, export=True, cpu=False):
    """examples/channel2d.py: closed channel, sloping bed, automatic CFL time step, volume check, exports"""
    lx, ly = 100e3, 3750.0
    mesh2d = RectangleMesh(nx, ny, lx, ly)
    P1_2d = get_functionspace(mesh2d, 'CG', 1)
    bathymetry_2d = Function(P1_2d, name='Bathymetry').interpolate(lambda x, y: 20.0 + (5.0 - 20.0)*x/lx)
    solver_obj = solver2d.FlowSolver2d(mesh2d, bathymetry_2d)
    o = solver_obj.options
    o.simulation_export_time = 100.0
    o.simulation_end_time = 400.0
    o.horizontal_velocity_scale = Constant(6.0)
    o.check_volume_conservation_2d = True
    o.fields_to_export = ['uv_2d', 'elev_2d']
    o.fields_to_export_hdf5 = ['uv_2d', 'elev_2d']
    o.output_directory = outdir
    o.no_exports = not export
    o.swe_timestepper_type = 'SSPRK33'
    elev_init = Function(P1_2d).interpolate(lambda x, y: np.where(x < 30e3, 6.0*(1 - x/30e3), 0.0))
    solver_obj.assign_initial_conditions(elev=elev_init)
    solver_obj.iterate()
    return solver_obj


def _forced(outdir, nx=20, ny=4, cpu=False, stepper='SSPRK33'):
    """tidal channel: elevation prescribed on the deep end by a Constant that ``update_forcings`` moves at every stage
    (examples/balzano/balzano.py:80-104 without the beach), a flux on the other end, linear drag, a Coriolis FIELD, a per-time-step
    callback - the stage-by-stage path of a partitioned run"""
    lx, ly = 40e3, 8e3
    mesh2d = RectangleMesh(nx, ny, lx, ly)
    P1_2d = get_functionspace(mesh2d, 'CG', 1)
    bathymetry_2d = Function(P1_2d).interpolate(lambda x, y: 12.0 - 4.0*x/lx + 0.5*np.sin(y/1500.0))
    solver_obj = solver2d.FlowSolver2d(mesh2d, bathymetry_2d)
    o = solver_obj.options
    o.swe_timestepper_type = stepper
    o.swe_timestepper_options.use_automatic_timestep = False
    o.timestep = 20.0 if stepper == 'SSPRK33' else 8.0
    o.simulation_end_time = 600.0
    o.simulation_export_time = 200.0
    o.linear_drag_coefficient = Constant(1e-4)
    o.coriolis_frequency = Function(P1_2d).interpolate(lambda x, y: 1e-4 + 2e-11*y)
    o.check_volume_conservation_2d = True
    o.no_exports = True
    o.output_directory = outdir
    bnd_elev = Constant(0.0)
    solver_obj.bnd_functions['shallow_water'] = {2: {'elev': bnd_elev}, 1: {'flux': Constant(-30.0)}}
    solver_obj.assign_initial_conditions(elev=Constant(0.0))
    seen = []

    class StepProbe(cb_mod.DiagnosticCallback):
        name = 'probe'

        def __call__(self):
            d = solver_obj.timestepper.diagnostics()
            return float(d[0]), float(d[1])

        def message_str(self, *v):
            return 'probe {:.6e} {:.6e}'.format(*v)
    solver_obj.add_callback(StepProbe(solver_obj, append_to_log=False), 'timestep')

    def update_forcings(t):
        seen.append(t)
        bnd_elev.assign(0.8*math.sin(2*math.pi*t/1800.0))
    solver_obj.iterate(update_forcings=update_forcings)
    solver_obj._forcing_times = seen
    return solver_obj


def _tracer(outdir, nx=18, ny=6, cpu=False, forced=False, limiter=True, stepper='SSPRK33'):
    """shallow water + one passive tracer + vertex limiter (test/tracerEq/test_consistency_2d.py shape), mass and overshoot
    checks; ``forced``: a tracer source Constant that ``update_forcings`` changes (coupled step driven stage by stage)"""
    lx, ly = 30e3, 10e3
    mesh2d = RectangleMesh(nx, ny, lx, ly)
    P1_2d = get_functionspace(mesh2d, 'CG', 1)
    bathymetry_2d = Function(P1_2d).interpolate(lambda x, y: 10.0 + 2.0*np.cos(math.pi*x/lx))
    solver_obj = solver2d.FlowSolver2d(mesh2d, bathymetry_2d)
    o = solver_obj.options
    src = Constant(0.0)
    o.add_tracer_2d('tracer_2d', 'Depth averaged tracer', 'Tracer2d', source=src if forced else None, diffusivity=None)
    o.swe_timestepper_type = stepper
    o.tracer_timestepper_type = stepper
    o.swe_timestepper_options.use_automatic_timestep = False
    o.tracer_timestepper_options.use_automatic_timestep = False
    o.timestep = 15.0 if stepper == 'SSPRK33' else 5.0
    o.simulation_end_time = 300.0
    o.simulation_export_time = 100.0
    o.use_limiter_for_tracers = limiter
    o.check_tracer_conservation = True
    o.check_tracer_overshoot = True
    o.check_volume_conservation_2d = True
    o.no_exports = True
    o.output_directory = outdir
    elev0 = Function(P1_2d).interpolate(lambda x, y: 0.3*np.exp(-((x - 0.4*lx)**2 + (y - 0.5*ly)**2)/(3e3)**2))
    q0 = Function(P1_2d).interpolate(lambda x, y: 1.0 + 1.0*(x > 0.5*lx))
    solver_obj.assign_initial_conditions(elev=elev0, tracer=q0)
    if forced:
        solver_obj.iterate(update_forcings=lambda t: src.assign(1e-4*math.sin(t/100.0)))
    else:
        solver_obj.iterate()
    return solver_obj


def _tracer_only(outdir, n=12, cpu=False):
    """examples/tracer2d.py (demos/demo_2d_tracer.py): quadrilaterals, frozen rotating velocity, tracer only, limiter"""
    mesh2d = UnitSquareMesh(n, n, quadrilateral=True)
    P1_2d = get_functionspace(mesh2d, 'CG', 1)
    solver_obj = solver2d.FlowSolver2d(mesh2d, Function(P1_2d).assign(1.0))
    o = solver_obj.options
    o.add_tracer_2d('tracer_2d', 'Depth averaged tracer', 'Tracer2d', source=None, diffusivity=None)
    o.tracer_only = True
    o.no_exports = True
    o.output_directory = outdir
    o.tracer_timestepper_type = 'SSPRK33'
    o.timestep = math.pi/300.0*40.0/n
    o.simulation_end_time = 30*o.timestep
    o.simulation_export_time = 10*o.timestep
    o.tracer_timestepper_options.use_automatic_timestep = False
    o.use_lax_friedrichs_tracer = False
    o.use_limiter_for_tracers = True
    q0 = Function(P1_2d).interpolate(lambda x, y: 1.0 + 1.0*((x - 0.5)**2 + (y - 0.75)**2 < 0.15**2))
    solver_obj.assign_initial_conditions(uv=lambda x, y: (0.5 - y, x - 0.5), tracer_2d=q0)
    solver_obj.iterate()
    return solver_obj


def _balzano(outdir, nx=12, ny=6, cpu=False):
    """examples/balzano.py: wetting-drying beach, Manning friction, tide through ``update_forcings`` (GPU only)"""
    lx, ly = 13800.0, 7200.0
    mesh2d = RectangleMesh(nx, ny, lx, ly)
    bathymetry = Function(get_functionspace(mesh2d, 'CG', 1)).interpolate(lambda x, y: x/2760.0)
    solver_obj = solver2d.FlowSolver2d(mesh2d, bathymetry)
    o = solver_obj.options
    o.swe_timestepper_type = 'SSPRK33'
    o.swe_timestepper_options.use_automatic_timestep = False
    o.timestep = 10.0
    o.simulation_end_time = 1800.0
    o.simulation_export_time = 600.0
    o.use_wetting_and_drying = True
    o.wetting_and_drying_alpha = Constant(0.4)
    o.manning_drag_coefficient = Constant(0.02)
    o.check_volume_conservation_2d = True
    o.no_exports = True
    o.output_directory = outdir
    bnd_elev = Constant(0.0)
    solver_obj.bnd_functions['shallow_water'] = {2: {'elev': bnd_elev}}
    solver_obj.assign_initial_conditions(elev=Constant(0.0))
    solver_obj.iterate(update_forcings=lambda t: bnd_elev.assign(-2.0*math.sin(2*math.pi*t/43200.0)))
    return solver_obj


def _restart(outdir, cpu=False):
    """examples/channel2d.py stopped after two exports and restarted from its checkpoint files by a NEW solver object
    (``load_state``, solver2d.py:820-921): on partitioned runs rank 0 writes the files, every rank reads them"""
    first = _channel(outdir, export=True)
    lx = 100e3
    mesh2d = RectangleMesh(24, 3, lx, 3750.0)
    P1_2d = get_functionspace(mesh2d, 'CG', 1)
    bathymetry_2d = Function(P1_2d, name='Bathymetry').interpolate(lambda x, y: 20.0 + (5.0 - 20.0)*x/lx)
    solver_obj = solver2d.FlowSolver2d(mesh2d, bathymetry_2d)
    o = solver_obj.options
    o.simulation_export_time = 100.0
    o.simulation_end_time = 600.0
    o.horizontal_velocity_scale = Constant(6.0)
    o.check_volume_conservation_2d = True
    o.fields_to_export = ['uv_2d', 'elev_2d']
    o.fields_to_export_hdf5 = ['uv_2d', 'elev_2d']
    o.output_directory = outdir
    o.swe_timestepper_type = 'SSPRK33'
    solver_obj.load_state(2)
    assert solver_obj.i_export == 2 and solver_obj.iteration == int(np.ceil(200.0/first.dt))
    solver_obj.iterate()
    return solver_obj


def _fields(outdir, nx=20, ny=6, cpu=False):
    """Function-valued data that ``update_forcings`` changes before every stage (GPU only): a tidal elevation FIELD on the open
    boundary (CG-P1 -> the values at the end nodes of the boundary facets), a wind stress field (CG-P1 vector -> per-vertex upload),
    an atmospheric pressure field held in DG-P1 (nodal upload), a spatially varying Manning coefficient and a viscosity field"""
    lx, ly = 40e3, 12e3
    mesh2d = RectangleMesh(nx, ny, lx, ly)
    P1_2d = get_functionspace(mesh2d, 'CG', 1)
    P1v_2d = get_functionspace(mesh2d, 'CG', 1, vector=True)
    P1DG_2d = get_functionspace(mesh2d, 'DG', 1)
    bathymetry_2d = Function(P1_2d).interpolate(lambda x, y: 15.0 - 5.0*x/lx + np.cos(y/2000.0))
    solver_obj = solver2d.FlowSolver2d(mesh2d, bathymetry_2d)
    o = solver_obj.options
    o.swe_timestepper_type = 'SSPRK33'
    o.swe_timestepper_options.use_automatic_timestep = False
    o.timestep = 12.0
    o.simulation_end_time = 360.0
    o.simulation_export_time = 120.0
    o.no_exports = True
    o.output_directory = outdir
    o.check_volume_conservation_2d = True
    tide = Function(P1_2d)
    wind = Function(P1v_2d)
    patm = Function(P1DG_2d)
    o.wind_stress = wind
    o.atmospheric_pressure = patm
    o.manning_drag_coefficient = Function(P1_2d).interpolate(lambda x, y: 0.02 + 0.01*x/lx)
    o.horizontal_viscosity = Function(P1_2d).interpolate(lambda x, y: 5.0 + 10.0*y/ly)
    solver_obj.bnd_functions['shallow_water'] = {1: {'elev': tide}, 2: {'un': Constant(0.01)}}
    solver_obj.assign_initial_conditions(elev=Constant(0.0))

    def update_forcings(t):
        tide.interpolate(lambda x, y: 0.5*math.sin(2*math.pi*t/3600.0)*(1.0 + 0.1*y/ly))
        wind.interpolate(lambda x, y: (0.1*math.cos(t/500.0) + 0*x, 0.05*np.sin(math.pi*x/lx)))
        patm.interpolate(lambda x, y: 50.0*np.sin(math.pi*y/ly)*math.sin(t/400.0))
    solver_obj.iterate(update_forcings=update_forcings)
    return solver_obj


def _periodic(outdir, nx=24, ny=8, cpu=False, tracer=True):
    """a channel periodic in x (test/swe2d/test_rossby_wave.py's mesh type) on an f-plane, with a tracer and the limiter: the first
    and the last strip are neighbours across the seam, the limiter's vertex patches wrap around it"""
    from thetis_amd import PeriodicRectangleMesh
    lx, ly = 48e3, 16e3
    mesh2d = PeriodicRectangleMesh(nx, ny, lx, ly, direction='x')
    P1_2d = get_functionspace(mesh2d, 'CG', 1)
    bathymetry_2d = Function(P1_2d).assign(30.0)
    solver_obj = solver2d.FlowSolver2d(mesh2d, bathymetry_2d)
    o = solver_obj.options
    if tracer:
        o.add_tracer_2d('tracer_2d', 'Depth averaged tracer', 'Tracer2d', source=None, diffusivity=None)
        o.tracer_timestepper_type = 'SSPRK33'
        o.tracer_timestepper_options.use_automatic_timestep = False
        o.check_tracer_conservation = True
        o.check_tracer_overshoot = True
    o.swe_timestepper_type = 'SSPRK33'
    o.swe_timestepper_options.use_automatic_timestep = False
    o.timestep = 5.0
    o.simulation_end_time = 150.0
    o.simulation_export_time = 50.0
    o.coriolis_frequency = Constant(1e-4)
    o.check_volume_conservation_2d = True
    o.no_exports = True
    o.output_directory = outdir
    bump = lambda x, y: 0.4*np.exp(-((np.minimum(x, lx - x))**2 + (y - 0.5*ly)**2)/(4e3)**2)       # sits ON the seam
    kw = {'tracer': Function(P1_2d).interpolate(lambda x, y: 1.0 + 1.0*(np.abs(x - 0.5*lx) > 0.3*lx))} if tracer else {}
    solver_obj.assign_initial_conditions(elev=Function(P1_2d).interpolate(bump), uv=Constant((0.3, 0.0)), **kw)
    solver_obj.iterate()
    return solver_obj


def _coast(outdir, cpu=False):
    """an unstructured mesh read from a Gmsh file (tests/golden/coast.msh, the shape of demos/north_sea.msh: arbitrary marker ids),
    cut into compact parts by recursive coordinate bisection: open boundary with a tide through ``update_forcings`` in the first half
    of the run (stage by stage), batches in the second, tracer with the limiter"""
    from thetis_amd import Mesh
    mesh2d = Mesh(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'coast.msh'))
    P1_2d = get_functionspace(mesh2d, 'CG', 1)
    x, y = mesh2d.vertex_xy.T
    bathymetry_2d = Function(P1_2d).assign(12.0 + 6.0*(x - x.min())/(x.max() - x.min()))
    solver_obj = solver2d.FlowSolver2d(mesh2d, bathymetry_2d)
    o = solver_obj.options
    o.add_tracer_2d('tracer_2d', 'Depth averaged tracer', 'Tracer2d', source=None, diffusivity=None)
    o.swe_timestepper_type = 'SSPRK33'
    o.tracer_timestepper_type = 'SSPRK33'
    o.swe_timestepper_options.use_automatic_timestep = False     # (the CG-P1 projection behind the automatic step undershoots to a
    o.tracer_timestepper_options.use_automatic_timestep = False  #  negative number on this strongly graded mesh: FlowSolver2d raises)
    o.timestep = 0.4
    o.simulation_export_time = 20.0
    o.simulation_end_time = 40.0
    o.check_volume_conservation_2d = True
    o.check_tracer_conservation = True
    o.no_exports = True
    o.output_directory = outdir
    tide = Constant(0.0)
    open_marker = sorted(mesh2d.boundary_markers)[-1]
    solver_obj.bnd_functions['shallow_water'] = {open_marker: {'elev': tide}}
    solver_obj.assign_initial_conditions(elev=Constant(0.0), tracer=Function(P1_2d).assign(1.0 + (y > y.mean())))
    it = solver_obj.create_iterator(update_forcings=lambda t: tide.assign(0.3*math.sin(2*math.pi*t/600.0)))
    for t in it:                                   # stage by stage up to the first export ...
        if solver_obj.i_export >= 1:
            break
    o.simulation_end_time = 80.0                   # ... then the rest in batches (no forcing updates)
    solver_obj.export_initial_state = False
    solver_obj.iterate()
    return solver_obj


CASES = {
    'coast': _coast,
    'periodic': _periodic,
    'fields': _fields,
    'restart': _restart,
    'channel': _channel,
    'channel_wide': lambda outdir, **kw: _channel(outdir, nx=84, ny=16, export=False, **kw),
    # BASELINE cfg 2 / cfg 3's mesh: 1 M triangles (the examples/channel2d.py physics on the bench mesh's cell count)
    'channel_1m': lambda outdir, **kw: _channel(outdir, nx=1000, ny=500, export=False, **kw),
    'forced': _forced,
    'forced_fe': lambda outdir, **kw: _forced(outdir, stepper='ForwardEuler', **kw),
    'tracer': _tracer,
    'tracer_forced': lambda outdir, **kw: _tracer(outdir, forced=True, **kw),
    'tracer_nolim': lambda outdir, **kw: _tracer(outdir, limiter=False, **kw),
    'tracer_fe': lambda outdir, **kw: _tracer(outdir, stepper='ForwardEuler', **kw),
    'tracer_only': _tracer_only,
    'balzano': _balzano,
}


def _file_digests(outdir):
    out = {}
    for root, _, files in os.walk(outdir):
        for f in sorted(files):
            if f.endswith(('.vtu', '.pvd', '.npz')):
                with open(os.path.join(root, f), 'rb') as fh:
                    out[os.path.relpath(os.path.join(root, f), outdir)] = hashlib.blake2b(fh.read(), digest_size=16).hexdigest()
    return out


def run(name, outdir, **kw):
    solver_obj = CASES[name](outdir, **kw)
    res = {'iteration': solver_obj.iteration, 'simulation_time': solver_obj.simulation_time, 'i_export': solver_obj.i_export,
           'dt': solver_obj.dt,
           'uv': solver_obj.fields.uv_2d.dat.data_ro.copy(), 'elev': solver_obj.fields.elev_2d.dat.data_ro.copy()}
    for label in solver_obj.options.tracer:
        res[label] = solver_obj.fields[label].dat.data_ro.copy()
    hist = {}
    for mode in ('export', 'timestep'):
        for cname, c in solver_obj.callbacks[mode].items():
            hist['{:}/{:}'.format(mode, cname)] = np.array(c.history, dtype=np.float64)
    res['callbacks'] = hist
    res['forcing_times'] = np.array(getattr(solver_obj, '_forcing_times', []), dtype=np.float64)
    solver_obj.comm.barrier()                   # rank 0 has finished writing
    res['files'] = _file_digests(outdir)
    dev = solver_obj.timestepper.device if hasattr(solver_obj.timestepper, 'device') else None
    res['exchange'] = getattr(dev, 'exchange', None)
    res['verify_report'] = getattr(getattr(dev, 'dist', None), 'verify_report', None)
    return res
