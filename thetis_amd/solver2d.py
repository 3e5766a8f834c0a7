"""
``FlowSolver2d``: the user-facing driver with the reference's surface (thetis/solver2d.py) - options, boundary
functions, ``assign_initial_conditions`` and the ``iterate`` / ``create_iterator`` time loop - whose time stepper is
the MI355X-resident SSPRK33 or ForwardEuler (thetis_amd/rungekutta.py -> C ABI -> HIP stage kernels), coupled with the
2D tracers and the vertex limiter when ``options.add_tracer_2d`` is used.

Kept from the reference: option names and defaults, generator semantics of ``create_iterator`` (yields the time
*before* it is incremented, solver2d.py:1122-1127), export cadence with ``t_epsilon = 1e-5`` (:1036),
``simulation_time = t0 + k*dt`` (:1127), ``print_state`` line format (:931-970), the ``steppers`` table (:662-672).
Exports are VTK (.vtu/.pvd) plus .npz checkpoints that ``load_state`` reads back (thetis_amd/exporter.py; h5py is not
available here).  Not kept: log files, the implicit steppers and the 3D/NH/sediment couplings.
"""
import math
import sys
from collections import OrderedDict
import time as time_mod

import numpy as np

import os

from . import callback, coupled_timeintegrator_2d, exporter
from .comm import get_comm
from .function import Function, FunctionSpace, MixedFunction, get_functionspace
from .log import print_output
from .options import Constant, ModelOptions2d
from .rungekutta import SSPRK33, ForwardEuler
from .shallowwater_eq import DepthExpression, ShallowWaterEquations, g_grav
from .limiter import VertexBasedP1DGLimiter
from .tracer_eq_2d import TracerEquation2D

__all__ = ['FlowSolver2d']


class AttrDict(dict):
    """Dictionary whose keys are attributes too (thetis/utility.py:87-100)."""

    def __init__(self, *args, **kwargs):
        super(AttrDict, self).__init__(*args, **kwargs)
        self.__dict__ = self


class FieldDict(AttrDict):
    """AttrDict that checks the values are fields (thetis/utility.py:103-132)."""


class FlowSolver2d(object):
    def __init__(self, mesh2d, bathymetry_2d, options=None, keep_log=False):
        self._initialized = False
        self.mesh2d = mesh2d
        self.dt = None
        self.options = ModelOptions2d()
        if options is not None:
            self.options.update(options)
        self.simulation_time = 0
        self.iteration = 0
        self.i_export = 0
        self.next_export_t = self.simulation_time + self.options.simulation_export_time
        self.callbacks = callback.CallbackManager()
        self.fields = FieldDict()
        self.function_spaces = AttrDict()
        self.fields.bathymetry_2d = bathymetry_2d
        self.export_initial_state = True
        self.bnd_functions = {'shallow_water': {}, 'tracer': {}, 'sediment': {}}
        self.solve_tracer = False
        self.keep_log = keep_log
        # one process per GPU (python -m torch.distributed.run --nproc-per-node N script.py, the counterpart of the reference's
        # mpiexec -n N): the communicator of this run; COLLECTIVE when WORLD_SIZE > 1 (thetis_amd/comm.py)
        self.comm = get_comm()
        self.device_id = self.comm.local_rank if self.comm.size > 1 else 0
        # (test seam: ``FlowSolver2d._device_cls``, if set, replaces ``Swe2dDevice`` as the class of the per-rank handle - the CPU
        #  tests of the rank-parallel host logic hand in a stand-in there, tests/cpu_device.py; the product never sets it)

    # ------------------------------------------------------------------ time step
    def compute_time_step(self, u_scale=0.0):
        """Maximum explicit time step from the CFL condition (solver2d.py:149-177): the L2 projection onto CG-P1 of
        ``h_elem_size / (sqrt(g max(h, 0.05)) + u_scale)`` with the consistent CG mass matrix, as ``solve(a == l)`` does."""
        from .cgproject import project_to_p1
        mesh = self.mesh2d
        csize = self.fields.h_elem_size_2d
        # per-cell nodal values: the bathymetry may be CG-P1 (per vertex) or a continuous field held in DG-P1 (per cell node)
        bath = np.maximum(self.fields.bathymetry_2d.cell_node_values(), 0.05)
        u = np.sqrt(float(g_grav)*bath) + float(u_scale)                   # (N, k) P1 nodal
        cs = csize.cell_node_values()
        # integrand csize/u at the quadrature points (project_to_p1 integrates over all cells, in mesh order)
        sol = project_to_p1(mesh, lambda lam, cells: (cs @ lam)/(u @ lam))
        out = Function(self.function_spaces.P1_2d)
        out.assign(sol)
        return out

    def set_time_step(self, alpha=0.05):
        """solver2d.py:213-248"""
        automatic_timestep = False
        for o in (self.options.swe_timestepper_options, self.options.tracer_timestepper_options):
            if hasattr(o, 'use_automatic_timestep') and o.use_automatic_timestep:
                automatic_timestep = True
        if automatic_timestep:
            mesh2d_dt = self.compute_time_step(u_scale=float(self.options.horizontal_velocity_scale))
            dt = self.options.cfl_2d*alpha*float(mesh2d_dt.dat.data_ro.min())
            # comm.allreduce(dt, op=MIN) (solver2d.py:240): the replicated meshes give every rank the same number; the reduction
            # makes sure of it (ranks stepping with different dt would never meet again)
            self.dt = float(self.comm.allreduce_min(dt)[0])
            if not self.dt > 0.0:
                # the consistent-mass L2 projections of solver2d.py:149-177 / utility.py:620-640 undershoot where the cell size
                # changes by an order of magnitude between neighbours; the reference would step with that number
                raise ValueError('the automatic time step is not positive ({:g}): the CG-P1 projection of cell size / wave speed '
                                 'undershoots on this strongly graded mesh - set options.timestep'.format(self.dt))
            if self.options.use_wetting_and_drying:
                # the explicit wetting-drying formulation carries waves of speed sqrt(g |H|) through dry ground
                # (|H| up to 2.4 alpha) and was measured stable up to ~0.4-0.5 of this step on the reference's Thacker and
                # Balzano set-ups (tests/test_wetting_drying.py)
                factor = float(getattr(self.options, 'wetting_and_drying_cfl_factor', 0.4))
                self.dt *= factor
                print_output('explicit wetting-drying: automatic dt scaled by wetting_and_drying_cfl_factor = {:g}'.format(factor))
        else:
            assert self.options.timestep is not None
            assert self.options.timestep > 0.0
            self.dt = self.options.timestep
        print_output('dt = {:}'.format(self.dt))

    def set_wetting_and_drying_alpha(self):
        """Wetting-drying parameter alpha ~ |L_x grad h| (solver2d.py:251-303, Kaernae et al. 2011): per cell the
        bounding widths (utility.py:729-738) dotted with |grad h|, clipped to [alpha_min, alpha_max], then brought to P1.
        Firedrake ``interpolate``s the cell-wise expression into P1 [FD-assumed: which adjacent cell a vertex takes its
        value from is unspecified]; here a vertex takes the maximum over its adjacent cells."""
        o = self.options
        if not o.use_wetting_and_drying:
            return
        if o.use_automatic_wetting_and_drying_alpha:
            m = self.mesh2d
            p = m.cell_xy()
            h = self.fields.bathymetry_2d.dat.data_ro[m.cells]
            widths = np.abs(p - np.roll(p, 1, axis=1)).max(axis=1)                      # (N, 2): max |dx|, max |dy|
            # grad h of the P1/Q1 bathymetry, from a least-squares plane through the cell's vertices
            d = p - p.mean(axis=1, keepdims=True)
            g = np.einsum('nij,nj->ni', np.linalg.pinv(d), h - h.mean(axis=1, keepdims=True))
            alpha_c = (widths*np.abs(g)).sum(axis=1)
            amax, amin = o.wetting_and_drying_alpha_max, o.wetting_and_drying_alpha_min
            if amax is not None:
                alpha_c = np.minimum(float(amax), alpha_c)
            if amin is not None:
                alpha_c = np.maximum(float(amin), alpha_c)
            av = np.zeros(m.num_vertices)
            for i in range(m.cells.shape[1]):
                np.maximum.at(av, m.cells[:, i], alpha_c)
            if not hasattr(self.function_spaces, 'P1_2d'):
                self.create_function_spaces()
            o.wetting_and_drying_alpha = Function(self.function_spaces.P1_2d).assign(av)
        alpha = o.wetting_and_drying_alpha
        if isinstance(alpha, Function):
            a = alpha.dat.data_ro
            assert a.min() >= 0.0
            print_output('Using spatially varying wetting and drying parameter (min {:.2f} max {:.2f})'.format(a.min(), a.max()))
        else:
            assert float(alpha) >= 0.0
            print_output('Using constant wetting and drying parameter (value {:.2f})'.format(float(alpha)))

    # ------------------------------------------------------------------ setup
    def create_function_spaces(self):
        """solver2d.py:307-352, 'dg-dg' branch"""
        m = self.mesh2d
        fs = self.function_spaces
        fs.P0_2d = get_functionspace(m, 'DG', 0, name='P0_2d')
        fs.P1_2d = get_functionspace(m, 'CG', 1, name='P1_2d')
        fs.P1v_2d = get_functionspace(m, 'CG', 1, name='P1v_2d', vector=True)
        fs.P1DG_2d = get_functionspace(m, 'DG', 1, name='P1DG_2d')
        fs.P1DGv_2d = get_functionspace(m, 'DG', 1, name='P1DGv_2d', vector=True)
        if self.options.element_family != 'dg-dg':
            raise NotImplementedError("only element_family='dg-dg' runs on the device path")
        fs.U_2d = get_functionspace(m, 'DG', self.options.polynomial_degree, name='U_2d', vector=True)
        fs.H_2d = get_functionspace(m, 'DG', self.options.polynomial_degree, name='H_2d')
        fs.V_2d = (fs.U_2d, fs.H_2d)
        fs.Q_2d = get_functionspace(m, 'DG', 1, name='Q_2d')

    def create_fields(self):
        """solver2d.py:389-449"""
        if not hasattr(self.function_spaces, 'U_2d'):
            self.create_function_spaces()
        uv_2d = Function(self.function_spaces.U_2d, name='uv_2d')
        elev_2d = Function(self.function_spaces.H_2d, name='elev_2d')
        self.fields.solution_2d = MixedFunction((uv_2d, elev_2d), name='solution_2d')
        self.fields.uv_2d = uv_2d
        self.fields.elev_2d = elev_2d
        for label, topts in self.options.tracer.items():
            self.fields[label] = topts.function if topts.function is not None else Function(self.function_spaces.Q_2d, name=label)
        # mesh element size: CG-P1 L2 projection of sqrt(cell area), utility.py:620-640 (needed for automatic dt only)
        self.fields.h_elem_size_2d = None
        self.set_wetting_and_drying_alpha()
        # H = h + eta (+ the wetting-drying displacement): the options that shape it, by the names DepthExpression takes them under
        # (utility.py:975-996, solver2d.py:142-146)
        depth_options = {name: getattr(self.options, name)
                         for name in ('use_nonlinear_equations', 'use_wetting_and_drying', 'wetting_and_drying_alpha')}
        self.depth = DepthExpression(self.fields.bathymetry_2d, **depth_options)

    def create_equations(self):
        """solver2d.py:453-539"""
        if not hasattr(self.fields, 'uv_2d'):
            self.create_fields()
        self.equations = AttrDict()
        self.equations.sw = ShallowWaterEquations(self.function_spaces.H_2d, self.depth, self.options)
        self.equations.sw.bnd_functions = self.bnd_functions['shallow_water']
        self.solve_tracer = len(self.options.tracer) > 0
        for label in self.options.tracer:
            self.equations[label] = TracerEquation2D(label, self.function_spaces.Q_2d, self.depth, self.options)
        if self.solve_tracer and self.options.use_limiter_for_tracers and self.options.polynomial_degree > 0:
            self.tracer_limiter = VertexBasedP1DGLimiter(self.function_spaces.Q_2d, device_id=self.device_id)
        else:
            self.tracer_limiter = None

    def get_swe_timestepper(self, integrator):
        """Gets shallow water timestepper object with appropriate parameters (solver2d.py:542-573)"""
        o = self.options
        fields = {
            'linear_drag_coefficient': o.linear_drag_coefficient,
            'quadratic_drag_coefficient': o.quadratic_drag_coefficient,
            'manning_drag_coefficient': o.manning_drag_coefficient,
            'nikuradse_bed_roughness': o.nikuradse_bed_roughness,
            'viscosity_h': o.horizontal_viscosity,
            'lax_friedrichs_velocity_scaling_factor': o.lax_friedrichs_velocity_scaling_factor,
            'coriolis': o.coriolis_frequency,
            'wind_stress': o.wind_stress,
            'atmospheric_pressure': o.atmospheric_pressure,
            'momentum_source': o.momentum_source_2d,
            'volume_source': o.volume_source_2d,
        }
        bnd_conditions = self.bnd_functions['shallow_water']
        return integrator(self.equations.sw, self.fields.solution_2d, fields, self.dt,
                          o.swe_timestepper_options, bnd_conditions, device_id=self.device_id, comm=self.comm,
                          spmd=self._partition_requirements(), device_cls=getattr(self, '_device_cls', None))

    def _partition_requirements(self):
        """What a partitioned run must know before it cuts the mesh (thetis_amd/spmd.py): the ghost layers depend on the number
        of tracers, on the limiter (vertex neighbours) and on the stages per step."""
        o = self.options
        stepper = o.tracer_timestepper_type if (o.tracer_only and o.tracer) else o.swe_timestepper_type
        return {'n_tracers': len(o.tracer), 'tracer_only': bool(o.tracer_only and o.tracer), 'stepper': stepper,
                'use_limiter': bool(o.tracer and o.use_limiter_for_tracers and o.polynomial_degree > 0)}

    def get_tracer_timestepper(self, integrator, system, swe_stepper):
        """Gets tracer timestepper object with appropriate parameters (solver2d.py:576-598)"""
        # the coefficient dict a tracer's stepper receives: the keys are the reference's by contract (solver2d.py:580-589), the
        # equation terms look them up by name
        o = self.options
        velocity, elevation = self.fields.solution_2d.subfunctions
        fields = dict(uv_2d=velocity, elev_2d=elevation)
        for key in ('lax_friedrichs_tracer_scaling_factor', 'tracer_advective_velocity_factor'):
            fields[key] = getattr(o, key)
        for label in system.split(','):
            tracer_options = o.tracer[label]
            fields['diffusivity_h-' + label] = tracer_options.diffusivity
            fields['source-' + label] = tracer_options.source
        bcs = {}
        if system in self.bnd_functions:
            bcs = self.bnd_functions[system]
        elif system[:-3] in self.bnd_functions:
            bcs = self.bnd_functions[system[:-3]]
        return integrator(self.equations[system], self.fields[system], fields, self.dt,
                          self.options.tracer_timestepper_options, bcs, swe_stepper)

    def create_timestepper(self):
        """solver2d.py:651-700"""
        if not hasattr(self, 'equations'):
            self.create_equations()
        if any(hasattr(o, 'use_automatic_timestep') and o.use_automatic_timestep
               for o in (self.options.swe_timestepper_options, self.options.tracer_timestepper_options)):
            from .cgproject import elem_size_p1
            self.fields.h_elem_size_2d = Function(self.function_spaces.P1_2d).assign(elem_size_p1(self.mesh2d))
        self.compute_mesh_stats()
        self.set_time_step()
        steppers = {'SSPRK33': SSPRK33, 'ForwardEuler': ForwardEuler}           # the explicit entries of solver2d.py:662-672
        tracer_steppers = {'SSPRK33': coupled_timeintegrator_2d.DeviceTracerSSPRK33,
                           'ForwardEuler': coupled_timeintegrator_2d.DeviceTracerForwardEuler}
        name = self.options.swe_timestepper_type
        if self.options.tracer_only and self.options.tracer:
            name = 'SSPRK33'        # the shallow water state is frozen (coupled_timeintegrator_2d.py:98): the stepper
            #                         object only holds the device-resident velocity, its type option is not used
        if name not in steppers:
            raise NotImplementedError("swe_timestepper_type {!r} needs a global (non)linear solve and is outside the "
                                      "explicit device path; use 'SSPRK33' or 'ForwardEuler'".format(name))
        if self.solve_tracer:
            if self.options.tracer_timestepper_type not in tracer_steppers:
                raise NotImplementedError("tracer_timestepper_type {!r} is outside the explicit device path; use "
                                          "'SSPRK33' or 'ForwardEuler'".format(self.options.tracer_timestepper_type))
            if self.comm.size > 1 and not self.options.tracer_only and self.options.tracer_timestepper_type != name:
                raise NotImplementedError('partitioned runs step the shallow water equations and the tracers with the same scheme '
                                          '(the ghost layers are counted in stages per step)')
            swe = self.get_swe_timestepper(steppers[name])
            tracers = {}
            for system in self.options.tracer_fields:
                tracers[system] = self.get_tracer_timestepper(tracer_steppers[self.options.tracer_timestepper_type], system, swe)
            self.timestepper = coupled_timeintegrator_2d.GeneralCoupledTimeIntegrator2D(self, swe, tracers)
        else:
            self.timestepper = self.get_swe_timestepper(steppers[name])
        print_output('Using time integrator: {:}'.format(self.timestepper.__class__.__name__))

    def compute_mesh_stats(self):
        """solver2d.py:179-211"""
        m = self.mesh2d
        print_output('Element family: {:}, degree: {:}'.format(self.options.element_family, self.options.polynomial_degree))
        print_output('2D cell type: {:}'.format('triangle' if m.cells.shape[1] == 3 else 'quadrilateral'))
        print_output('2D mesh: {:} vertices, {:} elements'.format(m.num_vertices, m.num_cells))
        a = np.sqrt(m.cell_areas())
        # comm.allreduce MIN / MAX (solver2d.py:192-193)
        lo, hi = float(self.comm.allreduce_min(a.min())[0]), float(self.comm.allreduce_max(a.max())[0])
        print_output('Horizontal element size: {:.2f} ... {:.2f} m'.format(lo, hi))
        if self.comm.size > 1:
            print_output('Partitioned over {:d} ranks (one per GPU)'.format(self.comm.size))
        print_output('Number of 2D elevation DOFs: {:}'.format(self.function_spaces.H_2d.dim()))
        print_output('Number of 2D velocity DOFs: {:}'.format(self.function_spaces.U_2d.dim()))

    def _field_metadata(self):
        meta = dict(exporter.field_metadata)
        for label, topts in self.options.tracer.items():
            meta[label] = topts.metadata
        return meta

    def create_exporters(self):
        """Creates file exporters (solver2d.py:704-730): VTK for fields_to_export, checkpoints for fields_to_export_hdf5."""
        self.exporters = OrderedDict()
        if self.options.no_exports:
            return
        o = self.options
        if o.fields_to_export:
            self.exporters['vtk'] = exporter.ExportManager(o.output_directory, o.fields_to_export, self.fields,
                                                           self._field_metadata(), export_type='vtk', comm=self.comm)
        if o.fields_to_export_hdf5:
            self.exporters['hdf5'] = exporter.ExportManager(os.path.join(o.output_directory, 'hdf5'), o.fields_to_export_hdf5,
                                                            self.fields, self._field_metadata(), export_type='hdf5', comm=self.comm)

    def initialize(self):
        """Whatever of spaces, equations, stepper and exporters a script has not created itself, in the order they depend on each
        other (solver2d.py:732-744: a script may call any create_* first, e.g. to add an equation before the stepper exists)."""
        for holder, attribute, create in ((self.function_spaces, 'U_2d', self.create_function_spaces),
                                          (self, 'equations', self.create_equations),
                                          (self, 'timestepper', self.create_timestepper),
                                          (self, 'exporters', self.create_exporters)):
            if not hasattr(holder, attribute):
                create()
        self._initialized = True

    def assign_initial_conditions(self, elev=None, uv=None, **tracers):
        """Assigns initial conditions by L2 projection (solver2d.py:747-785)"""
        if not self._initialized:
            self.initialize()
        velocity, elevation = self.fields.solution_2d.subfunctions
        for target, expression in ((elevation, elev), (velocity, uv)):
            if expression is not None:
                target.project(expression)
        for name, expression in tracers.items():
            # keyword 'salt' and keyword 'salt_2d' both name the tracer field salt_2d (solver2d.py:773-776)
            label = name if name.endswith('_2d') and len(name) > 3 else name + '_2d'
            if label not in self.options.tracer:
                raise AssertionError('Unknown tracer label {:}'.format(label))
            self.fields[label].project(expression)
        # (sediment: out of scope, options.py raises where it is selected)
        self.timestepper.initialize(self.fields.solution_2d)

    def add_callback(self, callback, eval_interval='export'):
        """solver2d.py:788-797"""
        self.callbacks.add(callback, eval_interval)

    def export(self, time=None):
        """Export all fields to disk and evaluate the export callbacks (solver2d.py:799-812)."""
        self.callbacks.evaluate(mode='export', index=self.i_export)
        for e in self.exporters.values():
            e.export(time=time)

    def load_state(self, i_stored, outputdir=None, t=None, iteration=None, i_export=None):
        """Loads simulation state from the checkpoint files of an earlier run and restores the export / iteration
        bookkeeping (solver2d.py:820-921).  Replaces :meth:`assign_initial_conditions`."""
        if not self._initialized:
            self.initialize()
        if outputdir is None:
            outputdir = self.options.output_directory
        self.comm.barrier()         # the files are written by rank 0 (of this run, or of the run that is being restarted)
        e = exporter.ExportManager(os.path.join(outputdir, 'hdf5'), ['uv_2d', 'elev_2d'], self.fields,
                                   self._field_metadata(), export_type='hdf5', comm=self.comm)
        metadata = {}
        metadata.update(e.exporters['uv_2d'].load(i_stored, self.fields.uv_2d))
        metadata.update(e.exporters['elev_2d'].load(i_stored, self.fields.elev_2d))
        self.assign_initial_conditions()
        # where the restarted run stands (solver2d.py:886-903): export index = the stored one unless told otherwise; the iteration
        # count and the time follow from it (the checkpoint's own time stamp wins over the estimate), the NEXT export is one interval on
        interval = self.options.simulation_export_time
        self.i_export = i_stored if i_export is None else i_export
        t_export = self.i_export*interval
        self.iteration = int(np.ceil(t_export/self.dt)) if iteration is None else iteration
        self.simulation_time = metadata.get('time', self.iteration*self.dt) if t is None else t
        self.next_export_t = t_export + interval
        # a restart that writes into a NEW directory exports its initial state and numbers its exports from i_export; a
        # continuation in the same directory does neither (solver2d.py:905-912)
        self.export_initial_state = outputdir != self.options.output_directory
        offset = 0 if self.export_initial_state else 1
        for ex in self.exporters.values():
            ex.set_next_export_ix(self.i_export + offset)

    # ------------------------------------------------------------------ time loop
    def print_state(self, cputime, print_header=False):
        """Print a summary of the model state on stdout (solver2d.py:923-971)"""
        entries = [('exp', self.i_export, '5d'), ('iter', self.iteration, '5d')]
        time_str = '{:.2f}'.format(self.simulation_time).rjust(15)
        entries += [('time', time_str, '15s')]
        if self.options.tracer_only:
            area = self.mesh2d.cell_areas()
            for label in self.options.tracer:
                q = self.fields[label].cell_node_values()
                if q.shape[1] == 3:
                    norm_q = math.sqrt(float(np.sum(area/12.0*(q.sum(axis=1)**2 + (q**2).sum(axis=1)))))
                else:
                    kq = 4*q + 2*np.roll(q, -1, axis=1) + 2*np.roll(q, 1, axis=1) + np.roll(q, 2, axis=1)
                    norm_q = math.sqrt(float(np.sum(area/36.0*(q*kq).sum(axis=1))))
                entries.append((label, norm_q, '10.4f'))
        else:
            d = self.timestepper.diagnostics()
            norm_h = math.sqrt(d[0])
            norm_u = math.sqrt(d[1])
            entries += [('eta norm', norm_h, '14.4f'), ('u norm', norm_u, '14.4f')]
        entries.append(('Tcpu', cputime, '6.2f'))
        if print_header:
            header = ' '.join([e[0].rjust(len('{:{fmt}}'.format(e[1], fmt=e[2]))) for e in entries])
            print_output(header)
        line = ' '.join(['{:{fmt}}'.format(e[1], fmt=e[2]) for e in entries])
        print_output(line)
        sys.stdout.flush()

    def iterate(self, update_forcings=None, export_func=None):
        """Runs the simulation (solver2d.py:974-994)"""
        for _ in self.create_iterator(update_forcings=update_forcings, export_func=export_func, _batch=True):
            pass

    def create_iterator(self, update_forcings=None, export_func=None, _batch=False):
        """Generator over the time loop (solver2d.py:997-1144).  ``_batch`` (used by :meth:`iterate`, which discards the
        yielded times): without forcing updates and per-time-step callbacks, all steps up to the next export are issued in
        one call into the library and the generator yields once per batch; times, iteration counts and export instants are
        exactly those of the step-by-step loop."""
        if not self._initialized:
            self.initialize()
        o = self.options
        assert o.simulation_end_time is not None, 'simulation_end_time must be set'
        eps = 1.0e-5                                    # the reference's t_epsilon (solver2d.py:1036)
        self._register_requested_checks()
        t_start, n_done = self.simulation_time, 0
        export_due = t_start + o.simulation_export_time
        wall = time_mod.perf_counter()
        self.print_state(0.0, print_header=True)
        if self.export_initial_state:
            self._export_now(export_func)
        stepper = self.timestepper
        can_batch = (_batch and update_forcings is None and not self.callbacks['timestep'] and hasattr(stepper, 'advance_steps'))

        def steps_to_next_event():
            """steps until the loop below would export or stop, by the loop's own arithmetic: t_k = t_start + k*dt"""
            n = 1
            while True:
                t_k = t_start + (n_done + n)*self.dt
                if t_k >= export_due - eps or not t_k <= o.simulation_end_time - eps:
                    return n
                n += 1

        while self.simulation_time <= o.simulation_end_time - eps:
            if can_batch:
                n = steps_to_next_event()
                stepper.advance_steps(self.simulation_time, n)
            else:
                n = 1
                stepper.advance(self.simulation_time, update_forcings)
            yield self.simulation_time                  # the time the step STARTED from, as the reference's generator does
            n_done += n
            self.iteration += n
            self.simulation_time = t_start + n_done*self.dt          # k*dt, never an accumulated sum (solver2d.py:1127)
            self.callbacks.evaluate(mode='timestep')
            if self.simulation_time >= export_due - eps:
                self.i_export += 1
                export_due += o.simulation_export_time
                now = time_mod.perf_counter()
                self.print_state(now - wall)
                wall = now
                self._export_now(export_func)
        return self.simulation_time

    def _export_now(self, export_func=None):
        self.export(time=self.simulation_time)
        if export_func is not None:
            export_func()

    def _register_requested_checks(self):
        """the conservation / overshoot checks the options ask for (solver2d.py:1040-1086), evaluated at every export"""
        o = self.options
        checks = []
        if o.check_volume_conservation_2d:
            checks.append(callback.VolumeConservation2DCallback(self))
        for label, tracer in o.tracer.items():
            if o.check_tracer_conservation:
                make = (callback.ConservativeTracerMassConservation2DCallback if tracer.use_conservative_form
                        else callback.TracerMassConservation2DCallback)
                checks.append(make(label, self))
            if o.check_tracer_overshoot:
                checks.append(callback.TracerOvershootCallBack(label, self))
        for c in checks:
            self.add_callback(c, eval_interval='export')
