"""
``Swe2dDevice``: one HIP device stepping one (partition of a) mesh - a thin object over the C ABI.

Arrays cross the boundary in the reference's dof layout (cell c owns nodes 3c..3c+2): ``uv`` (N,3,2),
``eta`` (N,3) float64.
"""
import ctypes
import os
import numpy as np

from . import _lib, ordering

__all__ = ['Swe2dDevice']

_dp = ctypes.POINTER(ctypes.c_double)


def _ptr(a):
    return a.ctypes.data_as(_dp)


def _iptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


class FacetValues(object):
    """Function-valued boundary data of one marker in compact form: the values at the two end nodes of every boundary
    facet, in the order of ``Swe2dDevice.boundary_facets`` - (n_facets, 2) or (n_facets, 2, 2) for a velocity."""

    def __init__(self, values):
        self.values = np.asarray(values, dtype=np.float64)


class Swe2dDevice(object):
    def __init__(self, mesh, bathymetry_vertex, dt, g_grav=9.81, use_nonlinear_equations=True,
                 use_lax_friedrichs_velocity=True, lax_friedrichs_velocity_scaling_factor=1.0,
                 device_id=0, n_owned=None, boundary_len=None, reorder='auto', ranges=None):
        """
        :arg mesh: object with ``cells`` (N,3), ``vertex_xy`` (V,2), ``cell_nbr`` (N,3), ``cell_nbr_facet`` (N,3)
        :arg bathymetry_vertex: (V,) CG-P1 bathymetry at the vertices
        :kwarg reorder: None | 'auto' | 'hilbert' | explicit cell permutation: device-side cell numbering (callers never see it)
        :kwarg ranges: cell ranges that a reordering must not mix, e.g. (n_interior, n_owned) of a partition
        """
        self.lib = _lib.load()
        self.n_cells = int(mesh.cells.shape[0])
        self.npc = int(mesh.cells.shape[1])
        self.n_owned = self.n_cells if n_owned is None else int(n_owned)
        c = np.ascontiguousarray
        cells0 = np.asarray(mesh.cells)
        xy0 = np.asarray(mesh.vertex_xy, dtype=np.float64)
        nbr0 = np.asarray(mesh.cell_nbr)
        nbf0 = np.asarray(mesh.cell_nbr_facet)
        bath0 = np.asarray(bathymetry_vertex, dtype=np.float64)
        if bath0.shape != (xy0.shape[0],):
            raise ValueError('bathymetry must have one value per vertex')
        # boundary markers are arbitrary positive ids in mesh files (e.g. 100, 200 in demos/north_sea.msh); the C ABI
        # indexes a small table, so they are mapped to slots 1..15 here
        markers = sorted(int(m) for m in np.unique(-nbr0[nbr0 < 0])) if (nbr0 < 0).any() else []
        if len(markers) >= _lib.MAX_MARKERS:
            raise NotImplementedError('more than {:d} distinct boundary markers'.format(_lib.MAX_MARKERS - 1))
        self._marker_slot = {m: i + 1 for i, m in enumerate(markers)}
        if markers and markers != list(range(1, len(markers) + 1)):
            lut = np.zeros(max(markers) + 1, dtype=np.int64)
            for m, sl in self._marker_slot.items():
                lut[m] = sl
            nbr0 = np.where(nbr0 < 0, -lut[np.where(nbr0 < 0, -nbr0, 0)], nbr0)
        self._caller_nbr = np.array(nbr0)        # (N, k), boundary facets = -slot, caller's cell numbering
        if boundary_len is not None:
            boundary_len = {self._marker_slot[m]: v for m, v in boundary_len.items() if m in self._marker_slot}
        # ---- device numbering: perm[i_dev] = i_caller
        self.perm = None
        self._vperm = None
        if reorder is not None:
            if isinstance(reorder, str):
                if reorder not in ('hilbert', 'auto'):
                    raise ValueError('unknown reorder {!r}'.format(reorder))
                cen = xy0[cells0].mean(axis=1)
                bounds = [0] + [int(b) for b in (ranges or (self.n_owned,))] + [self.n_cells]
                bounds = sorted(set(bounds))
                perm = np.arange(self.n_cells)
                for a, b in zip(bounds[:-1], bounds[1:]):
                    if b - a > 1 and a < self.n_owned:          # ghosts keep the order the halo messages deliver
                        if reorder == 'auto':
                            perm[a:b] = a + ordering.auto_cell_order(mesh, a, b)
                        else:
                            perm[a:b] = a + ordering.hilbert_cell_order(cen[a:b])
            else:
                perm = np.asarray(reorder, dtype=np.int64)
                assert sorted(perm) == list(range(self.n_cells))
            inv = np.empty_like(perm)
            inv[perm] = np.arange(self.n_cells)
            self.perm, self.inv_perm = perm, inv
            cells0 = cells0[perm]
            nb = nbr0[perm].astype(np.int64)
            pos = nb >= 0
            nb[pos] = inv[nb[pos]]
            nbr0 = nb
            nbf0 = nbf0[perm]
            vperm = ordering.first_touch_vertex_order(cells0)
            vinv = np.full(xy0.shape[0], -1, dtype=np.int64)
            vinv[vperm] = np.arange(len(vperm))
            cells0 = vinv[cells0]
            xy0 = xy0[vperm]
            bath0 = bath0[vperm]
            self._vperm = vperm
        self._topo_cells = None
        self._limiter_ready = False
        topo = getattr(mesh, 'topo_vertex', None)
        if topo is not None and not np.array_equal(np.asarray(topo), np.arange(len(topo))):
            tc = np.asarray(topo)[np.asarray(mesh.cells)]
            if self.perm is not None:
                tc = tc[self.perm]
            _, tc = np.unique(tc, return_inverse=True)                # compact ids
            self._topo_cells = tc.reshape(-1, self.npc)
        self._keep = [c(cells0, dtype=np.int32), c(xy0, dtype=np.float64),
                      c(nbr0, dtype=np.int32), c(nbf0, dtype=np.int8), c(bath0, dtype=np.float64)]
        cells, xy, nbr, nbf, bath = self._keep
        m = _lib.Swe2dMesh()
        m.n_cells = self.n_cells
        m.n_owned = self.n_owned
        m.n_vertices = xy.shape[0]
        m.nodes_per_cell = self.npc
        m.cell_vertices = cells.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        m.vertex_xy = _ptr(xy)
        m.cell_neighbours = nbr.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        m.cell_neighbour_facets = nbf.ctypes.data_as(ctypes.POINTER(ctypes.c_int8))
        m.bathymetry = _ptr(bath)
        if boundary_len is not None:
            bl = np.zeros(_lib.MAX_MARKERS)
            for k, v in boundary_len.items():
                if 0 < k < _lib.MAX_MARKERS:
                    bl[k] = v
            self._keep.append(bl)
            m.boundary_len = _ptr(bl)
        p = _lib.Swe2dParams()
        p.g_grav = g_grav
        p.dt = dt
        p.use_nonlinear_equations = int(bool(use_nonlinear_equations))
        p.use_lax_friedrichs_velocity = int(bool(use_lax_friedrichs_velocity))
        p.lax_friedrichs_velocity_scaling_factor = float(lax_friedrichs_velocity_scaling_factor)
        p.device_id = device_id
        self.h = ctypes.c_void_p()
        _lib.check(self.lib.swe2d_create(ctypes.byref(m), ctypes.byref(p), ctypes.byref(self.h)))
        # the library reads no environment: the THETIS_AMD_* switches of _lib.OPTION_ENV become options of this handle, once, here
        for opt, value in _lib.options_from_environment():
            self.set_option(opt, value)
        if (self.npc == 3 and isinstance(reorder, str) and self.n_owned == self.n_cells and 64 < self.n_cells <= 196608
                and os.environ.get('THETIS_AMD_FLOW_BLOCKS', '1') != '0'):
            # a mesh small enough for the dataflow kernel (swe2d_advance takes it by itself): its 64-cell blocks as compact tiles /
            # bisection boxes (ordering.flow_block_order) instead of 64 consecutive cells of the device numbering - a fifth to a
            # third fewer rim facets
            self.flow_set_order(ordering.flow_block_order(mesh))
        tt = os.environ.get('THETIS_AMD_TRIPLE_TILE', '11,8')
        if (self.npc == 3 and isinstance(reorder, str) and self.n_owned == self.n_cells and getattr(mesh, 'structured', False)
                and self.n_cells > 131072 and tt != '0'):
            # a mesh beyond the dataflow kernel: all three stages of a step in one launch (csrc/swe2d_fuse.h, swe_fuse123_kernel) on
            # two-ring tiles cut as patches of 11 x 8 quads - 176 triangles + rings of 38 + 42 = the 256 lanes - instead of as many
            # consecutive cells of the 16 x 6 numbering as fit (147 + 52 + 57, ragged: 1 M cells 110.7 -> 96.8 us per step, where the
            # fused pair takes 103.4; profiles/r06l_triple_tiles.txt).  THETIS_AMD_TRIPLE_TILE = "bx,by" | 0: A/B runs
            bx, by = (int(v) for v in tt.split(','))
            self.fused_set_triple_tiles(*ordering.triple_tile_order(mesh, bx, by))
        if self.npc == 4 and not getattr(mesh, 'affine', True):
            # a partition (LocalPartition.affine = the GLOBAL mesh's flag) whose own cells happen to be parallelograms takes the
            # general kernels like every other rank: ghost and owned copies of a cell then agree bit for bit
            self._ck(self.lib.swe2d_set_general_quadrilaterals(self.h, 1))

    # -- lifetime
    def close(self):
        if getattr(self, 'h', None) is not None and self.h:
            self.lib.swe2d_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        _lib.check(rc, self.h)

    # -- state
    def set_state(self, uv, eta):
        uv = np.asarray(uv, dtype=np.float64).reshape(self.n_cells, self.npc, 2)
        eta = np.asarray(eta, dtype=np.float64).reshape(self.n_cells, self.npc)
        if self.perm is not None:
            uv, eta = uv[self.perm], eta[self.perm]
        uv, eta = np.ascontiguousarray(uv), np.ascontiguousarray(eta)
        self._ck(self.lib.swe2d_set_state(self.h, _ptr(uv), _ptr(eta)))

    def get_state(self, i_stage=2):
        """The step result, or with ``i_stage`` = 0 / 1 the stage solution left by ``solve_stage(i_stage)``."""
        uv = np.empty((self.n_cells, self.npc, 2))
        eta = np.empty((self.n_cells, self.npc))
        self._ck(self.lib.swe2d_get_stage_state(self.h, int(i_stage), _ptr(uv), _ptr(eta)))
        if self.perm is not None:
            uv, eta = uv[self.inv_perm], eta[self.inv_perm]
        return uv, eta

    def snapshot(self, slot=0):
        """Save the time-stepping state (and every tracer) in a device-side copy; ``restore`` brings it back - exactly, also with
        wetting-drying, where ``set_state(*get_state())`` is the identity only up to rounding (the device carries D, not eta).
        ``slot``: one of _lib.SNAPSHOT_SLOTS independent copies (a long-lived one next to short-lived ones)."""
        self._ck(self.lib.swe2d_state_snapshot_slot(self.h, int(slot), 0))

    def restore(self, slot=0):
        self._ck(self.lib.swe2d_state_snapshot_slot(self.h, int(slot), 1))

    def set_option(self, option, value):
        """include/swe2d.h swe2d_option (``_lib.OPT_*``); ``None`` or -1: the library's own rule."""
        self._ck(self.lib.swe2d_set_option(self.h, int(option), -1 if value is None else int(value)))

    def get_option(self, option):
        v = ctypes.c_int(0)
        self._ck(self.lib.swe2d_get_option(self.h, int(option), ctypes.byref(v)))
        return v.value

    def set_dt(self, dt):
        self._ck(self.lib.swe2d_set_dt(self.h, float(dt)))

    def _slot(self, marker):
        try:
            return self._marker_slot[int(marker)]
        except KeyError:
            raise KeyError('the mesh has no boundary with marker {:}'.format(marker))

    def set_bc(self, marker, funcs):
        """``funcs``: dict with constant 'elev' / 'uv' / 'un' / 'flux' values, or None / {} for a closed boundary."""
        marker = self._slot(marker)
        kind = 0
        vals = np.zeros(5)
        is_field = lambda v: isinstance(v, FacetValues) or (isinstance(v, np.ndarray) and v.ndim >= 2)
        for key, value in (funcs or {}).items():
            if key == 'elev':
                kind |= _lib.BC_ELEV
                if is_field(value):
                    kind |= _lib.BC_ELEV_FIELD
                    self._set_bc_function(0, marker, value)
                else:
                    vals[0] = float(value)
            elif key == 'uv':
                kind |= _lib.BC_UV
                if is_field(value):
                    kind |= _lib.BC_UV_FIELD
                    self._set_bc_function(1, marker, value)
                else:
                    vals[1], vals[2] = float(value[0]), float(value[1])
            elif key == 'un':
                kind |= _lib.BC_UN
                if is_field(value):
                    kind |= _lib.BC_UN_FIELD
                    self._set_bc_function(2, marker, value)
                else:
                    vals[3] = float(value)
            elif key == 'flux':
                kind |= _lib.BC_FLUX
                if is_field(value):
                    kind |= _lib.BC_FLUX_FIELD
                    self._set_bc_function(3, marker, value)
                else:
                    vals[4] = float(value)
            elif key == 'drag':
                pass            # handled below
            else:
                raise Exception('Invalid boundary tag "{:}" specified on boundary {:}'.format(key, marker))
        self._ck(self.lib.swe2d_set_bc(self.h, int(marker), kind, _ptr(vals)))
        drag = (funcs or {}).get('drag')
        self._ck(self.lib.swe2d_set_boundary_drag(self.h, int(marker), -1.0 if drag is None else float(drag)))

    def set_wetting_and_drying(self, alpha):
        """Enable the explicit wetting-drying formulation; ``alpha``: constant or per-vertex array; None disables."""
        if alpha is None:
            self._ck(self.lib.swe2d_set_wetting_and_drying(self.h, 0, None))
            return
        nv = self._keep[1].shape[0]                      # device vertices (first-touch numbering)
        a = np.asarray(alpha, dtype=np.float64)
        if a.ndim == 0:
            a = np.full(nv, float(a))
        elif self._vperm is not None:
            a = a[self._vperm]
        a = np.ascontiguousarray(a)
        assert a.shape == (nv,), 'alpha must be a constant or have one value per vertex'
        self._ck(self.lib.swe2d_set_wetting_and_drying(self.h, 1, _ptr(a)))

    def _vertex_coefficient(self, value):
        """constant -> (None, float); per-vertex array (mesh numbering) -> (device-ordered array, 0.0)"""
        a = np.asarray(value, dtype=np.float64)
        if a.ndim == 0:
            return None, float(a)
        nv = self._keep[1].shape[0]
        if self._vperm is not None:
            a = a[self._vperm]
        a = np.ascontiguousarray(a)
        assert a.shape == (nv,), 'coefficient must be a constant or have one value per vertex'
        return a, 0.0

    def set_viscosity(self, nu, sipg_factor=1.0, use_grad_div_viscosity_term=False, use_grad_depth_viscosity_term=True):
        """SIPG horizontal viscosity (shallowwater_eq.py:554-616); ``nu``: constant, per-vertex array, or None (off)."""
        if nu is None:
            self._ck(self.lib.swe2d_set_viscosity(self.h, 0, None, 0.0, 1.0, 0, 0))
            return
        arr, const = self._vertex_coefficient(nu)
        self._ck(self.lib.swe2d_set_viscosity(self.h, 1, None if arr is None else _ptr(arr), const, float(sipg_factor),
                                              int(bool(use_grad_div_viscosity_term)), int(bool(use_grad_depth_viscosity_term))))

    def tracer_set_conservative(self, tracer_id, use_conservative_form=True):
        """The tracer field is the depth-integrated q = H*T (tracer_eq_2d.py:325-437)."""
        self._ck(self.lib.swe2d_tracer_set_conservative(self.h, int(tracer_id), int(bool(use_conservative_form))))

    def tracer_set_diffusivity(self, tracer_id, mu, sipg_factor_tracer=1.0):
        """SIPG horizontal diffusion of a tracer (tracer_eq_2d.py:226-278); constant, per-vertex array or None (off)."""
        if mu is None:
            self._ck(self.lib.swe2d_tracer_set_diffusivity(self.h, int(tracer_id), 0, None, 0.0, 1.0))
            return
        arr, const = self._vertex_coefficient(mu)
        self._ck(self.lib.swe2d_tracer_set_diffusivity(self.h, int(tracer_id), 1, None if arr is None else _ptr(arr), const,
                                                       float(sipg_factor_tracer)))

    def tracer_set_diffusion_bc(self, tracer_id, marker, kind, diff_flux=0.0):
        """kind: 0 none, 1 prescribed 'diff_flux', 2 constant 'value', 3 boundary dict without 'value', 4 Function 'value'."""
        self._ck(self.lib.swe2d_tracer_set_diffusion_bc(self.h, int(tracer_id), self._slot(marker), int(kind), float(diff_flux)))

    def _set_bc_function(self, which, slot, value):
        if isinstance(value, FacetValues):
            self.set_bc_facets(which, slot, value.values)
        else:
            self.set_bc_field(which, slot, value)

    def facet_node_values(self, marker, function_values, cells_of_vertices=None):
        """Compact boundary data of ``marker`` from a P1 field: ``function_values`` are DG nodal values (N, k[, 2]) or, with
        ``cells_of_vertices`` = the (N, k) cell-vertex table, CG vertex values (V[, 2])."""
        cells, facets = self.boundary_facets(self._slot(marker))
        nxt = (facets + 1) % self.npc
        d = np.asarray(function_values)
        if cells_of_vertices is None:
            return FacetValues(np.stack([d[cells, facets], d[cells, nxt]], axis=1))
        cv = np.asarray(cells_of_vertices)
        return FacetValues(np.stack([d[cv[cells, facets]], d[cv[cells, nxt]]], axis=1))

    def boundary_facets(self, slot):
        """(cells, facets) of the boundary facets carrying marker slot ``slot``, in the CALLER's cell numbering (cached)."""
        cache = self.__dict__.setdefault('_bnd_facets', {})
        if slot not in cache:
            c, f = np.nonzero(self._caller_nbr == -int(slot))
            cache[slot] = (np.ascontiguousarray(c.astype(np.int32)), np.ascontiguousarray(f.astype(np.int32)))
        return cache[slot]

    def _device_cells(self, cells):
        """caller cell ids -> device cell ids (cached per list object: boundary_facets hands out the same arrays)"""
        cache = self.__dict__.setdefault('_device_cell_lists', {})
        key = id(cells)
        if key not in cache:
            cache[key] = (cells, np.ascontiguousarray((cells if self.perm is None else self.inv_perm[cells]).astype(np.int32)))
        return cache[key][1]

    def set_bc_facets(self, which, slot, values):
        """Function-valued boundary data of one marker slot in compact form: ``values`` (n_facets, 2) [(n_facets, 2, 2) for
        which = 1] = the value at the first and second node of every facet of ``boundary_facets(slot)``."""
        cells, facets = self.boundary_facets(slot)
        v = np.ascontiguousarray(np.asarray(values, dtype=np.float64).reshape((len(cells), 2, 2) if which == 1 else (len(cells), 2)))
        self._ck(self.lib.swe2d_set_bc_facets(self.h, int(which), len(cells), _iptr(self._device_cells(cells)), _iptr(facets),
                                              _ptr(v)))

    def tracer_set_bc_facets(self, tid, slot, values):
        """compact form of ``tracer_set_bc`` with a Function value: ``values`` (n_facets, k) = the external value at every
        node of the boundary cells of ``boundary_facets(slot)``."""
        cells, facets = self.boundary_facets(slot)
        v = np.ascontiguousarray(np.asarray(values, dtype=np.float64).reshape(len(cells), self.npc))
        self._ck(self.lib.swe2d_tracer_set_bc_facets(self.h, int(tid), len(cells), _iptr(self._device_cells(cells)),
                                                     _iptr(facets), _ptr(v)))
        self._ck(self.lib.swe2d_tracer_set_bc(self.h, int(tid), int(slot), 2, 0.0))

    def set_bc_field(self, which, slot, nodal):
        """Function-valued boundary data of one marker slot: nodal DG values (N,k) [(N,k,2) for which = 1]; which = 0 elev,
        1 uv, 2 un, 3 flux."""
        shape = (self.n_cells, self.npc, 2) if which == 1 else (self.n_cells, self.npc)
        a = np.asarray(nodal, dtype=np.float64).reshape(shape)
        if self.perm is not None:
            a = a[self.perm]
        a = np.ascontiguousarray(a)
        self._ck(self.lib.swe2d_set_bc_field(self.h, int(which), int(slot), _ptr(a)))

    def set_field(self, field, nodal):
        if nodal is None:
            self._ck(self.lib.swe2d_set_field(self.h, field, None))
            return
        vec = field in (_lib.FIELD_MOMENTUM_SOURCE, _lib.FIELD_WIND_STRESS)
        shape = (self.n_cells, self.npc, 2) if vec else (self.n_cells, self.npc)
        a = np.broadcast_to(np.asarray(nodal, dtype=np.float64), shape)
        if self.perm is not None:
            a = a[self.perm]
        a = np.ascontiguousarray(a)
        self._ck(self.lib.swe2d_set_field(self.h, field, _ptr(a)))

    def set_field_vertex(self, field, vertex_values):
        """A continuous P1 coefficient by its vertex values (V,) or (V, 2) in the mesh's vertex numbering; the CG -> DG
        injection runs on the device."""
        vec = field in (_lib.FIELD_MOMENTUM_SOURCE, _lib.FIELD_WIND_STRESS)
        nv = self._keep[1].shape[0]
        a = np.asarray(vertex_values, dtype=np.float64).reshape((nv, 2) if vec else (nv,))
        if self._vperm is not None:
            a = a[self._vperm]
        a = np.ascontiguousarray(a)
        self._ck(self.lib.swe2d_set_field_vertex(self.h, field, _ptr(a)))

    def set_scalar(self, which, value):
        self._ck(self.lib.swe2d_set_scalar(self.h, which, -1.0 if value is None else float(value)))

    # -- time stepping
    def advance(self, n_steps=1):
        self._ck(self.lib.swe2d_advance(self.h, int(n_steps)))

    def solve_stage(self, i_stage):
        self._ck(self.lib.swe2d_solve_stage(self.h, int(i_stage)))

    def advance_forward_euler(self, n_steps=1):
        """timeintegrator.ForwardEuler steps (thetis/timeintegrator.py:115-165)."""
        self._ck(self.lib.swe2d_advance_forward_euler(self.h, int(n_steps)))

    def tracer_forward_euler(self, tid):
        self._ck(self.lib.swe2d_tracer_forward_euler(self.h, int(tid)))

    def solve_stage_cells(self, i_stage, cell_begin, cell_end):
        """Stage ``i_stage`` on device cells [cell_begin, cell_end) (partitions: may include ghost layers)."""
        self._ck(self.lib.swe2d_solve_stage_cells(self.h, int(i_stage), int(cell_begin), int(cell_end)))

    def solve_stage_pair_cells(self, cell_end_0, cell_end_1):
        """stage 0 on [0, cell_end_0) and stage 1 on [0, cell_end_1): one fused launch where the kernel covers the handle"""
        self._ck(self.lib.swe2d_solve_stage_pair_cells(self.h, int(cell_end_0), int(cell_end_1)))

    def solve_step_cells(self, cell_end):
        """all three stages of a step in one launch, stage 3 on [0, cell_end): swe2d_solve_step_cells (the state buffers change places)"""
        self._ck(self.lib.swe2d_solve_step_cells(self.h, int(cell_end)))

    def fused_step_info(self):
        """(a partition's steps should take solve_step_cells, tiles, ring-1 cells, ring-2 cells): swe2d_fused_step_info"""
        out = (ctypes.c_int32*4)()
        self._ck(self.lib.swe2d_fused_step_info(self.h, out))
        return bool(out[0]), int(out[1]), int(out[2]), int(out[3])

    def fused_set_order(self, cells_in_tile_order):
        """The order the tiles of the fused stage pair are cut from (``None``: the device numbering); caller's cell numbering."""
        if cells_in_tile_order is None:
            self._ck(self.lib.swe2d_fused_set_order(self.h, None))
            return
        order = np.asarray(cells_in_tile_order, dtype=np.int64)
        if self.perm is not None:
            order = self.inv_perm[order]
        order = np.ascontiguousarray(order, dtype=np.int32)
        self._ck(self.lib.swe2d_fused_set_order(self.h, _iptr(order)))

    def fused_set_triple_tiles(self, cells_in_tile_order, tile_starts=None):
        """The order the two-ring tiles (all three stages in one launch) are cut from and the positions of that order at which a
        tile must begin (``ordering.triple_tile_order``); ``None``: as the fused pair.  Caller's cell numbering."""
        if cells_in_tile_order is None:
            self._ck(self.lib.swe2d_fused_set_triple_tiles(self.h, None, None, 0))
            return
        order = np.asarray(cells_in_tile_order, dtype=np.int64)
        if self.perm is not None:
            order = self.inv_perm[order]
        order = np.ascontiguousarray(order, dtype=np.int32)
        starts = np.ascontiguousarray(np.zeros(0) if tile_starts is None else tile_starts, dtype=np.int32)
        self._ck(self.lib.swe2d_fused_set_triple_tiles(self.h, _iptr(order), _iptr(starts) if len(starts) else None, len(starts)))

    def forward_euler_cells(self, cell_begin, cell_end):
        """ForwardEuler step of device cells [cell_begin, cell_end) from state buffer 0 into buffer 1 (partitions)."""
        self._ck(self.lib.swe2d_forward_euler_cells(self.h, int(cell_begin), int(cell_end)))

    def swap_state_buffers(self):
        self._ck(self.lib.swe2d_swap_state_buffers(self.h))

    def solve_flow(self, cell_ends):
        """``len(cell_ends)`` (a multiple of 3) consecutive stages in ONE launch without grid-wide barriers (csrc/swe2d_flow.h):
        stage s updates the device cells [0, cell_ends[s]); bit for bit the ``solve_stage_cells`` calls it stands for."""
        ends = np.ascontiguousarray(cell_ends, dtype=np.int32)
        self._ck(self.lib.swe2d_solve_flow(self.h, int(len(ends)), _iptr(ends)))

    def solve_flow_exchange(self, n_cycles, cell_ends):
        """``n_cycles`` exchange cycles of ``len(cell_ends)`` stages each in ONE launch, the peer-to-peer halo exchange inside
        (csrc/swe2d_flow.h, FX kernels): the last cycle's push is received by the next such launch or by ``flow_unpack_pending``."""
        ends = np.ascontiguousarray(cell_ends, dtype=np.int32)
        self._ck(self.lib.swe2d_solve_flow_exchange(self.h, int(n_cycles), int(len(ends)), _iptr(ends)))

    def flow_unpack_pending(self):
        """Receive the last push of ``solve_flow_exchange`` outside a flow launch (ghost cells -> state planes)."""
        self._ck(self.lib.swe2d_flow_unpack_pending(self.h))

    def flow_prepare_exchange(self):
        """Tables of ``solve_flow_exchange`` ahead of its first launch (which must not allocate inside a capture)."""
        self._ck(self.lib.swe2d_flow_prepare_exchange(self.h))

    def flow_set_order(self, cells_in_flow_order):
        """The flow kernel's blocks = consecutive cells of this order (a permutation of the caller's cell ids; default: the
        device numbering).  For partitions: an order in which the ghost cells sit next to the owned cells they touch."""
        order = np.asarray(cells_in_flow_order, dtype=np.int64)
        if self.perm is not None:
            order = self.inv_perm[order]
        order = np.ascontiguousarray(order, dtype=np.int32)
        self._ck(self.lib.swe2d_flow_set_order(self.h, _iptr(order)))

    def flow_supported(self):
        """0: the flow kernel does not cover this handle (configuration, or more 64-cell blocks than the device holds
        resident); 1: covered; 2: covered and without source terms."""
        return int(self.lib.swe2d_flow_supported(self.h))

    def connectivity_info(self):
        """(compact records in use, cells that escape to the wide records): swe2d_connectivity_info"""
        out = (ctypes.c_int32*2)()
        self._ck(self.lib.swe2d_connectivity_info(self.h, out))
        return int(out[0]), int(out[1])

    def fused_pair_info(self):
        """(swe2d_advance takes the fused stage pair, tiles, ring cells, cells): swe2d_fused_pair_info"""
        out = (ctypes.c_int32*4)()
        self._ck(self.lib.swe2d_fused_pair_info(self.h, out))
        return bool(out[0]), int(out[1]), int(out[2]), int(out[3])

    def fused_triple_info(self):
        """(swe2d_advance takes all three stages in one launch, tiles, ring-1 cells, ring-2 cells): swe2d_fused_triple_info"""
        out = (ctypes.c_int32*4)()
        self._ck(self.lib.swe2d_fused_triple_info(self.h, out))
        return bool(out[0]), int(out[1]), int(out[2]), int(out[3])

    def flow_timeouts(self):
        n = ctypes.c_int32()
        self._ck(self.lib.swe2d_flow_status(self.h, ctypes.byref(n)))
        return n.value

    def advance_timed(self, n_steps, per_launch=False):
        """Returns (total ms, mean ms per stage-kernel launch), measured with HIP events on the launch stream."""
        tot = ctypes.c_float()
        avg = ctypes.c_float()
        self._ck(self.lib.swe2d_advance_timed(self.h, int(n_steps), int(per_launch), ctypes.byref(tot), ctypes.byref(avg)))
        return tot.value, avg.value

    def synchronize(self):
        self._ck(self.lib.swe2d_synchronize(self.h))

    def tendency(self):
        ku = np.empty((self.n_cells, self.npc, 2))
        ke = np.empty((self.n_cells, self.npc))
        self._ck(self.lib.swe2d_tendency(self.h, _ptr(ku), _ptr(ke)))
        if self.perm is not None:
            ku, ke = ku[self.inv_perm], ke[self.inv_perm]
        return ku, ke

    def diagnostics(self):
        """{int eta^2, int |u|^2, int (eta+h), min(h+eta)} over owned cells."""
        out = np.empty(4)
        self._ck(self.lib.swe2d_diagnostics(self.h, _ptr(out)))
        return out

    def diagnostics_limbs(self):
        """The three integrals of ``diagnostics`` as order-independent limb sums (int64 [3][4], include/swe2d.h) + the minimum
        depth: partitions add their limbs as integers and round once with ``limbs_to_double``."""
        limbs = np.zeros(3*_lib.SUM_LIMBS, dtype=np.int64)
        lo = ctypes.c_double()
        self._ck(self.lib.swe2d_diagnostics_limbs(self.h, limbs.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), ctypes.byref(lo)))
        return limbs.reshape(3, _lib.SUM_LIMBS), float(lo.value)

    def tracer_diagnostics_limbs(self, tid):
        """limb sums [2][4] of {int T*H dx, int T dx} + (min, max)"""
        limbs = np.zeros(2*_lib.SUM_LIMBS, dtype=np.int64)
        mm = np.empty(2)
        self._ck(self.lib.swe2d_tracer_diagnostics_limbs(self.h, int(tid), limbs.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), _ptr(mm)))
        return limbs.reshape(2, _lib.SUM_LIMBS), mm

    def limbs_to_double(self, limbs):
        """the total of limb sums [6] rounded to the nearest double (swe2d_sum_limbs_to_double)"""
        a = np.ascontiguousarray(limbs, dtype=np.int64).reshape(_lib.SUM_LIMBS)
        return float(self.lib.swe2d_sum_limbs_to_double(a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))))

    # -- tracers + limiter
    def _nodal_in(self, a):
        a = np.asarray(a, dtype=np.float64).reshape(self.n_cells, self.npc)
        if self.perm is not None:
            a = a[self.perm]
        return np.ascontiguousarray(a)

    def _nodal_out(self, a):
        return a[self.inv_perm] if self.perm is not None else a

    def add_tracer(self):
        tid = ctypes.c_int()
        self._ck(self.lib.swe2d_tracer_add(self.h, ctypes.byref(tid)))
        if self._topo_cells is not None and not self._limiter_ready:
            t = np.ascontiguousarray(self._topo_cells, dtype=np.int32)
            self._ck(self.lib.swe2d_limiter_setup(self.h, int(t.max()) + 1, t.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))))
            self._limiter_ready = True
        return tid.value

    def tracer_set_options(self, use_lax_friedrichs_tracer=False, lax_friedrichs_tracer_scaling_factor=1.0,
                           tracer_advective_velocity_factor=1.0):
        self._ck(self.lib.swe2d_tracer_set_options(self.h, int(bool(use_lax_friedrichs_tracer)),
                                                   float(lax_friedrichs_tracer_scaling_factor),
                                                   float(tracer_advective_velocity_factor)))

    def tracer_set_state(self, tid, nodal):
        a = self._nodal_in(nodal)
        self._ck(self.lib.swe2d_tracer_set_state(self.h, tid, _ptr(a)))

    def tracer_get_state(self, tid):
        a = np.empty((self.n_cells, self.npc))
        self._ck(self.lib.swe2d_tracer_get_state(self.h, tid, _ptr(a)))
        return self._nodal_out(a)

    def tracer_set_bc(self, tid, marker, value):
        """``value``: constant Dirichlet value, nodal DG array (N,k) of a Function, or None (default boundary term)."""
        if isinstance(value, np.ndarray) and value.ndim >= 2:
            self._ck(self.lib.swe2d_tracer_set_bc_field(self.h, tid, self._slot(marker), _ptr(self._nodal_in(value))))
            return
        self._ck(self.lib.swe2d_tracer_set_bc(self.h, tid, self._slot(marker), 0 if value is None else 1,
                                              0.0 if value is None else float(value)))

    def tracer_set_bc_velocity(self, tid, marker, uv=None, un=None, flux=None, elev=None):
        """External velocity of the tracer's boundary dict: 'uv' (2 components), 'flux' (with the dict's constant 'elev', if
        any) or 'un', in the precedence of tracer_eq_2d.py:100-112; none of them: uv_ext = uv_in.  Constants, or Function-
        valued entries as DG nodal arrays (N, k[, 2]) / ``FacetValues``."""
        is_field = lambda v: isinstance(v, FacetValues) or (isinstance(v, np.ndarray) and v.ndim >= 2)
        slot = self._slot(marker)
        for kind, v in ((1, uv), (3 if elev is None else 4, flux), (2, un)):
            if v is None:
                continue
            if is_field(v):
                fv = v if isinstance(v, FacetValues) else self.facet_node_values(marker, v)
                cells, facets = self.boundary_facets(slot)
                vals = np.ascontiguousarray(np.asarray(fv.values, dtype=np.float64).reshape(
                    (len(cells), 2, 2) if kind == 1 else (len(cells), 2)))
                self._ck(self.lib.swe2d_tracer_set_bc_velocity_facets(
                    self.h, int(tid), slot, kind, 0.0 if elev is None else float(elev), len(cells),
                    _iptr(self._device_cells(cells)), _iptr(facets), _ptr(vals)))
                return
            if kind == 1:
                u, w = float(v[0]), float(v[1])
            elif kind == 2:
                u, w = float(v), 0.0
            else:
                u, w = float(v), (0.0 if elev is None else float(elev))
            self._ck(self.lib.swe2d_tracer_set_bc_velocity(self.h, int(tid), slot, kind, u, w))
            return
        self._ck(self.lib.swe2d_tracer_set_bc_velocity(self.h, int(tid), slot, 0, 0.0, 0.0))

    def tracer_set_source(self, tid, nodal):
        if nodal is None:
            self._ck(self.lib.swe2d_tracer_set_source(self.h, tid, None))
        else:
            a = self._nodal_in(np.broadcast_to(np.asarray(nodal, dtype=np.float64), (self.n_cells, self.npc)))
            self._ck(self.lib.swe2d_tracer_set_source(self.h, tid, _ptr(a)))

    def tracer_solve_stage(self, tid, i_stage):
        self._ck(self.lib.swe2d_tracer_solve_stage(self.h, tid, int(i_stage)))

    def tracer_tendency(self, tid):
        a = np.empty((self.n_cells, self.npc))
        self._ck(self.lib.swe2d_tracer_tendency(self.h, tid, _ptr(a)))
        return self._nodal_out(a)

    def tracer_limit(self, tid):
        self._ck(self.lib.swe2d_tracer_limit(self.h, tid))

    # -- tracers on partitions (device cell ranges; ghosts keep the caller's order)
    def tracer_solve_stage_cells(self, tid, i_stage, cell_begin, cell_end):
        self._ck(self.lib.swe2d_tracer_solve_stage_cells(self.h, int(tid), int(i_stage), int(cell_begin), int(cell_end)))

    def tracer_swap_buffers(self, tid):
        """after ``tracer_solve_stage_cells(tid, 0, ...)`` on every range of a ForwardEuler step: buffer 1 becomes the tracer"""
        self._ck(self.lib.swe2d_tracer_swap_buffers(self.h, int(tid)))

    def tracer_limit_cells(self, tid, cell_end):
        """Limiter on cells [0, cell_end); means / vertex bounds over every local cell."""
        self._ck(self.lib.swe2d_tracer_limit_cells(self.h, int(tid), int(cell_end)))

    def tracer_halo_pack(self, tid, i_buffer, send_buf_ptr):
        self._ck(self.lib.swe2d_tracer_halo_pack(self.h, int(tid), int(i_buffer), ctypes.c_void_p(send_buf_ptr)))

    def tracer_halo_unpack(self, tid, i_buffer, recv_buf_ptr):
        self._ck(self.lib.swe2d_tracer_halo_unpack(self.h, int(tid), int(i_buffer), ctypes.c_void_p(recv_buf_ptr)))

    def tracer_diagnostics(self, tid):
        """{int T*H dx, int T dx, min nodal T, max nodal T}"""
        out = np.empty(4)
        self._ck(self.lib.swe2d_tracer_diagnostics(self.h, tid, _ptr(out)))
        return out

    def advance_coupled(self, n_steps=1, tracer_only=False, use_limiter=True):
        self._ck(self.lib.swe2d_advance_coupled(self.h, int(n_steps), int(bool(tracer_only)), int(bool(use_limiter))))

    # -- multi-GPU plumbing
    def halo_setup(self, send_cells, recv_cells):
        def dev_ids(c):
            a = np.asarray(c, dtype=np.int64)
            if self.perm is not None:
                a = self.inv_perm[a]
            return np.ascontiguousarray(a, dtype=np.int32)
        a, b = dev_ids(send_cells), dev_ids(recv_cells)
        ip = ctypes.POINTER(ctypes.c_int32)
        self._ck(self.lib.swe2d_halo_setup(self.h, a.size, a.ctypes.data_as(ip), b.size, b.ctypes.data_as(ip)))

    def halo_pack(self, i_buffer, send_buf_ptr):
        self._ck(self.lib.swe2d_halo_pack(self.h, i_buffer, ctypes.c_void_p(send_buf_ptr)))

    def halo_unpack(self, i_buffer, recv_buf_ptr):
        self._ck(self.lib.swe2d_halo_unpack(self.h, i_buffer, ctypes.c_void_p(recv_buf_ptr)))

    # -- peer-to-peer halo exchange through IPC-mapped memory (include/swe2d.h, csrc/swe2d_p2p.h)
    def p2p_create(self, widths):
        w = np.ascontiguousarray(widths, dtype=np.int32)
        self._ck(self.lib.swe2d_p2p_create(self.h, w.size, _iptr(w)))

    def p2p_export(self):
        """(64-byte IPC handle, device address of my landing zone, zone kind: 1 uncached, 2 fine-grained, 3 ordinary)"""
        buf = ctypes.create_string_buffer(_lib.IPC_HANDLE_BYTES)
        base = ctypes.c_void_p()
        kind = ctypes.c_int32()
        self._ck(self.lib.swe2d_p2p_export(self.h, buf, ctypes.byref(base), ctypes.byref(kind)))
        return buf.raw, int(base.value), int(kind.value)

    def p2p_open(self, ipc_handle):
        base = ctypes.c_void_p()
        if os.environ.get('THETIS_AMD_TEST_BREAK_P2P'):
            # tests (a node whose IPC mapping does not work): the library gets a handle that names no allocation and fails in
            # hipIpcOpenMemHandle, the way it would there - the hook is here, not in the library
            ipc_handle = bytes(_lib.IPC_HANDLE_BYTES)
        self._ck(self.lib.swe2d_p2p_open(self.h, ctypes.create_string_buffer(bytes(ipc_handle), _lib.IPC_HANDLE_BYTES),
                                         ctypes.byref(base)))
        return int(base.value)

    def p2p_connect(self, remote_base, send_offset, send_count, remote_recv_offset, remote_flag_index, remote_n_recv, n_from):
        n = len(remote_base)
        bases = (ctypes.c_void_p*max(n, 1))(*[ctypes.c_void_p(b) for b in remote_base])
        arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in (send_offset, send_count, remote_recv_offset,
                                                                   remote_flag_index, remote_n_recv)]
        self._ck(self.lib.swe2d_p2p_connect(self.h, n, bases, *[_iptr(a) for a in arrs], int(n_from)))

    def p2p_push(self, channel, i_buffer):
        self._ck(self.lib.swe2d_p2p_push(self.h, int(channel), int(i_buffer)))

    def p2p_wait_unpack(self, channel, i_buffer):
        self._ck(self.lib.swe2d_p2p_wait_unpack(self.h, int(channel), int(i_buffer)))

    def p2p_push_multi(self, channels, i_buffers):
        c, b = np.ascontiguousarray(channels, dtype=np.int32), np.ascontiguousarray(i_buffers, dtype=np.int32)
        self._ck(self.lib.swe2d_p2p_push_multi(self.h, len(c), _iptr(c), _iptr(b)))

    def p2p_wait_unpack_multi(self, channels, i_buffers):
        c, b = np.ascontiguousarray(channels, dtype=np.int32), np.ascontiguousarray(i_buffers, dtype=np.int32)
        self._ck(self.lib.swe2d_p2p_wait_unpack_multi(self.h, len(c), _iptr(c), _iptr(b)))

    def p2p_status(self, n_channels=1):
        """(epochs sent, epochs received, number of timed-out waits) after a stream synchronisation"""
        a = (ctypes.c_int64*n_channels)()
        b = (ctypes.c_int64*n_channels)()
        t = ctypes.c_int32()
        self._ck(self.lib.swe2d_p2p_status(self.h, a, b, ctypes.byref(t)))
        return list(a), list(b), int(t.value)

    def set_exchange_stream(self, stream_ptr):
        """the peer-to-peer exchange kernels on a stream of their own (None: the handle's stream); ordering by the caller's events"""
        self._ck(self.lib.swe2d_set_exchange_stream(self.h, ctypes.c_void_p(stream_ptr) if stream_ptr else None))

    def set_stream(self, stream_ptr):
        self._ck(self.lib.swe2d_set_stream(self.h, ctypes.c_void_p(stream_ptr) if stream_ptr else None))
