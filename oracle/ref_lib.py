"""
ORACLE - TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/libswe2d_ref.so (the C restatement).
Used by tests/ as the fast checker on meshes too big for the numpy oracle, and by bench.py's
``cpu_baseline`` leg.  Never imported by the product package.
"""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libswe2d_ref.so')

BC_CLOSED, BC_ELEV, BC_UV, BC_UN, BC_FLUX = 0, 1, 2, 4, 8
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)


class _RefStruct(ctypes.Structure):
    _fields_ = [('n_cells', ctypes.c_int), ('nbr', _ip), ('nbf', ctypes.POINTER(ctypes.c_byte)),
                ('xy', _dp), ('h', _dp), ('g', ctypes.c_double), ('nonlinear', ctypes.c_int),
                ('use_lf', ctypes.c_int), ('sigma_lf', ctypes.c_double),
                ('coriolis', _dp), ('linear_drag', ctypes.c_double), ('quad_drag', ctypes.c_double),
                ('manning', ctypes.c_double), ('norm_smoother', ctypes.c_double),
                ('patm', _dp), ('mom_src', _dp), ('vol_src', _dp),
                ('n_markers', ctypes.c_int), ('bc_kind', _ip), ('bc_elev', _dp), ('bc_uv', _dp),
                ('bc_un', _dp), ('bc_flux', _dp), ('bc_len', _dp), ('npc', ctypes.c_int), ('wd', ctypes.c_int), ('alpha', _dp), ('wind', _dp), ('bc_drag', _dp)]


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, 'swe2d_ref.c')):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'libswe2d_ref.so'], stdout=subprocess.DEVNULL)
    return _SO


def load():
    if not os.path.exists(_SO):
        build()
    lib = ctypes.CDLL(_SO)
    lib.swe2d_ref_tendency.argtypes = [ctypes.POINTER(_RefStruct), _dp, _dp, ctypes.c_double, _dp, _dp]
    lib.swe2d_ref_tendency.restype = None
    lib.swe2d_ref_advance.argtypes = [ctypes.POINTER(_RefStruct), _dp, _dp, ctypes.c_double, ctypes.c_int, _dp]
    lib.swe2d_ref_advance.restype = None
    lib.swe2d_ref_num_threads.restype = ctypes.c_int
    lib.swe2d_ref_set_num_threads.argtypes = [ctypes.c_int]
    return lib


def _ptr(a, tp=_dp):
    return None if a is None else a.ctypes.data_as(tp)


class RefSWE(object):
    """C restatement bound to plain mesh arrays (cell_xy (N,3,2), nbr (N,3), nbf (N,3), h (N,3))."""

    def __init__(self, cell_xy, cell_nbr, cell_nbr_facet, h_nodal, g=9.81, use_nonlinear_equations=True,
                 use_lax_friedrichs_velocity=True, lax_friedrichs_velocity_scaling_factor=1.0,
                 coriolis=None, linear_drag_coefficient=None, quadratic_drag_coefficient=None,
                 manning_drag_coefficient=None, norm_smoother=0.0, atmospheric_pressure=None,
                 momentum_source=None, volume_source=None, bnd_conditions=None, boundary_len=None,
                 use_wetting_and_drying=False, wetting_and_drying_alpha=0.5, wind_stress=None):
        self.lib = load()
        n = cell_xy.shape[0]
        self.n = n
        npc = cell_xy.shape[1]
        self.npc = npc
        c = np.ascontiguousarray
        self._keep = dict(
            xy=c(cell_xy, dtype=np.float64), nbr=c(cell_nbr, dtype=np.int32),
            nbf=c(cell_nbr_facet, dtype=np.int8), h=c(h_nodal, dtype=np.float64),
            coriolis=None if coriolis is None else c(np.broadcast_to(coriolis, (n, npc)), dtype=np.float64),
            patm=None if atmospheric_pressure is None else c(atmospheric_pressure, dtype=np.float64),
            mom_src=None if momentum_source is None else c(np.broadcast_to(momentum_source, (n, npc, 2)), dtype=np.float64),
            vol_src=None if volume_source is None else c(np.broadcast_to(volume_source, (n, npc)), dtype=np.float64))
        k = self._keep
        nm = 1
        bnd_conditions = bnd_conditions or {}
        if bnd_conditions:
            nm = max(bnd_conditions.keys()) + 1
        kind = np.zeros(nm, dtype=np.int32)
        elev = np.zeros(nm); uvb = np.zeros((nm, 2)); un = np.zeros(nm); flux = np.zeros(nm); blen = np.ones(nm)
        for mk, funcs in bnd_conditions.items():
            if 'elev' in funcs:
                kind[mk] |= BC_ELEV; elev[mk] = funcs['elev']
            if 'uv' in funcs:
                kind[mk] |= BC_UV; uvb[mk] = funcs['uv']
            elif 'un' in funcs:
                kind[mk] |= BC_UN; un[mk] = funcs['un']
            elif 'flux' in funcs:
                kind[mk] |= BC_FLUX; flux[mk] = funcs['flux']
            if boundary_len is not None and mk in boundary_len:
                blen[mk] = boundary_len[mk]
        k.update(kind=kind, elev=elev, uvb=uvb, un=un, flux=flux, blen=blen)
        s = _RefStruct()
        s.n_cells = n
        s.nbr = _ptr(k['nbr'], _ip)
        s.nbf = _ptr(k['nbf'], ctypes.POINTER(ctypes.c_byte))
        s.xy = _ptr(k['xy']); s.h = _ptr(k['h'])
        s.g = g; s.nonlinear = int(use_nonlinear_equations); s.use_lf = int(use_lax_friedrichs_velocity)
        s.sigma_lf = lax_friedrichs_velocity_scaling_factor
        s.coriolis = _ptr(k['coriolis'])
        s.linear_drag = -1.0 if linear_drag_coefficient is None else linear_drag_coefficient
        s.quad_drag = -1.0 if quadratic_drag_coefficient is None else quadratic_drag_coefficient
        s.manning = -1.0 if manning_drag_coefficient is None else manning_drag_coefficient
        s.norm_smoother = norm_smoother
        s.patm = _ptr(k['patm']); s.mom_src = _ptr(k['mom_src']); s.vol_src = _ptr(k['vol_src'])
        s.n_markers = nm
        s.bc_kind = _ptr(kind, _ip); s.bc_elev = _ptr(elev); s.bc_uv = _ptr(uvb); s.bc_un = _ptr(un)
        s.bc_flux = _ptr(flux); s.bc_len = _ptr(blen)
        s.npc = npc
        s.wd = int(bool(use_wetting_and_drying))
        alpha = c(np.broadcast_to(np.asarray(wetting_and_drying_alpha, dtype=np.float64), (n, npc)))
        k['alpha'] = alpha
        s.alpha = _ptr(alpha)
        wind = None if wind_stress is None else c(np.broadcast_to(wind_stress, (n, npc, 2)), dtype=np.float64)
        k['wind'] = wind
        s.wind = _ptr(wind)
        drag = np.full(nm, -1.0)
        for mk, funcs in bnd_conditions.items():
            if 'drag' in funcs:
                drag[mk] = funcs['drag']
        k['drag'] = drag
        s.bc_drag = _ptr(drag)
        self.s = s

    def tendency(self, uv, eta, dt):
        uv = np.ascontiguousarray(uv, dtype=np.float64); eta = np.ascontiguousarray(eta, dtype=np.float64)
        ku = np.empty_like(uv); ke = np.empty_like(eta)
        self.lib.swe2d_ref_tendency(ctypes.byref(self.s), _ptr(uv), _ptr(eta), dt, _ptr(ku), _ptr(ke))
        return ku, ke

    def advance(self, uv, eta, dt, n_steps):
        """n_steps SSPRK33 steps; returns new (uv, eta)."""
        uv = np.array(uv, dtype=np.float64, order='C'); eta = np.array(eta, dtype=np.float64, order='C')
        work = np.empty(6*self.npc*self.n)
        self.lib.swe2d_ref_advance(ctypes.byref(self.s), _ptr(uv), _ptr(eta), dt, n_steps, _ptr(work))
        return uv, eta

    def advance_blocked(self, uv, eta, dt, n_steps):
        """The timed CPU baseline (swe2d_ref_advance_blocked): same bits as ``advance``; returns (uv, eta, seconds in the
        step loop)."""
        uv = np.array(uv, dtype=np.float64, order='C'); eta = np.array(eta, dtype=np.float64, order='C')
        fn = self.lib.swe2d_ref_advance_blocked
        fn.restype = ctypes.c_double
        fn.argtypes = [ctypes.POINTER(_RefStruct), _dp, _dp, ctypes.c_double, ctypes.c_int]
        sec = fn(ctypes.byref(self.s), _ptr(uv), _ptr(eta), dt, int(n_steps))
        if sec < 0:
            raise RuntimeError('swe2d_ref_advance_blocked: configuration outside its scope (or out of memory)')
        return uv, eta, sec

    def num_threads(self):
        return self.lib.swe2d_ref_num_threads()

    def set_num_threads(self, n):
        self.lib.swe2d_ref_set_num_threads(n)


class _RefTracerStruct(ctypes.Structure):
    _fields_ = [('use_lf', ctypes.c_int), ('lf_factor', ctypes.c_double), ('velocity_factor', ctypes.c_double),
                ('source', _dp), ('n_markers', ctypes.c_int), ('bc_has_value', _ip), ('bc_value', _dp)]


class RefTracer(object):
    """Tracer advection + vertex limiter of the C restatement, on top of a ``RefSWE`` mesh."""

    def __init__(self, ref, use_lax_friedrichs_tracer=False, lax_friedrichs_tracer_scaling_factor=1.0,
                 tracer_advective_velocity_factor=1.0, source=None, bnd_values=None, cell_topo_vertices=None):
        self.ref = ref
        lib = ref.lib
        lib.swe2d_ref_tracer_tendency.argtypes = [ctypes.POINTER(_RefStruct), ctypes.POINTER(_RefTracerStruct), _dp, _dp,
                                                  ctypes.c_double, _dp]
        lib.swe2d_ref_tracer_tendency.restype = None
        lib.swe2d_ref_tracer_step.argtypes = [ctypes.POINTER(_RefStruct), ctypes.POINTER(_RefTracerStruct), _dp, _dp,
                                              ctypes.c_double, _dp]
        lib.swe2d_ref_tracer_step.restype = None
        lib.swe2d_ref_limit.argtypes = [ctypes.POINTER(_RefStruct), _ip, ctypes.c_int, _dp, _dp, _dp]
        lib.swe2d_ref_limit.restype = None
        n = ref.n
        bnd_values = bnd_values or {}
        nm = max(list(bnd_values.keys()) + [0]) + 1
        has = np.zeros(nm, dtype=np.int32)
        val = np.zeros(nm)
        for mk, v in bnd_values.items():
            has[mk] = 1
            val[mk] = v
        src = None if source is None else np.ascontiguousarray(np.broadcast_to(source, (n, ref.npc)), dtype=np.float64)
        self._keep = (has, val, src)
        t = _RefTracerStruct()
        t.use_lf = int(use_lax_friedrichs_tracer)
        t.lf_factor = lax_friedrichs_tracer_scaling_factor
        t.velocity_factor = tracer_advective_velocity_factor
        t.source = _ptr(src)
        t.n_markers = nm
        t.bc_has_value = _ptr(has, _ip)
        t.bc_value = _ptr(val)
        self.t = t
        self.topo = None if cell_topo_vertices is None else np.ascontiguousarray(cell_topo_vertices, dtype=np.int32)

    def tendency(self, T, uv, dt):
        T = np.ascontiguousarray(T, dtype=np.float64); uv = np.ascontiguousarray(uv, dtype=np.float64)
        k = np.empty_like(T)
        self.ref.lib.swe2d_ref_tracer_tendency(ctypes.byref(self.ref.s), ctypes.byref(self.t), _ptr(T), _ptr(uv), dt, _ptr(k))
        return k

    def step(self, T, uv, dt):
        T = np.array(T, dtype=np.float64, order='C'); uv = np.ascontiguousarray(uv, dtype=np.float64)
        work = np.empty(2*self.ref.npc*self.ref.n)
        self.ref.lib.swe2d_ref_tracer_step(ctypes.byref(self.ref.s), ctypes.byref(self.t), _ptr(T), _ptr(uv), dt, _ptr(work))
        return T

    def limit(self, T):
        assert self.topo is not None, 'cell_topo_vertices required for the limiter'
        T = np.array(T, dtype=np.float64, order='C')
        nv = int(self.topo.max()) + 1
        qmin = np.empty(nv); qmax = np.empty(nv)
        self.ref.lib.swe2d_ref_limit(ctypes.byref(self.ref.s), _ptr(self.topo, _ip), nv, _ptr(T), _ptr(qmin), _ptr(qmax))
        return T
