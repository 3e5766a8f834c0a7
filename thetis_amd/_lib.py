"""
ctypes binding of the C ABI declared in include/swe2d.h (libswe2d_hip.so).

There is deliberately no fallback: if the shared library is missing this raises, and if no HIP device is
visible ``swe2d_create`` fails with SWE2D_ERR_NO_DEVICE - the product never computes on the CPU.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('THETIS_AMD_LIB') or os.path.join(_HERE, 'libswe2d_hip.so')   # env: kernel A/B experiments

MAX_MARKERS = 16
BC_ELEV, BC_UV, BC_UN, BC_FLUX = 1, 2, 4, 8
BC_ELEV_FIELD, BC_UV_FIELD, BC_UN_FIELD, BC_FLUX_FIELD = 16, 32, 64, 128
FIELD_CORIOLIS, FIELD_ATMOSPHERIC_PRESSURE, FIELD_MOMENTUM_SOURCE, FIELD_VOLUME_SOURCE, FIELD_WIND_STRESS = 0, 1, 2, 3, 4
FIELD_LINEAR_DRAG, FIELD_QUADRATIC_DRAG, FIELD_MANNING_DRAG, FIELD_NIKURADSE = 5, 6, 7, 8
SCALAR_LINEAR_DRAG, SCALAR_QUADRATIC_DRAG, SCALAR_MANNING_DRAG, SCALAR_NORM_SMOOTHER, SCALAR_NIKURADSE = 0, 1, 2, 3, 4

IPC_HANDLE_BYTES = 64    # include/swe2d.h SWE2D_IPC_HANDLE_BYTES (sizeof(hipIpcMemHandle_t))
SUM_LIMBS = 6             # include/swe2d.h: limbs per order-independent sum (swe2d_diagnostics_limbs)
ABI_VERSION = 12         # include/swe2d.h SWE2D_ABI_VERSION
# include/swe2d.h swe2d_option
(OPT_FUSED_STAGES, OPT_FLOW, OPT_FLOW_WD, OPT_BND_INLINE, OPT_LDSX, OPT_ALTERNATE, OPT_COMPACT_IDX, OPT_VISC_FUSION, OPT_WALL_FAST,
 OPT_FLOW_POLL, OPT_FLOW_CAPACITY, OPT_FLOW_TIMEOUT_MS, OPT_P2P_TIMEOUT_MS, OPT_P2P_ZONE, OPT_ROCTX) = range(15)
OPT_COUNT = 15
SNAPSHOT_SLOTS = 2
OK, ERR_INVALID_ARGUMENT, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED, ERR_NOT_FINITE = 0, -1, -2, -3, -4, -5

# The library itself never reads the environment (include/swe2d.h, swe2d_set_option).  As a convenience of THIS binding the variables
# below are read once per handle, right after swe2d_create (Swe2dDevice.__init__), and handed over as options: A/B runs and the parity
# tests that force every kernel variant in turn set them around the construction of a device.  name -> (option, conversion)
OPTION_ENV = {
    'THETIS_AMD_FUSE12': (OPT_FUSED_STAGES, int),
    'THETIS_AMD_FLOW': (OPT_FLOW, int),
    'THETIS_AMD_FLOW_WD': (OPT_FLOW_WD, int),
    'THETIS_AMD_BND_INLINE': (OPT_BND_INLINE, int),
    'THETIS_AMD_LDSX': (OPT_LDSX, int),
    'THETIS_AMD_ALTERNATE': (OPT_ALTERNATE, int),
    'THETIS_AMD_COMPACT_IDX': (OPT_COMPACT_IDX, int),
    'THETIS_AMD_NO_VISC_FUSION': (OPT_VISC_FUSION, lambda v: 0),
    'THETIS_AMD_WALL_FAST': (OPT_WALL_FAST, int),
    'THETIS_AMD_FLOW_POLL': (OPT_FLOW_POLL, int),
    'THETIS_AMD_FLOW_CAPACITY': (OPT_FLOW_CAPACITY, int),
    'THETIS_AMD_FLOW_TIMEOUT_S': (OPT_FLOW_TIMEOUT_MS, lambda v: max(1, int(round(1e3*float(v))))),
    'THETIS_AMD_P2P_TIMEOUT_S': (OPT_P2P_TIMEOUT_MS, lambda v: max(1, int(round(1e3*float(v))))),
    'THETIS_AMD_P2P_ZONE': (OPT_P2P_ZONE, lambda v: {'uncached': 1, 'finegrained': 2, 'device': 3}[v]),
    'THETIS_AMD_ROCTX': (OPT_ROCTX, int),
}


def options_from_environment():
    """[(option, value)] for the variables of OPTION_ENV that are set."""
    out = []
    for name, (opt, conv) in OPTION_ENV.items():
        v = os.environ.get(name)
        if v is not None and v != '':
            out.append((opt, int(conv(v))))
    return out


_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)
_bp = ctypes.POINTER(ctypes.c_int8)


class Swe2dMesh(ctypes.Structure):
    _fields_ = [('n_cells', ctypes.c_int32), ('n_owned', ctypes.c_int32), ('n_vertices', ctypes.c_int32),
                ('nodes_per_cell', ctypes.c_int32), ('cell_vertices', _ip), ('vertex_xy', _dp),
                ('cell_neighbours', _ip), ('cell_neighbour_facets', _bp), ('bathymetry', _dp),
                ('boundary_len', _dp)]


class Swe2dParams(ctypes.Structure):
    _fields_ = [('g_grav', ctypes.c_double), ('dt', ctypes.c_double),
                ('use_nonlinear_equations', ctypes.c_int32), ('use_lax_friedrichs_velocity', ctypes.c_int32),
                ('lax_friedrichs_velocity_scaling_factor', ctypes.c_double), ('device_id', ctypes.c_int32)]


# every symbol include/swe2d.h declares: name -> (restype, argtypes)
_H = ctypes.c_void_p
SYMBOLS = {
    'swe2d_abi_version': (ctypes.c_int, []),
    'swe2d_connectivity_info': (ctypes.c_int, [_H, ctypes.POINTER(ctypes.c_int32)]),
    'swe2d_fused_pair_info': (ctypes.c_int, [_H, ctypes.POINTER(ctypes.c_int32)]),
    'swe2d_fused_triple_info': (ctypes.c_int, [_H, ctypes.POINTER(ctypes.c_int32)]),
    'swe2d_fused_set_order': (ctypes.c_int, [_H, ctypes.POINTER(ctypes.c_int32)]),
    'swe2d_solve_step_cells': (ctypes.c_int, [_H, ctypes.c_int32]),
    'swe2d_fused_step_info': (ctypes.c_int, [_H, ctypes.POINTER(ctypes.c_int32)]),
    'swe2d_fused_set_triple_tiles': (ctypes.c_int, [_H, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), ctypes.c_int32]),
    'swe2d_solve_stage_pair_cells': (ctypes.c_int, [_H, ctypes.c_int32, ctypes.c_int32]),
    'swe2d_device_count': (ctypes.c_int, []),
    'swe2d_ssprk33_coefficients': (None, [_dp, _dp, _dp]),
    'swe2d_create': (ctypes.c_int, [ctypes.POINTER(Swe2dMesh), ctypes.POINTER(Swe2dParams), ctypes.POINTER(_H)]),
    'swe2d_destroy': (None, [_H]),
    'swe2d_last_error': (ctypes.c_char_p, [_H]),
    'swe2d_set_state': (ctypes.c_int, [_H, _dp, _dp]),
    'swe2d_get_state': (ctypes.c_int, [_H, _dp, _dp]),
    'swe2d_get_stage_state': (ctypes.c_int, [_H, ctypes.c_int, _dp, _dp]),
    'swe2d_state_snapshot': (ctypes.c_int, [_H, ctypes.c_int]),
    'swe2d_state_snapshot_slot': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_int]),
    'swe2d_set_option': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_int]),
    'swe2d_get_option': (ctypes.c_int, [_H, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    'swe2d_set_dt': (ctypes.c_int, [_H, ctypes.c_double]),
    'swe2d_set_bc': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_int, _dp]),
    'swe2d_set_bc_field': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_int, _dp]),
    'swe2d_set_bc_facets': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_int32, _ip, _ip, _dp]),
    'swe2d_set_boundary_drag': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_double]),
    'swe2d_set_field': (ctypes.c_int, [_H, ctypes.c_int, _dp]),
    'swe2d_set_field_vertex': (ctypes.c_int, [_H, ctypes.c_int, _dp]),
    'swe2d_set_scalar': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_double]),
    'swe2d_set_wetting_and_drying': (ctypes.c_int, [_H, ctypes.c_int, _dp]),
    'swe2d_set_viscosity': (ctypes.c_int, [_H, ctypes.c_int, _dp, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int]),
    'swe2d_advance': (ctypes.c_int, [_H, ctypes.c_int]),
    'swe2d_solve_stage': (ctypes.c_int, [_H, ctypes.c_int]),
    'swe2d_advance_timed': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float),
                                           ctypes.POINTER(ctypes.c_float)]),
    'swe2d_synchronize': (ctypes.c_int, [_H]),
    'swe2d_tendency': (ctypes.c_int, [_H, _dp, _dp]),
    'swe2d_diagnostics': (ctypes.c_int, [_H, _dp]),
    'swe2d_diagnostics_limbs': (ctypes.c_int, [_H, ctypes.POINTER(ctypes.c_int64), _dp]),
    'swe2d_sum_limbs_to_double': (ctypes.c_double, [ctypes.POINTER(ctypes.c_int64)]),
    'swe2d_tracer_diagnostics_limbs': (ctypes.c_int, [_H, ctypes.c_int, ctypes.POINTER(ctypes.c_int64), _dp]),
    'swe2d_set_general_quadrilaterals': (ctypes.c_int, [_H, ctypes.c_int]),
    'swe2d_tracer_add': (ctypes.c_int, [_H, ctypes.POINTER(ctypes.c_int)]),
    'swe2d_tracer_set_options': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_double, ctypes.c_double]),
    'swe2d_tracer_set_state': (ctypes.c_int, [_H, ctypes.c_int, _dp]),
    'swe2d_tracer_get_state': (ctypes.c_int, [_H, ctypes.c_int, _dp]),
    'swe2d_tracer_set_bc': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]),
    'swe2d_tracer_set_bc_velocity': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double]),
    'swe2d_tracer_set_bc_velocity_facets': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int32,
                                                            _ip, _ip, _dp]),
    'swe2d_tracer_set_bc_field': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_int, _dp]),
    'swe2d_tracer_set_bc_facets': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_int32, _ip, _ip, _dp]),
    'swe2d_tracer_set_source': (ctypes.c_int, [_H, ctypes.c_int, _dp]),
    'swe2d_tracer_solve_stage_cells': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_int, ctypes.c_int32, ctypes.c_int32]),
    'swe2d_tracer_swap_buffers': (ctypes.c_int, [_H, ctypes.c_int]),
    'swe2d_tracer_limit_cells': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_int32]),
    'swe2d_tracer_halo_pack': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'swe2d_tracer_halo_unpack': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'swe2d_advance_forward_euler': (ctypes.c_int, [_H, ctypes.c_int]),
    'swe2d_tracer_forward_euler': (ctypes.c_int, [_H, ctypes.c_int]),
    'swe2d_tracer_set_conservative': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_int]),
    'swe2d_tracer_set_diffusivity': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_int, _dp, ctypes.c_double, ctypes.c_double]),
    'swe2d_tracer_set_diffusion_bc': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]),
    'swe2d_tracer_solve_stage': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_int]),
    'swe2d_tracer_tendency': (ctypes.c_int, [_H, ctypes.c_int, _dp]),
    'swe2d_limiter_setup': (ctypes.c_int, [_H, ctypes.c_int32, _ip]),
    'swe2d_tracer_limit': (ctypes.c_int, [_H, ctypes.c_int]),
    'swe2d_tracer_diagnostics': (ctypes.c_int, [_H, ctypes.c_int, _dp]),
    'swe2d_advance_coupled': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    'swe2d_debug_calibration_copy': (ctypes.c_int, [_H, ctypes.c_int]),
    'swe2d_halo_setup': (ctypes.c_int, [_H, ctypes.c_int32, _ip, ctypes.c_int32, _ip]),
    'swe2d_halo_pack': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_void_p]),
    'swe2d_halo_unpack': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_void_p]),
    'swe2d_p2p_create': (ctypes.c_int, [_H, ctypes.c_int32, _ip]),
    'swe2d_p2p_export': (ctypes.c_int, [_H, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), _ip]),
    'swe2d_p2p_open': (ctypes.c_int, [_H, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]),
    'swe2d_p2p_connect': (ctypes.c_int, [_H, ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p), _ip, _ip, _ip, _ip, _ip,
                                          ctypes.c_int32]),
    'swe2d_p2p_push': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_int]),
    'swe2d_p2p_wait_unpack': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_int]),
    'swe2d_p2p_push_multi': (ctypes.c_int, [_H, ctypes.c_int, _ip, _ip]),
    'swe2d_p2p_wait_unpack_multi': (ctypes.c_int, [_H, ctypes.c_int, _ip, _ip]),
    'swe2d_p2p_status': (ctypes.c_int, [_H, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), _ip]),
    'swe2d_solve_stage_cells': (ctypes.c_int, [_H, ctypes.c_int, ctypes.c_int32, ctypes.c_int32]),
    'swe2d_forward_euler_cells': (ctypes.c_int, [_H, ctypes.c_int32, ctypes.c_int32]),
    'swe2d_swap_state_buffers': (ctypes.c_int, [_H]),
    'swe2d_solve_flow': (ctypes.c_int, [_H, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32)]),
    'swe2d_flow_supported': (ctypes.c_int, [_H]),
    'swe2d_flow_prepare_exchange': (ctypes.c_int, [_H]),
    'swe2d_flow_unpack_pending': (ctypes.c_int, [_H]),
    'swe2d_solve_flow_exchange': (ctypes.c_int, [_H, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32)]),
    'swe2d_flow_set_order': (ctypes.c_int, [_H, ctypes.POINTER(ctypes.c_int32)]),
    'swe2d_flow_status': (ctypes.c_int, [_H, ctypes.POINTER(ctypes.c_int32)]),
    'swe2d_debug_flow_poke': (ctypes.c_int, [_H, ctypes.c_int32, ctypes.c_int32]),
    'swe2d_debug_flow_delay': (ctypes.c_int, [_H, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    'swe2d_debug_flow_tear': (ctypes.c_int, [_H, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    'swe2d_set_stream': (ctypes.c_int, [_H, ctypes.c_void_p]),
    'swe2d_set_exchange_stream': (ctypes.c_int, [_H, ctypes.c_void_p]),
}

_lib = None


class Swe2dError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('swe2d error {:d}: {:s}'.format(code, msg))
        self.code = code


def load():
    """Load libswe2d_hip.so (built by ``thetis_amd._build.build`` / ``__graft_entry__.build``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError('HIP extension {:} is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                          '(there is no CPU fallback)'.format(LIB_PATH))
    # PyTorch-ROCm bundles its own libamdhip64; two HIP runtimes in one process do not see each other's devices.
    # Import torch first (when present) so that the extension binds to the runtime torch uses for RCCL / streams.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)     # AttributeError if the library does not export a declared symbol
        except AttributeError:
            if os.environ.get('THETIS_AMD_LIB'):
                continue                # kernel A/B experiments against older builds (tools/kbench.py)
            raise
        fn.restype = restype
        fn.argtypes = argtypes
    if not os.environ.get('THETIS_AMD_LIB') and lib.swe2d_abi_version() != ABI_VERSION:
        raise ImportError('{:} implements ABI version {:d}, this package binds version {:d}: rebuild it '
                          '(python -c "import __graft_entry__ as g; g.build()")'.format(LIB_PATH, lib.swe2d_abi_version(),
                                                                                          ABI_VERSION))
    _lib = lib
    return lib


def check(rc, handle=None):
    if rc != OK:
        msg = load().swe2d_last_error(handle)
        raise Swe2dError(rc, msg.decode() if msg else '')
