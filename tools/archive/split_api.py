"""One-off: cut thetis_amd/csrc/swe2d_api.hip (2639 lines, one translation unit) into separately compiled parts.
Run once from the repo root (round 5); kept for the record of which line range went where."""
import re
import sys

src = open('thetis_amd/csrc/swe2d_api.hip').read().split('\n')
L = lambda a, b: src[a - 1:b]          # 1-based inclusive


def write(name, lines):
    open('thetis_amd/csrc/' + name, 'w').write('\n'.join(lines).rstrip('\n') + '\n')


# ---- swe2d_handle.h: includes, range-check wrappers, Handle, fail(), HIP_TRY, RoctxRange, has_sources, shared declarations
hdr = L(3, 7) + [''] + L(9, 21) + [''] + L(23, 23) + ['']
hdr += L(25, 81)                        # range-check build (unity build only)
hdr += ['', 'namespace swe2d_impl {', '',
        'extern thread_local std::string g_create_error;', '',
        '// Shu-Osher coefficients of SSPRK33 (swe2d_api.hip)',
        'extern const double kBeta[3], kAlpha0[3], kAlphaIn[3];', '',
        'extern std::atomic<unsigned long long> g_next_uid;', '']
hdr += L(102, 216) + ['']
hdr += ['int fail(Handle *h, int code, const std::string &msg);', '']
hdr += L(224, 265) + ['']
hdr += ['inline bool has_sources(const Handle *h)'] + L(268, 272) + ['']
hdr += ['inline int grid_for(int n) { return (n + 255)/256; }', '',
        '// ---- stage launches (swe2d_api.hip)',
        'void fill_stage_args(Handle *h, SweStageArgs &a, int in, int u0, int out, double a0, double a1, double beta, int c0, int c1);',
        'int launch_stage(Handle *h, int in, int u0, int out, double a0, double a1, double beta, int c0, int c1);',
        'int stage_on_range(Handle *h, int i_stage, int c0, int c1);',
        '// ---- dataflow stage loop (swe2d_api_flow.hip)',
        'int flow_build(Handle *h, const int32_t *order);',
        'bool flow_kernel_covers(const Handle *h);',
        'int flow_capacity(Handle *h);',
        'int flow_build_exchange(Handle *h);',
        'int launch_flow(Handle *h, int n_stages, const int32_t *cell_end, int n_cycles = 0);',
        'int flow_check(Handle *h);',
        '// ---- peer-to-peer halo (swe2d_api_p2p.hip)',
        'size_t p2p_channel_offset(const int *width, int c, int n_recv);',
        '',
        '}  // namespace swe2d_impl',
        'using namespace swe2d_impl;']
write('swe2d_handle.h', ['// swe2d_handle.h - what the translation units of the C ABI share: the handle, error plumbing, internal entry points.',
                         '// Not part of the boundary (include/swe2d.h is).',
                         '#pragma once'] + hdr)

# ---- kernel pickers: one translation unit per kernel family (the instantiations are what takes the compile time)
pick_h = '''// swe2d_pick.h - the kernel families, each instantiated in a translation unit of its own (swe2d_k_*.hip); the host code takes
// kernels as function pointers from these pickers.
#pragma once
struct SweStageArgs;
struct SweFlowArgs;
struct SweTracerArgs;
typedef void (*stage_kernel_t)(const SweStageArgs);
typedef void (*flow_kernel_t)(const SweFlowArgs);
typedef void (*tracer_kernel_t)(const SweTracerArgs);
// triangles; binl: 0 epilogue variant, 1 boundary-inline, 2 boundary-inline + LDS trace exchange
stage_kernel_t pick_kernel(bool nl, bool lf, bool u0, bool src, int binl);
// wetting-drying variants (nonlinear equations only); quad: 0 triangles, 1 parallelograms, 2 general quadrilaterals
stage_kernel_t pick_kernel_wd(bool lf, bool u0, bool src, int quad, bool binl);
// triangles with the horizontal viscosity fused in (swe_visc_interior)
stage_kernel_t pick_kernel_visc(bool nl, bool lf, bool u0, bool src);
stage_kernel_t pick_kernel_quad(bool nl, bool lf, bool u0, bool src, bool affine);
// wide: some block of the flow order has more than 64 rim facets (one more granule load per lane and polling trip)
flow_kernel_t pick_flow_kernel(bool nl, bool lf, bool src, bool fx = false, bool wide = false);
tracer_kernel_t pick_tracer_kernel(bool lf, bool t0, bool src);
tracer_kernel_t pick_tracer_kernel_diff(bool lf, bool t0, bool src);      // horizontal diffusion fused in (swe_diff_interior)
tracer_kernel_t pick_tracer_kernel_quad(bool lf, bool t0, bool src, bool affine = true);
'''
open('thetis_amd/csrc/swe2d_pick.h', 'w').write(pick_h)

k_head = lambda what, inc: ['// ' + what, '#include "swe2d_kernels.h"'] + inc + ['#include "swe2d_pick.h"', '']

# triangles, plain
tri = k_head('swe2d_k_tri.hip - the triangle stage kernels (swe_stage_kernel without wetting-drying and viscosity): instantiations + picker', [])
tri += L(276, 282) + L(297, 304)
write('swe2d_k_tri.hip', tri)
# wetting-drying (triangles + quadrilaterals) and fused viscosity
wd = k_head('swe2d_k_wd.hip - wetting-drying variants of the stage kernels (triangles and quadrilaterals) and the triangle kernels with the '
            'viscosity fused in', [])
wd += L(283, 296) + L(305, 318)
write('swe2d_k_wd.hip', wd)
quad = k_head('swe2d_k_quad.hip - the quadrilateral stage kernels (parallelograms and general cells)', [])
quad += L(320, 333)
write('swe2d_k_quad.hip', quad)
flow = k_head('swe2d_k_flow.hip - the dataflow stage loop (swe2d_flow.h): instantiations + picker', ['#include "swe2d_flow.h"'])
fl = L(595, 611)
fl = [l.replace('bool fx = false, bool wide = false', 'bool fx, bool wide') for l in fl]
flow += fl
write('swe2d_k_flow.hip', flow)
tr = k_head('swe2d_k_tracer.hip - the tracer stage kernels (triangles, with fused diffusion, quadrilaterals)', [])
t = L(1759, 1796)
t = [l.replace('bool src, bool affine = true)', 'bool src, bool affine)') for l in t]
tr += t
write('swe2d_k_tracer.hip', tr)

# ---- swe2d_api_flow.hip
fl = ['// swe2d_api_flow.hip - host side of the dataflow stage loop (swe2d_flow.h): slot tables, launches, the ABI entry points',
      '#include "swe2d_handle.h"', '#include "swe2d_pick.h"', '', 'namespace swe2d_impl {', '',
      '// the last flow launch per device of this process (launch_flow)'] + L(88, 91) + ['']
fl += L(484, 591) + [''] + L(613, 633) + [''] + L(643, 782)
fl += ['', '}  // namespace swe2d_impl', '', 'extern "C" {', ''] + L(1403, 1511) + ['', '}  // extern "C"']
fl = [l.replace('int launch_flow(Handle *h, int n_stages, const int32_t *cell_end, int n_cycles = 0)',
                'int launch_flow(Handle *h, int n_stages, const int32_t *cell_end, int n_cycles)') for l in fl]
write('swe2d_api_flow.hip', fl)

# ---- swe2d_api_tracer.hip
tr = ['// swe2d_api_tracer.hip - tracers + vertex-based limiter + the coupled step: host side and ABI entry points',
      '#include "swe2d_handle.h"', '#include "swe2d_pick.h"', '', 'namespace {', '']
tr += L(1798, 1962) + [''] + L(1964, 2352) + [''] + L(2380, 2407)
write('swe2d_api_tracer.hip', tr)

# ---- swe2d_api_p2p.hip
pp = ['// swe2d_api_p2p.hip - halo lists, pack / unpack and the peer-to-peer landing zones (swe2d_p2p.h): ABI entry points',
      '#include "swe2d_handle.h"', '', 'namespace swe2d_impl {', ''] + L(635, 641) + ['', '}  // namespace swe2d_impl', '',
     'extern "C" {', ''] + L(1692, 1748) + ['', '}  // extern "C"', ''] + L(2408, 2628)
write('swe2d_api_p2p.hip', pp)

# ---- swe2d_api.hip: what is left
api = L(1, 2) + ['#include "swe2d_handle.h"', '#include "swe2d_pick.h"', '', 'namespace swe2d_impl {', '',
                 'thread_local std::string g_create_error;',
                 'std::atomic<unsigned long long> g_next_uid{1ull};', '']
api += L(95, 100) + ['']
api = [l.replace('const double kBeta[3] =', 'extern const double kBeta[3] =').replace('const double kAlpha0[3] =', 'extern const double kAlpha0[3] =')
        .replace('const double kAlphaIn[3] =', 'extern const double kAlphaIn[3] =') for l in api]
api += L(218, 222) + ['']
api += L(335, 480) + ['', '}  // namespace swe2d_impl', '']
api += L(786, 1401) + [''] + L(1513, 1690) + [''] + L(2354, 2378) + ['', '}  // extern "C"', ''] + L(2630, 2638)
write('swe2d_api.hip', api)

# ---- the unity build (debug variants with device-side globals: -DSWE_RANGE_CHECK, -DSWE_FLOW_DELAY, -DSWE_WAVE_TIMING)
write('swe2d_unity.hip', ['// swe2d_unity.hip - every translation unit of the library in one: the -DSWE_RANGE_CHECK / -DSWE_FLOW_DELAY /',
                          '// -DSWE_WAVE_TIMING builds keep device-side globals that all kernels must share (tools/range_check.sh, tools/*timing.py).',
                          '#define SWE_UNITY 1'] + ['#include "%s"' % f for f in (
                              'swe2d_api.hip', 'swe2d_api_flow.hip', 'swe2d_api_tracer.hip', 'swe2d_api_p2p.hip', 'swe2d_k_tri.hip',
                              'swe2d_k_wd.hip', 'swe2d_k_quad.hip', 'swe2d_k_flow.hip', 'swe2d_k_tracer.hip')])
print('ok')
