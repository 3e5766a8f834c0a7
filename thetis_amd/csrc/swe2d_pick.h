// swe2d_pick.h - the kernel families, each instantiated in a translation unit of its own (swe2d_k_*.hip); the host code takes
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
// poll: granule loads per lane and polling trip (3 / 4 / 6 / 9 for blocks of at most 32 / 42 / 64 / more rim facets)
flow_kernel_t pick_flow_kernel(bool nl, bool lf, bool src, bool fx = false, int poll = 6);
flow_kernel_t pick_flow_kernel_wd(bool lf, bool src, bool fx, int poll);        // wetting-drying (swe2d_k_flow_wd.hip)
tracer_kernel_t pick_tracer_kernel(bool lf, bool t0, bool src);
tracer_kernel_t pick_tracer_kernel_diff(bool lf, bool t0, bool src);      // horizontal diffusion fused in (swe_diff_interior)
tracer_kernel_t pick_tracer_kernel_quad(bool lf, bool t0, bool src, bool affine = true);
// -DSWE_FLOW_DELAY (which = 0) / -DSWE_FLOW_TEAR (1) builds: sets the device-side switches of the adversary; -1 in other builds
int swe_flow_debug_config(int which, const int cfg[4]);
