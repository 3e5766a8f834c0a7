// swe2d_k_tracer.hip - the tracer stage kernels (triangles, with fused diffusion, quadrilaterals)
#include "swe2d_kernels.h"
#include "swe2d_pick.h"

template <bool LF, bool T0>
tracer_kernel_t pick_tracer_kernel_quad_general(bool src)
{
    return src ? swe_tracer_stage_kernel_quad<LF, T0, true, false> : swe_tracer_stage_kernel_quad<LF, T0, false, false>;
}
tracer_kernel_t pick_tracer_kernel_quad(bool lf, bool t0, bool src, bool affine)
{
    if (!affine) {
        if (lf) return t0 ? pick_tracer_kernel_quad_general<true, true>(src) : pick_tracer_kernel_quad_general<true, false>(src);
        return t0 ? pick_tracer_kernel_quad_general<false, true>(src) : pick_tracer_kernel_quad_general<false, false>(src);
    }
    if (lf) {
        if (t0) return src ? swe_tracer_stage_kernel_quad<true, true, true> : swe_tracer_stage_kernel_quad<true, true, false>;
        return src ? swe_tracer_stage_kernel_quad<true, false, true> : swe_tracer_stage_kernel_quad<true, false, false>;
    }
    if (t0) return src ? swe_tracer_stage_kernel_quad<false, true, true> : swe_tracer_stage_kernel_quad<false, true, false>;
    return src ? swe_tracer_stage_kernel_quad<false, false, true> : swe_tracer_stage_kernel_quad<false, false, false>;
}

tracer_kernel_t pick_tracer_kernel_diff(bool lf, bool t0, bool src)      // horizontal diffusion fused in (swe_diff_interior)
{
    if (lf) {
        if (t0) return src ? swe_tracer_stage_kernel<true, true, true, true> : swe_tracer_stage_kernel<true, true, false, true>;
        return src ? swe_tracer_stage_kernel<true, false, true, true> : swe_tracer_stage_kernel<true, false, false, true>;
    }
    if (t0) return src ? swe_tracer_stage_kernel<false, true, true, true> : swe_tracer_stage_kernel<false, true, false, true>;
    return src ? swe_tracer_stage_kernel<false, false, true, true> : swe_tracer_stage_kernel<false, false, false, true>;
}

tracer_kernel_t pick_tracer_kernel(bool lf, bool t0, bool src)
{
    if (lf) {
        if (t0) return src ? swe_tracer_stage_kernel<true, true, true> : swe_tracer_stage_kernel<true, true, false>;
        return src ? swe_tracer_stage_kernel<true, false, true> : swe_tracer_stage_kernel<true, false, false>;
    }
    if (t0) return src ? swe_tracer_stage_kernel<false, true, true> : swe_tracer_stage_kernel<false, true, false>;
    return src ? swe_tracer_stage_kernel<false, false, true> : swe_tracer_stage_kernel<false, false, false>;
}
