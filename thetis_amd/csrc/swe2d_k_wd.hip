// swe2d_k_wd.hip - wetting-drying variants of the stage kernels (triangles and quadrilaterals) and the triangle kernels with the viscosity fused in
#include "swe2d_kernels.h"
#include "swe2d_pick.h"

// wetting-drying variants (nonlinear equations only)
template <bool LF, bool U0>
stage_kernel_t pick_wd_src(bool src, int quad, bool binl)
{
    if (quad == 2) return src ? swe_stage_kernel_quad<true, LF, U0, true, true, false> : swe_stage_kernel_quad<true, LF, U0, false, true, false>;
    if (quad) return src ? swe_stage_kernel_quad<true, LF, U0, true, true> : swe_stage_kernel_quad<true, LF, U0, false, true>;
    if (binl) return src ? swe_stage_kernel<true, LF, U0, true, true, false, true> : swe_stage_kernel<true, LF, U0, false, true, false, true>;
    return src ? swe_stage_kernel<true, LF, U0, true, true> : swe_stage_kernel<true, LF, U0, false, true>;
}
stage_kernel_t pick_kernel_wd(bool lf, bool u0, bool src, int quad, bool binl)
{
    if (lf) return u0 ? pick_wd_src<true, true>(src, quad, binl) : pick_wd_src<true, false>(src, quad, binl);
    return u0 ? pick_wd_src<false, true>(src, quad, binl) : pick_wd_src<false, false>(src, quad, binl);
}
// triangles with the horizontal viscosity fused in (swe_visc_interior)
template <bool NL, bool LF, bool U0>
stage_kernel_t pickv_src(bool src)
{
    return src ? swe_stage_kernel<NL, LF, U0, true, false, true> : swe_stage_kernel<NL, LF, U0, false, false, true>;
}
template <bool NL, bool LF>
stage_kernel_t pickv_u0(bool u0, bool src) { return u0 ? pickv_src<NL, LF, true>(src) : pickv_src<NL, LF, false>(src); }
template <bool NL>
stage_kernel_t pickv_lf(bool lf, bool u0, bool src) { return lf ? pickv_u0<NL, true>(u0, src) : pickv_u0<NL, false>(u0, src); }
stage_kernel_t pick_kernel_visc(bool nl, bool lf, bool u0, bool src)
{
    return nl ? pickv_lf<true>(lf, u0, src) : pickv_lf<false>(lf, u0, src);
}
