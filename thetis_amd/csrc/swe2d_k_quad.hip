// swe2d_k_quad.hip - the quadrilateral stage kernels (parallelograms and general cells)
#include "swe2d_kernels.h"
#include "swe2d_pick.h"

template <bool NL, bool LF, bool U0>
stage_kernel_t pickq_src(bool src, bool affine)
{
    if (!affine) return src ? swe_stage_kernel_quad<NL, LF, U0, true, false, false> : swe_stage_kernel_quad<NL, LF, U0, false, false, false>;
    return src ? swe_stage_kernel_quad<NL, LF, U0, true, false> : swe_stage_kernel_quad<NL, LF, U0, false, false>;
}
template <bool NL, bool LF>
stage_kernel_t pickq_u0(bool u0, bool src, bool affine) { return u0 ? pickq_src<NL, LF, true>(src, affine) : pickq_src<NL, LF, false>(src, affine); }
template <bool NL>
stage_kernel_t pickq_lf(bool lf, bool u0, bool src, bool affine) { return lf ? pickq_u0<NL, true>(u0, src, affine) : pickq_u0<NL, false>(u0, src, affine); }
stage_kernel_t pick_kernel_quad(bool nl, bool lf, bool u0, bool src, bool affine)
{
    return nl ? pickq_lf<true>(lf, u0, src, affine) : pickq_lf<false>(lf, u0, src, affine);
}
