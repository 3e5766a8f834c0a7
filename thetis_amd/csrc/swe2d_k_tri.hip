// swe2d_k_tri.hip - the triangle stage kernels (swe_stage_kernel without wetting-drying and viscosity): instantiations + picker
#include "swe2d_kernels.h"
#include "swe2d_pick.h"

template <bool NL, bool LF, bool U0>
stage_kernel_t pick_src(bool src, int binl)          // binl: 0 epilogue variant, 1 boundary-inline, 2 boundary-inline + LDS exchange
{
    if (binl == 2) return src ? swe_stage_kernel<NL, LF, U0, true, false, false, true, true> : swe_stage_kernel<NL, LF, U0, false, false, false, true, true>;
    if (binl) return src ? swe_stage_kernel<NL, LF, U0, true, false, false, true> : swe_stage_kernel<NL, LF, U0, false, false, false, true>;
    return src ? swe_stage_kernel<NL, LF, U0, true, false> : swe_stage_kernel<NL, LF, U0, false, false>;
}
template <bool NL, bool LF>
stage_kernel_t pick_u0(bool u0, bool src, int binl) { return u0 ? pick_src<NL, LF, true>(src, binl) : pick_src<NL, LF, false>(src, binl); }
template <bool NL>
stage_kernel_t pick_lf(bool lf, bool u0, bool src, int binl) { return lf ? pick_u0<NL, true>(u0, src, binl) : pick_u0<NL, false>(u0, src, binl); }
stage_kernel_t pick_kernel(bool nl, bool lf, bool u0, bool src, int binl)
{
    return nl ? pick_lf<true>(lf, u0, src, binl) : pick_lf<false>(lf, u0, src, binl);
}
