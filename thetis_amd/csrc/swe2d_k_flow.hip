// swe2d_k_flow.hip - the dataflow stage loop (swe2d_flow.h): instantiations + picker
#include "swe2d_kernels.h"
#include "swe2d_flow.h"
#include "swe2d_pick.h"

template <bool NL, bool LF, int POLL>
flow_kernel_t pick_flow_src(bool src, bool fx)
{
    if (fx) return src ? swe_flow_kernel<NL, LF, true, true, POLL> : swe_flow_kernel<NL, LF, false, true, POLL>;
    return src ? swe_flow_kernel<NL, LF, true, false, POLL> : swe_flow_kernel<NL, LF, false, false, POLL>;
}
template <int POLL>
flow_kernel_t pick_flow_poll(bool nl, bool lf, bool src, bool fx)
{
    return nl ? (lf ? pick_flow_src<true, true, POLL>(src, fx) : pick_flow_src<true, false, POLL>(src, fx))
              : (lf ? pick_flow_src<false, true, POLL>(src, fx) : pick_flow_src<false, false, POLL>(src, fx));
}
// poll: granule loads per lane and polling trip - 3 where no block of the flow order has more than 32 rim facets (the 8 x 4-quad
// blocks of ordering.flow_block_order), 4 up to 42 (its bisection boxes on unstructured meshes), 6 up to 64, 9 beyond (a block that needs a second trip per pass paces the whole launch)
flow_kernel_t pick_flow_kernel(bool nl, bool lf, bool src, bool fx, int poll)
{
    return poll >= 9 ? pick_flow_poll<9>(nl, lf, src, fx) : (poll >= 6 ? pick_flow_poll<6>(nl, lf, src, fx)
         : (poll >= 4 ? pick_flow_poll<4>(nl, lf, src, fx) : pick_flow_poll<3>(nl, lf, src, fx)));
}

// the adversary builds' device-side switches live in this translation unit (with the kernels that read them)
int swe_flow_debug_config(int which, const int cfg[4])
{
#ifdef SWE_FLOW_DELAY
    if (which == 0) return hipMemcpyToSymbol(HIP_SYMBOL(swe_flow_delay), cfg, 4*sizeof(int)) == hipSuccess ? 0 : 1;
#endif
#ifdef SWE_FLOW_TEAR
    if (which == 1) return hipMemcpyToSymbol(HIP_SYMBOL(swe_flow_tear), cfg, 4*sizeof(int)) == hipSuccess ? 0 : 1;
#endif
    (void)which; (void)cfg;
    return -1;                               // this build has no such switch
}
