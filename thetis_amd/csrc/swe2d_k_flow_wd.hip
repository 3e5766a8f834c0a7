// swe2d_k_flow_wd.hip - the dataflow stage loop with wetting-drying (swe_flow_kernel<true, LF, SRC, FX, POLL, true>, round 5):
// instantiations + picker, in a translation unit of their own (they compile as long as all the other flow kernels together)
#include "swe2d_kernels.h"
#include "swe2d_flow.h"
#include "swe2d_pick.h"

template <bool LF, int POLL>
static flow_kernel_t pick_flow_wd_src(bool src, bool fx)
{
    if (fx) return src ? swe_flow_kernel<true, LF, true, true, POLL, true> : swe_flow_kernel<true, LF, false, true, POLL, true>;
    return src ? swe_flow_kernel<true, LF, true, false, POLL, true> : swe_flow_kernel<true, LF, false, false, POLL, true>;
}
// wetting-drying variants (nonlinear equations only); poll: see pick_flow_kernel
flow_kernel_t pick_flow_kernel_wd(bool lf, bool src, bool fx, int poll)
{
    if (poll >= 9) return lf ? pick_flow_wd_src<true, 9>(src, fx) : pick_flow_wd_src<false, 9>(src, fx);
    if (poll >= 6) return lf ? pick_flow_wd_src<true, 6>(src, fx) : pick_flow_wd_src<false, 6>(src, fx);
    if (poll >= 4) return lf ? pick_flow_wd_src<true, 4>(src, fx) : pick_flow_wd_src<false, 4>(src, fx);
    return lf ? pick_flow_wd_src<true, 3>(src, fx) : pick_flow_wd_src<false, 3>(src, fx);
}
