// swe2d_unity.hip - every translation unit of the library in one: the -DSWE_RANGE_CHECK / -DSWE_FLOW_DELAY /
// -DSWE_WAVE_TIMING builds keep device-side globals that all kernels must share (tools/range_check.sh, tools/*timing.py).
#define SWE_UNITY 1
#include "swe2d_api.hip"
#include "swe2d_api_flow.hip"
#include "swe2d_api_tracer.hip"
#include "swe2d_api_p2p.hip"
#include "swe2d_api_fuse.hip"
#include "swe2d_k_tri.hip"
#include "swe2d_k_wd.hip"
#include "swe2d_k_quad.hip"
#include "swe2d_k_flow.hip"
#include "swe2d_k_flow_wd.hip"
#include "swe2d_k_tracer.hip"
