// swe2d_api.hip - C ABI (include/swe2d.h) over the HIP stage kernels.  gfx950 only, no CPU fallback: every entry
// point fails with SWE2D_ERR_NO_DEVICE / SWE2D_ERR_HIP when the HIP runtime or a device is missing.
#include "../../include/swe2d.h"
#include "swe2d_kernels.h"
#include "swe2d_sipg.h"
#include "swe2d_flow.h"
#include "swe2d_p2p.h"

#include <dlfcn.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>
#include <atomic>
#include <mutex>

static_assert(SWE2D_MAX_MARKERS == SWE_MAX_MARKERS, "marker table size mismatch");

#ifdef SWE_RANGE_CHECK
// Range-checked build (swe2d_kernels.h): every device allocation of this library is recorded with its requested size;
// the sorted table is copied to the device before a launch whenever it changed.
#include <map>
#include <mutex>
namespace {
std::mutex g_chk_mutex;
std::map<unsigned long long, unsigned long long> g_chk_allocs;       // base -> end
bool g_chk_dirty = true;
unsigned long long g_chk_launches = 0;
hipError_t swe_chk_malloc(void **p, size_t n)
{
    const hipError_t e = hipMalloc(p, n);
    if (e == hipSuccess && *p) {
        std::lock_guard<std::mutex> lock(g_chk_mutex);
        const char *st = getenv("THETIS_AMD_RANGE_SELFTEST");                  // negative control: record half of every allocation
        g_chk_allocs[(unsigned long long)*p] = (unsigned long long)*p + ((st && atoi(st)) ? n/2 : n);
        g_chk_dirty = true;
    }
    return e;
}
template <class T> hipError_t swe_chk_malloc(T **p, size_t n) { return swe_chk_malloc((void **)p, n); }
hipError_t swe_chk_free(void *p)
{
    { std::lock_guard<std::mutex> lock(g_chk_mutex); g_chk_allocs.erase((unsigned long long)p); g_chk_dirty = true; }
    return hipFree(p);
}
void swe_chk_sync(hipStream_t stream)
{
    std::lock_guard<std::mutex> lock(g_chk_mutex);
    g_chk_launches++;
    if (!g_chk_dirty) return;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return;
    static SweChkTable t;
    t.n = 0;
    for (auto &kv : g_chk_allocs) if (t.n < SWE_CHK_MAX) { t.lo[t.n] = kv.first; t.hi[t.n] = kv.second; t.n++; }
    (void)hipDeviceSynchronize();
    (void)hipMemcpyToSymbol(HIP_SYMBOL(swe_chk_tab), &t, sizeof(t));
    g_chk_dirty = false;
}
}
#define hipMalloc(p, n) swe_chk_malloc(p, n)
#define hipFree(p) swe_chk_free(p)
#define SWE_CHK_SYNC(stream) swe_chk_sync(stream)
extern "C" int swe2d_debug_range_report(unsigned long long out[5])
{
    std::lock_guard<std::mutex> lock(g_chk_mutex);
    (void)hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(swe_chk_report), 4*sizeof(unsigned long long)) != hipSuccess) return 1;
    out[3] = g_chk_launches;
    out[4] = g_chk_allocs.size();
    return 0;
}
#else
#define SWE_CHK_SYNC(stream) ((void)0)
#endif

namespace {

thread_local std::string g_create_error;

// the last flow launch per device of this process (launch_flow)
struct FlowChain { hipEvent_t ev = nullptr; unsigned long long last_uid = 0ull; };
constexpr int kFlowChainDevices = 64;
FlowChain g_flow_chain[kFlowChainDevices];
std::mutex g_flow_chain_mu;
std::atomic<unsigned long long> g_next_uid{1ull};


// Shu-Osher coefficients of SSPRK33: output of thetis/rungekutta.py:13-87 (butcher_to_shuosher_form) for the
// tableau of rungekutta.py:342-346; pinned by tests/golden/shuosher_ssprk33.json.
//   U1 = 1*k0 + 1*U0;  U2 = 1/4*k1 + 3/4*U0 + 1/4*U1;  U3 = b32*k2 + a30*U0 + a32*U2
const double kBeta[3] = {1.0, 0.25, 0.6666666666666666};
const double kAlpha0[3] = {1.0, 0.75, 0.33333333333333337};   // weight of stage_sol[0]
const double kAlphaIn[3] = {0.0, 0.25, 0.6666666666666666};   // weight of the stage's input (stage 0: U0 itself)

struct Handle {
    unsigned long long uid = g_next_uid.fetch_add(1ull);   // never reused (a freed handle's address may be)
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t my_stream = nullptr;
    hipStream_t xstream = nullptr;                     // the exchange kernels' stream (swe2d_set_exchange_stream), null: `stream`
    int n_cells = 0, n_owned = 0, n_interior = 0, n_vertices = 0;
    int npc = 3;                                       // nodes per cell: 3 triangles, 4 quadrilaterals
    bool affine = true;                                // quadrilaterals: every cell a parallelogram (constant Jacobian, tensor mass inverse)
    bool affine_local = true;                          // ... as found in this handle's own cells (affine may be forced off: swe2d_set_general_quadrilaterals)
    size_t stride = 0;
    double *state[3] = {nullptr, nullptr, nullptr};   // A (U0 / step result), B (U1), C (U2)
    int *nbr = nullptr, *cv = nullptr;
    // compact boundary uploads (swe2d_set_bc_facets): the (cell, facet) lists of the last calls stay on the device, a repeated
    // call with the same lists (update_forcings at every stage) only uploads the values
    struct FacetList { std::vector<int32_t> cells, facets; int *dev = nullptr; };
    FacetList facet_lists[8];
    int facet_list_next = 0;
    int4 *opp4 = nullptr;                               // triangles: opposite vertices of the neighbours (fused viscosity)
    int *bnd_cells = nullptr;                           // cells with a boundary facet (boundary-only SIPG launch)
    int n_bnd = 0;
    bool fuse_visc = true;                              // THETIS_AMD_NO_VISC_FUSION=1: separate SIPG pass (A/B, debugging)
    int4 *idx4 = nullptr;                               // packed triangle connectivity (stage kernel), see SweStageArgs
    int2 *idx2 = nullptr;
    std::vector<int> h_nbr;                             // host copy of the packed neighbour codes [3][S] (triangles; flow_build)
    // dataflow stage loop (swe2d_flow.h): per-block stage counters, status word {timeouts, first late block + 1}
    unsigned *flow_flag = nullptr, *flow_status = nullptr;
    int4 *flow_xo4 = nullptr;                           // exchange slots of the rim facets (facets between two 64-cell blocks), see SweFlowArgs
    int2 *flow_xo2 = nullptr;
    int2 *flow_xblk = nullptr;
    int *flow_xsrc = nullptr;
    std::vector<int> h_send, h_recv;                    // host copies of the halo lists (swe2d_halo_setup)
    std::vector<int> flow_fpos;                         // cell -> flow position
    int2 *flow_xsend = nullptr;                         // FX: per position, the cell's places in the send list
    int *flow_xrecv = nullptr;                          // FX: per position, the cell's place in the receive list
    unsigned *flow_xtick = nullptr;
    int flow_push_blocks = 0, flow_recv_blocks = 0;
    bool flow_x_ready = false;                          // the FX tables match the halo lists and the flow order
    int *flow_cell = nullptr;                           // [flow_blocks*64] flow position -> cell (< 0: padding lane, -1 - cell to mimic)
    unsigned flow_parity_bytes = 0;
    void *flow_ex = nullptr;
    size_t flow_ex_bytes = 0;
    int flow_blocks = 0;                                // 64-cell blocks of the handle
    int flow_capacity = -1;                             // resident one-wave workgroups of the flow kernel on this device (-1: not asked yet)
    int flow_max_rim = 0;                               // most rim facets of a block in the current flow order (selects the polling width)
    int launch_parity = 0;                              // direction of the next large stage launch (launch_stage)
    bool flow_used = false;                             // a flow launch since the status word was last read
    double flow_timeout_s = 2.0;                        // THETIS_AMD_FLOW_TIMEOUT_S
    double *vx = nullptr, *vy = nullptr, *vh = nullptr;
    double *bc_field[4] = {nullptr, nullptr, nullptr, nullptr};  // Function-valued boundary data per facet: elev, uv, un, flux
    double *valpha = nullptr;                          // per-vertex wetting-drying alpha
    bool wd = false;
    // SIPG horizontal viscosity (optional pass after each stage kernel)
    bool visc = false;
    double *nu_v = nullptr;                            // per-vertex viscosity or null (constant)
    double nu_const = 0.0, sipg_factor = 1.0;
    int visc_grad_div = 0, visc_grad_depth = 1;
    double *field[SWE2D_FIELD_COUNT] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    double scalar[SWE2D_SCALAR_COUNT] = {-1.0, -1.0, -1.0, 0.0, -1.0};
    double *stage_uv = nullptr, *stage_eta = nullptr;  // device staging in host layout (6N + 3N)
    double *partial = nullptr;                         // diagnostics partial sums
    int n_partial_blocks = 0;
    unsigned long long *diag_acc = nullptr;            // limb sums of the diagnostics kernels (swe_sum_accumulate) + one counter
    int *send_cells = nullptr, *recv_cells = nullptr;
    int n_send = 0, n_recv = 0;
    // peer-to-peer halo (swe2d_p2p.h): my landing zone, the peers' zones mapped here, per-channel device counters
    struct P2p {
        void *zone = nullptr;
        size_t zone_bytes = 0;
        int zone_kind = 0;                               // 1 uncached, 2 fine-grained, 3 ordinary device memory
        int n_channels = 0;
        int width[SWE_P2P_MAX_CHANNELS] = {0};
        SweP2pCounters *ctr = nullptr;                   // [n_channels]
        std::vector<void *> opened;                      // hipIpcOpenMemHandle mappings to close
        int n_peers = 0, n_from = 0;
        int off[SWE_P2P_MAX_PEERS], cnt[SWE_P2P_MAX_PEERS], remote_off[SWE_P2P_MAX_PEERS], remote_flag[SWE_P2P_MAX_PEERS],
            remote_n_recv[SWE_P2P_MAX_PEERS];
        char *remote_base[SWE_P2P_MAX_PEERS];
        double timeout_s = 5.0;
    } p2p;
    // tracers + limiter
    struct Tracer {
        double *buf[3] = {nullptr, nullptr, nullptr};   // A (T0 / result), B, C: 3 planes each
        double *source = nullptr;
        bool conservative = false;                      // options.tracer[label].use_conservative_form
        double *bc_value_f = nullptr;                   // Function-valued 'value' boundaries, npc*npc planes
        int bc_vel_kind[SWE_MAX_MARKERS];               // 0 none, 1 'uv', 2 'un'
        double bc_u[SWE_MAX_MARKERS], bc_v[SWE_MAX_MARKERS];
        double *bc_vel_f = nullptr;                     // Function-valued 'uv' / 'un' / 'flux', 4*npc planes per facet layout
        int bc_vel_field[SWE_MAX_MARKERS];
        int bc_has_value[SWE_MAX_MARKERS];
        double bc_value[SWE_MAX_MARKERS];
        bool diff = false;                              // SIPG horizontal diffusion
        double *mu_v = nullptr;
        double mu_const = 0.0, sipg_factor = 1.0;
        int bc_diff_kind[SWE_MAX_MARKERS];
        double bc_diff_flux[SWE_MAX_MARKERS];
    };
    std::vector<Tracer> tracers;
    int tracer_use_lf = 0;
    double tracer_lf_factor = 1.0, tracer_vel_factor = 1.0;
    std::vector<int> host_cells;                 // [n][3] vertex ids as given (limiter default topology)
    std::vector<int> host_nbr;                   // [n][3]
    int lim_nv = 0;
    int *lim_v2c_off = nullptr, *lim_v2c_cell = nullptr, *lim_vbf_off = nullptr, *lim_vbf_facet = nullptr, *lim_tv = nullptr;
    double *lim_mean = nullptr, *lim_qmin = nullptr, *lim_qmax = nullptr;
    swe2d_params par{};
    SweBcTable bc{};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;
};

inline Handle *H(swe2d_handle *h) { return reinterpret_cast<Handle *>(h); }
inline const Handle *H(const swe2d_handle *h) { return reinterpret_cast<const Handle *>(h); }

int fail(Handle *h, int code, const std::string &msg)
{
    if (h) h->err = msg; else g_create_error = msg;
    return code;
}

#define HIP_TRY(h, expr)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(h, SWE2D_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

// Optional ROCTx ranges around the entry points that advance the state (THETIS_AMD_ROCTX=1): they show up as named ranges in
// `rocprofv3 --marker-trace` next to the kernel trace.  The tracing library is looked up at run time (rocprofiler-sdk's
// librocprofiler-sdk-roctx.so, else roctracer's libroctx64.so); without it, or without the variable, the ranges are no-ops.
struct RoctxRange {
    typedef int (*push_t)(const char *);
    typedef int (*pop_t)();
    static void resolve(push_t &push, pop_t &pop)
    {
        static bool done = false;
        static push_t p_push = nullptr;
        static pop_t p_pop = nullptr;
        if (!done) {
            done = true;
            if (std::getenv("THETIS_AMD_ROCTX")) {
                for (const char *name : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
                    if (void *lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                        p_push = reinterpret_cast<push_t>(dlsym(lib, "roctxRangePushA"));
                        p_pop = reinterpret_cast<pop_t>(dlsym(lib, "roctxRangePop"));
                        if (p_push && p_pop) break;
                        p_push = nullptr; p_pop = nullptr;
                    }
                }
            }
        }
        push = p_push; pop = p_pop;
    }
    pop_t pop_ = nullptr;
    explicit RoctxRange(const char *name)
    {
        push_t push;
        resolve(push, pop_);
        if (push) push(name); else pop_ = nullptr;
    }
    ~RoctxRange() { if (pop_) pop_(); }
};

bool has_sources(const Handle *h)
{
    for (int i = 0; i < SWE2D_FIELD_COUNT; i++) if (h->field[i]) return true;
    return h->scalar[SWE2D_SCALAR_LINEAR_DRAG] >= 0 || h->scalar[SWE2D_SCALAR_QUADRATIC_DRAG] >= 0
           || h->scalar[SWE2D_SCALAR_MANNING_DRAG] >= 0 || h->scalar[SWE2D_SCALAR_NIKURADSE] >= 0;
}

typedef void (*stage_kernel_t)(const SweStageArgs);

template <bool NL, bool LF, bool U0>
stage_kernel_t pick_src(bool src, int binl)          // binl: 0 epilogue variant, 1 boundary-inline, 2 boundary-inline + LDS exchange
{
    if (binl == 2) return src ? swe_stage_kernel<NL, LF, U0, true, false, false, true, true> : swe_stage_kernel<NL, LF, U0, false, false, false, true, true>;
    if (binl) return src ? swe_stage_kernel<NL, LF, U0, true, false, false, true> : swe_stage_kernel<NL, LF, U0, false, false, false, true>;
    return src ? swe_stage_kernel<NL, LF, U0, true, false> : swe_stage_kernel<NL, LF, U0, false, false>;
}
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
template <bool NL, bool LF>
stage_kernel_t pick_u0(bool u0, bool src, int binl) { return u0 ? pick_src<NL, LF, true>(src, binl) : pick_src<NL, LF, false>(src, binl); }
template <bool NL>
stage_kernel_t pick_lf(bool lf, bool u0, bool src, int binl) { return lf ? pick_u0<NL, true>(u0, src, binl) : pick_u0<NL, false>(u0, src, binl); }
stage_kernel_t pick_kernel(bool nl, bool lf, bool u0, bool src, int binl)
{
    return nl ? pick_lf<true>(lf, u0, src, binl) : pick_lf<false>(lf, u0, src, binl);
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

void fill_stage_args(Handle *h, SweStageArgs &a, int in, int u0, int out, double a0, double a1, double beta, int c0, int c1)
{
    a.uin = h->state[in];
    a.u0 = h->state[u0];
    a.uout = h->state[out];
    a.stride = h->stride;
    a.nbr = h->nbr;
    a.cv = h->cv;
    a.vx = h->vx; a.vy = h->vy; a.vh = h->vh;
    a.valpha = h->valpha;
    a.idx4 = h->idx4; a.idx2 = h->idx2;
    a.cell_begin = c0; a.cell_end = c1;
    a.reverse = 0;
    { const char *e = std::getenv("THETIS_AMD_WALL_FAST"); a.wall_general = (e && std::atoi(e) == 0) ? 1 : 0; }
    a.wd_skip_relax = (h->wd && h->visc) ? 1 : 0;
    a.g = h->par.g_grav;
    a.sigma_lf = h->par.lax_friedrichs_velocity_scaling_factor;
    a.dt = h->par.dt;
    a.a0 = a0; a.a1 = a1; a.beta = beta;
    a.coriolis = h->field[SWE2D_FIELD_CORIOLIS];
    a.patm = h->field[SWE2D_FIELD_ATMOSPHERIC_PRESSURE];
    a.msrc = h->field[SWE2D_FIELD_MOMENTUM_SOURCE];
    a.vsrc = h->field[SWE2D_FIELD_VOLUME_SOURCE];
    a.wind = h->field[SWE2D_FIELD_WIND_STRESS];
    a.bc_elev_f = h->bc_field[0]; a.bc_uv_f = h->bc_field[1]; a.bc_un_f = h->bc_field[2]; a.bc_flux_f = h->bc_field[3];
    a.npc_ = h->npc;
    a.linear_drag = h->scalar[SWE2D_SCALAR_LINEAR_DRAG];
    a.quad_drag = h->scalar[SWE2D_SCALAR_QUADRATIC_DRAG];
    a.manning = h->scalar[SWE2D_SCALAR_MANNING_DRAG];
    a.norm_smoother = h->scalar[SWE2D_SCALAR_NORM_SMOOTHER];
    a.nikuradse = h->scalar[SWE2D_SCALAR_NIKURADSE];
    a.lin_drag_f = h->field[SWE2D_FIELD_LINEAR_DRAG];
    a.quad_f = nullptr; a.quad_f_kind = 0;
    if (h->field[SWE2D_FIELD_QUADRATIC_DRAG]) { a.quad_f = h->field[SWE2D_FIELD_QUADRATIC_DRAG]; a.quad_f_kind = 1; }
    if (h->field[SWE2D_FIELD_MANNING_DRAG]) { a.quad_f = h->field[SWE2D_FIELD_MANNING_DRAG]; a.quad_f_kind = 2; }
    if (h->field[SWE2D_FIELD_NIKURADSE]) { a.quad_f = h->field[SWE2D_FIELD_NIKURADSE]; a.quad_f_kind = 3; }
    a.bc = h->bc;
    for (int m = 0; m < SWE_MAX_MARKERS; m++)
        if (a.bc.drag[m] >= 0.0) a.bc.kind[m] |= SWE_BC_HAS_DRAG;          // one table read per boundary facet in the kernel
    a.opp4 = h->opp4;
    a.nu_v = h->nu_v; a.nu_const = h->nu_const;
    a.visc_sipg = 3.0*h->sipg_factor;
    a.visc_grad_div = h->visc_grad_div; a.visc_grad_depth = h->visc_grad_depth;
}

// Launch one stage on cells [c0, c1).  in/out/u0 are state buffer indices.
int launch_stage(Handle *h, int in, int u0, int out, double a0, double a1, double beta, int c0, int c1)
{
    if (c1 <= c0) return SWE2D_OK;
    SweStageArgs a;
    fill_stage_args(h, a, in, u0, out, a0, a1, beta, c0, c1);
    const bool has_u0 = (a0 != 0.0);
    // triangles: cell integral and interior facets of the viscosity inside the stage kernel, boundary facets by a small
    // launch over the boundary cells
    const bool fused_visc = h->visc && h->fuse_visc && h->npc == 3 && !h->wd && h->opp4;
    // Boundary-inline variant of the triangle kernel (BINL, swe2d_kernels.h): faster than or equal to the epilogue variant at
    // every size (us/step, same box, production numbering: 125 k cells 27.9 -> 26.3, 250 k 41.0 -> 38.2, 500 k 66.1 -> 63.6,
    // 1 M 117.9 -> 117.8); both give the same bits.  THETIS_AMD_BND_INLINE=0 selects the epilogue variant (parity test, A/B).
    const char *env_binl_s = std::getenv("THETIS_AMD_BND_INLINE");       // read per launch: tests switch it inside one process
    const bool binl = !(env_binl_s && std::atoi(env_binl_s) == 0);
    // ... and for launches whose state no longer fits the Infinity Cache (three buffers of 24 B per node against 256 MB: beyond
    // ~1.24 M triangles) with the in-wave neighbour traces exchanged through LDS (LDSX; THETIS_AMD_LDSX=0/1 forces the choice).
    // With the device's tile-Hilbert numbering and the alternating direction below, same box, us/step without / with:
    // 1 M cells 115-118 / 118-119, 1.25 M 166-168 / 162-163, 1.5 M 206-210 / 197-199, 2 M 288-294 / 275, 2.5 M 363-364 / 345-347,
    // 3 M 420-428 / 394-398, 4 M 573-583 / 544 (profiles/r04w_ldsx_threshold.txt; rounds 2-3 took it from 3 M cells only).
    // Same bits in every variant: the kernel has no implicit contraction.
    const bool beyond_cache = (size_t)(c1 - c0)*h->npc*72 >= ((size_t)256 << 20);
    const char *env_ldsx_s = std::getenv("THETIS_AMD_LDSX");
    const bool ldsx = env_ldsx_s ? std::atoi(env_ldsx_s) != 0 : beyond_cache;
    stage_kernel_t kern = fused_visc
        ? pick_kernel_visc(h->par.use_nonlinear_equations != 0, h->par.use_lax_friedrichs_velocity != 0, has_u0, has_sources(h))
        : h->wd ? pick_kernel_wd(h->par.use_lax_friedrichs_velocity != 0, has_u0, has_sources(h), h->npc == 4 ? (h->affine ? 1 : 2) : 0, binl)
        : (h->npc == 4)
        ? pick_kernel_quad(h->par.use_nonlinear_equations != 0, h->par.use_lax_friedrichs_velocity != 0, has_u0, has_sources(h), h->affine)
        : pick_kernel(h->par.use_nonlinear_equations != 0, h->par.use_lax_friedrichs_velocity != 0, has_u0, has_sources(h), binl ? (ldsx ? 2 : 1) : 0);
    const int nblocks = (c1 - c0 + SWE_BLOCK - 1)/SWE_BLOCK;
    // the XCD-chunked block map needs a grid that is a multiple of 8; surplus blocks exit immediately
    const int grid = ((nblocks + 7)/8)*8;
    // Launches whose state no longer fits the Infinity Cache (three buffers of 24 B per node against 256 MB: beyond ~1.2 M
    // triangles / 0.9 M quadrilaterals) alternate the direction in which they walk the range: the cells a launch touched last -
    // still in the cache - are the first the next one reads.  Same results (cells are independent).  Same box, fraction of the
    // 8 TB/s roofline without / with: 2 M triangles 0.530 / 0.588, 4 M 0.581 / 0.602, 8 M 0.582 / 0.598 (1 M, which fits: 0.728 /
    // 0.732).  THETIS_AMD_ALTERNATE=0/1 forces the choice.
    if (!fused_visc) {
        const char *env_alt = std::getenv("THETIS_AMD_ALTERNATE");
        const bool alt = env_alt ? std::atoi(env_alt) != 0 : beyond_cache;
        if (alt) { a.reverse = h->launch_parity; h->launch_parity ^= 1; }
    }
    SWE_CHK_SYNC(h->stream);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SWE_BLOCK), 0, h->stream, a);
    HIP_TRY(h, hipGetLastError());
    if (h->visc) {
        // HorizontalViscosityTerm: U_out[uv] += beta*dt*M^-1 R_visc(U_in) on the same cells (swe2d_sipg.h)
        SweSipgArgs v{};
        v.in = h->state[in];
        v.out = h->state[out];
        v.stride = h->stride;
        v.nbr = h->nbr; v.cv = h->cv; v.vx = h->vx; v.vy = h->vy; v.vh = h->vh;
        v.mu_v = h->nu_v; v.mu_const = h->nu_const;
        v.sipg = (h->npc == 4 ? 4.0 : 3.0)*h->sipg_factor;              // sipg_factor * cp
        v.dt = h->par.dt; v.beta = beta;
        v.cell_begin = c0; v.cell_end = c1;
        v.grad_div = h->visc_grad_div; v.grad_depth = h->visc_grad_depth;
        v.nonlin = h->par.use_nonlinear_equations;
        v.wd = h->wd ? 1 : 0;
        v.valpha = h->valpha;
        v.eta = h->state[in] + (size_t)2*h->npc*h->stride;
        v.bc = h->bc;
        v.bc_elev_f = h->bc_field[0]; v.bc_uv_f = h->bc_field[1]; v.bc_un_f = h->bc_field[2]; v.bc_flux_f = h->bc_field[3];
        if (fused_visc) {
            // Dirichlet terms exist only where the boundary dict defines an external velocity (un / uv / flux)
            bool any = false;
            for (int m = 0; m < SWE_MAX_MARKERS; m++) any = any || (h->bc.kind[m] & (SWE_BC_UN | SWE_BC_UV | SWE_BC_FLUX)) != 0;
            v.cell_list = h->bnd_cells; v.n_list = h->n_bnd;
            if (any && h->n_bnd > 0)
                hipLaunchKernelGGL((swe_sipg_kernel<2, true>), dim3((h->n_bnd + SWE_BLOCK - 1)/SWE_BLOCK), dim3(SWE_BLOCK), 0,
                                   h->stream, v);
        } else if (h->npc == 4 && !h->affine) hipLaunchKernelGGL((swe_sipg_kernel_quad<2, false>), dim3(grid), dim3(SWE_BLOCK), 0, h->stream, v);
        else if (h->npc == 4) hipLaunchKernelGGL(swe_sipg_kernel_quad<2>, dim3(grid), dim3(SWE_BLOCK), 0, h->stream, v);
        else hipLaunchKernelGGL(swe_sipg_kernel<2>, dim3(grid), dim3(SWE_BLOCK), 0, h->stream, v);
        HIP_TRY(h, hipGetLastError());
        if (h->wd && !(a0 == 0.0 && a1 == 0.0)) {
            // wetting-drying: the dry-ground relaxation acts on the whole new velocity, viscous share included (not for the
            // tendency hook, a0 = a1 = 0)
            if (h->npc == 4)
                hipLaunchKernelGGL(swe_wd_relax_kernel<4>, dim3((c1 - c0 + 255)/256), dim3(256), 0, h->stream, h->state[out], h->stride,
                                   h->cv, h->vh, h->valpha, h->par.g_grav, beta*h->par.dt, c0, c1);
            else
                hipLaunchKernelGGL(swe_wd_relax_kernel<3>, dim3((c1 - c0 + 255)/256), dim3(256), 0, h->stream, h->state[out], h->stride,
                                   h->cv, h->vh, h->valpha, h->par.g_grav, beta*h->par.dt, c0, c1);
            HIP_TRY(h, hipGetLastError());
        }
    }
    return SWE2D_OK;
}

// stage i of the Shu-Osher SSPRK33 (rungekutta.py:930-946): buffers A -> B -> C -> A
int stage_on_range(Handle *h, int i_stage, int c0, int c1)
{
    switch (i_stage) {
    case 0: return launch_stage(h, 0, 0, 1, 0.0, 1.0, kBeta[0], c0, c1);          // U1 = k + U0 (U0 is the input)
    case 1: return launch_stage(h, 1, 0, 2, kAlpha0[1], kAlphaIn[1], kBeta[1], c0, c1);
    case 2: return launch_stage(h, 2, 0, 0, kAlpha0[2], kAlphaIn[2], kBeta[2], c0, c1);
    default: return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "i_stage must be 0, 1 or 2");
    }
}

int grid_for(int n) { return (n + 255)/256; }

// ---- dataflow stage loop (swe2d_flow.h): host tables
// The kernel's 64-cell blocks are consecutive positions of a FLOW ORDER of the cells (default: the device numbering; a
// partition passes an order in which its ghost layers - appended layer by layer to the numbering the stage ranges need - sit
// next to the owned cells they touch, swe2d_flow_set_order).  Rim facets = interior facets whose two cells sit in different
// blocks.  A block's exchange slots are contiguous and grouped by the block they face, so the chunk block A writes for block B
// is contiguous and B reads it coalesced.
int flow_build(Handle *h, const int32_t *order)
{
    const int n = h->n_cells;
    const size_t S = h->stride;
    if (h->npc != 3 || h->h_nbr.empty()) return SWE2D_OK;
    const int *nbr = h->h_nbr.data();
    const int nb = (n + SWE_BLOCK - 1)/SWE_BLOCK;
    std::vector<int> fcell((size_t)nb*SWE_BLOCK, -1), fpos((size_t)n, -1);
    for (int pp = 0; pp < n; pp++) {
        const int c = order ? order[pp] : pp;
        if (c < 0 || c >= n || fpos[c] >= 0) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "flow order: not a permutation of the cells");
        fcell[pp] = c;
        fpos[c] = pp;
    }
    for (int pp = n; pp < nb*SWE_BLOCK; pp++) fcell[pp] = -1 - fcell[n - 1];          // padding lanes mimic the last cell
    struct Rim { int nbblock, pos, f; };
    std::vector<int> own((size_t)3*nb*SWE_BLOCK, -1);                                // global slot of (position, f)
    std::vector<int2> blk((size_t)nb, int2{0, 0});
    int n_slots = 0;
    bool too_many = false;
    int max_rim = 0;
    std::vector<Rim> rim;
    for (int b = 0; b < nb; b++) {
        rim.clear();
        for (int pp = b*SWE_BLOCK; pp < std::min(n, (b + 1)*SWE_BLOCK); pp++)
            for (int f = 0; f < 3; f++) {
                const int code = nbr[(size_t)f*S + fcell[pp]];
                if (code >= 0 && fpos[code >> 2]/SWE_BLOCK != b) rim.push_back(Rim{fpos[code >> 2]/SWE_BLOCK, pp, f});
            }
        std::sort(rim.begin(), rim.end(), [](const Rim &x, const Rim &y) {
            return x.nbblock != y.nbblock ? x.nbblock < y.nbblock : (x.pos != y.pos ? x.pos < y.pos : x.f < y.f); });
        if ((int)rim.size() > SWE_FLOW_MAX_RIM) too_many = true;      // the kernel's staging area holds SWE_FLOW_MAX_RIM facets
        max_rim = std::max(max_rim, (int)rim.size());
        blk[b] = int2{n_slots, (int)rim.size()};
        for (const Rim &r : rim) own[(size_t)3*r.pos + r.f] = n_slots++;
    }
    // incoming list of a block: the slots its neighbours write for it, neighbour by neighbour in THEIR slot order;
    // entry = producer's slot << 6 | lane of the consuming cell; xin(position, f) = place of the slot facing (position, f)
    std::vector<int> xsrc((size_t)std::max(n_slots, 1), 0), xin((size_t)3*nb*SWE_BLOCK, -1);
    std::vector<std::pair<int, int>> inc;                                            // (producer's slot, consumer position*4 + f)
    for (int b = 0; b < nb; b++) {
        inc.clear();
        for (int pp = b*SWE_BLOCK; pp < std::min(n, (b + 1)*SWE_BLOCK); pp++)
            for (int f = 0; f < 3; f++)
                if (own[(size_t)3*pp + f] >= 0) {
                    const int code = nbr[(size_t)f*S + fcell[pp]];
                    inc.push_back({own[(size_t)3*fpos[code >> 2] + (code & 3)], (pp << 2) | f});
                }
        std::sort(inc.begin(), inc.end());                                           // by producer's slot = by neighbour block, then its order
        for (size_t i = 0; i < inc.size(); i++) {
            const int pp = inc[i].second >> 2, f = inc[i].second & 3;
            xsrc[(size_t)blk[b].x + i] = (inc[i].first << 6) | (pp & (SWE_BLOCK - 1));
            xin[(size_t)3*pp + f] = (int)i;
        }
    }
    // per position: {my slot of facet 0, 1, 2 counted from the block's first (-1: not a rim facet), w}, {w, w} with w = place of the
    // incoming slot (rim facet) or the lane of the neighbour inside the block (this lane itself for a boundary facet)
    std::vector<int4> p4((size_t)nb*SWE_BLOCK, int4{-1, -1, -1, 0});
    std::vector<int2> p2((size_t)nb*SWE_BLOCK, int2{0, 0});
    for (int pp = 0; pp < nb*SWE_BLOCK; pp++) {
        if (pp >= n) { p4[pp] = int4{-1, -1, -1, pp & (SWE_BLOCK - 1)}; p2[pp] = int2{pp & (SWE_BLOCK - 1), pp & (SWE_BLOCK - 1)}; continue; }
        const int b0 = blk[pp/SWE_BLOCK].x;
        int lo[3], w[3];
        for (int f = 0; f < 3; f++) {
            const int code = nbr[(size_t)f*S + fcell[pp]];
            const int o = own[(size_t)3*pp + f];
            lo[f] = o >= 0 ? o - b0 : -1;
            w[f] = o >= 0 ? xin[(size_t)3*pp + f] : (code >= 0 ? (fpos[code >> 2] & (SWE_BLOCK - 1)) : (pp & (SWE_BLOCK - 1)));
        }
        p4[pp] = int4{lo[0], lo[1], lo[2], w[0]};
        p2[pp] = int2{w[1], w[2]};
    }
    // (slot << 6 must fit an int, the exchange array must stay below SWE_FLOW_NOWHERE)
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (void *ptr : {(void *)h->flow_xblk, (void *)h->flow_xsrc, (void *)h->flow_xo4, (void *)h->flow_xo2, (void *)h->flow_ex, (void *)h->flow_cell})
        if (ptr) (void)hipFree(ptr);
    h->flow_xblk = nullptr; h->flow_xsrc = nullptr; h->flow_xo4 = nullptr; h->flow_xo2 = nullptr; h->flow_ex = nullptr; h->flow_cell = nullptr;
    // no flow kernel for this handle / this order: a block with more rim facets than the staging area holds (cells numbered without
    // locality), or slot numbers that do not fit
    if (too_many || !((size_t)3*n_slots*SWE_FLOW_SLOT_BYTES < ((size_t)1 << 31) && n_slots < (1 << 25))) return SWE2D_OK;
    h->flow_fpos = fpos;
    h->flow_max_rim = max_rim;
    if (const char *e = std::getenv("THETIS_AMD_FLOW_POLL")) h->flow_max_rim = std::atoi(e) > 8 ? 65 : std::min(max_rim, 64);   // A/B, tests
    h->flow_x_ready = false;
    h->flow_parity_bytes = (unsigned)((size_t)std::max(n_slots, 1)*SWE_FLOW_SLOT_BYTES);
    h->flow_ex_bytes = (size_t)3*h->flow_parity_bytes;
    HIP_TRY(h, hipMalloc(&h->flow_xblk, blk.size()*sizeof(int2)));
    HIP_TRY(h, hipMemcpy(h->flow_xblk, blk.data(), blk.size()*sizeof(int2), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMalloc(&h->flow_xsrc, xsrc.size()*sizeof(int)));
    HIP_TRY(h, hipMemcpy(h->flow_xsrc, xsrc.data(), xsrc.size()*sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMalloc(&h->flow_cell, fcell.size()*sizeof(int)));
    HIP_TRY(h, hipMemcpy(h->flow_cell, fcell.data(), fcell.size()*sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMalloc(&h->flow_xo4, p4.size()*sizeof(int4)));
    HIP_TRY(h, hipMalloc(&h->flow_xo2, p2.size()*sizeof(int2)));
    HIP_TRY(h, hipMemcpy(h->flow_xo4, p4.data(), p4.size()*sizeof(int4), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->flow_xo2, p2.data(), p2.size()*sizeof(int2), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMalloc(&h->flow_ex, h->flow_ex_bytes));
    HIP_TRY(h, hipMemset(h->flow_ex, 0, h->flow_ex_bytes));
    // the stage counters restart with the slots
    HIP_TRY(h, hipMemset(h->flow_flag, 0, (size_t)h->flow_blocks*SWE_FLOW_FLAG_STRIDE*sizeof(unsigned)));
    return SWE2D_OK;
}

// ---- dataflow stage loop (swe2d_flow.h)
typedef void (*flow_kernel_t)(const SweFlowArgs);
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
// wide: some block of the flow order has more than 64 rim facets (one more granule load per lane and polling trip)
flow_kernel_t pick_flow_kernel(bool nl, bool lf, bool src, bool fx = false, bool wide = false)
{
    return wide ? pick_flow_poll<9>(nl, lf, src, fx) : pick_flow_poll<8>(nl, lf, src, fx);
}

// the configurations the flow kernel covers (the step kernel's: triangles, no wetting-drying, no viscosity)
bool flow_kernel_covers(const Handle *h)
{
    const char *e = std::getenv("THETIS_AMD_BND_INLINE");
    return h->npc == 3 && !h->wd && !h->visc && h->idx4 && h->flow_flag && h->flow_ex && !(e && std::atoi(e) == 0);
}

// Resident one-wave workgroups of the flow kernel: every block of a launch must be resident (a block waits for its
// neighbours' flags), so the grid must not exceed what the device holds at once.
int flow_capacity(Handle *h)
{
    if (h->flow_capacity >= 0) return h->flow_capacity;
    h->flow_capacity = 0;
    int per_cu = 0, dev_cus = 0;
    flow_kernel_t kern = pick_flow_kernel(true, true, true, true, true);     // the largest variant
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern), SWE_BLOCK, 0) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess) return 0;
    if (const char *e = std::getenv("THETIS_AMD_FLOW_CAPACITY")) h->flow_capacity = std::atoi(e);      // tests: force the limit
    else h->flow_capacity = per_cu*dev_cus;
    return h->flow_capacity;
}

// byte offset of channel c's slot 0 in the landing zone of a rank with n_recv halo cells (both sides compute it)
size_t p2p_channel_offset(const int *width, int c, int n_recv)
{
    size_t off = SWE_P2P_HEADER_BYTES;
    for (int i = 0; i < c; i++) off += 2*(size_t)n_recv*width[i]*sizeof(double);
    return off;
}

// FX launches: the places of every flow position's cell in the halo lists, the blocks that hold send / ghost cells
int flow_build_exchange(Handle *h)
{
    if (h->flow_x_ready) return SWE2D_OK;
    const int np = h->flow_blocks*SWE_BLOCK;
    if ((int)h->flow_fpos.size() != h->n_cells) return fail(h, SWE2D_ERR_UNSUPPORTED, "flow: no tables");
    std::vector<int2> xs((size_t)np, int2{-1, -1});
    std::vector<int> xr((size_t)np, -1);
    for (int j = 0; j < h->n_send; j++) {
        int2 &e = xs[h->flow_fpos[h->h_send[j]]];
        if (e.x < 0) e.x = j;
        else if (e.y < 0) e.y = j;
        else return fail(h, SWE2D_ERR_UNSUPPORTED, "flow with the exchange inside: a cell is sent to more than two peers");
    }
    for (int j = 0; j < h->n_recv; j++) xr[h->flow_fpos[h->h_recv[j]]] = j;
    h->flow_push_blocks = h->flow_recv_blocks = 0;
    for (int b = 0; b < h->flow_blocks; b++) {
        bool anys = false, anyr = false;
        for (int l = 0; l < SWE_BLOCK; l++) { anys = anys || xs[(size_t)b*SWE_BLOCK + l].x >= 0; anyr = anyr || xr[(size_t)b*SWE_BLOCK + l] >= 0; }
        h->flow_push_blocks += anys; h->flow_recv_blocks += anyr;
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (!h->flow_xsend) HIP_TRY(h, hipMalloc(&h->flow_xsend, (size_t)np*sizeof(int2)));
    if (!h->flow_xrecv) HIP_TRY(h, hipMalloc(&h->flow_xrecv, (size_t)np*sizeof(int)));
    if (!h->flow_xtick) {
        HIP_TRY(h, hipMalloc(&h->flow_xtick, (2*SWE_FLOW_MAX_CYCLES + 32)*sizeof(unsigned)));
        HIP_TRY(h, hipMemset(h->flow_xtick, 0, (2*SWE_FLOW_MAX_CYCLES + 32)*sizeof(unsigned)));
    }
    HIP_TRY(h, hipMemcpy(h->flow_xsend, xs.data(), (size_t)np*sizeof(int2), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->flow_xrecv, xr.data(), (size_t)np*sizeof(int), hipMemcpyHostToDevice));
    h->flow_x_ready = true;
    return SWE2D_OK;
}

// n_stages stages (a multiple of 3) on the ranges [0, cell_end[s]) in ONE launch; n_cycles > 0: n_cycles exchange cycles of
// n_stages stages each with the peer-to-peer halo exchange (channel 0) inside the launch
int launch_flow(Handle *h, int n_stages, const int32_t *cell_end, int n_cycles = 0)
{
    if (!flow_kernel_covers(h)) return fail(h, SWE2D_ERR_UNSUPPORTED, "the flow kernel covers triangles without wetting-drying and viscosity");
    const bool fx = n_cycles > 0;
    const int total = n_stages*(fx ? n_cycles : 1);
    if (n_stages <= 0 || n_stages % 3 != 0 || total > SWE_FLOW_MAX_STAGES || n_cycles > SWE_FLOW_MAX_CYCLES)
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "flow: n_stages must be a multiple of 3, at most 384 stages and 64 cycles per launch");
    for (int s = 0; s < n_stages; s++)
        if (cell_end[s] < 0 || cell_end[s] > h->n_cells || (s > 0 && cell_end[s] > cell_end[s - 1]))
            return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "flow: the stage ranges must shrink and stay inside the mesh");
    const int grid = ((h->flow_blocks + 7)/8)*8;
    if (grid > flow_capacity(h)) return fail(h, SWE2D_ERR_UNSUPPORTED, "flow: more 64-cell blocks than the device holds resident at once");
    SweFlowArgs q{};
    if (fx) {
        auto &z = h->p2p;
        const int ch = z.n_channels - 1;                     // the granule channel: the last one, nine 16-byte granules per cell
        if (!z.zone || !z.ctr || z.n_peers == 0 || z.n_from == 0 || h->n_send == 0 || h->n_recv == 0)
            return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "flow with the exchange inside: the peer-to-peer halo is not connected");
        if (ch < 0 || z.width[ch] != 18)
            return fail(h, SWE2D_ERR_UNSUPPORTED, "flow with the exchange inside: the last peer-to-peer channel must have width 18 (nine granules per cell)");
        if (int rc = flow_build_exchange(h)) return rc;
        q.n_cycles = n_cycles; q.stages_per_cycle = n_stages;
        q.xsend = h->flow_xsend; q.xrecv = h->flow_xrecv; q.xtick = h->flow_xtick;
        q.xctr = z.ctr + ch;
        q.x_n_peers = z.n_peers;
        for (int i = 0; i < z.n_peers; i++) {                // as swe2d_p2p_push
            if (i > 0 && z.off[i] < z.off[i - 1]) return fail(h, SWE2D_ERR_UNSUPPORTED, "flow with the exchange inside: send segments must be sorted by offset");
            q.x_off[i] = z.off[i];
            char *base = z.remote_base[i];
            q.x_rdata[i] = base + p2p_channel_offset(z.width, ch, z.remote_n_recv[i]) + (size_t)z.remote_off[i]*144;
            q.x_rslot[i] = (unsigned)((size_t)z.remote_n_recv[i]*144);
            // my segment ends cnt cells after its start in both slots: the resource covers slot 0 .. the end of my segment in slot 1
            q.x_rbytes[i] = q.x_rslot[i] + (unsigned)((size_t)z.cnt[i]*144);
        }
        char *mine = static_cast<char *>(z.zone);            // as swe2d_p2p_wait_unpack
        q.x_zone = mine + p2p_channel_offset(z.width, ch, h->n_recv);
        q.x_slot = (unsigned)((size_t)h->n_recv*144);
        q.x_zbytes = 2*q.x_slot;
        q.x_timeout = (unsigned long long)(z.timeout_s*1e8);
    }
    fill_stage_args(h, q.st, 0, 0, 1, 0.0, 1.0, 1.0, 0, 0);
    for (int i = 0; i < 3; i++) q.buf[i] = h->state[i];
    q.flag = h->flow_flag; q.status = h->flow_status;
    q.xo4 = h->flow_xo4; q.xo2 = h->flow_xo2; q.ex = h->flow_ex;
    q.xblk = h->flow_xblk; q.xsrc = h->flow_xsrc; q.parity_bytes = h->flow_parity_bytes;
    q.fcell = h->flow_cell;
    q.n_blocks = h->flow_blocks; q.n_stages = total;
    for (int s = 0; s < SWE_FLOW_MAX_STAGES; s++) q.cell_end[s] = s < n_stages ? cell_end[s] : 0;
    for (int s = 0; s < 3; s++) { q.a0[s] = s ? kAlpha0[s] : 0.0; q.a1[s] = s ? kAlphaIn[s] : 1.0; q.beta[s] = kBeta[s]; }
    q.timeout_ticks = (unsigned long long)(h->flow_timeout_s*1e8);
    flow_kernel_t kern = pick_flow_kernel(h->par.use_nonlinear_equations != 0, h->par.use_lax_friedrichs_velocity != 0, has_sources(h), fx,
                                          h->flow_max_rim > 64);
    SWE_CHK_SYNC(h->stream);
    // Every block of a flow launch must be resident at once, and flow_capacity counts the whole device: two flow launches of
    // DIFFERENT handles (streams) of this process on one device could each get a part of it and wait for their missing blocks
    // until the timeout.  Launches that do not exchange with a peer are therefore chained per device: a launch waits for the
    // previous flow launch of another handle (an event wait on the stream, no host synchronisation).  FX launches are left alone
    // (peers inside one process must run side by side; across processes DistributedSwe2d does not choose the flow path by itself
    // when ranks share a device), and so are launches under stream capture (one handle per graph).
    bool chained = false;
    if (!fx) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const bool capturing = h->stream && hipStreamIsCapturing(h->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
        chained = !capturing && h->device >= 0 && h->device < kFlowChainDevices;
    }
    std::unique_lock<std::mutex> lock(g_flow_chain_mu, std::defer_lock);
    if (chained) {
        lock.lock();
        FlowChain &fc = g_flow_chain[h->device];
        if (fc.ev && fc.last_uid != h->uid) HIP_TRY(h, hipStreamWaitEvent(h->stream, fc.ev, 0));
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SWE_BLOCK), 0, h->stream, q);
    HIP_TRY(h, hipGetLastError());
    if (chained) {
        FlowChain &fc = g_flow_chain[h->device];
        if (!fc.ev) HIP_TRY(h, hipEventCreateWithFlags(&fc.ev, hipEventDisableTiming));
        HIP_TRY(h, hipEventRecord(fc.ev, h->stream));
        fc.last_uid = h->uid;
    }
    h->flow_used = true;
    return SWE2D_OK;
}

// after a synchronisation of the stream: did a wave of a flow launch give up waiting?  (then the state is wrong)
int flow_check(Handle *h)
{
    if (!h->flow_used || !h->flow_status) return SWE2D_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (h->stream && hipStreamIsCapturing(h->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return SWE2D_OK;
    unsigned st[2] = {0u, 0u};
    HIP_TRY(h, hipMemcpyAsync(st, h->flow_status, sizeof(st), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->flow_used = false;
    if (st[0] == 0u) return SWE2D_OK;
    // leave the handle usable: counters and flags back to a consistent start
    (void)hipMemsetAsync(h->flow_status, 0, 4*sizeof(unsigned), h->stream);
    (void)hipMemsetAsync(h->flow_flag, 0, (size_t)h->flow_blocks*SWE_FLOW_FLAG_STRIDE*sizeof(unsigned), h->stream);
    (void)hipMemsetAsync(h->flow_ex, 0, h->flow_ex_bytes, h->stream);
    (void)hipStreamSynchronize(h->stream);
    char msg[200];
    std::snprintf(msg, sizeof(msg), "flow kernel: %u block waits timed out (first: block %u) - blocks not resident together? The state is invalid",
                  st[0], st[1] - 1u);
    return fail(h, SWE2D_ERR_HIP, msg);
}

}  // namespace

extern "C" {

int swe2d_abi_version(void) { return SWE2D_ABI_VERSION; }

int swe2d_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return SWE2D_ERR_NO_DEVICE;
    return n;
}

void swe2d_ssprk33_coefficients(double alpha0[3], double alpha_in[3], double beta[3])
{
    // as used by stage_on_range: stage 0 reads U0 as its input, so its U0 weight is carried by alpha_in
    const double a0[3] = {0.0, kAlpha0[1], kAlpha0[2]};
    const double ai[3] = {kAlpha0[0], kAlphaIn[1], kAlphaIn[2]};
    for (int i = 0; i < 3; i++) { alpha0[i] = a0[i]; alpha_in[i] = ai[i]; beta[i] = kBeta[i]; }
}

const char *swe2d_last_error(const swe2d_handle *h)
{
    return h ? H(h)->err.c_str() : g_create_error.c_str();
}

int swe2d_create(const swe2d_mesh *mesh, const swe2d_params *params, swe2d_handle **out)
{
    if (!mesh || !params || !out) return fail(nullptr, SWE2D_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (mesh->nodes_per_cell != 3 && mesh->nodes_per_cell != 4)
        return fail(nullptr, SWE2D_ERR_UNSUPPORTED, "nodes_per_cell must be 3 (triangles) or 4 (quadrilaterals)");
    if (mesh->n_cells <= 0 || mesh->n_owned <= 0 || mesh->n_owned > mesh->n_cells || mesh->n_vertices <= 0)
        return fail(nullptr, SWE2D_ERR_INVALID_ARGUMENT, "bad mesh sizes");
    if (mesh->n_cells >= (1 << 29))
        return fail(nullptr, SWE2D_ERR_UNSUPPORTED, "more than 2^29 cells per device");
    if (!mesh->cell_vertices || !mesh->vertex_xy || !mesh->cell_neighbours || !mesh->cell_neighbour_facets
        || !mesh->bathymetry)
        return fail(nullptr, SWE2D_ERR_INVALID_ARGUMENT, "null mesh array");
    if (!(params->dt > 0.0) || !(params->g_grav > 0.0))
        return fail(nullptr, SWE2D_ERR_INVALID_ARGUMENT, "dt and g_grav must be positive");
    // the stage kernels address a group of nodes_per_cell planes through one 4 GiB raw buffer resource (32-bit offsets)
    if ((((size_t)mesh->n_cells + 255)/256*256)*(size_t)mesh->nodes_per_cell*sizeof(double) >= ((size_t)1 << 32)
        || (size_t)mesh->n_vertices*sizeof(double) >= ((size_t)1 << 32))
        return fail(nullptr, SWE2D_ERR_UNSUPPORTED,
                    "mesh too large for one device: nodes_per_cell*n_cells*8 bytes must stay below 4 GiB (partition it)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, SWE2D_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (params->device_id < 0 || params->device_id >= ndev)
        return fail(nullptr, SWE2D_ERR_INVALID_ARGUMENT, "device_id out of range");

    Handle *h = new (std::nothrow) Handle();
    if (!h) return fail(nullptr, SWE2D_ERR_INVALID_ARGUMENT, "out of host memory");
    h->device = params->device_id;
    h->par = *params;
    const int n = mesh->n_cells, nv = mesh->n_vertices;
    const int npc = mesh->nodes_per_cell;
    h->npc = npc;
    h->n_cells = n; h->n_owned = mesh->n_owned; h->n_interior = mesh->n_owned; h->n_vertices = nv;
    h->stride = ((size_t)n + 255)/256*256;

#define HIP_TRY_C(expr)                                                                                  \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            int rc_ = fail(nullptr, SWE2D_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
            swe2d_destroy(reinterpret_cast<swe2d_handle *>(h));                                          \
            return rc_;                                                                                  \
        }                                                                                                \
    } while (0)

    HIP_TRY_C(hipSetDevice(h->device));
    HIP_TRY_C(hipStreamCreateWithFlags(&h->my_stream, hipStreamNonBlocking));
    h->stream = h->my_stream; h->own_stream = true;
    HIP_TRY_C(hipEventCreate(&h->ev0));
    HIP_TRY_C(hipEventCreate(&h->ev1));
    const size_t S = h->stride;
    for (int b = 0; b < 3; b++) {
        HIP_TRY_C(hipMalloc(&h->state[b], 3*(size_t)npc*S*sizeof(double)));
        HIP_TRY_C(hipMemsetAsync(h->state[b], 0, 3*(size_t)npc*S*sizeof(double), h->stream));
    }
    HIP_TRY_C(hipMalloc(&h->nbr, (size_t)npc*S*sizeof(int)));
    HIP_TRY_C(hipMalloc(&h->cv, (size_t)npc*S*sizeof(int)));
    HIP_TRY_C(hipMalloc(&h->vx, (size_t)nv*sizeof(double)));
    HIP_TRY_C(hipMalloc(&h->vy, (size_t)nv*sizeof(double)));
    HIP_TRY_C(hipMalloc(&h->vh, (size_t)nv*sizeof(double)));
    HIP_TRY_C(hipMalloc(&h->stage_uv, 2*(size_t)npc*n*sizeof(double)));
    HIP_TRY_C(hipMalloc(&h->stage_eta, (size_t)npc*n*sizeof(double)));
    h->n_partial_blocks = (h->n_owned + SWE_BLOCK - 1)/SWE_BLOCK;
    HIP_TRY_C(hipMalloc(&h->partial, 4*(size_t)h->n_partial_blocks*sizeof(double)));
    HIP_TRY_C(hipMalloc(&h->diag_acc, SWE_DIAG_BUCKETS*SWE_DIAG_ACC*sizeof(unsigned long long)));

    // connectivity -> SoA planes, validated on the way
    std::vector<int> nbr((size_t)npc*S, 0), cv((size_t)npc*S, 0);
    std::vector<double> vx(nv), vy(nv), vh(nv);
    double blen[SWE2D_MAX_MARKERS];
    for (int m = 0; m < SWE2D_MAX_MARKERS; m++) blen[m] = 0.0;
    for (int i = 0; i < nv; i++) {
        vx[i] = mesh->vertex_xy[2*(size_t)i];
        vy[i] = mesh->vertex_xy[2*(size_t)i + 1];
        vh[i] = mesh->bathymetry[i];
    }
    h->host_cells.assign(mesh->cell_vertices, mesh->cell_vertices + (size_t)npc*n);
    h->host_nbr.assign(mesh->cell_neighbours, mesh->cell_neighbours + (size_t)npc*n);
    for (int k = 0; k < n; k++) {
        for (int f = 0; f < npc; f++) {
            const int vid = mesh->cell_vertices[(size_t)npc*k + f];
            if (vid < 0 || vid >= nv) {
                swe2d_destroy(reinterpret_cast<swe2d_handle *>(h));
                return fail(nullptr, SWE2D_ERR_INVALID_ARGUMENT, "cell_vertices entry out of range");
            }
            cv[(size_t)f*S + k] = vid;
            const int nb = mesh->cell_neighbours[(size_t)npc*k + f];
            int packed;
            if (nb >= 0) {
                const int f2 = mesh->cell_neighbour_facets[(size_t)npc*k + f];
                if (nb >= n || f2 < 0 || f2 >= npc) {
                    swe2d_destroy(reinterpret_cast<swe2d_handle *>(h));
                    return fail(nullptr, SWE2D_ERR_INVALID_ARGUMENT, "cell_neighbours entry out of range");
                }
                packed = (nb << 2) | f2;
            } else {
                const int marker = -nb;
                if (marker >= SWE2D_MAX_MARKERS && k < mesh->n_owned) {
                    swe2d_destroy(reinterpret_cast<swe2d_handle *>(h));
                    return fail(nullptr, SWE2D_ERR_UNSUPPORTED, "boundary marker >= SWE2D_MAX_MARKERS");
                }
                packed = -marker;
                if (k < mesh->n_owned && marker < SWE2D_MAX_MARKERS) {
                    const int va = mesh->cell_vertices[(size_t)npc*k + f], vb = mesh->cell_vertices[(size_t)npc*k + (f + 1) % npc];
                    blen[marker] += std::hypot(vx[vb] - vx[va], vy[vb] - vy[va]);
                }
            }
            nbr[(size_t)f*S + k] = packed;
        }
        // orientation check (and, for quadrilaterals, that the cell is a parallelogram: the kernel assumes an affine map)
        const int a = mesh->cell_vertices[(size_t)npc*k], b = mesh->cell_vertices[(size_t)npc*k + 1],
                  c = mesh->cell_vertices[(size_t)npc*k + npc - 1];
        const double area2 = (vx[b] - vx[a])*(vy[c] - vy[a]) - (vx[c] - vx[a])*(vy[b] - vy[a]);
        if (!(area2 > 0.0)) {
            swe2d_destroy(reinterpret_cast<swe2d_handle *>(h));
            return fail(nullptr, SWE2D_ERR_INVALID_ARGUMENT, "cells must be counter-clockwise with positive area");
        }
        if (npc == 4) {
            // a cell that is not a parallelogram selects the general bilinear kernels for the whole mesh (swe_quad_mass); it must
            // be convex: det J = d0 + d1 xi + d2 zeta > 0 at the four corners
            const int d = mesh->cell_vertices[4*(size_t)k + 2];
            const double sx = vx[a] - vx[b] + vx[d] - vx[c], sy = vy[a] - vy[b] + vy[d] - vy[c];
            if (std::fabs(sx) + std::fabs(sy) > 1e-9*std::sqrt(area2)) {
                h->affine = false;
                h->affine_local = false;
                const double ax = vx[b] - vx[a], ay = vy[b] - vy[a], bx = vx[c] - vx[a], by = vy[c] - vy[a];
                const double d1 = ax*sy - ay*sx, d2 = sx*by - sy*bx;
                if (!(area2 + d1 > 0.0 && area2 + d2 > 0.0 && area2 + d1 + d2 > 0.0)) {
                    swe2d_destroy(reinterpret_cast<swe2d_handle *>(h));
                    return fail(nullptr, SWE2D_ERR_INVALID_ARGUMENT, "quadrilateral cells must be convex");
                }
            }
        }
    }
    if (mesh->n_owned != mesh->n_cells && !mesh->boundary_len) {
        swe2d_destroy(reinterpret_cast<swe2d_handle *>(h));
        return fail(nullptr, SWE2D_ERR_INVALID_ARGUMENT, "boundary_len is required for a partitioned mesh");
    }
    std::memset(&h->bc, 0, sizeof(h->bc));
    for (int m = 0; m < SWE2D_MAX_MARKERS; m++) h->bc.drag[m] = -1.0;
    for (int m = 0; m < SWE2D_MAX_MARKERS; m++) h->bc.len[m] = mesh->boundary_len ? mesh->boundary_len[m] : blen[m];

    if (npc == 3) {
        h->h_nbr = nbr;
        std::vector<int4> p4((size_t)S, int4{0, 0, 0, 0});
        std::vector<int2> p2((size_t)S, int2{0, 0});
        for (int kk = 0; kk < n; kk++) {
            p4[kk] = int4{nbr[kk], nbr[S + kk], nbr[2*S + kk], cv[kk]};
            p2[kk] = int2{cv[S + kk], cv[2*S + kk]};
        }
        HIP_TRY_C(hipMalloc(&h->idx4, (size_t)S*sizeof(int4)));
        HIP_TRY_C(hipMalloc(&h->idx2, (size_t)S*sizeof(int2)));
        HIP_TRY_C(hipMemcpy(h->idx4, p4.data(), (size_t)S*sizeof(int4), hipMemcpyHostToDevice));
        HIP_TRY_C(hipMemcpy(h->idx2, p2.data(), (size_t)S*sizeof(int2), hipMemcpyHostToDevice));
        // fused viscosity: the neighbour's vertex opposite the shared facet (its node (f2 + 2) % 3) per facet, and the list
        // of cells that own a boundary facet
        std::vector<int> bnd;
        for (int kk = 0; kk < n; kk++) {
            int vo[3];
            bool on_bnd = false;
            for (int f = 0; f < 3; f++) {
                const int code = nbr[(size_t)f*S + kk];
                vo[f] = code >= 0 ? cv[(size_t)(((code & 3) + 2) % 3)*S + (code >> 2)] : cv[(size_t)f*S + kk];
                on_bnd = on_bnd || code < 0;
            }
            p4[kk] = int4{vo[0], vo[1], vo[2], 0};
            if (on_bnd) bnd.push_back(kk);
        }
        HIP_TRY_C(hipMalloc(&h->opp4, (size_t)S*sizeof(int4)));
        HIP_TRY_C(hipMemcpy(h->opp4, p4.data(), (size_t)S*sizeof(int4), hipMemcpyHostToDevice));
        h->n_bnd = (int)bnd.size();
        if (h->n_bnd) {
            HIP_TRY_C(hipMalloc(&h->bnd_cells, bnd.size()*sizeof(int)));
            HIP_TRY_C(hipMemcpy(h->bnd_cells, bnd.data(), bnd.size()*sizeof(int), hipMemcpyHostToDevice));
        }
        h->fuse_visc = std::getenv("THETIS_AMD_NO_VISC_FUSION") == nullptr;
        // dataflow stage loop (swe2d_flow.h): one stage counter per 64-cell block, the status word, and the exchange slots of
        // the rim facets - interior facets whose two cells sit in different blocks - numbered in cell order
        h->flow_blocks = (n + SWE_BLOCK - 1)/SWE_BLOCK;
        HIP_TRY_C(hipMalloc(&h->flow_flag, (size_t)h->flow_blocks*SWE_FLOW_FLAG_STRIDE*sizeof(unsigned)));
        HIP_TRY_C(hipMalloc(&h->flow_status, 4*sizeof(unsigned)));
        HIP_TRY_C(hipMemset(h->flow_flag, 0, (size_t)h->flow_blocks*SWE_FLOW_FLAG_STRIDE*sizeof(unsigned)));
        HIP_TRY_C(hipMemset(h->flow_status, 0, 4*sizeof(unsigned)));
        if (int rc = flow_build(h, nullptr)) { g_create_error = h->err; swe2d_destroy(reinterpret_cast<swe2d_handle *>(h)); return rc; }
        if (const char *e = std::getenv("THETIS_AMD_FLOW_TIMEOUT_S")) { const double t = std::atof(e); if (t > 0.0) h->flow_timeout_s = t; }
    }
    HIP_TRY_C(hipMemcpyAsync(h->nbr, nbr.data(), (size_t)npc*S*sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_TRY_C(hipMemcpyAsync(h->cv, cv.data(), (size_t)npc*S*sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_TRY_C(hipMemcpyAsync(h->vx, vx.data(), (size_t)nv*sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY_C(hipMemcpyAsync(h->vy, vy.data(), (size_t)nv*sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY_C(hipMemcpyAsync(h->vh, vh.data(), (size_t)nv*sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY_C(hipStreamSynchronize(h->stream));
#undef HIP_TRY_C
    *out = reinterpret_cast<swe2d_handle *>(h);
    return SWE2D_OK;
}

void swe2d_destroy(swe2d_handle *hh)
{
    Handle *h = H(hh);
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->my_stream) (void)hipStreamSynchronize(h->my_stream);
    for (int b = 0; b < 3; b++) if (h->state[b]) (void)hipFree(h->state[b]);
    for (int i = 0; i < SWE2D_FIELD_COUNT; i++) if (h->field[i]) (void)hipFree(h->field[i]);
    for (auto &t : h->tracers) {
        for (int b = 0; b < 3; b++) if (t.buf[b]) (void)hipFree(t.buf[b]);
        if (t.source) (void)hipFree(t.source);
        if (t.mu_v) (void)hipFree(t.mu_v);
        if (t.bc_value_f) (void)hipFree(t.bc_value_f);
        if (t.bc_vel_f) (void)hipFree(t.bc_vel_f);
    }
    void *ptrs[] = {h->nbr, h->cv, h->vx, h->vy, h->vh, h->stage_uv, h->stage_eta, h->partial, h->diag_acc, h->send_cells, h->recv_cells,
                    h->lim_v2c_off, h->lim_v2c_cell, h->lim_vbf_off, h->lim_vbf_facet, h->lim_tv, h->lim_mean,
                    h->lim_qmin, h->lim_qmax, h->valpha, h->bc_field[0], h->bc_field[1], h->bc_field[2], h->bc_field[3], h->nu_v, h->idx4, h->idx2, h->opp4, h->bnd_cells, h->flow_flag, h->flow_status, h->flow_xo4, h->flow_xo2, h->flow_ex, h->flow_xblk, h->flow_xsrc, h->flow_cell, h->flow_xsend, h->flow_xrecv, h->flow_xtick};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (auto &fl : h->facet_lists) if (fl.dev) (void)hipFree(fl.dev);
    for (void *m : h->p2p.opened) (void)hipIpcCloseMemHandle(m);
    if (h->p2p.zone) (void)hipFree(h->p2p.zone);
    if (h->p2p.ctr) (void)hipFree(h->p2p.ctr);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->my_stream) (void)hipStreamDestroy(h->my_stream);
    delete h;
}

int swe2d_set_stream(swe2d_handle *hh, void *hip_stream)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (hip_stream) { h->stream = reinterpret_cast<hipStream_t>(hip_stream); h->own_stream = false; }
    else { h->stream = h->my_stream; h->own_stream = true; }
    return SWE2D_OK;
}

int swe2d_set_state(swe2d_handle *hh, const double *uv, const double *eta)
{
    Handle *h = H(hh);
    if (!h || !uv || !eta) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t n = (size_t)h->n_cells*h->npc;
    HIP_TRY(h, hipMemcpyAsync(h->stage_uv, uv, 2*n*sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->stage_eta, eta, n*sizeof(double), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(swe_aos_to_planes, dim3(grid_for(h->n_cells)), dim3(256), 0, h->stream,
                       h->stage_uv, h->stage_eta, h->state[0], h->stride, h->n_cells, h->npc);
    HIP_TRY(h, hipGetLastError());
    if (h->wd) {
        // explicit wetting-drying: the admissible state next to the one handed in (nodal depths through the positivity
        // limiter; enable wetting-drying BEFORE setting the state)
        if (h->npc == 4) hipLaunchKernelGGL(swe_wd_clip_kernel<4>, dim3(grid_for(h->n_cells)), dim3(256), 0, h->stream,
                                            h->state[0], h->stride, h->cv, h->vh, h->valpha, h->n_cells,
                                            h->affine ? nullptr : h->vx, h->vy);
        else hipLaunchKernelGGL(swe_wd_clip_kernel<3>, dim3(grid_for(h->n_cells)), dim3(256), 0, h->stream,
                                h->state[0], h->stride, h->cv, h->vh, h->valpha, h->n_cells, nullptr, nullptr);
        HIP_TRY(h, hipGetLastError());
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));   // host buffers may be reused by the caller
    return SWE2D_OK;
}

int swe2d_get_state(swe2d_handle *hh, double *uv, double *eta) { return swe2d_get_stage_state(hh, 2, uv, eta); }

int swe2d_get_stage_state(swe2d_handle *hh, int i_stage, double *uv, double *eta)
{
    Handle *h = H(hh);
    if (!h || !uv || !eta) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null argument");
    if (i_stage < 0 || i_stage > 2) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "i_stage must be 0, 1 or 2");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t n = (size_t)h->n_cells*h->npc;
    hipLaunchKernelGGL(swe_planes_to_aos, dim3(grid_for(h->n_cells)), dim3(256), 0, h->stream,
                       h->state[(i_stage + 1) % 3], h->stage_uv, h->stage_eta, h->stride, h->n_cells, h->npc);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(uv, h->stage_uv, 2*n*sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(eta, h->stage_eta, n*sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return flow_check(h);
}

int swe2d_set_dt(swe2d_handle *hh, double dt)
{
    Handle *h = H(hh);
    if (!h || !(dt > 0.0)) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "dt must be positive");
    h->par.dt = dt;
    return SWE2D_OK;
}

int swe2d_set_bc(swe2d_handle *hh, int marker, int kind, const double values[5])
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    if (marker <= 0 || marker >= SWE2D_MAX_MARKERS) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "marker out of range");
    if (kind & ~(SWE2D_BC_ELEV | SWE2D_BC_UV | SWE2D_BC_UN | SWE2D_BC_FLUX | SWE2D_BC_ELEV_FIELD | SWE2D_BC_UV_FIELD
                 | SWE2D_BC_UN_FIELD | SWE2D_BC_FLUX_FIELD))
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "unknown boundary kind bits");
    if (((kind & SWE2D_BC_ELEV_FIELD) && !h->bc_field[0]) || ((kind & SWE2D_BC_UV_FIELD) && !h->bc_field[1])
        || ((kind & SWE2D_BC_UN_FIELD) && !h->bc_field[2]) || ((kind & SWE2D_BC_FLUX_FIELD) && !h->bc_field[3]))
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "boundary field not set (swe2d_set_bc_field)");
    if (kind != 0 && !values) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "values required");
    h->bc.kind[marker] = kind;
    if (values) {
        h->bc.elev[marker] = values[0];
        h->bc.u[marker] = values[1];
        h->bc.v[marker] = values[2];
        h->bc.un[marker] = values[3];
        h->bc.flux[marker] = values[4];
    }
    return SWE2D_OK;
}

int swe2d_set_bc_field(swe2d_handle *hh, int which, int marker, const double *nodal)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    if (which < 0 || which > 3 || !nodal) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad boundary field");
    if (marker <= 0 || marker >= SWE2D_MAX_MARKERS) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "marker out of range");
    HIP_TRY(h, hipSetDevice(h->device));
    const int ncomp = (which == 1) ? 2 : 1;
    const size_t bytes = (size_t)2*h->npc*ncomp*h->stride*sizeof(double);
    if (!h->bc_field[which]) {
        HIP_TRY(h, hipMalloc(&h->bc_field[which], bytes));
        HIP_TRY(h, hipMemsetAsync(h->bc_field[which], 0, bytes, h->stream));
    }
    const size_t n = (size_t)h->n_cells*h->npc;
    HIP_TRY(h, hipMemcpyAsync(h->stage_uv, nodal, (size_t)ncomp*n*sizeof(double), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(swe_bc_field_scatter, dim3(grid_for(h->n_cells)), dim3(256), 0, h->stream,
                       h->stage_uv, h->bc_field[which], h->stride, h->nbr, h->n_cells, ncomp, h->npc, marker);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SWE2D_OK;
}

// shared by swe2d_set_bc_facets / swe2d_tracer_set_bc_facets: upload the compact lists and scatter them into `planes`
static int scatter_facet_values(Handle *h, double *planes, int n, const int32_t *cells, const int32_t *facets,
                                const double *values, int ncomp, int nval)
{
    if (n == 0) return SWE2D_OK;
    const size_t nv = (size_t)n*nval*ncomp;
    if (nv*sizeof(double) > (size_t)2*h->npc*h->n_cells*sizeof(double))
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "more boundary facets than the staging buffer holds");
    // a small cache of the lists seen last (one per marker and field in practice), matched by content
    int slot = -1;
    for (int i = 0; i < 8 && slot < 0; i++) {
        const Handle::FacetList &c = h->facet_lists[i];
        if ((int)c.cells.size() == n && std::memcmp(c.cells.data(), cells, (size_t)n*sizeof(int32_t)) == 0
            && std::memcmp(c.facets.data(), facets, (size_t)n*sizeof(int32_t)) == 0)
            slot = i;
    }
    const bool same = slot >= 0;
    if (!same) slot = (h->facet_list_next++) & 7;
    Handle::FacetList &fl = h->facet_lists[slot];
    if (!same) {
        for (int t = 0; t < n; t++)
            if (cells[t] < 0 || cells[t] >= h->n_cells || facets[t] < 0 || facets[t] >= h->npc)
                return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "boundary facet list: cell or facet index out of range");
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (fl.dev) { HIP_TRY(h, hipFree(fl.dev)); fl.dev = nullptr; }
        HIP_TRY(h, hipMalloc(&fl.dev, 2*(size_t)n*sizeof(int)));
        HIP_TRY(h, hipMemcpy(fl.dev, cells, (size_t)n*sizeof(int), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(fl.dev + n, facets, (size_t)n*sizeof(int), hipMemcpyHostToDevice));
        fl.cells.assign(cells, cells + n);
        fl.facets.assign(facets, facets + n);
    }
    double *dv = h->stage_uv;                           // values staged in the uv staging buffer (2*npc*n_cells doubles)
    const int *dc = fl.dev, *df = fl.dev + n;
    HIP_TRY(h, hipMemcpyAsync(dv, values, nv*sizeof(double), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(swe_bc_facet_scatter, dim3(grid_for(n)), dim3(256), 0, h->stream, dv, planes, h->stride, dc, df, n, ncomp,
                       h->npc, nval);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(h->stream));      // host buffers may be reused by the caller
    return SWE2D_OK;
}

int swe2d_set_bc_facets(swe2d_handle *hh, int which, int n_facets, const int32_t *cells, const int32_t *facets,
                        const double *values)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    if (which < 0 || which > 3 || n_facets < 0 || (n_facets > 0 && (!cells || !facets || !values)))
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad boundary facet values");
    HIP_TRY(h, hipSetDevice(h->device));
    const int ncomp = (which == 1) ? 2 : 1;
    const size_t bytes = (size_t)2*h->npc*ncomp*h->stride*sizeof(double);
    if (!h->bc_field[which]) {
        HIP_TRY(h, hipMalloc(&h->bc_field[which], bytes));
        HIP_TRY(h, hipMemsetAsync(h->bc_field[which], 0, bytes, h->stream));
    }
    return scatter_facet_values(h, h->bc_field[which], n_facets, cells, facets, values, ncomp, 2);
}

int swe2d_set_boundary_drag(swe2d_handle *hh, int marker, double drag_coefficient)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    if (marker <= 0 || marker >= SWE2D_MAX_MARKERS) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "marker out of range");
    h->bc.drag[marker] = drag_coefficient;
    return SWE2D_OK;
}

static int set_field_impl(swe2d_handle *hh, int field, const double *nodal, bool per_vertex);

int swe2d_set_field(swe2d_handle *hh, int field, const double *nodal) { return set_field_impl(hh, field, nodal, false); }

int swe2d_set_field_vertex(swe2d_handle *hh, int field, const double *vertex_values)
{
    if (!vertex_values) return fail(H(hh), SWE2D_ERR_INVALID_ARGUMENT, "null vertex values (clear a field with swe2d_set_field(h, field, NULL))");
    return set_field_impl(hh, field, vertex_values, true);
}

static int set_field_impl(swe2d_handle *hh, int field, const double *nodal, bool per_vertex)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    if (field < 0 || field >= SWE2D_FIELD_COUNT) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "unknown field id");
    HIP_TRY(h, hipSetDevice(h->device));
    const int ncomp = (field == SWE2D_FIELD_MOMENTUM_SOURCE || field == SWE2D_FIELD_WIND_STRESS) ? 2 : 1;
    if (!nodal) {
        if (h->field[field]) {
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            HIP_TRY(h, hipFree(h->field[field]));
            h->field[field] = nullptr;
        }
        return SWE2D_OK;
    }
    if (field >= SWE2D_FIELD_QUADRATIC_DRAG && field <= SWE2D_FIELD_NIKURADSE) {
        // shallowwater_eq.py:686-696: at most one of quadratic / Manning / Nikuradse (fields and scalars alike)
        for (int f2 = SWE2D_FIELD_QUADRATIC_DRAG; f2 <= SWE2D_FIELD_NIKURADSE; f2++)
            if (f2 != field && h->field[f2])
                return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "Cannot set more than one of the quadratic / Manning / Nikuradse drag parameters");
        if (h->scalar[SWE2D_SCALAR_QUADRATIC_DRAG] >= 0.0 || h->scalar[SWE2D_SCALAR_MANNING_DRAG] >= 0.0
            || h->scalar[SWE2D_SCALAR_NIKURADSE] >= 0.0)
            return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "Cannot combine a drag coefficient field with a scalar quadratic / Manning / Nikuradse parameter");
    }
    if (field == SWE2D_FIELD_LINEAR_DRAG && h->scalar[SWE2D_SCALAR_LINEAR_DRAG] >= 0.0)
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "linear drag is already set as a scalar");
    if (!h->field[field]) {
        HIP_TRY(h, hipMalloc(&h->field[field], (size_t)h->npc*ncomp*h->stride*sizeof(double)));
        HIP_TRY(h, hipMemsetAsync(h->field[field], 0, (size_t)h->npc*ncomp*h->stride*sizeof(double), h->stream));
    }
    // stage through stage_uv (2kN doubles is enough for either shape; n_vertices <= kN)
    const size_t n = (size_t)h->n_cells*h->npc;
    if (per_vertex) {
        HIP_TRY(h, hipMemcpyAsync(h->stage_uv, nodal, (size_t)ncomp*h->n_vertices*sizeof(double), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(swe_vertex_to_planes, dim3(grid_for(h->n_cells)), dim3(256), 0, h->stream,
                           h->stage_uv, h->field[field], h->stride, h->cv, h->n_cells, ncomp, h->npc);
    } else {
        HIP_TRY(h, hipMemcpyAsync(h->stage_uv, nodal, (size_t)ncomp*n*sizeof(double), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(swe_nodal_to_planes, dim3(grid_for(h->n_cells)), dim3(256), 0, h->stream,
                           h->stage_uv, h->field[field], h->stride, h->n_cells, ncomp, h->npc);
    }
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SWE2D_OK;
}

int swe2d_set_scalar(swe2d_handle *hh, int which, double value)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    if (which < 0 || which >= SWE2D_SCALAR_COUNT) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "unknown scalar id");
    if (which == SWE2D_SCALAR_NORM_SMOOTHER && value < 0.0)
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "norm_smoother must be >= 0");
    if (which == SWE2D_SCALAR_MANNING_DRAG && value >= 0.0 && h->scalar[SWE2D_SCALAR_QUADRATIC_DRAG] >= 0.0)
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "Cannot set both dimensionless and Manning drag parameter");
    if (which == SWE2D_SCALAR_QUADRATIC_DRAG && value >= 0.0 && h->scalar[SWE2D_SCALAR_MANNING_DRAG] >= 0.0)
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "Cannot set both dimensionless and Manning drag parameter");
    if (which == SWE2D_SCALAR_NIKURADSE && value >= 0.0) {
        if (h->scalar[SWE2D_SCALAR_MANNING_DRAG] >= 0.0)
            return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "Cannot set both Nikuradse drag and Manning drag parameter");
        if (h->scalar[SWE2D_SCALAR_QUADRATIC_DRAG] >= 0.0)
            return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "Cannot set both dimensionless and Nikuradse drag parameter");
        if (value == 0.0) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "nikuradse_bed_roughness must be > 0");
    }
    if ((which == SWE2D_SCALAR_MANNING_DRAG || which == SWE2D_SCALAR_QUADRATIC_DRAG || which == SWE2D_SCALAR_NIKURADSE)
        && value >= 0.0
        && (h->field[SWE2D_FIELD_QUADRATIC_DRAG] || h->field[SWE2D_FIELD_MANNING_DRAG] || h->field[SWE2D_FIELD_NIKURADSE]))
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "Cannot combine a scalar quadratic / Manning / Nikuradse parameter with a drag coefficient field");
    if (which == SWE2D_SCALAR_LINEAR_DRAG && value >= 0.0 && h->field[SWE2D_FIELD_LINEAR_DRAG])
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "linear drag is already set as a field");
    if ((which == SWE2D_SCALAR_MANNING_DRAG || which == SWE2D_SCALAR_QUADRATIC_DRAG) && value >= 0.0
        && h->scalar[SWE2D_SCALAR_NIKURADSE] >= 0.0)
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "Cannot combine the Nikuradse drag with another quadratic drag parameter");
    h->scalar[which] = value;
    return SWE2D_OK;
}

int swe2d_set_wetting_and_drying(swe2d_handle *hh, int enable, const double *alpha_vertex)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    if (!enable) { h->wd = false; return SWE2D_OK; }
    if (!alpha_vertex) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "alpha_vertex is required");
    if (!h->par.use_nonlinear_equations)
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "wetting and drying needs use_nonlinear_equations");
    for (int i = 0; i < h->n_vertices; i++)
        if (!(alpha_vertex[i] >= 0.0)) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "alpha must be >= 0");
    HIP_TRY(h, hipSetDevice(h->device));
    if (!h->valpha) HIP_TRY(h, hipMalloc(&h->valpha, (size_t)h->n_vertices*sizeof(double)));
    HIP_TRY(h, hipMemcpyAsync(h->valpha, alpha_vertex, (size_t)h->n_vertices*sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->wd = true;
    return SWE2D_OK;
}

// shared by swe2d_set_viscosity / swe2d_tracer_set_diffusivity: upload (or drop) a per-vertex coefficient
static int upload_vertex_coefficient(Handle *h, const double *vertex_values, double **dev)
{
    HIP_TRY(h, hipSetDevice(h->device));
    if (!vertex_values) {
        if (*dev) { HIP_TRY(h, hipStreamSynchronize(h->stream)); HIP_TRY(h, hipFree(*dev)); *dev = nullptr; }
        return SWE2D_OK;
    }
    for (int i = 0; i < h->n_vertices; i++)
        if (!(vertex_values[i] >= 0.0)) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "diffusion coefficient must be >= 0");
    if (!*dev) HIP_TRY(h, hipMalloc(dev, (size_t)h->n_vertices*sizeof(double)));
    HIP_TRY(h, hipMemcpyAsync(*dev, vertex_values, (size_t)h->n_vertices*sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SWE2D_OK;
}

int swe2d_set_viscosity(swe2d_handle *hh, int enable, const double *nu_vertex, double nu_const, double sipg_factor,
                        int use_grad_div_viscosity_term, int use_grad_depth_viscosity_term)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    if (!enable) { h->visc = false; return SWE2D_OK; }
    if (!nu_vertex && !(nu_const >= 0.0)) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "viscosity must be >= 0");
    if (!(sipg_factor > 0.0)) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "sipg_factor must be > 0");
    int rc = upload_vertex_coefficient(h, nu_vertex, &h->nu_v);
    if (rc) return rc;
    h->nu_const = nu_const;
    h->sipg_factor = sipg_factor;
    h->visc_grad_div = use_grad_div_viscosity_term ? 1 : 0;
    h->visc_grad_depth = use_grad_depth_viscosity_term ? 1 : 0;
    h->visc = true;
    return SWE2D_OK;
}

int swe2d_solve_stage_cells(swe2d_handle *hh, int i_stage, int32_t cell_begin, int32_t cell_end)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    if (cell_begin < 0 || cell_end > h->n_cells || cell_begin > cell_end)
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad cell range");
    HIP_TRY(h, hipSetDevice(h->device));
    static const char *names[3] = {"swe2d_solve_stage[0]", "swe2d_solve_stage[1]", "swe2d_solve_stage[2]"};
    RoctxRange range(names[(i_stage >= 0 && i_stage < 3) ? i_stage : 0]);
    return stage_on_range(h, i_stage, cell_begin, cell_end);
}

int swe2d_solve_stage(swe2d_handle *hh, int i_stage)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    return swe2d_solve_stage_cells(hh, i_stage, 0, h->n_owned);
}

int swe2d_advance(swe2d_handle *hh, int n_steps)
{
    Handle *h = H(hh);
    if (!h || n_steps < 0) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad n_steps");
    if (h->n_owned != h->n_cells)
        return fail(h, SWE2D_ERR_UNSUPPORTED, "swe2d_advance on a partition: drive stages + halo exchange from the host");
    HIP_TRY(h, hipSetDevice(h->device));
    RoctxRange range("swe2d_advance");
    // Up to 128 steps per launch without grid barriers (swe2d_flow.h) where every 64-cell block of the mesh is resident at once
    // (<= 131 k cells) and the kernel covers the configuration.  Same box, us/step, three stage launches per step -> flow launches:
    // 15 k cells 16.5 -> 15.3, 62 k 20.1 -> 15.1, 125 k 24.3 -> 18.3 (the one-launch step kernel of round 2, which this replaces:
    // 14.1 / 16.8 / 24.9).  THETIS_AMD_FLOW=0 selects the stage launches (the same bits either way).
    {
        const char *env_fl = std::getenv("THETIS_AMD_FLOW");
        const bool want = env_fl ? std::atoi(env_fl) != 0 : true;
        if (want && n_steps > 0 && flow_kernel_covers(h) && ((h->flow_blocks + 7)/8)*8 <= flow_capacity(h)) {
            int32_t ends[SWE_FLOW_MAX_STAGES];
            for (int s = 0; s < SWE_FLOW_MAX_STAGES; s++) ends[s] = h->n_owned;
            for (int done = 0; done < n_steps;) {
                const int m = std::min(n_steps - done, SWE_FLOW_MAX_STAGES/3);
                int rc = launch_flow(h, 3*m, ends);
                if (rc) return rc;
                done += m;
            }
            return SWE2D_OK;
        }
    }
    for (int it = 0; it < n_steps; it++)
        for (int s = 0; s < 3; s++) {
            int rc = stage_on_range(h, s, 0, h->n_owned);
            if (rc) return rc;
        }
    return SWE2D_OK;
}

int swe2d_solve_flow(swe2d_handle *hh, int32_t n_stages, const int32_t *cell_end)
{
    Handle *h = H(hh);
    if (!h || !cell_end) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    RoctxRange range("swe2d_solve_flow");
    return launch_flow(h, n_stages, cell_end);
}

int swe2d_solve_flow_exchange(swe2d_handle *hh, int32_t n_cycles, int32_t stages_per_cycle, const int32_t *cell_end)
{
    Handle *h = H(hh);
    if (!h || !cell_end || n_cycles < 1) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    RoctxRange range("swe2d_solve_flow_exchange");
    return launch_flow(h, stages_per_cycle, cell_end, n_cycles);
}

int swe2d_flow_unpack_pending(swe2d_handle *hh)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    auto &z = h->p2p;
    const int ch = z.n_channels - 1;
    if (!z.zone || !z.ctr || ch < 0 || z.width[ch] != 18 || !h->flow_status)
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_flow_unpack_pending: no granule channel");
    if (h->n_recv == 0) return SWE2D_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    char *mine = static_cast<char *>(z.zone);
    const unsigned slot = (unsigned)((size_t)h->n_recv*144);
    hipLaunchKernelGGL(swe_flow_unpack_kernel, dim3(std::min(256, grid_for(h->n_recv))), dim3(256), 0, h->stream, h->state[0], h->stride,
                       h->recv_cells, h->n_recv, (void *)(mine + p2p_channel_offset(z.width, ch, h->n_recv)), 2*slot, slot, z.ctr + ch,
                       h->flow_status, (unsigned long long)(z.timeout_s*1e8));
    HIP_TRY(h, hipGetLastError());
    h->flow_used = true;
    return SWE2D_OK;
}

int swe2d_flow_prepare_exchange(swe2d_handle *hh)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    return flow_build_exchange(h);
}

int swe2d_flow_set_order(swe2d_handle *hh, const int32_t *cells_in_flow_order)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    if (!h->flow_flag) return fail(h, SWE2D_ERR_UNSUPPORTED, "the flow kernel covers triangles");
    HIP_TRY(h, hipSetDevice(h->device));
    if (int rc = flow_check(h)) return rc;
    return flow_build(h, cells_in_flow_order);
}

int swe2d_flow_supported(swe2d_handle *hh)
{
    Handle *h = H(hh);
    if (!h || !flow_kernel_covers(h)) return 0;
    if (hipSetDevice(h->device) != hipSuccess) return 0;
    return ((h->flow_blocks + 7)/8)*8 <= flow_capacity(h) ? (has_sources(h) ? 1 : 2) : 0;
}

int swe2d_flow_status(swe2d_handle *hh, int32_t *timeouts)
{
    Handle *h = H(hh);
    if (!h || !timeouts) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null argument");
    *timeouts = 0;
    if (!h->flow_status) return SWE2D_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    unsigned st[2] = {0u, 0u};
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(st, h->flow_status, sizeof(st), hipMemcpyDeviceToHost));
    *timeouts = (int32_t)st[0];
    return SWE2D_OK;
}

// test hook: adds `delta` to the stage counter of one block (tests/test_gpu_flow_kernel.py: a block whose neighbours wait for it)
int swe2d_debug_flow_poke(swe2d_handle *hh, int32_t block, int32_t delta)
{
    Handle *h = H(hh);
    if (!h || !h->flow_flag || block < 0 || block >= h->flow_blocks) return SWE2D_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    unsigned v = 0;
    HIP_TRY(h, hipMemcpy(&v, h->flow_flag + (size_t)block*SWE_FLOW_FLAG_STRIDE, sizeof(v), hipMemcpyDeviceToHost));
    v += (unsigned)delta;
    HIP_TRY(h, hipMemcpy(h->flow_flag + (size_t)block*SWE_FLOW_FLAG_STRIDE, &v, sizeof(v), hipMemcpyHostToDevice));
    return SWE2D_OK;
}

// test hook of the -DSWE_FLOW_DELAY build (csrc/swe2d_flow.h): block `block` of every flow launch of this process sleeps
// `microseconds` at the points in `where` of every `every`-th stage; block < 0 switches it off.  SWE2D_ERR_UNSUPPORTED in the product build.
int swe2d_debug_flow_delay(swe2d_handle *hh, int32_t block, int32_t where, int32_t microseconds, int32_t every)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
#ifdef SWE_FLOW_DELAY
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const int cfg[4] = {block, where, microseconds*100, every < 1 ? 1 : every};
    HIP_TRY(h, hipMemcpyToSymbol(HIP_SYMBOL(swe_flow_delay), cfg, sizeof(cfg)));
    return SWE2D_OK;
#else
    (void)block; (void)where; (void)microseconds; (void)every;
    return fail(h, SWE2D_ERR_UNSUPPORTED, "swe2d_debug_flow_delay: this library was built without -DSWE_FLOW_DELAY");
#endif
}

int swe2d_advance_forward_euler(swe2d_handle *hh, int n_steps)
{
    Handle *h = H(hh);
    if (!h || n_steps < 0) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad n_steps");
    if (h->n_owned != h->n_cells) return fail(h, SWE2D_ERR_UNSUPPORTED, "ForwardEuler is not available on partitions");
    HIP_TRY(h, hipSetDevice(h->device));
    for (int it = 0; it < n_steps; it++) {
        // U_new = U + dt M^-1 R(U): stage 0 of the Shu-Osher form; the result becomes buffer A by a pointer swap
        int rc = launch_stage(h, 0, 0, 1, 0.0, 1.0, 1.0, 0, h->n_owned);
        if (rc) return rc;
        std::swap(h->state[0], h->state[1]);
    }
    return SWE2D_OK;
}

// ForwardEuler on cell ranges (partitions): U_new = U + dt M^-1 R(U) from buffer 0 into buffer 1 on [cell_begin, cell_end);
// when every range of the step is launched, swe2d_swap_state_buffers makes buffer 1 the current state.
int swe2d_forward_euler_cells(swe2d_handle *hh, int32_t cell_begin, int32_t cell_end)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    if (cell_begin < 0 || cell_end > h->n_cells || cell_begin > cell_end)
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad cell range");
    HIP_TRY(h, hipSetDevice(h->device));
    return launch_stage(h, 0, 0, 1, 0.0, 1.0, 1.0, cell_begin, cell_end);
}

int swe2d_swap_state_buffers(swe2d_handle *hh)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    std::swap(h->state[0], h->state[1]);
    return SWE2D_OK;
}

int swe2d_advance_timed(swe2d_handle *hh, int n_steps, int per_launch, float *ms_total, float *ms_kernel_avg)
{
    Handle *h = H(hh);
    if (!h || n_steps <= 0 || !ms_total) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad argument");
    if (h->n_owned != h->n_cells)
        return fail(h, SWE2D_ERR_UNSUPPORTED, "swe2d_advance_timed on a partition is not supported");
    HIP_TRY(h, hipSetDevice(h->device));
    if (!per_launch) {
        HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
        int rc = swe2d_advance(hh, n_steps);
        if (rc) return rc;
        HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
        HIP_TRY(h, hipEventSynchronize(h->ev1));
        HIP_TRY(h, hipEventElapsedTime(ms_total, h->ev0, h->ev1));
        if (ms_kernel_avg) *ms_kernel_avg = *ms_total/(3.0f*n_steps);
        return SWE2D_OK;
    }
    // events around every stage launch, on the launch stream
    const int nl = 3*n_steps;
    std::vector<hipEvent_t> ev(2*(size_t)nl);
    for (auto &e : ev) HIP_TRY(h, hipEventCreate(&e));
    HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    int l = 0;
    for (int it = 0; it < n_steps; it++)
        for (int s = 0; s < 3; s++, l++) {
            HIP_TRY(h, hipEventRecord(ev[2*l], h->stream));
            int rc = stage_on_range(h, s, 0, h->n_owned);
            if (rc) return rc;
            HIP_TRY(h, hipEventRecord(ev[2*l + 1], h->stream));
        }
    HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    HIP_TRY(h, hipEventSynchronize(h->ev1));
    HIP_TRY(h, hipEventElapsedTime(ms_total, h->ev0, h->ev1));
    double sum = 0.0;
    for (int i = 0; i < nl; i++) {
        float ms = 0.f;
        HIP_TRY(h, hipEventElapsedTime(&ms, ev[2*i], ev[2*i + 1]));
        sum += ms;
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    if (ms_kernel_avg) *ms_kernel_avg = (float)(sum/nl);
    return SWE2D_OK;
}

int swe2d_synchronize(swe2d_handle *hh)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return flow_check(h);
}

int swe2d_tendency(swe2d_handle *hh, double *k_uv, double *k_eta)
{
    Handle *h = H(hh);
    if (!h || !k_uv || !k_eta) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    // k into buffer B: U_out = 1*k + 0*U0 + 0*U_in
    int rc = launch_stage(h, 0, 0, 1, 0.0, 0.0, 1.0, 0, h->n_owned);
    if (rc) return rc;
    const size_t n = (size_t)h->n_cells*h->npc;
    hipLaunchKernelGGL(swe_planes_to_aos, dim3(grid_for(h->n_cells)), dim3(256), 0, h->stream,
                       h->state[1], h->stage_uv, h->stage_eta, h->stride, h->n_cells, h->npc);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(k_uv, h->stage_uv, 2*n*sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(k_eta, h->stage_eta, n*sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SWE2D_OK;
}

// The total of limb sums (swe_sum_accumulate) rounded to the nearest double (ties to even), exactly: a pure function of the four
// integers, so a sum taken over one device and the same sum taken over the partitions of eight agree in every bit.
double swe2d_sum_limbs_to_double(const int64_t limbs[4])
{
    // carry-normalise: 0 <= L1, L2, L3 < 2^38, L0 signed;  V = L0 2^40 + L1 2^2 + L2 2^-36 + L3 2^-74
    int64_t L[4] = {limbs[0], limbs[1], limbs[2], limbs[3]};
    for (int j = 3; j > 0; j--) {
        const int64_t carry = L[j] >> 38;                     // arithmetic shift: floor
        L[j] -= carry*((int64_t)1 << 38);
        L[j - 1] += carry;
    }
    const __int128 hi = (__int128)L[0]*((__int128)1 << 38) + L[1];                      // units of 2^2
    const unsigned __int128 lo = ((unsigned __int128)(uint64_t)L[2] << 38) | (uint64_t)L[3];   // units of 2^-74, < 2^76
    if (hi == 0) return std::ldexp((double)lo, -74);          // 76 bits -> double: correctly rounded by the conversion
    // hi != 0:  V = (hi + f) 2^2, 0 <= f = lo 2^-76 < 1.  T = floor((hi + f) 2^s) with as many bits of f as fit into 127 bits, and a
    // sticky bit for the rest: T has >= 77 significant bits, the sticky bit sits far below the rounding position
    int bits = 0;
    for (unsigned __int128 m = (unsigned __int128)(hi < 0 ? -hi : hi); m; m >>= 1) bits++;
    const int sft = std::min(76, 125 - bits);
    __int128 T = hi*((__int128)1 << sft) + (__int128)(lo >> (76 - sft));
    if (sft < 76 && (lo & (((unsigned __int128)1 << (76 - sft)) - 1)) != 0) T |= 1;
    return std::ldexp((double)T, 2 - sft);
}

namespace {
// launches the shallow water diagnostics kernel: limb sums { int eta^2, int |u|^2, int (eta+h) } + min nodal depth of the owned cells
int run_diagnostics(Handle *h, int64_t limbs[3*SWE_SUM_LIMBS], double *min_depth)
{
    HIP_TRY(h, hipSetDevice(h->device));
    SWE_CHK_SYNC(h->stream);
    HIP_TRY(h, hipMemsetAsync(h->diag_acc, 0, SWE_DIAG_BUCKETS*SWE_DIAG_ACC*sizeof(unsigned long long), h->stream));
    if (h->npc == 4)
        hipLaunchKernelGGL(swe_diag_kernel_quad, dim3(h->n_partial_blocks), dim3(SWE_BLOCK), 0, h->stream,
                           h->state[0], h->stride, h->cv, h->vx, h->vy, h->vh, h->n_owned, h->partial,
                           h->wd ? h->valpha : nullptr, h->affine ? 1 : 0, h->diag_acc);
    else
        hipLaunchKernelGGL(swe_diag_kernel, dim3(h->n_partial_blocks), dim3(SWE_BLOCK), 0, h->stream,
                           h->state[0], h->stride, h->cv, h->vx, h->vy, h->vh, h->n_owned, h->partial,
                           h->wd ? h->valpha : nullptr, h->diag_acc);
    HIP_TRY(h, hipGetLastError());
    std::vector<double> part((size_t)h->n_partial_blocks);
    unsigned long long acc[SWE_DIAG_ACC] = {0}, copies[SWE_DIAG_BUCKETS*SWE_DIAG_ACC];
    HIP_TRY(h, hipMemcpyAsync(part.data(), h->partial, part.size()*sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(copies, h->diag_acc, sizeof(copies), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int b = 0; b < SWE_DIAG_BUCKETS; b++)
        for (int i = 0; i < SWE_DIAG_ACC; i++) acc[i] += copies[b*SWE_DIAG_ACC + i];        // mod 2^64 = two's complement sums
    *min_depth = 1e300;
    for (double v : part) *min_depth = std::fmin(*min_depth, v);
    for (int i = 0; i < 3*SWE_SUM_LIMBS; i++) limbs[i] = (int64_t)acc[i];
    if (int rc = flow_check(h)) return rc;
    if (acc[3*SWE_SUM_LIMBS] != 0)
        return fail(h, SWE2D_ERR_NOT_FINITE, "state is not finite");
    return SWE2D_OK;
}
}  // namespace

int swe2d_diagnostics_limbs(swe2d_handle *hh, int64_t limbs[12], double *min_depth)
{
    Handle *h = H(hh);
    if (!h || !limbs || !min_depth) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null argument");
    return run_diagnostics(h, limbs, min_depth);
}

int swe2d_diagnostics(swe2d_handle *hh, double out[4])
{
    Handle *h = H(hh);
    if (!h || !out) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null argument");
    int64_t limbs[3*SWE_SUM_LIMBS];
    if (int rc = run_diagnostics(h, limbs, &out[3])) return rc;
    for (int q = 0; q < 3; q++) out[q] = swe2d_sum_limbs_to_double(limbs + SWE_SUM_LIMBS*q);
    return SWE2D_OK;
}

int swe2d_halo_setup(swe2d_handle *hh, int32_t n_send, const int32_t *send_cells, int32_t n_recv, const int32_t *recv_cells)
{
    Handle *h = H(hh);
    if (!h || n_send < 0 || n_recv < 0 || (n_send > 0 && !send_cells) || (n_recv > 0 && !recv_cells))
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad halo lists");
    for (int i = 0; i < n_send; i++)
        if (send_cells[i] < 0 || send_cells[i] >= h->n_owned)
            return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "send cell is not an owned cell");
    // (a handle WITHOUT ghost cells may copy between its own cells: chunks of one mesh that carry copies of their neighbours' rim,
    // tools/chunkbench.py)
    for (int i = 0; i < n_recv; i++)
        if (recv_cells[i] < (h->n_owned < h->n_cells ? h->n_owned : 0) || recv_cells[i] >= h->n_cells)
            return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "receive cell is not a ghost cell");
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->send_cells) { HIP_TRY(h, hipFree(h->send_cells)); h->send_cells = nullptr; }
    if (h->recv_cells) { HIP_TRY(h, hipFree(h->recv_cells)); h->recv_cells = nullptr; }
    h->n_send = n_send;
    h->n_recv = n_recv;
    h->h_send.assign(send_cells, send_cells + n_send);
    h->h_recv.assign(recv_cells, recv_cells + n_recv);
    h->flow_x_ready = false;
    if (n_send > 0) {
        HIP_TRY(h, hipMalloc(&h->send_cells, (size_t)n_send*sizeof(int)));
        HIP_TRY(h, hipMemcpy(h->send_cells, send_cells, (size_t)n_send*sizeof(int), hipMemcpyHostToDevice));
    }
    if (n_recv > 0) {
        HIP_TRY(h, hipMalloc(&h->recv_cells, (size_t)n_recv*sizeof(int)));
        HIP_TRY(h, hipMemcpy(h->recv_cells, recv_cells, (size_t)n_recv*sizeof(int), hipMemcpyHostToDevice));
    }
    return SWE2D_OK;
}

int swe2d_halo_pack(swe2d_handle *hh, int i_buffer, double *send_buf_dev)
{
    Handle *h = H(hh);
    if (!h || i_buffer < 0 || i_buffer > 2) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad buffer index");
    if (h->n_send == 0) return SWE2D_OK;
    if (!send_buf_dev) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null send buffer");
    HIP_TRY(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(swe_halo_pack, dim3(grid_for(3*h->npc*h->n_send)), dim3(256), 0, h->stream,
                       h->state[i_buffer], h->stride, h->send_cells, h->n_send, send_buf_dev, 3*h->npc);
    HIP_TRY(h, hipGetLastError());
    return SWE2D_OK;
}

int swe2d_halo_unpack(swe2d_handle *hh, int i_buffer, const double *recv_buf_dev)
{
    Handle *h = H(hh);
    if (!h || i_buffer < 0 || i_buffer > 2) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad buffer index");
    if (h->n_recv == 0) return SWE2D_OK;
    if (!recv_buf_dev) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null recv buffer");
    HIP_TRY(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(swe_halo_unpack, dim3(grid_for(3*h->npc*h->n_recv)), dim3(256), 0, h->stream,
                       h->state[i_buffer], h->stride, h->recv_cells, h->n_recv, recv_buf_dev, 3*h->npc);
    HIP_TRY(h, hipGetLastError());
    return SWE2D_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------
// tracers + limiter
// ------------------------------------------------------------------------------------------------------------------
namespace {

typedef void (*tracer_kernel_t)(const SweTracerArgs);

template <bool LF, bool T0>
tracer_kernel_t pick_tracer_kernel_quad_general(bool src)
{
    return src ? swe_tracer_stage_kernel_quad<LF, T0, true, false> : swe_tracer_stage_kernel_quad<LF, T0, false, false>;
}
tracer_kernel_t pick_tracer_kernel_quad(bool lf, bool t0, bool src, bool affine = true)
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

int launch_tracer_stage(Handle *h, int id, int in, int out, double a0, double a1, double beta, int c0, int c1,
                        double *mean_out = nullptr)
{
    if (c1 <= c0) return SWE2D_OK;
    Handle::Tracer &t = h->tracers[id];
    SweTracerArgs a;
    a.tin = t.buf[in];
    a.t0 = t.buf[0];
    a.tout = t.buf[out];
    a.mean_out = mean_out;
    a.uv = h->state[0];
    a.stride = h->stride;
    a.nbr = h->nbr; a.cv = h->cv; a.vx = h->vx; a.vy = h->vy;
    a.idx4 = h->idx4; a.idx2 = h->idx2;
    a.cell_begin = c0; a.cell_end = c1;
    a.dt = h->par.dt; a.a0 = a0; a.a1 = a1; a.beta = beta;
    a.vel_factor = h->tracer_vel_factor;
    a.lf_factor = h->tracer_lf_factor;
    a.source = t.source;
    a.conservative = t.conservative ? 1 : 0;
    a.depth_mode = h->wd ? 2 : (h->par.use_nonlinear_equations ? 1 : 0);
    a.vh = h->vh; a.valpha = h->valpha;
    for (int m = 0; m < SWE_MAX_MARKERS; m++) { a.bc_has_value[m] = t.bc_has_value[m]; a.bc_value[m] = t.bc_value[m]; }
    a.bc_value_f = t.bc_value_f;
    for (int m = 0; m < SWE_MAX_MARKERS; m++) { a.bc_vel_kind[m] = t.bc_vel_kind[m]; a.bc_u[m] = t.bc_u[m]; a.bc_v[m] = t.bc_v[m]; a.bc_vel_field[m] = t.bc_vel_field[m]; }
    a.bc_vel_f = t.bc_vel_f;
    for (int m = 0; m < SWE_MAX_MARKERS; m++) a.bc_len[m] = h->bc.len[m];
    // triangles: cell integral and interior facets of the diffusion inside the stage kernel, boundary facets by a launch
    // over the boundary cells (only when a marker has a diffusive boundary term at all)
    const bool fused_diff = t.diff && h->fuse_visc && h->npc == 3 && h->opp4;
    a.opp4 = h->opp4;
    a.mu_v = t.mu_v; a.mu_const = t.mu_const;
    a.diff_sipg = 3.0*t.sipg_factor;
    tracer_kernel_t kern = (h->npc == 4) ? pick_tracer_kernel_quad(h->tracer_use_lf != 0, a0 != 0.0, t.source != nullptr, h->affine)
        : fused_diff ? pick_tracer_kernel_diff(h->tracer_use_lf != 0, a0 != 0.0, t.source != nullptr)
                     : pick_tracer_kernel(h->tracer_use_lf != 0, a0 != 0.0, t.source != nullptr);
    const int nblocks = (c1 - c0 + SWE_BLOCK - 1)/SWE_BLOCK;
    const int grid = ((nblocks + 7)/8)*8;
    SWE_CHK_SYNC(h->stream);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SWE_BLOCK), 0, h->stream, a);
    HIP_TRY(h, hipGetLastError());
    if (t.diff) {
        // HorizontalDiffusionTerm: T_out += beta*dt*M^-1 R_diff(T_in) (swe2d_sipg.h)
        SweSipgArgs v{};
        v.in = t.buf[in];
        v.out = t.buf[out];
        v.stride = h->stride;
        v.nbr = h->nbr; v.cv = h->cv; v.vx = h->vx; v.vy = h->vy; v.vh = h->vh;
        v.mu_v = t.mu_v; v.mu_const = t.mu_const;
        v.sipg = (h->npc == 4 ? 4.0 : 3.0)*t.sipg_factor;
        v.dt = h->par.dt; v.beta = beta;
        v.cell_begin = c0; v.cell_end = c1;
        v.uv = h->state[0];
        v.vel_factor = h->tracer_vel_factor;
        for (int m = 0; m < SWE_MAX_MARKERS; m++) { v.bc_diff_kind[m] = t.bc_diff_kind[m]; v.bc_diff_flux[m] = t.bc_diff_flux[m]; }
        v.bc_value_f = t.bc_value_f;
        for (int m = 0; m < SWE_MAX_MARKERS; m++) { v.bc_vel_kind[m] = t.bc_vel_kind[m]; v.bc_u[m] = t.bc_u[m]; v.bc_v[m] = t.bc_v[m]; }
        v.bc = h->bc;                                   // boundary lengths ('flux' key)
        v.depth_mode = a.depth_mode; v.valpha = h->valpha;
        if (fused_diff) {
            bool any = false;
            for (int m = 0; m < SWE_MAX_MARKERS; m++) any = any || t.bc_diff_kind[m] != SWE_SIPG_BC_NONE;
            v.cell_list = h->bnd_cells; v.n_list = h->n_bnd;
            if (any && h->n_bnd > 0)
                hipLaunchKernelGGL((swe_sipg_kernel<1, true>), dim3((h->n_bnd + SWE_BLOCK - 1)/SWE_BLOCK), dim3(SWE_BLOCK), 0,
                                   h->stream, v);
        } else if (h->npc == 4 && !h->affine) hipLaunchKernelGGL((swe_sipg_kernel_quad<1, false>), dim3(grid), dim3(SWE_BLOCK), 0, h->stream, v);
        else if (h->npc == 4) hipLaunchKernelGGL(swe_sipg_kernel_quad<1>, dim3(grid), dim3(SWE_BLOCK), 0, h->stream, v);
        else hipLaunchKernelGGL(swe_sipg_kernel<1>, dim3(grid), dim3(SWE_BLOCK), 0, h->stream, v);
        HIP_TRY(h, hipGetLastError());
    }
    return SWE2D_OK;
}

int tracer_stage(Handle *h, int id, int i_stage, int c0, int c1, double *mean_out = nullptr)
{
    if (mean_out) return launch_tracer_stage(h, id, 2, 0, kAlpha0[2], kAlphaIn[2], kBeta[2], c0, c1, mean_out);
    switch (i_stage) {
    case 0: return launch_tracer_stage(h, id, 0, 1, 0.0, 1.0, kBeta[0], c0, c1);
    case 1: return launch_tracer_stage(h, id, 1, 2, kAlpha0[1], kAlphaIn[1], kBeta[1], c0, c1);
    case 2: return launch_tracer_stage(h, id, 2, 0, kAlpha0[2], kAlphaIn[2], kBeta[2], c0, c1);
    default: return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "i_stage must be 0, 1 or 2");
    }
}

int check_tracer(Handle *h, int id)
{
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    if (id < 0 || id >= (int)h->tracers.size()) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "unknown tracer id");
    return SWE2D_OK;
}

// vertex -> cells CSR and vertex -> boundary facets CSR on the host, uploaded once
int limiter_build(Handle *h, int nv, const int *topo /* [n][3] */)
{
    const int n = h->n_cells, npc = h->npc;
    const size_t S = h->stride;
    std::vector<int> off(nv + 1, 0), boff(nv + 1, 0);
    for (int k = 0; k < n; k++)
        for (int i = 0; i < npc; i++) {
            const int v = topo[(size_t)npc*k + i];
            if (v < 0 || v >= nv) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "topological vertex id out of range");
            off[v + 1]++;
            if (h->host_nbr[(size_t)npc*k + i] < 0) { boff[v + 1]++; boff[topo[(size_t)npc*k + (i + 1) % npc] + 1]++; }
        }
    for (int v = 0; v < nv; v++) { off[v + 1] += off[v]; boff[v + 1] += boff[v]; }
    std::vector<int> cell(off[nv]), bf(std::max(1, boff[nv])), pos(off.begin(), off.end() - 1), bpos(boff.begin(), boff.end() - 1);
    std::vector<int> tv((size_t)npc*S, 0);
    for (int k = 0; k < n; k++)
        for (int i = 0; i < npc; i++) {
            const int v = topo[(size_t)npc*k + i];
            cell[pos[v]++] = k;
            tv[(size_t)i*S + k] = v;
            if (h->host_nbr[(size_t)npc*k + i] < 0) {          // facet i joins local vertices i and i+1
                const int v2 = topo[(size_t)npc*k + (i + 1) % npc];
                bf[bpos[v]++] = (k << 2) | i;
                bf[bpos[v2]++] = (k << 2) | i;
            }
        }
    int **ptrs[] = {&h->lim_v2c_off, &h->lim_v2c_cell, &h->lim_vbf_off, &h->lim_vbf_facet, &h->lim_tv};
    for (int **pp : ptrs) if (*pp) { HIP_TRY(h, hipFree(*pp)); *pp = nullptr; }
    double **dptrs[] = {&h->lim_mean, &h->lim_qmin, &h->lim_qmax};
    for (double **pp : dptrs) if (*pp) { HIP_TRY(h, hipFree(*pp)); *pp = nullptr; }
    HIP_TRY(h, hipMalloc(&h->lim_v2c_off, off.size()*sizeof(int)));
    HIP_TRY(h, hipMalloc(&h->lim_v2c_cell, cell.size()*sizeof(int)));
    HIP_TRY(h, hipMalloc(&h->lim_vbf_off, boff.size()*sizeof(int)));
    HIP_TRY(h, hipMalloc(&h->lim_vbf_facet, bf.size()*sizeof(int)));
    HIP_TRY(h, hipMalloc(&h->lim_tv, tv.size()*sizeof(int)));
    HIP_TRY(h, hipMalloc(&h->lim_mean, S*sizeof(double)));
    HIP_TRY(h, hipMalloc(&h->lim_qmin, (size_t)nv*sizeof(double)));
    HIP_TRY(h, hipMalloc(&h->lim_qmax, (size_t)nv*sizeof(double)));
    HIP_TRY(h, hipMemcpy(h->lim_v2c_off, off.data(), off.size()*sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->lim_v2c_cell, cell.data(), cell.size()*sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->lim_vbf_off, boff.data(), boff.size()*sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->lim_vbf_facet, bf.data(), bf.size()*sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->lim_tv, tv.data(), tv.size()*sizeof(int), hipMemcpyHostToDevice));
    h->lim_nv = nv;
    return SWE2D_OK;
}

// Means and vertex bounds over every local cell / vertex, limited values written to cells [0, cell_end).  On a
// partition cell_end excludes the outermost ghost layer, whose vertex neighbourhoods are incomplete (partition.py).
int limiter_apply(Handle *h, int id, int cell_end, bool means_done = false)
{
    if (cell_end < 0 || cell_end > h->n_cells) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad cell range");
    if (h->lim_nv == 0) {
        int rc = limiter_build(h, h->n_vertices, h->host_cells.data());
        if (rc) return rc;
    }
    double *t = h->tracers[id].buf[0];
    const int n = h->n_cells, nv = h->lim_nv;
    if (!means_done)       // swe2d_advance_coupled has the last tracer stage write the means
        hipLaunchKernelGGL(swe_limiter_cell_mean, dim3(grid_for(n)), dim3(256), 0, h->stream, t, h->stride, n, h->lim_mean, h->npc,
                           h->affine ? nullptr : h->cv, h->vx, h->vy);
    hipLaunchKernelGGL(swe_limiter_vertex_bounds, dim3(grid_for(nv)), dim3(256), 0, h->stream, h->lim_v2c_off,
                       h->lim_v2c_cell, h->lim_vbf_off, h->lim_vbf_facet, h->lim_mean, t, h->stride, nv, h->lim_qmin,
                       h->lim_qmax, h->npc);
    if (cell_end > 0)
        hipLaunchKernelGGL(swe_limiter_apply, dim3(grid_for(cell_end)), dim3(256), 0, h->stream, t, h->stride, cell_end,
                           h->lim_tv, h->lim_qmin, h->lim_qmax, h->npc, h->affine ? nullptr : h->lim_mean);
    HIP_TRY(h, hipGetLastError());
    return SWE2D_OK;
}

}  // namespace

extern "C" {

int swe2d_tracer_add(swe2d_handle *hh, int *tracer_id)
{
    Handle *h = H(hh);
    if (!h || !tracer_id) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    Handle::Tracer t;
    for (int m = 0; m < SWE_MAX_MARKERS; m++) {
        t.bc_has_value[m] = 0; t.bc_value[m] = 0.0;
        t.bc_diff_kind[m] = SWE_SIPG_BC_NONE; t.bc_diff_flux[m] = 0.0;
        t.bc_vel_kind[m] = 0; t.bc_u[m] = 0.0; t.bc_v[m] = 0.0; t.bc_vel_field[m] = 0;
    }
    for (int b = 0; b < 3; b++) {
        HIP_TRY(h, hipMalloc(&t.buf[b], (size_t)h->npc*h->stride*sizeof(double)));
        HIP_TRY(h, hipMemsetAsync(t.buf[b], 0, (size_t)h->npc*h->stride*sizeof(double), h->stream));
    }
    h->tracers.push_back(t);
    *tracer_id = (int)h->tracers.size() - 1;
    return SWE2D_OK;
}

int swe2d_tracer_set_options(swe2d_handle *hh, int use_lax_friedrichs_tracer, double lax_friedrichs_tracer_scaling_factor,
                             double tracer_advective_velocity_factor)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    h->tracer_use_lf = use_lax_friedrichs_tracer ? 1 : 0;
    h->tracer_lf_factor = lax_friedrichs_tracer_scaling_factor;
    h->tracer_vel_factor = tracer_advective_velocity_factor;
    return SWE2D_OK;
}

int swe2d_tracer_set_state(swe2d_handle *hh, int id, const double *nodal)
{
    Handle *h = H(hh);
    int rc = check_tracer(h, id);
    if (rc) return rc;
    if (!nodal) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpyAsync(h->stage_eta, nodal, (size_t)h->npc*h->n_cells*sizeof(double), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(swe_nodal_to_planes, dim3(grid_for(h->n_cells)), dim3(256), 0, h->stream,
                       h->stage_eta, h->tracers[id].buf[0], h->stride, h->n_cells, 1, h->npc);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SWE2D_OK;
}

static int tracer_read_back(Handle *h, const double *planes, double *nodal)
{
    hipLaunchKernelGGL(swe_planes_to_nodal, dim3(grid_for(h->n_cells)), dim3(256), 0, h->stream,
                       planes, h->stage_eta, h->stride, h->n_cells, h->npc);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(nodal, h->stage_eta, (size_t)h->npc*h->n_cells*sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SWE2D_OK;
}

int swe2d_tracer_get_state(swe2d_handle *hh, int id, double *nodal)
{
    Handle *h = H(hh);
    int rc = check_tracer(h, id);
    if (rc) return rc;
    if (!nodal) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    return tracer_read_back(h, h->tracers[id].buf[0], nodal);
}

int swe2d_tracer_set_bc(swe2d_handle *hh, int id, int marker, int has_value, double value)
{
    Handle *h = H(hh);
    int rc = check_tracer(h, id);
    if (rc) return rc;
    if (marker <= 0 || marker >= SWE2D_MAX_MARKERS) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "marker out of range");
    // 0: no value, 1: constant, 2: Function values uploaded by swe2d_tracer_set_bc_field / swe2d_tracer_set_bc_facets
    h->tracers[id].bc_has_value[marker] = has_value == 2 ? 2 : (has_value ? 1 : 0);
    h->tracers[id].bc_value[marker] = value;
    return SWE2D_OK;
}

int swe2d_tracer_set_bc_velocity(swe2d_handle *hh, int id, int marker, int kind, double u, double v)
{
    Handle *h = H(hh);
    int rc = check_tracer(h, id);
    if (rc) return rc;
    if (marker <= 0 || marker >= SWE2D_MAX_MARKERS) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "marker out of range");
    if (kind < 0 || kind > 4)
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "velocity kind must be 0 (none), 1 ('uv'), 2 ('un'), 3 ('flux') or 4 ('flux' + 'elev')");
    h->tracers[id].bc_vel_kind[marker] = kind;
    h->tracers[id].bc_u[marker] = u;
    h->tracers[id].bc_v[marker] = v;
    h->tracers[id].bc_vel_field[marker] = 0;
    return SWE2D_OK;
}

int swe2d_tracer_set_bc_velocity_facets(swe2d_handle *hh, int id, int marker, int kind, double elev, int n_facets,
                                        const int32_t *cells, const int32_t *facets, const double *values)
{
    Handle *h = H(hh);
    int rc = check_tracer(h, id);
    if (rc) return rc;
    if (marker <= 0 || marker >= SWE2D_MAX_MARKERS) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "marker out of range");
    if (kind < 1 || kind > 4) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "velocity kind must be 1 ('uv'), 2 ('un'), 3 ('flux') or 4 ('flux' + 'elev')");
    if (n_facets < 0 || (n_facets > 0 && (!cells || !facets || !values)))
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad boundary facet values");
    HIP_TRY(h, hipSetDevice(h->device));
    Handle::Tracer &t = h->tracers[id];
    const size_t bytes = (size_t)4*h->npc*h->stride*sizeof(double);
    if (!t.bc_vel_f) {
        HIP_TRY(h, hipMalloc(&t.bc_vel_f, bytes));
        HIP_TRY(h, hipMemsetAsync(t.bc_vel_f, 0, bytes, h->stream));
    }
    t.bc_vel_kind[marker] = kind;
    t.bc_vel_field[marker] = 1;
    t.bc_u[marker] = 0.0;
    t.bc_v[marker] = elev;                              // constant 'elev' of a 'flux' entry (kind 4)
    return scatter_facet_values(h, t.bc_vel_f, n_facets, cells, facets, values, kind == 1 ? 2 : 1, 2);
}

int swe2d_tracer_set_bc_facets(swe2d_handle *hh, int id, int n_facets, const int32_t *cells, const int32_t *facets,
                               const double *values)
{
    Handle *h = H(hh);
    int rc = check_tracer(h, id);
    if (rc) return rc;
    if (n_facets < 0 || (n_facets > 0 && (!cells || !facets || !values)))
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad boundary facet values");
    HIP_TRY(h, hipSetDevice(h->device));
    Handle::Tracer &t = h->tracers[id];
    const size_t bytes = (size_t)h->npc*h->npc*h->stride*sizeof(double);
    if (!t.bc_value_f) {
        HIP_TRY(h, hipMalloc(&t.bc_value_f, bytes));
        HIP_TRY(h, hipMemsetAsync(t.bc_value_f, 0, bytes, h->stream));
    }
    return scatter_facet_values(h, t.bc_value_f, n_facets, cells, facets, values, 1, h->npc);
}

int swe2d_tracer_set_bc_field(swe2d_handle *hh, int id, int marker, const double *nodal)
{
    Handle *h = H(hh);
    int rc = check_tracer(h, id);
    if (rc) return rc;
    if (marker <= 0 || marker >= SWE2D_MAX_MARKERS || !nodal) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad boundary field");
    HIP_TRY(h, hipSetDevice(h->device));
    Handle::Tracer &t = h->tracers[id];
    const size_t bytes = (size_t)h->npc*h->npc*h->stride*sizeof(double);
    if (!t.bc_value_f) {
        HIP_TRY(h, hipMalloc(&t.bc_value_f, bytes));
        HIP_TRY(h, hipMemsetAsync(t.bc_value_f, 0, bytes, h->stream));
    }
    HIP_TRY(h, hipMemcpyAsync(h->stage_eta, nodal, (size_t)h->npc*h->n_cells*sizeof(double), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(swe_bc_cellfield_scatter, dim3(grid_for(h->n_cells)), dim3(256), 0, h->stream,
                       h->stage_eta, t.bc_value_f, h->stride, h->nbr, h->n_cells, h->npc, marker);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    t.bc_has_value[marker] = 2;
    return SWE2D_OK;
}

int swe2d_tracer_set_source(swe2d_handle *hh, int id, const double *nodal)
{
    Handle *h = H(hh);
    int rc = check_tracer(h, id);
    if (rc) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    Handle::Tracer &t = h->tracers[id];
    if (!nodal) {
        if (t.source) { HIP_TRY(h, hipStreamSynchronize(h->stream)); HIP_TRY(h, hipFree(t.source)); t.source = nullptr; }
        return SWE2D_OK;
    }
    if (!t.source) HIP_TRY(h, hipMalloc(&t.source, (size_t)h->npc*h->stride*sizeof(double)));
    HIP_TRY(h, hipMemcpyAsync(h->stage_eta, nodal, (size_t)h->npc*h->n_cells*sizeof(double), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(swe_nodal_to_planes, dim3(grid_for(h->n_cells)), dim3(256), 0, h->stream,
                       h->stage_eta, t.source, h->stride, h->n_cells, 1, h->npc);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SWE2D_OK;
}

int swe2d_tracer_forward_euler(swe2d_handle *hh, int id)
{
    Handle *h = H(hh);
    int rc = check_tracer(h, id);
    if (rc) return rc;
    if (h->n_owned != h->n_cells) return fail(h, SWE2D_ERR_UNSUPPORTED, "ForwardEuler is not available on partitions");
    HIP_TRY(h, hipSetDevice(h->device));
    rc = launch_tracer_stage(h, id, 0, 1, 0.0, 1.0, 1.0, 0, h->n_owned);
    if (rc) return rc;
    std::swap(h->tracers[id].buf[0], h->tracers[id].buf[1]);
    return SWE2D_OK;
}

// ForwardEuler on cell ranges (partitions): swe2d_tracer_solve_stage_cells(id, 0, ...) is the step from tracer buffer 0 into
// buffer 1; when every range of the step is launched this makes buffer 1 the tracer.
int swe2d_tracer_swap_buffers(swe2d_handle *hh, int id)
{
    Handle *h = H(hh);
    int rc = check_tracer(h, id);
    if (rc) return rc;
    std::swap(h->tracers[id].buf[0], h->tracers[id].buf[1]);
    return SWE2D_OK;
}

int swe2d_tracer_set_conservative(swe2d_handle *hh, int id, int use_conservative_form)
{
    Handle *h = H(hh);
    int rc = check_tracer(h, id);
    if (rc) return rc;
    h->tracers[id].conservative = use_conservative_form != 0;
    return SWE2D_OK;
}

int swe2d_tracer_set_diffusivity(swe2d_handle *hh, int id, int enable, const double *mu_vertex, double mu_const,
                                 double sipg_factor_tracer)
{
    Handle *h = H(hh);
    int rc = check_tracer(h, id);
    if (rc) return rc;
    Handle::Tracer &t = h->tracers[id];
    if (!enable) { t.diff = false; return SWE2D_OK; }
    if (!mu_vertex && !(mu_const >= 0.0)) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "diffusivity must be >= 0");
    if (!(sipg_factor_tracer > 0.0)) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "sipg_factor_tracer must be > 0");
    rc = upload_vertex_coefficient(h, mu_vertex, &t.mu_v);
    if (rc) return rc;
    t.mu_const = mu_const;
    t.sipg_factor = sipg_factor_tracer;
    t.diff = true;
    return SWE2D_OK;
}

int swe2d_tracer_set_diffusion_bc(swe2d_handle *hh, int id, int marker, int kind, double diff_flux)
{
    Handle *h = H(hh);
    int rc = check_tracer(h, id);
    if (rc) return rc;
    if (marker <= 0 || marker >= SWE2D_MAX_MARKERS) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "marker out of range");
    if (kind < SWE_SIPG_BC_NONE || kind > SWE_SIPG_BC_VALUE_FIELD) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad diffusion boundary kind");
    if (kind == SWE_SIPG_BC_VALUE_FIELD && !h->tracers[id].bc_value_f)
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "boundary field not set (swe2d_tracer_set_bc_field)");
    h->tracers[id].bc_diff_kind[marker] = kind;
    h->tracers[id].bc_diff_flux[marker] = diff_flux;
    return SWE2D_OK;
}

int swe2d_tracer_solve_stage(swe2d_handle *hh, int id, int i_stage)
{
    Handle *h = H(hh);
    int rc = check_tracer(h, id);
    if (rc) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    return tracer_stage(h, id, i_stage, 0, h->n_owned);
}

int swe2d_tracer_tendency(swe2d_handle *hh, int id, double *k_nodal)
{
    Handle *h = H(hh);
    int rc = check_tracer(h, id);
    if (rc) return rc;
    if (!k_nodal) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    rc = launch_tracer_stage(h, id, 0, 1, 0.0, 0.0, 1.0, 0, h->n_owned);
    if (rc) return rc;
    return tracer_read_back(h, h->tracers[id].buf[1], k_nodal);
}

int swe2d_limiter_setup(swe2d_handle *hh, int32_t n_topo_vertices, const int32_t *cell_topo_vertices)
{
    Handle *h = H(hh);
    if (!h || n_topo_vertices <= 0 || !cell_topo_vertices) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad limiter topology");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return limiter_build(h, n_topo_vertices, cell_topo_vertices);
}

int swe2d_tracer_limit(swe2d_handle *hh, int id)
{
    Handle *h = H(hh);
    int rc = check_tracer(h, id);
    if (rc) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->n_owned != h->n_cells)
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "on a partition use swe2d_tracer_limit_cells (the outermost ghost layer cannot be limited)");
    return limiter_apply(h, id, h->n_cells);
}

int swe2d_tracer_limit_cells(swe2d_handle *hh, int id, int32_t cell_end)
{
    Handle *h = H(hh);
    int rc = check_tracer(h, id);
    if (rc) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    return limiter_apply(h, id, cell_end);
}

int swe2d_tracer_solve_stage_cells(swe2d_handle *hh, int id, int i_stage, int32_t cell_begin, int32_t cell_end)
{
    Handle *h = H(hh);
    int rc = check_tracer(h, id);
    if (rc) return rc;
    if (cell_begin < 0 || cell_end > h->n_cells || cell_begin > cell_end) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad cell range");
    HIP_TRY(h, hipSetDevice(h->device));
    return tracer_stage(h, id, i_stage, cell_begin, cell_end);
}

int swe2d_tracer_halo_pack(swe2d_handle *hh, int id, int i_buffer, double *send_buf_dev)
{
    Handle *h = H(hh);
    int rc = check_tracer(h, id);
    if (rc) return rc;
    if (i_buffer < 0 || i_buffer > 2) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad buffer index");
    if (h->n_send == 0) return SWE2D_OK;
    if (!send_buf_dev) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null send buffer");
    HIP_TRY(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(swe_halo_pack, dim3(grid_for(h->npc*h->n_send)), dim3(256), 0, h->stream,
                       h->tracers[id].buf[i_buffer], h->stride, h->send_cells, h->n_send, send_buf_dev, h->npc);
    HIP_TRY(h, hipGetLastError());
    return SWE2D_OK;
}

int swe2d_tracer_halo_unpack(swe2d_handle *hh, int id, int i_buffer, const double *recv_buf_dev)
{
    Handle *h = H(hh);
    int rc = check_tracer(h, id);
    if (rc) return rc;
    if (i_buffer < 0 || i_buffer > 2) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad buffer index");
    if (h->n_recv == 0) return SWE2D_OK;
    if (!recv_buf_dev) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null recv buffer");
    HIP_TRY(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(swe_halo_unpack, dim3(grid_for(h->npc*h->n_recv)), dim3(256), 0, h->stream,
                       h->tracers[id].buf[i_buffer], h->stride, h->recv_cells, h->n_recv, recv_buf_dev, h->npc);
    HIP_TRY(h, hipGetLastError());
    return SWE2D_OK;
}

namespace {
int run_tracer_diagnostics(Handle *h, int id, int64_t limbs[2*SWE_SUM_LIMBS], double minmax[2])
{
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemsetAsync(h->diag_acc, 0, SWE_DIAG_BUCKETS*SWE_DIAG_ACC*sizeof(unsigned long long), h->stream));
    if (h->npc == 4)
        hipLaunchKernelGGL(swe_tracer_diag_kernel_quad, dim3(h->n_partial_blocks), dim3(SWE_BLOCK), 0, h->stream,
                           h->tracers[id].buf[0], h->state[0], h->stride, h->cv, h->vx, h->vy, h->vh,
                           h->par.use_nonlinear_equations, h->n_owned, h->partial, h->wd ? h->valpha : nullptr, h->affine ? 1 : 0,
                           h->diag_acc);
    else
        hipLaunchKernelGGL(swe_tracer_diag_kernel, dim3(h->n_partial_blocks), dim3(SWE_BLOCK), 0, h->stream,
                           h->tracers[id].buf[0], h->state[0], h->stride, h->cv, h->vx, h->vy, h->vh,
                           h->par.use_nonlinear_equations, h->n_owned, h->partial, h->wd ? h->valpha : nullptr, h->diag_acc);
    HIP_TRY(h, hipGetLastError());
    std::vector<double> part(2*(size_t)h->n_partial_blocks);
    unsigned long long acc[SWE_DIAG_ACC] = {0}, copies[SWE_DIAG_BUCKETS*SWE_DIAG_ACC];
    HIP_TRY(h, hipMemcpyAsync(part.data(), h->partial, part.size()*sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(copies, h->diag_acc, sizeof(copies), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int b = 0; b < SWE_DIAG_BUCKETS; b++)
        for (int i = 0; i < SWE_DIAG_ACC; i++) acc[i] += copies[b*SWE_DIAG_ACC + i];        // mod 2^64 = two's complement sums
    minmax[0] = 1e300; minmax[1] = -1e300;
    for (int b = 0; b < h->n_partial_blocks; b++) {
        minmax[0] = std::fmin(minmax[0], part[2*(size_t)b]);
        minmax[1] = std::fmax(minmax[1], part[2*(size_t)b + 1]);
    }
    for (int i = 0; i < 2*SWE_SUM_LIMBS; i++) limbs[i] = (int64_t)acc[i];
    if (acc[2*SWE_SUM_LIMBS] != 0)
        return fail(h, SWE2D_ERR_NOT_FINITE, "tracer is not finite");
    return SWE2D_OK;
}
}  // namespace

int swe2d_tracer_diagnostics_limbs(swe2d_handle *hh, int id, int64_t limbs[8], double minmax[2])
{
    Handle *h = H(hh);
    int rc = check_tracer(h, id);
    if (rc) return rc;
    if (!limbs || !minmax) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null argument");
    return run_tracer_diagnostics(h, id, limbs, minmax);
}

int swe2d_tracer_diagnostics(swe2d_handle *hh, int id, double out[4])
{
    Handle *h = H(hh);
    int rc = check_tracer(h, id);
    if (rc) return rc;
    if (!out) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null argument");
    int64_t limbs[2*SWE_SUM_LIMBS];
    if ((rc = run_tracer_diagnostics(h, id, limbs, out + 2))) return rc;
    out[0] = swe2d_sum_limbs_to_double(limbs);
    out[1] = swe2d_sum_limbs_to_double(limbs + SWE_SUM_LIMBS);
    return SWE2D_OK;
}

int swe2d_set_general_quadrilaterals(swe2d_handle *hh, int on)
{
    Handle *h = H(hh);
    if (!h) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null handle");
    if (h->npc != 4) return on ? fail(h, SWE2D_ERR_INVALID_ARGUMENT, "not a quadrilateral mesh") : SWE2D_OK;
    if (!on && !h->affine_local) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "the mesh has cells that are not parallelograms");
    h->affine = !on;
    return SWE2D_OK;
}

int swe2d_debug_calibration_copy(swe2d_handle *hh, int n_times)
{
    Handle *h = H(hh);
    if (!h || n_times < 0) return SWE2D_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t n = 3*(size_t)h->npc*h->stride;
    for (int i = 0; i < n_times; i++) {
        // buffer C is dead between steps: copying A -> C does not disturb the state
        hipLaunchKernelGGL(swe_calibration_copy, dim3((unsigned)((n + SWE_BLOCK - 1)/SWE_BLOCK)), dim3(SWE_BLOCK), 0, h->stream,
                           h->state[0], h->state[2], n);
    }
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SWE2D_OK;
}

int swe2d_advance_coupled(swe2d_handle *hh, int n_steps, int tracer_only, int use_limiter)
{
    Handle *h = H(hh);
    if (!h || n_steps < 0) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad n_steps");
    if (h->n_owned != h->n_cells) return fail(h, SWE2D_ERR_UNSUPPORTED, "on a partition the host drives the coupled step (stages on cell ranges + halo exchanges, thetis_amd/distributed.py)");
    HIP_TRY(h, hipSetDevice(h->device));
    RoctxRange range("swe2d_advance_coupled");
    for (int it = 0; it < n_steps; it++) {
        if (!tracer_only)
            for (int s = 0; s < 3; s++) { int rc = stage_on_range(h, s, 0, h->n_owned); if (rc) return rc; }
        for (int id = 0; id < (int)h->tracers.size(); id++) {
            // without a diffusion pass behind it the last stage kernel also writes the cell means the limiter starts from
            const bool fuse_mean = use_limiter && !h->tracers[id].diff;
            if (fuse_mean && h->lim_nv == 0) {
                int rc = limiter_build(h, h->n_vertices, h->host_cells.data());
                if (rc) return rc;
            }
            for (int s = 0; s < 3; s++) {
                int rc = tracer_stage(h, id, s, 0, h->n_owned, (s == 2 && fuse_mean) ? h->lim_mean : nullptr);
                if (rc) return rc;
            }
            if (use_limiter) { int rc = limiter_apply(h, id, h->n_cells, fuse_mean); if (rc) return rc; }
        }
    }
    return SWE2D_OK;
}

}  // extern "C"


// ------------------------------------------------------------------------------------------------------------------
// peer-to-peer halo exchange (swe2d_p2p.h)

extern "C" {

int swe2d_p2p_create(swe2d_handle *hh, int32_t n_channels, const int32_t *widths)
{
    Handle *h = H(hh);
    if (!h || n_channels < 1 || n_channels > SWE_P2P_MAX_CHANNELS || !widths)
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_create: 1..8 channels");
    if (h->p2p.zone) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_create: already created");
    HIP_TRY(h, hipSetDevice(h->device));
    auto &z = h->p2p;
    z.n_channels = n_channels;
    for (int c = 0; c < n_channels; c++) {
        if (widths[c] < 1) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_create: bad channel width");
        z.width[c] = widths[c];
    }
    z.zone_bytes = p2p_channel_offset(z.width, n_channels, h->n_recv);
    z.zone_bytes = (z.zone_bytes + 4095)/4096*4096;
    // remote GPUs write here and local kernels poll it: keep it out of the (non-coherent) L2 when the runtime allows
    const char *force = std::getenv("THETIS_AMD_P2P_ZONE");        // "uncached" | "finegrained" | "device" (debugging)
    const int want = !force ? 0 : !std::strcmp(force, "uncached") ? 1 : !std::strcmp(force, "finegrained") ? 2 : 3;
    z.zone_kind = 0;
    if ((want == 0 || want == 1) && hipExtMallocWithFlags(&z.zone, z.zone_bytes, hipDeviceMallocUncached) == hipSuccess) z.zone_kind = 1;
    if (!z.zone_kind) (void)hipGetLastError();
    if (!z.zone_kind && (want == 0 || want == 2)
        && hipExtMallocWithFlags(&z.zone, z.zone_bytes, hipDeviceMallocFinegrained) == hipSuccess) z.zone_kind = 2;
    if (!z.zone_kind) {
        (void)hipGetLastError();
        HIP_TRY(h, hipMalloc(&z.zone, z.zone_bytes));
        z.zone_kind = 3;
    }
    HIP_TRY(h, hipMemset(z.zone, 0, z.zone_bytes));
    HIP_TRY(h, hipMalloc(&z.ctr, n_channels*sizeof(SweP2pCounters)));
    HIP_TRY(h, hipMemset(z.ctr, 0, n_channels*sizeof(SweP2pCounters)));
    HIP_TRY(h, hipDeviceSynchronize());
    if (const char *t = std::getenv("THETIS_AMD_P2P_TIMEOUT_S")) z.timeout_s = std::atof(t);
    return SWE2D_OK;
}

int swe2d_p2p_export(swe2d_handle *hh, void *ipc_handle_out, void **local_base, int32_t *zone_kind)
{
    Handle *h = H(hh);
    if (!h || !h->p2p.zone) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_export: no landing zone");
    static_assert(sizeof(hipIpcMemHandle_t) == SWE2D_IPC_HANDLE_BYTES, "IPC handle size");
    HIP_TRY(h, hipSetDevice(h->device));
    if (ipc_handle_out) {
        hipIpcMemHandle_t mh;
        HIP_TRY(h, hipIpcGetMemHandle(&mh, h->p2p.zone));
        std::memcpy(ipc_handle_out, &mh, sizeof(mh));
    }
    if (local_base) *local_base = h->p2p.zone;
    if (zone_kind) *zone_kind = h->p2p.zone_kind;
    return SWE2D_OK;
}

int swe2d_p2p_open(swe2d_handle *hh, const void *ipc_handle, void **remote_base)
{
    Handle *h = H(hh);
    if (!h || !ipc_handle || !remote_base) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_open: null argument");
    if (std::getenv("THETIS_AMD_TEST_BREAK_P2P"))             // tests: a node whose IPC mapping does not work
        return fail(h, SWE2D_ERR_HIP, "swe2d_p2p_open: disabled by THETIS_AMD_TEST_BREAK_P2P");
    HIP_TRY(h, hipSetDevice(h->device));
    hipIpcMemHandle_t mh;
    std::memcpy(&mh, ipc_handle, sizeof(mh));
    void *p = nullptr;
    HIP_TRY(h, hipIpcOpenMemHandle(&p, mh, hipIpcMemLazyEnablePeerAccess));
    h->p2p.opened.push_back(p);
    // First contact with a peer's memory must fail with an error code, never with a memory fault inside the push kernel (which
    // would take the process down): the mapping has to be a device pointer of this process, a host-initiated copy into the
    // unused upper half of the zone header has to round-trip, and so has a store + load from a kernel of this device.
    hipPointerAttribute_t attr;
    HIP_TRY(h, hipPointerGetAttributes(&attr, p));
    unsigned long long *probe = reinterpret_cast<unsigned long long *>(static_cast<char *>(p) + SWE_P2P_HEADER_BYTES/2)
                                + (unsigned)getpid() % (SWE_P2P_HEADER_BYTES/16);
    const unsigned long long pattern = 0x5157453244503250ull ^ ((unsigned long long)getpid() << 20) ^ (unsigned long long)h->device;
    unsigned long long back = 0;
    HIP_TRY(h, hipMemcpy(probe, &pattern, sizeof(pattern), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(&back, probe, sizeof(back), hipMemcpyDeviceToHost));
    if (back != pattern) return fail(h, SWE2D_ERR_HIP, "swe2d_p2p_open: a copy into the peer's landing zone does not read back");
    unsigned long long *dback = nullptr;
    HIP_TRY(h, hipMalloc(&dback, sizeof(*dback)));
    hipLaunchKernelGGL(swe_p2p_probe_kernel, dim3(1), dim3(64), 0, h->stream, probe, ~pattern, dback);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) e = hipMemcpy(&back, dback, sizeof(back), hipMemcpyDeviceToHost);
    (void)hipFree(dback);
    if (e != hipSuccess) return fail(h, SWE2D_ERR_HIP, "swe2d_p2p_open: the probe kernel failed on the peer's landing zone");
    if (back != ~pattern) return fail(h, SWE2D_ERR_HIP, "swe2d_p2p_open: a kernel store into the peer's landing zone does not read back");
    *remote_base = p;
    return SWE2D_OK;
}

int swe2d_p2p_connect(swe2d_handle *hh, int32_t n_peers, void *const *remote_base, const int32_t *send_offset,
                      const int32_t *send_count, const int32_t *remote_recv_offset, const int32_t *remote_flag_index,
                      const int32_t *remote_n_recv, int32_t n_from)
{
    Handle *h = H(hh);
    if (!h || !h->p2p.zone) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_connect: no landing zone");
    if (n_peers < 0 || n_peers > SWE_P2P_MAX_PEERS || n_from < 0 || n_from > SWE_P2P_MAX_PEERS)
        return fail(h, SWE2D_ERR_UNSUPPORTED, "swe2d_p2p_connect: at most 8 peers per rank");
    auto &z = h->p2p;
    int end = 0;
    for (int i = 0; i < n_peers; i++) {
        if (!remote_base[i] || send_offset[i] != end || send_count[i] < 0 || remote_flag_index[i] < 0
            || remote_flag_index[i] >= SWE_P2P_MAX_PEERS || remote_recv_offset[i] < 0
            || remote_recv_offset[i] + send_count[i] > remote_n_recv[i])
            return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_connect: segments must tile the send list in order");
        end += send_count[i];
        z.remote_base[i] = static_cast<char *>(remote_base[i]);
        z.off[i] = send_offset[i]; z.cnt[i] = send_count[i];
        z.remote_off[i] = remote_recv_offset[i]; z.remote_flag[i] = remote_flag_index[i]; z.remote_n_recv[i] = remote_n_recv[i];
    }
    if (end != h->n_send) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_connect: segments do not cover the send list");
    z.n_peers = n_peers;
    z.n_from = n_from;
    return SWE2D_OK;
}

namespace {
int p2p_field(Handle *h, int channel, int i_buffer, double **planes, int *np)
{
    if (channel < 0 || channel >= h->p2p.n_channels || i_buffer < 0 || i_buffer > 2)
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "p2p: bad channel or buffer index");
    if (channel == 0) { *planes = h->state[i_buffer]; *np = 3*h->npc; }
    else {
        if (channel - 1 >= (int)h->tracers.size()) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "p2p: channel has no tracer");
        *planes = h->tracers[channel - 1].buf[i_buffer]; *np = h->npc;
    }
    if (*np != h->p2p.width[channel]) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "p2p: channel width mismatch");
    return SWE2D_OK;
}
}  // namespace

int swe2d_p2p_push(swe2d_handle *hh, int channel, int i_buffer)
{
    Handle *h = H(hh);
    if (!h || !h->p2p.zone) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_push: not connected");
    double *planes; int np;
    if (int rc = p2p_field(h, channel, i_buffer, &planes, &np)) return rc;
    auto &z = h->p2p;
    if (z.n_peers == 0 || h->n_send == 0) return SWE2D_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    SweP2pPushArgs a{};
    a.planes = planes; a.stride = h->stride; a.send_cells = h->send_cells; a.n_send = h->n_send; a.np = np;
    a.n_peers = z.n_peers;
    for (int i = 0; i < z.n_peers; i++) {
        a.off[i] = z.off[i]; a.cnt[i] = z.cnt[i];
        char *base = z.remote_base[i];
        a.rdata[i] = reinterpret_cast<double *>(base + p2p_channel_offset(z.width, channel, z.remote_n_recv[i]))
                     + (size_t)z.remote_off[i]*np;
        a.rslot[i] = (size_t)z.remote_n_recv[i]*np;
        a.rflag[i] = reinterpret_cast<unsigned long long *>(base)
                     + (size_t)(channel*SWE_P2P_MAX_PEERS + z.remote_flag[i])*SWE_P2P_FLAG_STRIDE;
    }
    a.ctr = z.ctr + channel;
    hipLaunchKernelGGL(swe_p2p_push_kernel, dim3(std::min(SWE_P2P_MAX_BLOCKS, grid_for(np*h->n_send))), dim3(256), 0, h->xstream ? h->xstream : h->stream, a);
    HIP_TRY(h, hipGetLastError());
    return SWE2D_OK;
}

int swe2d_p2p_wait_unpack(swe2d_handle *hh, int channel, int i_buffer)
{
    Handle *h = H(hh);
    if (!h || !h->p2p.zone) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_wait_unpack: not connected");
    double *planes; int np;
    if (int rc = p2p_field(h, channel, i_buffer, &planes, &np)) return rc;
    auto &z = h->p2p;
    if (z.n_from == 0 || h->n_recv == 0) return SWE2D_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    SweP2pUnpackArgs a{};
    a.planes = planes; a.stride = h->stride; a.recv_cells = h->recv_cells; a.n_recv = h->n_recv; a.np = np;
    a.n_from = z.n_from;
    char *base = static_cast<char *>(z.zone);
    for (int i = 0; i < z.n_from; i++)
        a.flag[i] = reinterpret_cast<const unsigned long long *>(base) + (size_t)(channel*SWE_P2P_MAX_PEERS + i)*SWE_P2P_FLAG_STRIDE;
    a.zone = reinterpret_cast<const double *>(base + p2p_channel_offset(z.width, channel, h->n_recv));
    a.slot = (size_t)h->n_recv*np;
    a.timeout_ticks = (unsigned long long)(z.timeout_s*1e8);
    a.ctr = z.ctr + channel;
    a.fence = z.zone_kind == 3;
    hipLaunchKernelGGL(swe_p2p_unpack_kernel, dim3(std::min(SWE_P2P_MAX_BLOCKS, grid_for(np*h->n_recv))), dim3(256), 0, h->xstream ? h->xstream : h->stream, a);
    HIP_TRY(h, hipGetLastError());
    return SWE2D_OK;
}

// The exchange kernels (swe2d_p2p_push / swe2d_p2p_wait_unpack: a few thousand cells, 5-6 us each, mostly latency) on a stream of
// their own: the caller orders it against the handle's stream with events (push after the send cells' stage, the next reader of the
// ghost cells after the unpack) and the stage kernels of the interior run meanwhile.  Null: back on the handle's stream.  No
// synchronisation here (usable around a stream capture that forks into this stream and joins again).
int swe2d_set_exchange_stream(swe2d_handle *hh, void *hip_stream)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    h->xstream = reinterpret_cast<hipStream_t>(hip_stream);
    return SWE2D_OK;
}

int swe2d_p2p_status(swe2d_handle *hh, int64_t *epochs_sent, int64_t *epochs_received, int32_t *timeouts)
{
    Handle *h = H(hh);
    if (!h || !h->p2p.ctr) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_status: not created");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->xstream) HIP_TRY(h, hipStreamSynchronize(h->xstream));
    std::vector<SweP2pCounters> c(h->p2p.n_channels);
    HIP_TRY(h, hipMemcpy(c.data(), h->p2p.ctr, c.size()*sizeof(SweP2pCounters), hipMemcpyDeviceToHost));
    int to = 0;
    for (int i = 0; i < h->p2p.n_channels; i++) {
        if (epochs_sent) epochs_sent[i] = (int64_t)c[i].epoch_send;
        if (epochs_received) epochs_received[i] = (int64_t)c[i].epoch_recv;
        to += (int)c[i].timeouts;
    }
    if (timeouts) *timeouts = to;
    return SWE2D_OK;
}

}  // extern "C"

#ifdef SWE_WAVE_TIMING
// profiling build only: copies the time stamps of the LAST stage launch (5 x SWE_WT_MAX, 100 MHz ticks)
extern "C" int swe2d_debug_read_wave_timing(swe2d_handle *hh, unsigned long long *out)
{
    Handle *h = H(hh);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpyFromSymbol(out, HIP_SYMBOL(swe_wave_ts), sizeof(unsigned long long)*6*SWE_WT_MAX));
    return SWE2D_OK;
}
#endif
