// swe2d_handle.h - what the translation units of the C ABI share: the handle, error plumbing, internal entry points.
// Not part of the boundary (include/swe2d.h is).
#pragma once
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

namespace swe2d_impl {

extern thread_local std::string g_create_error;

// Shu-Osher coefficients of SSPRK33 (swe2d_api.hip)
extern const double kBeta[3], kAlpha0[3], kAlphaIn[3];

extern std::atomic<unsigned long long> g_next_uid;

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
    int opt[SWE2D_OPT_COUNT];                           // swe2d_set_option: -1 = the library's own rule (set in the constructor below)
    Handle() { for (int i = 0; i < SWE2D_OPT_COUNT; i++) opt[i] = -1; }
    // which of the stage buffers hold the stage solutions of the step made last (swe2d_get_stage_state): the fused and the dataflow
    // kernels keep U(1) (and U(2)) on chip
    bool stage_valid[2] = {false, false};
    int capture_swaps = 0;                              // buffer swaps made while the stream was capturing (launch_fuse123): must be even per captured sequence
    int4 *idx4 = nullptr;                               // packed triangle connectivity (stage kernel), see SweStageArgs
    int2 *idx2 = nullptr;
    int4 *idxc = nullptr;                               // ... in 16 B (swe_conn_pack), what the stage kernels read (SWE2D_OPT_COMPACT_IDX)
    // stages 1 + 2 of a step in one launch by overlapped tiles (swe2d_fuse.h; SWE2D_OPT_FUSED_STAGES): tile tables, built at first use
    int2 *fuse_tile = nullptr;
    int *fuse_inner = nullptr;
    int fuse_n_tiles = 0;
    int fuse_state = 0;                                 // -1: the numbering gives poor tiles, -2: first use inside a stream capture: stage launches
    long long fuse_ring_cells = 0;
    // ... the stage pair on quadrilaterals (swe_fuse12_quad_kernel)
    int4 *fuseq_tile = nullptr;
    int *fuseq_inner = nullptr;
    int fuseq_n_tiles = 0;
    long long fuseq_ring_cells = 0;
    // ... all three stages in one launch, two rings per tile (SWE2D_OPT_FUSED_STAGES = 3): tile tables, built at first use
    int2 *fuse3_tile = nullptr, *fuse3_cnt = nullptr;
    int fuse3_n_tiles = 0;
    long long fuse3_ring1 = 0, fuse3_ring2 = 0;
    std::vector<int> fuse_order;                        // cells in the order the tiles are cut from (swe2d_fused_set_order); empty: the numbering
    std::vector<int> fuse3_order;                       // the same for the two-ring tiles alone (swe2d_fused_set_triple_tiles); empty: fuse_order
    std::vector<unsigned char> fuse3_start;             // [n_cells] 1 = a two-ring tile must begin at this position of the order; empty: none
    int n_conn_escapes = 0;                             // cells whose record is an escape to the wide ones
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
    double *vx = nullptr, *vy = nullptr, *vh = nullptr;
    double *bc_field[4] = {nullptr, nullptr, nullptr, nullptr};  // Function-valued boundary data per facet: elev, uv, un, flux
    double *valpha = nullptr;                          // per-vertex wetting-drying alpha
    bool wd = false;
    bool state_holds_D = false;                        // wetting-drying: the elevation planes of buffer A hold the displaced depth D
    struct Snapshot { void *data = nullptr; size_t bytes = 0; bool holds_D = false; };
    Snapshot snapshot[SWE2D_SNAPSHOT_SLOTS];           // swe2d_state_snapshot: buffer A + the tracers' buffers A, per slot
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

int fail(Handle *h, int code, const std::string &msg);

#define HIP_TRY(h, expr)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            /* the error is reported through the return code; left in the runtime's "last error" it would surface in the NEXT  \
               caller that checks it - e.g. torch's launch check after a peer mapping that failed (first contact with a node) */ \
            (void)hipGetLastError();                                                             \
            return fail(h, SWE2D_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
        }                                                                                        \
    } while (0)

// Optional ROCTx ranges around the entry points that advance the state (SWE2D_OPT_ROCTX = 1): they show up as named ranges in
// `rocprofv3 --marker-trace` next to the kernel trace.  The tracing library is looked up at run time (rocprofiler-sdk's
// librocprofiler-sdk-roctx.so, else roctracer's libroctx64.so) when the first handle asks for it; without it, or without the
// option, the ranges are no-ops.
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
            {
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
    RoctxRange(const Handle *h, const char *name);
    ~RoctxRange() { if (pop_) pop_(); }
};
inline RoctxRange::RoctxRange(const Handle *h, const char *name)
{
    if (!h || h->opt[SWE2D_OPT_ROCTX] <= 0) return;
    push_t push;
    resolve(push, pop_);
    if (push) push(name); else pop_ = nullptr;
}


inline bool has_sources(const Handle *h)
{
    for (int i = 0; i < SWE2D_FIELD_COUNT; i++) if (h->field[i]) return true;
    return h->scalar[SWE2D_SCALAR_LINEAR_DRAG] >= 0 || h->scalar[SWE2D_SCALAR_QUADRATIC_DRAG] >= 0
           || h->scalar[SWE2D_SCALAR_MANNING_DRAG] >= 0 || h->scalar[SWE2D_SCALAR_NIKURADSE] >= 0;
}

inline int grid_for(int n) { return (n + 255)/256; }

// options (swe2d_set_option): on unless switched off / the value in seconds with its default
inline bool opt_on(const Handle *h, int o) { return h->opt[o] != 0; }
inline double opt_seconds(const Handle *h, int o, double dflt) { return h->opt[o] > 0 ? 1e-3*h->opt[o] : dflt; }

// ---- stage launches (swe2d_api.hip)
// Where the 16-B connectivity records (swe2d_conn.h) pay: launches that stream from memory - from ~250 k cells, where the 8 B
// they save per cell are 2-3 % of a stage's traffic (profiles/r05zc: 1 M triangles 113.1 -> 110.3 us per step, 500 k 63.6 -> 62.7);
// a launch of 125 k cells is latency-bound and the ~20 integer instructions of the decode make it 1 % slower (24.0 against 23.8),
// and so are the kernels bound by their arithmetic (wetting-drying + Manning: 80.5 against 79.8; tracer with fused diffusion).
inline bool conn_pays(const Handle *h, int n_cells_of_launch, bool arithmetic_bound)
{
    const int o = h->opt[SWE2D_OPT_COMPACT_IDX];
    return h->idxc && o != 0 && (o == 2 || (n_cells_of_launch >= 250000 && !arithmetic_bound));
}
bool fuse12_covers(const Handle *h);
int fuse12_build(Handle *h);
int launch_fuse12(Handle *h, int cell_end);
bool fuse123_wanted(const Handle *h);
int fuse123_build(Handle *h);
int launch_fuse123(Handle *h, int cell_end);
int capture_parity_check(Handle *h);                    // SWE2D_ERR_UNSUPPORTED once after a capture that swapped the state buffers an odd number of times
int step_swe(Handle *h);                               // one SSPRK33 step of the shallow-water state on the whole mesh: fused pair + stage 3, or stage launches
void fill_stage_args(Handle *h, SweStageArgs &a, int in, int u0, int out, double a0, double a1, double beta, int c0, int c1);
int launch_stage(Handle *h, int in, int u0, int out, double a0, double a1, double beta, int c0, int c1);
int stage_on_range(Handle *h, int i_stage, int c0, int c1);
// per-facet values -> the facet planes of a boundary field; a per-vertex coefficient -> the device (swe2d_api.hip)
int scatter_facet_values(Handle *h, double *planes, int n, const int32_t *cells, const int32_t *facets,
                         const double *values, int ncomp, int nval);
int upload_vertex_coefficient(Handle *h, const double *vertex_values, double **dev);
// ---- dataflow stage loop (swe2d_api_flow.hip)
int flow_build(Handle *h, const int32_t *order);
bool flow_kernel_covers(const Handle *h);
int flow_capacity(Handle *h);
int flow_build_exchange(Handle *h);
int launch_flow(Handle *h, int n_stages, const int32_t *cell_end, int n_cycles = 0);
int flow_check(Handle *h);
// ---- peer-to-peer halo (swe2d_api_p2p.hip)
size_t p2p_channel_offset(const int *width, int c, int n_recv);

}  // namespace swe2d_impl
using namespace swe2d_impl;
