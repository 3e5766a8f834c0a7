// swe2d_api.hip - C ABI (include/swe2d.h) over the HIP stage kernels.  gfx950 only, no CPU fallback: every entry
// point fails with SWE2D_ERR_NO_DEVICE / SWE2D_ERR_HIP when the HIP runtime or a device is missing.
#include "swe2d_handle.h"
#include "swe2d_pick.h"

namespace swe2d_impl {

thread_local std::string g_create_error;
std::atomic<unsigned long long> g_next_uid{1ull};

// Shu-Osher coefficients of SSPRK33: output of thetis/rungekutta.py:13-87 (butcher_to_shuosher_form) for the
// tableau of rungekutta.py:342-346; pinned by tests/golden/shuosher_ssprk33.json.
//   U1 = 1*k0 + 1*U0;  U2 = 1/4*k1 + 3/4*U0 + 1/4*U1;  U3 = b32*k2 + a30*U0 + a32*U2
extern const double kBeta[3] = {1.0, 0.25, 0.6666666666666666};
extern const double kAlpha0[3] = {1.0, 0.75, 0.33333333333333337};   // weight of stage_sol[0]
extern const double kAlphaIn[3] = {0.0, 0.25, 0.6666666666666666};   // weight of the stage's input (stage 0: U0 itself)

int fail(Handle *h, int code, const std::string &msg)
{
    if (h) h->err = msg; else g_create_error = msg;
    return code;
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
    a.idxc = conn_pays(h, c1 - c0, h->wd) ? h->idxc : nullptr;
    a.cell_begin = c0; a.cell_end = c1;
    a.reverse = 0;
    a.wall_general = opt_on(h, SWE2D_OPT_WALL_FAST) ? 0 : 1;
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
    const bool fused_visc = h->visc && opt_on(h, SWE2D_OPT_VISC_FUSION) && h->npc == 3 && !h->wd && h->opp4;
    // Boundary-inline variant of the triangle kernel (BINL, swe2d_kernels.h): faster than or equal to the epilogue variant at
    // every size (us/step, same box, production numbering: 125 k cells 27.9 -> 26.3, 250 k 41.0 -> 38.2, 500 k 66.1 -> 63.6,
    // 1 M 117.9 -> 117.8); both give the same bits.  SWE2D_OPT_BND_INLINE = 0 selects the epilogue variant (parity test, A/B).
    const bool binl = h->opt[SWE2D_OPT_BND_INLINE] != 0;
    // Wetting-drying (round 5, the device carries D): the epilogue variant (158 VGPRs, 3 waves per SIMD) beats the boundary-inline
    // one (190 VGPRs, 2 waves) - cfg 5 at 500 k cells 83.5 against 89.3 us per step, same box (profiles/r05c_cfg5.txt); with the
    // eta-carrying kernels of rounds 2-4 the two ran the same.  SWE2D_OPT_BND_INLINE = 1 forces the inline variant (same bits).
    const bool binl_wd = h->opt[SWE2D_OPT_BND_INLINE] > 0;
    // ... and for launches whose state no longer fits the Infinity Cache (three buffers of 24 B per node against 256 MB: beyond
    // ~1.24 M triangles) with the in-wave neighbour traces exchanged through LDS (LDSX; SWE2D_OPT_LDSX = 0 / 1 forces the choice).
    // With the device's tile-Hilbert numbering and the alternating direction below, same box, us/step without / with:
    // 1 M cells 115-118 / 118-119, 1.25 M 166-168 / 162-163, 1.5 M 206-210 / 197-199, 2 M 288-294 / 275, 2.5 M 363-364 / 345-347,
    // 3 M 420-428 / 394-398, 4 M 573-583 / 544 (profiles/r04w_ldsx_threshold.txt; rounds 2-3 took it from 3 M cells only).
    // Same bits in every variant: the kernel has no implicit contraction.
    const bool beyond_cache = (size_t)(c1 - c0)*h->npc*72 >= ((size_t)256 << 20);
    const bool ldsx = h->opt[SWE2D_OPT_LDSX] >= 0 ? h->opt[SWE2D_OPT_LDSX] != 0 : beyond_cache;
    stage_kernel_t kern = fused_visc
        ? pick_kernel_visc(h->par.use_nonlinear_equations != 0, h->par.use_lax_friedrichs_velocity != 0, has_u0, has_sources(h))
        : h->wd ? pick_kernel_wd(h->par.use_lax_friedrichs_velocity != 0, has_u0, has_sources(h), h->npc == 4 ? (h->affine ? 1 : 2) : 0, binl_wd)
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
    // 0.732).  SWE2D_OPT_ALTERNATE = 0 / 1 forces the choice.
    if (!fused_visc) {
        const bool alt = h->opt[SWE2D_OPT_ALTERNATE] >= 0 ? h->opt[SWE2D_OPT_ALTERNATE] != 0 : beyond_cache;
        if (alt) { a.reverse = h->launch_parity; h->launch_parity ^= 1; }
    }
    SWE_CHK_SYNC(h->stream);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SWE_BLOCK), 0, h->stream, a);
    HIP_TRY(h, hipGetLastError());
    if (out == 1 || out == 2) h->stage_valid[out - 1] = true;      // a stage launch leaves its stage solution in its buffer
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

int scatter_facet_values(Handle *h, double *planes, int n, const int32_t *cells, const int32_t *facets,
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

int upload_vertex_coefficient(Handle *h, const double *vertex_values, double **dev)
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

}  // namespace swe2d_impl

extern "C" {

int swe2d_abi_version(void) { return SWE2D_ABI_VERSION; }

static bool advance_takes_flow(Handle *h);
int swe2d_fused_pair_info(swe2d_handle *hh, int32_t out[4])
{
    Handle *h = H(hh);
    if (!h || !out) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null argument");
    out[0] = out[1] = out[2] = 0; out[3] = h->n_cells;
    // (small whole meshes: swe2d_advance takes the dataflow kernel first; a partition's stage pairs are driven by the host)
    if (!fuse12_covers(h) || (h->n_owned == h->n_cells && advance_takes_flow(h))) return SWE2D_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    if (int rc = fuse12_build(h)) return rc;
    if (h->npc == 4) { if (h->fuseq_tile) { out[0] = 1; out[1] = h->fuseq_n_tiles; out[2] = (int32_t)h->fuseq_ring_cells; } }
    else if (h->fuse_tile) { out[0] = 1; out[1] = h->fuse_n_tiles; out[2] = (int32_t)h->fuse_ring_cells; }
    return SWE2D_OK;
}

int swe2d_connectivity_info(swe2d_handle *hh, int32_t out[2])
{
    Handle *h = H(hh);
    if (!h || !out) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null argument");
    const bool on = h->idxc && h->opt[SWE2D_OPT_COMPACT_IDX] != 0;
    out[0] = on ? 1 : 0;
    out[1] = on ? h->n_conn_escapes : 0;
    return SWE2D_OK;
}

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

int swe2d_set_option(swe2d_handle *hh, int option, int value)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    if (option < 0 || option >= SWE2D_OPT_COUNT) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_set_option: unknown option");
    if (value < -1) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_set_option: value must be >= -1 (-1: the library's own rule)");
    h->opt[option] = value;
    if (option == SWE2D_OPT_FUSED_STAGES && h->fuse_state == -1) h->fuse_state = 0;     // tiles judged not worth it: judged again under the new setting
    if (option == SWE2D_OPT_FLOW_CAPACITY) h->flow_capacity = -1;
    return SWE2D_OK;
}

int swe2d_get_option(swe2d_handle *hh, int option, int *value)
{
    Handle *h = H(hh);
    if (!h || !value) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null argument");
    if (option < 0 || option >= SWE2D_OPT_COUNT) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_get_option: unknown option");
    *value = h->opt[option];
    return SWE2D_OK;
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
            (void)hipGetLastError();                                                                     \
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
            // (the same test as thetis_amd/mesh.py, Mesh2d.affine: a cell between two different tests would make the global mesh
            //  'affine' on the host while the handle that owns it takes the general kernels - ghost / owner bit mismatch)
            if (std::max(std::fabs(sx), std::fabs(sy)) > 1e-9*std::sqrt(area2)) {
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

    h->h_nbr = nbr;                                    // (host copy of the packed neighbour codes: tile tables of the fused stage pair)
    if (npc == 3) {
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
        {   // the 16-B form of the same records (swe_conn_pack): what the stage kernels read
            {
                std::vector<int4> pc((size_t)S, int4{0, 0, 0, (int)0x80000000u});
                h->n_conn_escapes = 0;
                for (int kk = 0; kk < n; kk++) {
                    const int nb3[3] = {nbr[kk], nbr[S + kk], nbr[2*S + kk]}, v3[3] = {cv[kk], cv[S + kk], cv[2*S + kk]};
                    pc[kk] = swe_conn_pack(kk, nb3, v3);
                    if (pc[kk].w < 0) h->n_conn_escapes++;
                }
                HIP_TRY_C(hipMalloc(&h->idxc, (size_t)S*sizeof(int4)));
                HIP_TRY_C(hipMemcpy(h->idxc, pc.data(), (size_t)S*sizeof(int4), hipMemcpyHostToDevice));
            }
        }
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
        // dataflow stage loop (swe2d_flow.h): one stage counter per 64-cell block, the status word, and the exchange slots of
        // the rim facets - interior facets whose two cells sit in different blocks - numbered in cell order
        h->flow_blocks = (n + SWE_BLOCK - 1)/SWE_BLOCK;
        HIP_TRY_C(hipMalloc(&h->flow_flag, (size_t)h->flow_blocks*SWE_FLOW_FLAG_STRIDE*sizeof(unsigned)));
        HIP_TRY_C(hipMalloc(&h->flow_status, 4*sizeof(unsigned)));
        HIP_TRY_C(hipMemset(h->flow_flag, 0, (size_t)h->flow_blocks*SWE_FLOW_FLAG_STRIDE*sizeof(unsigned)));
        HIP_TRY_C(hipMemset(h->flow_status, 0, 4*sizeof(unsigned)));
        if (int rc = flow_build(h, nullptr)) { g_create_error = h->err; swe2d_destroy(reinterpret_cast<swe2d_handle *>(h)); return rc; }
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
                    h->lim_qmin, h->lim_qmax, h->valpha, h->snapshot[0].data, h->snapshot[1].data, h->bc_field[0], h->bc_field[1], h->bc_field[2], h->bc_field[3], h->nu_v, h->idx4, h->idx2, h->idxc, h->fuse_tile, h->fuse_inner, h->fuse3_tile, h->fuse3_cnt, h->fuseq_tile, h->fuseq_inner, h->opp4, h->bnd_cells, h->flow_flag, h->flow_status, h->flow_xo4, h->flow_xo2, h->flow_ex, h->flow_xblk, h->flow_xsrc, h->flow_cell, h->flow_xsend, h->flow_xrecv, h->flow_xtick};
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
    h->state_holds_D = h->wd;                      // wetting-drying: the elevation planes now hold the displaced depth D
    h->stage_valid[0] = h->stage_valid[1] = false; // no stage of this state has run yet
    HIP_TRY(h, hipStreamSynchronize(h->stream));   // host buffers may be reused by the caller
    return SWE2D_OK;
}

// Library-internal save / restore of the time-stepping state (buffer A and every tracer's buffer A) on the device: what graph
// capture, verification replays and benchmarks need around steps they must undo.  Exact - unlike swe2d_get_state followed by
// swe2d_set_state with wetting-drying, where the planes hold D and eta -> D -> eta is the identity only up to rounding.
int swe2d_state_snapshot(swe2d_handle *hh, int restore) { return swe2d_state_snapshot_slot(hh, 0, restore); }

int swe2d_state_snapshot_slot(swe2d_handle *hh, int slot, int restore)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    if (slot < 0 || slot >= SWE2D_SNAPSHOT_SLOTS) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_state_snapshot: no such slot");
    HIP_TRY(h, hipSetDevice(h->device));
    Handle::Snapshot &sn = h->snapshot[slot];
    const size_t nb = (size_t)3*h->npc*h->stride*sizeof(double), nt = (size_t)h->npc*h->stride*sizeof(double);
    const size_t total = nb + h->tracers.size()*nt;
    if (restore) {
        if (!sn.data || sn.bytes != total) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_state_snapshot: nothing to restore");
        HIP_TRY(h, hipMemcpyAsync(h->state[0], sn.data, nb, hipMemcpyDeviceToDevice, h->stream));
        for (size_t t = 0; t < h->tracers.size(); t++)
            HIP_TRY(h, hipMemcpyAsync(h->tracers[t].buf[0], (char *)sn.data + nb + t*nt, nt, hipMemcpyDeviceToDevice, h->stream));
        h->state_holds_D = sn.holds_D;
        h->stage_valid[0] = h->stage_valid[1] = false;     // the stage buffers belong to the steps that are being undone
        return SWE2D_OK;
    }
    if (sn.data && sn.bytes != total) { HIP_TRY(h, hipStreamSynchronize(h->stream)); (void)hipFree(sn.data); sn.data = nullptr; }
    if (!sn.data) { HIP_TRY(h, hipMalloc(&sn.data, total)); sn.bytes = total; }
    HIP_TRY(h, hipMemcpyAsync(sn.data, h->state[0], nb, hipMemcpyDeviceToDevice, h->stream));
    for (size_t t = 0; t < h->tracers.size(); t++)
        HIP_TRY(h, hipMemcpyAsync((char *)sn.data + nb + t*nt, h->tracers[t].buf[0], nt, hipMemcpyDeviceToDevice, h->stream));
    sn.holds_D = h->state_holds_D;
    return SWE2D_OK;
}

int swe2d_get_state(swe2d_handle *hh, double *uv, double *eta) { return swe2d_get_stage_state(hh, 2, uv, eta); }

int swe2d_get_stage_state(swe2d_handle *hh, int i_stage, double *uv, double *eta)
{
    Handle *h = H(hh);
    if (!h || !uv || !eta) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null argument");
    if (i_stage < 0 || i_stage > 2) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "i_stage must be 0, 1 or 2");
    // stage_sol[i] of the reference always is what stage i left (rungekutta.py:930-946).  Here the fused and the dataflow kernels keep
    // the intermediate stage solutions on chip: a buffer they did not write is not handed out as if they had
    if (int rc = capture_parity_check(h)) return rc;
    if (i_stage < 2 && !h->stage_valid[i_stage])
        return fail(h, SWE2D_ERR_UNSUPPORTED, "swe2d_get_stage_state: the last step did not leave this stage solution in memory (fused stages / "
                    "dataflow kernel, or no stage has run since the state was set); run the step with swe2d_solve_stage to read it");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t n = (size_t)h->n_cells*h->npc;
    // (wetting-drying: the planes of all three buffers hold D, the host gets eta)
    const bool isD = h->wd && (i_stage != 2 || h->state_holds_D);
    hipLaunchKernelGGL(swe_planes_to_aos, dim3(grid_for(h->n_cells)), dim3(256), 0, h->stream,
                       h->state[(i_stage + 1) % 3], h->stage_uv, h->stage_eta, h->stride, h->n_cells, h->npc,
                       isD ? h->cv : nullptr, isD ? h->vh : nullptr, isD ? h->valpha : nullptr);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(uv, h->stage_uv, 2*n*sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(eta, h->stage_eta, n*sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (int rc = capture_parity_check(h)) return rc;
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
    if (enable) {
        if (!alpha_vertex) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "alpha_vertex is required");
        if (!h->par.use_nonlinear_equations)
            return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "wetting and drying needs use_nonlinear_equations");
        for (int i = 0; i < h->n_vertices; i++)
            if (!(alpha_vertex[i] >= 0.0)) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "alpha must be >= 0");
    }
    HIP_TRY(h, hipSetDevice(h->device));
    // The device carries D (swe2d_kernels.h, swe_wd_eta) while wetting-drying is on.  A state that is resident when the switch or
    // alpha changes keeps its ELEVATION: D -> eta with the alpha it was formed with, then eta -> D with the new one.
    if (h->state_holds_D) {
        hipLaunchKernelGGL(swe_wd_planes_to_eta, dim3(grid_for(h->n_cells)), dim3(256), 0, h->stream, h->state[0], h->stride, h->n_cells,
                           h->npc, h->cv, h->vh, h->valpha);
        HIP_TRY(h, hipGetLastError());
        h->state_holds_D = false;
    }
    if (!enable) { h->wd = false; return SWE2D_OK; }
    if (!h->valpha) HIP_TRY(h, hipMalloc(&h->valpha, (size_t)h->n_vertices*sizeof(double)));
    HIP_TRY(h, hipMemcpyAsync(h->valpha, alpha_vertex, (size_t)h->n_vertices*sizeof(double), hipMemcpyHostToDevice, h->stream));
    h->wd = true;
    if (h->npc == 4) hipLaunchKernelGGL(swe_wd_clip_kernel<4>, dim3(grid_for(h->n_cells)), dim3(256), 0, h->stream,
                                        h->state[0], h->stride, h->cv, h->vh, h->valpha, h->n_cells, h->affine ? nullptr : h->vx, h->vy);
    else hipLaunchKernelGGL(swe_wd_clip_kernel<3>, dim3(grid_for(h->n_cells)), dim3(256), 0, h->stream,
                            h->state[0], h->stride, h->cv, h->vh, h->valpha, h->n_cells, nullptr, nullptr);
    HIP_TRY(h, hipGetLastError());
    h->state_holds_D = true;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SWE2D_OK;
}

// shared by swe2d_set_viscosity / swe2d_tracer_set_diffusivity: upload (or drop) a per-vertex coefficient
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
    RoctxRange range(h, names[(i_stage >= 0 && i_stage < 3) ? i_stage : 0]);
    return stage_on_range(h, i_stage, cell_begin, cell_end);
}

int swe2d_solve_stage(swe2d_handle *hh, int i_stage)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    return swe2d_solve_stage_cells(hh, i_stage, 0, h->n_owned);
}

// does swe2d_advance run this handle's steps in the dataflow kernel?
static bool advance_takes_flow(Handle *h)
{
    return opt_on(h, SWE2D_OPT_FLOW) && flow_kernel_covers(h) && ((h->flow_blocks + 7)/8)*8 <= flow_capacity(h);
}

int swe2d_advance(swe2d_handle *hh, int n_steps)
{
    Handle *h = H(hh);
    if (!h || n_steps < 0) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad n_steps");
    if (h->n_owned != h->n_cells)
        return fail(h, SWE2D_ERR_UNSUPPORTED, "swe2d_advance on a partition: drive stages + halo exchange from the host");
    HIP_TRY(h, hipSetDevice(h->device));
    RoctxRange range(h, "swe2d_advance");
    // Up to 128 steps per launch without grid barriers (swe2d_flow.h) where every 64-cell block of the mesh is resident at once
    // (<= 131 k cells) and the kernel covers the configuration.  Same box, us/step, three stage launches per step -> flow launches:
    // 15 k cells 16.5 -> 15.3, 62 k 20.1 -> 15.1, 125 k 24.3 -> 18.3 (the one-launch step kernel of round 2, which this replaces:
    // 14.1 / 16.8 / 24.9).  SWE2D_OPT_FLOW = 0 selects the stage launches (the same bits either way).
    {
        if (n_steps > 0 && advance_takes_flow(h)) {
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
    // stages 1 and 2 in one launch by overlapped tiles (swe2d_fuse.h) + stage 3 as a stage launch where that kernel covers the
    // handle, three stage launches otherwise: the same bits (step_swe, swe2d_api_fuse.hip)
    for (int it = 0; it < n_steps; it++)
        if (int rc = step_swe(h)) return rc;
    return SWE2D_OK;
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
    h->stage_valid[0] = h->stage_valid[1] = false;
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
    // events around every launch of a step, on the launch stream: three stage launches, the fused stage pair + stage 3, or the one
    // launch of all three stages (the mean is per element-update - a third of a step - in every case)
    const bool triple = fuse123_wanted(h);
    if (!triple && fuse12_covers(h)) { if (int rc = fuse12_build(h)) return rc; }
    const bool fused = !triple && fuse12_covers(h) && (h->npc == 4 ? h->fuseq_tile != nullptr : h->fuse_tile != nullptr);
    const int lps = triple ? 1 : (fused ? 2 : 3);
    const int nl = lps*n_steps;
    std::vector<hipEvent_t> ev(2*(size_t)nl);
    for (auto &e : ev) HIP_TRY(h, hipEventCreate(&e));
    HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    int l = 0;
    for (int it = 0; it < n_steps; it++)
        for (int s = 0; s < lps; s++, l++) {
            HIP_TRY(h, hipEventRecord(ev[2*l], h->stream));
            int rc = triple ? launch_fuse123(h, h->n_owned)
                            : (fused ? (s == 0 ? launch_fuse12(h, h->n_owned) : stage_on_range(h, 2, 0, h->n_owned)) : stage_on_range(h, s, 0, h->n_owned));
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
    if (ms_kernel_avg) *ms_kernel_avg = (float)(sum/(3.0*n_steps));
    return SWE2D_OK;
}

int swe2d_synchronize(swe2d_handle *hh)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (int rc = capture_parity_check(h)) return rc;
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
    h->stage_valid[0] = false;                               // buffer B holds the tendency, not a stage solution
    const size_t n = (size_t)h->n_cells*h->npc;
    hipLaunchKernelGGL(swe_planes_to_aos, dim3(grid_for(h->n_cells)), dim3(256), 0, h->stream,
                       h->state[1], h->stage_uv, h->stage_eta, h->stride, h->n_cells, h->npc);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(k_uv, h->stage_uv, 2*n*sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(k_eta, h->stage_eta, n*sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SWE2D_OK;
}

// The total of limb sums (swe_sum_accumulate) rounded to the nearest double (ties to even), exactly: a pure function of the six
// integers, so a sum taken over one device and the same sum taken over the partitions of eight agree in every bit.
//   V = sum_j L_j 2^(40 - 38 j),  j = 0 .. 5
double swe2d_sum_limbs_to_double(const int64_t limbs[SWE_SUM_LIMBS])
{
    constexpr int N = SWE_SUM_LIMBS;
    // carry-normalise: 0 <= L_1 .. L_5 < 2^38, L_0 signed (the sums of at most 2^25 limbs of 38 bits fit 64 bits with room)
    int64_t L[N];
    for (int j = 0; j < N; j++) L[j] = limbs[j];
    auto normalise = [&]() {
        for (int j = N - 1; j > 0; j--) {
            const int64_t carry = L[j] >> 38;                 // arithmetic shift: floor
            L[j] -= carry*((int64_t)1 << 38);
            L[j - 1] += carry;
        }
    };
    normalise();
    const bool neg = L[0] < 0;
    if (neg) { for (int j = 0; j < N; j++) L[j] = -L[j]; normalise(); }      // |V|: now every limb is >= 0
    int t = 0;
    while (t < N && L[t] == 0) t++;
    if (t == N) return neg ? -0.0 : 0.0;
    // the three leading limbs (114 bits, at least 77 of them significant once L_0 needs more than one: L_0 < 2^63 -> take it apart)
    unsigned __int128 T;
    int unit;                                                 // exponent of T's last bit
    bool sticky = false;
    if (t == 0 && (L[0] >> 38) != 0) {
        // a top limb wider than 38 bits (up to 63): T = L_0 2^38 + L_1 (101 bits), the rest is sticky
        T = ((unsigned __int128)(uint64_t)L[0] << 38) | (uint64_t)L[1];
        unit = 40 - 38;
        for (int j = 2; j < N; j++) sticky = sticky || L[j] != 0;
    } else {
        const int64_t a = L[t], b = t + 1 < N ? L[t + 1] : 0, c = t + 2 < N ? L[t + 2] : 0;
        T = ((unsigned __int128)(uint64_t)a << 76) | ((unsigned __int128)(uint64_t)b << 38) | (uint64_t)c;
        unit = 40 - 38*(t + 2);
        for (int j = t + 3; j < N; j++) sticky = sticky || L[j] != 0;
    }
    // T has at most 114 bits; bring its leading bit up to bit 120 so that a sticky bit at position 0 sits at least 60 bits below the
    // rounding position of the conversion (the compiler's 128-bit -> double conversion rounds to nearest even)
    int bits = 0;
    for (unsigned __int128 m = T; m; m >>= 1) bits++;
    const int sft = 120 - bits;
    T <<= sft;
    if (sticky) T |= 1;
    const double r = std::ldexp((double)T, unit - sft);
    return neg ? -r : r;
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

int swe2d_diagnostics_limbs(swe2d_handle *hh, int64_t limbs[3*SWE_SUM_LIMBS], double *min_depth)
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
