// swe2d_api_tracer.hip - tracers + vertex-based limiter + the coupled step: host side and ABI entry points
#include "swe2d_handle.h"
#include "swe2d_pick.h"

namespace {

// what a tracer stage kernel needs besides its buffers, weights and cell range
void fill_tracer_args(Handle *h, int id, SweTracerArgs &a, int in, int out, double a0, double a1, double beta, int c0, int c1, double *mean_out)
{
    Handle::Tracer &t = h->tracers[id];
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
    a.opp4 = h->opp4;
    a.mu_v = t.mu_v; a.mu_const = t.mu_const;
    a.diff_sipg = 3.0*t.sipg_factor;
    a.idxc = nullptr;
}

int launch_tracer_stage(Handle *h, int id, int in, int out, double a0, double a1, double beta, int c0, int c1,
                        double *mean_out = nullptr)
{
    if (c1 <= c0) return SWE2D_OK;
    Handle::Tracer &t = h->tracers[id];
    SweTracerArgs a;
    fill_tracer_args(h, id, a, in, out, a0, a1, beta, c0, c1, mean_out);
    // triangles: cell integral and interior facets of the diffusion inside the stage kernel, boundary facets by a launch
    // over the boundary cells (only when a marker has a diffusive boundary term at all)
    const bool fused_diff = t.diff && opt_on(h, SWE2D_OPT_VISC_FUSION) && h->npc == 3 && h->opp4;
    a.idxc = conn_pays(h, c1 - c0, fused_diff) ? h->idxc : nullptr;
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
    // Small cell ranges (a rank of eight: every launch is latency) on cells whose mean is the nodal average: the vertex kernel forms the
    // means itself - one launch less (3.5 us of 77 per step on a rank of eight of cfg 4, profiles/r06c_cfg4_rank8_kernel_stats.csv); on
    // a large mesh every cell would be averaged once per vertex instead of once (72 B read per cell instead of 24 + 24)
    const bool inline_means = !means_done && (h->npc == 3 || h->affine) && n < 250000;
    if (!means_done && !inline_means)       // (swe2d_advance_coupled has the last tracer stage write the means)
        hipLaunchKernelGGL(swe_limiter_cell_mean, dim3(grid_for(n)), dim3(256), 0, h->stream, t, h->stride, n, h->lim_mean, h->npc,
                           h->affine ? nullptr : h->cv, h->vx, h->vy);
    hipLaunchKernelGGL(swe_limiter_vertex_bounds, dim3(grid_for(nv)), dim3(256), 0, h->stream, h->lim_v2c_off,
                       h->lim_v2c_cell, h->lim_vbf_off, h->lim_vbf_facet, inline_means ? nullptr : h->lim_mean, t, h->stride, nv, h->lim_qmin,
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

int swe2d_tracer_diagnostics_limbs(swe2d_handle *hh, int id, int64_t limbs[2*SWE_SUM_LIMBS], double minmax[2])
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

int swe2d_advance_coupled(swe2d_handle *hh, int n_steps, int tracer_only, int use_limiter)
{
    Handle *h = H(hh);
    if (!h || n_steps < 0) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad n_steps");
    if (h->n_owned != h->n_cells) return fail(h, SWE2D_ERR_UNSUPPORTED, "on a partition the host drives the coupled step (stages on cell ranges + halo exchanges, thetis_amd/distributed.py)");
    HIP_TRY(h, hipSetDevice(h->device));
    RoctxRange range(h, "swe2d_advance_coupled");
    for (int it = 0; it < n_steps; it++) {
        // the shallow-water step as swe2d_advance makes it on a mesh beyond the dataflow kernel: fused stage pair + stage 3 where
        // that covers the handle (cfg 4 on 1 M triangles), stage launches otherwise
        if (!tracer_only) { if (int rc = step_swe(h)) return rc; }
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
