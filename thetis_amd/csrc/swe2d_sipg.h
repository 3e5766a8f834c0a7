// Symmetric-interior-penalty diffusion operators of the explicit path, DG-P1 triangles:
//   NC = 2: HorizontalViscosityTerm        thetis/shallowwater_eq.py:554-616  (rows = velocity components)
//   NC = 1: tracer HorizontalDiffusionTerm thetis/tracer_eq_2d.py:226-278
// Both are optional (coefficient None in the reference => the term returns 0), so they are NOT part of the fused stage
// kernels: when enabled, this pass runs right after the stage kernel on the same cell range and accumulates
//   U_out[rows] += beta*dt*M^-1 R_sipg(U_in)
// (the stage update is linear in the residual).  One lane per cell, both sides of a facet evaluate the same symmetric
// flux => no atomics, deterministic.  Unlike the advective fluxes the SIPG terms need the neighbour's gradient, i.e. its
// third node and third vertex (one extra dependent gather per facet).
//
// Coefficients (viscosity nu / diffusivity mu) are a constant or a continuous P1 field given per vertex, so that
// avg(nu) = nu on a facet and avg(nu grad c) = nu avg(grad c); a discontinuous coefficient is rejected on the host.
#pragma once
#include "swe2d_kernels.h"

#define SWE_SIPG_BC_NONE 0          // funcs is None: no boundary term
#define SWE_SIPG_BC_DIFF_FLUX 1     // tracer 'diff_flux'                            tracer_eq_2d.py:267-268
#define SWE_SIPG_BC_UPWIND 2        // tracer, constant 'value': -phi mu s grad(c).n (grad c_ext = 0)   tracer_eq_2d.py:270-276
#define SWE_SIPG_BC_GRAD_IN 3       // tracer, funcs without 'value' (c_ext = c_in): -phi mu grad(c).n
#define SWE_SIPG_BC_VALUE_FIELD 4   // tracer, Function 'value': -phi mu (s grad(c) + (1-s) grad(c_ext)).n

struct SweSipgArgs {
    const double *in;       // 3*NC planes: row c, node i at in[(3c + i)*S + k]
    double *out;            // same layout; accumulated into
    size_t stride;
    const int *nbr, *cv;
    const double *vx, *vy, *vh;
    const double *mu_v;     // per-vertex coefficient or null
    double mu_const;
    double sipg;            // sipg_factor * cp,  cp = (p+1)(p+2)/2 = 3          shallowwater_eq.py:571-576
    double dt, beta;
    int cell_begin, cell_end;
    // BND_ONLY launches (the viscosity of triangles is otherwise fused into the stage kernel): lane t works on cell
    // cell_list[t], cells outside [cell_begin, cell_end) are skipped
    const int *cell_list;
    int n_list;
    // viscosity only
    int grad_div, grad_depth, nonlin;
    int wd;                 // wetting-drying: the total depth is the nodally displaced depth D (valpha per vertex)
    const double *eta;      // 3 planes (total depth of the grad-depth term and of 'flux' boundaries)
    SweBcTable bc;
    const double *bc_elev_f, *bc_uv_f, *bc_un_f, *bc_flux_f;     // per-facet planes, see SweStageArgs
    // tracer only
    const double *uv;       // velocity planes (upwind switch of the boundary term)
    double vel_factor;
    int bc_diff_kind[SWE_MAX_MARKERS];
    double bc_diff_flux[SWE_MAX_MARKERS];
    const double *bc_value_f;   // 9 planes (3f + i), see SweTracerArgs
    int bc_vel_kind[SWE_MAX_MARKERS];        // external velocity of the boundary dict, see SweTracerArgs
    int depth_mode;                          // tracer 'flux' boundaries: total depth rule (SweTracerArgs), with vh / valpha
    const double *valpha;
    double bc_u[SWE_MAX_MARKERS], bc_v[SWE_MAX_MARKERS];
};

// BND_ONLY: only the boundary-facet terms (cells taken from p.cell_list); the cell integral and the interior facets of the
// viscosity are then evaluated inside the stage kernel (swe_visc_interior).
template <int NC, bool BND_ONLY = false>
__global__ __launch_bounds__(SWE_BLOCK) void swe_sipg_kernel(const SweSipgArgs p)
{
    int k;
    if (BND_ONLY) {
        const int t = blockIdx.x*SWE_BLOCK + (int)threadIdx.x;
        if (t >= p.n_list) return;
        k = p.cell_list[t];
        if (k < p.cell_begin || k >= p.cell_end) return;
    } else {
        const int lb = swe_logical_block(blockIdx.x, gridDim.x);
        k = p.cell_begin + lb*SWE_BLOCK + (int)threadIdx.x;
        if (k >= p.cell_end) return;
    }
    const size_t S = p.stride;
    const bool gd = (NC == 2) && p.grad_div;
    // raw buffer addressing (swe_ld, swe2d_kernels.h): SGPR plane offsets, one 32-bit lane offset per gathered cell
    const unsigned S8 = (unsigned)S*8u, k8 = (unsigned)k*8u, S4 = (unsigned)S*4u, k4 = (unsigned)k*4u;
    const swe_rsrc_t rcv = swe_rsrc(p.cv), rvx = swe_rsrc(p.vx), rvy = swe_rsrc(p.vy);
    swe_rsrc_t rin[NC];
#pragma unroll
    for (int r = 0; r < NC; r++) rin[r] = swe_rsrc(p.in + (size_t)3*r*S);

    int nb[3], vid[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        nb[i] = swe_ldi(swe_rsrc(p.nbr), k4, i*S4);
        vid[i] = swe_ldi(rcv, k4, i*S4);
    }
    double c[NC][3];
#pragma unroll
    for (int r = 0; r < NC; r++)
#pragma unroll
        for (int i = 0; i < 3; i++) c[r][i] = swe_ld(rin[r], k8, i*S8);
    double px[3], py[3], mu[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        px[i] = swe_ld(rvx, (unsigned)vid[i]*8u, 0);
        py[i] = swe_ld(rvy, (unsigned)vid[i]*8u, 0);
        mu[i] = p.mu_v ? swe_ld(swe_rsrc(p.mu_v), (unsigned)vid[i]*8u, 0) : p.mu_const;
    }
    double nx[3], ny[3];
#pragma unroll
    for (int f = 0; f < 3; f++) {
        const int b = (f + 1) % 3;
        nx[f] = py[b] - py[f];
        ny[f] = px[f] - px[b];
    }
    const double twoA = nx[0]*ny[1] - ny[0]*nx[1];
    const double A = 0.5*twoA, r2A = swe_rcp(twoA);
    double gx[3], gy[3];                              // grad(phi_i) = -nF_{i+1}/(2A)
#pragma unroll
    for (int i = 0; i < 3; i++) {
        gx[i] = -nx[(i + 1) % 3]*r2A;
        gy[i] = -ny[(i + 1) % 3]*r2A;
    }
    // own gradient G[r][j] = d c_r / d x_j and the stress without its coefficient, S0 = G (+ G^T with grad-div)
    double G[NC][2], S0[NC][2];
#pragma unroll
    for (int r = 0; r < NC; r++) {
        G[r][0] = c[r][0]*gx[0] + c[r][1]*gx[1] + c[r][2]*gx[2];
        G[r][1] = c[r][0]*gy[0] + c[r][1]*gy[1] + c[r][2]*gy[2];
    }
#pragma unroll
    for (int r = 0; r < NC; r++)
#pragma unroll
        for (int j = 0; j < 2; j++) S0[r][j] = G[r][j] + ((NC == 2 && gd) ? G[j % NC][r] : 0.0);

    double b[NC][3];                                  // assembled residual R = -f
    {
        const double am = A*(mu[0] + mu[1] + mu[2])*(1.0/3.0);       // int mu dx
#pragma unroll
        for (int r = 0; r < NC; r++)
#pragma unroll
            for (int i = 0; i < 3; i++) b[r][i] = BND_ONLY ? 0.0 : -am*(gx[i]*S0[r][0] + gy[i]*S0[r][1]);     // inner(grad test, stress)*dx
    }
    double eo[3] = {0.0, 0.0, 0.0}, ho[3] = {0.0, 0.0, 0.0}, alo[3] = {0.0, 0.0, 0.0}, dno[3] = {0.0, 0.0, 0.0};
    if (NC == 2) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            ho[i] = swe_ld(swe_rsrc(p.vh), (unsigned)vid[i]*8u, 0);
            eo[i] = swe_ld(swe_rsrc(p.eta), k8, i*S8);
            if (p.wd) {                      // the planes hold the displaced depth D (swe2d_kernels.h, swe_wd_eta)
                alo[i] = swe_ld(swe_rsrc(p.valpha), (unsigned)vid[i]*8u, 0);
                dno[i] = eo[i];
                eo[i] = dno[i] - 0.25*alo[i]*alo[i]/dno[i] - ho[i];
            }
        }
    }
    if (NC == 2 && p.grad_depth && !BND_ONLY) {
        // -dot(test, dot(grad(H)/H, stress))*dx, shallowwater_eq.py:611-612; 6-point rule as the drag terms
        double Hn[3];
#pragma unroll
        for (int i = 0; i < 3; i++) Hn[i] = p.wd ? dno[i] : (p.nonlin ? ho[i] + eo[i] : ho[i]);
        const double gHx = Hn[0]*gx[0] + Hn[1]*gx[1] + Hn[2]*gx[2], gHy = Hn[0]*gy[0] + Hn[1]*gy[1] + Hn[2]*gy[2];
        double t[2];
#pragma unroll
        for (int r = 0; r < 2; r++) t[r] = gHx*S0[0][r % NC] + gHy*S0[1 % NC][r % NC];        // a_k S0[k][r]
        const double a1 = 0.445948490915965, b1 = 0.108103018168070, w1 = 0.223381589678011;
        const double a2 = 0.091576213509771, b2 = 0.816847572980459, w2 = 0.109951743655322;
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const double aa = q < 3 ? a1 : a2, bb = q < 3 ? b1 : b2, ww = q < 3 ? w1 : w2;
            double l[3] = {aa, aa, aa};
            l[q % 3] = bb;
            const double Hq = l[0]*Hn[0] + l[1]*Hn[1] + l[2]*Hn[2];
            const double muq = l[0]*mu[0] + l[1]*mu[1] + l[2]*mu[2];
            const double fac = ww*A*muq*swe_rcp(Hq);
#pragma unroll
            for (int r = 0; r < NC; r++)
#pragma unroll
                for (int i = 0; i < 3; i++) b[r][i] += fac*l[i]*t[r % 2];
        }
    }

#pragma unroll
    for (int f = 0; f < 3; f++) {
        const int a = f, bb = (f + 1) % 3;
        const double nxs = nx[f], nys = ny[f];
        double L, rL;
        swe_sqrt_rsqrt(nxs*nxs + nys*nys, L, rL);
        const double n0 = nxs*rL, n1 = nys*rL;
        const double w = 0.5*L;                                    // Gauss weight * facet length
        if (nb[f] >= 0) {
            if (BND_ONLY) continue;
            const int kn = nb[f] >> 2, f2 = nb[f] & 3;
            const int na = (f2 == 2) ? 0 : f2 + 1, no = (f2 == 0) ? 2 : f2 - 1;      // neighbour nodes on my a, opposite
            const unsigned sel_b = f2 == 0 ? 0u : (f2 == 1 ? 1u : 2u), sel_a = (unsigned)na, sel_o = (unsigned)no;
            const int vo = swe_ldi(rcv, (unsigned)kn*4u + (sel_o == 0 ? 0u : (sel_o == 1 ? S4 : 2u*S4)), 0);
            const unsigned kn8 = (unsigned)kn*8u;
            const unsigned oa = kn8 + (sel_a == 0 ? 0u : (sel_a == 1 ? S8 : 2u*S8));
            const unsigned ob = kn8 + (sel_b == 0 ? 0u : (sel_b == 1 ? S8 : 2u*S8));
            const unsigned oo = kn8 + (sel_o == 0 ? 0u : (sel_o == 1 ? S8 : 2u*S8));
            const double e1x = px[bb] - px[a], e1y = py[bb] - py[a];
            const double e2x = swe_ld(rvx, (unsigned)vo*8u, 0) - px[a], e2y = swe_ld(rvy, (unsigned)vo*8u, 0) - py[a];
            const double det = e1x*e2y - e1y*e2x;                  // -2 A_n (the neighbour lies to the right of a -> b)
            const double rdet = swe_rcp(det);
            const double An = 0.5*fabs(det);
            double ca[NC], cb[NC], S0n[NC][2], Gn[NC][2];
#pragma unroll
            for (int r = 0; r < NC; r++) {
                ca[r] = swe_ld(rin[r], oa, 0);
                cb[r] = swe_ld(rin[r], ob, 0);
                const double co = swe_ld(rin[r], oo, 0);
                const double d1 = cb[r] - ca[r], d2 = co - ca[r];
                Gn[r][0] = (d1*e2y - d2*e1y)*rdet;
                Gn[r][1] = (d2*e1x - d1*e2x)*rdet;
            }
#pragma unroll
            for (int r = 0; r < NC; r++)
#pragma unroll
                for (int j = 0; j < 2; j++) S0n[r][j] = Gn[r][j] + ((NC == 2 && gd) ? Gn[j % NC][r] : 0.0);
            const double sigma = p.sipg*L*swe_rcp(fmin(A, An));             // max over the two sides of sipg*cp*|F|/A
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const double xb = q ? SWE_XI1 : SWE_XI0, xa = 1.0 - xb;
                const double muq = xa*mu[a] + xb*mu[bb];
                double jmp[NC];
#pragma unroll
                for (int r = 0; r < NC; r++) jmp[r] = (xa*c[r][a] + xb*c[r][bb]) - (xa*ca[r] + xb*cb[r]);
                const double nn[2] = {n0, n1};
#pragma unroll
                for (int r = 0; r < NC; r++) {
                    // stress_jump[r][j] = mu (jmp_r n_j (+ jmp_j n_r)),  avg(stress)[r][j] = mu (S0 + S0n)/2
                    const double sj0 = muq*(jmp[r]*n0 + ((NC == 2 && gd) ? jmp[0]*nn[r] : 0.0));
                    const double sj1 = muq*(jmp[r]*n1 + ((NC == 2 && gd) ? jmp[1 % NC]*nn[r] : 0.0));
                    const double sjn = sj0*n0 + sj1*n1;
                    const double avn = 0.5*muq*((S0[r][0] + S0n[r][0])*n0 + (S0[r][1] + S0n[r][1])*n1);
                    const double val = sigma*sjn - avn;            // facet nodes only (test function trace)
                    b[r][a] -= w*xa*val;
                    b[r][bb] -= w*xb*val;
#pragma unroll
                    for (int i = 0; i < 3; i++) b[r][i] += w*0.5*(gx[i]*sj0 + gy[i]*sj1);   // -inner(avg(grad test), stress_jump)
                }
            }
        } else {
            const int marker = -nb[f];
            if (marker >= SWE_MAX_MARKERS) continue;
            if (NC == 2) {
                // Dirichlet terms where the boundary condition defines an external velocity, shallowwater_eq.py:584-609
                const int kind = p.bc.kind[marker];
                if (!(kind & (SWE_BC_UN | SWE_BC_UV | SWE_BC_FLUX))) continue;
                const double sigma = p.sipg*L/A;
                double fua = 0.0, fub = 0.0, fva = 0.0, fvb = 0.0, fna = 0.0, fnb = 0.0, fea = 0.0, feb = 0.0, fxa = 0.0, fxb = 0.0;
                const size_t pa = (size_t)(2*a)*S + k, pb = pa + S;
                if ((kind & SWE_BC_UV_FIELD) && p.bc_uv_f) {
                    fua = p.bc_uv_f[pa]; fub = p.bc_uv_f[pb];
                    fva = p.bc_uv_f[6*S + pa]; fvb = p.bc_uv_f[6*S + pb];
                }
                if ((kind & SWE_BC_UN_FIELD) && p.bc_un_f) { fna = p.bc_un_f[pa]; fnb = p.bc_un_f[pb]; }
                if ((kind & SWE_BC_ELEV_FIELD) && p.bc_elev_f) { fea = p.bc_elev_f[pa]; feb = p.bc_elev_f[pb]; }
                if ((kind & SWE_BC_FLUX_FIELD) && p.bc_flux_f) { fxa = p.bc_flux_f[pa]; fxb = p.bc_flux_f[pb]; }
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const double xb = q ? SWE_XI1 : SWE_XI0, xa = 1.0 - xb;
                    const double muq = xa*mu[a] + xb*mu[bb];
                    const double uq = xa*c[0][a] + xb*c[0][bb], vq = xa*c[1 % NC][a] + xb*c[1 % NC][bb];
                    double dlt[2];
                    if (kind & SWE_BC_UN) {
                        const double un_ext = (kind & SWE_BC_UN_FIELD) ? xa*fna + xb*fnb : p.bc.un[marker];
                        const double d = uq*n0 + vq*n1 - un_ext;
                        dlt[0] = d*n0; dlt[1] = d*n1;
                    } else if (kind & SWE_BC_UV) {
                        dlt[0] = uq - ((kind & SWE_BC_UV_FIELD) ? xa*fua + xb*fub : p.bc.u[marker]);
                        dlt[1] = vq - ((kind & SWE_BC_UV_FIELD) ? xa*fva + xb*fvb : p.bc.v[marker]);
                    } else {
                        const double eq = xa*eo[a] + xb*eo[bb], hq = xa*ho[a] + xb*ho[bb];
                        const double e_ext = (kind & SWE_BC_ELEV) ? ((kind & SWE_BC_ELEV_FIELD) ? xa*fea + xb*feb : p.bc.elev[marker]) : eq;
                        const double H0 = p.wd ? swe_wd_depth(hq + e_ext, xa*alo[a] + xb*alo[bb]) : (p.nonlin ? hq + e_ext : hq);
                        const double s = ((kind & SWE_BC_FLUX_FIELD) ? xa*fxa + xb*fxb : p.bc.flux[marker])/(H0*p.bc.len[marker]);
                        dlt[0] = uq - s*n0; dlt[1] = vq - s*n1;
                    }
                    const double nn[2] = {n0, n1};
#pragma unroll
                    for (int r = 0; r < NC; r++) {
                        const double sj0 = muq*(dlt[r % 2]*n0 + (gd ? dlt[0]*nn[r % 2] : 0.0));
                        const double sj1 = muq*(dlt[r % 2]*n1 + (gd ? dlt[1]*nn[r % 2] : 0.0));
                        const double val = sigma*(sj0*n0 + sj1*n1) - muq*(S0[r][0]*n0 + S0[r][1]*n1);
                        b[r][a] -= w*xa*val;
                        b[r][bb] -= w*xb*val;
#pragma unroll
                        for (int i = 0; i < 3; i++) b[r][i] += w*(gx[i]*sj0 + gy[i]*sj1);    // -inner(grad test, stress_jump)
                    }
                }
            } else {
                const int kd = p.bc_diff_kind[marker];
                if (kd == SWE_SIPG_BC_NONE) continue;
                double ua = 0.0, ub = 0.0, va = 0.0, vb = 0.0, gex = 0.0, gey = 0.0;
                if (kd == SWE_SIPG_BC_UPWIND || kd == SWE_SIPG_BC_VALUE_FIELD) {
                    ua = p.vel_factor*p.uv[(size_t)a*S + k]; ub = p.vel_factor*p.uv[(size_t)bb*S + k];
                    va = p.vel_factor*p.uv[(size_t)(3 + a)*S + k]; vb = p.vel_factor*p.uv[(size_t)(3 + bb)*S + k];
                }
                if (kd == SWE_SIPG_BC_VALUE_FIELD) {               // cell gradient of the boundary Function
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        const double ce = p.bc_value_f[(size_t)(3*f + i)*S + k];
                        gex += ce*gx[i];
                        gey += ce*gy[i];
                    }
                }
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const double xb = q ? SWE_XI1 : SWE_XI0, xa = 1.0 - xb;
                    double val;
                    if (kd == SWE_SIPG_BC_DIFF_FLUX) {
                        val = -p.bc_diff_flux[marker];
                    } else {
                        const double muq = xa*mu[a] + xb*mu[bb];
                        double uq = xa*ua + xb*ub, vq = xa*va + xb*vb, ue = uq, ve = vq;
                        if (p.bc_vel_kind[marker] == 1) { ue = p.vel_factor*p.bc_u[marker]; ve = p.vel_factor*p.bc_v[marker]; }
                        else if (p.bc_vel_kind[marker] == 2) { ue = p.bc_u[marker]*n0; ve = p.bc_u[marker]*n1; }
                        else if (p.bc_vel_kind[marker] >= 3) {                               // 'flux'
                            const unsigned va8 = (unsigned)vid[a]*8u, vb8 = (unsigned)vid[bb]*8u;
                            const double hq = xa*swe_ld(swe_rsrc(p.vh), va8, 0) + xb*swe_ld(swe_rsrc(p.vh), vb8, 0);
                            const double alq = p.depth_mode == 2
                                ? xa*swe_ld(swe_rsrc(p.valpha), va8, 0) + xb*swe_ld(swe_rsrc(p.valpha), vb8, 0) : 0.0;
                            double ea_ = p.uv[(size_t)(6 + a)*S + k], eb_ = p.uv[(size_t)(6 + bb)*S + k];
                            if (p.depth_mode == 2) {           // the planes hold D: the nodal elevations by the closed form
                                const double aa_ = swe_ld(swe_rsrc(p.valpha), va8, 0), ab_ = swe_ld(swe_rsrc(p.valpha), vb8, 0);
                                ea_ = ea_ - 0.25*aa_*aa_/ea_ - swe_ld(swe_rsrc(p.vh), va8, 0);
                                eb_ = eb_ - 0.25*ab_*ab_/eb_ - swe_ld(swe_rsrc(p.vh), vb8, 0);
                            }
                            const double eq = xa*ea_ + xb*eb_;
                            const double sp = swe_tracer_flux_speed(p.depth_mode, hq, eq, alq, p.bc_vel_kind[marker] == 4,
                                                                    p.bc_v[marker], p.bc_u[marker], p.bc.len[marker], p.vel_factor);
                            ue = sp*n0; ve = sp*n1;
                        }
                        const double un = 0.5*((uq + ue)*n0 + (vq + ve)*n1);                 // uv_av . n
                        const double s = (kd == SWE_SIPG_BC_GRAD_IN) ? 1.0 : (un > 0.0 ? 1.0 : (un < 0.0 ? 0.0 : 0.5));
                        val = -muq*((s*G[0][0] + (1.0 - s)*gex)*n0 + (s*G[0][1] + (1.0 - s)*gey)*n1);
                    }
                    b[0][a] -= w*xa*val;
                    b[0][bb] -= w*xb*val;
                }
            }
        }
    }
    const double s = 6.0*p.dt*p.beta*r2A;
#pragma unroll
    for (int r = 0; r < NC; r++) {
        const double sb = b[r][0] + b[r][1] + b[r][2];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const swe_rsrc_t ro = swe_rsrc(p.out + (size_t)3*r*S);
            swe_st(ro, k8, i*S8, swe_ld(ro, k8, i*S8) + s*(4.0*b[r][i] - sb));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same operators on parallelogram quadrilaterals (DQ-1): cp = (p+1)^2 = 4; the gradients of the bilinear basis vary
// along a facet, so both sides' gradients are evaluated at the facet quadrature points through the affine maps
// x = p0 + xi a + zeta b of the cell and of the neighbour (whose two far vertices are gathered).  2 x 2 Gauss rule in the
// cell (exact for the polynomial integrands), tensor mass inverse.  Boundary-field planes: 2*4 per component.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void swe_q1_basis(double xi, double ze, double gxi_x, double gxi_y, double gze_x, double gze_y,
                                             double phi[4], double gx[4], double gy[4])
{
    const double dxi[4] = {-(1.0 - ze), (1.0 - ze), ze, -ze};
    const double dze[4] = {-(1.0 - xi), -xi, xi, (1.0 - xi)};
    phi[0] = (1.0 - xi)*(1.0 - ze); phi[1] = xi*(1.0 - ze); phi[2] = xi*ze; phi[3] = (1.0 - xi)*ze;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        gx[i] = dxi[i]*gxi_x + dze[i]*gze_x;
        gy[i] = dxi[i]*gxi_y + dze[i]*gze_y;
    }
}

// grad(xi), grad(zeta) at the reference point (xi, zeta) of a general quadrilateral x = p0 + xi a + zeta b + xi zeta c (see
// swe_quad_mass in swe2d_kernels.h): J^-1 = adj(J)/det J with x_xi = a + zeta c, x_zeta = b + xi c.  Returns det J.
__device__ __forceinline__ double swe_q1_map_grads(double ax, double ay, double bx, double by, double cx, double cy, double xi,
                                                   double ze, double &gxi_x, double &gxi_y, double &gze_x, double &gze_y)
{
    const double xxi = ax + cx*ze, yxi = ay + cy*ze, xze = bx + cx*xi, yze = by + cy*xi;
    const double det = xxi*yze - yxi*xze, rd = swe_rcp(det);
    gxi_x = yze*rd; gxi_y = -xze*rd; gze_x = -yxi*rd; gze_y = xxi*rd;
    return det;
}

// AFFINE = false: general quadrilaterals - the gradients through the Jacobian at every quadrature point (cell and facets, both
// sides), det J at the point as the cell weight, the true cell areas in the penalty, the 4 x 4 mass solve of swe_quad_mass.
template <int NC, bool AFFINE = true>
__global__ __launch_bounds__(SWE_BLOCK) void swe_sipg_kernel_quad(const SweSipgArgs p)
{
    const int lb = swe_logical_block(blockIdx.x, gridDim.x);
    const int k = p.cell_begin + lb*SWE_BLOCK + (int)threadIdx.x;
    if (k >= p.cell_end) return;
    const size_t S = p.stride;
    const bool gd = (NC == 2) && p.grad_div;
    const unsigned S8 = (unsigned)S*8u, k8 = (unsigned)k*8u, S4 = (unsigned)S*4u, k4 = (unsigned)k*4u;    // see swe_ld
    const swe_rsrc_t rcv = swe_rsrc(p.cv), rvx = swe_rsrc(p.vx), rvy = swe_rsrc(p.vy);
    swe_rsrc_t rin[NC];
#pragma unroll
    for (int r = 0; r < NC; r++) rin[r] = swe_rsrc(p.in + (size_t)4*r*S);
    const double RX[4] = {0.0, 1.0, 1.0, 0.0}, RZ[4] = {0.0, 0.0, 1.0, 1.0};      // reference corners

    int nb[4], vid[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        nb[i] = swe_ldi(swe_rsrc(p.nbr), k4, i*S4);
        vid[i] = swe_ldi(rcv, k4, i*S4);
    }
    double c[NC][4];
#pragma unroll
    for (int r = 0; r < NC; r++)
#pragma unroll
        for (int i = 0; i < 4; i++) c[r][i] = swe_ld(rin[r], k8, i*S8);
    double px[4], py[4], mu[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        px[i] = swe_ld(rvx, (unsigned)vid[i]*8u, 0);
        py[i] = swe_ld(rvy, (unsigned)vid[i]*8u, 0);
        mu[i] = p.mu_v ? swe_ld(swe_rsrc(p.mu_v), (unsigned)vid[i]*8u, 0) : p.mu_const;
    }
    const double ax = px[1] - px[0], ay = py[1] - py[0], bx = px[3] - px[0], by = py[3] - py[0];
    const double A0 = ax*by - ay*bx, rA = swe_rcp(A0);
    double gxi_x = by*rA, gxi_y = -bx*rA, gze_x = -ay*rA, gze_y = ax*rA;     // grad(xi), grad(zeta) (AFFINE: everywhere in the cell)
    const double cx = AFFINE ? 0.0 : (px[0] - px[1]) + (px[2] - px[3]), cy = AFFINE ? 0.0 : (py[0] - py[1]) + (py[2] - py[3]);
    const double d1 = AFFINE ? 0.0 : ax*cy - ay*cx, d2 = AFFINE ? 0.0 : cx*by - cy*bx;
    const double A = AFFINE ? A0 : A0 + 0.5*(d1 + d2);                       // cell area
    double eo[4] = {0.0, 0.0, 0.0, 0.0}, ho[4] = {0.0, 0.0, 0.0, 0.0}, alo[4] = {0.0, 0.0, 0.0, 0.0}, dno[4] = {0.0, 0.0, 0.0, 0.0};
    if (NC == 2) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            ho[i] = swe_ld(swe_rsrc(p.vh), (unsigned)vid[i]*8u, 0);
            eo[i] = swe_ld(swe_rsrc(p.eta), k8, i*S8);
            if (p.wd) {                      // the planes hold the displaced depth D (swe2d_kernels.h, swe_wd_eta)
                alo[i] = swe_ld(swe_rsrc(p.valpha), (unsigned)vid[i]*8u, 0);
                dno[i] = eo[i];
                eo[i] = dno[i] - 0.25*alo[i]*alo[i]/dno[i] - ho[i];
            }
        }
    }
    double b[NC][4];
#pragma unroll
    for (int r = 0; r < NC; r++)
#pragma unroll
        for (int i = 0; i < 4; i++) b[r][i] = 0.0;

    // ---- cell integrals, 2 x 2 Gauss-Legendre (weight A/4)
#pragma unroll
    for (int qi = 0; qi < 2; qi++) {
#pragma unroll
        for (int qz = 0; qz < 2; qz++) {
            double phi[4], gx[4], gy[4];
            double Aq = A;
            if (!AFFINE) Aq = swe_q1_map_grads(ax, ay, bx, by, cx, cy, qi ? SWE_XI1 : SWE_XI0, qz ? SWE_XI1 : SWE_XI0, gxi_x, gxi_y, gze_x, gze_y);
            swe_q1_basis(qi ? SWE_XI1 : SWE_XI0, qz ? SWE_XI1 : SWE_XI0, gxi_x, gxi_y, gze_x, gze_y, phi, gx, gy);
            double muq = 0.0, Hq = 0.0, gHx = 0.0, gHy = 0.0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                muq += phi[i]*mu[i];
                if (NC == 2 && p.grad_depth) {
                    const double Hn = p.wd ? dno[i] : (p.nonlin ? ho[i] + eo[i] : ho[i]);
                    Hq += phi[i]*Hn; gHx += gx[i]*Hn; gHy += gy[i]*Hn;
                }
            }
            double G[NC][2], S0[NC][2];
#pragma unroll
            for (int r = 0; r < NC; r++) {
                G[r][0] = 0.0; G[r][1] = 0.0;
#pragma unroll
                for (int i = 0; i < 4; i++) { G[r][0] += c[r][i]*gx[i]; G[r][1] += c[r][i]*gy[i]; }
            }
#pragma unroll
            for (int r = 0; r < NC; r++)
#pragma unroll
                for (int j = 0; j < 2; j++) S0[r][j] = G[r][j] + ((NC == 2 && gd) ? G[j % NC][r] : 0.0);
            const double w = 0.25*Aq*muq;
            const double rHq = (NC == 2 && p.grad_depth) ? swe_rcp(Hq) : 0.0;
#pragma unroll
            for (int r = 0; r < NC; r++) {
                const double tr = (NC == 2 && p.grad_depth) ? (gHx*S0[0][r % NC] + gHy*S0[1 % NC][r % NC])*rHq : 0.0;
#pragma unroll
                for (int i = 0; i < 4; i++) b[r][i] += w*(phi[i]*tr - (gx[i]*S0[r][0] + gy[i]*S0[r][1]));
            }
        }
    }

    // ---- facets
#pragma unroll
    for (int f = 0; f < 4; f++) {
        const int a = f, bb = (f + 1) & 3;
        const double nxs = py[bb] - py[a], nys = px[a] - px[bb];
        double L, rL;
        swe_sqrt_rsqrt(nxs*nxs + nys*nys, L, rL);
        const double n0 = nxs*rL, n1 = nys*rL;
        const double w = 0.5*L;
        const double nn[2] = {n0, n1};
        if (nb[f] >= 0) {
            const int kn = nb[f] >> 2, f2 = nb[f] & 3;
            const int na = (f2 + 1) & 3, n2 = (f2 + 2) & 3, n3 = (f2 + 3) & 3;   // neighbour: na on my a, f2 on my bb
            // neighbour geometry from its four vertices (two shared, two gathered), indexed by ITS local numbering
            double qx[4], qy[4], cn[NC][4];
            qx[na] = px[a]; qy[na] = py[a]; qx[f2] = px[bb]; qy[f2] = py[bb];
            {
                const unsigned kn4 = (unsigned)kn*4u;
                const int v2 = swe_ldi(rcv, kn4 + ((n2 & 1) ? S4 : 0u) + ((n2 & 2) ? 2u*S4 : 0u), 0);
                const int v3 = swe_ldi(rcv, kn4 + ((n3 & 1) ? S4 : 0u) + ((n3 & 2) ? 2u*S4 : 0u), 0);
                qx[n2] = swe_ld(rvx, (unsigned)v2*8u, 0); qy[n2] = swe_ld(rvy, (unsigned)v2*8u, 0);
                qx[n3] = swe_ld(rvx, (unsigned)v3*8u, 0); qy[n3] = swe_ld(rvy, (unsigned)v3*8u, 0);
            }
#pragma unroll
            for (int r = 0; r < NC; r++)
#pragma unroll
                for (int i = 0; i < 4; i++) cn[r][i] = swe_ld(rin[r], (unsigned)kn*8u, i*S8);
            const double anx = qx[1] - qx[0], any_ = qy[1] - qy[0], bnx = qx[3] - qx[0], bny = qy[3] - qy[0];
            const double An0 = anx*bny - any_*bnx, rAn = swe_rcp(An0);
            double hxi_x = bny*rAn, hxi_y = -bnx*rAn, hze_x = -any_*rAn, hze_y = anx*rAn;
            const double cnx = AFFINE ? 0.0 : (qx[0] - qx[1]) + (qx[2] - qx[3]), cny = AFFINE ? 0.0 : (qy[0] - qy[1]) + (qy[2] - qy[3]);
            const double An = AFFINE ? An0 : An0 + 0.5*((anx*cny - any_*cnx) + (cnx*bny - cny*bnx));
            const double sigma = p.sipg*L*swe_rcp(fmin(A, An));
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const double s = q ? SWE_XI1 : SWE_XI0;
                double phi[4], gx[4], gy[4], phn[4], hx[4], hy[4];
                if (!AFFINE) {
                    swe_q1_map_grads(ax, ay, bx, by, cx, cy, (1.0 - s)*RX[a] + s*RX[bb], (1.0 - s)*RZ[a] + s*RZ[bb], gxi_x, gxi_y, gze_x, gze_y);
                    swe_q1_map_grads(anx, any_, bnx, bny, cnx, cny, (1.0 - s)*RX[na] + s*RX[f2], (1.0 - s)*RZ[na] + s*RZ[f2], hxi_x, hxi_y,
                                     hze_x, hze_y);
                }
                swe_q1_basis((1.0 - s)*RX[a] + s*RX[bb], (1.0 - s)*RZ[a] + s*RZ[bb], gxi_x, gxi_y, gze_x, gze_y, phi, gx, gy);
                swe_q1_basis((1.0 - s)*RX[na] + s*RX[f2], (1.0 - s)*RZ[na] + s*RZ[f2], hxi_x, hxi_y, hze_x, hze_y, phn, hx, hy);
                const double muq = (1.0 - s)*mu[a] + s*mu[bb];
                double jmp[NC], S0[NC][2], S0n[NC][2], G[NC][2], Gn[NC][2];
#pragma unroll
                for (int r = 0; r < NC; r++) {
                    jmp[r] = ((1.0 - s)*c[r][a] + s*c[r][bb]) - ((1.0 - s)*cn[r][na] + s*cn[r][f2]);
                    G[r][0] = G[r][1] = Gn[r][0] = Gn[r][1] = 0.0;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        G[r][0] += c[r][i]*gx[i]; G[r][1] += c[r][i]*gy[i];
                        Gn[r][0] += cn[r][i]*hx[i]; Gn[r][1] += cn[r][i]*hy[i];
                    }
                }
#pragma unroll
                for (int r = 0; r < NC; r++)
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        S0[r][j] = G[r][j] + ((NC == 2 && gd) ? G[j % NC][r] : 0.0);
                        S0n[r][j] = Gn[r][j] + ((NC == 2 && gd) ? Gn[j % NC][r] : 0.0);
                    }
#pragma unroll
                for (int r = 0; r < NC; r++) {
                    const double sj0 = muq*(jmp[r]*n0 + ((NC == 2 && gd) ? jmp[0]*nn[r] : 0.0));
                    const double sj1 = muq*(jmp[r]*n1 + ((NC == 2 && gd) ? jmp[1 % NC]*nn[r] : 0.0));
                    const double avn = 0.5*muq*((S0[r][0] + S0n[r][0])*n0 + (S0[r][1] + S0n[r][1])*n1);
                    const double val = sigma*(sj0*n0 + sj1*n1) - avn;
                    b[r][a] -= w*(1.0 - s)*val;
                    b[r][bb] -= w*s*val;
#pragma unroll
                    for (int i = 0; i < 4; i++) b[r][i] += w*0.5*(gx[i]*sj0 + gy[i]*sj1);
                }
            }
        } else {
            const int marker = -nb[f];
            if (marker >= SWE_MAX_MARKERS) continue;
            const size_t pa = (size_t)(2*a)*S + k, pb = pa + S;              // per-facet boundary-field planes
            if (NC == 2) {
                const int kind = p.bc.kind[marker];
                if (!(kind & (SWE_BC_UN | SWE_BC_UV | SWE_BC_FLUX))) continue;
                const double sigma = p.sipg*L/A;
                double fua = 0.0, fub = 0.0, fva = 0.0, fvb = 0.0, fna = 0.0, fnb = 0.0, fea = 0.0, feb = 0.0, fxa = 0.0, fxb = 0.0;
                if ((kind & SWE_BC_UV_FIELD) && p.bc_uv_f) {
                    fua = p.bc_uv_f[pa]; fub = p.bc_uv_f[pb];
                    fva = p.bc_uv_f[8*S + pa]; fvb = p.bc_uv_f[8*S + pb];
                }
                if ((kind & SWE_BC_UN_FIELD) && p.bc_un_f) { fna = p.bc_un_f[pa]; fnb = p.bc_un_f[pb]; }
                if ((kind & SWE_BC_ELEV_FIELD) && p.bc_elev_f) { fea = p.bc_elev_f[pa]; feb = p.bc_elev_f[pb]; }
                if ((kind & SWE_BC_FLUX_FIELD) && p.bc_flux_f) { fxa = p.bc_flux_f[pa]; fxb = p.bc_flux_f[pb]; }
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const double s = q ? SWE_XI1 : SWE_XI0, xa = 1.0 - s, xb = s;
                    double phi[4], gx[4], gy[4];
                    if (!AFFINE) swe_q1_map_grads(ax, ay, bx, by, cx, cy, xa*RX[a] + xb*RX[bb], xa*RZ[a] + xb*RZ[bb], gxi_x, gxi_y, gze_x, gze_y);
                    swe_q1_basis(xa*RX[a] + xb*RX[bb], xa*RZ[a] + xb*RZ[bb], gxi_x, gxi_y, gze_x, gze_y, phi, gx, gy);
                    const double muq = xa*mu[a] + xb*mu[bb];
                    const double uq = xa*c[0][a] + xb*c[0][bb], vq = xa*c[1 % NC][a] + xb*c[1 % NC][bb];
                    double dlt[2];
                    if (kind & SWE_BC_UN) {
                        const double un_ext = (kind & SWE_BC_UN_FIELD) ? xa*fna + xb*fnb : p.bc.un[marker];
                        const double d = uq*n0 + vq*n1 - un_ext;
                        dlt[0] = d*n0; dlt[1] = d*n1;
                    } else if (kind & SWE_BC_UV) {
                        dlt[0] = uq - ((kind & SWE_BC_UV_FIELD) ? xa*fua + xb*fub : p.bc.u[marker]);
                        dlt[1] = vq - ((kind & SWE_BC_UV_FIELD) ? xa*fva + xb*fvb : p.bc.v[marker]);
                    } else {
                        const double eq = xa*eo[a] + xb*eo[bb], hq = xa*ho[a] + xb*ho[bb];
                        const double e_ext = (kind & SWE_BC_ELEV) ? ((kind & SWE_BC_ELEV_FIELD) ? xa*fea + xb*feb : p.bc.elev[marker]) : eq;
                        const double H0 = p.wd ? swe_wd_depth(hq + e_ext, xa*alo[a] + xb*alo[bb]) : (p.nonlin ? hq + e_ext : hq);
                        const double sc = ((kind & SWE_BC_FLUX_FIELD) ? xa*fxa + xb*fxb : p.bc.flux[marker])/(H0*p.bc.len[marker]);
                        dlt[0] = uq - sc*n0; dlt[1] = vq - sc*n1;
                    }
                    double G[NC][2];
#pragma unroll
                    for (int r = 0; r < NC; r++) {
                        G[r][0] = G[r][1] = 0.0;
#pragma unroll
                        for (int i = 0; i < 4; i++) { G[r][0] += c[r][i]*gx[i]; G[r][1] += c[r][i]*gy[i]; }
                    }
#pragma unroll
                    for (int r = 0; r < NC; r++) {
                        const double s00 = G[r][0] + (gd ? G[0][r % 2] : 0.0), s01 = G[r][1] + (gd ? G[1 % NC][r % 2] : 0.0);
                        const double sj0 = muq*(dlt[r % 2]*n0 + (gd ? dlt[0]*nn[r % 2] : 0.0));
                        const double sj1 = muq*(dlt[r % 2]*n1 + (gd ? dlt[1]*nn[r % 2] : 0.0));
                        const double val = sigma*(sj0*n0 + sj1*n1) - muq*(s00*n0 + s01*n1);
                        b[r][a] -= w*xa*val;
                        b[r][bb] -= w*xb*val;
#pragma unroll
                        for (int i = 0; i < 4; i++) b[r][i] += w*(gx[i]*sj0 + gy[i]*sj1);
                    }
                }
            } else {
                const int kd = p.bc_diff_kind[marker];
                if (kd == SWE_SIPG_BC_NONE) continue;
                double ua = 0.0, ub = 0.0, va = 0.0, vb = 0.0, ce[4] = {0.0, 0.0, 0.0, 0.0};
                if (kd == SWE_SIPG_BC_UPWIND || kd == SWE_SIPG_BC_VALUE_FIELD) {
                    ua = p.vel_factor*p.uv[(size_t)a*S + k]; ub = p.vel_factor*p.uv[(size_t)bb*S + k];
                    va = p.vel_factor*p.uv[(size_t)(4 + a)*S + k]; vb = p.vel_factor*p.uv[(size_t)(4 + bb)*S + k];
                }
                if (kd == SWE_SIPG_BC_VALUE_FIELD) {
#pragma unroll
                    for (int i = 0; i < 4; i++) ce[i] = p.bc_value_f[(size_t)(4*f + i)*S + k];
                }
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const double s = q ? SWE_XI1 : SWE_XI0, xa = 1.0 - s, xb = s;
                    double val;
                    if (kd == SWE_SIPG_BC_DIFF_FLUX) {
                        val = -p.bc_diff_flux[marker];
                    } else {
                        double phi[4], gx[4], gy[4];
                        if (!AFFINE) swe_q1_map_grads(ax, ay, bx, by, cx, cy, xa*RX[a] + xb*RX[bb], xa*RZ[a] + xb*RZ[bb], gxi_x, gxi_y, gze_x, gze_y);
                        swe_q1_basis(xa*RX[a] + xb*RX[bb], xa*RZ[a] + xb*RZ[bb], gxi_x, gxi_y, gze_x, gze_y, phi, gx, gy);
                        double g0 = 0.0, g1 = 0.0, e0_ = 0.0, e1_ = 0.0;
#pragma unroll
                        for (int i = 0; i < 4; i++) { g0 += c[0][i]*gx[i]; g1 += c[0][i]*gy[i]; e0_ += ce[i]*gx[i]; e1_ += ce[i]*gy[i]; }
                        const double muq = xa*mu[a] + xb*mu[bb];
                        double uq = xa*ua + xb*ub, vq = xa*va + xb*vb, ue = uq, ve = vq;
                        if (p.bc_vel_kind[marker] == 1) { ue = p.vel_factor*p.bc_u[marker]; ve = p.vel_factor*p.bc_v[marker]; }
                        else if (p.bc_vel_kind[marker] == 2) { ue = p.bc_u[marker]*n0; ve = p.bc_u[marker]*n1; }
                        else if (p.bc_vel_kind[marker] >= 3) {                               // 'flux'
                            const unsigned va8 = (unsigned)vid[a]*8u, vb8 = (unsigned)vid[bb]*8u;
                            const double hq = xa*swe_ld(swe_rsrc(p.vh), va8, 0) + xb*swe_ld(swe_rsrc(p.vh), vb8, 0);
                            const double alq = p.depth_mode == 2
                                ? xa*swe_ld(swe_rsrc(p.valpha), va8, 0) + xb*swe_ld(swe_rsrc(p.valpha), vb8, 0) : 0.0;
                            double ea_ = p.uv[(size_t)(8 + a)*S + k], eb_ = p.uv[(size_t)(8 + bb)*S + k];
                            if (p.depth_mode == 2) {           // the planes hold D: the nodal elevations by the closed form
                                const double aa_ = swe_ld(swe_rsrc(p.valpha), va8, 0), ab_ = swe_ld(swe_rsrc(p.valpha), vb8, 0);
                                ea_ = ea_ - 0.25*aa_*aa_/ea_ - swe_ld(swe_rsrc(p.vh), va8, 0);
                                eb_ = eb_ - 0.25*ab_*ab_/eb_ - swe_ld(swe_rsrc(p.vh), vb8, 0);
                            }
                            const double eq = xa*ea_ + xb*eb_;
                            const double sp = swe_tracer_flux_speed(p.depth_mode, hq, eq, alq, p.bc_vel_kind[marker] == 4,
                                                                    p.bc_v[marker], p.bc_u[marker], p.bc.len[marker], p.vel_factor);
                            ue = sp*n0; ve = sp*n1;
                        }
                        const double un = 0.5*((uq + ue)*n0 + (vq + ve)*n1);
                        const double sw = (kd == SWE_SIPG_BC_GRAD_IN) ? 1.0 : (un > 0.0 ? 1.0 : (un < 0.0 ? 0.0 : 0.5));
                        val = -muq*((sw*g0 + (1.0 - sw)*e0_)*n0 + (sw*g1 + (1.0 - sw)*e1_)*n1);
                    }
                    b[0][a] -= w*xa*val;
                    b[0][bb] -= w*xb*val;
                }
            }
        }
    }
    // ---- tensor mass inverse (M^-1 b)_i = (16 b_i - 8 b_{i+1} - 8 b_{i-1} + 4 b_{i+2})/A
    if constexpr (AFFINE) {
    const double sc = p.dt*p.beta*rA;
#pragma unroll
    for (int r = 0; r < NC; r++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const swe_rsrc_t ro = swe_rsrc(p.out + (size_t)4*r*S);
            swe_st(ro, k8, i*S8, swe_ld(ro, k8, i*S8)
                   + sc*(16.0*b[r][i] - 8.0*b[r][(i + 1) & 3] - 8.0*b[r][(i + 3) & 3] + 4.0*b[r][(i + 2) & 3]));
        }
    } else {
    SweQuadMass M;
    SweQuadLDL F;
    swe_quad_mass(A0, d1, d2, M);
    swe_quad_mass_factor(M, F);
    const double sc = p.dt*p.beta;
#pragma unroll
    for (int r = 0; r < NC; r++) {
        swe_quad_mass_solve(F, b[r]);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const swe_rsrc_t ro = swe_rsrc(p.out + (size_t)4*r*S);
            swe_st(ro, k8, i*S8, swe_ld(ro, k8, i*S8) + sc*b[r][i]);
        }
    }
    }
}
