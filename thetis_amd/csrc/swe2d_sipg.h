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
    // viscosity only
    int grad_div, grad_depth, nonlin;
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
    double bc_u[SWE_MAX_MARKERS], bc_v[SWE_MAX_MARKERS];
};

template <int NC>
__global__ __launch_bounds__(SWE_BLOCK) void swe_sipg_kernel(const SweSipgArgs p)
{
    const int lb = swe_logical_block(blockIdx.x, gridDim.x);
    const int k = p.cell_begin + lb*SWE_BLOCK + (int)threadIdx.x;
    if (k >= p.cell_end) return;
    const size_t S = p.stride;
    const bool gd = (NC == 2) && p.grad_div;

    int nb[3], vid[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        nb[i] = p.nbr[(size_t)i*S + k];
        vid[i] = p.cv[(size_t)i*S + k];
    }
    double c[NC][3];
#pragma unroll
    for (int r = 0; r < NC; r++)
#pragma unroll
        for (int i = 0; i < 3; i++) c[r][i] = p.in[(size_t)(3*r + i)*S + k];
    double px[3], py[3], mu[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        px[i] = p.vx[vid[i]];
        py[i] = p.vy[vid[i]];
        mu[i] = p.mu_v ? p.mu_v[vid[i]] : p.mu_const;
    }
    double nx[3], ny[3];
#pragma unroll
    for (int f = 0; f < 3; f++) {
        const int b = (f + 1) % 3;
        nx[f] = py[b] - py[f];
        ny[f] = px[f] - px[b];
    }
    const double twoA = nx[0]*ny[1] - ny[0]*nx[1];
    const double A = 0.5*twoA, r2A = 1.0/twoA;
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
            for (int i = 0; i < 3; i++) b[r][i] = -am*(gx[i]*S0[r][0] + gy[i]*S0[r][1]);     // inner(grad test, stress)*dx
    }
    double eo[3] = {0.0, 0.0, 0.0}, ho[3] = {0.0, 0.0, 0.0};
    if (NC == 2) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            ho[i] = p.vh[vid[i]];
            eo[i] = p.eta[(size_t)i*S + k];
        }
    }
    if (NC == 2 && p.grad_depth) {
        // -dot(test, dot(grad(H)/H, stress))*dx, shallowwater_eq.py:611-612; 6-point rule as the drag terms
        double Hn[3];
#pragma unroll
        for (int i = 0; i < 3; i++) Hn[i] = p.nonlin ? ho[i] + eo[i] : ho[i];
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
            const double fac = ww*A*muq/Hq;
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
        const double L = sqrt(nxs*nxs + nys*nys);
        const double n0 = nxs/L, n1 = nys/L;
        const double w = 0.5*L;                                    // Gauss weight * facet length
        if (nb[f] >= 0) {
            const int kn = nb[f] >> 2, f2 = nb[f] & 3;
            const int na = (f2 == 2) ? 0 : f2 + 1, no = (f2 == 0) ? 2 : f2 - 1;      // neighbour nodes on my a, opposite
            const int vo = p.cv[(size_t)no*S + kn];
            const double e1x = px[bb] - px[a], e1y = py[bb] - py[a];
            const double e2x = p.vx[vo] - px[a], e2y = p.vy[vo] - py[a];
            const double det = e1x*e2y - e1y*e2x;                  // -2 A_n (the neighbour lies to the right of a -> b)
            const double rdet = 1.0/det;
            const double An = 0.5*fabs(det);
            double ca[NC], cb[NC], S0n[NC][2], Gn[NC][2];
#pragma unroll
            for (int r = 0; r < NC; r++) {
                ca[r] = p.in[(size_t)(3*r + na)*S + kn];
                cb[r] = p.in[(size_t)(3*r + f2)*S + kn];
                const double co = p.in[(size_t)(3*r + no)*S + kn];
                const double d1 = cb[r] - ca[r], d2 = co - ca[r];
                Gn[r][0] = (d1*e2y - d2*e1y)*rdet;
                Gn[r][1] = (d2*e1x - d1*e2x)*rdet;
            }
#pragma unroll
            for (int r = 0; r < NC; r++)
#pragma unroll
                for (int j = 0; j < 2; j++) S0n[r][j] = Gn[r][j] + ((NC == 2 && gd) ? Gn[j % NC][r] : 0.0);
            const double sigma = p.sipg*L/fmin(A, An);             // max over the two sides of sipg*cp*|F|/A
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
                        const double H0 = p.nonlin ? hq + e_ext : hq;
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
        for (int i = 0; i < 3; i++) p.out[(size_t)(3*r + i)*S + k] += s*(4.0*b[r][i] - sb);
    }
}
