/*
 * ORACLE - TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Plain-C (OpenMP) restatement of the DG-P1 2D shallow-water residual + block mass inverse + SSPRK33
 * Shu-Osher update that Thetis builds in UFL and hands to Firedrake.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load the shared object built from this file.
 *
 * PARITY UNPINNED at round-off level (see oracle/swe2d_oracle.py header and DESIGN.md): the reference's
 * arithmetic lives in Firedrake/PETSc, which cannot be installed or run here.  This file is pinned against
 * oracle/swe2d_oracle.py (literal UFL restatement, quadrature everywhere) and through it against the
 * reference's own known-answer tests.
 *
 * Formulation: element-centric (every cell evaluates the numerical flux of its three facets itself, so both
 * sides of an interior facet compute it once each), closed-form P1 cell integrals
 *   int l_i = A/3,  int l_i l_j = A(1+d_ij)/12        
 * and 2-point Gauss-Legendre on facets.  Reference forms (file:line under /root/reference/thetis):
 *   ExternalPressureGradientTerm shallowwater_eq.py:360-381     HUDivTerm shallowwater_eq.py:421-442
 *   HorizontalAdvectionTerm      shallowwater_eq.py:470-510     get_bnd_functions shallowwater_eq.py:232-272
 *   CoriolisTerm :623-634  LinearDragTerm :734-740  QuadraticDragTerm :679-701  AtmosphericPressureTerm :658-663
 *   MomentumSourceTerm :805-811  ContinuitySourceTerm :825-831
 *   mass inverse equation.py:105 (M_K = A/12 [[2,1,1],[1,2,1],[1,1,2]])   SSPRK33 rungekutta.py:326-347,908-946
 *
 * Data layout: cell-major AoS like the numpy oracle: uv[N][k][2], eta[N][k], xy[N][k][2], h[N][k], k = npc.
 * Quadrilaterals (k = 4, parallelograms): bilinear basis on the unit square, nodes counter-clockwise from (0,0);
 * cell integrals by the 2 x 2 Gauss-Legendre rule, M^-1 = (1/A) m^-1 (x) m^-1 with m^-1 = [[4,-2],[-2,4]].
 */
#define _POSIX_C_SOURCE 200112L
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BC_CLOSED 0
#define BC_ELEV   1
#define BC_UV     2
#define BC_UN     4
#define BC_FLUX   8

typedef struct {
    int n_cells;
    const int *nbr;            /* [N][3]  >=0 neighbour cell, <0: -(marker)            */
    const signed char *nbf;    /* [N][3]  local facet id inside the neighbour          */
    const double *xy;          /* [N][3][2]                                            */
    const double *h;           /* [N][3]  bathymetry at the cell nodes (CG-P1)         */
    double g;
    int nonlinear;             /* use_nonlinear_equations                              */
    int use_lf;                /* use_lax_friedrichs_velocity                          */
    double sigma_lf;
    /* optional cell-local terms: NULL / 0 = off */
    const double *coriolis;    /* [N][3]                                               */
    double linear_drag;        /* constant, <0 = off                                   */
    double quad_drag;          /* constant C_D, <0 = off                               */
    double manning;            /* constant mu, <0 = off                                */
    double norm_smoother;
    const double *patm;        /* [N][3]                                               */
    const double *mom_src;     /* [N][3][2]                                            */
    const double *vol_src;     /* [N][3]                                               */
    /* open boundaries, indexed by marker (0..n_markers-1 -> marker = index) */
    int n_markers;
    const int *bc_kind;        /* bitmask of BC_*                                      */
    const double *bc_elev;     /* per marker                                           */
    const double *bc_uv;       /* per marker [2]                                       */
    const double *bc_un;       /* per marker                                           */
    const double *bc_flux;     /* per marker                                           */
    const double *bc_len;      /* per marker total boundary length                     */
    int npc;                   /* nodes per cell: 3 (DG-P1 triangles) or 4 (DQ-1 parallelograms) */
    /* explicit wetting-drying (this build's own nodal formulation, see oracle/swe2d_oracle.py header) */
    int wd;                    /* use_wetting_and_drying                                */
    const double *alpha;       /* [N][k] wetting_and_drying_alpha at the cell nodes     */
    const double *wind;        /* [N][k][2] wind stress (WindStressTerm shallowwater_eq.py:643-649) or NULL */
    const double *bc_drag;     /* per marker boundary drag C_D (BoundaryDragTerm :712-725), <0 = none, or NULL */
} swe2d_ref_t;

static const double GL_XI[2] = {0.21132486540518713, 0.78867513459481287};

/* Dunavant degree-4 6-point rule for the non-polynomial (Manning) cell integrand */
static const double TRI_B[6][3] = {
    {0.108103018168070, 0.445948490915965, 0.445948490915965},
    {0.445948490915965, 0.108103018168070, 0.445948490915965},
    {0.445948490915965, 0.445948490915965, 0.108103018168070},
    {0.816847572980459, 0.091576213509771, 0.091576213509771},
    {0.091576213509771, 0.816847572980459, 0.091576213509771},
    {0.091576213509771, 0.091576213509771, 0.816847572980459}};
static const double TRI_W[6] = {0.223381589678011, 0.223381589678011, 0.223381589678011,
                                0.109951743655322, 0.109951743655322, 0.109951743655322};

static inline double total_depth(const swe2d_ref_t *m, double h, double eta)
{
    return m->nonlinear ? h + eta : h;
}

/* displaced depth D = (H + sqrt(H^2 + a^2))/2, H = h + eta   (utility.py:975-993) */
static inline double wd_depth(double h, double eta, double a)
{
    const double H = h + eta;
    return 0.5*(H + sqrt(H*H + a*a));
}

/* pointwise depth for external (boundary) states */
static inline double depth_pt(const swe2d_ref_t *m, double h, double eta, double a)
{
    return m->wd ? wd_depth(h, eta, a) : total_depth(m, h, eta);
}

/* int a*b over the cell, a,b P1 */
static inline double int2(double A, const double a[3], const double b[3])
{
    double sa = a[0] + a[1] + a[2], sb = b[0] + b[1] + b[2];
    return A/12.0*(sa*sb + a[0]*b[0] + a[1]*b[1] + a[2]*b[2]);
}

/* cell integrals of a parallelogram (DQ-1), 2 x 2 Gauss-Legendre; returns the area */
static double quad_cell_terms(const swe2d_ref_t *m, int k, const double *p, const double u[4], const double v[4],
                              const double e[4], const double H[4], double bu[4], double bv[4], double be[4])
{
    const double g = m->g;
    const double ax = p[2] - p[0], ay = p[3] - p[1];          /* a = p1 - p0 */
    const double bx = p[6] - p[0], by = p[7] - p[1];          /* b = p3 - p0 */
    const double A = ax*by - ay*bx;
    const double xix = by/A, xiy = -bx/A, zex = -ay/A, zey = ax/A;     /* grad xi, grad zeta */
    const double *cor = m->coriolis ? m->coriolis + 4*(size_t)k : 0;
    const double *pa = m->patm ? m->patm + 4*(size_t)k : 0;
    const double *ms = m->mom_src ? m->mom_src + 8*(size_t)k : 0;
    const double *vs = m->vol_src ? m->vol_src + 4*(size_t)k : 0;
    for (int qi = 0; qi < 2; qi++) for (int qz = 0; qz < 2; qz++) {
        const double xi = GL_XI[qi], ze = GL_XI[qz], w = 0.25*A;
        const double phi[4] = {(1 - xi)*(1 - ze), xi*(1 - ze), xi*ze, (1 - xi)*ze};
        const double dxi[4] = {-(1 - ze), (1 - ze), ze, -ze};
        const double dze[4] = {-(1 - xi), -xi, xi, (1 - xi)};
        double gx[4], gy[4], uq = 0, vq = 0, eq = 0, Hq = 0, divu = 0;
        for (int i = 0; i < 4; i++) {
            gx[i] = dxi[i]*xix + dze[i]*zex;
            gy[i] = dxi[i]*xiy + dze[i]*zey;
            uq += phi[i]*u[i]; vq += phi[i]*v[i]; eq += phi[i]*e[i]; Hq += phi[i]*H[i];
            divu += gx[i]*u[i] + gy[i]*v[i];
        }
        double corq = 0, gpx = 0, gpy = 0, sx = 0, sy = 0, sv = 0;
        if (m->wind) {
            const double *ws = m->wind + 8*(size_t)k;
            for (int i = 0; i < 4; i++) { sx += phi[i]*ws[2*i]/(Hq*1000.0); sy += phi[i]*ws[2*i + 1]/(Hq*1000.0); }
        }
        for (int i = 0; i < 4; i++) {
            if (cor) corq += phi[i]*cor[i];
            if (pa) { gpx += gx[i]*pa[i]; gpy += gy[i]*pa[i]; }
            if (ms) { sx += phi[i]*ms[2*i]; sy += phi[i]*ms[2*i + 1]; }
            if (vs) sv += phi[i]*vs[i];
        }
        double drag = 0;
        if (m->quad_drag >= 0 || m->manning >= 0) {
            const double cd = (m->manning >= 0) ? g*m->manning*m->manning/cbrt(Hq) : m->quad_drag;
            drag = cd*sqrt(uq*uq + vq*vq + m->norm_smoother*m->norm_smoother)/Hq;
        }
        if (m->linear_drag >= 0) drag += m->linear_drag;
        for (int i = 0; i < 4; i++) {
            double fu = g*eq*gx[i], fv = g*eq*gy[i];                              /* :361 */
            if (m->nonlinear) {                                                   /* :478 */
                const double adv = phi[i]*divu + uq*gx[i] + vq*gy[i];
                fu += adv*uq; fv += adv*vq;
            }
            fu += corq*vq*phi[i]; fv -= corq*uq*phi[i];                           /* :632-633 */
            fu -= drag*phi[i]*uq; fv -= drag*phi[i]*vq;                           /* :700, :738 */
            fu -= gpx/1000.0*phi[i]; fv -= gpy/1000.0*phi[i];                     /* :662 */
            fu += sx*phi[i]; fv += sy*phi[i];                                     /* :810 */
            bu[i] += w*fu; bv[i] += w*fv;
            be[i] += w*(Hq*(gx[i]*uq + gy[i]*vq) + sv*phi[i]);                    /* :422, :830 */
        }
    }
    return A;
}

/* k = M^-1 (dt R(U)) for one cell */
/* k_uv, k_eta: the output of THIS cell (2*npc and npc doubles) */
static void cell_tendency(const swe2d_ref_t *m, int k, const double *uv, const double *eta, double dt,
                          double *k_uv, double *k_eta)
{
    const double g = m->g;
    const int npc = m->npc;
    const double *p = m->xy + 2*(size_t)npc*k;
    const double *hk = m->h + (size_t)npc*k;
    double u[4], v[4], e[4], H[4], al[4] = {0, 0, 0, 0};
    for (int i = 0; i < npc; i++) {
        u[i] = uv[2*(size_t)npc*k + 2*i];
        v[i] = uv[2*(size_t)npc*k + 2*i + 1];
        e[i] = eta[(size_t)npc*k + i];
        if (m->wd) al[i] = m->alpha[(size_t)npc*k + i];
        H[i] = m->wd ? wd_depth(hk[i], e[i], al[i]) : total_depth(m, hk[i], e[i]);
    }
    double bu[4] = {0, 0, 0, 0}, bv[4] = {0, 0, 0, 0}, be[4] = {0, 0, 0, 0};
    double A;
    if (npc == 4) {
        A = quad_cell_terms(m, k, p, u, v, e, H, bu, bv, be);
    } else {
    A = 0.5*((p[2] - p[0])*(p[5] - p[1]) - (p[4] - p[0])*(p[3] - p[1]));
    double gx[3], gy[3];       /* grad phi_i */
    for (int i = 0; i < 3; i++) {
        int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
        gx[i] = (p[2*i1 + 1] - p[2*i2 + 1])/(2*A);
        gy[i] = (p[2*i2] - p[2*i1])/(2*A);
    }

    /* ---- cell integrals */
    const double esum = e[0] + e[1] + e[2];
    const double IHu = int2(A, H, u), IHv = int2(A, H, v);
    double Iuu = 0, Iuv = 0, Ivv = 0, divu = 0;
    if (m->nonlinear) {
        Iuu = int2(A, u, u); Iuv = int2(A, u, v); Ivv = int2(A, v, v);
        for (int i = 0; i < 3; i++) divu += gx[i]*u[i] + gy[i]*v[i];
    }
    const double usum = u[0] + u[1] + u[2], vsum = v[0] + v[1] + v[2];
    for (int i = 0; i < 3; i++) {
        bu[i] += g*gx[i]*A/3.0*esum;                               /* +g eta div(psi)          :361 */
        bv[i] += g*gy[i]*A/3.0*esum;
        be[i] += gx[i]*IHu + gy[i]*IHv;                            /* +grad(phi).(H u)         :422 */
        if (m->nonlinear) {                                        /* +(psi div u + u.grad psi) u :478 */
            bu[i] += divu*A/12.0*(usum + u[i]) + gx[i]*Iuu + gy[i]*Iuv;
            bv[i] += divu*A/12.0*(vsum + v[i]) + gx[i]*Iuv + gy[i]*Ivv;
        }
    }
    if (m->coriolis) {                                             /* :632-633, f P1: cubic integrand */
        const double *f = m->coriolis + 3*(size_t)k;
        for (int i = 0; i < 3; i++) {
            double su = 0, sv = 0;
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
                /* int l_i l_a l_b = A*{6,2,1}/60 */
                int same = (i == a) + (i == b) + (a == b);
                double w = (same == 3) ? 6.0 : (same == 1 ? 2.0 : 1.0);
                su += w*f[a]*u[b]; sv += w*f[a]*v[b];
            }
            bu[i] += A/60.0*sv;       /* -f*(-v psi_x) */
            bv[i] -= A/60.0*su;       /* -f*( u psi_y) */
        }
    }
    if (m->linear_drag >= 0) {                                     /* :738 */
        for (int i = 0; i < 3; i++) {
            bu[i] -= m->linear_drag*A/12.0*(usum + u[i]);
            bv[i] -= m->linear_drag*A/12.0*(vsum + v[i]);
        }
    }
    if (m->quad_drag >= 0 || m->manning >= 0) {                    /* :685-700, quadrature */
        for (int q = 0; q < 6; q++) {
            const double *b = TRI_B[q];
            double uq = b[0]*u[0] + b[1]*u[1] + b[2]*u[2];
            double vq = b[0]*v[0] + b[1]*v[1] + b[2]*v[2];
            double Hq = b[0]*H[0] + b[1]*H[1] + b[2]*H[2];
            double cd = (m->manning >= 0) ? g*m->manning*m->manning/cbrt(Hq) : m->quad_drag;
            double s = TRI_W[q]*A*cd*sqrt(uq*uq + vq*vq + m->norm_smoother*m->norm_smoother)/Hq;
            for (int i = 0; i < 3; i++) { bu[i] -= s*b[i]*uq; bv[i] -= s*b[i]*vq; }
        }
    }
    if (m->wind) {                                                 /* :648, +tau.psi/(H rho0), 6-point rule */
        const double *ws = m->wind + 6*(size_t)k;
        for (int q = 0; q < 6; q++) {
            const double *b = TRI_B[q];
            double Hq = b[0]*H[0] + b[1]*H[1] + b[2]*H[2];
            double wx = b[0]*ws[0] + b[1]*ws[2] + b[2]*ws[4], wy = b[0]*ws[1] + b[1]*ws[3] + b[2]*ws[5];
            double s = TRI_W[q]*A/(Hq*1000.0);
            for (int i = 0; i < 3; i++) { bu[i] += s*b[i]*wx; bv[i] += s*b[i]*wy; }
        }
    }
    if (m->patm) {                                                 /* :662 */
        const double *pa = m->patm + 3*(size_t)k;
        double px = gx[0]*pa[0] + gx[1]*pa[1] + gx[2]*pa[2];
        double py = gy[0]*pa[0] + gy[1]*pa[1] + gy[2]*pa[2];
        for (int i = 0; i < 3; i++) { bu[i] -= px/1000.0*A/3.0; bv[i] -= py/1000.0*A/3.0; }
    }
    if (m->mom_src) {                                              /* :810 */
        const double *s = m->mom_src + 6*(size_t)k;
        double sx = s[0] + s[2] + s[4], sy = s[1] + s[3] + s[5];
        for (int i = 0; i < 3; i++) { bu[i] += A/12.0*(sx + s[2*i]); bv[i] += A/12.0*(sy + s[2*i + 1]); }
    }
    if (m->vol_src) {                                              /* :830 */
        const double *s = m->vol_src + 3*(size_t)k;
        double ss = s[0] + s[1] + s[2];
        for (int i = 0; i < 3; i++) be[i] += A/12.0*(ss + s[i]);
    }

    }   /* npc == 3 */

    /* ---- facets */
    for (int f = 0; f < npc; f++) {
        const int a = f, b = (f + 1) % npc;
        const double dx = p[2*b] - p[2*a], dy = p[2*b + 1] - p[2*a + 1];
        const double len = sqrt(dx*dx + dy*dy);
        const double nx = dy/len, ny = -dx/len;
        const int nb = m->nbr[(size_t)npc*k + f];
        double ua_n = 0, ub_n = 0, va_n = 0, vb_n = 0, ea_n = 0, eb_n = 0;
        int kind = BC_CLOSED, marker = 0;
        if (nb >= 0) {
            const int f2 = m->nbf[(size_t)npc*k + f];
            const int na = (f2 + 1) % npc, nbb = f2;      /* neighbour traverses the facet backwards */
            const size_t o = 2*(size_t)npc*nb, oe = (size_t)npc*nb;
            ua_n = uv[o + 2*na];   va_n = uv[o + 2*na + 1];   ea_n = eta[oe + na];
            ub_n = uv[o + 2*nbb];  vb_n = uv[o + 2*nbb + 1];  eb_n = eta[oe + nbb];
        } else {
            marker = -nb;
            if (marker < m->n_markers && m->bc_kind) kind = m->bc_kind[marker];
        }
        for (int q = 0; q < 2; q++) {
            const double xb = GL_XI[q], xa = 1.0 - xb;
            const double w = 0.5*len;
            const double uq = xa*u[a] + xb*u[b], vq = xa*v[a] + xb*v[b], eq = xa*e[a] + xb*e[b];
            const double hq = xa*hk[a] + xb*hk[b];
            const double alq = xa*al[a] + xb*al[b];
            const double Hq = m->wd ? xa*H[a] + xb*H[b] : total_depth(m, hq, eq);
            double fu = 0, fv = 0, fe = 0;     /* the form f; residual is -f */
            if (nb >= 0) {
                const double un_ = xa*ua_n + xb*ub_n, vn_ = xa*va_n + xb*vb_n, en_ = xa*ea_n + xb*eb_n;
                const double Hn = m->wd ? xa*wd_depth(hk[a], ea_n, al[a]) + xb*wd_depth(hk[b], eb_n, al[b])
                                        : total_depth(m, hq, en_);
                const double Hav = 0.5*(Hq + Hn);
                const double uav = 0.5*(uq + un_), vav = 0.5*(vq + vn_);
                const double jump_un = (uq - un_)*nx + (vq - vn_)*ny;
                const double head_star = 0.5*(eq + en_) + sqrt(Hav/g)*jump_un;           /* :363 */
                fu += g*head_star*nx; fv += g*head_star*ny;                               /* :366 */
                const double c2 = sqrt(g/Hav)*(eq - en_);
                fe += Hav*((uav + c2*nx)*nx + (vav + c2*ny)*ny);                          /* :424-427 */
                if (m->nonlinear) {
                    const double un_own = uq*nx + vq*ny;
                    fu += uav*un_own; fv += vav*un_own;                                   /* :483 */
                    if (m->use_lf) {
                        const double gamma = 0.5*fabs(uav*nx + vav*ny)*m->sigma_lf;       /* :487 */
                        fu += gamma*(uq - un_); fv += gamma*(vq - vn_);                   /* :488 */
                    }
                }
            } else if (kind == BC_CLOSED) {
                const double un_own = uq*nx + vq*ny;
                const double head_rie = eq + sqrt(Hq/g)*un_own;                           /* :379-380 */
                fu += g*head_rie*nx; fv += g*head_rie*ny;
                if (m->nonlinear && m->use_lf) {
                    const double gamma = 0.5*fabs(un_own)*m->sigma_lf;                    /* :496 */
                    fu += gamma*2.0*un_own*nx; fv += gamma*2.0*un_own*ny;                 /* :497 */
                }
            } else {
                /* external state, get_bnd_functions :243-267 */
                double e_ext = eq, u_ext = uq, v_ext = vq;
                if (kind & BC_ELEV) e_ext = m->bc_elev[marker];
                if (kind & BC_UV) { u_ext = m->bc_uv[2*marker]; v_ext = m->bc_uv[2*marker + 1]; }
                else if (kind & BC_UN) { u_ext = m->bc_un[marker]*nx; v_ext = m->bc_un[marker]*ny; }
                else if (kind & BC_FLUX) {
                    const double H_ext0 = depth_pt(m, hq, e_ext, alq);
                    const double s = m->bc_flux[marker]/(H_ext0*m->bc_len[marker]);
                    u_ext = s*nx; v_ext = s*ny;
                }
                const double H_ext = depth_pt(m, hq, e_ext, alq);
                const double un_jump = (uq - u_ext)*nx + (vq - v_ext)*ny;
                const double eta_rie = 0.5*(eq + e_ext) + sqrt(Hq/g)*un_jump;             /* :374 */
                fu += g*eta_rie*nx; fv += g*eta_rie*ny;                                   /* :375 */
                const double h_av = 0.5*(Hq + H_ext);
                const double eta_jump = eq - e_ext;
                const double un_avg = 0.5*((uq + u_ext)*nx + (vq + v_ext)*ny);
                const double un_rie = un_avg + sqrt(g/h_av)*eta_jump;                     /* :438 */
                const double eta_rie2 = 0.5*(eq + e_ext) + sqrt(h_av/g)*un_jump;          /* :440 */
                fe += depth_pt(m, hq, eta_rie2, alq)*un_rie;                              /* :441-442 */
                if (m->nonlinear) {
                    const double un_rie3 = un_avg + sqrt(g/Hq)*eta_jump;                  /* :507 */
                    fu += un_rie3*0.5*(u_ext + uq); fv += un_rie3*0.5*(v_ext + vq);       /* :508-509 */
                }
            }
            if (nb < 0 && m->bc_drag && marker < m->n_markers && m->bc_drag[marker] >= 0) {      /* :717-724 */
                const double un_own = uq*nx + vq*ny;
                const double utx = uq - un_own*nx, uty = vq - un_own*ny;
                const double mag = sqrt(utx*utx + uty*uty);
                fu += m->bc_drag[marker]*mag*utx; fv += m->bc_drag[marker]*mag*uty;
            }
            bu[a] -= w*xa*fu; bu[b] -= w*xb*fu;
            bv[a] -= w*xa*fv; bv[b] -= w*xb*fv;
            be[a] -= w*xa*fe; be[b] -= w*xb*fe;
        }
    }
    if (npc == 4) {
        /* (M^-1 b)_i = (16 b_i - 8 b_{i+1} - 8 b_{i-1} + 4 b_{i+2})/A  (tensor of the 1D inverse [[4,-2],[-2,4]]) */
        const double s4 = dt/A;
        for (int i = 0; i < 4; i++) {
            const int n1 = (i + 1) % 4, n2 = (i + 2) % 4, n3 = (i + 3) % 4;
            k_uv[2*i] = s4*(16.0*bu[i] - 8.0*bu[n1] - 8.0*bu[n3] + 4.0*bu[n2]);
            k_uv[2*i + 1] = s4*(16.0*bv[i] - 8.0*bv[n1] - 8.0*bv[n3] + 4.0*bv[n2]);
            k_eta[i] = s4*(16.0*be[i] - 8.0*be[n1] - 8.0*be[n3] + 4.0*be[n2]);
        }
        return;
    }
    /* ---- mass inverse: (M^-1 b)_i = 3/A (4 b_i - sum b) */
    const double s = 3.0*dt/A;
    const double su = bu[0] + bu[1] + bu[2], sv = bv[0] + bv[1] + bv[2], se = be[0] + be[1] + be[2];
    for (int i = 0; i < 3; i++) {
        k_uv[2*i] = s*(4.0*bu[i] - su);
        k_uv[2*i + 1] = s*(4.0*bv[i] - sv);
        k_eta[i] = s*(4.0*be[i] - se);
    }
}

/* k = M^-1 (dt R(U)) for the whole mesh */
void swe2d_ref_tendency(const swe2d_ref_t *m, const double *uv, const double *eta, double dt,
                        double *k_uv, double *k_eta)
{
#pragma omp parallel for schedule(static)
    for (int k = 0; k < m->n_cells; k++)
        cell_tendency(m, k, uv, eta, dt, k_uv + 2*(size_t)m->npc*(size_t)k, k_eta + (size_t)m->npc*(size_t)k);
}

/* n_steps SSPRK33 steps in place; work must hold 4 states = 4*9*N doubles.  Shu-Osher form, expression
 * order of rungekutta.py:911-913: tendency*beta + sum_j stage_sol[j]*alpha.  Constant-in-time forcing only. */
/* explicit wetting-drying: the continuity equation advances zeta = D - h; eta is recovered from H = D - a^2/(4 D) */
/* One cell of a wetting-drying stage: zeta_new = bk + a0 zeta(e0) + a1 zeta(e1) at every node, D = zeta + h, the positivity
 * limiter, eta from D.  The nodal values of a DG-P1 depth can undershoot although the cell mean stays positive (the mass
 * inverse amplifies nodal changes by up to 9), and eta(D) = D - a^2/(4D) - h has a pole at D = 0: one negative nodal D and
 * the step is lost.  Limiter (Xing-Zhang type, cell-local, keeps the cell mean = conservative): the nodal deviations from the
 * mean are scaled by theta = min(1, (mean - floor)/(mean - min)) so that every node has D >= floor = WD_FLOOR * alpha (the
 * depth at which H = h + eta = -(1/(4 WD_FLOOR) - WD_FLOOR) alpha, i.e. the water table |H| of a dry node is bounded by a
 * multiple of alpha and with it the speed sqrt(g |H|) of the waves the modified continuity equation carries there).  A cell
 * whose MEAN fell below the floor is flattened to its mean (conservative) and only below a tenth of the floor raised to that.
 * Restated in numpy as SWEOracle.wd_finish_stage (oracle/swe2d_oracle.py). */
#define WD_FLOOR 0.1             /* oracle/swe2d_oracle.py WD_FLOOR */
static inline void wd_stage_cell(const swe2d_ref_t *m, long c, const double *bk, const double *e0, double a0, const double *e1,
                                 double a1, double *out)
{
    const int k = m->npc;
    double D[4], mean = 0.0, dmin = 1e300, fl = 0.0;
    for (int i = 0; i < k; i++) {
        const double h = m->h[c*k + i], a = m->alpha[c*k + i];
        D[i] = bk[i] + (wd_depth(h, e0[i], a) - h)*a0 + (wd_depth(h, e1[i], a) - h)*a1 + h;
        mean += D[i];
        if (D[i] < dmin) dmin = D[i];
        if (WD_FLOOR*a > fl) fl = WD_FLOOR*a;
    }
    mean /= k;
    if (dmin < fl) {
        if (mean <= fl) {
            /* the whole cell is below the floor: flatten it (no volume added) down to a hard floor of a tenth of it */
            const double flat = mean > 0.1*fl ? mean : 0.1*fl;
            for (int i = 0; i < k; i++) D[i] = flat;
        }
        else {
            const double theta = (mean - fl)/(mean - dmin);
            for (int i = 0; i < k; i++) D[i] = mean + theta*(D[i] - mean);
        }
    }
    for (int i = 0; i < k; i++) {
        const double h = m->h[c*k + i], a = m->alpha[c*k + i];
        out[i] = D[i] - a*a/(4.0*D[i]) - h;
    }
}

/* dry-ground velocity relaxation of the explicit wetting-drying scheme: u <- u exp(-dt_stage/tau * psi^2),
 * tau = WD_TAU_FACTOR * sqrt(alpha/g), psi = clamp(-H/alpha - 1, 0, 1): nothing where the water table is less than alpha below
 * the bed, full strength from 2 alpha on (oracle/swe2d_oracle.py, module docstring). */
#define WD_TAU_FACTOR 10.0       /* oracle/swe2d_oracle.py WD_TAU */
static inline void wd_damp_cell(const swe2d_ref_t *m, long c, const double *eta_new, double dt_stage, double *uv_cell)
{
    const int k = m->npc;
    for (int i = 0; i < k; i++) {
        const double h = m->h[c*k + i], a = m->alpha[c*k + i];
        double psi = -(h + eta_new[i])/a - 1.0;
        psi = psi < 0.0 ? 0.0 : (psi > 1.0 ? 1.0 : psi);
        const double fac = exp(-dt_stage/(WD_TAU_FACTOR*sqrt(a/m->g))*psi*psi);
        uv_cell[2*i] *= fac;
        uv_cell[2*i + 1] *= fac;
    }
}

void swe2d_ref_advance(const swe2d_ref_t *m, double *uv, double *eta, double dt, int n_steps, double *work)
{
    const size_t nu = 2*(size_t)m->npc*(size_t)m->n_cells, ne = (size_t)m->npc*(size_t)m->n_cells;
    double *u0 = work, *e0 = u0 + nu;
    double *ku = e0 + ne, *ke = ku + nu;
    static const double A30 = 0.33333333333333337, A32 = 0.6666666666666666, B32 = 0.6666666666666666;
    for (int it = 0; it < n_steps; it++) {
        memcpy(u0, uv, nu*sizeof(double));
        memcpy(e0, eta, ne*sizeof(double));
        /* stage 0: U1 = k*1 + U0*1 */
        swe2d_ref_tendency(m, uv, eta, dt, ku, ke);
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)nu; i++) uv[i] = ku[i]*1.0 + u0[i]*1.0;
        if (m->wd) {
#pragma omp parallel for schedule(static)
            for (long c = 0; c < (long)m->n_cells; c++) {
                double b[4];
                for (int i = 0; i < m->npc; i++) b[i] = ke[c*m->npc + i]*1.0;
                wd_stage_cell(m, c, b, e0 + c*m->npc, 1.0, e0 + c*m->npc, 0.0, eta + c*m->npc);
                wd_damp_cell(m, c, eta + c*m->npc, 1.0*dt, uv + 2*c*m->npc);
            }
        } else {
#pragma omp parallel for schedule(static)
            for (long i = 0; i < (long)ne; i++) eta[i] = ke[i]*1.0 + e0[i]*1.0;
        }
        /* stage 1: U2 = k*0.25 + U0*0.75 + U1*0.25   (U1 is the current solution) */
        swe2d_ref_tendency(m, uv, eta, dt, ku, ke);
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)nu; i++) uv[i] = ku[i]*0.25 + u0[i]*0.75 + uv[i]*0.25;
        if (m->wd) {
#pragma omp parallel for schedule(static)
            for (long c = 0; c < (long)m->n_cells; c++) {
                double b[4];
                for (int i = 0; i < m->npc; i++) b[i] = ke[c*m->npc + i]*0.25;
                wd_stage_cell(m, c, b, e0 + c*m->npc, 0.75, eta + c*m->npc, 0.25, eta + c*m->npc);
                wd_damp_cell(m, c, eta + c*m->npc, 0.25*dt, uv + 2*c*m->npc);
            }
        } else {
#pragma omp parallel for schedule(static)
            for (long i = 0; i < (long)ne; i++) eta[i] = ke[i]*0.25 + e0[i]*0.75 + eta[i]*0.25;
        }
        /* stage 2: U3 = k*B32 + U0*A30 + U1*0 + U2*A32 */
        swe2d_ref_tendency(m, uv, eta, dt, ku, ke);
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)nu; i++) uv[i] = ku[i]*B32 + u0[i]*A30 + uv[i]*A32;
        if (m->wd) {
#pragma omp parallel for schedule(static)
            for (long c = 0; c < (long)m->n_cells; c++) {
                double b[4];
                for (int i = 0; i < m->npc; i++) b[i] = ke[c*m->npc + i]*B32;
                wd_stage_cell(m, c, b, e0 + c*m->npc, A30, eta + c*m->npc, A32, eta + c*m->npc);
                wd_damp_cell(m, c, eta + c*m->npc, B32*dt, uv + 2*c*m->npc);
            }
        } else {
#pragma omp parallel for schedule(static)
            for (long i = 0; i < (long)ne; i++) eta[i] = ke[i]*B32 + e0[i]*A30 + eta[i]*A32;
        }
    }
}

/* ---- the timed CPU baseline of bench.py ------------------------------------------------------------------------------
 * The same n_steps SSPRK33 steps as swe2d_ref_advance, bit for bit (tests/test_oracle_known_answers.py), organised the way
 * a production CPU code would be: ONE parallel region for the whole run (a persistent thread team, three barriers per step),
 * every thread owns a contiguous block of cells, residual + mass inverse + Shu-Osher combine fused per cell on three rotating
 * state buffers (no tendency arrays, no copies of U0), and every per-cell array - mesh data and state - is copied into
 * buffers that the owning thread touches first, so that with OMP_PROC_BIND=close the pages of a block live on the NUMA node
 * of the cores that sweep it.  Closed walls / constant forcing, no wetting-drying (the bench workload); returns the seconds
 * spent in the step loop (copy-in / copy-out excluded), or -1 if the configuration is outside that scope. */
static void *ref_alloc(size_t bytes)
{
    void *p = NULL;
    return posix_memalign(&p, 4096, bytes ? bytes : 4096) == 0 ? p : NULL;
}

double swe2d_ref_advance_blocked(const swe2d_ref_t *m, double *uv, double *eta, double dt, int n_steps)
{
    if (m->wd) return -1.0;
    const size_t n = (size_t)m->n_cells, k = (size_t)m->npc;
    static const double A30 = 0.33333333333333337, A32 = 0.6666666666666666, B32 = 0.6666666666666666;
    swe2d_ref_t loc = *m;
    int *nbr = ref_alloc(n*k*sizeof(int));
    signed char *nbf = ref_alloc(n*k);
    double *xy = ref_alloc(n*k*2*sizeof(double)), *h = ref_alloc(n*k*sizeof(double));
    double *U[3], *E[3];
    for (int b = 0; b < 3; b++) { U[b] = ref_alloc(n*k*2*sizeof(double)); E[b] = ref_alloc(n*k*sizeof(double)); }
    double seconds = -1.0;
    if (nbr && nbf && xy && h && U[0] && U[1] && U[2] && E[0] && E[1] && E[2]) {
        loc.nbr = nbr; loc.nbf = nbf; loc.xy = xy; loc.h = h;
#ifdef _OPENMP
        double t0 = 0.0;
#pragma omp parallel
#else
        clock_t c0 = 0;
#endif
        {
            /* first touch = owner: the static schedule below is the one every later loop uses */
#pragma omp for schedule(static)
            for (long c = 0; c < (long)n; c++) {
                memcpy(nbr + c*k, m->nbr + c*k, k*sizeof(int));
                memcpy(nbf + c*k, m->nbf + c*k, k);
                memcpy(xy + c*k*2, m->xy + c*k*2, k*2*sizeof(double));
                memcpy(h + c*k, m->h + c*k, k*sizeof(double));
                memcpy(U[0] + c*k*2, uv + c*k*2, k*2*sizeof(double));
                memcpy(E[0] + c*k, eta + c*k, k*sizeof(double));
                memset(U[1] + c*k*2, 0, k*2*sizeof(double)); memset(U[2] + c*k*2, 0, k*2*sizeof(double));
                memset(E[1] + c*k, 0, k*sizeof(double)); memset(E[2] + c*k, 0, k*sizeof(double));
            }
#ifdef _OPENMP
#pragma omp master
            t0 = omp_get_wtime();
#else
            c0 = clock();
#endif
            for (int it = 0; it < n_steps; it++) {
                /* stage i reads buffer i (and U0 = buffer 0), writes buffer (i + 1) % 3; the implicit barrier of each
                 * `omp for` separates the stages; stage 2 overwrites U0 in place (only its own cell reads it) */
                for (int st = 0; st < 3; st++) {
                    const double *ui = U[st], *ei = E[st];
                    double *uo = U[(st + 1) % 3], *eo = E[(st + 1) % 3];
#pragma omp for schedule(static)
                    for (long c = 0; c < (long)n; c++) {
                        double ku[8], ke[4];
                        cell_tendency(&loc, (int)c, ui, ei, dt, ku, ke);
                        const double *u0 = U[0] + c*k*2, *e0 = E[0] + c*k, *u1 = ui + c*k*2, *e1 = ei + c*k;
                        double *ou = uo + c*k*2, *oe = eo + c*k;
                        if (st == 0) {
                            for (size_t i = 0; i < 2*k; i++) ou[i] = ku[i]*1.0 + u0[i]*1.0;
                            for (size_t i = 0; i < k; i++) oe[i] = ke[i]*1.0 + e0[i]*1.0;
                        } else if (st == 1) {
                            for (size_t i = 0; i < 2*k; i++) ou[i] = ku[i]*0.25 + u0[i]*0.75 + u1[i]*0.25;
                            for (size_t i = 0; i < k; i++) oe[i] = ke[i]*0.25 + e0[i]*0.75 + e1[i]*0.25;
                        } else {
                            for (size_t i = 0; i < 2*k; i++) ou[i] = ku[i]*B32 + u0[i]*A30 + u1[i]*A32;
                            for (size_t i = 0; i < k; i++) oe[i] = ke[i]*B32 + e0[i]*A30 + e1[i]*A32;
                        }
                    }
                }
            }
#ifdef _OPENMP
#pragma omp master
            seconds = omp_get_wtime() - t0;
#pragma omp barrier
#else
            seconds = (double)(clock() - c0)/CLOCKS_PER_SEC;
#endif
#pragma omp for schedule(static)
            for (long c = 0; c < (long)n; c++) {
                memcpy(uv + c*k*2, U[0] + c*k*2, k*2*sizeof(double));
                memcpy(eta + c*k, E[0] + c*k, k*sizeof(double));
            }
        }
    }
    free(nbr); free(nbf); free(xy); free(h);
    for (int b = 0; b < 3; b++) { free(U[b]); free(E[b]); }
    return seconds;
}

int swe2d_ref_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void swe2d_ref_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ===================================================================================================================
 * 2D tracer (non-conservative form) and the vertex-based P1DG limiter.
 *   tracer_eq_2d.HorizontalAdvectionTerm  thetis/tracer_eq_2d.py:147-193   SourceTerm :281-298
 *   VertexBasedP1DGLimiter               thetis/limiter.py:48-198 (+ firedrake.VertexBasedLimiter [FD-assumed])
 * Closed walls / boundaries without a tracer BC use  f += c (u.n) phi  (:189-191); boundaries with a 'value' use the
 * upwind blend between c and value with the interior velocity (:181-188, no 'uv'/'un'/'flux' given).
 * =================================================================================================================== */
typedef struct {
    int use_lf;                /* use_lax_friedrichs_tracer                                */
    double lf_factor;          /* lax_friedrichs_tracer_scaling_factor                     */
    double velocity_factor;    /* tracer_advective_velocity_factor                         */
    const double *source;      /* [N][3] or NULL                                           */
    int n_markers;
    const int *bc_has_value;   /* per marker: 1 if a 'value' boundary condition is set     */
    const double *bc_value;    /* per marker                                               */
} swe2d_ref_tracer_t;

static void cell_tracer_tendency(const swe2d_ref_t *m, const swe2d_ref_tracer_t *tp, int k, const double *T,
                                 const double *uv, double dt, double *kT)
{
    const int npc = m->npc;
    const double *p = m->xy + 2*(size_t)npc*k;
    const double cf = tp->velocity_factor;
    double u[4], v[4], c[4];
    for (int i = 0; i < npc; i++) {
        u[i] = cf*uv[2*(size_t)npc*k + 2*i];
        v[i] = cf*uv[2*(size_t)npc*k + 2*i + 1];
        c[i] = T[(size_t)npc*k + i];
    }
    double b[4] = {0, 0, 0, 0};
    double A;
    const double *src = tp->source ? tp->source + (size_t)npc*k : 0;
    if (npc == 4) {
        const double ax = p[2] - p[0], ay = p[3] - p[1], bx = p[6] - p[0], by = p[7] - p[1];
        A = ax*by - ay*bx;
        const double xix = by/A, xiy = -bx/A, zex = -ay/A, zey = ax/A;
        for (int qi = 0; qi < 2; qi++) for (int qz = 0; qz < 2; qz++) {
            const double xi = GL_XI[qi], ze = GL_XI[qz], w = 0.25*A;
            const double phi[4] = {(1 - xi)*(1 - ze), xi*(1 - ze), xi*ze, (1 - xi)*ze};
            const double dxi[4] = {-(1 - ze), (1 - ze), ze, -ze};
            const double dze[4] = {-(1 - xi), -xi, xi, (1 - xi)};
            double gx[4], gy[4], uq = 0, vq = 0, cq = 0, divu = 0, sq = 0;
            for (int i = 0; i < 4; i++) {
                gx[i] = dxi[i]*xix + dze[i]*zex;
                gy[i] = dxi[i]*xiy + dze[i]*zey;
                uq += phi[i]*u[i]; vq += phi[i]*v[i]; cq += phi[i]*c[i];
                divu += gx[i]*u[i] + gy[i]*v[i];
                if (src) sq += phi[i]*src[i];
            }
            for (int i = 0; i < 4; i++) b[i] += w*((phi[i]*divu + uq*gx[i] + vq*gy[i])*cq + sq*phi[i]);
        }
    } else {
        A = 0.5*((p[2] - p[0])*(p[5] - p[1]) - (p[4] - p[0])*(p[3] - p[1]));
        double gx[3], gy[3];
        for (int i = 0; i < 3; i++) {
            int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
            gx[i] = (p[2*i1 + 1] - p[2*i2 + 1])/(2*A);
            gy[i] = (p[2*i2] - p[2*i1])/(2*A);
        }
        /* cell: +(phi div u + u.grad phi) c */
        double divu = 0;
        for (int i = 0; i < 3; i++) divu += gx[i]*u[i] + gy[i]*v[i];
        const double Iuc = int2(A, u, c), Ivc = int2(A, v, c);
        const double csum = c[0] + c[1] + c[2];
        for (int i = 0; i < 3; i++) b[i] += divu*A/12.0*(csum + c[i]) + gx[i]*Iuc + gy[i]*Ivc;
        if (src) {
            double ss = src[0] + src[1] + src[2];
            for (int i = 0; i < 3; i++) b[i] += A/12.0*(ss + src[i]);
        }
    }
    for (int f = 0; f < npc; f++) {
        const int a = f, bb = (f + 1) % npc;
        const double dx = p[2*bb] - p[2*a], dy = p[2*bb + 1] - p[2*a + 1];
        const double len = sqrt(dx*dx + dy*dy);
        const double nx = dy/len, ny = -dx/len;
        const int nb = m->nbr[(size_t)npc*k + f];
        double ua_n = 0, ub_n = 0, va_n = 0, vb_n = 0, ca_n = 0, cb_n = 0;
        if (nb >= 0) {
            const int f2 = m->nbf[(size_t)npc*k + f];
            const int na = (f2 + 1) % npc, nbb = f2;
            const size_t o = 2*(size_t)npc*nb, oc = (size_t)npc*nb;
            ua_n = cf*uv[o + 2*na];  va_n = cf*uv[o + 2*na + 1];  ca_n = T[oc + na];
            ub_n = cf*uv[o + 2*nbb]; vb_n = cf*uv[o + 2*nbb + 1]; cb_n = T[oc + nbb];
        }
        for (int q = 0; q < 2; q++) {
            const double xb = GL_XI[q], xa = 1.0 - xb, w = 0.5*len;
            const double uq = xa*u[a] + xb*u[bb], vq = xa*v[a] + xb*v[bb], cq = xa*c[a] + xb*c[bb];
            const double un_own = uq*nx + vq*ny;
            double fq;
            if (nb >= 0) {
                const double un_ = xa*ua_n + xb*ub_n, vn_ = xa*va_n + xb*vb_n, cn_ = xa*ca_n + xb*cb_n;
                const double un_av = 0.5*((uq + un_)*nx + (vq + vn_)*ny);      /* seen from this cell */
                const double c_up = un_av > 0 ? cq : (un_av < 0 ? cn_ : 0.5*(cq + cn_));
                fq = c_up*un_own;
                if (tp->use_lf) fq += 0.5*fabs(un_av)*tp->lf_factor*(cq - cn_);
            } else {
                const int marker = -nb;
                if (tp->bc_has_value && marker < tp->n_markers && tp->bc_has_value[marker]) {
                    const double c_up = un_own > 0 ? cq : (un_own < 0 ? tp->bc_value[marker] : 0.5*(cq + tp->bc_value[marker]));
                    fq = c_up*un_own;
                } else {
                    fq = cq*un_own;
                }
            }
            b[a] -= w*xa*fq; b[bb] -= w*xb*fq;
        }
    }
    if (npc == 4) {
        const double s4 = dt/A;
        for (int i = 0; i < 4; i++)
            kT[4*(size_t)k + i] = s4*(16.0*b[i] - 8.0*b[(i + 1) % 4] - 8.0*b[(i + 3) % 4] + 4.0*b[(i + 2) % 4]);
        return;
    }
    const double s = 3.0*dt/A, sb = b[0] + b[1] + b[2];
    for (int i = 0; i < 3; i++) kT[3*(size_t)k + i] = s*(4.0*b[i] - sb);
}

void swe2d_ref_tracer_tendency(const swe2d_ref_t *m, const swe2d_ref_tracer_t *tp, const double *T, const double *uv,
                               double dt, double *kT)
{
#pragma omp parallel for schedule(static)
    for (int k = 0; k < m->n_cells; k++) cell_tracer_tendency(m, tp, k, T, uv, dt, kT);
}

/* one tracer SSPRK33 step with frozen velocity; work holds 2*3*N doubles */
void swe2d_ref_tracer_step(const swe2d_ref_t *m, const swe2d_ref_tracer_t *tp, double *T, const double *uv, double dt,
                           double *work)
{
    const size_t n = (size_t)m->npc*(size_t)m->n_cells;
    double *T0 = work, *kT = work + n;
    static const double A30 = 0.33333333333333337, A32 = 0.6666666666666666, B32 = 0.6666666666666666;
    memcpy(T0, T, n*sizeof(double));
    swe2d_ref_tracer_tendency(m, tp, T, uv, dt, kT);
    for (size_t i = 0; i < n; i++) T[i] = kT[i]*1.0 + T0[i]*1.0;
    swe2d_ref_tracer_tendency(m, tp, T, uv, dt, kT);
    for (size_t i = 0; i < n; i++) T[i] = kT[i]*0.25 + T0[i]*0.75 + T[i]*0.25;
    swe2d_ref_tracer_tendency(m, tp, T, uv, dt, kT);
    for (size_t i = 0; i < n; i++) T[i] = kT[i]*B32 + T0[i]*A30 + T[i]*A32;
}

/* vertex-based limiter; cell_vertex = [N][3] topological vertex ids, qmin/qmax = work arrays of n_vertices */
void swe2d_ref_limit(const swe2d_ref_t *m, const int *cell_vertex, int n_vertices, double *T, double *qmin, double *qmax)
{
    const int n = m->n_cells, npc = m->npc;
    for (int v = 0; v < n_vertices; v++) { qmax[v] = -1.0e10; qmin[v] = 1.0e10; }
    for (int k = 0; k < n; k++) {
        const double *c = T + (size_t)npc*k;
        double mean = 0;
        for (int i = 0; i < npc; i++) mean += c[i];
        mean /= npc;
        for (int i = 0; i < npc; i++) {
            const int v = cell_vertex[(size_t)npc*k + i];
            qmax[v] = fmax(qmax[v], mean); qmin[v] = fmin(qmin[v], mean);
        }
    }
    for (int k = 0; k < n; k++)
        for (int f = 0; f < npc; f++)
            if (m->nbr[(size_t)npc*k + f] < 0) {
                const int a = f, b = (f + 1) % npc;
                const double fm = (T[(size_t)npc*k + a] + T[(size_t)npc*k + b])/2.0;
                const int va = cell_vertex[(size_t)npc*k + a], vb = cell_vertex[(size_t)npc*k + b];
                qmax[va] = fmax(qmax[va], fm); qmin[va] = fmin(qmin[va], fm);
                qmax[vb] = fmax(qmax[vb], fm); qmin[vb] = fmin(qmin[vb], fm);
            }
    for (int k = 0; k < n; k++) {
        double *c = T + (size_t)npc*k;
        double mean = 0;
        for (int i = 0; i < npc; i++) mean += c[i];
        mean /= npc;
        double alpha = 1.0;
        for (int i = 0; i < npc; i++) {
            const int v = cell_vertex[(size_t)npc*k + i];
            if (c[i] > mean) alpha = fmin(alpha, fmin(1.0, (qmax[v] - mean)/(c[i] - mean)));
            else if (c[i] < mean) alpha = fmin(alpha, fmin(1.0, (mean - qmin[v])/(mean - c[i])));
        }
        for (int i = 0; i < npc; i++) c[i] = mean + alpha*(c[i] - mean);
    }
}
