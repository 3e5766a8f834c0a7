// swe2d_kernels.h - device code of the MI355X (gfx950) DG-P1 shallow-water stage kernel.
//
// One launch = one SSPRK33 stage for a range of cells:   U_out = beta*k + a0*U0 + a1*U_in,
//   k = M^-1 (dt R(U_in)),  R = ExternalPressureGradient + HUDiv + HorizontalAdvection(+Lax-Friedrichs)
//   (+ Coriolis, drag, atmospheric pressure, sources) of thetis/shallowwater_eq.py:335-510,619-831,
//   M^-1 from thetis/equation.py:105, update from thetis/rungekutta.py:908-946.
//
// Design (HBM-bound FP64 gather/stream kernel; no dense contraction, hence no MFMA):
//  * one lane = one triangle; everything a cell needs is recomputed by the cell itself (both sides of an interior
//    facet evaluate the same numerical flux) -> no atomics, no inter-lane reduction, bitwise deterministic.
//  * state is 9 SoA planes (u0 u1 u2 v0 v1 v2 e0 e1 e2) of `stride` doubles: own-cell loads/stores are fully
//    coalesced 512-B wave transactions; neighbour traces are 8-B gathers that hit L2 because consecutive cells are
//    mesh neighbours and the block->cell-range map keeps each XCD on one contiguous chunk of the mesh.
//  * residual, 3x3 mass inverse and the Shu-Osher combine are fused: per stage each cell's state is read once
//    (+U0 once in stages 1,2) and written once: 180/252/252 algorithmic bytes.
//  * closed forms for the P1 cell integrals (no cell quadrature, no division: A*grad(phi_i) = -nF_{i+1}/2) and one
//    sqrt per facet quadrature point: g*sqrt(H/g) = H*sqrt(g/H) = sqrt(g*H).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "swe2d_conn.h"

#define SWE_MAX_MARKERS 16
#define SWE_BC_ELEV 1
#define SWE_BC_UV 2
#define SWE_BC_UN 4
#define SWE_BC_FLUX 8
#define SWE_BC_ELEV_FIELD 16     // the external elevation is a nodal field (Function-valued boundary data)
#define SWE_BC_UV_FIELD 32
#define SWE_BC_UN_FIELD 64
#define SWE_BC_FLUX_FIELD 128
#define SWE_BC_HAS_DRAG 256       // set by the host in the table handed to a launch: the marker has a boundary drag >= 0
#ifndef SWE_BLOCK
#define SWE_BLOCK 64
#endif
#ifndef SWE_MIN_WAVES
#define SWE_MIN_WAVES 1          // __launch_bounds__ 2nd argument: minimum waves per SIMD
#endif
#ifndef SWE_FAST_SQRT
#define SWE_FAST_SQRT 1          // rsq/rcp + Goldschmidt/Newton without the denormal-range scaling of sqrt()/operator/
#endif

struct SweBcTable {
    int kind[SWE_MAX_MARKERS];
    double elev[SWE_MAX_MARKERS];
    double u[SWE_MAX_MARKERS];
    double v[SWE_MAX_MARKERS];
    double un[SWE_MAX_MARKERS];
    double flux[SWE_MAX_MARKERS];
    double len[SWE_MAX_MARKERS];
    double drag[SWE_MAX_MARKERS];     // boundary drag C_D ('drag' key, BoundaryDragTerm), < 0: none
};

struct SweStageArgs {
    const double *uin;     // 9 planes, state entering the stage
    const double *u0;      // 9 planes, stage_sol[0]
    double *uout;          // 9 planes
    size_t stride;         // plane length (doubles)
    const int *nbr;        // 3 planes: (cell<<2)|facet_in_neighbour, or -(marker)
    const int *cv;         // 3 planes: vertex ids
    const double *vx, *vy, *vh;   // per vertex: coordinates, bathymetry
    const double *valpha;         // per vertex: wetting-drying parameter alpha (WD variant)
    // packed copy of the triangle connectivity for the stage kernel: idx4[k] = {nbr0, nbr1, nbr2, cv0}, idx2[k] = {cv1, cv2}
    // (one 16-B and one 8-B load per lane instead of six 4-B loads from six planes)
    const int4 *idx4;
    const int2 *idx2;
    // ... and the same in 16 B (swe_conn_pack / swe_conn_load below), or null: what the stage kernels read (idx4 / idx2 stay as the
    // escape for the few cells whose differences do not fit)
    const int4 *idxc;
    // horizontal viscosity fused into the triangle stage kernel (VISC variants; swe_visc_interior)
    const int4 *opp4;             // {vertex of neighbour 0 / 1 / 2 opposite the shared facet, 0}
    const double *nu_v;           // per-vertex viscosity or null (then nu_const)
    double nu_const, visc_sipg;   // visc_sipg = sipg_factor * cp, cp = 3
    int visc_grad_div, visc_grad_depth;
    int cell_begin, cell_end;
    int wd_skip_relax;            // wetting-drying + viscosity: the dry-ground relaxation of the velocity follows the viscosity pass
    int reverse;                  // walk the blocks of the range from its end (launches beyond the Infinity Cache alternate, see launch_stage)
    int wall_general;             // 1: closed walls through the general boundary path (SWE2D_OPT_WALL_FAST = 0: parity test of the wall path)
    double g, sigma_lf, dt;
    double a0, a1, beta;   // U_out = beta*k + a0*U0 + a1*U_in
    // optional cell-local terms (SRC variant)
    const double *coriolis;   // 3 planes or null
    const double *patm;       // 3 planes or null
    const double *msrc;       // 6 planes (x0 x1 x2 y0 y1 y2) or null
    const double *vsrc;       // 3 planes or null
    const double *wind;       // 6 planes (x0.. y0..) wind stress or null
    // Function-valued boundary data, stored PER FACET (a corner cell has two boundary facets with different markers that
    // share a node): plane 2f holds the value at the facet's first node f, plane 2f+1 at its second node f+1
    const double *bc_elev_f;  // 2k planes: external elevation, or null
    const double *bc_uv_f;    // 4k planes (u: 0..2k-1, v: 2k..4k-1)
    const double *bc_un_f;    // 2k planes
    const double *bc_flux_f;  // 2k planes
    int npc_;                 // nodes per cell (plane offsets of the vector boundary field)
    double linear_drag, quad_drag, manning, norm_smoother;   // <0: off
    double nikuradse;                                        // Nikuradse bed roughness length k_s, <0: off
    // spatially varying drag coefficients (k planes each, or null): the field replaces the scalar of its kind
    const double *lin_drag_f;
    const double *quad_f;     // quadratic C_D, Manning mu or Nikuradse k_s, according to quad_f_kind
    int quad_f_kind;          // 0 none, 1 quadratic, 2 Manning, 3 Nikuradse
    SweBcTable bc;
};

// 2-point Gauss-Legendre on [0,1] (facet rule of degree 3, shallowwater_eq.py:225-230 [FD-assumed])
#define SWE_XI0 0.21132486540518713
#define SWE_XI1 0.78867513459481287

// Block -> logical block: the dispatcher places block b on XCD b%8 (observed; used for speed only).  Give every XCD
// one contiguous eighth of the cell range so that facet-neighbour gathers are served by the XCD's own L2.
__device__ __forceinline__ int swe_logical_block(int b, int nblocks)
{
    const int per = (nblocks + 7) >> 3;
    return (b & 7)*per + (b >> 3);
}

// ---- Compact triangle connectivity (swe2d_conn.h): 16 B per cell instead of 24, the wide records where a difference does not fit
__device__ __forceinline__ void swe_conn_load(const int4 *idxc, const int4 *idx4, const int2 *idx2, int k, int nb[3], int vid[3])
{
    if (idxc) {                                          // uniform
        if (swe_conn_unpack(idxc[k], k, nb, vid)) {      // escape (rare): the wide records
            const int4 q4 = idx4[k];
            const int2 q2 = idx2[k];
            nb[0] = q4.x; nb[1] = q4.y; nb[2] = q4.z;
            vid[0] = q4.w; vid[1] = q2.x; vid[2] = q2.y;
        }
    } else {
        const int4 q4 = idx4[k];
        const int2 q2 = idx2[k];
        nb[0] = q4.x; nb[1] = q4.y; nb[2] = q4.z;
        vid[0] = q4.w; vid[1] = q2.x; vid[2] = q2.y;
    }
}

// sqrt(x) and 1/sqrt(x) for normal-range x > 0: v_rsq_f64 seed, one Goldschmidt iteration, two residual corrections
// (the sequence LLVM emits for f64 sqrt, minus its 2^-767 rescaling and class checks).  ~1 ulp.
__device__ __forceinline__ void swe_sqrt_rsqrt(double x, double &s, double &rs)
{
#if SWE_FAST_SQRT
    const double y = __builtin_amdgcn_rsq(x);
    double g = x*y, h = 0.5*y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double d0 = fma(-g, g, x);
    g = fma(d0, h, g);
    const double d1 = fma(-g, g, x);
    g = fma(d1, h, g);
    s = g;
    rs = h + h;
#else
    s = sqrt(x);
    rs = 1.0/s;
#endif
}

// sqrt(x) for x >= 0 (x == 0 -> 0; x < 0 -> NaN, like sqrt).  The seed is taken at x + 1e-300 (= x for every x >= 1e-283; below,
// the corrections still converge): v_rsq_f64 then gives a finite number at 0, which the sequence carries to g = 0 exactly, and NaN
// for negative x as before - one addition instead of a compare and two selects on the result.
__device__ __forceinline__ double swe_sqrt(double x)
{
#if SWE_FAST_SQRT
    const double y = __builtin_amdgcn_rsq(x + 1e-300);
    double g = x*y, h = 0.5*y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double d0 = fma(-g, g, x);
    g = fma(d0, h, g);
    const double d1 = fma(-g, g, x);
    return fma(d1, h, g);
#else
    return sqrt(x);
#endif
}

// sqrt(x) for x >= 0 that need not be exact below 1e-150 (sums of squares): the argument is clamped away from 0, where
// v_rsq_f64 gives inf, instead of the compare and the two selects of swe_sqrt; x = 0 -> 1e-150
__device__ __forceinline__ double swe_sqrt_sumsq(double x)
{
#if SWE_FAST_SQRT
    double s, rs;
    swe_sqrt_rsqrt(fmax(x, 1e-300), s, rs);
    return s;
#else
    return sqrt(x);
#endif
}

// 1/x for normal-range x: v_rcp_f64 seed + two Newton steps
__device__ __forceinline__ double swe_rcp(double x)
{
#if SWE_FAST_SQRT
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
#else
    return 1.0/x;
#endif
}

// Wetting-drying (explicit nodal formulation, see oracle/swe2d_oracle.py header and DESIGN.md): displaced total depth
// D = (H + sqrt(H^2 + a^2))/2 with H = h + eta (thetis/utility.py:975-993), and its inverse H = D - a^2/(4 D).
__device__ __forceinline__ double swe_wd_depth(double H, double a)
{
    return 0.5*(H + swe_sqrt_sumsq(fma(H, H, a*a)));
}
// ROUND 5: with wetting-drying the DEVICE carries D, not eta.  The "elevation" planes of the three state buffers hold the nodal
// displaced depth D (the quantity the explicit scheme advances: zeta = D - h), the elevation is recovered where the equations need
// it - the pressure gradient and the elevation jumps of the fluxes - by eta = D - alpha^2/(4 D) - h: a reciprocal, where the
// eta-carrying kernels of rounds 1-4 took twelve square roots per cell and stage to get D back (own nodes, U(0), six neighbour
// nodes) and three reciprocals to store eta again.  Host layout and ABI are unchanged: swe2d_set_state converts eta -> D (and brings
// it to the admissible set, swe_wd_clip_kernel), swe2d_get_state D -> eta (swe_planes_to_aos); eta -> D -> eta is the identity up
// to rounding (one ulp of D), not bit for bit - library-internal save / restore goes through swe2d_state_snapshot instead.
__device__ __forceinline__ double swe_wd_eta(double D, double h, double a)
{
    return D - 0.25*a*a*swe_rcp(D) - h;
}

// End of a wetting-drying stage in one cell (explicit formulation, oracle/swe2d_oracle.py module docstring and
// SWEOracle.wd_finish_stage): oe[] holds zeta = D - h on entry and the limited depth D on exit (the state the device carries).
//  (1) positivity limiter on the nodal depths: deviations from the cell mean scaled so that every node keeps
//      D >= SWE_WD_FLOOR * alpha (mean unchanged = conservative); a cell whose mean is below the floor is flattened to its
//      mean, below a tenth of the floor raised to that;
//  (2) [eta = D - alpha^2/(4 D) - h is no longer formed here: the device carries D]
//  (3) relaxation of the velocity on dry ground: u *= exp(-dt_stage/tau psi^2), tau = SWE_WD_TAU sqrt(alpha/g),
//      psi = clamp(-H/alpha - 1, 0, 1), H = h + eta = D - alpha^2/(4 D).
#define SWE_WD_FLOOR 0.1
#define SWE_WD_TAU 10.0
// ``mw``: weights of the cell mean (general quadrilaterals: int phi_i dx / area, swe_quad_mean_weights); nullptr: 1/K.
template <int K>
__device__ __forceinline__ void swe_wd_finish(double g, double dt_stage, const double h[K], const double al[K], double ou[K],
                                              double ov[K], double oe[K], bool relax = true, const double *mw = nullptr)
{
#pragma clang fp contract(off)
    double D[K], mean = 0.0, dmin = 1e300, fl = 0.0;
#pragma unroll
    for (int i = 0; i < K; i++) {
        D[i] = oe[i] + h[i];
        mean += mw ? mw[i]*D[i] : D[i];
        dmin = fmin(dmin, D[i]);
        fl = fmax(fl, SWE_WD_FLOOR*al[i]);
    }
    if (!mw) mean *= (K == 3 ? (1.0/3.0) : 0.25);
    if (dmin < fl) {
        if (mean <= fl) {
            const double flat = fmax(mean, 0.1*fl);
#pragma unroll
            for (int i = 0; i < K; i++) D[i] = flat;
        } else {
            const double theta = (mean - fl)*swe_rcp(mean - dmin);
#pragma unroll
            for (int i = 0; i < K; i++) D[i] = mean + theta*(D[i] - mean);
        }
    }
#pragma unroll
    for (int i = 0; i < K; i++) {
        oe[i] = D[i];
        if (!relax) continue;      // viscous runs: the relaxation follows the viscosity pass (swe_wd_relax_kernel)
        // psi > 0 <=> the water table H = D - alpha^2/(4 D) lies more than alpha below the bed <=> alpha^2 > 4 D (D + alpha)
        // (D > 0): decided without a quotient, which only dry nodes then pay for (a node within rounding of the threshold gets
        // exp(-O(1e-32)) = 1 from the formula either way)
        if (al[i]*al[i] > 4.0*D[i]*(D[i] + al[i])) {
            const double ral = swe_rcp(al[i]);
            const double Hw = D[i] - 0.25*al[i]*al[i]*swe_rcp(D[i]);
            const double psi = fmin(1.0, fmax(0.0, -Hw*ral - 1.0));
            const double fac = exp(-dt_stage*(1.0/SWE_WD_TAU)*swe_sqrt(g*ral)*psi*psi);     // 1/sqrt(alpha/g) = sqrt(g/alpha)
            ou[i] *= fac;
            ov[i] *= fac;
        }
    }
}

// x^(-1/3) for normal-range x > 0 (Manning: C_D = g mu^2 / H^(1/3), shallowwater_eq.py:693): f32 seed through
// v_log_f32 / v_exp_f32 (1e-7), then ONE step of third order: with r = 1 - x y^3 the root is y (1 - r)^(-1/3) =
// y (1 + r/3 + 2 r^2/9 + 14 r^3/81 ...), truncated after r^2 (remainder 14/81 (3e-7)^3 ~ 5e-21; r itself carries 2e-16 from the
// two rounded products) - six FP64 instructions where two Newton steps took ten
__device__ __forceinline__ double swe_rcbrt(double x)
{
#if SWE_FAST_SQRT
    const float lf = __builtin_amdgcn_logf((float)x);              // log2
    const double y = (double)__builtin_amdgcn_exp2f(-0.33333334f*lf);
    const double r = fma(-(x*y), y*y, 1.0);
    return fma(y*r, fma(r, 2.0/9.0, 1.0/3.0), y);
#else
    return 1.0/cbrt(x);
#endif
}

// total depth of a pointwise (external / Riemann) state
template <bool NONLIN, bool WD>
__device__ __forceinline__ double swe_depth_pt(double h, double eta, double a)
{
    return WD ? swe_wd_depth(h + eta, a) : (NONLIN ? h + eta : h);
}

// a*b + c*d with the contraction spelled out.  The boundary code below is inlined into two variants of the stage kernel (and
// into the quadrilateral kernel); a sum of two products leaves it to the compiler which product is fused, and it chooses
// differently from one context to the other - one ulp, which would break the bitwise agreement of the variants.
__device__ __forceinline__ double swe_dot2(double a, double b, double c, double d) { return fma(a, b, c*d); }

// 12/A * int a*b dx for P1 a, b
__device__ __forceinline__ double swe_int2(const double a[3], const double b[3])
{
    return fma(a[2], b[2], fma(a[1], b[1], fma(a[0], b[0], (a[0] + a[1] + a[2])*(b[0] + b[1] + b[2]))));
}

// A P1 trace (A at the facet's first node, B at its second) at the two Gauss points: the first lies SWE_XI0 from A, the second
// SWE_XI0 from B.  One difference and two fmas for the pair (xa*A + xb*B per point takes four).
__device__ __forceinline__ void swe_gl2(double A, double B, double &p0, double &p1)
{
    const double D = B - A;
    p0 = fma(SWE_XI0, D, A);
    p1 = fma(-SWE_XI0, D, B);
}

// The numerical fluxes of ONE interior facet seen from this cell, integrated against the facet's two nodal basis functions with the
// two-point rule (shallowwater_eq.py:360-366, :421-427, :480-488): shared by the stage kernels and the dataflow kernel
// (swe2d_flow.h), which must give the same bits.  (.)a / (.)b: this cell's values at the facet's nodes, (.)na / (.)nb: the
// neighbour's on the same nodes; Ha, Hb, Dna, Dnb: nodal total depths of both sides (wetting-drying only); nxs, nys = |F| n, L = |F|.
// The jump and the average come from one interpolation each: [u] = u - u_n, {u} = u - [u]/2.
template <bool NONLIN, bool LF, bool WD>
__device__ __forceinline__ void swe_facet_flux(double g, double sigma_lf, double ua, double ub, double va, double vb, double ea, double eb,
                                               double ha, double hb, double Ha, double Hb, double una, double unb, double vna, double vnb,
                                               double ena, double enb, double Dna, double Dnb, double nxs, double nys, double L, double rL,
                                               double &Fau, double &Fbu, double &Fav, double &Fbv, double &Fae, double &Fbe)
{
#pragma clang fp contract(off)
    double uq[2], vq[2], eq[2], hq[2], un[2], vn[2], en[2], Hs[2] = {0.0, 0.0};
    swe_gl2(ua, ub, uq[0], uq[1]);
    swe_gl2(va, vb, vq[0], vq[1]);
    swe_gl2(ea, eb, eq[0], eq[1]);
    swe_gl2(una, unb, un[0], un[1]);
    swe_gl2(vna, vnb, vn[0], vn[1]);
    swe_gl2(ena, enb, en[0], en[1]);
    if (WD) swe_gl2(0.5*(Ha + Dna), 0.5*(Hb + Dnb), Hs[0], Hs[1]);      // {H} of the nodal displaced depths
    else swe_gl2(ha, hb, hq[0], hq[1]);
    Fau = 0.0; Fbu = 0.0; Fav = 0.0; Fbv = 0.0; Fae = 0.0; Fbe = 0.0;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const double xb = q ? SWE_XI1 : SWE_XI0, xa = 1.0 - xb;
        const double de = eq[q] - en[q];
        const double eav = fma(-0.5, de, eq[q]);
        const double Hav = WD ? Hs[q] : (NONLIN ? hq[q] + eav : hq[q]);
        const double c = swe_sqrt(g*Hav);
        const double du = uq[q] - un[q], dv = vq[q] - vn[q];
        const double dun = swe_dot2(du, nxs, dv, nys);            // |F| jump(u.n)
        const double spg = fma(c*dun, rL, g*eav);                 // g*head_star            :363
        double fu = spg*nxs, fv = spg*nys;                        //                        :366
        const double uav = fma(-0.5, du, uq[q]), vav = fma(-0.5, dv, vq[q]);
        const double uavn = swe_dot2(uav, nxs, vav, nys);         // |F| {u}.n
        const double fe = fma(c*de, L, Hav*uavn);                 // {H}({u}+sqrt(g/{H})[eta n]).n  :424-427
        if (NONLIN) {
            const double unown = swe_dot2(uq[q], nxs, vq[q], nys);
            fu = fma(uav, unown, fu);                             //                        :483
            fv = fma(vav, unown, fv);
            if (LF) {
                const double gam = 0.5*fabs(uavn)*sigma_lf;       //                        :487
                fu = fma(gam, du, fu);                            //                        :488
                fv = fma(gam, dv, fv);
            }
        }
        Fau = fma(xa, fu, Fau); Fbu = fma(xb, fu, Fbu);
        Fav = fma(xa, fv, Fav); Fbv = fma(xb, fv, Fbv);
        Fae = fma(xa, fe, Fae); Fbe = fma(xb, fe, Fbe);
    }
}

// Boundary facet (closed wall or open boundary); rare, so written for clarity with unit normals.
// Returns the form values f (residual is -f) already multiplied by the facet length.
struct SweBcFieldValues { double elev, u, v, un, flux; };   // Function-valued boundary data at the quadrature point

template <bool NONLIN, bool LF, bool WD>
__device__ __forceinline__ void swe_boundary_flux(const SweStageArgs &p, int marker, double uq, double vq, double eq,
                                               double hq, double Hq, double alq, double nxs, double nys, double L,
                                               double rL, const SweBcFieldValues &bf, double &fu, double &fv, double &fe,
                                               int kind_all)
{
    // No implicit contraction in the boundary code: it is inlined into several kernels (epilogue / inline-boundary variants,
    // quadrilaterals), which must agree bit for bit, and where the compiler fuses depends on the surrounding code.  Every
    // operation below is individually rounded; fma() is written out where wanted.
#pragma clang fp contract(off)
    const double g = p.g;
    // square roots and quotients through the v_rsq / v_rcp based helpers (~10 instructions each) instead of the IEEE library
    // sequences (~40-60): a wave that owns one boundary cell executes this code for all its lanes, and on a small partition
    // a quarter of the waves do
    const double rg = swe_rcp(g);
    const double nx = nxs*rL, ny = nys*rL;
    // kind_all: the marker's entry of the table incl. SWE_BC_HAS_DRAG, read ONCE per facet by the caller (an entry of the
    // kernel-argument table indexed per lane is a load from memory: here it would be a dependent one per use)
    const int kind = kind_all & 0xff;
    const double un_own = swe_dot2(uq, nx, vq, ny);
    if (kind == 0) {
        // land boundary, shallowwater_eq.py:377-381 and :489-497
        const double head_rie = eq + swe_sqrt(Hq*rg)*un_own;
        fu = g*head_rie*nx;
        fv = g*head_rie*ny;
        fe = 0.0;
        if (NONLIN && LF) {
            const double gamma = 0.5*fabs(un_own)*p.sigma_lf;
            fu += gamma*2.0*un_own*nx;
            fv += gamma*2.0*un_own*ny;
        }
    } else {
        // external state, get_bnd_functions shallowwater_eq.py:243-267
        double e_ext = eq, u_ext = uq, v_ext = vq;
        if (kind & SWE_BC_ELEV) e_ext = (kind & SWE_BC_ELEV_FIELD) ? bf.elev : p.bc.elev[marker];
        if (kind & SWE_BC_UV) {
            u_ext = (kind & SWE_BC_UV_FIELD) ? bf.u : p.bc.u[marker];
            v_ext = (kind & SWE_BC_UV_FIELD) ? bf.v : p.bc.v[marker];
        } else if (kind & SWE_BC_UN) {
            const double un_ext = (kind & SWE_BC_UN_FIELD) ? bf.un : p.bc.un[marker];
            u_ext = un_ext*nx;
            v_ext = un_ext*ny;
        } else if (kind & SWE_BC_FLUX) {
            const double H0 = swe_depth_pt<NONLIN, WD>(hq, e_ext, alq);
            const double s = ((kind & SWE_BC_FLUX_FIELD) ? bf.flux : p.bc.flux[marker])*swe_rcp(H0*p.bc.len[marker]);
            u_ext = s*nx;
            v_ext = s*ny;
        }
        const double H_ext = swe_depth_pt<NONLIN, WD>(hq, e_ext, alq);
        const double un_jump = swe_dot2(uq - u_ext, nx, vq - v_ext, ny);
        double sq_Hg, sq_gH;                                                           // sqrt(H/g), sqrt(g/H)
        swe_sqrt_rsqrt(Hq*rg, sq_Hg, sq_gH);
        const double eta_rie = 0.5*(eq + e_ext) + sq_Hg*un_jump;                       // :374
        fu = g*eta_rie*nx;
        fv = g*eta_rie*ny;
        const double h_av = 0.5*(Hq + H_ext);
        const double eta_jump = eq - e_ext;
        const double un_avg = 0.5*swe_dot2(uq + u_ext, nx, vq + v_ext, ny);
        double sq_hg, sq_gh;                                                           // sqrt(h_av/g), sqrt(g/h_av)
        swe_sqrt_rsqrt(h_av*rg, sq_hg, sq_gh);
        const double un_rie = un_avg + sq_gh*eta_jump;                                 // :438
        const double eta_rie2 = 0.5*(eq + e_ext) + sq_hg*un_jump;                      // :440
        fe = swe_depth_pt<NONLIN, WD>(hq, eta_rie2, alq)*un_rie;                       // :441-442
        if (NONLIN) {
            const double un_rie3 = un_avg + sq_gH*eta_jump;                            // :507
            fu += un_rie3*0.5*(u_ext + uq);
            fv += un_rie3*0.5*(v_ext + vq);
        }
    }
    if (kind_all & SWE_BC_HAS_DRAG) {                                                  // BoundaryDragTerm :717-724
        const double cdb = p.bc.drag[marker];
        const double utx = uq - un_own*nx, uty = vq - un_own*ny;
        const double mag = swe_sqrt(swe_dot2(utx, utx, uty, uty));
        fu += cdb*mag*utx;
        fv += cdb*mag*uty;
    }
    fu *= L;
    fv *= L;
    fe *= L;
}

// Both quadrature points of a boundary facet; kept out of line of the interior fast path (few cells take it).
template <bool NONLIN, bool LF, bool WD, bool WALLFAST = true>
__device__ __forceinline__ void swe_boundary_facet(const SweStageArgs &p, int marker, int k, int a, int b, double ua,
                                                   double ub, double va,
                                                   double vb, double ea, double eb, double ha, double hb, double Ha,
                                                   double Hb, double ala, double alb, double nxs,
                                                   double nys, double L, double rL, double &Fau, double &Fbu,
                                                   double &Fav, double &Fbv, double &Fae, double &Fbe, int kind_in = -1)
{
#pragma clang fp contract(off)
    // Function-valued boundary data live on the same DG nodes as the state: read the two facet nodes of this cell
    // (kind_in >= 0: the marker's table entry, already fetched by the caller together with its other loads)
    const int kind_all = kind_in >= 0 ? kind_in : ((marker < SWE_MAX_MARKERS) ? p.bc.kind[marker] : 0);
    const int kind = kind_all & 0xff;
    // (not in the -DSWE_RANGE_CHECK build: there the kernels call a non-inlined checker, and with this early return the
    //  quadrilateral wetting-drying + source-term instance produced NaN in its third stage - a code generation difference that was
    //  not chased; the path touches no memory, so the checked build loses nothing by taking the general one)
#if !defined(SWE_NO_WALL_FAST_PATH) && !defined(SWE_RANGE_CHECK)
    if (WALLFAST && kind_all == 0 && !p.wall_general) {
        // Closed wall without boundary drag - what most boundary facets are, and in a dataflow launch the blocks that own them set
        // the pace: the land branch of swe_boundary_flux (shallowwater_eq.py:377-381, :489-497) operation for operation, without
        // the interpolation of boundary data that is not there and the tests of the boundary kind.  Same bits as the general path.
        const double g = p.g;
        const double rg = swe_rcp(g);
        const double nx = nxs*rL, ny = nys*rL;
#pragma unroll 1
        for (int q = 0; q < 2; q++) {
            const double xb = q ? SWE_XI1 : SWE_XI0, xa = 1.0 - xb;
            const double uq = swe_dot2(xa, ua, xb, ub), vq = swe_dot2(xa, va, xb, vb), eq = swe_dot2(xa, ea, xb, eb);
            const double Hq = swe_dot2(xa, Ha, xb, Hb);
            const double un_own = swe_dot2(uq, nx, vq, ny);
            const double head_rie = eq + swe_sqrt(Hq*rg)*un_own;
            double fu = g*head_rie*nx;
            double fv = g*head_rie*ny;
            if (NONLIN && LF) {
                const double gamma = 0.5*fabs(un_own)*p.sigma_lf;
                fu += gamma*2.0*un_own*nx;
                fv += gamma*2.0*un_own*ny;
            }
            fu *= L;
            fv *= L;
            Fau += xa*fu; Fbu += xb*fu;
            Fav += xa*fv; Fbv += xb*fv;
        }
        return;
    }
#endif
    const size_t S = p.stride;
    double fea = 0.0, feb = 0.0, fua = 0.0, fub = 0.0, fva = 0.0, fvb = 0.0, fna = 0.0, fnb = 0.0, fxa = 0.0, fxb = 0.0;
    const size_t pa = (size_t)(2*a)*S + k, pb = pa + S;            // per-facet planes: facet index = first node a
    if ((kind & SWE_BC_ELEV_FIELD) && p.bc_elev_f) { fea = p.bc_elev_f[pa]; feb = p.bc_elev_f[pb]; }
    if ((kind & SWE_BC_UV_FIELD) && p.bc_uv_f) {
        const size_t ov = (size_t)(2*p.npc_)*S;
        fua = p.bc_uv_f[pa]; fub = p.bc_uv_f[pb];
        fva = p.bc_uv_f[ov + pa]; fvb = p.bc_uv_f[ov + pb];
    }
    if ((kind & SWE_BC_UN_FIELD) && p.bc_un_f) { fna = p.bc_un_f[pa]; fnb = p.bc_un_f[pb]; }
    if ((kind & SWE_BC_FLUX_FIELD) && p.bc_flux_f) { fxa = p.bc_flux_f[pa]; fxb = p.bc_flux_f[pb]; }
#pragma unroll 1
    for (int q = 0; q < 2; q++) {
        const double xb = q ? SWE_XI1 : SWE_XI0, xa = 1.0 - xb;
        SweBcFieldValues bf;
        bf.elev = swe_dot2(xa, fea, xb, feb); bf.u = swe_dot2(xa, fua, xb, fub); bf.v = swe_dot2(xa, fva, xb, fvb);
        bf.un = swe_dot2(xa, fna, xb, fnb);
        bf.flux = swe_dot2(xa, fxa, xb, fxb);
        const double uq = swe_dot2(xa, ua, xb, ub), vq = swe_dot2(xa, va, xb, vb), eq = swe_dot2(xa, ea, xb, eb),
                     hq = swe_dot2(xa, ha, xb, hb);
        const double Hq = swe_dot2(xa, Ha, xb, Hb), alq = swe_dot2(xa, ala, xb, alb);   // Ha, Hb: nodal total depth (h, h + eta or D)
        double fu, fv, fe;
        swe_boundary_flux<NONLIN, LF, WD>(p, marker, uq, vq, eq, hq, Hq, alq, nxs, nys, L, rL, bf, fu, fv, fe, kind_all);
        Fau += xa*fu; Fbu += xb*fu;
        Fav += xa*fv; Fbv += xb*fv;
        Fae += xa*fe; Fbe += xb*fe;
    }
}

// Raw buffer addressing (SGPR resource + 32-bit per-lane BYTE offset + uniform SGPR byte offset): one VGPR offset serves
// every plane of a cell, the plane offsets stay in SGPRs, and no 64-bit per-lane address arithmetic is issued.
// A resource spans 4 GiB: one resource per group of three planes, swe2d_create rejects 3*stride*8 >= 2^32.
typedef unsigned int swe_u32x2 __attribute__((ext_vector_type(2)));
#ifndef SWE_ST_AUX
#define SWE_ST_AUX 0             // cache policy of the state stores (experiments: 16 = sc1 write-through, 2 = nt)
#endif
#ifndef SWE_RANGE_CHECK
typedef __amdgpu_buffer_rsrc_t swe_rsrc_t;
__device__ __forceinline__ swe_rsrc_t swe_rsrc(const void *base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0xffffffff, 0x00020000);
}
__device__ __forceinline__ double swe_ld(swe_rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ int swe_ldi(swe_rsrc_t r, unsigned voff, unsigned soff)
{
    return (int)__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
}
__device__ __forceinline__ void swe_st(swe_rsrc_t r, unsigned voff, unsigned soff, double x)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(swe_u32x2, x), r, voff, soff, SWE_ST_AUX);
}
#else
// Range-checked build (tools/range_check.sh; never the shipped library): every raw-buffer access of the stage, tracer and
// viscosity kernels is tested against the table of the library's own device allocations (exact requested sizes, filled by
// the host before each launch).  An access outside every allocation is counted, its address and source line recorded, and
// not performed.  The raw-buffer resources of the product build span 4 GiB, i.e. the hardware checks nothing there.
#define SWE_CHK_MAX 4096
struct SweChkTable { unsigned long long lo[SWE_CHK_MAX], hi[SWE_CHK_MAX]; int n; };
__device__ SweChkTable swe_chk_tab;                          // sorted by lo, disjoint
__device__ unsigned long long swe_chk_report[4];             // violations, first address, its source line, (host) checked launches
struct swe_rsrc_t { __amdgpu_buffer_rsrc_t r; unsigned long long base; };
__device__ __forceinline__ swe_rsrc_t swe_rsrc(const void *base)
{
    swe_rsrc_t x;
    x.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0xffffffff, 0x00020000);
    x.base = (unsigned long long)base;
    return x;
}
__device__ __noinline__ bool swe_chk(unsigned long long addr, unsigned bytes, int line)
{
    int lo = 0, hi = swe_chk_tab.n;                          // last entry with lo <= addr
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (swe_chk_tab.lo[mid] <= addr) lo = mid; else hi = mid; }
    const bool ok = swe_chk_tab.n > 0 && swe_chk_tab.lo[lo] <= addr && addr + bytes <= swe_chk_tab.hi[lo];
    if (!ok && atomicAdd(&swe_chk_report[0], 1ull) == 0ull) { swe_chk_report[1] = addr; swe_chk_report[2] = (unsigned long long)line; }
    return ok;
}
__device__ __forceinline__ double swe_ld_chk(swe_rsrc_t r, unsigned voff, unsigned soff, int line)
{
    if (!swe_chk(r.base + voff + soff, 8, line)) return 0.0;
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r.r, voff, soff, 0));
}
__device__ __forceinline__ int swe_ldi_chk(swe_rsrc_t r, unsigned voff, unsigned soff, int line)
{
    if (!swe_chk(r.base + voff + soff, 4, line)) return 0;
    return (int)__builtin_amdgcn_raw_buffer_load_b32(r.r, voff, soff, 0);
}
__device__ __forceinline__ void swe_st_chk(swe_rsrc_t r, unsigned voff, unsigned soff, double x, int line)
{
    if (!swe_chk(r.base + voff + soff, 8, line)) return;
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(swe_u32x2, x), r.r, voff, soff, SWE_ST_AUX);
}
#define swe_ld(r, v, s) swe_ld_chk(r, v, s, __LINE__)
#define swe_ldi(r, v, s) swe_ldi_chk(r, v, s, __LINE__)
#define swe_st(r, v, s, x) swe_st_chk(r, v, s, x, __LINE__)
#endif

// Optional cell-local terms (SRC kernel variant): Coriolis, linear / quadratic / Manning drag, atmospheric pressure
// gradient, momentum and volume sources.  b-vectors are the assembled integrals (before the mass inverse).
__device__ __forceinline__ void swe_source_terms(const SweStageArgs &p, int k, size_t S, double twoA, const double u[3],
                                                 const double v[3], const double H[3], const double gxs[3],
                                                 const double gys[3], double bu[3], double bv[3], double be[3])
{
    // No implicit contraction (the pragma of a kernel does not reach into the functions inlined into it): this code is inlined
    // into every SRC variant of the stage kernels and into the step kernel, which must all give the same bits.
#pragma clang fp contract(off)
    const double g = p.g;
    const double A = 0.5*twoA;
    const unsigned S8 = (unsigned)S*8u, k8 = (unsigned)k*8u;
    const double us = u[0] + u[1] + u[2], vs = v[0] + v[1] + v[2];
    const double A60 = A*(1.0/60.0), A12 = A*(1.0/12.0);
    if (p.coriolis) {                                    // shallowwater_eq.py:632-633
        double f[3];
#pragma unroll
        for (int i = 0; i < 3; i++) f[i] = swe_ld(swe_rsrc(p.coriolis), k8, i*S8);
        const double fs = f[0] + f[1] + f[2];
        const double fu_ = fma(f[2], u[2], fma(f[1], u[1], f[0]*u[0])), fv_ = fma(f[2], v[2], fma(f[1], v[1], f[0]*v[0]));
#pragma unroll
        for (int i = 0; i < 3; i++) {
            // 60/A int phi_i f w = fs*ws + sum f_a w_a + f_i*ws + w_i*fs + 2 f_i w_i
            const double tv = fma(2.0*f[i], v[i], fma(v[i], fs, fma(f[i], vs, fma(fs, vs, fv_))));
            const double tu = fma(2.0*f[i], u[i], fma(u[i], fs, fma(f[i], us, fma(fs, us, fu_))));
            bu[i] = fma(A60, tv, bu[i]);
            bv[i] = fma(-A60, tu, bv[i]);
        }
    }
    if (p.lin_drag_f) {                                  // LinearDragTerm with a P1 coefficient: int phi_i c w, cubic
        double c[3];
#pragma unroll
        for (int i = 0; i < 3; i++) c[i] = swe_ld(swe_rsrc(p.lin_drag_f), k8, i*S8);
        const double cs = c[0] + c[1] + c[2];
        const double cu_ = fma(c[2], u[2], fma(c[1], u[1], c[0]*u[0])), cv_ = fma(c[2], v[2], fma(c[1], v[1], c[0]*v[0]));
#pragma unroll
        for (int i = 0; i < 3; i++) {
            bu[i] = fma(-A60, fma(2.0*c[i], u[i], fma(u[i], cs, fma(c[i], us, fma(cs, us, cu_)))), bu[i]);
            bv[i] = fma(-A60, fma(2.0*c[i], v[i], fma(v[i], cs, fma(c[i], vs, fma(cs, vs, cv_)))), bv[i]);
        }
    } else if (p.linear_drag >= 0.0) {                   // shallowwater_eq.py:738
        const double cA = p.linear_drag*A12;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            bu[i] = fma(-cA, us + u[i], bu[i]);
            bv[i] = fma(-cA, vs + v[i], bv[i]);
        }
    }
    if (p.quad_drag >= 0.0 || p.manning >= 0.0 || p.nikuradse >= 0.0 || p.quad_f) {   // shallowwater_eq.py:685-700, 6-point rule
        // The rule's points come in two orbits of three, barycentric (a, a, a) with one entry replaced by b.  A P1 field at the
        // point that has b at node i is a*(x_0 + x_1 + x_2) + (b - a)*x_i, and the three points' contributions s_q w_q to the
        // node integrals add up to a*(s_0 + s_1 + s_2) + (b - a)*s_i: one fma per field and point instead of three, and two
        // instructions per node, orbit and component instead of three fmas per node, POINT and component (the kernel's time
        // follows its instruction count on meshes that sit in the Infinity Cache).
        const double a1 = 0.445948490915965, b1 = 0.108103018168070, w1 = 0.223381589678011;
        const double a2 = 0.091576213509771, b2 = 0.816847572980459, w2 = 0.109951743655322;
        const bool fld = p.quad_f != nullptr;
        const int kind = fld ? p.quad_f_kind : (p.manning >= 0.0 ? 2 : (p.nikuradse >= 0.0 ? 3 : 1));
        const double Hs = H[0] + H[1] + H[2];
        const double sm2 = p.norm_smoother*p.norm_smoother;
        if (!fld && kind == 2) {
            // Manning with a constant coefficient (cfg 5, options.manning_drag_coefficient = Constant): a path of its own, so that
            // neither the selects between field and constant nor the other laws' code sit between its instructions
            const double gm2 = g*p.manning*p.manning;
#pragma unroll
            for (int o = 0; o < 2; o++) {
                const double aa = o ? a2 : a1, dd = o ? b2 - a2 : b1 - a1, wAc = (o ? w2 : w1)*A*gm2;
                const double au = aa*us, av = aa*vs, aH = aa*Hs;
                double su[3], sv[3];
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const double uq = fma(dd, u[i], au), vq = fma(dd, v[i], av), Hq = fma(dd, H[i], aH);
                    const double y = swe_rcbrt(Hq), y2 = y*y;          // C_D/H = g mu^2 H^(-4/3) = g mu^2 (H^(-1/3))^4: no reciprocal
                    const double s = (wAc*(y2*y2))*swe_sqrt_sumsq(fma(uq, uq, fma(vq, vq, sm2)));
                    su[i] = s*uq;
                    sv[i] = s*vq;
                }
                const double aSu = aa*(su[0] + su[1] + su[2]), aSv = aa*(sv[0] + sv[1] + sv[2]);
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    bu[i] -= fma(dd, su[i], aSu);
                    bv[i] -= fma(dd, sv[i], aSv);
                }
            }
        } else {
            const double c0 = kind == 2 ? p.manning : (kind == 3 ? p.nikuradse : p.quad_drag);
            double cf[3] = {0.0, 0.0, 0.0};
            if (fld) {
#pragma unroll
                for (int i = 0; i < 3; i++) cf[i] = swe_ld(swe_rsrc(p.quad_f), k8, i*S8);
            }
            const double cfs = cf[0] + cf[1] + cf[2];
#pragma unroll
            for (int o = 0; o < 2; o++) {
                const double aa = o ? a2 : a1, dd = o ? b2 - a2 : b1 - a1, wA = (o ? w2 : w1)*A;
                const double au = aa*us, av = aa*vs, aH = aa*Hs, ac = aa*cfs;
                double su[3], sv[3];
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const double uq = fma(dd, u[i], au), vq = fma(dd, v[i], av), Hq = fma(dd, H[i], aH);
                    const double coef = fld ? fma(dd, cf[i], ac) : c0;      // field coefficient at the point
                    double cdh;                              // C_D / H
                    if (kind == 2) {                         // Manning with a coefficient field
                        const double y = swe_rcbrt(Hq), y2 = y*y;
                        cdh = g*coef*coef*(y2*y2);
                    } else {
                        double cd = coef;
                        if (kind == 3) {                     // C_D = 2 kappa^2 / ln(11.036 H/k_s)^2 for H > k_s, else 0   :696-697
                            const double lg = log(11.036*Hq/coef);
                            cd = (Hq > coef) ? 0.32/(lg*lg) : 0.0;
                        }
                        cdh = cd*swe_rcp(Hq);
                    }
                    const double s = wA*cdh*swe_sqrt_sumsq(fma(uq, uq, fma(vq, vq, sm2)));
                    su[i] = s*uq;
                    sv[i] = s*vq;
                }
                const double aSu = aa*(su[0] + su[1] + su[2]), aSv = aa*(sv[0] + sv[1] + sv[2]);
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    bu[i] -= fma(dd, su[i], aSu);
                    bv[i] -= fma(dd, sv[i], aSv);
                }
            }
        }
    }
    if (p.wind) {                                        // shallowwater_eq.py:648, + tau.psi/(H rho0), 6-point rule
        // (by orbits, like the quadratic drag above; the quotient through the v_rcp helper)
        const double a1 = 0.445948490915965, b1 = 0.108103018168070, w1 = 0.223381589678011;
        const double a2 = 0.091576213509771, b2 = 0.816847572980459, w2 = 0.109951743655322;
        double tx[3], ty[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            tx[i] = swe_ld(swe_rsrc(p.wind), k8, i*S8);
            ty[i] = swe_ld(swe_rsrc(p.wind + 3*S), k8, i*S8);
        }
        const double txs = tx[0] + tx[1] + tx[2], tys = ty[0] + ty[1] + ty[2], Hs = H[0] + H[1] + H[2];
#pragma unroll
        for (int o = 0; o < 2; o++) {
            const double aa = o ? a2 : a1, dd = o ? b2 - a2 : b1 - a1, wA = (o ? w2 : w1)*A*(1.0/1000.0);
            const double ax = aa*txs, ay = aa*tys, aH = aa*Hs;
            double wx[3], wy[3];
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const double s = wA*swe_rcp(fma(dd, H[i], aH));
                wx[i] = s*fma(dd, tx[i], ax);
                wy[i] = s*fma(dd, ty[i], ay);
            }
            const double aSx = aa*(wx[0] + wx[1] + wx[2]), aSy = aa*(wy[0] + wy[1] + wy[2]);
#pragma unroll
            for (int i = 0; i < 3; i++) {
                bu[i] += fma(dd, wx[i], aSx);
                bv[i] += fma(dd, wy[i], aSy);
            }
        }
    }
    if (p.patm) {                                        // shallowwater_eq.py:662, rho0 = 1000
        double gpx = 0.0, gpy = 0.0;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const double pa = swe_ld(swe_rsrc(p.patm), k8, i*S8);
            gpx = fma(gxs[i], pa, gpx);
            gpy = fma(gys[i], pa, gpy);
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            bu[i] = fma(-gpx, 1.0/3000.0, bu[i]);
            bv[i] = fma(-gpy, 1.0/3000.0, bv[i]);
        }
    }
    if (p.msrc) {                                        // shallowwater_eq.py:810
        double sx[3], sy[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            sx[i] = swe_ld(swe_rsrc(p.msrc), k8, i*S8);
            sy[i] = swe_ld(swe_rsrc(p.msrc + 3*S), k8, i*S8);
        }
        const double ssx = sx[0] + sx[1] + sx[2], ssy = sy[0] + sy[1] + sy[2];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            bu[i] = fma(A12, ssx + sx[i], bu[i]);
            bv[i] = fma(A12, ssy + sy[i], bv[i]);
        }
    }
    if (p.vsrc) {                                        // shallowwater_eq.py:830
        double sv_[3];
#pragma unroll
        for (int i = 0; i < 3; i++) sv_[i] = swe_ld(swe_rsrc(p.vsrc), k8, i*S8);
        const double ss = sv_[0] + sv_[1] + sv_[2];
#pragma unroll
        for (int i = 0; i < 3; i++) be[i] = fma(A12, ss + sv_[i], be[i]);
    }
}

// Boundary facets (walls, open boundaries, boundary drag) are < 0.3 % of the work but their code needs ~60 registers of
// temporaries.  In the TRIANGLE kernel they are therefore not evaluated inside the facet loop (which treats them as zero)
// but in this epilogue, which runs for cells with a boundary facet after the rest of the cell update is finished - when
// the large working set of the main part is dead - reloads the few inputs it needs (L1/L2 hits) and adds the correction
// beta*dt*M^-1(boundary flux) to the output values that are still in registers.  The residual is linear in the facet
// contributions, so the result is the same up to summation order.  (The quadrilateral kernel runs at 2 waves/SIMD either
// way and keeps its boundary facets inline: the epilogue costs 14 % there.)
// 2A of a triangle for the boundary correction, with the contraction spelled out: the two variants of the stage kernel
// (swe_boundary_epilogue / BINL) see these operands in different forms (reloaded / shared with the facet normals) and, left to
// the compiler, fuse a different one of the two products - a one-ulp difference in sfac
__device__ __forceinline__ double swe_cross2a(double x0, double y0, double x1, double y1, double x2, double y2)
{
    return fma(x1 - x0, y2 - y0, -((y1 - y0)*(x2 - x0)));
}

template <bool NONLIN, bool LF, bool WD>
__device__ __forceinline__ void swe_boundary_epilogue(const SweStageArgs &p, int k, int nb0, int nb1, int nb2,
                                                      double ou[3], double ov[3], double oe[3])
{
#pragma clang fp contract(off)
    const size_t S = p.stride;
    double sfac;                                         // 6 dt beta / (2A)
    {
        const int v0 = p.cv[k], v1 = p.cv[S + k], v2 = p.cv[2*S + k];
        const double x0 = p.vx[v0], y0 = p.vy[v0];
        sfac = 6.0*p.dt*p.beta*swe_rcp(swe_cross2a(x0, y0, p.vx[v1], p.vy[v1], p.vx[v2], p.vy[v2]));
    }
    // one facet at a time, everything addressed in memory by plane index: no dynamically indexed register arrays
#pragma unroll 1
    for (int f = 0; f < 3; f++) {
        const int nbf = (f == 0) ? nb0 : (f == 1) ? nb1 : nb2;
        if (nbf >= 0) continue;
        const int a = f, b = (f == 2) ? 0 : f + 1;
        const int va = p.cv[(size_t)a*S + k], vb = p.cv[(size_t)b*S + k];
        const double xa_ = p.vx[va], ya_ = p.vy[va], xb_ = p.vx[vb], yb_ = p.vy[vb];
        const double ha = p.vh[va], hb = p.vh[vb];
        const double ala = WD ? p.valpha[va] : 0.0, alb = WD ? p.valpha[vb] : 0.0;
        const double ua = p.uin[(size_t)a*S + k], ub = p.uin[(size_t)b*S + k];
        const double va_ = p.uin[(size_t)(3 + a)*S + k], vb_ = p.uin[(size_t)(3 + b)*S + k];
        const double da_ = p.uin[(size_t)(6 + a)*S + k], db_ = p.uin[(size_t)(6 + b)*S + k];      // eta, or D with wetting-drying
        const double ea = WD ? swe_wd_eta(da_, ha, ala) : da_, eb = WD ? swe_wd_eta(db_, hb, alb) : db_;
        const double Ha = WD ? da_ : (NONLIN ? ha + ea : ha);
        const double Hb = WD ? db_ : (NONLIN ? hb + eb : hb);
        const double nxs = yb_ - ya_, nys = xa_ - xb_;
        double L, rL;
        swe_sqrt_rsqrt(swe_dot2(nxs, nxs, nys, nys), L, rL);
        double Fau = 0.0, Fbu = 0.0, Fav = 0.0, Fbv = 0.0, Fae = 0.0, Fbe = 0.0;
        swe_boundary_facet<NONLIN, LF, WD>(p, -nbf, k, a, b, ua, ub, va_, vb_, ea, eb, ha, hb, Ha, Hb, ala, alb, nxs, nys,
                                           L, rL, Fau, Fbu, Fav, Fbv, Fae, Fbe);
        const double dau = -0.5*Fau, dbu = -0.5*Fbu, dav = -0.5*Fav, dbv = -0.5*Fbv, dae = -0.5*Fae, dbe = -0.5*Fbe;
        // M^-1 of a vector that is non-zero at nodes a and b only (4 d_i - (d_a + d_b)), static output indices
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const double wa = (i == a) ? 3.0 : -1.0, wb = (i == b) ? 3.0 : -1.0;
            // explicit fma: the inline-boundary variant of the stage kernel repeats these lines with compile-time wa, wb and
            // must round identically (left to the compiler, constant weights contract differently from run-time ones)
            ou[i] = fma(sfac, fma(wa, dau, wb*dbu), ou[i]);
            ov[i] = fma(sfac, fma(wa, dav, wb*dbv), ov[i]);
            oe[i] = fma(sfac, fma(wa, dae, wb*dbe), oe[i]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// HorizontalViscosityTerm (SIPG, thetis/shallowwater_eq.py:554-616) of a triangle WITHOUT its boundary-facet terms: the
// cell integral, the optional grad-depth term and the interior facets, added to the assembled momentum residuals bu, bv.
// Same arithmetic as swe_sipg_kernel<2> (swe2d_sipg.h), but inside the stage kernel: the own state, the facet traces of
// the neighbours, the geometry and the mass inverse are already there; extra per facet are the neighbour's third node
// (2 gathers) and third vertex (id from the opp4 table, 2 gathers).  Boundary facets: swe_sipg_kernel<2, true> launch.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void swe_visc_interior(const SweStageArgs &p, int k, unsigned S8, swe_rsrc_t gu, swe_rsrc_t gv,
                                                  const int nb[3], const int vid[3], const double u[3], const double v[3],
                                                  const double una[3], const double unb[3], const double vna[3],
                                                  const double vnb[3], const double px[3], const double py[3],
                                                  const double nx[3], const double ny[3], const double Lf[3],
                                                  const double rLf[3], double twoA, const double Hn[3], double bu[3],
                                                  double bv[3])
{
    const bool gd = p.visc_grad_div != 0;
    const int4 o4 = p.opp4[k];
    const swe_rsrc_t rvx = swe_rsrc(p.vx), rvy = swe_rsrc(p.vy);
    // every extra load up front
    double uo[3], vo_[3], xo[3], yo[3], mu[3];
#pragma unroll
    for (int f = 0; f < 3; f++) {
        const int nbf = nb[f];
        const int kn = nbf >= 0 ? (nbf >> 2) : k;
        const int f2 = nbf >= 0 ? (nbf & 3) : f;
        const unsigned oo = (unsigned)kn*8u + (f2 == 0 ? 2u*S8 : (f2 == 1 ? 0u : S8));      // node (f2 + 2) % 3
        uo[f] = swe_ld(gu, oo, 0);
        vo_[f] = swe_ld(gv, oo, 0);
        const unsigned v8 = (unsigned)(f == 0 ? o4.x : (f == 1 ? o4.y : o4.z))*8u;
        xo[f] = swe_ld(rvx, v8, 0);
        yo[f] = swe_ld(rvy, v8, 0);
        mu[f] = p.nu_v ? swe_ld(swe_rsrc(p.nu_v), (unsigned)vid[f]*8u, 0) : p.nu_const;
    }
    const double A = 0.5*twoA, r2A = swe_rcp(twoA);
    double gx[3], gy[3];                              // grad(phi_i) = -nF_{i+1}/(2A)
#pragma unroll
    for (int i = 0; i < 3; i++) {
        gx[i] = -nx[(i + 1) % 3]*r2A;
        gy[i] = -ny[(i + 1) % 3]*r2A;
    }
    double G[2][2], S0[2][2];
    G[0][0] = u[0]*gx[0] + u[1]*gx[1] + u[2]*gx[2];
    G[0][1] = u[0]*gy[0] + u[1]*gy[1] + u[2]*gy[2];
    G[1][0] = v[0]*gx[0] + v[1]*gx[1] + v[2]*gx[2];
    G[1][1] = v[0]*gy[0] + v[1]*gy[1] + v[2]*gy[2];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int j = 0; j < 2; j++) S0[r][j] = G[r][j] + (gd ? G[j][r] : 0.0);
    double b[2][3];
    {
        const double am = A*(mu[0] + mu[1] + mu[2])*(1.0/3.0);
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int i = 0; i < 3; i++) b[r][i] = -am*(gx[i]*S0[r][0] + gy[i]*S0[r][1]);
    }
    if (p.visc_grad_depth) {                          // -dot(test, dot(grad(H)/H, stress))*dx          :611-612
        const double gHx = Hn[0]*gx[0] + Hn[1]*gx[1] + Hn[2]*gx[2], gHy = Hn[0]*gy[0] + Hn[1]*gy[1] + Hn[2]*gy[2];
        double t[2];
#pragma unroll
        for (int r = 0; r < 2; r++) t[r] = gHx*S0[0][r] + gHy*S0[1][r];
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
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int i = 0; i < 3; i++) b[r][i] += fac*l[i]*t[r];
        }
    }
#pragma unroll
    for (int f = 0; f < 3; f++) {
        if (nb[f] < 0) continue;
        const int a = f, bb = (f + 1) % 3;
        const double nxs = nx[f], nys = ny[f];
        const double L = Lf[f], rL = rLf[f];                       // facet lengths of the hyperbolic facet loop
        const double n0 = nxs*rL, n1 = nys*rL;
        const double w = 0.5*L;
        const double e1x = px[bb] - px[a], e1y = py[bb] - py[a];
        const double e2x = xo[f] - px[a], e2y = yo[f] - py[a];
        const double det = e1x*e2y - e1y*e2x;                      // -2 A_n
        const double rdet = swe_rcp(det);
        const double An = 0.5*fabs(det);
        const double ca[2] = {una[f], vna[f]}, cb[2] = {unb[f], vnb[f]}, co[2] = {uo[f], vo_[f]};
        const double c[2][3] = {{u[0], u[1], u[2]}, {v[0], v[1], v[2]}};
        double S0n[2][2], Gn[2][2];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const double d1 = cb[r] - ca[r], d2 = co[r] - ca[r];
            Gn[r][0] = (d1*e2y - d2*e1y)*rdet;
            Gn[r][1] = (d2*e1x - d1*e2x)*rdet;
        }
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int j = 0; j < 2; j++) S0n[r][j] = Gn[r][j] + (gd ? Gn[j][r] : 0.0);
        const double sigma = p.visc_sipg*L*swe_rcp(fmin(A, An));
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const double xb = q ? SWE_XI1 : SWE_XI0, xa = 1.0 - xb;
            const double muq = xa*mu[a] + xb*mu[bb];
            double jmp[2];
#pragma unroll
            for (int r = 0; r < 2; r++) jmp[r] = (xa*c[r][a] + xb*c[r][bb]) - (xa*ca[r] + xb*cb[r]);
            const double nn[2] = {n0, n1};
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const double sj0 = muq*(jmp[r]*n0 + (gd ? jmp[0]*nn[r] : 0.0));
                const double sj1 = muq*(jmp[r]*n1 + (gd ? jmp[1]*nn[r] : 0.0));
                const double sjn = sj0*n0 + sj1*n1;
                const double avn = 0.5*muq*((S0[r][0] + S0n[r][0])*n0 + (S0[r][1] + S0n[r][1])*n1);
                const double val = sigma*sjn - avn;
                b[r][a] -= w*xa*val;
                b[r][bb] -= w*xb*val;
#pragma unroll
                for (int i = 0; i < 3; i++) b[r][i] += w*0.5*(gx[i]*sj0 + gy[i]*sj1);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        bu[i] += b[0][i];
        bv[i] += b[1][i];
    }
}

// Register budget: see the -Rpass-analysis output quoted in DESIGN.md; forcing more waves per SIMD than the allocation
// gives naturally spills (20 B/lane at 128 VGPRs for the first stage: +16 MB scratch writes per launch, not faster;
// 5-6 waves: 2-3x slower).
#ifdef SWE_WAVE_TIMING
// profiling build only (tools/wavetiming.py): per one-wave workgroup the 100 MHz wall clock at five points of the stage kernel
#define SWE_WT_MAX 8192
__device__ unsigned long long swe_wave_ts[6][SWE_WT_MAX];
#define SWE_WT(i) do { if (threadIdx.x == 0 && blockIdx.x < SWE_WT_MAX) swe_wave_ts[i][blockIdx.x] = wall_clock64(); } while (0)
#define SWE_WT_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define SWE_WT(i)
#define SWE_WT_DRAIN()
#endif

// BINL: boundary fluxes are evaluated after the cell's outputs are finished from the values the lane still holds (one pass per
// lane over ITS boundary facet), instead of reloading the facet's nodes in swe_boundary_epilogue: no dependent memory round
// trips in the waves that own boundary cells (a quarter of the waves of a 125 k-cell partition, and the last to finish).
// 164-166 VGPRs, no scratch, 3 waves/SIMD (the boundary markers travel packed in one register, h + eta is re-formed).
// Same bits as the epilogue path (tests/test_gpu_parity.py).
template <bool NONLIN, bool LF, bool HASU0, bool SRC, bool WD, bool VISC = false, bool BINL = false, bool LDSX = false>
__global__ __launch_bounds__(SWE_BLOCK, SWE_MIN_WAVES) void swe_stage_kernel(const SweStageArgs p)
{
    // No implicit contraction in this kernel: which of two products the compiler fuses depends on the code around it, and the
    // variants of this template (boundary handling, LDS exchange, sources ...) must give the same bits - a partition takes a
    // different variant than the whole mesh.  Every fused multiply-add below is written out.
#pragma clang fp contract(off)
    SWE_WT(0);
#ifdef SWE_WAVE_TIMING
    if (threadIdx.x == 0 && blockIdx.x < SWE_WT_MAX) {      // where does the wave run?  HW_ID (wave, simd, cu, sh, se) | XCC_ID << 32
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        swe_wave_ts[5][blockIdx.x] = (unsigned long long)hw | ((unsigned long long)(xcc & 0xf) << 32);
    }
#endif
#ifdef SWE_NO_XCD_MAP
    int lb = blockIdx.x;
#else
    int lb = swe_logical_block(blockIdx.x, gridDim.x);
#endif
    if (p.reverse) {              // the blocks dispatched first take the END of the range: what the previous launch touched last
        lb = (p.cell_end - p.cell_begin + SWE_BLOCK - 1)/SWE_BLOCK - 1 - lb;
        if (lb < 0) return;
    }
    const int k = p.cell_begin + lb*SWE_BLOCK + (int)threadIdx.x;
    if (k >= p.cell_end) return;
    const size_t S = p.stride;
    const double g = p.g;
    const unsigned S8 = (unsigned)S*8u, k8 = (unsigned)k*8u;
    const swe_rsrc_t gu = swe_rsrc(p.uin), gv = swe_rsrc(p.uin + 3*S), ge = swe_rsrc(p.uin + 6*S);

    // ---- every load of the cell is issued here, unconditionally, so that a wave pays two memory round trips
    //      (own data + indices, then the gathers) instead of one per facet and one per output plane.
    double u[3], v[3], e[3];
    int nb[3], vid[3];
    swe_conn_load(p.idxc, p.idx4, p.idx2, k, nb, vid);
    // boundary markers of the three facets in one register (0: interior facet): all the boundary pass of the BINL variant
    // needs of nb[] at the end of the kernel
    const int bmarkers = (nb[0] < 0 ? -nb[0] : 0) | (nb[1] < 0 ? (-nb[1]) << 8 : 0) | (nb[2] < 0 ? (-nb[2]) << 16 : 0);
    // ... and the table entry of the lane's first boundary facet, requested now with all the other loads: read inside the boundary
    // pass it was a dependent trip to memory in every boundary wave (+0.9 us of "loads" in their wave timing)
    int bkind1 = 0;
    if (BINL && bmarkers != 0) {
        const int m1 = (bmarkers & 0xff) ? (bmarkers & 0xff) : ((bmarkers & 0xff00) ? ((bmarkers >> 8) & 0xff) : (bmarkers >> 16));
        bkind1 = m1 < SWE_MAX_MARKERS ? p.bc.kind[m1] : 0;
    }
#ifdef SWE_WAVE_TIMING
    if (nb[0] == 0x7fffffff) return;          // forces the index loads to land before the time stamp
    SWE_WT(1);
#endif
#pragma unroll
    for (int i = 0; i < 3; i++) {
        u[i] = swe_ld(gu, k8, i*S8);
        v[i] = swe_ld(gv, k8, i*S8);
        e[i] = swe_ld(ge, k8, i*S8);
    }
    // w = a0*U0 + a1*U_in, the part of the Shu-Osher combine that does not depend on the tendency
    double wu[3], wv[3], we[3];
    swe_rsrc_t g0u, g0v, g0e;
    if (HASU0) { g0u = swe_rsrc(p.u0); g0v = swe_rsrc(p.u0 + 3*S); g0e = swe_rsrc(p.u0 + 6*S); }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        wu[i] = p.a1*u[i];
        wv[i] = p.a1*v[i];
        we[i] = p.a1*e[i];
        if (HASU0) {
            wu[i] = fma(p.a0, swe_ld(g0u, k8, i*S8), wu[i]);
            wv[i] = fma(p.a0, swe_ld(g0v, k8, i*S8), wv[i]);
            if (!WD) we[i] = fma(p.a0, swe_ld(g0e, k8, i*S8), we[i]);
        }
    }
    double e0[3] = {0.0, 0.0, 0.0};
    if (WD && HASU0) {
#pragma unroll
        for (int i = 0; i < 3; i++) e0[i] = swe_ld(g0e, k8, i*S8);
    }
    // neighbour traces: the neighbour traverses the shared facet backwards, its node (f2+1)%3 sits on my node f and
    // its node f2 on my node f+1.  Boundary facets read this cell itself (value unused) to keep the loads branch-free.
    double una[3], unb[3], vna[3], vnb[3], ena[3], enb[3];
    if constexpr (LDSX) {
    // LDSX (large launches): neighbour traces of cells inside this wave's own 64-cell block (~85 % with the tile-Hilbert numbering) come from LDS: the
    // wave publishes its nine nodal values there (one-wave workgroup: the barrier is local to the wave), only the others are
    // gathered from memory, issued before the exchange.  Same values either way.  Measured with the device's tile-Hilbert
    // numbering (us/step, same box, without / with): 250 k cells 37.9 / 42.2, 500 k 63.5 / 68.3, 1 M 115-118 / 118-119,
    // 2 M 295-312 / 297-298, 4 M 573-583 / 544 (-6 %): it pays where the state no longer fits the Infinity Cache and the
    // gathers compete with the streaming loads for HBM-side bandwidth; below that the extra LDS round trip costs more than it
    // saves, so the host picks the variant by launch size.  (With a row-major cell numbering, where a third of the traces
    // come from the next row, the gain is larger and starts at 1 M cells: -5 % there, -14 % at 4 M.)
    __shared__ double xs[9][SWE_BLOCK];
    const int lane = (int)threadIdx.x, kw0 = k - lane, kw1 = min(kw0 + SWE_BLOCK, p.cell_end);
    int lnb[3], nodeb[3];
    bool inw[3];
#pragma unroll
    for (int f = 0; f < 3; f++) {
        const int nbf = nb[f];
        const int kn = nbf >= 0 ? (nbf >> 2) : k;
        const int f2 = nbf >= 0 ? (nbf & 3) : f;
        inw[f] = kn >= kw0 && kn < kw1;
        lnb[f] = kn - kw0; nodeb[f] = f2;
        if (!inw[f]) {
            const unsigned kn8 = (unsigned)kn*8u;
            const unsigned ob = kn8 + (f2 == 0 ? 0u : (f2 == 1 ? S8 : 2u*S8));
            const unsigned oa = kn8 + (f2 == 0 ? S8 : (f2 == 1 ? 2u*S8 : 0u));
            una[f] = swe_ld(gu, oa, 0);
            unb[f] = swe_ld(gu, ob, 0);
            vna[f] = swe_ld(gv, oa, 0);
            vnb[f] = swe_ld(gv, ob, 0);
            ena[f] = swe_ld(ge, oa, 0);
            enb[f] = swe_ld(ge, ob, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 3; i++) { xs[i][lane] = u[i]; xs[3 + i][lane] = v[i]; xs[6 + i][lane] = e[i]; }
    __syncthreads();
#pragma unroll
    for (int f = 0; f < 3; f++) {
        if (inw[f]) {
            const int nb_ = nodeb[f], na_ = nb_ == 2 ? 0 : nb_ + 1;
            una[f] = xs[na_][lnb[f]];     unb[f] = xs[nb_][lnb[f]];
            vna[f] = xs[3 + na_][lnb[f]]; vnb[f] = xs[3 + nb_][lnb[f]];
            ena[f] = xs[6 + na_][lnb[f]]; enb[f] = xs[6 + nb_][lnb[f]];
        }
    }
    } else {
#pragma unroll
    for (int f = 0; f < 3; f++) {
        const int nbf = nb[f];
        const int kn = nbf >= 0 ? (nbf >> 2) : k;
        const int f2 = nbf >= 0 ? (nbf & 3) : f;
        const unsigned kn8 = (unsigned)kn*8u;
        const unsigned ob = kn8 + (f2 == 0 ? 0u : (f2 == 1 ? S8 : 2u*S8));         // node f2
        const unsigned oa = kn8 + (f2 == 0 ? S8 : (f2 == 1 ? 2u*S8 : 0u));         // node (f2 + 1) % 3
        una[f] = swe_ld(gu, oa, 0);
        unb[f] = swe_ld(gu, ob, 0);
        vna[f] = swe_ld(gv, oa, 0);
        vnb[f] = swe_ld(gv, ob, 0);
        ena[f] = swe_ld(ge, oa, 0);
        enb[f] = swe_ld(ge, ob, 0);
    }
    }
    double px[3], py[3], h[3], H[3], al[3] = {0.0, 0.0, 0.0};
    const swe_rsrc_t rvx = swe_rsrc(p.vx), rvy = swe_rsrc(p.vy), rvh = swe_rsrc(p.vh);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const unsigned v8 = (unsigned)vid[i]*8u;
        px[i] = swe_ld(rvx, v8, 0);
        py[i] = swe_ld(rvy, v8, 0);
        h[i] = swe_ld(rvh, v8, 0);
        if (WD) al[i] = swe_ld(swe_rsrc(p.valpha), v8, 0);
        if (WD) {           // the planes hold D (U(0)'s too); the continuity equation advances zeta = D - h
            H[i] = e[i];
            we[i] = p.a1*(H[i] - h[i]);
            if (HASU0) we[i] = fma(p.a0, e0[i] - h[i], we[i]);
            e[i] = swe_wd_eta(H[i], h[i], al[i]);
        } else H[i] = NONLIN ? h[i] + e[i] : h[i];
    }
    SWE_WT_DRAIN();
    SWE_WT(2);
    // scaled outward normals nF_f = |F| n of facet f (vertex f -> f+1); counter-clockwise cell
    double nx[3], ny[3];
#pragma unroll
    for (int f = 0; f < 3; f++) {
        const int b = (f + 1) % 3;
        nx[f] = py[b] - py[f];
        ny[f] = px[f] - px[b];
    }
    const double twoA = fma(nx[0], ny[1], -(ny[0]*nx[1]));
    // A*grad(phi_i) = -nF_{i+1}/2
    double gxs[3], gys[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        gxs[i] = -0.5*nx[(i + 1) % 3];
        gys[i] = -0.5*ny[(i + 1) % 3];
    }

    // ---- cell integrals (closed form)
    double bu[3], bv[3], be[3];
    {
        const double ge3 = g*(e[0] + e[1] + e[2])*(1.0/3.0);
        const double SHu = swe_int2(H, u)*(1.0/12.0), SHv = swe_int2(H, v)*(1.0/12.0);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            bu[i] = gxs[i]*ge3;                              // +g eta div(psi)       shallowwater_eq.py:361
            bv[i] = gys[i]*ge3;
            be[i] = swe_dot2(gxs[i], SHu, gys[i], SHv);      // +grad(phi).(H u)      shallowwater_eq.py:422
        }
        if (NONLIN) {                                        // +(psi div u + u.grad psi).u  shallowwater_eq.py:478
            const double Suu = swe_int2(u, u)*(1.0/12.0), Suv = swe_int2(u, v)*(1.0/12.0),
                         Svv = swe_int2(v, v)*(1.0/12.0);
            const double D12 = fma(gys[2], v[2], fma(gys[1], v[1], fma(gys[0], v[0],
                               fma(gxs[2], u[2], fma(gxs[1], u[1], gxs[0]*u[0])))))*(1.0/12.0);
            const double us = u[0] + u[1] + u[2], vs = v[0] + v[1] + v[2];
#pragma unroll
            for (int i = 0; i < 3; i++) {
                bu[i] = fma(gys[i], Suv, fma(gxs[i], Suu, fma(D12, us + u[i], bu[i])));
                bv[i] = fma(gys[i], Svv, fma(gxs[i], Suv, fma(D12, vs + v[i], bv[i])));
            }
        }
    }

    // ---- facet integrals: 2-point Gauss-Legendre, numerical fluxes seen from this cell
    double Lf[3] = {0.0, 0.0, 0.0}, rLf[3] = {0.0, 0.0, 0.0};     // kept for the fused viscosity only
#pragma unroll
    for (int f = 0; f < 3; f++) {
        const int a = f, b = (f + 1) % 3;
        const double nxs = nx[f], nys = ny[f];
        const double len2 = swe_dot2(nxs, nxs, nys, nys);
        double L, rL;
        swe_sqrt_rsqrt(len2, L, rL);
        if (VISC) { Lf[f] = L; rLf[f] = rL; }
        double Fau = 0.0, Fbu = 0.0, Fav = 0.0, Fbv = 0.0, Fae = 0.0, Fbe = 0.0;
        // Branch-free: a boundary facet carries this cell's own values as "neighbour" traces (finite numbers), its flux is
        // evaluated like any other and discarded below.  With `if (nb[f] >= 0)` around this block the compiler sank the six
        // trace loads of a facet into the branch - a third dependent trip to memory in every wave.  (The fused-viscosity variant
        // keeps the branch: its traces stay live for swe_visc_interior and the branch-free form needs 256 VGPRs = 1 wave/SIMD,
        // 248 instead of 185-198 us/step at 1 M cells.)
        if (!(VISC || WD) || nb[f] >= 0) {
            // neighbour's nodal depth on this facet: what its planes hold; its elevation by the closed form (bathymetry and alpha
            // are continuous: same vertices)
            const double Dna = WD ? ena[f] : 0.0, Dnb = WD ? enb[f] : 0.0;
            const double ena_ = WD ? swe_wd_eta(Dna, h[a], al[a]) : ena[f], enb_ = WD ? swe_wd_eta(Dnb, h[b], al[b]) : enb[f];
            swe_facet_flux<NONLIN, LF, WD>(g, p.sigma_lf, u[a], u[b], v[a], v[b], e[a], e[b], h[a], h[b], H[a], H[b], una[f], unb[f],
                                           vna[f], vnb[f], ena_, enb_, Dna, Dnb, nxs, nys, L, rL, Fau, Fbu, Fav, Fbv, Fae, Fbe);
        }
        if (nb[f] < 0) { Fau = 0.0; Fbu = 0.0; Fav = 0.0; Fbv = 0.0; Fae = 0.0; Fbe = 0.0; }   // see swe_boundary_epilogue / BINL
        bu[a] = fma(-0.5, Fau, bu[a]); bu[b] = fma(-0.5, Fbu, bu[b]);
        bv[a] = fma(-0.5, Fav, bv[a]); bv[b] = fma(-0.5, Fbv, bv[b]);
        be[a] = fma(-0.5, Fae, be[a]); be[b] = fma(-0.5, Fbe, be[b]);
    }

    // optional cell-local terms AFTER the facet loop: the 18 neighbour traces are dead by now, which keeps the SRC variants
    // at 146 (162 with wetting-drying) VGPRs = 3 waves/SIMD instead of 188 (194) = 2
    if (SRC) swe_source_terms(p, k, S, twoA, u, v, H, gxs, gys, bu, bv, be);
    if (VISC) swe_visc_interior(p, k, S8, gu, gv, nb, vid, u, v, una, unb, vna, vnb, px, py, nx, ny, Lf, rLf, twoA, H, bu, bv);

    // ---- mass inverse (M^-1 b)_i = 3/A (4 b_i - sum b), times dt, and the Shu-Osher combine
    const double s = 6.0*p.dt*p.beta*swe_rcp(twoA);
    const double su = bu[0] + bu[1] + bu[2], sv = bv[0] + bv[1] + bv[2], se = be[0] + be[1] + be[2];
    double ou[3], ov[3], oe[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        ou[i] = fma(s, fma(4.0, bu[i], -su), wu[i]);
        ov[i] = fma(s, fma(4.0, bv[i], -sv), wv[i]);
        oe[i] = fma(s, fma(4.0, be[i], -se), we[i]);          // eta, or zeta = D - h with wetting-drying
    }
    // boundary facets were skipped above; their correction is added to the finished outputs (see swe_boundary_epilogue)
    if (BINL && bmarkers != 0) {
        // Boundary facets from the values this lane still holds (no second trip to memory: on a small grid the dependent
        // reloads of swe_boundary_epilogue made the boundary waves the last ones to finish).  Every lane works on ITS
        // boundary facet - the wave runs the boundary code once (twice where a corner cell is among its lanes), not once per
        // facet slot - in ascending facet order and with the arithmetic of swe_boundary_epilogue: the same bits.
#pragma clang fp contract(off)
        // 2A exactly as swe_boundary_epilogue forms it, from the facet normals that are still live: x1 - x0 = -ny[0],
        // y2 - y0 = -nx[2], y1 - y0 = nx[0], x2 - x0 = ny[2] (negation is exact)
        const double sfac = 6.0*p.dt*p.beta*swe_rcp(fma(-ny[0], -nx[2], -(nx[0]*ny[2])));
        int rem = ((bmarkers & 0xff) ? 1 : 0) | ((bmarkers & 0xff00) ? 2 : 0) | ((bmarkers & 0xff0000) ? 4 : 0);
        int kind_next = bkind1;                     // first pass: prefetched; a second pass (corner cell) reads the table
#define SWE_SEL3(x, i) ((i) == 0 ? (x)[0] : ((i) == 1 ? (x)[1] : (x)[2]))
#pragma unroll 1
        while (rem) {
            const int f = (rem & 1) ? 0 : ((rem & 2) ? 1 : 2);
            rem &= rem - 1;
            const int a = f, b = (f == 2) ? 0 : f + 1;
            const double nxs = SWE_SEL3(nx, f), nys = SWE_SEL3(ny, f);
            double L, rL;
            swe_sqrt_rsqrt(swe_dot2(nxs, nxs, nys, nys), L, rL);
            double Fau = 0.0, Fbu = 0.0, Fav = 0.0, Fbv = 0.0, Fae = 0.0, Fbe = 0.0;
            // (nodal depth: h + eta re-formed here for the plain nonlinear case instead of keeping H[] live - the same sum)
            const double Ha_ = (WD || !NONLIN) ? SWE_SEL3(H, a) : SWE_SEL3(h, a) + SWE_SEL3(e, a);
            const double Hb_ = (WD || !NONLIN) ? SWE_SEL3(H, b) : SWE_SEL3(h, b) + SWE_SEL3(e, b);
            swe_boundary_facet<NONLIN, LF, WD>(p, (bmarkers >> (8*f)) & 0xff, k, a, b, SWE_SEL3(u, a), SWE_SEL3(u, b), SWE_SEL3(v, a),
                                               SWE_SEL3(v, b), SWE_SEL3(e, a), SWE_SEL3(e, b), SWE_SEL3(h, a), SWE_SEL3(h, b),
                                               Ha_, Hb_, SWE_SEL3(al, a), SWE_SEL3(al, b), nxs, nys, L, rL,
                                               Fau, Fbu, Fav, Fbv, Fae, Fbe, kind_next);
            kind_next = -1;
            const double dau = -0.5*Fau, dbu = -0.5*Fbu, dav = -0.5*Fav, dbv = -0.5*Fbv, dae = -0.5*Fae, dbe = -0.5*Fbe;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const double wa = (i == a) ? 3.0 : -1.0, wb = (i == b) ? 3.0 : -1.0;
                ou[i] = fma(sfac, fma(wa, dau, wb*dbu), ou[i]);      // as in swe_boundary_epilogue, bit for bit
                ov[i] = fma(sfac, fma(wa, dav, wb*dbv), ov[i]);
                oe[i] = fma(sfac, fma(wa, dae, wb*dbe), oe[i]);
            }
        }
#undef SWE_SEL3
    }
    if (!BINL && (nb[0] | nb[1] | nb[2]) < 0) swe_boundary_epilogue<NONLIN, LF, WD>(p, k, nb[0], nb[1], nb[2], ou, ov, oe);
    const swe_rsrc_t gou = swe_rsrc(p.uout), gov = swe_rsrc(p.uout + 3*S), goe = swe_rsrc(p.uout + 6*S);
#ifdef SWE_WAVE_TIMING
    if (ou[0] == 1.2345e300) return;          // the arithmetic has to be finished before the time stamp
    SWE_WT(3);
#endif
    // zeta = D - h -> limited depth D (stored); dry-ground relaxation.  (Not for the parity hook swe2d_tendency, a0 = a1 = 0: it
    // returns the raw tendencies of (u, v, zeta).)
    if (WD && !(p.a0 == 0.0 && p.a1 == 0.0)) swe_wd_finish<3>(g, p.beta*p.dt, h, al, ou, ov, oe, !p.wd_skip_relax);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        swe_st(gou, k8, i*S8, ou[i]);
        swe_st(gov, k8, i*S8, ov[i]);
        swe_st(goe, k8, i*S8, oe[i]);
    }
    SWE_WT_DRAIN();
    SWE_WT(4);
}

// ---------------------------------------------------------------------------------------------------------------
// layout conversion: host (Firedrake-like) AoS  uv[kN][2], eta[kN]  <->  3k SoA planes (k = nodes per cell)
static __global__ void swe_aos_to_planes(const double *uv, const double *eta, double *planes, size_t stride, int n, int npc)
{
    const int k = blockIdx.x*blockDim.x + threadIdx.x;
    if (k >= n) return;
    for (int i = 0; i < npc; i++) {
        planes[(size_t)i*stride + k] = uv[2*((size_t)npc*k + i)];
        planes[(size_t)(npc + i)*stride + k] = uv[2*((size_t)npc*k + i) + 1];
        planes[(size_t)(2*npc + i)*stride + k] = eta[(size_t)npc*k + i];
    }
}

// valpha != nullptr: wetting-drying, the elevation planes hold the displaced depth D (swe_wd_eta): the host gets
// eta = D - alpha^2/(4 D) - h (IEEE division: this is the boundary, not the hot path)
static __global__ void swe_planes_to_aos(const double *planes, double *uv, double *eta, size_t stride, int n, int npc,
                                         const int *cv = nullptr, const double *vh = nullptr, const double *valpha = nullptr)
{
    const int k = blockIdx.x*blockDim.x + threadIdx.x;
    if (k >= n) return;
    for (int i = 0; i < npc; i++) {
        uv[2*((size_t)npc*k + i)] = planes[(size_t)i*stride + k];
        uv[2*((size_t)npc*k + i) + 1] = planes[(size_t)(npc + i)*stride + k];
        double e = planes[(size_t)(2*npc + i)*stride + k];
        if (valpha) {
            const int v = cv[(size_t)i*stride + k];
            e = e - 0.25*valpha[v]*valpha[v]/e - vh[v];
        }
        eta[(size_t)npc*k + i] = e;
    }
}
// the inverse of the above for buffers whose elevation planes hold D although wetting-drying is about to be switched off or its
// alpha changed (swe2d_set_wetting_and_drying with a resident state): D -> eta in place
static __global__ void swe_wd_planes_to_eta(double *planes, size_t stride, int n, int npc, const int *cv, const double *vh, const double *valpha)
{
    const int k = blockIdx.x*blockDim.x + threadIdx.x;
    if (k >= n) return;
    for (int i = 0; i < npc; i++) {
        const int v = cv[(size_t)i*stride + k];
        const double D = planes[(size_t)(2*npc + i)*stride + k];
        planes[(size_t)(2*npc + i)*stride + k] = D - 0.25*valpha[v]*valpha[v]/D - vh[v];
    }
}

// nodal scalar field (kN) -> k planes;  vector (kN,2) -> 2k planes (x0.. y0..)
static __global__ void swe_nodal_to_planes(const double *nodal, double *planes, size_t stride, int n, int ncomp, int npc)
{
    const int k = blockIdx.x*blockDim.x + threadIdx.x;
    if (k >= n) return;
    for (int i = 0; i < npc; i++)
        for (int c = 0; c < ncomp; c++)
            planes[(size_t)(npc*c + i)*stride + k] = nodal[(size_t)ncomp*((size_t)npc*k + i) + c];
}

// continuous P1 coefficient given per VERTEX -> nodal planes (the CG -> DG injection done on the device: a time-dependent
// wind / pressure field costs one value per vertex over PCIe instead of one per DG node plus a host-side gather)
static __global__ void swe_vertex_to_planes(const double *vert, double *planes, size_t stride, const int *cv, int n, int ncomp, int npc)
{
    const int k = blockIdx.x*blockDim.x + threadIdx.x;
    if (k >= n) return;
    for (int i = 0; i < npc; i++) {
        const size_t v = (size_t)cv[(size_t)i*stride + k];
        for (int c = 0; c < ncomp; c++) planes[(size_t)(npc*c + i)*stride + k] = vert[v*ncomp + c];
    }
}

// Function-valued boundary data of ONE marker: copy the two end-node values of every boundary facet carrying `marker` from a
// nodal field in host layout into the per-facet planes (plane 2f: node f, plane 2f+1: node f+1; component c: + 2*npc*c)
static __global__ void swe_bc_field_scatter(const double *nodal, double *planes, size_t stride, const int *nbr, int n, int ncomp,
                                     int npc, int marker)
{
    const int k = blockIdx.x*blockDim.x + threadIdx.x;
    if (k >= n) return;
    for (int f = 0; f < npc; f++) {
        if (nbr[(size_t)f*stride + k] != -marker) continue;
        const int b = (f + 1 == npc) ? 0 : f + 1;
        for (int c = 0; c < ncomp; c++) {
            planes[(size_t)(2*npc*c + 2*f)*stride + k] = nodal[(size_t)ncomp*((size_t)npc*k + f) + c];
            planes[(size_t)(2*npc*c + 2*f + 1)*stride + k] = nodal[(size_t)ncomp*((size_t)npc*k + b) + c];
        }
    }
}

// The same from a COMPACT list: entry t = boundary facet `facet[t]` of cell `cell[t]` with `nval` values per component
// (2 = the facet's end nodes -> planes 2f, 2f+1 of component c at + 2*npc*c; npc = all nodes of the cell -> planes npc*f + i).
// What a time-dependent boundary Function costs per update: a few KB over PCIe instead of the whole nodal field.
static __global__ void swe_bc_facet_scatter(const double *vals, double *planes, size_t stride, const int *cell, const int *facet, int n,
                                     int ncomp, int npc, int nval)
{
    const int t = blockIdx.x*blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int k = cell[t], f = facet[t];
    for (int c = 0; c < ncomp; c++)
        for (int j = 0; j < nval; j++) {
            const size_t plane = (nval == 2) ? (size_t)(2*npc*c + 2*f + j) : (size_t)(npc*f + j);
            planes[plane*stride + k] = vals[((size_t)t*nval + j)*ncomp + c];
        }
}

// Function-valued tracer boundary value of ONE marker: all npc nodal values of the cell are kept per boundary facet (the
// diffusive boundary term needs the cell gradient of the external value, tracer_eq_2d.py:270-276): plane npc*f + i
static __global__ void swe_bc_cellfield_scatter(const double *nodal, double *planes, size_t stride, const int *nbr, int n, int npc,
                                         int marker)
{
    const int k = blockIdx.x*blockDim.x + threadIdx.x;
    if (k >= n) return;
    for (int f = 0; f < npc; f++) {
        if (nbr[(size_t)f*stride + k] != -marker) continue;
        for (int i = 0; i < npc; i++) planes[(size_t)(npc*f + i)*stride + k] = nodal[(size_t)npc*k + i];
    }
}

// halo: message layout [n][np] (cell-major, np = 3k planes), so the per-peer segments of one buffer are contiguous
static __global__ void swe_halo_pack(const double *planes, size_t stride, const int *cells, int n, double *buf, int np)
{
    const int t = blockIdx.x*blockDim.x + threadIdx.x;
    if (t >= np*n) return;
    const int j = t/np, q = t - np*j;
    buf[t] = planes[(size_t)q*stride + cells[j]];
}

static __global__ void swe_halo_unpack(double *planes, size_t stride, const int *cells, int n, const double *buf, int np)
{
    const int t = blockIdx.x*blockDim.x + threadIdx.x;
    if (t >= np*n) return;
    const int j = t/np, q = t - np*j;
    planes[(size_t)q*stride + cells[j]] = buf[t];
}

// wetting-drying: the elevation planes as handed in by the caller (eta) become the device's D planes, brought to the admissible
// set of the explicit scheme (every nodal depth through the positivity limiter of swe_wd_finish; velocities untouched) - otherwise
// the first stage would do it and the volume of the initial state would not be the volume the run conserves
__device__ __forceinline__ void swe_quad_mean_weights(double d0, double d1, double d2, double w[4]);
// (vx != nullptr: general quadrilaterals, the limiter's cell mean is mass-weighted)
template <int K>
__global__ void swe_wd_clip_kernel(double *planes, size_t stride, const int *cv, const double *vh, const double *valpha, int n,
                                   const double *vx = nullptr, const double *vy = nullptr)
{
    const int k = blockIdx.x*blockDim.x + threadIdx.x;
    if (k >= n) return;
    double h[K], al[K], zu[K], zv[K], ze[K], px[K], py[K];
    for (int i = 0; i < K; i++) {
        const int v = cv[(size_t)i*stride + k];
        h[i] = vh[v]; al[i] = valpha[v];
        px[i] = vx ? vx[v] : 0.0; py[i] = vx ? vy[v] : 0.0;
        zu[i] = 0.0; zv[i] = 0.0;
        ze[i] = swe_wd_depth(h[i] + planes[(size_t)(2*K + i)*stride + k], al[i]) - h[i];
    }
    if (K == 4 && vx) {
        const double ax = px[1] - px[0], ay = py[1] - py[0], bx = px[K - 1] - px[0], by = py[K - 1] - py[0];
        const double cx = (px[0] - px[1]) + (px[2] - px[K - 1]), cy = (py[0] - py[1]) + (py[2] - py[K - 1]);
        double mw[4];
        swe_quad_mean_weights(ax*by - ay*bx, ax*cy - ay*cx, cx*by - cy*bx, mw);
        swe_wd_finish<K>(9.81, 0.0, h, al, zu, zv, ze, true, mw);
    } else swe_wd_finish<K>(9.81, 0.0, h, al, zu, zv, ze);
    for (int i = 0; i < K; i++) planes[(size_t)(2*K + i)*stride + k] = ze[i];
}

// Dry-ground relaxation of the velocity as a pass of its own (the last part of swe_wd_finish): viscous runs with wetting-drying
// apply it AFTER the viscosity pass has added its share of the stage update, so that the whole new velocity is relaxed -
// u <- u exp(-dt_stage/tau psi^2), tau = SWE_WD_TAU sqrt(alpha/g), psi = clamp(-H/alpha - 1, 0, 1) with the stage's new eta.
template <int K>
__global__ void swe_wd_relax_kernel(double *planes, size_t stride, const int *cv, const double *vh, const double *valpha, double g,
                                    double dt_stage, int c0, int c1)
{
    const int k = c0 + blockIdx.x*blockDim.x + threadIdx.x;
    if (k >= c1) return;
    for (int i = 0; i < K; i++) {
        const int vtx = cv[(size_t)i*stride + k];
        const double al = valpha[vtx], D = planes[(size_t)(2*K + i)*stride + k];          // the planes hold D
        const double ral = swe_rcp(al);
        const double psi = fmin(1.0, fmax(0.0, -(D - 0.25*al*al*swe_rcp(D))*ral - 1.0));
        if (psi > 0.0) {
            const double fac = exp(-dt_stage*(1.0/SWE_WD_TAU)*swe_sqrt(g*ral)*psi*psi);
            planes[(size_t)i*stride + k] *= fac;
            planes[(size_t)(K + i)*stride + k] *= fac;
        }
    }
}

// ---- order-independent sums for the diagnostics ---------------------------------------------------------------------------
// A floating-point sum depends on the order of its terms, i.e. on how a mesh is cut into blocks and partitions; the integrals
// print_state and the conservation callbacks show (solver2d.py:955-956, callback.py:323-328, all-reduced over the ranks in the
// reference, callback.py:478-482) would then differ in their last digits between a run on one GPU and the same run on eight.
// Every per-cell contribution x is therefore split EXACTLY into six signed 38-bit limbs of units 2^40, 2^2, 2^-36, 2^-74, 2^-112,
// 2^-150 (|x| < 2^78 ~ 3e23; what lies below 2^-150 ~ 7e-46 is truncated, per term, in the same way wherever the term is computed)
// and the limbs are added as 64-bit integers - associative, so lanes, blocks and ranks may add them in any order (2^25 terms fit).
// (Rounds 4 had four limbs, i.e. a floor of 2^-74 ~ 5e-23 per term: integrals of small fields on small cells - a lake-at-rest
// residual of 1e-14 on a unit-square mesh, eta^2 A ~ 1e-32 - lost most of their digits or came out as exactly 0 where the reference's
// floating-point all-reduce keeps them.  228 bits below 2^78 leave nothing of a double behind that a sum of doubles
// of one magnitude would keep.)
// swe2d_sum_limbs_to_double (csrc/swe2d_api.hip) rounds the total to the nearest double, once.
#define SWE_SUM_LIMBS 6
#define SWE_DIAG_ACC (3*SWE_SUM_LIMBS + 1)             // limb sums of up to three integrals + the counter of unsummable terms
#define SWE_DIAG_BUCKETS 64                            // copies of the accumulators (block b adds to copy b % 64; the host adds the copies):
                                                       //  15 625 blocks x 12 atomics on ONE set of addresses took 0.9 ms, serialised in L2
__device__ __forceinline__ void swe_sum_split(double x, long long q[SWE_SUM_LIMBS], unsigned &bad)
{
    if (!(fabs(x) < 0x1p78)) { bad = 1u; x = 0.0; }                      // NaN, Inf, or out of range: reported, not summed
    double t = trunc(x*0x1p-40); q[0] = (long long)t; x -= t*0x1p40;      // every product and difference here is exact
    t = trunc(x*0x1p-2);  q[1] = (long long)t; x -= t*0x1p2;
    t = trunc(x*0x1p36);  q[2] = (long long)t; x -= t*0x1p-36;
    t = trunc(x*0x1p74);  q[3] = (long long)t; x -= t*0x1p-74;
    t = trunc(x*0x1p112); q[4] = (long long)t; x -= t*0x1p-112;
    t = trunc(x*0x1p150); q[5] = (long long)t;
}
// adds the block's (one wave's) sum of x to acc[0..3]; acc[4*n_sums] of the launch counts the terms that were not summed
__device__ __forceinline__ void swe_sum_accumulate(double x, unsigned long long *acc, unsigned long long *bad_counter)
{
    long long q[SWE_SUM_LIMBS];
    unsigned bad = 0u;
    swe_sum_split(x, q, bad);
    for (int j = 0; j < SWE_SUM_LIMBS; j++) {
        long long v = q[j];
        for (int off = SWE_BLOCK/2; off > 0; off >>= 1) v += __shfl_down(v, off, SWE_BLOCK);
        if (threadIdx.x == 0 && v != 0) atomicAdd(acc + j, (unsigned long long)v);
    }
    if (bad) atomicAdd(bad_counter, 1ull);
}
__device__ __forceinline__ double swe_wave_min(double v)
{
    for (int off = SWE_BLOCK/2; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, SWE_BLOCK));
    return v;
}
__device__ __forceinline__ double swe_wave_max(double v)
{
    for (int off = SWE_BLOCK/2; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, SWE_BLOCK));
    return v;
}

// diagnostics: { int eta^2, int |u|^2, int (eta+h) } as limb sums in acc[12] (+ acc[12]: terms out of range), min(h+eta) per block
static __global__ __launch_bounds__(SWE_BLOCK) void swe_diag_kernel(const double *planes, size_t stride, const int *cv,
                                                             const double *vx, const double *vy, const double *vh,
                                                             int n, double *partial, const double *valpha,
                                                             unsigned long long *acc)
{
    const int k = blockIdx.x*SWE_BLOCK + threadIdx.x;
    double s_e2 = 0.0, s_u2 = 0.0, s_vol = 0.0, s_min = 1e300;
    if (k < n) {
        double u[3], v[3], e[3], px[3], py[3], h[3];
        for (int i = 0; i < 3; i++) {
            u[i] = planes[(size_t)i*stride + k];
            v[i] = planes[(size_t)(3 + i)*stride + k];
            e[i] = planes[(size_t)(6 + i)*stride + k];
            const int vid = cv[(size_t)i*stride + k];
            px[i] = vx[vid]; py[i] = vy[vid]; h[i] = vh[vid];
            if (valpha) { const double D = e[i]; e[i] = D - 0.25*valpha[vid]*valpha[vid]/D - h[i]; h[i] = D; }   // the planes hold D
            else h[i] = h[i] + e[i];                                                   // nodal total depth
        }
        const double A = 0.5*((px[1] - px[0])*(py[2] - py[0]) - (px[2] - px[0])*(py[1] - py[0]));
        s_e2 = A*(1.0/12.0)*swe_int2(e, e);
        s_u2 = A*(1.0/12.0)*(swe_int2(u, u) + swe_int2(v, v));
        s_vol = A*(1.0/3.0)*(h[0] + h[1] + h[2]);
        s_min = fmin(fmin(h[0], h[1]), h[2]);
    }
    acc += (size_t)(blockIdx.x & (SWE_DIAG_BUCKETS - 1))*SWE_DIAG_ACC;
    swe_sum_accumulate(s_e2, acc, acc + 3*SWE_SUM_LIMBS);
    swe_sum_accumulate(s_u2, acc + SWE_SUM_LIMBS, acc + 3*SWE_SUM_LIMBS);
    swe_sum_accumulate(s_vol, acc + 2*SWE_SUM_LIMBS, acc + 3*SWE_SUM_LIMBS);
    s_min = swe_wave_min(s_min);
    if (threadIdx.x == 0) partial[blockIdx.x] = s_min;
}

// ===============================================================================================================
// 2D tracer advection stage (non-conservative form) and the vertex-based P1DG limiter
//   thetis/tracer_eq_2d.py:147-193 (HorizontalAdvectionTerm), :281-298 (SourceTerm)
//   thetis/limiter.py:48-198 + firedrake.VertexBasedLimiter [FD-assumed]
//   thetis/coupled_timeintegrator_2d.py:93-113: SWE step, then every tracer with the UPDATED velocity, then the limiter
// Same structure as the SWE stage: one lane per triangle, all loads up front, closed-form cell integrals, fused mass
// inverse and Shu-Osher combine.  Algorithmic bytes per cell per stage: T read 24 + write 24 (+24 T0 in stages 2,3)
// + u,v read 48 + static 36.
// ===============================================================================================================
struct SweTracerArgs {
    const double *tin;     // 3 planes
    const double *t0;      // 3 planes (stage_sol[0])
    double *tout;          // 3 planes
    double *mean_out;      // or null: cell means of the output for the limiter (= swe_limiter_cell_mean, same arithmetic)
    // horizontal diffusion fused into the triangle tracer kernel (DIFF variants; swe_diff_interior)
    const int4 *opp4;      // see SweStageArgs
    const double *mu_v;    // per-vertex diffusivity or null (then mu_const)
    double mu_const, diff_sipg;
    const double *uv;      // SWE state planes (u0 u1 u2 v0 v1 v2 ...): the advecting velocity
    size_t stride;
    const int *nbr, *cv;
    const int4 *idx4;      // packed triangle connectivity, see SweStageArgs (null for quadrilaterals)
    const int2 *idx2;
    const int4 *idxc;      // its 16-B form or null, see SweStageArgs
    const double *vx, *vy;
    int cell_begin, cell_end;
    double dt, a0, a1, beta;
    double vel_factor;     // tracer_advective_velocity_factor
    double lf_factor;      // lax_friedrichs_tracer_scaling_factor
    const double *source;  // 3 planes or null
    int conservative;      // ConservativeHorizontalAdvectionTerm / ConservativeSourceTerm (tracer_eq_2d.py:325-437)
    int depth_mode;        // total depth of the conservative source term: 0 = h, 1 = h + eta, 2 = wetting-drying D
    const double *vh, *valpha;
    int bc_has_value[SWE_MAX_MARKERS];   // 0: no 'value', 1: constant, 2: Function (bc_value_f)
    double bc_value[SWE_MAX_MARKERS];
    const double *bc_value_f;            // k*k planes: plane k*f + i = node i of the cell for its boundary facet f, or null
    int bc_vel_kind[SWE_MAX_MARKERS];    // external velocity of the boundary dict: 0 = uv_in, 1 = 'uv' (times vel_factor), 2 = 'un'*n,
                                         // 3 = 'flux' (bc_u) with elev_in, 4 = 'flux' with the constant 'elev' in bc_v
    double bc_len[SWE_MAX_MARKERS];      // boundary lengths (the 'flux' key divides by H_ext * length)
    double bc_u[SWE_MAX_MARKERS], bc_v[SWE_MAX_MARKERS];      // 'uv' components, or 'un' in bc_u
    // Function-valued 'uv' / 'un' / 'flux' of the boundary dict, per boundary facet like the shallow water boundary fields:
    // plane 2f (+1) = value at the first (second) node of facet f, second component at + 2k planes; or null
    const double *bc_vel_f;
    int bc_vel_field[SWE_MAX_MARKERS];   // 1: the marker's 'uv' / 'un' / 'flux' comes from bc_vel_f
};

// boundary facet of the tracer stage kernels (tracer_eq_2d.py:177-191 and :380-393): upwind value / flux with the external
// state of get_bnd_functions (:70-110).  nxs, nys: normal scaled by the facet length; returns the form value times |F|.
// |uv_ext| of the tracers' 'flux' boundary key (tracer_eq_2d.py:105-109): corr_factor*flux/(H_ext*boundary_len), H_ext the total
// depth at the external elevation (the constant 'elev' of the dict, else the interior one)
__device__ __forceinline__ double swe_tracer_flux_speed(int depth_mode, double hq, double eq, double alq, int elev_given,
                                                        double elev_ext, double flux, double len, double vel_factor)
{
#pragma clang fp contract(off)
    const double ee = elev_given ? elev_ext : eq;
    const double Hx = depth_mode == 2 ? swe_wd_depth(hq + ee, alq) : (depth_mode == 1 ? hq + ee : hq);
    return vel_factor*flux/(Hx*len);
}

// xa, xb: weights of the facet's end nodes at the quadrature point; k, f, npc_, S: where the Function-valued boundary
// velocity of this facet lives in bc_vel_f
__device__ __forceinline__ double swe_tracer_boundary_flux(const SweTracerArgs &p, int marker, double cq, double cext,
                                                           double uq, double vq, double nxs, double nys, double hq,
                                                           double eq, double alq, double xa, double xb, int k, int f,
                                                           int npc_, size_t S)
{
    // No implicit contraction: inlined into the stage kernels, their boundary epilogues and the fused three-stage kernel, which must all
    // give the same bits (which product of a*b + c*d is fused depends on the surrounding code otherwise)
#pragma clang fp contract(off)
    double ue = uq, ve = vq;
    const int vk = p.bc_vel_kind[marker];
    double bu = p.bc_u[marker], bv = p.bc_v[marker];
    if (p.bc_vel_field[marker] && p.bc_vel_f) {                  // tracer_eq_2d.py:100-109 with Function-valued entries
        const size_t pa = (size_t)(2*f)*S + k;
        bu = swe_dot2(xa, p.bc_vel_f[pa], xb, p.bc_vel_f[pa + S]);
        if (vk == 1) bv = swe_dot2(xa, p.bc_vel_f[pa + (size_t)(2*npc_)*S], xb, p.bc_vel_f[pa + (size_t)(2*npc_ + 1)*S]);
    }
    if (vk >= 3) {
        const double rl = 1.0/sqrt(swe_dot2(nxs, nxs, nys, nys));
        const double sp = swe_tracer_flux_speed(p.depth_mode, hq, eq, alq, vk == 4, bv, bu, p.bc_len[marker], p.vel_factor);
        ue = sp*nxs*rl;
        ve = sp*nys*rl;
    } else if (vk == 1) {
        ue = p.vel_factor*bu;
        ve = p.vel_factor*bv;
    } else if (vk == 2) {
        const double rl = 1.0/sqrt(swe_dot2(nxs, nxs, nys, nys));
        ue = bu*nxs*rl;
        ve = bu*nys*rl;
    }
    const double unav = 0.5*swe_dot2(uq + ue, nxs, vq + ve, nys);
    if (p.conservative) {                    // flux_up = c_in*uv*s + c_ext*uv_ext*(1-s)
        const double fin = cq*swe_dot2(uq, nxs, vq, nys), fex = cext*swe_dot2(ue, nxs, ve, nys);
        return unav > 0.0 ? fin : (unav < 0.0 ? fex : 0.5*(fin + fex));
    }
    const double cup = unav > 0.0 ? cq : (unav < 0.0 ? cext : 0.5*(cq + cext));
    return cup*unav;
}

// tracer HorizontalDiffusionTerm (SIPG, thetis/tracer_eq_2d.py:226-278) of a triangle without its boundary-facet terms,
// inside the tracer stage kernel: same arithmetic as swe_sipg_kernel<1>; see swe_visc_interior.
__device__ __forceinline__ void swe_diff_interior(const SweTracerArgs &p, int k, unsigned S8, swe_rsrc_t gt, const int nb[3],
                                                  const int vid[3], const double c[3], const double cna[3],
                                                  const double cnb[3], const double px[3], const double py[3],
                                                  const double nx[3], const double ny[3], double twoA, double b[3])
{
    const int4 o4 = p.opp4[k];
    const swe_rsrc_t rvx = swe_rsrc(p.vx), rvy = swe_rsrc(p.vy);
    double co[3], xo[3], yo[3], mu[3];
#pragma unroll
    for (int f = 0; f < 3; f++) {
        const int nbf = nb[f];
        const int kn = nbf >= 0 ? (nbf >> 2) : k;
        const int f2 = nbf >= 0 ? (nbf & 3) : f;
        co[f] = swe_ld(gt, (unsigned)kn*8u + (f2 == 0 ? 2u*S8 : (f2 == 1 ? 0u : S8)), 0);      // node (f2 + 2) % 3
        const unsigned v8 = (unsigned)(f == 0 ? o4.x : (f == 1 ? o4.y : o4.z))*8u;
        xo[f] = swe_ld(rvx, v8, 0);
        yo[f] = swe_ld(rvy, v8, 0);
        mu[f] = p.mu_v ? swe_ld(swe_rsrc(p.mu_v), (unsigned)vid[f]*8u, 0) : p.mu_const;
    }
    const double A = 0.5*twoA, r2A = swe_rcp(twoA);
    double gx[3], gy[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        gx[i] = -nx[(i + 1) % 3]*r2A;
        gy[i] = -ny[(i + 1) % 3]*r2A;
    }
    const double G0 = c[0]*gx[0] + c[1]*gx[1] + c[2]*gx[2], G1 = c[0]*gy[0] + c[1]*gy[1] + c[2]*gy[2];
    double d[3];
    {
        const double am = A*(mu[0] + mu[1] + mu[2])*(1.0/3.0);
#pragma unroll
        for (int i = 0; i < 3; i++) d[i] = -am*(gx[i]*G0 + gy[i]*G1);
    }
#pragma unroll
    for (int f = 0; f < 3; f++) {
        if (nb[f] < 0) continue;
        const int a = f, bb = (f + 1) % 3;
        const double nxs = nx[f], nys = ny[f];
        double L, rL;
        swe_sqrt_rsqrt(nxs*nxs + nys*nys, L, rL);
        const double n0 = nxs*rL, n1 = nys*rL;
        const double w = 0.5*L;
        const double e1x = px[bb] - px[a], e1y = py[bb] - py[a];
        const double e2x = xo[f] - px[a], e2y = yo[f] - py[a];
        const double det = e1x*e2y - e1y*e2x;
        const double rdet = swe_rcp(det);
        const double An = 0.5*fabs(det);
        const double d1 = cnb[f] - cna[f], d2 = co[f] - cna[f];
        const double Gn0 = (d1*e2y - d2*e1y)*rdet, Gn1 = (d2*e1x - d1*e2x)*rdet;
        const double sigma = p.diff_sipg*L*swe_rcp(fmin(A, An));
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const double xb = q ? SWE_XI1 : SWE_XI0, xa = 1.0 - xb;
            const double muq = xa*mu[a] + xb*mu[bb];
            const double jmp = (xa*c[a] + xb*c[bb]) - (xa*cna[f] + xb*cnb[f]);
            const double sj0 = muq*(jmp*n0), sj1 = muq*(jmp*n1);
            const double sjn = sj0*n0 + sj1*n1;
            const double avn = 0.5*muq*((G0 + Gn0)*n0 + (G1 + Gn1)*n1);
            const double val = sigma*sjn - avn;
            d[a] -= w*xa*val;
            d[bb] -= w*xb*val;
#pragma unroll
            for (int i = 0; i < 3; i++) d[i] += w*0.5*(gx[i]*sj0 + gy[i]*sj1);
        }
    }
#pragma unroll
    for (int i = 0; i < 3; i++) b[i] += d[i];
}

// Boundary facets of a triangle whose marker carries a boundary value or an external velocity, after the outputs (the counterpart of
// swe_tracer_boundary_epilogue_quad): inputs reloaded by plane index, (M^-1 d)_i = 3/A (4 d_i - sum d) with s = 6 dt beta / (2A).
__device__ __forceinline__ void swe_tracer_boundary_epilogue_tri(const SweTracerArgs &p, int k, unsigned bdefer, double s, double o[3])
{
    const size_t S = p.stride;
    const double cf = p.vel_factor;
    int vid[3];
    double px[3], py[3];
    for (int i = 0; i < 3; i++) { vid[i] = p.cv[(size_t)i*S + k]; px[i] = p.vx[vid[i]]; py[i] = p.vy[vid[i]]; }
#pragma unroll 1
    for (int f = 0; f < 3; f++) {
        if (!((bdefer >> f) & 1u)) continue;
        const int a = f, bb = (f == 2) ? 0 : f + 1;
        const int marker = -p.nbr[(size_t)f*S + k];
        const double ua = cf*p.uv[(size_t)a*S + k], ub = cf*p.uv[(size_t)bb*S + k];
        const double va = cf*p.uv[(size_t)(3 + a)*S + k], vb = cf*p.uv[(size_t)(3 + bb)*S + k];
        const double ca = p.tin[(size_t)a*S + k], cb = p.tin[(size_t)bb*S + k];
        const double nxs = py[bb] - py[a], nys = px[a] - px[bb];
        double Fa = 0.0, Fb = 0.0;
#pragma unroll 1
        for (int q = 0; q < 2; q++) {
            const double xb = q ? SWE_XI1 : SWE_XI0, xa = 1.0 - xb;
            const double uq = xa*ua + xb*ub, vq = xa*va + xb*vb, cq = xa*ca + xb*cb;
            const double cext = (p.bc_has_value[marker] == 2)
                ? xa*p.bc_value_f[(size_t)(3*f + a)*S + k] + xb*p.bc_value_f[(size_t)(3*f + bb)*S + k]
                : (p.bc_has_value[marker] ? p.bc_value[marker] : cq);
            double hq = 0.0, eq = 0.0, alq = 0.0;
            if (p.bc_vel_kind[marker] >= 3) {                  // 'flux': total depth at the quadrature point
                const double ha = p.vh[vid[a]], hb = p.vh[vid[bb]];
                hq = xa*ha + xb*hb;
                double ea_ = p.uv[(size_t)(6 + a)*S + k], eb_ = p.uv[(size_t)(6 + bb)*S + k];
                if (p.depth_mode == 2) {           // the planes hold D: the nodal elevations by the closed form
                    const double aa_ = p.valpha[vid[a]], ab_ = p.valpha[vid[bb]];
                    alq = xa*aa_ + xb*ab_;
                    ea_ = ea_ - 0.25*aa_*aa_/ea_ - ha;
                    eb_ = eb_ - 0.25*ab_*ab_/eb_ - hb;
                }
                eq = xa*ea_ + xb*eb_;
            }
            const double fq = swe_tracer_boundary_flux(p, marker, cq, cext, uq, vq, nxs, nys, hq, eq, alq, xa, xb, k, f, 3, S);
            Fa += xa*fq;
            Fb += xb*fq;
        }
        const double da = -0.5*Fa, db = -0.5*Fb, sd = da + db;
#pragma unroll
        for (int i = 0; i < 3; i++) o[i] += s*(4.0*((i == a) ? da : ((i == bb) ? db : 0.0)) - sd);
    }
}

// Advective right-hand side integrals of ONE triangle for the tracer stage (tracer_eq_2d.py:147-193, :341-395): cell integral, source,
// the three facets.  Shared by swe_tracer_stage_kernel and the three-stages-in-one-launch kernel of swe2d_fuse.h, which must give
// the same bits: no implicit contraction, every fused multiply-add written out (round 6; rounds 1-5 left the contraction to the
// compiler, whose choice for a*b + c*d depends on the code around it).
// u, v: the cell's velocity times tracer_advective_velocity_factor; una / unb (vna / vnb, cna / cnb): the neighbour's velocity
// (tracer) at its node on my node f + 1 / on my node f, for a boundary facet the cell's own.
// DEFER: boundary facets whose bit is set in `bdefer` contribute nothing here (the instances with the diffusion fused in evaluate
// them after the outputs, swe_tracer_boundary_epilogue_tri); otherwise every boundary facet is evaluated in place.
template <bool LF, bool SRC, bool DEFER>
__device__ __forceinline__ void swe_tracer_rhs_tri(const SweTracerArgs &p, int k, unsigned k8, unsigned S8, const int nb[3], const int vid[3],
                                                   const double u[3], const double v[3], const double c[3], const double una[3],
                                                   const double unb[3], const double vna[3], const double vnb[3], const double cna[3],
                                                   const double cnb[3], const double nx[3], const double ny[3], double twoA,
                                                   unsigned bdefer, double b[3])
{
#pragma clang fp contract(off)
    const size_t S = p.stride;
    double gxs[3], gys[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        gxs[i] = -0.5*nx[(i + 1) % 3];
        gys[i] = -0.5*ny[(i + 1) % 3];
    }
    // cell integral  +(phi div u + u.grad phi) c                                      tracer_eq_2d.py:157-158
    {
        const double D12 = fma(gys[2], v[2], fma(gys[1], v[1], fma(gys[0], v[0],
                           fma(gxs[2], u[2], fma(gxs[1], u[1], gxs[0]*u[0])))))*(1.0/12.0);
        const double Suc = swe_int2(u, c)*(1.0/12.0), Svc = swe_int2(v, c)*(1.0/12.0);
        const double cs = c[0] + c[1] + c[2];
        // conservative form: +grad(phi).u q only                                      tracer_eq_2d.py:358-359
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const double t = swe_dot2(gys[i], Svc, gxs[i], Suc);
            b[i] = p.conservative ? t : fma(D12, cs + c[i], t);
        }
    }
    if (SRC) {                                                                         // tracer_eq_2d.py:293-297
        const double A = 0.5*twoA;
        double s[3];
#pragma unroll
        for (int i = 0; i < 3; i++) s[i] = swe_ld(swe_rsrc(p.source), k8, i*S8);
        const double ss = s[0] + s[1] + s[2];
        if (p.conservative) {                                                          // H*source, :434-436
            double H[3];
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const double hh = p.vh[vid[i]];
                const double ee = swe_ld(swe_rsrc(p.uv + 6*S), k8, i*S8);
                H[i] = p.depth_mode == 2 ? ee : (p.depth_mode == 1 ? hh + ee : hh);       // wetting-drying: the planes hold D
            }
            const double Hs = H[0] + H[1] + H[2], Hss = fma(H[2], s[2], fma(H[1], s[1], H[0]*s[0]));
            const double A60 = A*(1.0/60.0);
#pragma unroll
            for (int i = 0; i < 3; i++)      // 60/A int phi_i H s = Hs*ss + sum H_a s_a + H_i*ss + s_i*Hs + 2 H_i s_i
                b[i] = fma(A60, fma(2.0*H[i], s[i], fma(s[i], Hs, fma(H[i], ss, fma(Hs, ss, Hss)))), b[i]);
        } else {
            const double A12 = A*(1.0/12.0);
#pragma unroll
            for (int i = 0; i < 3; i++) b[i] = fma(A12, ss + s[i], b[i]);
        }
    }
#pragma unroll
    for (int f = 0; f < 3; f++) {
        const int a = f, bb = (f + 1) % 3;
        const double nxs = nx[f], nys = ny[f];
        double Fa = 0.0, Fb = 0.0;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const double xb = q ? SWE_XI1 : SWE_XI0, xa = 1.0 - xb;
            const double uq = swe_dot2(xa, u[a], xb, u[bb]), vq = swe_dot2(xa, v[a], xb, v[bb]), cq = swe_dot2(xa, c[a], xb, c[bb]);
            const double unown = swe_dot2(uq, nxs, vq, nys);            // |F| u.n of this side   :170-171
            double fq;
            if (nb[f] >= 0) {
                const double un = swe_dot2(xa, una[f], xb, unb[f]), vn = swe_dot2(xa, vna[f], xb, vnb[f]), cn = swe_dot2(xa, cna[f], xb, cnb[f]);
                const double uavn = 0.5*swe_dot2(uq + un, nxs, vq + vn, nys);              // :163-165
                const double cup = uavn > 0.0 ? cq : (uavn < 0.0 ? cn : 0.5*(cq + cn));   // :166-168
                fq = cup*unown;
                if (p.conservative) {                           // upwind FLUX q u (both from the upwind side), :366-372
                    const double fn = cn*swe_dot2(un, nxs, vn, nys);
                    fq = uavn > 0.0 ? cq*unown : (uavn < 0.0 ? fn : 0.5*(cq*unown + fn));
                }
                if (LF) fq = fma(0.5*fabs(uavn)*p.lf_factor, cq - cn, fq);                 // :173-175
            } else if constexpr (DEFER) {
                // boundary facet of the instances with the diffusion fused in (176-192 VGPRs with the code below inlined: two waves per
                // SIMD): with a boundary value or an external velocity it is evaluated after the outputs (swe_tracer_boundary_epilogue_tri,
                // see swe_tracer_stage_kernel_quad) - 150-162 VGPRs, three waves, 1 M cells 127.2 -> 108.9 us per step; here only the
                // default, the interior state on both sides                                                        :189-191
                fq = (bdefer >> f) & 1u ? 0.0 : cq*unown;
            } else {
                // (the plain instances keep the boundary code inline: at 154 VGPRs they run three waves already, and the epilogue
                //  form measured 3.5 % slower for them - 102.7 against 99.2 us per step, profiles/r05z3)
                const int marker = -nb[f];
                if (marker < SWE_MAX_MARKERS && (p.bc_has_value[marker] || p.bc_vel_kind[marker])) {     // :181-188
                    const double cext = (p.bc_has_value[marker] == 2)
                        ? swe_dot2(xa, p.bc_value_f[(size_t)(3*f + a)*S + k], xb, p.bc_value_f[(size_t)(3*f + bb)*S + k])
                        : (p.bc_has_value[marker] ? p.bc_value[marker] : cq);
                    double hq = 0.0, eq = 0.0, alq = 0.0;
                    if (p.bc_vel_kind[marker] >= 3) {                  // 'flux': total depth at the quadrature point
                        const unsigned va8 = (unsigned)vid[a]*8u, vb8 = (unsigned)vid[bb]*8u;
                        hq = swe_dot2(xa, swe_ld(swe_rsrc(p.vh), va8, 0), xb, swe_ld(swe_rsrc(p.vh), vb8, 0));
                        if (p.depth_mode == 2) alq = swe_dot2(xa, swe_ld(swe_rsrc(p.valpha), va8, 0), xb, swe_ld(swe_rsrc(p.valpha), vb8, 0));
                        const swe_rsrc_t ge = swe_rsrc(p.uv + 6*S);
                        double ea_ = swe_ld(ge, k8, a*S8), eb_ = swe_ld(ge, k8, bb*S8);
                        if (p.depth_mode == 2) {           // the planes hold D: the nodal elevations by the closed form
                            const double aa_ = swe_ld(swe_rsrc(p.valpha), va8, 0), ab_ = swe_ld(swe_rsrc(p.valpha), vb8, 0);
                            ea_ = ea_ - 0.25*aa_*aa_/ea_ - swe_ld(swe_rsrc(p.vh), va8, 0);
                            eb_ = eb_ - 0.25*ab_*ab_/eb_ - swe_ld(swe_rsrc(p.vh), vb8, 0);
                        }
                        eq = swe_dot2(xa, ea_, xb, eb_);
                    }
                    fq = swe_tracer_boundary_flux(p, marker, cq, cext, uq, vq, nxs, nys, hq, eq, alq, xa, xb, k, f, 3, S);
                } else {
                    fq = cq*unown;                                                         // :189-191
                }
            }
            Fa = fma(xa, fq, Fa);
            Fb = fma(xb, fq, Fb);
        }
        b[a] = fma(-0.5, Fa, b[a]);
        b[bb] = fma(-0.5, Fb, b[bb]);
    }
}

// mass inverse and Shu-Osher combine of a triangle's tracer stage: o = beta dt M^-1 b + w, (M^-1 b)_i = 3/A (4 b_i - sum b)
__device__ __forceinline__ double swe_tracer_finish_tri(double dt, double beta, double twoA, const double b[3], const double w[3], double o[3])
{
#pragma clang fp contract(off)
    const double s = 6.0*dt*beta*swe_rcp(twoA);
    const double sb = b[0] + b[1] + b[2];
#pragma unroll
    for (int i = 0; i < 3; i++) o[i] = fma(s, fma(4.0, b[i], -sb), w[i]);
    return s;
}

template <bool LF, bool HAST0, bool SRC, bool DIFF = false>
__global__ __launch_bounds__(SWE_BLOCK) void swe_tracer_stage_kernel(const SweTracerArgs p)
{
#pragma clang fp contract(off)
#ifdef SWE_NO_XCD_MAP
    const int lb = blockIdx.x;
#else
    const int lb = swe_logical_block(blockIdx.x, gridDim.x);
#endif
    const int k = p.cell_begin + lb*SWE_BLOCK + (int)threadIdx.x;
    if (k >= p.cell_end) return;
    const size_t S = p.stride;
    const double cf = p.vel_factor;
    const unsigned S8 = (unsigned)S*8u, k8 = (unsigned)k*8u;       // raw buffer addressing, see swe_ld
    const swe_rsrc_t gu = swe_rsrc(p.uv), gv = swe_rsrc(p.uv + 3*S), gt = swe_rsrc(p.tin);

    double u[3], v[3], c[3], w[3];
    int nb[3], vid[3];
    swe_conn_load(p.idxc, p.idx4, p.idx2, k, nb, vid);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        u[i] = cf*swe_ld(gu, k8, i*S8);
        v[i] = cf*swe_ld(gv, k8, i*S8);
        c[i] = swe_ld(gt, k8, i*S8);
        w[i] = p.a1*c[i];
        if (HAST0) w[i] = fma(p.a0, swe_ld(swe_rsrc(p.t0), k8, i*S8), w[i]);
    }
    unsigned bdefer = 0u;             // boundary facets whose marker carries a boundary value or an external velocity: bit f
    if (DIFF && (nb[0] | nb[1] | nb[2]) < 0) {      // (the tables are read by the lanes - and waves - that own a boundary facet only)
#pragma unroll
        for (int f = 0; f < 3; f++) {
            const int marker = nb[f] < 0 ? -nb[f] : 0;
            if (marker > 0 && marker < SWE_MAX_MARKERS && (p.bc_has_value[marker] || p.bc_vel_kind[marker])) bdefer |= 1u << f;
        }
    }
    double una[3], unb[3], vna[3], vnb[3], cna[3], cnb[3];
#pragma unroll
    for (int f = 0; f < 3; f++) {
        const int nbf = nb[f];
        const int kn = nbf >= 0 ? (nbf >> 2) : k;
        const int f2 = nbf >= 0 ? (nbf & 3) : f;
        const unsigned kn8 = (unsigned)kn*8u;
        const unsigned ob = kn8 + (f2 == 0 ? 0u : (f2 == 1 ? S8 : 2u*S8));         // node f2
        const unsigned oa = kn8 + (f2 == 0 ? S8 : (f2 == 1 ? 2u*S8 : 0u));         // node (f2 + 1) % 3
        una[f] = cf*swe_ld(gu, oa, 0);
        unb[f] = cf*swe_ld(gu, ob, 0);
        vna[f] = cf*swe_ld(gv, oa, 0);
        vnb[f] = cf*swe_ld(gv, ob, 0);
        cna[f] = swe_ld(gt, oa, 0);
        cnb[f] = swe_ld(gt, ob, 0);
    }
    double px[3], py[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        px[i] = swe_ld(swe_rsrc(p.vx), (unsigned)vid[i]*8u, 0);
        py[i] = swe_ld(swe_rsrc(p.vy), (unsigned)vid[i]*8u, 0);
    }
    double nx[3], ny[3];
#pragma unroll
    for (int f = 0; f < 3; f++) {
        const int b = (f + 1) % 3;
        nx[f] = py[b] - py[f];
        ny[f] = px[f] - px[b];
    }
    const double twoA = fma(nx[0], ny[1], -(ny[0]*nx[1]));
    double b[3];
    swe_tracer_rhs_tri<LF, SRC, DIFF>(p, k, k8, S8, nb, vid, u, v, c, una, unb, vna, vnb, cna, cnb, nx, ny, twoA, bdefer, b);
    if (DIFF) swe_diff_interior(p, k, S8, gt, nb, vid, c, cna, cnb, px, py, nx, ny, twoA, b);
    double msum = 0.0, o[3];
    const double s = swe_tracer_finish_tri(p.dt, p.beta, twoA, b, w, o);
    if (bdefer) swe_tracer_boundary_epilogue_tri(p, k, bdefer, s, o);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        swe_st(swe_rsrc(p.tout), k8, i*S8, o[i]);
        msum += o[i];
    }
    if (p.mean_out) p.mean_out[k] = msum/3.0;
}

// ---- limiter, step 1: cell means (P0 projection of an affine P1 / Q1 field = mean of the nodal values)
// (general quadrilaterals, cv != nullptr: the P0 projection is the mass-weighted mean, see swe_quad_mean_weights below)
__device__ __forceinline__ void swe_quad_mean_weights(double d0, double d1, double d2, double w[4]);
static __global__ void swe_limiter_cell_mean(const double *t, size_t stride, int n, double *mean, int npc, const int *cv,
                                      const double *vx, const double *vy)
{
    const int k = blockIdx.x*blockDim.x + threadIdx.x;
    if (k >= n) return;
    if (cv) {
        double px[4], py[4], w[4];
        for (int i = 0; i < 4; i++) { const int v = cv[(size_t)i*stride + k]; px[i] = vx[v]; py[i] = vy[v]; }
        const double ax = px[1] - px[0], ay = py[1] - py[0], bx = px[3] - px[0], by = py[3] - py[0];
        const double cx = (px[0] - px[1]) + (px[2] - px[3]), cy = (py[0] - py[1]) + (py[2] - py[3]);
        swe_quad_mean_weights(ax*by - ay*bx, ax*cy - ay*cx, cx*by - cy*bx, w);
        double s = 0.0;
        for (int i = 0; i < 4; i++) s += w[i]*t[(size_t)i*stride + k];
        mean[k] = s;
        return;
    }
    double s = 0.0;
    for (int i = 0; i < npc; i++) s += t[(size_t)i*stride + k];
    mean[k] = s/(double)npc;
}

// ---- limiter, step 2: per (topological) vertex min/max of the means of the cells around it (CSR gather: no atomics,
// deterministic) + Thetis's boundary-facet means (limiter.py:109-145)
static __global__ void swe_limiter_vertex_bounds(const int *v2c_off, const int *v2c_cell, const int *vbf_off, const int *vbf_facet,
                                          const double *mean, const double *t, size_t stride, int nv,
                                          double *qmin, double *qmax, int npc)
{
    const int v = blockIdx.x*blockDim.x + threadIdx.x;
    if (v >= nv) return;
    double lo = 1.0e10, hi = -1.0e10;
    for (int j = v2c_off[v]; j < v2c_off[v + 1]; j++) {
        const int kc = v2c_cell[j];
        double m;
        if (mean) m = mean[kc];
        else {                                   // no means array: the nodal average formed here (swe_limiter_cell_mean's sum, same bits)
            double s = 0.0;
            for (int i = 0; i < npc; i++) s += t[(size_t)i*stride + kc];
            m = s/(double)npc;
        }
        lo = fmin(lo, m);
        hi = fmax(hi, m);
    }
    for (int j = vbf_off[v]; j < vbf_off[v + 1]; j++) {
        const int packed = vbf_facet[j];
        const int k = packed >> 2, a = packed & 3, b = (a + 1 == npc) ? 0 : a + 1;
        const double fm = (t[(size_t)a*stride + k] + t[(size_t)b*stride + k])/2.0;
        lo = fmin(lo, fm);
        hi = fmax(hi, fm);
    }
    qmin[v] = lo;
    qmax[v] = hi;
}

// ---- limiter, step 3: per-cell scaling towards the mean
// (``mean_in`` != nullptr, general quadrilaterals: the mass-weighted means of step 1 instead of the nodal average)
static __global__ void swe_limiter_apply(double *t, size_t stride, int n, const int *tv, const double *qmin, const double *qmax,
                                  int npc, const double *mean_in)
{
    const int k = blockIdx.x*blockDim.x + threadIdx.x;
    if (k >= n) return;
    double c[4];
    int vv[4];
    double s = 0.0;
    for (int i = 0; i < npc; i++) {
        c[i] = t[(size_t)i*stride + k];
        vv[i] = tv[(size_t)i*stride + k];
        s += c[i];
    }
    const double mean = mean_in ? mean_in[k] : s/(double)npc;
    double alpha = 1.0;
    for (int i = 0; i < npc; i++) {
        if (c[i] > mean) alpha = fmin(alpha, fmin(1.0, (qmax[vv[i]] - mean)/(c[i] - mean)));
        else if (c[i] < mean) alpha = fmin(alpha, fmin(1.0, (mean - qmin[vv[i]])/(mean - c[i])));
    }
    for (int i = 0; i < npc; i++) t[(size_t)i*stride + k] = mean + alpha*(c[i] - mean);
}

// ---- tracer diagnostics: per-block { int T*H dx, int T dx, min T, max T }
static __global__ __launch_bounds__(SWE_BLOCK) void swe_tracer_diag_kernel(const double *t, const double *state, size_t stride,
                                                                    const int *cv, const double *vx, const double *vy,
                                                                    const double *vh, int nonlinear, int n, double *partial,
                                                                    const double *valpha, unsigned long long *acc)
{
    const int k = blockIdx.x*SWE_BLOCK + threadIdx.x;
    double s_m = 0.0, s_i = 0.0, s_min = 1e300, s_max = -1e300;
    if (k < n) {
        double c[3], H[3], px[3], py[3];
        for (int i = 0; i < 3; i++) {
            c[i] = t[(size_t)i*stride + k];
            const int vid = cv[(size_t)i*stride + k];
            px[i] = vx[vid]; py[i] = vy[vid];
            H[i] = vh[vid] + (nonlinear ? state[(size_t)(6 + i)*stride + k] : 0.0);
            if (valpha) H[i] = state[(size_t)(6 + i)*stride + k];      // total depth with wetting-drying: the displaced depth D the planes hold
        }
        const double A = 0.5*((px[1] - px[0])*(py[2] - py[0]) - (px[2] - px[0])*(py[1] - py[0]));
        s_m = A*(1.0/12.0)*swe_int2(c, H);
        s_i = A*(1.0/3.0)*(c[0] + c[1] + c[2]);
        s_min = fmin(fmin(c[0], c[1]), c[2]);
        s_max = fmax(fmax(c[0], c[1]), c[2]);
    }
    acc += (size_t)(blockIdx.x & (SWE_DIAG_BUCKETS - 1))*SWE_DIAG_ACC;
    swe_sum_accumulate(s_m, acc, acc + 2*SWE_SUM_LIMBS);
    swe_sum_accumulate(s_i, acc + SWE_SUM_LIMBS, acc + 2*SWE_SUM_LIMBS);
    s_min = swe_wave_min(s_min);
    s_max = swe_wave_max(s_max);
    if (threadIdx.x == 0) { partial[2*(size_t)blockIdx.x] = s_min; partial[2*(size_t)blockIdx.x + 1] = s_max; }
}

// scalar nodal field (3N) <-> 3 planes
static __global__ void swe_planes_to_nodal(const double *planes, double *nodal, size_t stride, int n, int npc)
{
    const int k = blockIdx.x*blockDim.x + threadIdx.x;
    if (k >= n) return;
    for (int i = 0; i < npc; i++) nodal[(size_t)npc*k + i] = planes[(size_t)i*stride + k];
}

// PMC calibration aid (MI355X_MICROARCH.md section HBM: "calibrate on a known byte count in your own access pattern"):
// streams n doubles per plane with the same 8-B/lane coalesced loads and stores the stage kernel uses.
static __global__ __launch_bounds__(SWE_BLOCK) void swe_calibration_copy(const double *src, double *dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x*SWE_BLOCK + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// ===============================================================================================================
// DQ-1 on parallelogram quadrilaterals (demos/demo_2d_tracer.py:19 mesh type; 'DQ' branch of solver2d.py:319).
// 12 SoA planes u0..u3 v0..v3 e0..e3; bilinear basis on the unit square, nodes counter-clockwise from (0,0):
//   x = p0 + xi a + zeta b,  a = p1 - p0,  b = p3 - p0,  detJ = a x b = area,
//   grad xi = (b_y, -b_x)/A,  grad zeta = (-a_y, a_x)/A.
// Cell integrals: 2 x 2 Gauss-Legendre (degree 3 per direction, exact for every polynomial integrand of the path);
// facets: the same 2-point rule and numerical fluxes as the triangles (a facet carries two nodes);
// mass inverse: (M^-1 b)_i = (16 b_i - 8 b_{i+1} - 8 b_{i-1} + 4 b_{i+2})/A.
// Algorithmic bytes per cell per stage: 96 read + 96 write (+96 U0 in stages 2,3) + 56 static.
//
// One lane per cell.  Rounds 1-4: 172-223 VGPRs, two waves per SIMD.  Round 5: boundary facets after the outputs
// (swe_boundary_epilogue_quad), U(0) requested after the facet loop, optional terms in a pass of their own - 140-167 VGPRs, three
// waves per SIMD for the parallelogram kernels, +2.5 ... 4 % (profiles/r05g_quad_three_waves.txt).  Splitting a cell over TWO lanes (half-turn local frames, partner values
// by DPP quad_perm, 118-120 VGPRs = four waves per SIMD, twice the instruction streams) was built, passed the same parity tests
// and ran at the SAME speed at every size from 122 k to 2 M cells (1 M: 201-207 against 193-203 us/step, with 3 or 4 waves per
// SIMD alike): this kernel is not bound by its occupancy.  PMC: HBM traffic x 1.05 of the algorithmic bytes, 1182 VALU
// instructions per wave (~45 % of the SIMD cycles), the rest is the index -> gather -> arithmetic -> store chain of a wave
// against HBM latency, which more waves of half the size do not shorten.  A lesson from the same experiment: a cold code path
// (the boundary-facet copy) that SPILLS costs every wave - the kernel then carries a scratch allocation, 251 against 201 us/step
// although no wave of the timed mesh interior ever touches the scratch.  profiles/r03j_quad_two_lanes.txt; the kernel is in the
// history (commit "Experiment: quadrilateral stage kernel with a cell split over two lanes").
// ===============================================================================================================
// ---- general (non-affine) quadrilaterals: bilinear map x = p0 + xi a + zeta b + xi zeta c,  a = p1 - p0,  b = p3 - p0,
// c = p0 - p1 + p2 - p3 (zero on a parallelogram).  Jacobian columns x_xi = a + zeta c, x_zeta = b + xi c;
//   det J = d0 + d1 xi + d2 zeta,  d0 = a x b,  d1 = a x c,  d2 = c x b      (c x c = 0: the determinant is LINEAR);
//   det J * grad(phi) = adj(J)^T grad_ref(phi): the kernels' "A * grad" terms need no division, only x_xi, x_zeta at the point.
// Mass matrix M_ij = int det J phi_i phi_j = d0 (m (x) m) + d1 (m1 (x) m) + d2 (m (x) m1) with the 1D matrices m = [[1/3, 1/6],
// [1/6, 1/3]] and m1 = int t N_a N_b = [[1/12, 1/12], [1/12, 1/4]] - closed form, equal to what the 2 x 2 Gauss rule of the
// oracle (oracle/swe2d_oracle.py: mass_matrix) gives, which is exact for this integrand.  Node i = (xi index, zeta index):
// 0 = (0,0), 1 = (1,0), 2 = (1,1), 3 = (0,1).  The 4 x 4 solve is an LDL^T factorisation in registers (M is SPD with a
// condition number of ~9).  Reference: thetis/solver2d.py:340-345 accepts any quadrilateral mesh (DQ-1 with the true mass matrix).
struct SweQuadMass { double m[10]; };          // upper triangle, row-major: 00 01 02 03 11 12 13 22 23 33
__device__ __forceinline__ void swe_quad_mass(double d0, double d1, double d2, SweQuadMass &M)
{
#pragma clang fp contract(off)
    // 1D entries: m[a][b], m1[a][b]
    const double m_[2][2] = {{1.0/3.0, 1.0/6.0}, {1.0/6.0, 1.0/3.0}};
    const double m1[2][2] = {{1.0/12.0, 1.0/12.0}, {1.0/12.0, 1.0/4.0}};
    const int ia[4] = {0, 1, 1, 0}, ib[4] = {0, 0, 1, 1};
    int q = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = i; j < 4; j++) {
            const double c0 = m_[ia[i]][ia[j]]*m_[ib[i]][ib[j]], c1 = m1[ia[i]][ia[j]]*m_[ib[i]][ib[j]],
                         c2 = m_[ia[i]][ia[j]]*m1[ib[i]][ib[j]];
            M.m[q++] = fma(d2, c2, fma(d1, c1, d0*c0));
        }
}
// row sums of M = int det J phi_i (the weights of the P0 projection / cell mean), divided by the cell area
__device__ __forceinline__ void swe_quad_mean_weights(double d0, double d1, double d2, double w[4])
{
#pragma clang fp contract(off)
    const double rA = swe_rcp(d0 + 0.5*(d1 + d2));
    w[0] = (0.25*d0 + (1.0/12.0)*d1 + (1.0/12.0)*d2)*rA;
    w[1] = (0.25*d0 + (1.0/6.0)*d1 + (1.0/12.0)*d2)*rA;
    w[2] = (0.25*d0 + (1.0/6.0)*d1 + (1.0/6.0)*d2)*rA;
    w[3] = (0.25*d0 + (1.0/12.0)*d1 + (1.0/6.0)*d2)*rA;
}
struct SweQuadLDL { double l10, l20, l30, l21, l31, l32, r0, r1, r2, r3; };
__device__ __forceinline__ void swe_quad_mass_factor(const SweQuadMass &M, SweQuadLDL &F)
{
#pragma clang fp contract(off)
    const double *m = M.m;               // 0:00 1:01 2:02 3:03 4:11 5:12 6:13 7:22 8:23 9:33
    F.r0 = swe_rcp(m[0]);
    F.l10 = m[1]*F.r0; F.l20 = m[2]*F.r0; F.l30 = m[3]*F.r0;
    F.r1 = swe_rcp(m[4] - F.l10*m[1]);
    const double t21 = m[5] - F.l20*m[1], t31 = m[6] - F.l30*m[1];
    F.l21 = t21*F.r1; F.l31 = t31*F.r1;
    F.r2 = swe_rcp(m[7] - F.l20*m[2] - F.l21*t21);
    const double t32 = m[8] - F.l30*m[2] - F.l31*t21;
    F.l32 = t32*F.r2;
    F.r3 = swe_rcp(m[9] - F.l30*m[3] - F.l31*t31 - F.l32*t32);
}
__device__ __forceinline__ void swe_quad_mass_solve(const SweQuadLDL &F, double b[4])      // b <- M^-1 b
{
#pragma clang fp contract(off)
    const double y0 = b[0];
    const double y1 = b[1] - F.l10*y0;
    const double y2 = b[2] - F.l20*y0 - F.l21*y1;
    const double y3 = b[3] - F.l30*y0 - F.l31*y1 - F.l32*y2;
    const double x3 = y3*F.r3;
    const double x2 = y2*F.r2 - F.l32*x3;
    const double x1 = y1*F.r1 - F.l21*x2 - F.l31*x3;
    const double x0 = y0*F.r0 - F.l10*x1 - F.l20*x2 - F.l30*x3;
    b[0] = x0; b[1] = x1; b[2] = x2; b[3] = x3;
}

// Boundary facets of a quadrilateral, evaluated AFTER the cell's outputs are finished (the counterpart of swe_boundary_epilogue):
// the facet's inputs are reloaded by plane index (L1 / L2 hits: this wave has just read them), the flux goes through the mass
// inverse of a vector that is non-zero at the facet's two nodes and is added to the outputs still in registers.  Four inlined
// copies of swe_boundary_facet inside the facet loop cost every wave 24-36 VGPRs (172-201 = two waves per SIMD, rounds 1-4); with
// the boundary code here, where little else is live, the kernel fits three (140-167; profiles/r05g_quad_three_waves.txt).
// The residual is linear in the facet contributions: the same result up to summation order.
template <bool NONLIN, bool LF, bool WD, bool AFFINE>
// uin / Su / ku: where the stage's input lives - the state planes (p.uin, the stride, the cell) for the stage launches, the LDS planes of a
// tile (the lane) for the fused stage pair of swe2d_fuse.h; beta: the stage's weight
__device__ __forceinline__ void swe_boundary_epilogue_quad(const SweStageArgs &p, int k, unsigned bmarkers, double ou[4], double ov[4],
                                                           double oe[4], const double *uin, size_t Su, int ku, double beta)
{
#pragma clang fp contract(off)
    const size_t S = p.stride;
    // the cell's frame (as the stage kernel forms it)
    double A, d1 = 0.0, d2 = 0.0;
    {
        const int v0 = p.cv[k], v1 = p.cv[S + k], v3 = p.cv[3*S + k];
        const double x0 = p.vx[v0], y0 = p.vy[v0], x1 = p.vx[v1], y1 = p.vy[v1], x3 = p.vx[v3], y3 = p.vy[v3];
        const double ax = x1 - x0, ay = y1 - y0, bx = x3 - x0, by = y3 - y0;
        A = fma(ax, by, -(ay*bx));
        if (!AFFINE) {
            const int v2 = p.cv[2*S + k];
            const double x2 = p.vx[v2], y2 = p.vy[v2];
            const double cx = (x0 - x1) + (x2 - x3), cy = (y0 - y1) + (y2 - y3);
            d1 = fma(ax, cy, -(ay*cx));
            d2 = fma(cx, by, -(cy*bx));
        }
    }
    const double sb = p.dt*beta;
    const double s = AFFINE ? sb*swe_rcp(A) : sb;
    SweQuadLDL Fb;
    if constexpr (!AFFINE) {
        SweQuadMass Mb;
        swe_quad_mass(A, d1, d2, Mb);
        swe_quad_mass_factor(Mb, Fb);
    }
    // one facet at a time, everything addressed in memory by plane index: no dynamically indexed register arrays
#pragma unroll 1
    for (int f = 0; f < 4; f++) {
        const int marker = (int)((bmarkers >> (8*f)) & 0xffu);
        if (marker == 0) continue;
        const int a = f, b = (f + 1) & 3;
        const int va = p.cv[(size_t)a*S + k], vb = p.cv[(size_t)b*S + k];
        const double xa_ = p.vx[va], ya_ = p.vy[va], xb_ = p.vx[vb], yb_ = p.vy[vb];
        const double ha = p.vh[va], hb = p.vh[vb];
        const double ala = WD ? p.valpha[va] : 0.0, alb = WD ? p.valpha[vb] : 0.0;
        const double ua = uin[(size_t)a*Su + ku], ub = uin[(size_t)b*Su + ku];
        const double va_ = uin[(size_t)(4 + a)*Su + ku], vb_ = uin[(size_t)(4 + b)*Su + ku];
        const double da_ = uin[(size_t)(8 + a)*Su + ku], db_ = uin[(size_t)(8 + b)*Su + ku];      // eta, or D with wetting-drying
        const double ea = WD ? swe_wd_eta(da_, ha, ala) : da_, eb = WD ? swe_wd_eta(db_, hb, alb) : db_;
        const double Ha = WD ? da_ : (NONLIN ? ha + ea : ha);
        const double Hb = WD ? db_ : (NONLIN ? hb + eb : hb);
        const double nxs = yb_ - ya_, nys = xa_ - xb_;
        double L, rL;
        swe_sqrt_rsqrt(swe_dot2(nxs, nxs, nys, nys), L, rL);
        double Fau = 0.0, Fbu = 0.0, Fav = 0.0, Fbv = 0.0, Fae = 0.0, Fbe = 0.0;
        swe_boundary_facet<NONLIN, LF, WD>(p, marker, k, a, b, ua, ub, va_, vb_, ea, eb, ha, hb, Ha, Hb, ala, alb, nxs, nys,
                                           L, rL, Fau, Fbu, Fav, Fbv, Fae, Fbe);
        const double dau = -0.5*Fau, dbu = -0.5*Fbu, dav = -0.5*Fav, dbv = -0.5*Fbv, dae = -0.5*Fae, dbe = -0.5*Fbe;
        if constexpr (AFFINE) {
#pragma unroll
            for (int i = 0; i < 4; i++) {            // row i of the tensor mass inverse (16, -8, 4, -8)/A on (da at a, db at b)
                const double ca = (i == a) ? 16.0 : (i == (a ^ 2) ? 4.0 : -8.0), cb = (i == b) ? 16.0 : (i == (b ^ 2) ? 4.0 : -8.0);
                ou[i] = fma(s, fma(ca, dau, cb*dbu), ou[i]);
                ov[i] = fma(s, fma(ca, dav, cb*dbv), ov[i]);
                oe[i] = fma(s, fma(ca, dae, cb*dbe), oe[i]);
            }
        } else {
            double ru[4], rv[4], re[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                ru[i] = (i == a) ? dau : ((i == b) ? dbu : 0.0);
                rv[i] = (i == a) ? dav : ((i == b) ? dbv : 0.0);
                re[i] = (i == a) ? dae : ((i == b) ? dbe : 0.0);
            }
            swe_quad_mass_solve(Fb, ru);
            swe_quad_mass_solve(Fb, rv);
            swe_quad_mass_solve(Fb, re);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                ou[i] = fma(s, ru[i], ou[i]);
                ov[i] = fma(s, rv[i], ov[i]);
                oe[i] = fma(s, re[i], oe[i]);
            }
        }
    }
}

// One stage of one quadrilateral from its inputs in registers to its outputs in registers: geometry, the four facets (u ... e: the
// cell's nodal values as the planes hold them; una / unb ...: the neighbour's values at its node on my node f / on my node f + 1, for
// a boundary facet the cell's own), U(0), the cell quadrature, the optional terms, mass inverse, Shu-Osher combine, boundary facets,
// wetting-drying.  Shared by swe_stage_kernel_quad and the fused stage pair of swe2d_fuse.h (TILE: the stage's input and the lane's own
// U(0) live in LDS - the tile's [12] planes of stage values `tin` with stride Su, the [12] planes `t0` of U(0) with stride S0, lane ku -
// instead of the state planes):
// the same operations in the same order, the same bits.
template <bool NONLIN, bool LF, bool HASU0, bool SRC, bool WD, bool AFFINE, bool TILE>
__device__ __forceinline__ void swe_quad_stage_cell(const SweStageArgs &p, int k, unsigned k8, unsigned S8, const int nb[4], const int vid[4],
                                                    unsigned bmarkers, double u[4], double v[4], double e[4], const double una[4],
                                                    const double unb[4], const double vna[4], const double vnb[4], const double ena[4],
                                                    const double enb[4], double a0, double a1, double beta, const double *tin, size_t Su,
                                                    int ku, const double *t0, size_t S0, double ou[4], double ov[4], double oe[4])
{
#pragma clang fp contract(off)
    double wu[4], wv[4], we[4];
    const size_t S = p.stride;
    const double g = p.g;
    double px[4], py[4], h[4], H[4], al[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const unsigned v8 = (unsigned)vid[i]*8u;
        px[i] = swe_ld(swe_rsrc(p.vx), v8, 0);
        py[i] = swe_ld(swe_rsrc(p.vy), v8, 0);
        h[i] = swe_ld(swe_rsrc(p.vh), v8, 0);
        if (WD) al[i] = swe_ld(swe_rsrc(p.valpha), v8, 0);
        if (WD) {           // the planes hold D (U(0)'s too); the continuity equation advances zeta = D - h
            H[i] = e[i];
            e[i] = swe_wd_eta(H[i], h[i], al[i]);
        } else H[i] = NONLIN ? h[i] + e[i] : h[i];
    }
    const double ax = px[1] - px[0], ay = py[1] - py[0];
    const double bx = px[3] - px[0], by = py[3] - py[0];
    const double A = fma(ax, by, -(ay*bx));                      // AFFINE: the cell area; else d0 = det J at (0, 0)
    const double rA = swe_rcp(A);
    // A * grad(xi), A * grad(zeta)
    const double xix = by, xiy = -bx, zex = -ay, zey = ax;
    // general quadrilateral: c = p0 - p1 + p2 - p3, det J = A + d1 xi + d2 zeta
    const double cx = AFFINE ? 0.0 : (px[0] - px[1]) + (px[2] - px[3]), cy = AFFINE ? 0.0 : (py[0] - py[1]) + (py[2] - py[3]);
    const double d1 = AFFINE ? 0.0 : fma(ax, cy, -(ay*cx)), d2 = AFFINE ? 0.0 : fma(cx, by, -(cy*bx));

    double bu[4] = {0.0, 0.0, 0.0, 0.0}, bv[4] = {0.0, 0.0, 0.0, 0.0}, be[4] = {0.0, 0.0, 0.0, 0.0};
    // ---- facets first: the 24 neighbour traces (48 VGPRs) are dead before the cell quadrature starts
#pragma unroll
    for (int f = 0; f < 4; f++) {
        const int a = f, b = (f + 1) & 3;
        const double nxs = py[b] - py[a], nys = px[a] - px[b];
        const double len2 = swe_dot2(nxs, nxs, nys, nys);
        double L, rL;
        swe_sqrt_rsqrt(len2, L, rL);
        double Fau = 0.0, Fbu = 0.0, Fav = 0.0, Fbv = 0.0, Fae = 0.0, Fbe = 0.0;
        {   // branch-free (see swe_stage_kernel): a boundary facet carries the cell's own values as traces, its flux is discarded
            // neighbour's nodal depth on this facet: what its planes hold; its elevation by the closed form (bathymetry and alpha
            // are continuous: same vertices)
            const double Dna = WD ? ena[f] : 0.0, Dnb = WD ? enb[f] : 0.0;
            const double ena_ = WD ? swe_wd_eta(Dna, h[a], al[a]) : ena[f], enb_ = WD ? swe_wd_eta(Dnb, h[b], al[b]) : enb[f];
            swe_facet_flux<NONLIN, LF, WD>(g, p.sigma_lf, u[a], u[b], v[a], v[b], e[a], e[b], h[a], h[b], H[a], H[b], una[f], unb[f],
                                           vna[f], vnb[f], ena_, enb_, Dna, Dnb, nxs, nys, L, rL, Fau, Fbu, Fav, Fbv, Fae, Fbe);
        }
        if (nb[f] < 0) { Fau = 0.0; Fbu = 0.0; Fav = 0.0; Fbv = 0.0; Fae = 0.0; Fbe = 0.0; }     // see swe_boundary_epilogue_quad
        bu[a] = fma(-0.5, Fau, bu[a]); bu[b] = fma(-0.5, Fbu, bu[b]);
        bv[a] = fma(-0.5, Fav, bv[a]); bv[b] = fma(-0.5, Fbv, bv[b]);
        be[a] = fma(-0.5, Fae, be[a]); be[b] = fma(-0.5, Fbe, be[b]);
    }

    // U(0) is requested here: the 24 neighbour traces are dead, the cell quadrature hides the trip
    double u0u[4], u0v[4], u0e[4];
    if (HASU0) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if constexpr (TILE) {                          // the lane's own U(0) waits in LDS (swe2d_fuse.h)
                u0u[i] = t0[(size_t)i*S0 + ku];
                u0v[i] = t0[(size_t)(4 + i)*S0 + ku];
                u0e[i] = t0[(size_t)(8 + i)*S0 + ku];
            } else {
                u0u[i] = swe_ld(swe_rsrc(p.u0), k8, i*S8);
                u0v[i] = swe_ld(swe_rsrc(p.u0 + 4*S), k8, i*S8);
                u0e[i] = swe_ld(swe_rsrc(p.u0 + 8*S), k8, i*S8);
            }
        }
    }
    // ---- cell integrals, 2 x 2 Gauss-Legendre; weights A/4, gradients carry 1/A  ->  factor 1/4 on gradient terms
#pragma unroll
    for (int qi = 0; qi < 2; qi++) {
#pragma unroll
        for (int qz = 0; qz < 2; qz++) {
            const double xi = qi ? SWE_XI1 : SWE_XI0, ze = qz ? SWE_XI1 : SWE_XI0;
            const double phi[4] = {(1.0 - xi)*(1.0 - ze), xi*(1.0 - ze), xi*ze, (1.0 - xi)*ze};
            const double dxi[4] = {-(1.0 - ze), (1.0 - ze), ze, -ze};
            const double dze[4] = {-(1.0 - xi), -xi, xi, (1.0 - xi)};
            double gx[4], gy[4];                           // A * grad(phi_i)  (general cell: det J * grad(phi_i) at the point)
            double uq = 0.0, vq = 0.0, eq = 0.0, Hq = 0.0, D = 0.0;
            // adj(J)^T at the point: x_zeta = b + xi c, x_xi = a + zeta c
            const double xix_q = AFFINE ? xix : fma(cy, xi, by), xiy_q = AFFINE ? xiy : -fma(cx, xi, bx);
            const double zex_q = AFFINE ? zex : -fma(cy, ze, ay), zey_q = AFFINE ? zey : fma(cx, ze, ax);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                gx[i] = swe_dot2(dxi[i], xix_q, dze[i], zex_q);
                gy[i] = swe_dot2(dxi[i], xiy_q, dze[i], zey_q);
                uq = fma(phi[i], u[i], uq);
                vq = fma(phi[i], v[i], vq);
                eq = fma(phi[i], e[i], eq);
                Hq = fma(phi[i], H[i], Hq);
                D = fma(gy[i], v[i], fma(gx[i], u[i], D));  // A * div(u)
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                double fu = g*eq*gx[i], fv = g*eq*gy[i];                       // shallowwater_eq.py:361
                if (NONLIN) {                                                  // :478
                    const double adv = fma(vq, gy[i], fma(uq, gx[i], phi[i]*D));
                    fu = fma(adv, uq, fu);
                    fv = fma(adv, vq, fv);
                }
                bu[i] = fma(0.25, fu, bu[i]);
                bv[i] = fma(0.25, fv, bv[i]);
                be[i] = fma(0.25, Hq*swe_dot2(gx[i], uq, gy[i], vq), be[i]);          // :422
            }
        }
    }
    // ---- optional cell-local terms (Coriolis, drag, wind, sources, atmospheric pressure) in a pass of their own over the four
    // quadrature points, rolled: inside the unrolled loop above they cost 286-306 VGPRs, inside a rolled copy of it 197-225 (two
    // waves per SIMD, rounds 1-4); here the main loop is the one of the plain kernel and this pass lives in the registers it
    // leaves (167 with the parallelogram kernel: three waves)
    if (SRC) {
#pragma unroll 1
        for (int q = 0; q < 4; q++) {
            const double xi = (q & 2) ? SWE_XI1 : SWE_XI0, ze = (q & 1) ? SWE_XI1 : SWE_XI0;
            const double phi[4] = {(1.0 - xi)*(1.0 - ze), xi*(1.0 - ze), xi*ze, (1.0 - xi)*ze};
            double uq = 0.0, vq = 0.0, Hq = 0.0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                uq = fma(phi[i], u[i], uq);
                vq = fma(phi[i], v[i], vq);
                Hq = fma(phi[i], H[i], Hq);
            }
            const double Aq = AFFINE ? A : fma(d2, ze, fma(d1, xi, A));          // det J at the point
            double corq = 0.0, gpx = 0.0, gpy = 0.0, sx = 0.0, sy = 0.0, sv = 0.0;
            if (p.patm) {
                const double dxi[4] = {-(1.0 - ze), (1.0 - ze), ze, -ze};
                const double dze[4] = {-(1.0 - xi), -xi, xi, (1.0 - xi)};
                const double xix_q = AFFINE ? xix : fma(cy, xi, by), xiy_q = AFFINE ? xiy : -fma(cx, xi, bx);
                const double zex_q = AFFINE ? zex : -fma(cy, ze, ay), zey_q = AFFINE ? zey : fma(cx, ze, ax);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const double pa = swe_ld(swe_rsrc(p.patm), k8, i*S8);
                    gpx += swe_dot2(dxi[i], xix_q, dze[i], zex_q)*pa;
                    gpy += swe_dot2(dxi[i], xiy_q, dze[i], zey_q)*pa;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (p.coriolis) corq += phi[i]*swe_ld(swe_rsrc(p.coriolis), k8, i*S8);
                if (p.msrc) {
                    sx += phi[i]*swe_ld(swe_rsrc(p.msrc), k8, i*S8);
                    sy += phi[i]*swe_ld(swe_rsrc(p.msrc + 4*S), k8, i*S8);
                }
                if (p.vsrc) sv += phi[i]*swe_ld(swe_rsrc(p.vsrc), k8, i*S8);
                if (p.wind) {
                    sx += phi[i]*swe_ld(swe_rsrc(p.wind), k8, i*S8)/(Hq*1000.0);
                    sy += phi[i]*swe_ld(swe_rsrc(p.wind + 4*S), k8, i*S8)/(Hq*1000.0);
                }
            }
            double drag = 0.0;
            if (p.quad_drag >= 0.0 || p.manning >= 0.0 || p.nikuradse >= 0.0 || p.quad_f) {
                double cq = 0.0;
                if (p.quad_f) {
#pragma unroll
                    for (int i = 0; i < 4; i++) cq += phi[i]*swe_ld(swe_rsrc(p.quad_f), k8, i*S8);
                }
                const int kind = p.quad_f ? p.quad_f_kind : (p.manning >= 0.0 ? 2 : (p.nikuradse >= 0.0 ? 3 : 1));
                const double coef = p.quad_f ? cq : (kind == 2 ? p.manning : (kind == 3 ? p.nikuradse : p.quad_drag));
                double cdh;                                  // C_D / H
                if (kind == 2) {                             // Manning: g mu^2 H^(-4/3) = g mu^2 (H^(-1/3))^4, no reciprocal
                    const double y = swe_rcbrt(Hq), y2 = y*y;
                    cdh = g*coef*coef*(y2*y2);
                } else {
                    double cd = coef;
                    if (kind == 3) {
                        const double lg = log(11.036*Hq/coef);
                        cd = (Hq > coef) ? 0.32/(lg*lg) : 0.0;
                    }
                    cdh = cd*swe_rcp(Hq);
                }
                drag = cdh*swe_sqrt_sumsq(uq*uq + vq*vq + p.norm_smoother*p.norm_smoother);
            }
            if (p.lin_drag_f) {
#pragma unroll
                for (int i = 0; i < 4; i++) drag += phi[i]*swe_ld(swe_rsrc(p.lin_drag_f), k8, i*S8);
            } else if (p.linear_drag >= 0.0) drag += p.linear_drag;
            const double cu = Aq*(corq*vq - drag*uq + sx) - gpx*(1.0/1000.0);
            const double cv_ = Aq*(-corq*uq - drag*vq + sy) - gpy*(1.0/1000.0);
            const double ce = Aq*sv;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                bu[i] = fma(0.25, cu*phi[i], bu[i]);
                bv[i] = fma(0.25, cv_*phi[i], bv[i]);
                be[i] = fma(0.25, ce*phi[i], be[i]);
            }
        }
    }

    // ---- tensor mass inverse and Shu-Osher combine (w = a0*U0 + a1*U_in; with wetting-drying the continuity equation advances
    //      zeta = D - h: the planes hold D, U(0)'s too)
#pragma unroll
    for (int i = 0; i < 4; i++) {
        wu[i] = a1*u[i];
        wv[i] = a1*v[i];
        we[i] = WD ? a1*(H[i] - h[i]) : a1*e[i];
        if (HASU0) {
            wu[i] = fma(a0, u0u[i], wu[i]);
            wv[i] = fma(a0, u0v[i], wv[i]);
            we[i] = fma(a0, WD ? u0e[i] - h[i] : u0e[i], we[i]);
        }
    }
    if constexpr (AFFINE) {
    const double s = p.dt*beta*rA;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int n1 = (i + 1) & 3, n2 = (i + 2) & 3, n3 = (i + 3) & 3;
        ou[i] = fma(s, fma(4.0, bu[n2], fma(-8.0, bu[n3], fma(-8.0, bu[n1], 16.0*bu[i]))), wu[i]);
        ov[i] = fma(s, fma(4.0, bv[n2], fma(-8.0, bv[n3], fma(-8.0, bv[n1], 16.0*bv[i]))), wv[i]);
        oe[i] = fma(s, fma(4.0, be[n2], fma(-8.0, be[n3], fma(-8.0, be[n1], 16.0*be[i]))), we[i]);
    }
    } else {
    SweQuadMass M;
    SweQuadLDL F;
    swe_quad_mass(A, d1, d2, M);
    swe_quad_mass_factor(M, F);
    swe_quad_mass_solve(F, bu);
    swe_quad_mass_solve(F, bv);
    swe_quad_mass_solve(F, be);
    const double s = p.dt*beta;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        ou[i] = fma(s, bu[i], wu[i]);
        ov[i] = fma(s, bv[i], wv[i]);
        oe[i] = fma(s, be[i], we[i]);
    }
    }
    // boundary facets were skipped in the facet loop
    if (bmarkers != 0u) {
        if constexpr (TILE) swe_boundary_epilogue_quad<NONLIN, LF, WD, AFFINE>(p, k, bmarkers, ou, ov, oe, tin, Su, ku, beta);
        else swe_boundary_epilogue_quad<NONLIN, LF, WD, AFFINE>(p, k, bmarkers, ou, ov, oe, p.uin, p.stride, k, beta);
    }
    if (WD && !(a0 == 0.0 && a1 == 0.0)) {
        if constexpr (AFFINE) swe_wd_finish<4>(p.g, beta*p.dt, h, al, ou, ov, oe, !p.wd_skip_relax);
        else {
            double mw[4];
            swe_quad_mean_weights(A, d1, d2, mw);
            swe_wd_finish<4>(p.g, beta*p.dt, h, al, ou, ov, oe, !p.wd_skip_relax, mw);
        }
    }
}

// AFFINE = false: general quadrilaterals (see above)
template <bool NONLIN, bool LF, bool HASU0, bool SRC, bool WD, bool AFFINE = true>
__global__ __launch_bounds__(SWE_BLOCK) __attribute__((amdgpu_waves_per_eu(3))) void swe_stage_kernel_quad(const SweStageArgs p)
{
    // no implicit contraction (see swe_stage_kernel)
#pragma clang fp contract(off)
#ifdef SWE_NO_XCD_MAP
    int lb = blockIdx.x;
#else
    int lb = swe_logical_block(blockIdx.x, gridDim.x);
#endif
    if (p.reverse) {              // see swe_stage_kernel: launches beyond the Infinity Cache alternate their direction
        lb = (p.cell_end - p.cell_begin + SWE_BLOCK - 1)/SWE_BLOCK - 1 - lb;
        if (lb < 0) return;
    }
    const int k = p.cell_begin + lb*SWE_BLOCK + (int)threadIdx.x;
    if (k >= p.cell_end) return;
    const size_t S = p.stride;
    // raw buffer addressing (see swe_ld): one resource per group of four planes, 4*stride*8 < 2^32 checked at create
    const unsigned S8 = (unsigned)S*8u, k8 = (unsigned)k*8u, S4 = (unsigned)S*4u, k4 = (unsigned)k*4u;
    const swe_rsrc_t gu = swe_rsrc(p.uin), gv = swe_rsrc(p.uin + 4*S), ge = swe_rsrc(p.uin + 8*S);

    double u[4], v[4], e[4];
    int nb[4], vid[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        nb[i] = swe_ldi(swe_rsrc(p.nbr), k4, i*S4);
        vid[i] = swe_ldi(swe_rsrc(p.cv), k4, i*S4);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u[i] = swe_ld(gu, k8, i*S8);
        v[i] = swe_ld(gv, k8, i*S8);
        e[i] = swe_ld(ge, k8, i*S8);
    }
    // boundary markers of the four facets in one register (0: interior facet): all the boundary pass at the end needs of nb[]
    const unsigned bmarkers = (nb[0] < 0 ? (unsigned)(-nb[0]) : 0u) | (nb[1] < 0 ? (unsigned)(-nb[1]) << 8 : 0u) |
                              (nb[2] < 0 ? (unsigned)(-nb[2]) << 16 : 0u) | (nb[3] < 0 ? (unsigned)(-nb[3]) << 24 : 0u);
    // (U(0) of the velocity and of eta is NOT requested here: its twelve values would be live through the facet loop - 24 of the
    //  registers that kept this kernel at two waves per SIMD; see below)
    double una[4], unb[4], vna[4], vnb[4], ena[4], enb[4];
    {   // (an LDS exchange of the in-wave traces as in swe_stage_kernel<..., LDSX> was measured: 201 vs 203 us/step at 1 M
        //  quadrilaterals - this kernel is bound by its 198 VGPRs and its arithmetic, not by the gathers; not kept)
#pragma unroll
        for (int f = 0; f < 4; f++) {
            const int nbf = nb[f];
            const int kn = nbf >= 0 ? (nbf >> 2) : k;
            const int f2 = nbf >= 0 ? (nbf & 3) : f;
            const int na = (f2 + 1) & 3;
            const unsigned kn8 = (unsigned)kn*8u;
            const unsigned ob = kn8 + ((f2 & 1) ? S8 : 0u) + ((f2 & 2) ? 2u*S8 : 0u);
            const unsigned oa = kn8 + ((na & 1) ? S8 : 0u) + ((na & 2) ? 2u*S8 : 0u);
            una[f] = swe_ld(gu, oa, 0);
            unb[f] = swe_ld(gu, ob, 0);
            vna[f] = swe_ld(gv, oa, 0);
            vnb[f] = swe_ld(gv, ob, 0);
            ena[f] = swe_ld(ge, oa, 0);
            enb[f] = swe_ld(ge, ob, 0);
        }
    }
    double ou[4], ov[4], oe[4];
    swe_quad_stage_cell<NONLIN, LF, HASU0, SRC, WD, AFFINE, false>(p, k, k8, S8, nb, vid, bmarkers, u, v, e, una, unb, vna, vnb, ena, enb, p.a0,
                                                                  p.a1, p.beta, p.uin, S, k, nullptr, 0, ou, ov, oe);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        swe_st(swe_rsrc(p.uout), k8, i*S8, ou[i]);
        swe_st(swe_rsrc(p.uout + 4*S), k8, i*S8, ov[i]);
        swe_st(swe_rsrc(p.uout + 8*S), k8, i*S8, oe[i]);
    }
}

// quad diagnostics: int a*b over a parallelogram = A/36 * a^T K b, K = [[4,2,1,2],[2,4,2,1],[1,2,4,2],[2,1,2,4]]
__device__ __forceinline__ double swe_int2_quad(const double a[4], const double b[4])
{
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 4; i++)
        s += a[i]*(4.0*b[i] + 2.0*b[(i + 1) & 3] + 2.0*b[(i + 3) & 3] + b[(i + 2) & 3]);
    return s;
}

// a^T M b for the general quadrilateral mass matrix
__device__ __forceinline__ double swe_quad_form(const SweQuadMass &M, const double a[4], const double b[4])
{
    const double *m = M.m;               // 0:00 1:01 2:02 3:03 4:11 5:12 6:13 7:22 8:23 9:33
    return m[0]*a[0]*b[0] + m[4]*a[1]*b[1] + m[7]*a[2]*b[2] + m[9]*a[3]*b[3]
         + m[1]*(a[0]*b[1] + a[1]*b[0]) + m[2]*(a[0]*b[2] + a[2]*b[0]) + m[3]*(a[0]*b[3] + a[3]*b[0])
         + m[5]*(a[1]*b[2] + a[2]*b[1]) + m[6]*(a[1]*b[3] + a[3]*b[1]) + m[8]*(a[2]*b[3] + a[3]*b[2]);
}

static __global__ __launch_bounds__(SWE_BLOCK) void swe_diag_kernel_quad(const double *planes, size_t stride, const int *cv,
                                                                  const double *vx, const double *vy, const double *vh,
                                                                  int n, double *partial, const double *valpha, int affine,
                                                                  unsigned long long *acc)
{
    const int k = blockIdx.x*SWE_BLOCK + threadIdx.x;
    double s_e2 = 0.0, s_u2 = 0.0, s_vol = 0.0, s_min = 1e300;
    if (k < n) {
        double u[4], v[4], e[4], px[4], py[4], h[4];
        for (int i = 0; i < 4; i++) {
            u[i] = planes[(size_t)i*stride + k];
            v[i] = planes[(size_t)(4 + i)*stride + k];
            e[i] = planes[(size_t)(8 + i)*stride + k];
            const int vid = cv[(size_t)i*stride + k];
            px[i] = vx[vid]; py[i] = vy[vid]; h[i] = vh[vid];
            if (valpha) { const double D = e[i]; e[i] = D - 0.25*valpha[vid]*valpha[vid]/D - h[i]; h[i] = D; }   // the planes hold D
            else h[i] = h[i] + e[i];                                                   // nodal total depth
        }
        const double A = (px[1] - px[0])*(py[3] - py[0]) - (py[1] - py[0])*(px[3] - px[0]);
        if (affine) {
            s_e2 = A*(1.0/36.0)*swe_int2_quad(e, e);
            s_u2 = A*(1.0/36.0)*(swe_int2_quad(u, u) + swe_int2_quad(v, v));
            s_vol = A*0.25*(h[0] + h[1] + h[2] + h[3]);
        } else {
            const double cx = (px[0] - px[1]) + (px[2] - px[3]), cy = (py[0] - py[1]) + (py[2] - py[3]);
            const double d1 = (px[1] - px[0])*cy - (py[1] - py[0])*cx, d2 = cx*(py[3] - py[0]) - cy*(px[3] - px[0]);
            SweQuadMass M;
            swe_quad_mass(A, d1, d2, M);
            s_e2 = swe_quad_form(M, e, e);
            s_u2 = swe_quad_form(M, u, u) + swe_quad_form(M, v, v);
            double w[4];
            swe_quad_mean_weights(A, d1, d2, w);
            s_vol = (A + 0.5*(d1 + d2))*(w[0]*h[0] + w[1]*h[1] + w[2]*h[2] + w[3]*h[3]);
        }
        s_min = fmin(fmin(h[0], h[1]), fmin(h[2], h[3]));
    }
    acc += (size_t)(blockIdx.x & (SWE_DIAG_BUCKETS - 1))*SWE_DIAG_ACC;
    swe_sum_accumulate(s_e2, acc, acc + 3*SWE_SUM_LIMBS);
    swe_sum_accumulate(s_u2, acc + SWE_SUM_LIMBS, acc + 3*SWE_SUM_LIMBS);
    swe_sum_accumulate(s_vol, acc + 2*SWE_SUM_LIMBS, acc + 3*SWE_SUM_LIMBS);
    s_min = swe_wave_min(s_min);
    if (threadIdx.x == 0) partial[blockIdx.x] = s_min;
}

// Boundary facets of a quadrilateral whose marker carries a boundary value or an external velocity (tracer_eq_2d.py:177-188, :380-393),
// evaluated AFTER the cell's outputs are finished: inputs reloaded by plane index, the facet's two node integrals through the mass
// inverse into the outputs (see swe_boundary_epilogue_quad).  bdefer: bit f = facet f is such a facet.
template <bool AFFINE>
__device__ __forceinline__ void swe_tracer_boundary_epilogue_quad(const SweTracerArgs &p, int k, unsigned bdefer, double o[4])
{
    const size_t S = p.stride;
    const double cf = p.vel_factor;
    int vid[4];
    double px[4], py[4];
    for (int i = 0; i < 4; i++) { vid[i] = p.cv[(size_t)i*S + k]; px[i] = p.vx[vid[i]]; py[i] = p.vy[vid[i]]; }
    const double ax = px[1] - px[0], ay = py[1] - py[0], bx = px[3] - px[0], by = py[3] - py[0];
    const double A = ax*by - ay*bx;
    const double cx = AFFINE ? 0.0 : (px[0] - px[1]) + (px[2] - px[3]), cy = AFFINE ? 0.0 : (py[0] - py[1]) + (py[2] - py[3]);
    const double d1 = AFFINE ? 0.0 : ax*cy - ay*cx, d2 = AFFINE ? 0.0 : cx*by - cy*bx;
    SweQuadLDL F;
    if constexpr (!AFFINE) {
        SweQuadMass M;
        swe_quad_mass(A, d1, d2, M);
        swe_quad_mass_factor(M, F);
    }
    const double s = AFFINE ? p.dt*p.beta*swe_rcp(A) : p.dt*p.beta;
#pragma unroll 1
    for (int f = 0; f < 4; f++) {
        if (!((bdefer >> f) & 1u)) continue;
        const int a = f, bb = (f + 1) & 3;
        const int marker = -p.nbr[(size_t)f*S + k];
        const double ua = cf*p.uv[(size_t)a*S + k], ub = cf*p.uv[(size_t)bb*S + k];
        const double va = cf*p.uv[(size_t)(4 + a)*S + k], vb = cf*p.uv[(size_t)(4 + bb)*S + k];
        const double ca = p.tin[(size_t)a*S + k], cb = p.tin[(size_t)bb*S + k];
        const double nxs = py[bb] - py[a], nys = px[a] - px[bb];
        double Fa = 0.0, Fb = 0.0;
#pragma unroll 1
        for (int q = 0; q < 2; q++) {
            const double xb = q ? SWE_XI1 : SWE_XI0, xa = 1.0 - xb;
            const double uq = xa*ua + xb*ub, vq = xa*va + xb*vb, cq = xa*ca + xb*cb;
            const double cext = (p.bc_has_value[marker] == 2)
                ? xa*p.bc_value_f[(size_t)(4*f + a)*S + k] + xb*p.bc_value_f[(size_t)(4*f + bb)*S + k]
                : (p.bc_has_value[marker] ? p.bc_value[marker] : cq);
            double hq = 0.0, eq = 0.0, alq = 0.0;
            if (p.bc_vel_kind[marker] >= 3) {                  // 'flux': total depth at the quadrature point
                const double ha = p.vh[vid[a]], hb = p.vh[vid[bb]];
                hq = xa*ha + xb*hb;
                double ea_ = p.uv[(size_t)(8 + a)*S + k], eb_ = p.uv[(size_t)(8 + bb)*S + k];
                if (p.depth_mode == 2) {           // the planes hold D: the nodal elevations by the closed form
                    const double aa_ = p.valpha[vid[a]], ab_ = p.valpha[vid[bb]];
                    alq = xa*aa_ + xb*ab_;
                    ea_ = ea_ - 0.25*aa_*aa_/ea_ - ha;
                    eb_ = eb_ - 0.25*ab_*ab_/eb_ - hb;
                }
                eq = xa*ea_ + xb*eb_;
            }
            const double fq = swe_tracer_boundary_flux(p, marker, cq, cext, uq, vq, nxs, nys, hq, eq, alq, xa, xb, k, f, 4, S);
            Fa += xa*fq;
            Fb += xb*fq;
        }
        const double da = -0.5*Fa, db = -0.5*Fb;
        if constexpr (AFFINE) {
#pragma unroll
            for (int i = 0; i < 4; i++) {            // row i of the tensor mass inverse (16, -8, 4, -8)/A on (da at a, db at bb)
                const double wa = (i == a) ? 16.0 : (i == (a ^ 2) ? 4.0 : -8.0), wb = (i == bb) ? 16.0 : (i == (bb ^ 2) ? 4.0 : -8.0);
                o[i] += s*(wa*da + wb*db);
            }
        } else {
            double r[4];
#pragma unroll
            for (int i = 0; i < 4; i++) r[i] = (i == a) ? da : ((i == bb) ? db : 0.0);
            swe_quad_mass_solve(F, r);
#pragma unroll
            for (int i = 0; i < 4; i++) o[i] += s*r[i];
        }
    }
}

// ---- tracer stage on parallelogram quadrilaterals (see swe_tracer_stage_kernel and swe_stage_kernel_quad)
template <bool LF, bool HAST0, bool SRC, bool AFFINE = true>
__global__ __launch_bounds__(SWE_BLOCK) void swe_tracer_stage_kernel_quad(const SweTracerArgs p)
{
#ifdef SWE_NO_XCD_MAP
    const int lb = blockIdx.x;
#else
    const int lb = swe_logical_block(blockIdx.x, gridDim.x);
#endif
    const int k = p.cell_begin + lb*SWE_BLOCK + (int)threadIdx.x;
    if (k >= p.cell_end) return;
    const size_t S = p.stride;
    const unsigned S8 = (unsigned)S*8u, k8 = (unsigned)k*8u, S4 = (unsigned)S*4u, k4 = (unsigned)k*4u;    // see swe_ld
    const swe_rsrc_t gu = swe_rsrc(p.uv), gv = swe_rsrc(p.uv + 4*S), gt = swe_rsrc(p.tin);
    const double cf = p.vel_factor;

    double u[4], v[4], c[4], w[4];
    int nb[4], vid[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        nb[i] = swe_ldi(swe_rsrc(p.nbr), k4, i*S4);
        vid[i] = swe_ldi(swe_rsrc(p.cv), k4, i*S4);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u[i] = cf*swe_ld(gu, k8, i*S8);
        v[i] = cf*swe_ld(gv, k8, i*S8);
        c[i] = swe_ld(gt, k8, i*S8);
        w[i] = p.a1*c[i];
        if (HAST0) w[i] += p.a0*swe_ld(swe_rsrc(p.t0), k8, i*S8);
    }
    // boundary facets whose marker carries a boundary value or an external velocity: bit f (the tables are kernel arguments: an
    // entry indexed per lane is a load, issued here with all the others)
    unsigned bdefer = 0u;
    if ((nb[0] | nb[1] | nb[2] | nb[3]) < 0) {      // (read by the lanes - and waves - that own a boundary facet only)
#pragma unroll
        for (int f = 0; f < 4; f++) {
            const int marker = nb[f] < 0 ? -nb[f] : 0;
            if (marker > 0 && marker < SWE_MAX_MARKERS && (p.bc_has_value[marker] || p.bc_vel_kind[marker])) bdefer |= 1u << f;
        }
    }
    double una[4], unb[4], vna[4], vnb[4], cna[4], cnb[4];
#pragma unroll
    for (int f = 0; f < 4; f++) {
        const int nbf = nb[f];
        const int kn = nbf >= 0 ? (nbf >> 2) : k;
        const int f2 = nbf >= 0 ? (nbf & 3) : f;
        const int na = (f2 + 1) & 3;
        const unsigned kn8 = (unsigned)kn*8u;
        const unsigned ob = kn8 + ((f2 & 1) ? S8 : 0u) + ((f2 & 2) ? 2u*S8 : 0u);
        const unsigned oa = kn8 + ((na & 1) ? S8 : 0u) + ((na & 2) ? 2u*S8 : 0u);
        una[f] = cf*swe_ld(gu, oa, 0);
        unb[f] = cf*swe_ld(gu, ob, 0);
        vna[f] = cf*swe_ld(gv, oa, 0);
        vnb[f] = cf*swe_ld(gv, ob, 0);
        cna[f] = swe_ld(gt, oa, 0);
        cnb[f] = swe_ld(gt, ob, 0);
    }
    double px[4], py[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        px[i] = swe_ld(swe_rsrc(p.vx), (unsigned)vid[i]*8u, 0);
        py[i] = swe_ld(swe_rsrc(p.vy), (unsigned)vid[i]*8u, 0);
    }
    const double ax = px[1] - px[0], ay = py[1] - py[0];
    const double bx = px[3] - px[0], by = py[3] - py[0];
    const double A = ax*by - ay*bx;
    const double xix = by, xiy = -bx, zex = -ay, zey = ax;       // A*grad(xi), A*grad(zeta)
    // general quadrilateral (see swe_quad_mass): c = p0 - p1 + p2 - p3, det J = A + d1 xi + d2 zeta
    const double cx = AFFINE ? 0.0 : (px[0] - px[1]) + (px[2] - px[3]), cy = AFFINE ? 0.0 : (py[0] - py[1]) + (py[2] - py[3]);
    const double d1 = AFFINE ? 0.0 : ax*cy - ay*cx, d2 = AFFINE ? 0.0 : cx*by - cy*bx;
    double b[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int qi = 0; qi < 2; qi++) {
#pragma unroll
        for (int qz = 0; qz < 2; qz++) {
            const double xi = qi ? SWE_XI1 : SWE_XI0, ze = qz ? SWE_XI1 : SWE_XI0;
            const double phi[4] = {(1.0 - xi)*(1.0 - ze), xi*(1.0 - ze), xi*ze, (1.0 - xi)*ze};
            const double dxi[4] = {-(1.0 - ze), (1.0 - ze), ze, -ze};
            const double dze[4] = {-(1.0 - xi), -xi, xi, (1.0 - xi)};
            double gx[4], gy[4], uq = 0.0, vq = 0.0, cq = 0.0, D = 0.0, sq = 0.0;
            const double xix_q = AFFINE ? xix : by + cy*xi, xiy_q = AFFINE ? xiy : -(bx + cx*xi);
            const double zex_q = AFFINE ? zex : -(ay + cy*ze), zey_q = AFFINE ? zey : ax + cx*ze;
            const double Aq = AFFINE ? A : A + d1*xi + d2*ze;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                gx[i] = dxi[i]*xix_q + dze[i]*zex_q;
                gy[i] = dxi[i]*xiy_q + dze[i]*zey_q;
                uq += phi[i]*u[i];
                vq += phi[i]*v[i];
                cq += phi[i]*c[i];
                D += gx[i]*u[i] + gy[i]*v[i];
                if (SRC) sq += phi[i]*swe_ld(swe_rsrc(p.source), k8, i*S8);
            }
            if (SRC && p.conservative) {                                               // H*source, :434-436
                double Hq = 0.0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const double hh = p.vh[vid[i]];
                    const double ee = swe_ld(swe_rsrc(p.uv + 8*S), k8, i*S8);
                    Hq += phi[i]*(p.depth_mode == 2 ? ee : (p.depth_mode == 1 ? hh + ee : hh));      // wetting-drying: the planes hold D
                }
                sq *= Hq;
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
                b[i] += 0.25*(((p.conservative ? 0.0 : phi[i]*D) + uq*gx[i] + vq*gy[i])*cq + Aq*sq*phi[i]);
        }
    }
#pragma unroll
    for (int f = 0; f < 4; f++) {
        const int a = f, bb = (f + 1) & 3;
        const double nxs = py[bb] - py[a], nys = px[a] - px[bb];
        double Fa = 0.0, Fb = 0.0;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const double xb = q ? SWE_XI1 : SWE_XI0, xa = 1.0 - xb;
            const double uq = xa*u[a] + xb*u[bb], vq = xa*v[a] + xb*v[bb], cq = xa*c[a] + xb*c[bb];
            const double unown = uq*nxs + vq*nys;
            double fq;
            if (nb[f] >= 0) {
                const double un = xa*una[f] + xb*unb[f], vn = xa*vna[f] + xb*vnb[f], cn = xa*cna[f] + xb*cnb[f];
                const double uavn = 0.5*((uq + un)*nxs + (vq + vn)*nys);
                const double cup = uavn > 0.0 ? cq : (uavn < 0.0 ? cn : 0.5*(cq + cn));
                fq = cup*unown;
                if (p.conservative) {
                    const double fn = cn*(un*nxs + vn*nys);
                    fq = uavn > 0.0 ? cq*unown : (uavn < 0.0 ? fn : 0.5*(cq*unown + fn));
                }
                if (LF) fq += 0.5*fabs(uavn)*p.lf_factor*(cq - cn);
            } else {
                // boundary facet: with a boundary value or an external velocity it is evaluated AFTER the outputs are finished
                // (swe_tracer_boundary_epilogue_quad: four inlined copies of that code cost every wave ~30 VGPRs and the third wave
                // per SIMD); here only the default, the interior state on both sides (tracer_eq_2d.py:189-191)
                fq = (bdefer >> f) & 1u ? 0.0 : cq*unown;
            }
            Fa += xa*fq;
            Fb += xb*fq;
        }
        b[a] -= 0.5*Fa;
        b[bb] -= 0.5*Fb;
    }
    if constexpr (AFFINE) {
    const double s = p.dt*p.beta*swe_rcp(A);
    double msum = 0.0, o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) o[i] = s*(16.0*b[i] - 8.0*b[(i + 1) & 3] - 8.0*b[(i + 3) & 3] + 4.0*b[(i + 2) & 3]) + w[i];
    if (bdefer) swe_tracer_boundary_epilogue_quad<true>(p, k, bdefer, o);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        swe_st(swe_rsrc(p.tout), k8, i*S8, o[i]);
        msum += o[i];
    }
    if (p.mean_out) p.mean_out[k] = msum/4.0;
    } else {
    SweQuadMass M;
    SweQuadLDL F;
    swe_quad_mass(A, d1, d2, M);
    swe_quad_mass_factor(M, F);
    swe_quad_mass_solve(F, b);
    double mw[4], msum = 0.0, o[4];
    swe_quad_mean_weights(A, d1, d2, mw);
    const double s = p.dt*p.beta;
#pragma unroll
    for (int i = 0; i < 4; i++) o[i] = s*b[i] + w[i];
    if (bdefer) swe_tracer_boundary_epilogue_quad<false>(p, k, bdefer, o);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        swe_st(swe_rsrc(p.tout), k8, i*S8, o[i]);
        msum += mw[i]*o[i];
    }
    if (p.mean_out) p.mean_out[k] = msum;         // P0 projection: int o dx / area
    }
}

// tracer diagnostics on quadrilaterals
static __global__ __launch_bounds__(SWE_BLOCK) void swe_tracer_diag_kernel_quad(const double *t, const double *state, size_t stride,
                                                                         const int *cv, const double *vx, const double *vy,
                                                                         const double *vh, int nonlinear, int n, double *partial,
                                                                         const double *valpha, int affine, unsigned long long *acc)
{
    const int k = blockIdx.x*SWE_BLOCK + threadIdx.x;
    double s_m = 0.0, s_i = 0.0, s_min = 1e300, s_max = -1e300;
    if (k < n) {
        double c[4], H[4], px[4], py[4];
        for (int i = 0; i < 4; i++) {
            c[i] = t[(size_t)i*stride + k];
            const int vid = cv[(size_t)i*stride + k];
            px[i] = vx[vid]; py[i] = vy[vid];
            H[i] = vh[vid] + (nonlinear ? state[(size_t)(8 + i)*stride + k] : 0.0);
            if (valpha) H[i] = state[(size_t)(8 + i)*stride + k];      // wetting-drying: the displaced depth D the planes hold
        }
        const double A = (px[1] - px[0])*(py[3] - py[0]) - (py[1] - py[0])*(px[3] - px[0]);
        if (affine) {
            s_m = A*(1.0/36.0)*swe_int2_quad(c, H);
            s_i = A*0.25*(c[0] + c[1] + c[2] + c[3]);
        } else {
            const double cx = (px[0] - px[1]) + (px[2] - px[3]), cy = (py[0] - py[1]) + (py[2] - py[3]);
            const double d1 = (px[1] - px[0])*cy - (py[1] - py[0])*cx, d2 = cx*(py[3] - py[0]) - cy*(px[3] - px[0]);
            SweQuadMass M;
            swe_quad_mass(A, d1, d2, M);
            s_m = swe_quad_form(M, c, H);
            double w[4];
            swe_quad_mean_weights(A, d1, d2, w);
            s_i = (A + 0.5*(d1 + d2))*(w[0]*c[0] + w[1]*c[1] + w[2]*c[2] + w[3]*c[3]);
        }
        s_min = fmin(fmin(c[0], c[1]), fmin(c[2], c[3]));
        s_max = fmax(fmax(c[0], c[1]), fmax(c[2], c[3]));
    }
    acc += (size_t)(blockIdx.x & (SWE_DIAG_BUCKETS - 1))*SWE_DIAG_ACC;
    swe_sum_accumulate(s_m, acc, acc + 2*SWE_SUM_LIMBS);
    swe_sum_accumulate(s_i, acc + SWE_SUM_LIMBS, acc + 2*SWE_SUM_LIMBS);
    s_min = swe_wave_min(s_min);
    s_max = swe_wave_max(s_max);
    if (threadIdx.x == 0) { partial[2*(size_t)blockIdx.x] = s_min; partial[2*(size_t)blockIdx.x + 1] = s_max; }
}
