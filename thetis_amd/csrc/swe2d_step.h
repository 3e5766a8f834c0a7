// swe2d_step.h - a whole SSPRK33 step of the triangle DG-P1 shallow water equations in ONE launch (gfx950).
//
// The per-stage kernel (swe_stage_kernel, swe2d_kernels.h) moves 228 B per cell and stage through HBM and pays one kernel
// boundary per stage: at a million cells it runs at 0.7 of the HBM roofline, at the 125 k cells of one rank of eight it is
// latency-bound (three launches of ~9.5 us).  Both limits come from the stage-by-stage structure of
// thetis/rungekutta.py:316-339 (solve_stage i, then solve_stage i + 1 on the whole mesh), not from the arithmetic.
// This kernel keeps the stages of a TILE of cells on the CU:
//
//   * a workgroup owns a tile = C consecutive cells of the device numbering (compact in space: tile-Hilbert order) plus
//     the three rings of facet-neighbours around them (host-built slot lists: core | ring 1 | ring 2 | ring 3);
//   * every slot's lane loads its cell's U(0) (and ring 3, which is only ever read, rides along on the first lanes) and
//     publishes it in LDS: from here on no stage touches memory for a neighbour;
//   * stage 1 runs on core + ring 1 + ring 2, stage 2 on core + ring 1, stage 3 on the core cells: the six neighbour traces of a
//     facet are read from LDS right where the facet's flux is formed (six live values instead of the eighteen in-flight gathers
//     of the stage kernel), a stage's results replace the tile's values in LDS between two barriers, the last stage stores the
//     step result (into the OTHER state buffer: neighbouring tiles still read the step input; the host swaps the buffer
//     pointers after the step);
//   * a lane keeps its cell's geometry in registers across the three stages; ring lanes retire as soon as their last value is
//     published (whole waves leave the workgroup's barrier set).
//
// Per cell and STEP memory sees: the step input once (72 B, the ring reads hit the L2), the tile lists (~35 B), the result once
// (72 B) - about a third of three per-stage launches; the price is the redundant ring work ((3C + 2 r1 + r2)/(3C) = 1.3 for
// 128-cell tiles, r1 ~ 38, r2 ~ 41 on the bench mesh; whole waves: 9 wave-stages where 6 are the minimum).  168 VGPRs, two of
// them spilled = three waves per SIMD = three 256-lane workgroups per compute unit - but only because the stage loop hides its
// invariants from the optimiser (see the asm statement there): hoisted out of the loop they cost 228 VGPRs or 51 spills.
// Measured (MI355X, same box, us/step, three stage launches -> one step launch): 2.5 k cells 17.3 -> 14.0, 10 k 16.8 -> 14.1,
// 31 k 18.0 -> 13.3, 62 k 20.3 -> 17.0, 90 k 24.1 -> 22.2, 125 k 24.3 -> 24.9, 250 k 38.2 -> 42.4, 1 M 117 -> 131 (256-cell tiles
// in 384-lane workgroups: 153 - two six-wave workgroups do not share a compute unit's SIMDs evenly).  The step kernel wins where a
// step is latency - one launch, one trip to memory, three short stages - and loses beyond ~100 k cells, where throughput counts:
// there it is bound by the FP64 issue rate (1.95 us of SIMD time per wave-stage at 3 waves per SIMD against 1.1 us of pure
// issue), and the 1.3x ring work costs more than the HBM traffic it saves.  swe2d_advance takes it for meshes of at most 80 k cells.
//
// The arithmetic is that of swe_stage_kernel<NONLIN, LF, ., SRC, false, false, true(BINL)>, operation for operation and under
// the same `fp contract(off)`: the step result is bit for bit the one of three stage launches (tests/test_gpu_step_kernel.py).
#pragma once
#include "swe2d_kernels.h"

#ifndef SWE_STEP_OCCUPANCY
#define SWE_STEP_OCCUPANCY __attribute__((amdgpu_waves_per_eu(3, 3)))      // 168 VGPRs: three 256-lane workgroups per CU
#endif
#define SWE_STEP_NO_SLOT 0x3ffu

#define SWE_STEP_EXTRA 128         // slots beyond the workgroup size: ring-3 cells (loaded and published by the first lanes)

struct SweStepArgs {
    SweStageArgs st;               // uin: step input (state buffer 0), uout: step result (state buffer 1); a0/a1/beta unused
    const int4 *tile_slot;         // [n_tiles][B + SWE_STEP_EXTRA] per slot {device cell, neighbour slots (3 x 10 bits), markers | facing
                                   //  facets << 24, vertex 0}; slots >= n2 (ring 3): only .x is used
    const int2 *tile_vert;         // [n_tiles][B] vertices 1, 2 of the slot's cell
    const int4 *tile_n;            // per tile {n_core, + n_ring1, + n_ring2, + n_ring3}
    int n_tiles, B;
    double a0[3], a1[3], beta[3];  // Shu-Osher weights per stage (swe2d_ssprk33_coefficients)
};

// right-hand side integrals of one cell: cell integrals + interior facet fluxes (boundary facets contribute zero here).
// xs: the tile's stage values in LDS, plane stride XS; lnb: the neighbours' slots; meta: boundary markers | the neighbours' facing facets << 24.
// The neighbour traverses the shared facet backwards: its node (f2 + 1) % 3 sits on my node f and its node f2 on my node
// f + 1; a boundary facet reads this cell's own slot (finite values, flux discarded) like the stage kernel reads its own cell.
template <bool NONLIN, bool LF, bool SRC>
__device__ __forceinline__ void swe_step_rhs(const SweStageArgs &p, int k, const double u[3], const double v[3], const double e[3],
                                             const double h[3], const double *xs, int XS, int j, unsigned lnb, int meta,
                                             const double nx[3], const double ny[3], double twoA, double bu[3], double bv[3],
                                             double be[3])
{
#pragma clang fp contract(off)
    const double g = p.g;
    double H[3], gxs[3], gys[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        H[i] = NONLIN ? h[i] + e[i] : h[i];
        gxs[i] = -0.5*nx[(i + 1) % 3];                     // A*grad(phi_i) = -nF_{i+1}/2
        gys[i] = -0.5*ny[(i + 1) % 3];
    }
    {
        const double ge3 = g*(e[0] + e[1] + e[2])*(1.0/3.0);
        const double SHu = swe_int2(H, u)*(1.0/12.0), SHv = swe_int2(H, v)*(1.0/12.0);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            bu[i] = gxs[i]*ge3;
            bv[i] = gys[i]*ge3;
            be[i] = swe_dot2(gxs[i], SHu, gys[i], SHv);
        }
        if (NONLIN) {
            const double Suu = swe_int2(u, u)*(1.0/12.0), Suv = swe_int2(u, v)*(1.0/12.0), Svv = swe_int2(v, v)*(1.0/12.0);
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
#pragma unroll
    for (int f = 0; f < 3; f++) {
        const int a = f, b = (f + 1) % 3;
        const bool bnd = ((meta >> (8*f)) & 0xff) != 0;
        const int ls = bnd ? j : (int)((lnb >> (10*f)) & 0x3ffu);
        const int nb_ = (meta >> (24 + 2*f)) & 3, na_ = nb_ == 2 ? 0 : nb_ + 1;
        const double una = xs[na_*XS + ls], unb = xs[nb_*XS + ls];
        const double vna = xs[(3 + na_)*XS + ls], vnb = xs[(3 + nb_)*XS + ls];
        const double ena = xs[(6 + na_)*XS + ls], enb = xs[(6 + nb_)*XS + ls];
        const double nxs = nx[f], nys = ny[f];
        double Lf, rLf;
        swe_sqrt_rsqrt(swe_dot2(nxs, nxs, nys, nys), Lf, rLf);
        double Fau = 0.0, Fbu = 0.0, Fav = 0.0, Fbv = 0.0, Fae = 0.0, Fbe = 0.0;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const double xb = q ? SWE_XI1 : SWE_XI0, xa = 1.0 - xb;
            const double uq = swe_dot2(xa, u[a], xb, u[b]), vq = swe_dot2(xa, v[a], xb, v[b]), eq = swe_dot2(xa, e[a], xb, e[b]);
            const double hq = swe_dot2(xa, h[a], xb, h[b]);
            const double un = swe_dot2(xa, una, xb, unb), vn = swe_dot2(xa, vna, xb, vnb), en = swe_dot2(xa, ena, xb, enb);
            const double eav = 0.5*(eq + en);
            const double Hav = NONLIN ? hq + eav : hq;
            const double c = swe_sqrt(g*Hav);
            const double du = uq - un, dv = vq - vn;
            const double dun = swe_dot2(du, nxs, dv, nys);
            const double spg = fma(c*dun, rLf, g*eav);
            double fu = spg*nxs, fv = spg*nys;
            const double uav = 0.5*(uq + un), vav = 0.5*(vq + vn);
            const double uavn = swe_dot2(uav, nxs, vav, nys);
            const double fe = fma(c*(eq - en), Lf, Hav*uavn);
            if (NONLIN) {
                const double unown = swe_dot2(uq, nxs, vq, nys);
                fu = fma(uav, unown, fu);
                fv = fma(vav, unown, fv);
                if (LF) {
                    const double gam = 0.5*fabs(uavn)*p.sigma_lf;
                    fu = fma(gam, du, fu);
                    fv = fma(gam, dv, fv);
                }
            }
            Fau = fma(xa, fu, Fau); Fbu = fma(xb, fu, Fbu);
            Fav = fma(xa, fv, Fav); Fbv = fma(xb, fv, Fbv);
            Fae = fma(xa, fe, Fae); Fbe = fma(xb, fe, Fbe);
        }
        if (bnd) { Fau = 0.0; Fbu = 0.0; Fav = 0.0; Fbv = 0.0; Fae = 0.0; Fbe = 0.0; }
        bu[a] = fma(-0.5, Fau, bu[a]); bu[b] = fma(-0.5, Fbu, bu[b]);
        bv[a] = fma(-0.5, Fav, bv[a]); bv[b] = fma(-0.5, Fbv, bv[b]);
        be[a] = fma(-0.5, Fae, be[a]); be[b] = fma(-0.5, Fbe, be[b]);
    }
    if (SRC) swe_source_terms(p, k, p.stride, twoA, u, v, H, gxs, gys, bu, bv, be);
}

// mass inverse, Shu-Osher combine and the boundary facets of the cell (the BINL pass of swe_stage_kernel)
template <bool NONLIN, bool LF>
__device__ __forceinline__ void swe_step_finish(const SweStageArgs &p, int k, double beta, const double u[3], const double v[3],
                                                const double e[3], const double h[3], const double nx[3], const double ny[3],
                                                double twoA, int bmarkers, int bkind1, const double bu[3], const double bv[3], const double be[3],
                                                const double wu[3], const double wv[3], const double we[3], double ou[3],
                                                double ov[3], double oe[3])
{
#pragma clang fp contract(off)
    const double s = 6.0*p.dt*beta*swe_rcp(twoA);
    const double su = bu[0] + bu[1] + bu[2], sv = bv[0] + bv[1] + bv[2], se = be[0] + be[1] + be[2];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        ou[i] = fma(s, fma(4.0, bu[i], -su), wu[i]);
        ov[i] = fma(s, fma(4.0, bv[i], -sv), wv[i]);
        oe[i] = fma(s, fma(4.0, be[i], -se), we[i]);
    }
    if (bmarkers != 0) {
        const double sfac = 6.0*p.dt*beta*swe_rcp(fma(-ny[0], -nx[2], -(nx[0]*ny[2])));
        int rem = ((bmarkers & 0xff) ? 1 : 0) | ((bmarkers & 0xff00) ? 2 : 0) | ((bmarkers & 0xff0000) ? 4 : 0);
        int kind_next = bkind1;
        // scalars, not the parameter arrays: a select between loads through a pointer parameter becomes a load from a selected
        // address and the arrays end up in scratch
        const double u_0 = u[0], u_1 = u[1], u_2 = u[2], v_0 = v[0], v_1 = v[1], v_2 = v[2], e_0 = e[0], e_1 = e[1], e_2 = e[2];
        const double h_0 = h[0], h_1 = h[1], h_2 = h[2], nx_0 = nx[0], nx_1 = nx[1], nx_2 = nx[2], ny_0 = ny[0], ny_1 = ny[1], ny_2 = ny[2];
#define SWE_SEL3(x, i) ((i) == 0 ? x##_0 : ((i) == 1 ? x##_1 : x##_2))
#pragma unroll 1
        while (rem) {
            const int f = (rem & 1) ? 0 : ((rem & 2) ? 1 : 2);
            rem &= rem - 1;
            const int a = f, b = (f == 2) ? 0 : f + 1;
            const double nxs = SWE_SEL3(nx, f), nys = SWE_SEL3(ny, f);
            double Lf, rLf;
            swe_sqrt_rsqrt(swe_dot2(nxs, nxs, nys, nys), Lf, rLf);
            double Fau = 0.0, Fbu = 0.0, Fav = 0.0, Fbv = 0.0, Fae = 0.0, Fbe = 0.0;
            const double Ha_ = !NONLIN ? SWE_SEL3(h, a) : SWE_SEL3(h, a) + SWE_SEL3(e, a);
            const double Hb_ = !NONLIN ? SWE_SEL3(h, b) : SWE_SEL3(h, b) + SWE_SEL3(e, b);
            swe_boundary_facet<NONLIN, LF, false>(p, (bmarkers >> (8*f)) & 0xff, k, a, b, SWE_SEL3(u, a), SWE_SEL3(u, b), SWE_SEL3(v, a),
                                                  SWE_SEL3(v, b), SWE_SEL3(e, a), SWE_SEL3(e, b), SWE_SEL3(h, a), SWE_SEL3(h, b),
                                                  Ha_, Hb_, 0.0, 0.0, nxs, nys, Lf, rLf, Fau, Fbu, Fav, Fbv, Fae, Fbe, kind_next);
            kind_next = -1;
            const double dau = -0.5*Fau, dbu = -0.5*Fbu, dav = -0.5*Fav, dbv = -0.5*Fbv, dae = -0.5*Fae, dbe = -0.5*Fbe;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const double wa = (i == a) ? 3.0 : -1.0, wb = (i == b) ? 3.0 : -1.0;
                ou[i] = fma(sfac, fma(wa, dau, wb*dbu), ou[i]);
                ov[i] = fma(sfac, fma(wa, dav, wb*dbv), ov[i]);
                oe[i] = fma(sfac, fma(wa, dae, wb*dbe), oe[i]);
            }
        }
#undef SWE_SEL3
    }
}

// BLOCK: the workgroup size as a compile-time constant - the LDS plane offsets of every trace read become instruction immediates
template <bool NONLIN, bool LF, bool SRC, int BLOCK>
__global__ __launch_bounds__(BLOCK) SWE_STEP_OCCUPANCY void swe_step_kernel(const SweStepArgs q)
{
#pragma clang fp contract(off)
    extern __shared__ double swe_step_xs[];                    // [9][B + SWE_STEP_EXTRA]: the stage values of the tile's slots
    const SweStageArgs &p = q.st;
    const int tile = swe_logical_block(blockIdx.x, gridDim.x);
    if (tile >= q.n_tiles) return;
    constexpr int B = BLOCK, XS = BLOCK + SWE_STEP_EXTRA;
    const int j = (int)threadIdx.x;
    const int4 tn = q.tile_n[tile];
    double *xs = swe_step_xs;
    const size_t S = p.stride;
    const unsigned S8 = (unsigned)S*8u;
    const swe_rsrc_t gu = swe_rsrc(p.uin), gv = swe_rsrc(p.uin + 3*S), ge = swe_rsrc(p.uin + 6*S);
    const int4 *slots = q.tile_slot + (size_t)tile*XS;
    // ring-3 cells beyond the workgroup size: the first lanes fetch and publish their step input too
    if (B + j < tn.w) {
        const unsigned kx8 = (unsigned)slots[B + j].x*8u;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            xs[i*XS + B + j] = swe_ld(gu, kx8, i*S8);
            xs[(3 + i)*XS + B + j] = swe_ld(gv, kx8, i*S8);
            xs[(6 + i)*XS + B + j] = swe_ld(ge, kx8, i*S8);
        }
    }
    const bool in_tile = j < tn.w;
    const int4 sl = in_tile ? slots[j] : int4{0, 0, 0, 0};
    const int k = sl.x, meta = sl.z;
    const unsigned k8 = (unsigned)k*8u;
    unsigned lnb_v = (unsigned)sl.y;
    int meta_v = sl.z;
    double u[3], v[3], e[3];
    if (in_tile) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            u[i] = swe_ld(gu, k8, i*S8);
            v[i] = swe_ld(gv, k8, i*S8);
            e[i] = swe_ld(ge, k8, i*S8);
        }
    }
    double h[3] = {0.0, 0.0, 0.0}, nx[3] = {0.0, 0.0, 0.0}, ny[3] = {0.0, 0.0, 0.0};
    int bkind1 = 0;
    if (j < tn.z) {
        const int2 v12 = q.tile_vert[(size_t)tile*B + j];
        const int vid[3] = {sl.w, v12.x, v12.y};
        double px[3], py[3];
        const swe_rsrc_t rvx = swe_rsrc(p.vx), rvy = swe_rsrc(p.vy), rvh = swe_rsrc(p.vh);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const unsigned v8 = (unsigned)vid[i]*8u;
            px[i] = swe_ld(rvx, v8, 0);
            py[i] = swe_ld(rvy, v8, 0);
            h[i] = swe_ld(rvh, v8, 0);
        }
        const int bmarkers = meta & 0xffffff;
        if (bmarkers != 0) {
            const int m1 = (bmarkers & 0xff) ? (bmarkers & 0xff) : ((bmarkers & 0xff00) ? ((bmarkers >> 8) & 0xff) : (bmarkers >> 16));
            bkind1 = m1 < SWE_MAX_MARKERS ? p.bc.kind[m1] : 0;
        }
#pragma unroll
        for (int f = 0; f < 3; f++) {
            const int b = (f + 1) % 3;
            nx[f] = py[b] - py[f];
            ny[f] = px[f] - px[b];
        }
    }
    if (in_tile) {
#pragma unroll
        for (int i = 0; i < 3; i++) { xs[i*XS + j] = u[i]; xs[(3 + i)*XS + j] = v[i]; xs[(6 + i)*XS + j] = e[i]; }
    }
    __syncthreads();
    if (j >= tn.z) return;                                     // ring 3 and unused slots: whole waves leave, the rest is masked
    double ou[3], ov[3], oe[3];
#pragma unroll 1
    for (int s = 0; s < 3; s++) {
        // Opaque to the optimiser: without this every stage-invariant quantity (facet lengths, reciprocals, selected slots and
        // LDS addresses, gradients ...) is hoisted out of the stage loop and kept live across it - past the register budget.
        // The per-stage kernel recomputes them in every stage as well.
#pragma unroll
        for (int i = 0; i < 3; i++) asm volatile("" : "+v"(nx[i]), "+v"(ny[i]), "+v"(h[i]));
        asm volatile("" : "+v"(lnb_v), "+v"(meta_v));
        const unsigned lnb = lnb_v;
        const int meta = meta_v, bmarkers = meta_v & 0xffffff;
        double bu[3], bv[3], be[3], wu[3], wv[3], we[3];
        const double twoA = fma(nx[0], ny[1], -(ny[0]*nx[1]));
        swe_step_rhs<NONLIN, LF, SRC>(p, k, u, v, e, h, xs, XS, j, lnb, meta, nx, ny, twoA, bu, bv, be);
        // w = a0*U(0) + a1*U_in: the first stage has no U(0) term (swe_stage_kernel<., ., HASU0 = false>); later stages read
        // U(0) of the cell again (L2) instead of keeping it in registers
        const double a0 = q.a0[s], a1 = q.a1[s];
#pragma unroll
        for (int i = 0; i < 3; i++) { wu[i] = a1*u[i]; wv[i] = a1*v[i]; we[i] = a1*e[i]; }
        if (s > 0) {
#pragma unroll
            for (int i = 0; i < 3; i++) {
                wu[i] = fma(a0, swe_ld(gu, k8, i*S8), wu[i]);
                wv[i] = fma(a0, swe_ld(gv, k8, i*S8), wv[i]);
                we[i] = fma(a0, swe_ld(ge, k8, i*S8), we[i]);
            }
        }
        swe_step_finish<NONLIN, LF>(p, k, q.beta[s], u, v, e, h, nx, ny, twoA, bmarkers, bkind1, bu, bv, be, wu, wv, we, ou, ov, oe);
        if (s == 2) break;
        __syncthreads();                                       // every trace of this stage's input has been read
#pragma unroll
        for (int i = 0; i < 3; i++) { xs[i*XS + j] = ou[i]; xs[(3 + i)*XS + j] = ov[i]; xs[(6 + i)*XS + j] = oe[i]; }
        __syncthreads();
        if (j >= (s == 0 ? tn.y : tn.x)) return;               // ring 2 is done after stage 1, ring 1 after stage 2
#pragma unroll
        for (int i = 0; i < 3; i++) { u[i] = ou[i]; v[i] = ov[i]; e[i] = oe[i]; }
    }
    const swe_rsrc_t gou = swe_rsrc(p.uout), gov = swe_rsrc(p.uout + 3*S), goe = swe_rsrc(p.uout + 6*S);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        swe_st(gou, k8, i*S8, ou[i]);
        swe_st(gov, k8, i*S8, ov[i]);
        swe_st(goe, k8, i*S8, oe[i]);
    }
}
