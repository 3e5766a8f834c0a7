// swe2d_flow.h - many SSPRK33 stages of the triangle DG-P1 shallow water equations in ONE launch, without a grid barrier:
// a dataflow stage loop for cell ranges whose 64-cell blocks are all resident at once (gfx950; <= ~190 k cells).
//
// What it replaces: the reference runs solve_stage(i) on the whole mesh and then solve_stage(i + 1) (thetis/rungekutta.py:930-952),
// and under mpiexec every par_loop is preceded by a halo exchange (examples/README.md:51-56).  On one rank of eight of the 1 M
// triangle bench mesh (125 k cells = 1954 one-wave workgroups, two per SIMD) a stage LAUNCH is latency, not bandwidth: 0.5 us
// index loads + 2.5 us loads + 2.1 us arithmetic + 0.3 us stores in lock step, a tail of boundary waves, 1.5-1.9 us of kernel
// boundary and a reload of the state into L2s that the boundary invalidated.  Here
//
//   * one wave OWNS one 64-cell block of the device numbering for the whole launch: connectivity, geometry and the block's own
//     nodal values stay in registers from stage to stage (no index loads, no own-cell loads after the first stage);
//   * a block at stage s + 1 waits only for the blocks its facets touch to have finished stage s: every block publishes a
//     monotonically growing stage counter (flag), a lane polls the counters of ITS three neighbour cells' blocks - no grid
//     barrier, so the load phase of some blocks overlaps the arithmetic of others and nobody waits for the slowest wave of the
//     grid;
//   * stage values travel through memory, write-through: `sc1` stores, every storing wave drains them (s_waitcnt vmcnt(0))
//     before ONE lane raises the block's flag with an agent-scope store; consumers poll with agent-scope loads and gather the
//     neighbour traces with `sc1` loads (past the CU's L1, which other CUs' stores never refresh).  This is the
//     placement-independent hand-off of MI355X_MICROARCH.md ("sc1 loads may replace the acquire only when the producer stored
//     sc1"); the XCD-chunked block map (swe_logical_block) is used for speed only.
//
// Deadlock freedom needs every block of the launch resident: the host launches this kernel only when the grid fits the
// occupancy the runtime reports (swe2d_api.hip: flow_capacity), and every spin is bounded by the wall clock - a timeout is
// counted in the status word, the wave carries on (the result is then wrong and the host reports SWE2D_ERR_HIP at the next
// synchronisation point), it never hangs the device.
//
// Flags never need re-initialisation: a block that retires (its cells are outside the range of the remaining stages) raises
// its flag to `base + n_stages` as well, so after a launch every flag of the handle holds the same value, which is the next
// launch's base (read from the block's own flag).  The grid therefore always covers ALL blocks of the handle, also those that
// take part in no stage.
//
// The arithmetic is that of swe_stage_kernel<NONLIN, LF, ., SRC, false, false, true(BINL)>, operation for operation and under
// the same `fp contract(off)`: bit for bit the result of the stage launches on the same ranges (tests/test_gpu_flow_kernel.py).
#pragma once
#include "swe2d_kernels.h"

#ifndef SWE_FLOW_OCCUPANCY
#define SWE_FLOW_OCCUPANCY __attribute__((amdgpu_waves_per_eu(3, 3)))      // <= 168 VGPRs: three one-wave workgroups per SIMD
#endif
#define SWE_FLOW_MAX_STAGES 48             // 16 time steps per launch
#ifndef SWE_FLOW_FLAG_STRIDE
#define SWE_FLOW_FLAG_STRIDE 16           // unsigned words between two blocks' flags (64 B)
#endif

struct SweFlowArgs {
    SweStageArgs st;                       // geometry, connectivity, boundary tables, sources; uin/u0/uout/a0/a1/beta/cell_* unused
    double *buf[3];                        // state buffers A (U0 / step result), B, C
    unsigned *flag;                        // [n_blocks][SWE_FLOW_FLAG_STRIDE] stages finished by the block, monotonic over launches
    unsigned *status;                      // [0] timeouts, [1] first block that timed out + 1
    int n_blocks;                          // blocks of the handle (cells rounded up to 64)
    int n_stages;                          // a multiple of 3: stage s is Shu-Osher stage s % 3
    int cell_end[SWE_FLOW_MAX_STAGES];     // stage s updates the cells [0, cell_end[s]); non-increasing
    double a0[3], a1[3], beta[3];          // Shu-Osher weights per stage (swe2d_ssprk33_coefficients)
    unsigned long long timeout_ticks;      // wall_clock64 ticks (100 MHz)
};

#ifndef SWE_FLOW_LD_AUX
#define SWE_FLOW_LD_AUX 16                // cache policy of the stage-value loads / stores: 16 = sc1 (experiments: 0 = plain)
#endif
#ifndef SWE_FLOW_ST_AUX
#define SWE_FLOW_ST_AUX 16
#endif
__device__ __forceinline__ double swe_ld_sc1(swe_rsrc_t r, unsigned voff, unsigned soff)
{
#ifndef SWE_RANGE_CHECK
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, SWE_FLOW_LD_AUX));       // aux 16 = sc1
#else
    return swe_ld(r, voff, soff);
#endif
}
__device__ __forceinline__ void swe_st_sc1(swe_rsrc_t r, unsigned voff, unsigned soff, double x)
{
#ifndef SWE_RANGE_CHECK
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(swe_u32x2, x), r, voff, soff, SWE_FLOW_ST_AUX);
#else
    swe_st(r, voff, soff, x);
#endif
}

// right-hand side integrals of one cell from values in registers: cell integrals + interior facet fluxes (boundary facets
// contribute zero here, their flux is evaluated from the cell's own values and discarded - the branch-free facet loop of
// swe_stage_kernel).  Lines as in swe_stage_kernel, same order.
template <bool NONLIN, bool LF, bool SRC>
__device__ __forceinline__ void swe_flow_rhs(const SweStageArgs &p, int k, const double u[3], const double v[3], const double e[3],
                                             const double h[3], const double una[3], const double unb[3], const double vna[3],
                                             const double vnb[3], const double ena[3], const double enb[3], int bmarkers,
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
        const bool bnd = ((bmarkers >> (8*f)) & 0xff) != 0;
        const double nxs = nx[f], nys = ny[f];
        double Lf, rLf;
        swe_sqrt_rsqrt(swe_dot2(nxs, nxs, nys, nys), Lf, rLf);
        double Fau = 0.0, Fbu = 0.0, Fav = 0.0, Fbv = 0.0, Fae = 0.0, Fbe = 0.0;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const double xb = q ? SWE_XI1 : SWE_XI0, xa = 1.0 - xb;
            const double uq = swe_dot2(xa, u[a], xb, u[b]), vq = swe_dot2(xa, v[a], xb, v[b]), eq = swe_dot2(xa, e[a], xb, e[b]);
            const double hq = swe_dot2(xa, h[a], xb, h[b]);
            const double un = swe_dot2(xa, una[f], xb, unb[f]), vn = swe_dot2(xa, vna[f], xb, vnb[f]), en = swe_dot2(xa, ena[f], xb, enb[f]);
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
__device__ __forceinline__ void swe_flow_finish(const SweStageArgs &p, int k, double beta, const double u[3], const double v[3],
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

__device__ __forceinline__ unsigned swe_flow_ld_flag(const unsigned *f)
{
    return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // global_load_dword sc1
}

#ifdef SWE_WAVE_TIMING
// profiling build only (tools/flowtiming.py): the 100 MHz wall clock of every block at five points of ONE stage of the launch
#ifndef SWE_FLOW_TS_STAGE
#define SWE_FLOW_TS_STAGE 7
#endif
#define SWE_FT(i) do { if (s == SWE_FLOW_TS_STAGE && lane == 0 && lb < SWE_WT_MAX) swe_wave_ts[i][lb] = wall_clock64(); } while (0)
#else
#define SWE_FT(i)
#endif

template <bool NONLIN, bool LF, bool SRC>
__global__ __launch_bounds__(SWE_BLOCK) SWE_FLOW_OCCUPANCY void swe_flow_kernel(const SweFlowArgs q)
{
#pragma clang fp contract(off)
    const SweStageArgs &p = q.st;
    const int lb = swe_logical_block(blockIdx.x, gridDim.x);
    if (lb >= q.n_blocks) return;                              // padding of the grid to a multiple of 8
    const int lane = (int)threadIdx.x;
    const int kraw = lb*SWE_BLOCK + lane;
    const size_t S = p.stride;
    const unsigned S8 = (unsigned)S*8u;
    unsigned *const myflag = q.flag + (size_t)lb*SWE_FLOW_FLAG_STRIDE;
    const unsigned base = *myflag;                             // written by this block's wave in the previous launch (or 0)
    const unsigned fin = base + (unsigned)q.n_stages;
    if (lb*SWE_BLOCK >= q.cell_end[0]) {                       // this block takes part in no stage of the launch
        if (lane == 0) __hip_atomic_store(myflag, fin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // lanes beyond the first stage's range mimic the range's last cell: finite values, never stored
    const int k = min(kraw, q.cell_end[0] - 1);
    const unsigned k8 = (unsigned)k*8u;

    // ---- launch invariants of the cell: connectivity, neighbour addresses, geometry
    int bmarkers, bkind1 = 0;
    unsigned oa[3], ob[3];                 // byte offsets of the neighbour's nodes on my nodes f and f + 1 (inside a 3-plane group)
    int kn[3];                             // neighbour cells (this cell itself for a boundary facet)
    double h[3], nx[3], ny[3];
    {
        const int4 q4 = p.idx4[k];
        const int2 q2 = p.idx2[k];
        const int nb[3] = {q4.x, q4.y, q4.z};
        const int vid[3] = {q4.w, q2.x, q2.y};
        bmarkers = (nb[0] < 0 ? -nb[0] : 0) | (nb[1] < 0 ? (-nb[1]) << 8 : 0) | (nb[2] < 0 ? (-nb[2]) << 16 : 0);
        if (bmarkers != 0) {
            const int m1 = (bmarkers & 0xff) ? (bmarkers & 0xff) : ((bmarkers & 0xff00) ? ((bmarkers >> 8) & 0xff) : (bmarkers >> 16));
            bkind1 = m1 < SWE_MAX_MARKERS ? p.bc.kind[m1] : 0;
        }
#pragma unroll
        for (int f = 0; f < 3; f++) {
            const int nbf = nb[f];
            kn[f] = nbf >= 0 ? (nbf >> 2) : k;
            const int f2 = nbf >= 0 ? (nbf & 3) : f;
            const unsigned kn8 = (unsigned)kn[f]*8u;
            ob[f] = kn8 + (f2 == 0 ? 0u : (f2 == 1 ? S8 : 2u*S8));         // node f2
            oa[f] = kn8 + (f2 == 0 ? S8 : (f2 == 1 ? 2u*S8 : 0u));         // node (f2 + 1) % 3
        }
        double px[3], py[3];
        const swe_rsrc_t rvx = swe_rsrc(p.vx), rvy = swe_rsrc(p.vy), rvh = swe_rsrc(p.vh);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const unsigned v8 = (unsigned)vid[i]*8u;
            px[i] = swe_ld(rvx, v8, 0);
            py[i] = swe_ld(rvy, v8, 0);
            h[i] = swe_ld(rvh, v8, 0);
        }
#pragma unroll
        for (int f = 0; f < 3; f++) {
            const int b = (f + 1) % 3;
            nx[f] = py[b] - py[f];
            ny[f] = px[f] - px[b];
        }
    }
    // the launch's input: written by earlier kernels, plain loads
    double u[3], v[3], e[3];
    {
        const swe_rsrc_t gu = swe_rsrc(q.buf[0]), gv = swe_rsrc(q.buf[0] + 3*S), ge = swe_rsrc(q.buf[0] + 6*S);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            u[i] = swe_ld(gu, k8, i*S8);
            v[i] = swe_ld(gv, k8, i*S8);
            e[i] = swe_ld(ge, k8, i*S8);
        }
    }
    unsigned long long t_start = 0ull;
    bool late = false;

#pragma unroll 1
    for (int s = 0; s < q.n_stages; s++) {
        const int end_s = q.cell_end[s];
        if (lb*SWE_BLOCK >= end_s) break;                      // retired: the ranges only shrink
        const bool act = kraw < end_s;
        const int i3 = s % 3;
        SWE_FT(0);
#ifdef SWE_WAVE_TIMING
        if (s == SWE_FLOW_TS_STAGE && lane == 0 && lb < SWE_WT_MAX) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            swe_wave_ts[5][lb] = (unsigned long long)hw | ((unsigned long long)(xcc & 0xf) << 32);
        }
#endif
        const double *bin = q.buf[i3];
        double *bout = q.buf[i3 == 2 ? 0 : i3 + 1];
        // Opaque to the optimiser: without this every stage-invariant quantity (facet lengths, reciprocals, gradients ...) is
        // hoisted out of the stage loop and kept live across it - past the register budget.  The per-stage kernel recomputes
        // them in every stage as well.
#pragma unroll
        for (int i = 0; i < 3; i++) asm volatile("" : "+v"(nx[i]), "+v"(ny[i]), "+v"(h[i]));
        asm volatile("" : "+v"(bmarkers));
        // w = a0*U(0) + a1*U_in; U(0) of the cell is this block's own stage-3 result of the previous step (or the launch input):
        // requested before the wait
        const double a0 = q.a0[i3], a1 = q.a1[i3];
        double wu[3], wv[3], we[3];
#pragma unroll
        for (int i = 0; i < 3; i++) { wu[i] = a1*u[i]; wv[i] = a1*v[i]; we[i] = a1*e[i]; }
        double u0[3], v0[3], e0[3];
        if (i3 > 0) {
            const swe_rsrc_t g0u = swe_rsrc(q.buf[0]), g0v = swe_rsrc(q.buf[0] + 3*S), g0e = swe_rsrc(q.buf[0] + 6*S);
#pragma unroll
            for (int i = 0; i < 3; i++) {
                u0[i] = swe_ld_sc1(g0u, k8, i*S8);
                v0[i] = swe_ld_sc1(g0v, k8, i*S8);
                e0[i] = swe_ld_sc1(g0e, k8, i*S8);
            }
        }
        // ---- wait until the blocks of my three neighbour cells have finished stage s - 1 (a neighbour that stage s - 1 did not
        //      update - outside its range - is read as it is, like a stage launch would)
        if (s > 0 && !late) {
            const int end_p = q.cell_end[s - 1];
            const unsigned need = base + (unsigned)s;
            const unsigned *f0 = q.flag + (size_t)(kn[0] < end_p ? (kn[0] >> 6) : lb)*SWE_FLOW_FLAG_STRIDE;
            const unsigned *f1 = q.flag + (size_t)(kn[1] < end_p ? (kn[1] >> 6) : lb)*SWE_FLOW_FLAG_STRIDE;
            const unsigned *f2 = q.flag + (size_t)(kn[2] < end_p ? (kn[2] >> 6) : lb)*SWE_FLOW_FLAG_STRIDE;
            for (unsigned spins = 0;; spins++) {
#ifdef SWE_FLOW_LOCAL_EXPERIMENT
                const size_t lo = (size_t)q.n_blocks*SWE_FLOW_FLAG_STRIDE;
                const unsigned g0 = swe_flow_ld_flag(f0), g1 = swe_flow_ld_flag(f1), g2 = swe_flow_ld_flag(f2);
                const unsigned l0 = swe_flow_ld_flag(f0 + lo), l1 = swe_flow_ld_flag(f1 + lo), l2 = swe_flow_ld_flag(f2 + lo);
                const unsigned c0 = (int)(l0 - g0) > 0 ? l0 : g0, c1 = (int)(l1 - g1) > 0 ? l1 : g1, c2 = (int)(l2 - g2) > 0 ? l2 : g2;
#else
                const unsigned c0 = swe_flow_ld_flag(f0), c1 = swe_flow_ld_flag(f1), c2 = swe_flow_ld_flag(f2);
#endif
                const bool ok = (int)(c0 - need) >= 0 && (int)(c1 - need) >= 0 && (int)(c2 - need) >= 0;
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
                if ((spins & 63u) == 63u) {
                    const unsigned long long now = wall_clock64();
                    if (t_start == 0ull) t_start = now;
                    else if (now - t_start > q.timeout_ticks) { late = true; break; }
                }
            }
            t_start = 0ull;
            if (late && lane == 0) {
                if (atomicAdd(q.status, 1u) == 0u) q.status[1] = (unsigned)lb + 1u;
            }
        }
        asm volatile("" ::: "memory");
        SWE_FT(1);
        // ---- neighbour traces of stage s - 1 (sc1: past this CU's L1, which other CUs' stores never refresh)
        double una[3], unb[3], vna[3], vnb[3], ena[3], enb[3];
        {
            const swe_rsrc_t gu = swe_rsrc(bin), gv = swe_rsrc(bin + 3*S), ge = swe_rsrc(bin + 6*S);
#pragma unroll
            for (int f = 0; f < 3; f++) {
                una[f] = swe_ld_sc1(gu, oa[f], 0);
                unb[f] = swe_ld_sc1(gu, ob[f], 0);
                vna[f] = swe_ld_sc1(gv, oa[f], 0);
                vnb[f] = swe_ld_sc1(gv, ob[f], 0);
                ena[f] = swe_ld_sc1(ge, oa[f], 0);
                enb[f] = swe_ld_sc1(ge, ob[f], 0);
            }
        }
        if (i3 > 0) {
#pragma unroll
            for (int i = 0; i < 3; i++) {
                wu[i] = fma(a0, u0[i], wu[i]);
                wv[i] = fma(a0, v0[i], wv[i]);
                we[i] = fma(a0, e0[i], we[i]);
            }
        }
#ifdef SWE_WAVE_TIMING
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SWE_FT(2);
#endif
        double bu[3], bv[3], be[3], ou[3], ov[3], oe[3];
        const double twoA = fma(nx[0], ny[1], -(ny[0]*nx[1]));
        swe_flow_rhs<NONLIN, LF, SRC>(p, k, u, v, e, h, una, unb, vna, vnb, ena, enb, bmarkers, nx, ny, twoA, bu, bv, be);
        swe_flow_finish<NONLIN, LF>(p, k, q.beta[i3], u, v, e, h, nx, ny, twoA, bmarkers, bkind1, bu, bv, be, wu, wv, we, ou, ov, oe);
#ifdef SWE_WAVE_TIMING
        if (ou[0] == 1.2345e300) return;          // the arithmetic has to be finished before the time stamp
        SWE_FT(3);
#endif
        // ---- publish: write-through stores, drained by this (the only storing) wave, then the block's flag
        if (act) {
            const swe_rsrc_t gou = swe_rsrc(bout), gov = swe_rsrc(bout + 3*S), goe = swe_rsrc(bout + 6*S);
#pragma unroll
            for (int i = 0; i < 3; i++) {
                swe_st_sc1(gou, k8, i*S8, ou[i]);
                swe_st_sc1(gov, k8, i*S8, ov[i]);
                swe_st_sc1(goe, k8, i*S8, oe[i]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SWE_FT(4);
#ifdef SWE_FLOW_LOCAL_EXPERIMENT
        if (lane == 0 && s + 1 < q.n_stages) *(volatile unsigned *)(myflag + (size_t)q.n_blocks*SWE_FLOW_FLAG_STRIDE) = base + (unsigned)s + 1u;
#endif
        if (lane == 0 && s + 1 < q.n_stages) __hip_atomic_store(myflag, base + (unsigned)s + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int i = 0; i < 3; i++) { u[i] = ou[i]; v[i] = ov[i]; e[i] = oe[i]; }
    }
    // retired or finished: every flag of the handle ends the launch at base + n_stages
    if (lane == 0) __hip_atomic_store(myflag, fin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
