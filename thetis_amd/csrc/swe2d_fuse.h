// swe2d_fuse.h - stages 1 and 2 of an SSPRK33 step in ONE launch by overlapped tiles.  What swe2d_advance takes by itself from
// 250 k triangles on a whole mesh the kernel covers (swe2d_api_fuse.hip: fuse12_covers; SWE2D_OPT_FUSED_STAGES forces or forbids it).
//
// A stage launch streams the state: per triangle and step 72 B read + 72 B written in stage 1, 72 + 72 B read + 72 B written in
// stage 2.  U(1) is read by stage 2 and by nothing else (rungekutta.py:ERKGenericShuOsher, stage 3 takes
// U(2) and U(0)), so a workgroup that owns a TILE of the mesh can keep it on chip: 256 lanes = up to 192 interior cells (lanes
// 0 .. n_inner-1: three waves) + the cells that share a facet with them (the ring, at most 64: the fourth wave).  Stage 1 is
// evaluated for every cell of the tile - the ring's redundantly, its outer neighbours' traces gathered from the state planes -,
// the results go to LDS, and stage 2 is evaluated for the interior cells with every neighbour trace from LDS.  Per interior
// cell the launch reads U(0) once (x 1.24 for the ring on the bench mesh) and writes U(2): 192 MB per launch at 1 M cells by the
// counters where the two stage launches move 428 MB.
// The arithmetic is the dataflow kernel's (swe_flow_rhs_cell / swe_flow_rhs_facets / swe_flow_finish = swe_stage_kernel's
// operations in its order): bit for bit the stage launches (tests/test_gpu_parity.py::test_fused_stage_pair_gives_the_bits_...),
// and checked directly against the oracle (::test_fused_stage_pair_matches_the_c_restatement...).
// Covers: triangles, whole mesh or the owned + ghost ranges of a partition, with or without source terms; no wetting-drying, no
// viscosity (those keep the stage launches).
#pragma once
#include "swe2d_kernels.h"
#include "swe2d_flow.h"

#ifndef SWE_FUSE_WG
#define SWE_FUSE_WG 256
#endif
#ifndef SWE_FUSE_MIN_WG
#define SWE_FUSE_MIN_WG 3                                 // workgroups per CU the compiler has to make room for (3: 168 VGPRs)
#endif
#ifndef SWE_FUSE_INNER
#define SWE_FUSE_INNER 192
#endif
#define SWE_FUSE_FBITS 10                                // bits per facet in the tile table's packed word: 9 of lane / slot + the flag
#define SWE_FUSE_RING (SWE_FUSE_WG - SWE_FUSE_INNER)
#define SWE_FUSE_XG (9*SWE_FUSE_WG)                    // staging area of the traces from outside the tile: [slot][6]
#define SWE_FUSE_MAX_OUT (2*SWE_FUSE_RING)             // a ring cell has a facet towards the interior: at most two towards the outside
#define SWE_FUSE_LDS (SWE_FUSE_XG + 6*SWE_FUSE_MAX_OUT)

struct SweFuseArgs {
    SweStageArgs st;          // uin = U(0) (state buffer A); geometry, connectivity, boundary tables; dt, g, sigma_lf
    const int2 *tile;         // [n_tiles][256]: {cell or -1, per facet 10 bits: [8:0] lane of the neighbour in the tile (a boundary
                              //  facet: the lane itself) or, with bit 9 set, the staging slot of a neighbour outside the tile}
    const int *n_inner;       // [n_tiles]: lanes 0 .. n_inner-1 hold the interior cells
    int n_tiles;
    int cell_end;             // stage 2 updates the interior cells < cell_end (a partition's shrinking stage ranges; else n_cells)
    double beta1;             // stage 1: U(1) = U(0) + beta1 dt M^-1 R(U(0))
    double a0_2, a1_2, beta2; // stage 2: U(2) = a0 U(0) + a1 U(1) + beta2 dt M^-1 R(U(1))
    double *out;              // 9 planes: U(2) (state buffer C)
};

template <bool NONLIN, bool LF, bool SRC = false>
__global__ __launch_bounds__(SWE_FUSE_WG, SWE_FUSE_MIN_WG) void swe_fuse12_kernel(const SweFuseArgs q)
{
#pragma clang fp contract(off)
    __shared__ double lds[SWE_FUSE_LDS];
    __shared__ double lw[9][SWE_FUSE_INNER];               // a0 U(0) + a1 U(1) of the interior cells: 18 registers stage 2 cannot spare
    const SweStageArgs &p = q.st;
    const int tile = swe_logical_block(blockIdx.x, gridDim.x);
    if (tile >= q.n_tiles) return;                         // padding of the grid to a multiple of 8
    const int lane = (int)threadIdx.x;
    const int2 tl = q.tile[(size_t)tile*SWE_FUSE_WG + lane];
    const bool real = tl.x >= 0;
    const int k = real ? tl.x : 0;
    const int n_inner = q.n_inner[tile];
    const size_t S = p.stride;
    const unsigned S8 = (unsigned)S*8u, k8 = (unsigned)k*8u;

    int bmarkers = 0, bkind1 = 0;
    // LDS addresses of the traces of every facet (swe_flow_rhs_facets): component 0's pair, the others follow from it - 3*256 apart in
    // the planes, 1 apart in the staging area (six registers less than one pair per component: what three workgroups per CU need)
    unsigned tr[3][1];
    double h[3], nx[3], ny[3], u[3], v[3], e[3];
    if (real) {
        int nb[3], vid[3];
        swe_conn_load(p.idxc, p.idx4, p.idx2, k, nb, vid);
        bmarkers = (nb[0] < 0 ? -nb[0] : 0) | (nb[1] < 0 ? (-nb[1]) << 8 : 0) | (nb[2] < 0 ? (-nb[2]) << 16 : 0);
        if (bmarkers != 0) {
            const int m1 = (bmarkers & 0xff) ? (bmarkers & 0xff) : ((bmarkers & 0xff00) ? ((bmarkers >> 8) & 0xff) : (bmarkers >> 16));
            bkind1 = m1 < SWE_MAX_MARKERS ? p.bc.kind[m1] : 0;
        }
        const swe_rsrc_t gu = swe_rsrc(p.uin), gv = swe_rsrc(p.uin + 3*S), ge = swe_rsrc(p.uin + 6*S);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            u[i] = swe_ld(gu, k8, i*S8);
            v[i] = swe_ld(gv, k8, i*S8);
            e[i] = swe_ld(ge, k8, i*S8);
        }
        double r0[3][6];
        bool outside[3];
#pragma unroll
        for (int f = 0; f < 3; f++) {
            const int nbf = nb[f];
            const unsigned w = ((unsigned)tl.y >> (SWE_FUSE_FBITS*f)) & 0x3ffu;
            outside[f] = (w & 0x200u) != 0u;
            const unsigned at = w & 0x1ffu;                                   // lane in the tile, or staging slot
            // the neighbour traverses the shared facet backwards: its node f2 sits on my node f + 1, its node (f2 + 1) % 3 on my node f
            const int f2 = nbf >= 0 ? (nbf & 3) : f, f2a = f2 == 2 ? 0 : f2 + 1;
            {
                const unsigned ab = outside[f] ? (unsigned)(SWE_FUSE_XG + 6*at) : (unsigned)(f2*SWE_FUSE_WG + at);
                const unsigned aa = outside[f] ? (unsigned)(SWE_FUSE_XG + 6*at + 3) : (unsigned)(f2a*SWE_FUSE_WG + at);
                tr[f][0] = ab | (aa << 16);
            }
            // (issued for every facet: a facet inside the tile reads this cell itself, value unused - no branch around the loads)
            const int code = outside[f] ? nbf : ((k << 2) | f);
            const unsigned kn8 = (unsigned)(code >> 2)*8u;
            const int g2 = code & 3;
            const unsigned ob = kn8 + (g2 == 0 ? 0u : (g2 == 1 ? S8 : 2u*S8));         // node g2
            const unsigned oa = kn8 + (g2 == 0 ? S8 : (g2 == 1 ? 2u*S8 : 0u));         // node (g2 + 1) % 3
            r0[f][0] = swe_ld(gu, ob, 0); r0[f][1] = swe_ld(gv, ob, 0); r0[f][2] = swe_ld(ge, ob, 0);
            r0[f][3] = swe_ld(gu, oa, 0); r0[f][4] = swe_ld(gv, oa, 0); r0[f][5] = swe_ld(ge, oa, 0);
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
#pragma unroll
        for (int i = 0; i < 3; i++) { lds[i*SWE_FUSE_WG + lane] = u[i]; lds[(3 + i)*SWE_FUSE_WG + lane] = v[i]; lds[(6 + i)*SWE_FUSE_WG + lane] = e[i]; }
#pragma unroll
        for (int f = 0; f < 3; f++) {
            if (outside[f]) {
                const unsigned at = ((unsigned)tl.y >> (SWE_FUSE_FBITS*f)) & 0x1ffu;
#pragma unroll
                for (int j = 0; j < 6; j++) lds[SWE_LDSI(SWE_FUSE_XG + 6*at + j, SWE_FUSE_LDS)] = r0[f][j];
            }
        }
    }
    __syncthreads();
    // ---- stage 1 on every cell of the tile: U(1) = U(0) + beta1 dt M^-1 R(U(0))
    double o1u[3], o1v[3], o1e[3];
    double twoA = 0.0;
    if (real) {
        twoA = fma(nx[0], ny[1], -(ny[0]*nx[1]));
        double bu[3], bv[3], be[3], wu[3], wv[3], we[3];
        swe_flow_rhs_cell<NONLIN>(p, u, v, e, h, nx, ny, bu, bv, be);
        swe_flow_rhs_facets<NONLIN, LF, SRC, 1, false, SWE_FUSE_XG, SWE_FUSE_WG, SWE_FUSE_LDS>(p, k, u, v, e, h, lds, tr, bmarkers, nx, ny, twoA, bu, bv, be);
#pragma unroll
        for (int i = 0; i < 3; i++) { wu[i] = 1.0*u[i]; wv[i] = 1.0*v[i]; we[i] = 1.0*e[i]; }
        swe_flow_finish<NONLIN, LF, true>(p, k, q.beta1, u, v, e, h, nx, ny, twoA, bmarkers, bkind1, bu, bv, be, wu, wv, we, o1u, o1v, o1e);
    }
    __syncthreads();                                       // every lane has read its traces of U(0)
    if (real) {
#pragma unroll
        for (int i = 0; i < 3; i++) { lds[i*SWE_FUSE_WG + lane] = o1u[i]; lds[(3 + i)*SWE_FUSE_WG + lane] = o1v[i]; lds[(6 + i)*SWE_FUSE_WG + lane] = o1e[i]; }
    }
    const bool act2 = lane < n_inner && k < q.cell_end;    // (a partition: stage 2's range ends before stage 1's)
    if (act2) {                                            // the part of stage 2's combine that does not depend on its tendency; U(0) is dead after this
#pragma unroll
        for (int i = 0; i < 3; i++) {
            lw[i][lane] = fma(q.a0_2, u[i], q.a1_2*o1u[i]);
            lw[3 + i][lane] = fma(q.a0_2, v[i], q.a1_2*o1v[i]);
            lw[6 + i][lane] = fma(q.a0_2, e[i], q.a1_2*o1e[i]);
        }
    }
    __syncthreads();
    // ---- stage 2 on the interior cells (every neighbour is a cell of the tile): U(2) = a0 U(0) + a1 U(1) + beta2 dt M^-1 R(U(1))
    if (act2) {
        // opaque to the optimiser (as in swe_flow_kernel): what stage 1 derived from the geometry - facet lengths, reciprocals,
        // gradients - would otherwise stay live across the barrier for stage 2, past the register budget of three waves per SIMD
#pragma unroll
        for (int i = 0; i < 3; i++) asm volatile("" : "+v"(nx[i]), "+v"(ny[i]), "+v"(h[i]));
#pragma unroll
        for (int f = 0; f < 3; f++) asm volatile("" : "+v"(tr[f][0]));
        asm volatile("" : "+v"(bmarkers), "+v"(twoA));
        double bu[3], bv[3], be[3], wu[3], wv[3], we[3], ou[3], ov[3], oe[3];
        swe_flow_rhs_cell<NONLIN>(p, o1u, o1v, o1e, h, nx, ny, bu, bv, be);
        swe_flow_rhs_facets<NONLIN, LF, SRC, 1, false, SWE_FUSE_XG, SWE_FUSE_WG, SWE_FUSE_LDS>(p, k, o1u, o1v, o1e, h, lds, tr, bmarkers, nx, ny, twoA, bu, bv, be);
#pragma unroll
        for (int i = 0; i < 3; i++) { wu[i] = lw[i][lane]; wv[i] = lw[3 + i][lane]; we[i] = lw[6 + i][lane]; }
        swe_flow_finish<NONLIN, LF, true>(p, k, q.beta2, o1u, o1v, o1e, h, nx, ny, twoA, bmarkers, bkind1, bu, bv, be, wu, wv, we, ou, ov, oe);
        const swe_rsrc_t gou = swe_rsrc(q.out), gov = swe_rsrc(q.out + 3*S), goe = swe_rsrc(q.out + 6*S);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            swe_st(gou, k8, i*S8, ou[i]);
            swe_st(gov, k8, i*S8, ov[i]);
            swe_st(goe, k8, i*S8, oe[i]);
        }
    }
}

// ---- all THREE stages of a step in one launch: two rings per tile (round 6, VERDICT r05 "next 2d") ---------------------------------
// A tile = interior cells (lanes 0 .. n_inner-1) + ring 1 (their facet neighbours, lanes n_inner .. n_mid-1) + ring 2 (the facet
// neighbours of ring 1, lanes n_mid .. n_all-1), 256 lanes in all.  Stage 1 on every cell of the tile (ring 2's outer neighbours'
// traces gathered from the state planes), stage 2 on interior + ring 1, stage 3 on the interior: U(1) and U(2) never leave the chip,
// per interior cell the launch reads U(0) once (x the tile's redundancy) and writes U(3).  U(3) goes to ANOTHER buffer than U(0) -
// a tile reads ring cells whose owners may already have finished - and the host swaps the two pointers after the launch.
// LDS: P0 [9][256] = U(0) (traces of stage 1; every lane's own U(0) for the Shu-Osher weights of stages 2 and 3), P1 [9][256] = the
// running stage values (U(1), then U(2)), then the staging area of the traces from outside the tile: 36.9 KB + 6 x 8 B per slot.
// Same arithmetic, same order as the stage launches: the same bits (tests/test_gpu_parity.py::test_fused_stage_triple...).
#define SWE_FUSE3_XG (18*SWE_FUSE_WG)                   // the staging area follows P0 and P1
#ifndef SWE_FUSE3_MAX_OUT
#define SWE_FUSE3_MAX_OUT 224                           // staging slots (a ring-2 cell has at most two facets towards the outside)
#endif
#define SWE_FUSE3_LDS (SWE_FUSE3_XG + 6*SWE_FUSE3_MAX_OUT)
#ifndef SWE_FUSE3_SRC_MIN_WG
#define SWE_FUSE3_SRC_MIN_WG 2                          // the source-term instances need 187-199 VGPRs: two workgroups per CU (at three: 80-132 B of scratch per lane)
#endif

struct SweFuse3Args {
    SweStageArgs st;          // uin = U(0) (state buffer A); geometry, connectivity, boundary tables; dt, g, sigma_lf
    const int2 *tile;         // [n_tiles][256]: as SweFuseArgs::tile (lane of the neighbour in the tile, or bit 9 + staging slot)
    const int2 *counts;       // [n_tiles]: {n_inner, n_mid}
    int n_tiles;
    int cell_end;             // stage 3 (the only one that leaves the chip) on cells [0, cell_end): a partition's last stage range
    double a0[3], a1[3], beta[3];   // Shu-Osher weights per stage (swe2d_ssprk33_coefficients)
    double *out;              // 9 planes: U(3), NOT the buffer of U(0)
};

template <bool NONLIN, bool LF, bool SRC = false>
__global__ __launch_bounds__(SWE_FUSE_WG, SRC ? SWE_FUSE3_SRC_MIN_WG : SWE_FUSE_MIN_WG) void swe_fuse123_kernel(const SweFuse3Args q)
{
#pragma clang fp contract(off)
    __shared__ double lds[SWE_FUSE3_LDS];
    const SweStageArgs &p = q.st;
    const int tile = swe_logical_block(blockIdx.x, gridDim.x);
    if (tile >= q.n_tiles) return;                         // padding of the grid to a multiple of 8
    const int lane = (int)threadIdx.x;
    const int2 tl = q.tile[(size_t)tile*SWE_FUSE_WG + lane];
    const bool real = tl.x >= 0;
    const int k = real ? tl.x : 0;
    const int2 cnt = q.counts[tile];
    const size_t S = p.stride;
    const unsigned S8 = (unsigned)S*8u, k8 = (unsigned)k*8u;
    double *const P1 = lds + 9*SWE_FUSE_WG;

    int bmarkers = 0, bkind1 = 0;
    unsigned tr[3][1];
    double h[3], nx[3], ny[3], u[3], v[3], e[3];
    if (real) {
        int nb[3], vid[3];
        swe_conn_load(p.idxc, p.idx4, p.idx2, k, nb, vid);
        bmarkers = (nb[0] < 0 ? -nb[0] : 0) | (nb[1] < 0 ? (-nb[1]) << 8 : 0) | (nb[2] < 0 ? (-nb[2]) << 16 : 0);
        if (bmarkers != 0) {
            const int m1 = (bmarkers & 0xff) ? (bmarkers & 0xff) : ((bmarkers & 0xff00) ? ((bmarkers >> 8) & 0xff) : (bmarkers >> 16));
            bkind1 = m1 < SWE_MAX_MARKERS ? p.bc.kind[m1] : 0;
        }
        const swe_rsrc_t gu = swe_rsrc(p.uin), gv = swe_rsrc(p.uin + 3*S), ge = swe_rsrc(p.uin + 6*S);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            u[i] = swe_ld(gu, k8, i*S8);
            v[i] = swe_ld(gv, k8, i*S8);
            e[i] = swe_ld(ge, k8, i*S8);
        }
        double r0[3][6];
        bool outside[3];
#pragma unroll
        for (int f = 0; f < 3; f++) {
            const int nbf = nb[f];
            const unsigned w = ((unsigned)tl.y >> (SWE_FUSE_FBITS*f)) & 0x3ffu;
            outside[f] = (w & 0x200u) != 0u;
            const unsigned at = w & 0x1ffu;                                   // lane in the tile, or staging slot
            const int f2 = nbf >= 0 ? (nbf & 3) : f, f2a = f2 == 2 ? 0 : f2 + 1;
            {
                const unsigned ab = outside[f] ? (unsigned)(SWE_FUSE3_XG + 6*at) : (unsigned)(f2*SWE_FUSE_WG + at);
                const unsigned aa = outside[f] ? (unsigned)(SWE_FUSE3_XG + 6*at + 3) : (unsigned)(f2a*SWE_FUSE_WG + at);
                tr[f][0] = ab | (aa << 16);
            }
            const int code = outside[f] ? nbf : ((k << 2) | f);
            const unsigned kn8 = (unsigned)(code >> 2)*8u;
            const int g2 = code & 3;
            const unsigned ob = kn8 + (g2 == 0 ? 0u : (g2 == 1 ? S8 : 2u*S8));         // node g2
            const unsigned oa = kn8 + (g2 == 0 ? S8 : (g2 == 1 ? 2u*S8 : 0u));         // node (g2 + 1) % 3
            r0[f][0] = swe_ld(gu, ob, 0); r0[f][1] = swe_ld(gv, ob, 0); r0[f][2] = swe_ld(ge, ob, 0);
            r0[f][3] = swe_ld(gu, oa, 0); r0[f][4] = swe_ld(gv, oa, 0); r0[f][5] = swe_ld(ge, oa, 0);
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
#pragma unroll
        for (int i = 0; i < 3; i++) { lds[i*SWE_FUSE_WG + lane] = u[i]; lds[(3 + i)*SWE_FUSE_WG + lane] = v[i]; lds[(6 + i)*SWE_FUSE_WG + lane] = e[i]; }
#pragma unroll
        for (int f = 0; f < 3; f++) {
            if (outside[f]) {
                const unsigned at = ((unsigned)tl.y >> (SWE_FUSE_FBITS*f)) & 0x1ffu;
#pragma unroll
                for (int j = 0; j < 6; j++) lds[SWE_LDSI(SWE_FUSE3_XG + 6*at + j, SWE_FUSE3_LDS)] = r0[f][j];
            }
        }
    }
    __syncthreads();
    // ---- stage 1 on every cell of the tile: U(1) = U(0) + beta1 dt M^-1 R(U(0)), into P1 (nobody reads P1 yet: no barrier in front)
    double twoA = 0.0;
    if (real) {
        twoA = fma(nx[0], ny[1], -(ny[0]*nx[1]));
        double bu[3], bv[3], be[3], wu[3], wv[3], we[3], ou[3], ov[3], oe[3];
        swe_flow_rhs_cell<NONLIN>(p, u, v, e, h, nx, ny, bu, bv, be);
        swe_flow_rhs_facets<NONLIN, LF, SRC, 1, false, SWE_FUSE3_XG, SWE_FUSE_WG, SWE_FUSE3_LDS>(p, k, u, v, e, h, lds, tr, bmarkers, nx, ny, twoA, bu, bv, be);
#pragma unroll
        for (int i = 0; i < 3; i++) { wu[i] = 1.0*u[i]; wv[i] = 1.0*v[i]; we[i] = 1.0*e[i]; }
        swe_flow_finish<NONLIN, LF, true>(p, k, q.beta[0], u, v, e, h, nx, ny, twoA, bmarkers, bkind1, bu, bv, be, wu, wv, we, ou, ov, oe);
#pragma unroll
        for (int i = 0; i < 3; i++) { u[i] = ou[i]; v[i] = ov[i]; e[i] = oe[i]; }
#pragma unroll
        for (int i = 0; i < 3; i++) { P1[i*SWE_FUSE_WG + lane] = u[i]; P1[(3 + i)*SWE_FUSE_WG + lane] = v[i]; P1[(6 + i)*SWE_FUSE_WG + lane] = e[i]; }
    }
    __syncthreads();
    // ---- stage 2 on interior + ring 1, stage 3 on the interior: every neighbour is a cell of the tile, traces from P1; the lane's own
    //      U(0) for the weights from P0
#pragma unroll 1
    for (int s = 1; s < 3; s++) {
        const bool act = lane < (s == 1 ? cnt.y : cnt.x) && (s == 1 || k < q.cell_end);
        double ou[3], ov[3], oe[3];
        if (act) {
            // opaque to the optimiser (as in swe_flow_kernel): what the previous stage derived from the geometry would otherwise stay
            // live across the barrier, past the register budget of three waves per SIMD
#pragma unroll
            for (int i = 0; i < 3; i++) asm volatile("" : "+v"(nx[i]), "+v"(ny[i]), "+v"(h[i]));
#pragma unroll
            for (int f = 0; f < 3; f++) asm volatile("" : "+v"(tr[f][0]));
            asm volatile("" : "+v"(bmarkers), "+v"(twoA));
            double bu[3], bv[3], be[3], wu[3], wv[3], we[3];
            swe_flow_rhs_cell<NONLIN>(p, u, v, e, h, nx, ny, bu, bv, be);
            swe_flow_rhs_facets<NONLIN, LF, SRC, 1, false, SWE_FUSE3_XG, SWE_FUSE_WG, 9*SWE_FUSE_WG>(p, k, u, v, e, h, P1, tr, bmarkers, nx, ny, twoA, bu, bv, be);
            const double a0 = q.a0[s], a1 = q.a1[s];
#pragma unroll
            for (int i = 0; i < 3; i++) {
                wu[i] = fma(a0, lds[i*SWE_FUSE_WG + lane], a1*u[i]);
                wv[i] = fma(a0, lds[(3 + i)*SWE_FUSE_WG + lane], a1*v[i]);
                we[i] = fma(a0, lds[(6 + i)*SWE_FUSE_WG + lane], a1*e[i]);
            }
            swe_flow_finish<NONLIN, LF, true>(p, k, q.beta[s], u, v, e, h, nx, ny, twoA, bmarkers, bkind1, bu, bv, be, wu, wv, we, ou, ov, oe);
        }
        if (s == 1) {
            __syncthreads();                               // every lane has read its traces of U(1)
            if (act) {
#pragma unroll
                for (int i = 0; i < 3; i++) { P1[i*SWE_FUSE_WG + lane] = ou[i]; P1[(3 + i)*SWE_FUSE_WG + lane] = ov[i]; P1[(6 + i)*SWE_FUSE_WG + lane] = oe[i]; }
#pragma unroll
                for (int i = 0; i < 3; i++) { u[i] = ou[i]; v[i] = ov[i]; e[i] = oe[i]; }
            }
            __syncthreads();
        } else if (act) {
            const swe_rsrc_t gou = swe_rsrc(q.out), gov = swe_rsrc(q.out + 3*S), goe = swe_rsrc(q.out + 6*S);
#pragma unroll
            for (int i = 0; i < 3; i++) {
                swe_st(gou, k8, i*S8, ou[i]);
                swe_st(gov, k8, i*S8, ov[i]);
                swe_st(goe, k8, i*S8, oe[i]);
            }
        }
    }
}

// (Round 6, measured and removed: the three stages of a TRACER step in one launch on these two-ring tiles - the tracer in LDS between
//  the stages, the velocity traces in registers, bit for bit the stage launches.  Slower at every size: tracer only at 1 M cells 89.7
//  against 78.0 us per step, at 4 M 424.9 against 391.6 - profiles/r06d_cfgs*.txt, r06e_cfgs_4m.txt, DESIGN_ANNEX.md A8.)
//
// (Also measured and not kept: tile records that carry the cell's connectivity - 32 B per lane, two int4 - so that a lane issues its
//  loads after ONE trip to memory instead of two (tile entry, then the connectivity record): 2-6 % slower at 250 k ... 4 M cells,
//  the 12 B per cell it adds cost more than the trip it saves - profiles/r06f_fused_sizes.txt against r06b_fused_sizes.txt.)

// ---- stages 1 and 2 of a step in one launch on QUADRILATERALS (round 6): the triangle kernel's tiles with four facets per cell ------
// 256 lanes = up to 192 interior cells + their ring of at most 64 (a 16 x 12 tile of a structured mesh: ring 56); the arithmetic is
// swe_quad_stage_cell, the function the stage launches call - the same bits (tests/test_quads.py::test_fused_stage_pair_on_quad...).
// LDS: [12][256] stage values (u0..3 v0..3 e0..3 of every lane: U(0), then U(1)), the staging area of the traces from outside the tile
// ([slot][6]: u, v, e at the neighbour's node on my node f + 1, then at its node on my node f), and [12][192] = U(0) of the interior
// lanes for stage 2's Shu-Osher weights: 52.2 KB, three workgroups per CU.  At 1 M quadrilaterals the three state buffers (288 MB) do
// not fit the Infinity Cache: the two stage launches this replaces move 592 B per cell, the fused launch ~290.
#define SWE_QFUSE_INNER 192
#define SWE_QFUSE_RING (SWE_FUSE_WG - SWE_QFUSE_INNER)
#define SWE_QFUSE_XG (12*SWE_FUSE_WG)
#define SWE_QFUSE_MAX_OUT (3*SWE_QFUSE_RING)            // a ring cell has a facet towards the interior: at most three towards the outside
#define SWE_QFUSE_LDS (SWE_QFUSE_XG + 6*SWE_QFUSE_MAX_OUT)

struct SweFuseQuadArgs {
    SweStageArgs st;          // uin = U(0) (state buffer A); geometry, connectivity planes, boundary tables; dt, g, sigma_lf
    const int4 *tile;         // [n_tiles][256]: {cell or -1, facets 0-2 (10 bits each: lane of the neighbour in the tile - a boundary facet:
                              //  the lane itself - or bit 9 + staging slot), facet 3, 0}
    const int *n_inner;       // [n_tiles]
    int n_tiles;
    int cell_end;             // stage 2 updates the interior cells < cell_end
    double beta1, a0_2, a1_2, beta2;
    double *out;              // 12 planes: U(2) (state buffer C)
};

template <bool NONLIN, bool LF, bool SRC, bool AFFINE>
__global__ __launch_bounds__(SWE_FUSE_WG, SWE_FUSE_MIN_WG) void swe_fuse12_quad_kernel(const SweFuseQuadArgs q)
{
#pragma clang fp contract(off)
    __shared__ double lds[SWE_QFUSE_LDS];
    __shared__ double lu0[12*SWE_QFUSE_INNER];
    const SweStageArgs &p = q.st;
    const int tile = swe_logical_block(blockIdx.x, gridDim.x);
    if (tile >= q.n_tiles) return;
    const int lane = (int)threadIdx.x;
    const int4 tl = q.tile[(size_t)tile*SWE_FUSE_WG + lane];
    const bool real = tl.x >= 0;
    const int k = real ? tl.x : 0;
    const int n_inner = q.n_inner[tile];
    const size_t S = p.stride;
    const unsigned S8 = (unsigned)S*8u, k8 = (unsigned)k*8u, S4 = (unsigned)S*4u, k4 = (unsigned)k*4u;

    unsigned tr[4] = {0u, 0u, 0u, 0u}, bmarkers = 0u;
    int nb[4] = {-1, -1, -1, -1}, vid[4] = {0, 0, 0, 0};
    double u[4], v[4], e[4];
    if (real) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            nb[i] = swe_ldi(swe_rsrc(p.nbr), k4, i*S4);
            vid[i] = swe_ldi(swe_rsrc(p.cv), k4, i*S4);
        }
        const swe_rsrc_t gu = swe_rsrc(p.uin), gv = swe_rsrc(p.uin + 4*S), ge = swe_rsrc(p.uin + 8*S);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            u[i] = swe_ld(gu, k8, i*S8);
            v[i] = swe_ld(gv, k8, i*S8);
            e[i] = swe_ld(ge, k8, i*S8);
        }
        bmarkers = (nb[0] < 0 ? (unsigned)(-nb[0]) : 0u) | (nb[1] < 0 ? (unsigned)(-nb[1]) << 8 : 0u) |
                   (nb[2] < 0 ? (unsigned)(-nb[2]) << 16 : 0u) | (nb[3] < 0 ? (unsigned)(-nb[3]) << 24 : 0u);
        double r0[4][6];
        bool outside[4];
#pragma unroll
        for (int f = 0; f < 4; f++) {
            const int nbf = nb[f];
            const unsigned w = f < 3 ? (((unsigned)tl.y >> (SWE_FUSE_FBITS*f)) & 0x3ffu) : ((unsigned)tl.z & 0x3ffu);
            outside[f] = (w & 0x200u) != 0u;
            const unsigned at = w & 0x1ffu;                                   // lane in the tile, or staging slot
            // the neighbour traverses the shared facet backwards: its node f2 sits on my node f + 1, its node (f2 + 1) & 3 on my node f
            const int f2 = nbf >= 0 ? (nbf & 3) : f, na = (f2 + 1) & 3;
            const unsigned ab = outside[f] ? (unsigned)(SWE_QFUSE_XG + 6*at) : (unsigned)(f2*SWE_FUSE_WG + at);
            const unsigned aa = outside[f] ? (unsigned)(SWE_QFUSE_XG + 6*at + 3) : (unsigned)(na*SWE_FUSE_WG + at);
            tr[f] = ab | (aa << 16);
            // (issued for every facet: a facet inside the tile reads this cell itself, value unused - no branch around the loads)
            const int code = outside[f] ? nbf : ((k << 2) | f);
            const unsigned kn8 = (unsigned)(code >> 2)*8u;
            const int g2 = code & 3, ga = (g2 + 1) & 3;
            const unsigned ob = kn8 + ((g2 & 1) ? S8 : 0u) + ((g2 & 2) ? 2u*S8 : 0u);
            const unsigned oa = kn8 + ((ga & 1) ? S8 : 0u) + ((ga & 2) ? 2u*S8 : 0u);
            r0[f][0] = swe_ld(gu, ob, 0); r0[f][1] = swe_ld(gv, ob, 0); r0[f][2] = swe_ld(ge, ob, 0);
            r0[f][3] = swe_ld(gu, oa, 0); r0[f][4] = swe_ld(gv, oa, 0); r0[f][5] = swe_ld(ge, oa, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) { lds[i*SWE_FUSE_WG + lane] = u[i]; lds[(4 + i)*SWE_FUSE_WG + lane] = v[i]; lds[(8 + i)*SWE_FUSE_WG + lane] = e[i]; }
#pragma unroll
        for (int f = 0; f < 4; f++) {
            if (outside[f]) {
                const unsigned at = (f < 3 ? ((unsigned)tl.y >> (SWE_FUSE_FBITS*f)) : (unsigned)tl.z) & 0x1ffu;
#pragma unroll
                for (int j = 0; j < 6; j++) lds[SWE_LDSI(SWE_QFUSE_XG + 6*at + j, SWE_QFUSE_LDS)] = r0[f][j];
            }
        }
    }
    __syncthreads();
    // the six traces of facet f from LDS: component c is 4 planes further inside the tile, one double further in the staging area
#define SWE_QFUSE_TRACES(una, unb, vna, vnb, ena, enb) \
    _Pragma("unroll") \
    for (int f = 0; f < 4; f++) { \
        const unsigned ab = tr[f] & 0xffffu, aa = tr[f] >> 16; \
        const unsigned step = ab >= (unsigned)SWE_QFUSE_XG ? 1u : (unsigned)(4*SWE_FUSE_WG); \
        unb[f] = lds[SWE_LDSI(ab, SWE_QFUSE_LDS)]; vnb[f] = lds[SWE_LDSI(ab + step, SWE_QFUSE_LDS)]; enb[f] = lds[SWE_LDSI(ab + 2u*step, SWE_QFUSE_LDS)]; \
        una[f] = lds[SWE_LDSI(aa, SWE_QFUSE_LDS)]; vna[f] = lds[SWE_LDSI(aa + step, SWE_QFUSE_LDS)]; ena[f] = lds[SWE_LDSI(aa + 2u*step, SWE_QFUSE_LDS)]; \
    }
    // ---- stage 1 on every cell of the tile: U(1) = U(0) + beta1 dt M^-1 R(U(0))
    double o1u[4], o1v[4], o1e[4];
    if (real) {
        double una[4], unb[4], vna[4], vnb[4], ena[4], enb[4];
        SWE_QFUSE_TRACES(una, unb, vna, vnb, ena, enb)
        swe_quad_stage_cell<NONLIN, LF, false, SRC, false, AFFINE, true>(p, k, k8, S8, nb, vid, bmarkers, u, v, e, una, unb, vna, vnb, ena, enb,
                                                                        0.0, 1.0, q.beta1, lds, SWE_FUSE_WG, lane, nullptr, 0, o1u, o1v, o1e);
    }
    __syncthreads();                                       // every lane has read its traces (and its boundary facets' inputs) of U(0)
    const bool act2 = lane < n_inner && k < q.cell_end;
    if (real) {
#pragma unroll
        for (int i = 0; i < 4; i++) { lds[i*SWE_FUSE_WG + lane] = o1u[i]; lds[(4 + i)*SWE_FUSE_WG + lane] = o1v[i]; lds[(8 + i)*SWE_FUSE_WG + lane] = o1e[i]; }
    }
    if (act2) {
#pragma unroll
        for (int i = 0; i < 4; i++) { lu0[i*SWE_QFUSE_INNER + lane] = u[i]; lu0[(4 + i)*SWE_QFUSE_INNER + lane] = v[i]; lu0[(8 + i)*SWE_QFUSE_INNER + lane] = e[i]; }
    }
    __syncthreads();
    // ---- stage 2 on the interior cells (every neighbour is a cell of the tile): U(2) = a0 U(0) + a1 U(1) + beta2 dt M^-1 R(U(1))
    if (act2) {
#pragma unroll
        for (int f = 0; f < 4; f++) asm volatile("" : "+v"(tr[f]));
        asm volatile("" : "+v"(bmarkers));
        double una[4], unb[4], vna[4], vnb[4], ena[4], enb[4], ou[4], ov[4], oe[4];
        SWE_QFUSE_TRACES(una, unb, vna, vnb, ena, enb)
        swe_quad_stage_cell<NONLIN, LF, true, SRC, false, AFFINE, true>(p, k, k8, S8, nb, vid, bmarkers, o1u, o1v, o1e, una, unb, vna, vnb, ena, enb,
                                                                       q.a0_2, q.a1_2, q.beta2, lds, SWE_FUSE_WG, lane, lu0, SWE_QFUSE_INNER, ou, ov, oe);
        const swe_rsrc_t gou = swe_rsrc(q.out), gov = swe_rsrc(q.out + 4*S), goe = swe_rsrc(q.out + 8*S);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            swe_st(gou, k8, i*S8, ou[i]);
            swe_st(gov, k8, i*S8, ov[i]);
            swe_st(goe, k8, i*S8, oe[i]);
        }
    }
#undef SWE_QFUSE_TRACES
}
