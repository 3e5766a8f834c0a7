// swe2d_mflow.h - the dataflow stage loop of swe2d_flow.h for cell ranges LARGER than the device holds resident as one block per
// wave: a wave owns K consecutive 64-cell blocks of the flow order and walks them stage after stage (round 5; VERDICT r04 items
// "missing 4" / "next 4": ranks of two and four of the 1 M-triangle bench mesh, 250-500 k cells, ran stage launches at
// 1.7x / 2.8x of one device).
//
// What it replaces: thetis/rungekutta.py:930-952 (solve_stage i on the whole mesh, then i + 1) under mpiexec -n 2 / -n 4
// (examples/README.md:51-56), as swe2d_flow.h does for ranks of eight.
//
// What is different from swe_flow_kernel:
//   * the stage values of a block cannot stay in registers from stage to stage (the wave works on its other blocks in between):
//     they live where the stage LAUNCHES keep them - the three state buffers, A -> B -> C -> A per time step (include/swe2d.h,
//     swe2d_solve_stage) - and every (stage, block) visit loads the block's own nodal values, its connectivity and geometry again
//     (L2 hits: the wave wrote or read them K - 1 visits ago) and stores its result; U(0) of the running step is re-read from buffer A
//     in stages 2 and 3 like swe_stage_kernel<..., HASU0> does.  Per visit that is the traffic of a stage launch; what goes away is
//     the launch boundary (1.5-1.9 us), the quarter-full last round of a 250 k-cell launch and the lock step of its waves;
//   * neighbour traces: inside the block through LDS, across block rims through the tagged granules of swe2d_flow.h - same slots,
//     same tags, same check word, same polling pass - also between two blocks of the SAME wave (the granules of the block visited
//     just before are in L2 by then; no special case);
//   * consecutive blocks per wave: block j of wave w is flow block w K + j.  A block's rim neighbours are mostly the blocks before
//     and after it in the flow order, i.e. visited by the same wave shortly before / after: only the first and last block of a wave
//     wait for another wave, and that wave published what they need K - 1 visits ago.  (Interleaved ownership - block j of wave w =
//     j W + w - would put every block's neighbours into other waves at the same visit: the chain of swe2d_flow.h, K times per stage.)
//   * no deadlock: visit (s, j) of a wave needs visits (s - 1, *) of its neighbours only; by induction on s every wave gets through
//     stage s - 1 with all its blocks before anybody has to wait at stage s for good.  Slot reuse is as in swe2d_flow.h (a producer
//     reaches the end of stage s + 2 only after the consumer has read stage s).  All waves must be resident: the host sizes K so
//     that they are (launch_flow).
//
// No exchange inside the launch (FX): a partition's cycle is one launch of this kernel followed by the push and unpack kernels of
// swe2d_p2p.h, like `_cycle_swe_flow` with the one-block kernel.
//
// Arithmetic: swe_flow_rhs_cell / swe_flow_rhs_facets / swe_flow_finish, i.e. swe_stage_kernel<..., BINL>'s operation for operation:
// bit for bit the stage launches (tests/test_gpu_flow_kernel.py).
#pragma once
#include "swe2d_flow.h"

#ifndef SWE_MFLOW_OCCUPANCY
#define SWE_MFLOW_OCCUPANCY __attribute__((amdgpu_waves_per_eu(2, 3)))
#endif

template <bool NONLIN, bool LF, bool SRC, int POLL = 8>
__global__ __launch_bounds__(SWE_BLOCK) SWE_MFLOW_OCCUPANCY void swe_mflow_kernel(const SweFlowArgs q)
{
#pragma clang fp contract(off)
    __shared__ double lds[SWE_FLOW_LDS_DOUBLES];
    __shared__ int xsrc[SWE_FLOW_MAX_RIM];
    __shared__ int lact[SWE_BLOCK];
    __shared__ unsigned char lpub[SWE_FLOW_MAX_RIM];
    const SweStageArgs &p = q.st;
    const int K = q.blocks_per_wave;
    const int n_waves = (q.n_blocks + K - 1)/K;
    const int w = swe_logical_block(blockIdx.x, gridDim.x);
    if (w >= n_waves) return;                                  // padding of the grid to a multiple of 8
    const int lane = (int)threadIdx.x;
    const size_t S = p.stride;
    const unsigned S8 = (unsigned)S*8u;
    const int b0 = w*K, b1 = min(b0 + K, q.n_blocks);
    // all blocks' stage counters are equal between launches (swe2d_flow.h): one read serves the wave's blocks
    const unsigned base = q.flag[(size_t)b0*SWE_FLOW_FLAG_STRIDE];
    const unsigned fin = base + (unsigned)q.n_stages;
    const __amdgpu_buffer_rsrc_t rex = __builtin_amdgcn_make_buffer_rsrc(q.ex, 0, 3*q.parity_bytes, 0x00020000);
    const swe_rsrc_t rvx = swe_rsrc(p.vx), rvy = swe_rsrc(p.vy), rvh = swe_rsrc(p.vh);
    unsigned long long t_start = 0ull;
    bool late = false;

#pragma unroll 1
    for (int s = 0; s < q.n_stages; s++) {
        const int i3 = s % 3;
        const int end_s = q.cell_end[s];
        // buffers of the stage: A -> B -> C -> A (swe2d_solve_stage); U(0) = A
        const double *bin = q.buf[i3 == 0 ? 0 : i3], *bout_c = q.buf[(i3 + 1) % 3];
        double *bout = const_cast<double *>(bout_c);
        const swe_rsrc_t gu = swe_rsrc(bin), gv = swe_rsrc(bin + 3*S), ge = swe_rsrc(bin + 6*S);
        const swe_rsrc_t g0u = swe_rsrc(q.buf[0]), g0v = swe_rsrc(q.buf[0] + 3*S), g0e = swe_rsrc(q.buf[0] + 6*S);
        const swe_rsrc_t gou = swe_rsrc(bout), gov = swe_rsrc(bout + 3*S), goe = swe_rsrc(bout + 6*S);
#pragma unroll 1
        for (int lb = b0; lb < b1; lb++) {
            const int kcode = q.fcell[lb*SWE_BLOCK + lane];
            const bool real = kcode >= 0;
            const int k = real ? kcode : -1 - kcode;
            const bool act = real && k < end_s;
            if (!__any(act)) continue;                         // (ranges only shrink: the block is done for this launch)
            const unsigned k8 = (unsigned)k*8u;
            const int2 myslots = q.xblk[lb];
            const int nrim = myslots.y;
            __syncthreads();                                   // the previous visit's LDS reads are over
            for (int i = lane; i < nrim; i += SWE_BLOCK) xsrc[SWE_LDSI(i, SWE_FLOW_MAX_RIM)] = q.xsrc[myslots.x + i];
            // ---- the block's connectivity, exchange slots, geometry and stage input
            int bmarkers, bkind1 = 0;
            unsigned tr[3][3];
            int xown[3];
            double h[3], nx[3], ny[3], u[3], v[3], e[3];
            double wu[3], wv[3], we[3];
            {
                const int4 q4 = p.idx4[k];
                const int2 q2 = p.idx2[k];
                const int4 x4 = q.xo4[lb*SWE_BLOCK + lane];
                const int2 x2 = q.xo2[lb*SWE_BLOCK + lane];
                const int nb[3] = {q4.x, q4.y, q4.z};
                const int vid[3] = {q4.w, q2.x, q2.y};
                const int xin[3] = {x4.w, x2.x, x2.y};
                xown[0] = x4.x; xown[1] = x4.y; xown[2] = x4.z;
                bmarkers = (nb[0] < 0 ? -nb[0] : 0) | (nb[1] < 0 ? (-nb[1]) << 8 : 0) | (nb[2] < 0 ? (-nb[2]) << 16 : 0);
                if (bmarkers != 0) {
                    const int m1 = (bmarkers & 0xff) ? (bmarkers & 0xff) : ((bmarkers & 0xff00) ? ((bmarkers >> 8) & 0xff) : (bmarkers >> 16));
                    bkind1 = m1 < SWE_MAX_MARKERS ? p.bc.kind[m1] : 0;
                }
#pragma unroll
                for (int i = 0; i < 3; i++) {             // (past the L1: this wave stored them one stage ago, and read the line before that)
                    u[i] = swe_ld_l2(gu, k8, i*S8);
                    v[i] = swe_ld_l2(gv, k8, i*S8);
                    e[i] = swe_ld_l2(ge, k8, i*S8);
                }
                // w = a0*U(0) + a1*U_in (the first stage of a step has no U(0) term)
                const double a0 = q.a0[i3], a1 = q.a1[i3];
#pragma unroll
                for (int i = 0; i < 3; i++) { wu[i] = a1*u[i]; wv[i] = a1*v[i]; we[i] = a1*e[i]; }
                if (i3 > 0) {
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        wu[i] = fma(a0, swe_ld_l2(g0u, k8, i*S8), wu[i]);
                        wv[i] = fma(a0, swe_ld_l2(g0v, k8, i*S8), wv[i]);
                        we[i] = fma(a0, swe_ld_l2(g0e, k8, i*S8), we[i]);
                    }
                }
                double r0[3][6];                               // first stage of the launch: the rim traces come from the state planes
#pragma unroll
                for (int f = 0; f < 3; f++) {
                    const int nbf = nb[f];
                    const bool rim = xown[f] >= 0;
                    const bool inw = nbf >= 0 && !rim;
                    const int ls = rim ? lane : xin[f];
                    const int f2 = inw ? (nbf & 3) : f, f2a = f2 == 2 ? 0 : f2 + 1;
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const unsigned ab = rim ? (unsigned)(SWE_FLOW_XG + 6*xin[f] + c) : (unsigned)((3*c + f2)*SWE_BLOCK + ls);
                        const unsigned aa = rim ? (unsigned)(SWE_FLOW_XG + 6*xin[f] + 3 + c) : (unsigned)((3*c + f2a)*SWE_BLOCK + ls);
                        tr[f][c] = ab | (aa << 16);
                    }
                    if (s == 0) {
                        const int code = rim ? nbf : ((k << 2) | f);
                        const unsigned kn8 = (unsigned)(code >> 2)*8u;
                        const int g2 = code & 3;
                        const unsigned ob = kn8 + (g2 == 0 ? 0u : (g2 == 1 ? S8 : 2u*S8));
                        const unsigned oa = kn8 + (g2 == 0 ? S8 : (g2 == 1 ? 2u*S8 : 0u));
                        r0[f][0] = swe_ld(gu, ob, 0); r0[f][1] = swe_ld(gv, ob, 0); r0[f][2] = swe_ld(ge, ob, 0);
                        r0[f][3] = swe_ld(gu, oa, 0); r0[f][4] = swe_ld(gv, oa, 0); r0[f][5] = swe_ld(ge, oa, 0);
                    }
                }
                double px[3], py[3];
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
                if (s == 0) {
#pragma unroll
                    for (int f = 0; f < 3; f++) {
                        if (xown[f] >= 0) {
#pragma unroll
                            for (int j = 0; j < 6; j++) lds[SWE_LDSI(SWE_FLOW_XG + 6*xin[f] + j, SWE_FLOW_LDS_DOUBLES)] = r0[f][j];
                        }
                    }
                }
            }
            // ---- the block's stage values for its own lanes
#pragma unroll
            for (int i = 0; i < 3; i++) { lds[SWE_LDSI(i*SWE_BLOCK + lane, SWE_FLOW_LDS_DOUBLES)] = u[i]; lds[SWE_LDSI((3 + i)*SWE_BLOCK + lane, SWE_FLOW_LDS_DOUBLES)] = v[i]; lds[SWE_LDSI((6 + i)*SWE_BLOCK + lane, SWE_FLOW_LDS_DOUBLES)] = e[i]; }
            lact[SWE_LDSI(lane, SWE_BLOCK)] = act ? 1 : 0;
            double bu[3], bv[3], be[3];
            const double twoA = fma(nx[0], ny[1], -(ny[0]*nx[1]));
            swe_flow_rhs_cell<NONLIN>(p, u, v, e, h, nx, ny, bu, bv, be);
            // ---- traces across the rim: the granules of stage s - 1 (swe2d_flow.h: tags, check word, polling pass)
            if (s > 0) {
                const unsigned need = base + (unsigned)(s - 1) + 1u;
                const unsigned par = ((unsigned)(s - 1) & 1u)*q.parity_bytes;
                __syncthreads();
                for (unsigned spins = 0;; spins++) {
                    bool ok = true;
                    for (int c0 = 0; c0 < 8*nrim; c0 += POLL*SWE_BLOCK) {
                        swe_u32x4 gr[POLL];
                        unsigned poff[POLL];
                        int ent_[POLL], act_[POLL];
#pragma unroll
                        for (int j = 0; j < POLL; j++) ent_[j] = xsrc[SWE_LDSI(min((c0 + j*SWE_BLOCK + lane) >> 3, SWE_FLOW_MAX_RIM - 1), SWE_FLOW_MAX_RIM)];
#pragma unroll
                        for (int j = 0; j < POLL; j++) act_[j] = lact[SWE_LDSI(ent_[j] & (SWE_BLOCK - 1), SWE_BLOCK)];
#pragma unroll
                        for (int j = 0; j < POLL; j++) {
                            const int t = c0 + j*SWE_BLOCK + lane;
                            poff[j] = (t < 8*nrim && act_[j]) ? (unsigned)(ent_[j] >> 6)*SWE_FLOW_SLOT_BYTES + 16u*(unsigned)(t & 7) + par : SWE_FLOW_NOWHERE;
                        }
#pragma unroll
                        for (int j = 0; j < POLL; j++) gr[j] = swe_flow_get(rex, poff[j]);
#pragma unroll
                        for (int j = 0; j < POLL; j++) {
                            const int t = c0 + j*SWE_BLOCK + lane;
                            if (poff[j] != SWE_FLOW_NOWHERE) {
                                ok = ok && swe_flow_arrived(gr[j], need);
                                if ((t & 7) < 6) lds[SWE_LDSI(SWE_FLOW_XG + 6*(t >> 3) + (t & 7), SWE_FLOW_LDS_DOUBLES)] = swe_flow_val(gr[j]);
                            }
                        }
                    }
                    if (__all(ok) || late) break;
                    __builtin_amdgcn_s_sleep(2);
                    if ((spins & 31u) == 31u) {
                        const unsigned long long now = wall_clock64();
                        if (t_start == 0ull) t_start = now;
                        else if (now - t_start > q.timeout_ticks) {
                            late = true;
                            if (lane == 0 && atomicAdd(q.status, 1u) == 0u) q.status[1] = (unsigned)lb + 1u;
                        }
                    }
                }
                t_start = 0ull;
            }
            __syncthreads();
            double ou[3], ov[3], oe[3];
            swe_flow_rhs_facets<NONLIN, LF, SRC, 3>(p, k, u, v, e, h, lds, tr, bmarkers, nx, ny, twoA, bu, bv, be);
            swe_flow_finish<NONLIN, LF, !SRC>(p, k, q.beta[i3], u, v, e, h, nx, ny, twoA, act ? bmarkers : 0, bkind1, bu, bv, be, wu, wv, we, ou, ov, oe);
            // ---- publish the rim traces of this stage's result (nobody reads the last stage of the launch)
            if (s + 1 < q.n_stages) {
                const unsigned tag_ = base + (unsigned)s + 1u;
                const unsigned par_ = ((unsigned)s & 1u)*q.parity_bytes + (unsigned)myslots.x*SWE_FLOW_SLOT_BYTES;
                __syncthreads();
#pragma unroll
                for (int f = 0; f < 3; f++) {
                    if (xown[f] >= 0 && act) {
                        const int a_ = f, b_ = (f + 1) % 3;
                        double *d_ = lds + SWE_LDSI(SWE_FLOW_XG + 6*xown[f] + 5, SWE_FLOW_LDS_DOUBLES) - 5;
                        d_[0] = ou[a_]; d_[1] = ov[a_]; d_[2] = oe[a_]; d_[3] = ou[b_]; d_[4] = ov[b_]; d_[5] = oe[b_];
                    }
                    if (xown[f] >= 0) lpub[SWE_LDSI(xown[f], SWE_FLOW_MAX_RIM)] = act ? 1 : 0;
                }
                __syncthreads();
                for (int t_ = lane; t_ < 8*nrim; t_ += SWE_BLOCK) {
                    const int gi_ = t_ & 7;
                    const double x_ = gi_ < 6 ? lds[SWE_LDSI(SWE_FLOW_XG + 6*(t_ >> 3) + gi_, SWE_FLOW_LDS_DOUBLES)] : 0.0;
                    if (lpub[SWE_LDSI(t_ >> 3, SWE_FLOW_MAX_RIM)]) swe_flow_put(rex, par_ + 16u*(unsigned)t_, x_, tag_);
                }
            }
            // ---- the stage result: buffer B / C / A, read by this wave's next visit of the block (and by later launches)
            if (act) {
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    swe_st(gou, k8, i*S8, ou[i]);
                    swe_st(gov, k8, i*S8, ov[i]);
                    swe_st(goe, k8, i*S8, oe[i]);
                }
            }
        }
    }
    // every block's counter ends the launch at base + n_stages
    for (int lb = b0 + lane; lb < b1; lb += SWE_BLOCK) q.flag[(size_t)lb*SWE_FLOW_FLAG_STRIDE] = fin;
}
