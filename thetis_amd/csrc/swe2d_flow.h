// swe2d_flow.h - many SSPRK33 stages of the triangle DG-P1 shallow water equations in ONE launch, without a grid barrier:
// a dataflow stage loop for cell ranges whose 64-cell blocks are all resident at once (gfx950; <= ~130-190 k cells).
//
// What it replaces: the reference runs solve_stage(i) on the whole mesh and then solve_stage(i + 1) (thetis/rungekutta.py:930-952),
// and under mpiexec every par_loop is preceded by a halo exchange (examples/README.md:51-56).  On one rank of eight of the 1 M
// triangle bench mesh (125 k cells = 1954 one-wave workgroups, two per SIMD) a stage LAUNCH is latency, not bandwidth: 0.5 us
// index loads + 2.5 us loads + 2.1 us arithmetic + 0.3 us stores in lock step, a tail of boundary waves, 1.5-1.9 us of kernel
// boundary and a reload of the state into L2s that the boundary invalidated.  Here
//
//   * one wave OWNS one 64-cell block of the device numbering for the whole launch: connectivity, geometry, the block's stage
//     values and its U(0) stay in registers from stage to stage; state planes are read once at the start of the launch and
//     written once per time step (the step result, buffer 0);
//   * traces of neighbours INSIDE the block (~85 % with the tile numbering) go through LDS: the wave publishes its nine nodal
//     values per cell there at the top of a stage and reads its neighbours' - no memory, no waiting;
//   * traces of neighbours in OTHER blocks travel as 16-byte granules {value, stage tag, check word}: a facet on the block's rim
//     owns a slot (host-built numbering: a block's slots are contiguous and grouped by the block they face), its six trace values
//     are stored there after every stage, each granule by ONE 16-byte `sc1` (write-through) store of one lane, tag = stage counter;
//     the block across the rim re-reads the granules with `sc1` loads (past its CU's L1) until all carry the tag of the stage
//     it needs.  The data IS the flag:
//     no drain (s_waitcnt vmcnt(0)) before a flag store, no flag store, no separate poll before a dependent gather - measured
//     on the first version of this kernel (stage values through the state planes + one stage counter per block) those three
//     hops were 2.1 of the 4.0 us a stage took with one block per compute unit.  A naturally aligned 16-byte store of one lane
//     was observed to land untorn on gfx950 (MI355X_MICROARCH.md, "observed untorn, also for 16-B sc1 halves; not an
//     architectural guarantee"); the protocol does not rely on it: the check word (tag ^ lo32 ^ hi32 of the value, round 5) ties
//     the tag to the value it was stored with, and a granule whose halves come from two different stores is re-polled like one
//     that has not arrived (swe_flow_arrived; adversary build -DSWE_FLOW_TEAR).  Two slots per facet alternate by stage parity:
//     the producer overwrites the values of stage s at the end of stage s + 2, which it can only reach after the consumer has
//     finished stage s + 1, i.e. has read them.
//     Both directions are coalesced through an LDS staging area: the rim lanes drop their values there by slot and the wave
//     writes the block's whole slot range with consecutive lanes on consecutive granules (full 128-byte lines); the chunk a
//     neighbour block wrote for this block is contiguous too, the wave reads it the same way and the rim lanes pick their
//     traces from LDS.  (One 16-byte access per lane and granule is one partial-line request each: 216 per block, stage and
//     polling pass instead of ~36 lines - measured 3 us just to issue the stores at 125 k cells, and the polling passes of
//     eight waves saturate a compute unit's memory pipeline.)
//   * no grid barrier and no per-block flag wait: a block starts stage s + 1 as soon as the granules of ITS rim facets have
//     arrived, so the load phase of some blocks overlaps the arithmetic of others.
//
// Placement-independent (MI355X_MICROARCH.md, inter-workgroup visibility): every shared word is written with sc1 stores and
// read with sc1 loads; the XCD-chunked block map (swe_logical_block) is used for speed only.
//
// Deadlock freedom needs every block of the launch resident: the host launches this kernel only when the grid fits the
// occupancy the runtime reports (swe2d_api_flow.hip: flow_capacity), and every spin is bounded by the wall clock - a timeout is
// counted in the status word, the wave carries on (the result is then wrong and the host reports SWE2D_ERR_HIP at the next
// synchronisation point), it never hangs the device.
//
// Stage tags never need re-initialisation: they count stages over ALL launches of the handle.  Every block keeps the count in
// its own word (flag[block]); a block that retires early (its cells are outside the range of the remaining stages) or takes
// part in no stage advances its word by n_stages as well, so all words stay equal and the grid always covers ALL blocks.
//
// Ranges: stage s updates the cells [0, cell_end[s]), non-increasing, and every cell of stage s + 1's range must have its three
// facet neighbours inside stage s's range - the shrinking ranges of a halo-exchange cycle (partition.py: stage_range), or the
// whole mesh throughout.  After the launch state buffer 0 holds what the stage launches would have left there; buffers 1, 2
// (the intermediate stage solutions) are not written.
//
// The arithmetic is that of swe_stage_kernel<NONLIN, LF, ., SRC, false, false, true(BINL)>, operation for operation and under
// the same `fp contract(off)`: bit for bit the result of the stage launches on the same ranges (tests/test_gpu_flow_kernel.py).
#pragma once
#include "swe2d_kernels.h"
#include "swe2d_p2p.h"

#ifndef SWE_FLOW_OCCUPANCY
#define SWE_FLOW_OCCUPANCY __attribute__((amdgpu_waves_per_eu(2, 2)))      // <= 256 VGPRs: two one-wave workgroups per SIMD
#endif
#define SWE_FLOW_MAX_STAGES 384            // 128 time steps per launch
#define SWE_FLOW_MAX_CYCLES 64             // exchange cycles per launch (FX kernels)
#ifndef SWE_FLOW_FLAG_STRIDE
#define SWE_FLOW_FLAG_STRIDE 16           // unsigned words between two blocks' stage counters (64 B)
#endif
#define SWE_FLOW_SLOT_BYTES 128            // exchange slot of one rim facet and stage parity: 8 granules of 16 B {value, tag, check} (6 values + 2 pads)
// t / 6 for 0 <= t < 2^14 (a polling pass visits the SIX value granules of every incoming 128-byte slot, lane after lane: the two
// padding granules that make a slot one full line for the producer's write-through stores are never read - 32 rim facets per three
// load instructions, where one load per granule of the slot needed four, round 5)
#define SWE_FLOW_DIV6(t_) ((int)(((unsigned)(t_)*43691u) >> 18))
#ifndef SWE_FLOW_POLL_SLEEP
#define SWE_FLOW_POLL_SLEEP 2              // s_sleep (x 64 cycles) between two polling passes of a block that is still waiting
#endif
#ifndef SWE_FLOW_NTR_SRC
#define SWE_FLOW_NTR_SRC 3                 // (SRC ? 1 : 3): the trace addresses of the source-term variants packed, see swe_flow_rhs_facets
#endif
#ifndef SWE_FLOW_WALLFAST_SRC
#define SWE_FLOW_WALLFAST_SRC true         // the closed-wall fast path of swe_boundary_facet also in the variants with source terms
#endif
#ifndef SWE_FLOW_RX
#define SWE_FLOW_RX 3                      // FX receive: granule loads per lane in flight in a trip of a pass
#endif
#define SWE_FLOW_MAX_RIM 160               // rim facets of a block (3*64 at worst: such flow orders are refused, see flow_build)

typedef unsigned int swe_u32x4 __attribute__((ext_vector_type(4)));
#define SWE_FLOW_NOWHERE 0x80000000u       // byte offset beyond the exchange array (< 2 GiB): a load returns zeros without touching memory
// LDS layout of a block: nine planes of 64 stage values, then six trace values per incoming rim facet
#define SWE_FLOW_XG (9*SWE_BLOCK)
#define SWE_FLOW_LDS_DOUBLES (SWE_FLOW_XG + 6*SWE_FLOW_MAX_RIM)
// LDS indices are computed from host-built tables (xo4 / xo2 / xsrc, flow_build): the -DSWE_RANGE_CHECK build (tools/range_check.sh)
// tests every one of them against the array it indexes; a violation is reported like an out-of-range memory access and redirected
// to element 0.
#ifdef SWE_RANGE_CHECK
__device__ __forceinline__ unsigned swe_lds_index_chk(unsigned i, unsigned n, int line)
{
    if (i < n) return i;
    if (atomicAdd(&swe_chk_report[0], 1ull) == 0ull) { swe_chk_report[1] = 0x1d5000000000ull | i; swe_chk_report[2] = (unsigned long long)line; }
    return 0u;
}
#define SWE_LDSI(i, n) swe_lds_index_chk((unsigned)(i), (unsigned)(n), __LINE__)
#else
#define SWE_LDSI(i, n) (i)
#endif

// -DSWE_FLOW_DELAY (tools/range_check.sh builds it next to the range-checked library): an adversary for the granule protocol.
// One chosen block sleeps `ticks` of the 100 MHz wall clock (two to three stage periods) at a chosen point of every `every`-th
// stage - before its polling pass (1), before it publishes (2), before an FX receive (4), before an FX push (8) - so that its
// neighbours run ahead as far as the protocol lets them, or wait for it as long as it takes.  The slot-parity argument (a slot of
// stage s is overwritten at the end of stage s + 2, which its producer cannot reach before every consumer that still waits for
// has read it) and the "push n + 2 only after receive n + 1" argument of the FX exchange say the result cannot change; the tests
// compare it bit for bit with the stage launches (tests/test_gpu_flow_kernel.py, tests/test_distributed.py).
#ifdef SWE_FLOW_DELAY
__device__ int swe_flow_delay[4] = {-1, 0, 0, 1};              // block, where (bit mask), ticks, every n-th stage
#define SWE_FLOW_DELAY_AT(w) do {                                                                                            \
        if (swe_flow_delay[0] == lb && (swe_flow_delay[1] & (w)) && (swe_flow_delay[3] <= 1 || (s % swe_flow_delay[3]) == 0)) {    \
            const unsigned long long t0_ = wall_clock64();                                                                   \
            while (wall_clock64() - t0_ < (unsigned long long)swe_flow_delay[2]) __builtin_amdgcn_s_sleep(64);                \
        }                                                                                                                    \
    } while (0)
#else
#define SWE_FLOW_DELAY_AT(w) do { } while (0)
#endif

struct SweFlowArgs {
    SweStageArgs st;                       // geometry, connectivity, boundary tables, sources; uin/u0/uout/a0/a1/beta/cell_* unused
    double *buf[3];                        // state buffers A (U0 / step result), B, C (B, C are not touched)
    unsigned *flag;                        // [n_blocks][SWE_FLOW_FLAG_STRIDE] stages counted by the block over all launches
    unsigned *status;                      // [0] timeouts, [1] first block that timed out + 1
    const int *fcell;                      // [n_blocks*64] the cell of every flow position (block*64 + lane); < 0: padding lane, mimics cell -1 - x
    const int4 *xo4;                       // per position, counted from the block's first slot: {my slot of facet 0, 1, 2 (-1: not a rim facet),
    const int2 *xo2;                       //  w0}, {w1, w2}: w = place of the facet's incoming slot in the block's incoming list (rim facet) or the
                                           //  lane of the neighbour inside the block (the lane itself for a boundary facet)
    const int2 *xblk;                      // per block {first slot, number of slots}: a block's slots are contiguous
    const int *xsrc;                       // [n_slots] incoming list of every block at its own slot range: entry i = (slot the
                                           //  neighbour block writes for my i-th incoming facet) << 6 | lane of my cell that reads it
    void *ex;                              // [3 slot sets][n_slots] exchange slots, SWE_FLOW_SLOT_BYTES each: two stage parities + the cycle inputs (FX)
    unsigned parity_bytes;                 // n_slots * SWE_FLOW_SLOT_BYTES
    int n_blocks;                          // blocks of the handle (cells rounded up to 64)
    int n_stages;                          // a multiple of 3: stage s is Shu-Osher stage s % 3
    int cell_end[SWE_FLOW_MAX_STAGES];     // stage s updates the cells [0, cell_end[s]); non-increasing
    double a0[3], a1[3], beta[3];          // Shu-Osher weights per stage (swe2d_ssprk33_coefficients)
    unsigned long long timeout_ticks;      // wall_clock64 ticks (100 MHz)
    // ---- FX kernels: the halo exchange of a partition inside the launch, cell by cell through tagged granules in the peers'
    //      landing zones (a peer-to-peer channel of swe2d_p2p.h of width 18 = nine granules per cell; its epoch flags are not
    //      used).  The launch runs n_cycles exchange cycles of stages_per_cycle stages (n_stages = their product; cell_end[g] holds
    //      the ranges of ONE cycle): a cycle starts with the receive of the previous cycle's push (every ghost lane waits for the
    //      nine granules of ITS cell to carry the push number), ends with the push of the send cells into the peers' zones.
    int n_cycles, stages_per_cycle;
    const int2 *xsend;                     // per position: the cell's (up to two) places in the send list, -1: none
    const int *xrecv;                      // per position: the cell's place in the receive list, -1: not a ghost cell
    unsigned *xtick;                       // arrival counter of the launch's blocks (the last one advances the epochs)
    SweP2pCounters *xctr;                  // pushes made / received so far (the granule channel's counters)
    int x_n_peers;
    int x_off[SWE_P2P_MAX_PEERS];                                            // per peer: start of its segment of the send list (cells)
    void *x_rdata[SWE_P2P_MAX_PEERS];                                        // peer's landing segment for me, slot 0
    unsigned x_rbytes[SWE_P2P_MAX_PEERS];                                    // its size in bytes (both slots)
    unsigned x_rslot[SWE_P2P_MAX_PEERS];                                     // bytes between the peer's slot 0 and slot 1
    void *x_zone;                                                            // my landing data, slot 0
    unsigned x_zbytes, x_slot;                                               // its size (both slots), bytes between slot 0 and slot 1
    unsigned long long x_timeout;                                            // wall_clock64 ticks
};

// one granule: {value, tag, check} written / read by ONE 16-byte access of one lane, sc1 (aux 16): write-through / past the L1.
// check = tag ^ lo32(value) ^ hi32(value).  An untorn 16-byte store of one lane is what gfx950 was OBSERVED to do, on one device; it
// is not an architectural guarantee, and across devices (the FX exchange: a peer's IPC-mapped zone over xGMI) it had never run at
// all.  The consumer therefore takes a granule only when the tag is the one it waits for AND the check word agrees
// with the value it travelled with: a store that lands in two halves (value before tag or tag before value, 8 + 8 or 4 + 12 bytes)
// shows either an old tag - not taken anyway - or a new tag whose check word belongs to another value - re-polled like a granule
// that has not arrived.  (An old value with the new tag passes only if lo ^ hi of the two values agree, i.e. never for a changed
// value by less than one chance in 2^32 - and a granule whose value did not change is right whichever half is read.)  Three VALU
// operations on either side.  -DSWE_FLOW_TEAR (tools/range_check.sh) is the adversary that splits the stores on purpose;
// -DSWE_FLOW_NOCHECK its negative control (the consumer ignores the check word: the torn build must then give wrong bits).
__device__ __forceinline__ unsigned swe_flow_check_word(unsigned lo, unsigned hi, unsigned tag) { return tag ^ lo ^ hi; }
#ifdef SWE_FLOW_TEAR
// block (-2: every block), ticks of the 100 MHz clock between the two halves of a store, every n-th publish, 1 = also the pushes across ranks
__device__ int swe_flow_tear[4] = {-1, 0, 1, 0};
__device__ __forceinline__ void swe_flow_put_torn(__amdgpu_buffer_rsrc_t r, unsigned off, swe_u32x4 g, int aux_sys, int ticks)
{
    // the dangerous order: the half with the NEW tag first, the value it belongs to a few microseconds later
    const swe_u32x2 hi2 = {g.z, g.w}, lo2 = {g.x, g.y};
    if (aux_sys) __builtin_amdgcn_raw_buffer_store_b64(hi2, r, off + 8u, 0, 17); else __builtin_amdgcn_raw_buffer_store_b64(hi2, r, off + 8u, 0, 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(16);
    if (aux_sys) __builtin_amdgcn_raw_buffer_store_b64(lo2, r, off, 0, 17); else __builtin_amdgcn_raw_buffer_store_b64(lo2, r, off, 0, 16);
}
#define SWE_FLOW_TORN(lb_, pc_, sys_) ((swe_flow_tear[0] == (lb_) || swe_flow_tear[0] == -2) && swe_flow_tear[1] > 0 \
                                       && ((pc_) % (swe_flow_tear[2] < 1 ? 1 : swe_flow_tear[2])) == 0 && (!(sys_) || swe_flow_tear[3]))
#endif
__device__ __forceinline__ void swe_flow_put(__amdgpu_buffer_rsrc_t r, unsigned off, double x, unsigned tag, bool torn = false)
{
    const swe_u32x2 xb = __builtin_bit_cast(swe_u32x2, x);
    const swe_u32x4 g = {xb.x, xb.y, tag, swe_flow_check_word(xb.x, xb.y, tag)};
#ifdef SWE_FLOW_TEAR
    if (torn) { swe_flow_put_torn(r, off, g, 0, swe_flow_tear[1]); return; }
#endif
    (void)torn;
    __builtin_amdgcn_raw_buffer_store_b128(g, r, off, 0, 16);
}
__device__ __forceinline__ swe_u32x4 swe_flow_get(__amdgpu_buffer_rsrc_t r, unsigned off)       // value, tag, check
{
    return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16);
}
// the same across devices: system scope (sc0 sc1, aux 17), tag = push number
__device__ __forceinline__ void swe_flow_put_sys(__amdgpu_buffer_rsrc_t r, unsigned off, double x, unsigned tag, bool torn = false)
{
    const swe_u32x2 xb = __builtin_bit_cast(swe_u32x2, x);
    const swe_u32x4 g = {xb.x, xb.y, tag, swe_flow_check_word(xb.x, xb.y, tag)};
#ifdef SWE_FLOW_TEAR
    if (torn) { swe_flow_put_torn(r, off, g, 1, swe_flow_tear[1]); return; }
#endif
    (void)torn;
    __builtin_amdgcn_raw_buffer_store_b128(g, r, off, 0, 17);
}
__device__ __forceinline__ swe_u32x4 swe_flow_get_sys(__amdgpu_buffer_rsrc_t r, unsigned off)
{
    return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 17);
}
__device__ __forceinline__ double swe_flow_val(swe_u32x4 g)
{
    const swe_u32x2 xb = {g.x, g.y};
    return __builtin_bit_cast(double, xb);
}
// has the granule of publish / push `need` (or a later one) arrived, whole?
__device__ __forceinline__ bool swe_flow_arrived(swe_u32x4 g, unsigned need)
{
#ifdef SWE_FLOW_NOCHECK
    return (int)(g.z - need) >= 0;
#else
    return (int)(g.z - need) >= 0 && g.w == swe_flow_check_word(g.x, g.y, g.z);
#endif
}

// right-hand side integrals of one cell: cell integrals + interior facet fluxes (boundary facets contribute zero here, their
// flux is evaluated from the cell's own values and discarded - the branch-free facet loop of swe_stage_kernel).  The six traces
// of facet f are read from LDS where the flux is formed: tr[f] = the LDS addresses (in doubles from `lds`) of {u, v, e at the
// neighbour's node on my node f + 1; u, v, e at its node on my node f} - the block's own stage values for a neighbour inside
// the block (a boundary facet points at this cell itself), the incoming staging entry for a rim facet.
// Lines as in swe_stage_kernel, same order.
// WD (wetting-drying, round 5): e[] is the elevation recovered from the displaced depth D the planes hold, Dn[] that depth = the
// nodal total depth (swe_stage_kernel<..., WD>)
template <bool NONLIN, bool WD = false>
__device__ __forceinline__ void swe_flow_rhs_cell(const SweStageArgs &p, const double u[3], const double v[3], const double e[3],
                                                  const double h[3], const double nx[3], const double ny[3], double bu[3], double bv[3],
                                                  double be[3], const double *Dn = nullptr)
{
#pragma clang fp contract(off)
    const double g = p.g;
    double H[3], gxs[3], gys[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        H[i] = WD ? Dn[i] : (NONLIN ? h[i] + e[i] : h[i]);
        gxs[i] = -0.5*nx[(i + 1) % 3];                     // A*grad(phi_i) = -nF_{i+1}/2
        gys[i] = -0.5*ny[(i + 1) % 3];
    }
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

// NTR = 3: tr[f][c] holds the two addresses of component c; NTR = 1 (the variants with source terms, which have no registers to
// spare: 28 B/lane of scratch with the exchange inside, round 4-5): only component 0's pair is kept, the others follow from it -
// the components of a trace are 1 apart in the staging area (a rim facet: address >= SWE_FLOW_XG) and 3*64 apart in the block's own
// planes - at the price of eight integer instructions per facet and stage.
// (XG, PLANE: where the staging area starts and how long a plane of the block's own values is - the fused stage pair of swe2d_fuse.h
//  has 256-lane planes)
// (NLDS: the size of the array `lds` points into - what the -DSWE_RANGE_CHECK build tests every index against)
template <bool NONLIN, bool LF, bool SRC, int NTR, bool WD = false, int XG = SWE_FLOW_XG, int PLANE = SWE_BLOCK, int NLDS = SWE_FLOW_LDS_DOUBLES>
__device__ __forceinline__ void swe_flow_rhs_facets(const SweStageArgs &p, int k, const double u[3], const double v[3], const double e[3],
                                                    const double h[3], const double *lds, const unsigned tr[3][NTR], int bmarkers,
                                                    const double nx[3], const double ny[3], double twoA, double bu[3], double bv[3],
                                                    double be[3], const double *Dn = nullptr, const double *al = nullptr)
{
#pragma clang fp contract(off)
    const double g = p.g;
#pragma unroll
    for (int f = 0; f < 3; f++) {
        const int a = f, b = (f + 1) % 3;
        const bool bnd = ((bmarkers >> (8*f)) & 0xff) != 0;
        // tr[f][c] = address of component c (u, v, e) at the neighbour's node on my node f + 1 | the same on my node f << 16
        const unsigned ab0 = tr[f][0] & 0xffffu, aa0 = tr[f][0] >> 16;
        const unsigned step = ab0 >= (unsigned)XG ? 1u : (unsigned)(3*PLANE);
        const unsigned ab1 = NTR == 3 ? tr[f][NTR - 2] & 0xffffu : ab0 + step, aa1 = NTR == 3 ? tr[f][NTR - 2] >> 16 : aa0 + step;
        const unsigned ab2 = NTR == 3 ? tr[f][NTR - 1] & 0xffffu : ab0 + 2u*step, aa2 = NTR == 3 ? tr[f][NTR - 1] >> 16 : aa0 + 2u*step;
        const double unb = lds[SWE_LDSI(ab0, NLDS)], una = lds[SWE_LDSI(aa0, NLDS)];
        const double vnb = lds[SWE_LDSI(ab1, NLDS)], vna = lds[SWE_LDSI(aa1, NLDS)];
        const double enb = lds[SWE_LDSI(ab2, NLDS)], ena = lds[SWE_LDSI(aa2, NLDS)];
        const double nxs = nx[f], nys = ny[f];
        double Lf, rLf;
        swe_sqrt_rsqrt(swe_dot2(nxs, nxs, nys, nys), Lf, rLf);
        double Fau, Fbu, Fav, Fbv, Fae, Fbe;
        if constexpr (WD) {
            // the neighbour's traces carry its nodal depth D; its elevation by the closed form (bathymetry and alpha are continuous)
            const double ena_ = swe_wd_eta(ena, h[a], al[a]), enb_ = swe_wd_eta(enb, h[b], al[b]);
            swe_facet_flux<NONLIN, LF, true>(g, p.sigma_lf, u[a], u[b], v[a], v[b], e[a], e[b], h[a], h[b], Dn[a], Dn[b], una, unb, vna, vnb,
                                             ena_, enb_, ena, enb, nxs, nys, Lf, rLf, Fau, Fbu, Fav, Fbv, Fae, Fbe);
        } else
        swe_facet_flux<NONLIN, LF, false>(g, p.sigma_lf, u[a], u[b], v[a], v[b], e[a], e[b], h[a], h[b], 0.0, 0.0, una, unb, vna, vnb, ena, enb,
                                          0.0, 0.0, nxs, nys, Lf, rLf, Fau, Fbu, Fav, Fbv, Fae, Fbe);
        if (bnd) { Fau = 0.0; Fbu = 0.0; Fav = 0.0; Fbv = 0.0; Fae = 0.0; Fbe = 0.0; }
        bu[a] = fma(-0.5, Fau, bu[a]); bu[b] = fma(-0.5, Fbu, bu[b]);
        bv[a] = fma(-0.5, Fav, bv[a]); bv[b] = fma(-0.5, Fbv, bv[b]);
        be[a] = fma(-0.5, Fae, be[a]); be[b] = fma(-0.5, Fbe, be[b]);
    }
    if (SRC) {
        double H[3], gxs[3], gys[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            H[i] = WD ? Dn[i] : (NONLIN ? h[i] + e[i] : h[i]);
            gxs[i] = -0.5*nx[(i + 1) % 3];
            gys[i] = -0.5*ny[(i + 1) % 3];
        }
        swe_source_terms(p, k, p.stride, twoA, u, v, H, gxs, gys, bu, bv, be);
    }
}

// mass inverse, Shu-Osher combine and the boundary facets of the cell (the BINL pass of swe_stage_kernel)
// (WALLFAST: the closed-wall path of swe_boundary_facet; not in the variants with source terms - no registers to spare, 68 B of scratch)
template <bool NONLIN, bool LF, bool WALLFAST, bool WD = false>
__device__ __forceinline__ void swe_flow_finish(const SweStageArgs &p, int k, double beta, const double u[3], const double v[3],
                                                const double e[3], const double h[3], const double nx[3], const double ny[3],
                                                double twoA, int bmarkers, int bkind1, const double bu[3], const double bv[3], const double be[3],
                                                const double wu[3], const double wv[3], const double we[3], double ou[3],
                                                double ov[3], double oe[3], const double *Dn = nullptr, const double *al = nullptr)
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
        const double D_0 = WD ? Dn[0] : 0.0, D_1 = WD ? Dn[1] : 0.0, D_2 = WD ? Dn[2] : 0.0;
        const double al_0 = WD ? al[0] : 0.0, al_1 = WD ? al[1] : 0.0, al_2 = WD ? al[2] : 0.0;
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
            const double Ha_ = WD ? SWE_SEL3(D, a) : (!NONLIN ? SWE_SEL3(h, a) : SWE_SEL3(h, a) + SWE_SEL3(e, a));
            const double Hb_ = WD ? SWE_SEL3(D, b) : (!NONLIN ? SWE_SEL3(h, b) : SWE_SEL3(h, b) + SWE_SEL3(e, b));
            swe_boundary_facet<NONLIN, LF, WD, WALLFAST>(p, (bmarkers >> (8*f)) & 0xff, k, a, b, SWE_SEL3(u, a), SWE_SEL3(u, b), SWE_SEL3(v, a),
                                                  SWE_SEL3(v, b), SWE_SEL3(e, a), SWE_SEL3(e, b), SWE_SEL3(h, a), SWE_SEL3(h, b),
                                                  Ha_, Hb_, WD ? SWE_SEL3(al, a) : 0.0, WD ? SWE_SEL3(al, b) : 0.0, nxs, nys, Lf, rLf, Fau, Fbu, Fav,
                                                  Fbv, Fae, Fbe, kind_next);
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
    // zeta = D - h -> the limited depth D the device carries; dry-ground relaxation of the velocity (swe_stage_kernel<..., WD>)
    if constexpr (WD) swe_wd_finish<3>(p.g, beta*p.dt, h, al, ou, ov, oe, !p.wd_skip_relax);
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

// LDS of a block, in doubles: [0, 9*64) the block's stage values xs[plane][lane] (u0 u1 u2 v0 v1 v2 e0 e1 e2),
// then the rim staging area xg[slot][6] (incoming traces at the top of a stage, outgoing ones at its end)

// FX = false: n_stages stages on the ranges cell_end[0 .. n_stages); the rim traces of the first stage come from the state planes.
// FX = true:  n_cycles exchange cycles (see SweFlowArgs); every cycle starts by publishing its input across the rims (the ghost
//             cells' input arrives in the landing zone, not in the planes), so a stage always finds its rim traces in granules.
// Tags count publishes: publish number pc of the launch carries tag base + pc + 1.  Stage results go to slot set pc & 1, a cycle's
// input to a third set: a block whose cells sit in the outer ghost layers skips the late stages of a cycle and is back at the next
// cycle's start long before its neighbours have read its last stage results - it may overwrite its previous INPUT (read by stage 0
// of the previous cycle, which every neighbour that needs it has finished before this rank's push, hence before the flags that
// lets this block start the cycle: it also waits for this rank's own previous push to be complete) but nothing else.
// POLL: granule loads per lane in flight in a polling trip = POLL*64/6 rim facets per trip (6 value granules per facet, 64 lanes:
// 32 / 64 / 96 facets with 3 / 6 / 9 loads; rounds 3-5 read the padding granules too: 8 / 9 loads for 64 / 72 facets).  The block
// with the most rim facets sets the pace of the whole launch, and a block that needs a second trip per pass (+1.3 us per stage)
// slowed a 131 k-cell mesh from 18.7 to 22.1 us per step; a ninth load in EVERY block costs 0.3 us per step (its issue slot and its
// place in the return queue, even when all lanes point nowhere).  The host picks POLL = 3 when no block of the flow order has more
// than 32 rim facets (the 8 x 4-quad blocks of ordering.flow_block_order), 4 / 6 up to 42 / 64, 9 otherwise (launch_flow); same results
// whichever instance runs.  Round 5: the slowest rank of eight 19.1 us per step with nine loads on two-row blocks, 17.1 with four
// on tiles (padding granules still read), ... with three (profiles/r05s_flow_block_order.txt, r05t_flow_poll6.txt).
// WD (round 5): wetting-drying.  The "elevation" values of the block - registers, LDS planes, rim granules, U(0), the exchange records -
// are the displaced depth D the state planes hold; the elevation is recovered per stage (own nodes before the wave starts to wait,
// the neighbours' where the fluxes are formed), the stage ends with swe_wd_finish: swe_stage_kernel<true, LF, ., SRC, true>'s
// operations.  alpha of the cell's vertices stays in registers with the rest of the geometry.
template <bool NONLIN, bool LF, bool SRC, bool FX, int POLL = 6, bool WD = false>
__global__ __launch_bounds__(SWE_BLOCK) SWE_FLOW_OCCUPANCY void swe_flow_kernel(const SweFlowArgs q)
{
#pragma clang fp contract(off)
    __shared__ double lds[SWE_FLOW_LDS_DOUBLES];
    __shared__ int xsrc[SWE_FLOW_MAX_RIM];                     // the block's incoming list (SweFlowArgs::xsrc)
    __shared__ int lact[SWE_BLOCK];                            // is the lane's cell inside the running stage's range?
    __shared__ unsigned lrec[SWE_BLOCK];                       // FX: places of the block's ghost / send cells' records in a landing zone
    __shared__ int lpeer[SWE_BLOCK];
    __shared__ unsigned char lpub[SWE_FLOW_MAX_RIM];           // does the cell that owns the block's i-th slot publish in the running stage?
    __shared__ double lu0[9][SWE_BLOCK];                       // U(0) of the running time step (18 registers the stage loop cannot spare)
    const SweStageArgs &p = q.st;
    const int lb = swe_logical_block(blockIdx.x, gridDim.x);
    if (lb >= q.n_blocks) return;                              // padding of the grid to a multiple of 8
    const int lane = (int)threadIdx.x;
    const int kcode = q.fcell[lb*SWE_BLOCK + lane];
    const bool real = kcode >= 0;                              // padding lanes mimic a cell: finite values, never stored or published
    const int k = real ? kcode : -1 - kcode;
    const size_t S = p.stride;
    const unsigned S8 = (unsigned)S*8u;
    unsigned *const myflag = q.flag + (size_t)lb*SWE_FLOW_FLAG_STRIDE;
    const unsigned base = *myflag;                             // publishes counted so far: written by this block's wave in the previous launch
    const unsigned fin = base + (unsigned)q.n_stages;
    if (!FX && !__any(real && k < q.cell_end[0])) {            // this block takes part in no stage of the launch
        if (lane == 0) *myflag = fin;
        return;
    }
    const unsigned k8 = (unsigned)k*8u;
    // bounds-checked resource: a load from SWE_FLOW_NOWHERE costs no memory access
    const __amdgpu_buffer_rsrc_t rex = __builtin_amdgcn_make_buffer_rsrc(q.ex, 0, 3*q.parity_bytes, 0x00020000);
    const int2 myslots = q.xblk[lb];                           // uniform: first slot, count
    const int nrim = myslots.y;
    for (int i = lane; i < nrim; i += SWE_BLOCK) xsrc[SWE_LDSI(i, SWE_FLOW_MAX_RIM)] = q.xsrc[myslots.x + i];

    // ---- launch invariants of the cell: connectivity, exchange slots, geometry
    int bmarkers, bkind1 = 0;
    constexpr int NTR = WD ? 1 : SWE_FLOW_NTR_SRC;          // (wetting-drying: no registers to spare, see swe_flow_rhs_facets)
    unsigned tr[3][NTR];                   // LDS addresses of the six traces of every facet (see swe_flow_rhs_facets)
    int xown[3];                           // rim facets: my slot, counted from the block's first slot; else -1
    double h[3], nx[3], ny[3], al[3] = {0.0, 0.0, 0.0};
    double u[3], v[3], e[3];
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
        // the launch's input: written by earlier kernels, plain loads
        const swe_rsrc_t gu = swe_rsrc(q.buf[0]), gv = swe_rsrc(q.buf[0] + 3*S), ge = swe_rsrc(q.buf[0] + 6*S);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            u[i] = swe_ld(gu, k8, i*S8);
            v[i] = swe_ld(gv, k8, i*S8);
            e[i] = swe_ld(ge, k8, i*S8);
        }
        double r0[3][6];                   // first stage (FX = false): the rim traces come from the state planes too
#pragma unroll
        for (int f = 0; f < 3; f++) {
            const int nbf = nb[f];
            const bool rim = xown[f] >= 0;                                 // interior facet whose neighbour lives in another block
            const bool inw = nbf >= 0 && !rim;
            // the neighbour traverses the shared facet backwards: its node f2 sits on my node f + 1, its node (f2 + 1) % 3 on my node f
            const int ls = rim ? lane : xin[f];
            const int f2 = inw ? (nbf & 3) : f, f2a = f2 == 2 ? 0 : f2 + 1;
#pragma unroll
            for (int c = 0; c < NTR; c++) {
                const unsigned ab = rim ? (unsigned)(SWE_FLOW_XG + 6*xin[f] + c) : (unsigned)((3*c + f2)*SWE_BLOCK + ls);
                const unsigned aa = rim ? (unsigned)(SWE_FLOW_XG + 6*xin[f] + 3 + c) : (unsigned)((3*c + f2a)*SWE_BLOCK + ls);
                tr[f][c] = ab | (aa << 16);
            }
            if (!FX) {
                const int code = rim ? nbf : ((k << 2) | f);
                const unsigned kn8 = (unsigned)(code >> 2)*8u;
                const int g2 = code & 3;
                const unsigned ob = kn8 + (g2 == 0 ? 0u : (g2 == 1 ? S8 : 2u*S8));         // node g2
                const unsigned oa = kn8 + (g2 == 0 ? S8 : (g2 == 1 ? 2u*S8 : 0u));         // node (g2 + 1) % 3
                r0[f][0] = swe_ld(gu, ob, 0); r0[f][1] = swe_ld(gv, ob, 0); r0[f][2] = swe_ld(ge, ob, 0);
                r0[f][3] = swe_ld(gu, oa, 0); r0[f][4] = swe_ld(gv, oa, 0); r0[f][5] = swe_ld(ge, oa, 0);
            }
        }
        double px[3], py[3];
        const swe_rsrc_t rvx = swe_rsrc(p.vx), rvy = swe_rsrc(p.vy), rvh = swe_rsrc(p.vh);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const unsigned v8 = (unsigned)vid[i]*8u;
            px[i] = swe_ld(rvx, v8, 0);
            py[i] = swe_ld(rvy, v8, 0);
            h[i] = swe_ld(rvh, v8, 0);
            if (WD) al[i] = swe_ld(swe_rsrc(p.valpha), v8, 0);
        }
#pragma unroll
        for (int f = 0; f < 3; f++) {
            const int b = (f + 1) % 3;
            nx[f] = py[b] - py[f];
            ny[f] = px[f] - px[b];
        }
        if (!FX) {
#pragma unroll
            for (int f = 0; f < 3; f++) {
                if (xown[f] >= 0) {
#pragma unroll
                    for (int j = 0; j < 6; j++) lds[SWE_LDSI(SWE_FLOW_XG + 6*xin[f] + j, SWE_FLOW_LDS_DOUBLES)] = r0[f][j];
                }
            }
        }
    }
    // ---- FX: this cell's places in the halo lists, the epochs the launch starts from
    int xs1 = -1, xs2 = -1, xr = -1;
    unsigned long long S0 = 0ull, R0 = 0ull;
    bool pend0 = false, has_ghost = false, has_send = false;
    if (FX) {
        const int2 t = q.xsend[lb*SWE_BLOCK + lane];
        xs1 = real ? t.x : -1; xs2 = real ? t.y : -1;
        xr = real ? q.xrecv[lb*SWE_BLOCK + lane] : -1;
        // both counters are only written by the launch's LAST block to finish (below): stable for every wave of the launch
        S0 = q.xctr->epoch_send; R0 = q.xctr->epoch_recv;
        pend0 = S0 > R0;                   // a push of the previous launch has not been received yet
        has_ghost = __any(xr >= 0);
        has_send = __any(xs1 >= 0);
    }
    const int ncyc = FX ? q.n_cycles : 1, spc = FX ? q.stages_per_cycle : q.n_stages;
    unsigned long long t_start = 0ull;
    bool late = false;
    int s = 0;                             // stage counter of the launch

    // publish the rim traces held in (pu, pv, pe) of the lanes in `who`: facet f carries my nodes f (granules 0-2) and f + 1
    // (granules 3-5).  The values go to the staging area by slot, then the wave stores the block's whole slot range, consecutive
    // lanes on consecutive granules (full lines).  The slots of rim cells outside `who` (cells that have dropped out of the shrinking
    // stage range) are NOT stored: value and tag stay those of the cell's last active stage.  The two-parity argument above - a
    // producer reaches the end of stage s + 2 only after the consumer has read stage s - holds for cells the producer still waits
    // for; a cell active in stage s but not later makes its block wait for nothing from the block it faces, so the block may run
    // two stages ahead of a consumer that is stalled on another neighbour and must not touch the slot that consumer has yet to read.
#ifdef SWE_FLOW_TEAR
#define SWE_FLOW_TORN_HERE(pc_, sys_) SWE_FLOW_TORN(lb, pc_, sys_)
#else
#define SWE_FLOW_TORN_HERE(pc_, sys_) false
#endif
// (Measured and not kept, round 5: the LDS reads of a trip of the store loop issued together - four granules per lane and trip,
//  clamped indices - instead of one read, wait and store per granule: 19.3 against 19.3 us per step for rank 3 of eight, 17.2
//  against 17.1 on one device at 125 k cells, and 92 instead of 28 B/lane of scratch in the source-term + exchange variant;
//  profiles/r05d_flow_loops_ab.txt.)
#define SWE_FLOW_PUBLISH_STORES(pc_) \
        for (int t_ = lane; t_ < 8*nrim; t_ += SWE_BLOCK) {                                                                       \
            const int gi_ = t_ & 7;                                                                                               \
            const double x_ = gi_ < 6 ? lds[SWE_LDSI(SWE_FLOW_XG + 6*(t_ >> 3) + gi_, SWE_FLOW_LDS_DOUBLES)] : 0.0;               \
            if (lpub[SWE_LDSI(t_ >> 3, SWE_FLOW_MAX_RIM)]) swe_flow_put(rex, par_ + 16u*(unsigned)t_, x_, tag_, SWE_FLOW_TORN_HERE(pc_, 0));   \
        }
#define SWE_FLOW_PUBLISH(pu, pv, pe, who, pc_, set_) do {                                                                               \
        const unsigned tag_ = base + (unsigned)(pc_) + 1u;                                                                        \
        const unsigned par_ = (unsigned)(set_)*q.parity_bytes + (unsigned)myslots.x*SWE_FLOW_SLOT_BYTES;                    \
        __syncthreads();                                       /* every lane has read its incoming traces */                      \
        _Pragma("unroll")                                                                                                         \
        for (int f = 0; f < 3; f++) {                                                                                             \
            if (xown[f] >= 0 && (who)) {                                                                                          \
                const int a_ = f, b_ = (f + 1) % 3;                                                                               \
                double *d_ = lds + SWE_LDSI(SWE_FLOW_XG + 6*xown[f] + 5, SWE_FLOW_LDS_DOUBLES) - 5;                                                                       \
                d_[0] = pu[a_]; d_[1] = pv[a_]; d_[2] = pe[a_]; d_[3] = pu[b_]; d_[4] = pv[b_]; d_[5] = pe[b_];                   \
            }                                                                                                                     \
            if (xown[f] >= 0) lpub[SWE_LDSI(xown[f], SWE_FLOW_MAX_RIM)] = (who) ? 1 : 0;                                          \
        }                                                                                                                         \
        __syncthreads();                                                                                                          \
SWE_FLOW_PUBLISH_STORES(pc_)                                                            \
    } while (0)

#pragma unroll 1
    for (int c = 0; c < ncyc; c++) {
        if (FX) {
            // ---- receive: what the peers pushed at the end of their previous cycle (or launch).  Every ghost lane waits for the nine
            //      granules of ITS cell to carry that push's number - no flag per rank: a ghost cell is ready as soon as the one block
            //      of the peer that owns it has finished, and the peer may overwrite the slot two pushes later only after it has
            //      received this rank's next push, which depends on this cell having been read
            SWE_FLOW_DELAY_AT(4);
            if (((c > 0) || pend0) && has_ghost) {
                const unsigned target = (unsigned)(R0 + (unsigned long long)c + (pend0 ? 1ull : 0ull));
                const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(q.x_zone, 0, q.x_zbytes, 0x00020000);
                const unsigned zo = xr >= 0 ? (target & 1u)*q.x_slot + (unsigned)xr*144u : SWE_FLOW_NOWHERE;
                // A block of outer ghost cells skips the late stages of a cycle and waits here for most of it: ONE lane watches ONE
                // granule, slowly (every polling pass of every lane is nine fabric reads per ghost cell - MI355X_MICROARCH.md,
                // polling-cost), the full passes start when that one has arrived
                {
                    const unsigned long long gm = __ballot(xr >= 0);
                    const int first = (int)__builtin_ctzll(gm);
                    const unsigned hint = __builtin_amdgcn_readlane(zo, first);
                    for (unsigned spins = 0; !late; spins++) {
                        const swe_u32x4 g1 = swe_flow_get_sys(rz, lane == 0 ? hint : SWE_FLOW_NOWHERE);
                        if (__any(lane == 0 && swe_flow_arrived(g1, target))) break;
                        __builtin_amdgcn_s_sleep(8);
                        if ((spins & 15u) == 15u) {
                            const unsigned long long now = wall_clock64();
                            if (t_start == 0ull) t_start = now;
                            else if (now - t_start > q.x_timeout) {
                                late = true;
                                if (lane == 0 && atomicAdd(q.status, 1u) == 0u) q.status[1] = (unsigned)lb + 1u;
                            }
                        }
                    }
                    t_start = 0ull;
                }
                // the records (nine granules = 144 B per ghost cell) are read by nine consecutive lanes each - a 16-byte access per
                // lane and granule would be nine fabric reads per cell and pass - into the staging area, ghost cell after ghost cell
                // of the block; the ghost lanes then pick their nine values from LDS
                const unsigned long long gm = __ballot(xr >= 0);
                const int ng = (int)__builtin_popcountll(gm);
                const int ci = (int)__builtin_popcountll(gm & ((1ull << lane) - 1ull));       // this ghost lane's rank among them
                __syncthreads();
                if (xr >= 0) lrec[SWE_LDSI(ci, SWE_BLOCK)] = zo;
                __syncthreads();
                for (unsigned spins = 0;; spins++) {
                    bool ok = true;
                    // the loads of a pass are issued TOGETHER (three per lane and trip: 21 ghost cells; a block next to a cut holds
                    // ~30): one round trip to the zone - uncached memory the peers write over the fabric - per trip instead of one
                    // per 64 granules (the rolled loop waited for every load before it issued the next, round 5)
                    for (int c0 = 0; c0 < 9*ng; c0 += SWE_FLOW_RX*SWE_BLOCK) {
                        swe_u32x4 gz[SWE_FLOW_RX];
                        unsigned zoff[SWE_FLOW_RX];
#pragma unroll
                        for (int j = 0; j < SWE_FLOW_RX; j++) {
                            const int t = min(c0 + j*SWE_BLOCK + lane, 9*ng - 1);
                            const int cc = (t*7282) >> 16;                                         // t / 9
                            zoff[j] = lrec[SWE_LDSI(cc, SWE_BLOCK)] + 16u*(unsigned)(t - 9*cc);
                        }
#pragma unroll
                        for (int j = 0; j < SWE_FLOW_RX; j++) gz[j] = swe_flow_get_sys(rz, c0 + j*SWE_BLOCK + lane < 9*ng ? zoff[j] : SWE_FLOW_NOWHERE);
#pragma unroll
                        for (int j = 0; j < SWE_FLOW_RX; j++) {
                            const int t = c0 + j*SWE_BLOCK + lane;
                            if (t < 9*ng) {
                                ok = ok && swe_flow_arrived(gz[j], target);
                                lds[SWE_LDSI(SWE_FLOW_XG + t, SWE_FLOW_LDS_DOUBLES)] = swe_flow_val(gz[j]);
                            }
                        }
                    }
                    if (__all(ok) || late) break;
                    __builtin_amdgcn_s_sleep(4);
                    if ((spins & 31u) == 31u) {
                        const unsigned long long now = wall_clock64();
                        if (t_start == 0ull) t_start = now;
                        else if (now - t_start > q.x_timeout) {
                            late = true;
                            if (lane == 0 && atomicAdd(q.status, 1u) == 0u) q.status[1] = (unsigned)lb + 1u;
                        }
                    }
                }
                t_start = 0ull;
                __syncthreads();
                if (xr >= 0) {
#pragma unroll
                    for (int i = 0; i < 3; i++) { u[i] = lds[SWE_LDSI(SWE_FLOW_XG + 9*ci + i, SWE_FLOW_LDS_DOUBLES)]; v[i] = lds[SWE_LDSI(SWE_FLOW_XG + 9*ci + 3 + i, SWE_FLOW_LDS_DOUBLES)]; e[i] = lds[SWE_LDSI(SWE_FLOW_XG + 9*ci + 6 + i, SWE_FLOW_LDS_DOUBLES)]; }
                }
                // ... and into the state planes, for the kernels after this launch: what the LAST cycle receives (an earlier cycle's
                // copy is overwritten by the next one's before anything reads it - profiles/r05v_flow_last_result_store.txt)
                if (xr >= 0 && c == ncyc - 1) {
                    const swe_rsrc_t gou = swe_rsrc(q.buf[0]), gov = swe_rsrc(q.buf[0] + 3*S), goe = swe_rsrc(q.buf[0] + 6*S);
#pragma unroll
                    for (int i = 0; i < 3; i++) { swe_st(gou, k8, i*S8, u[i]); swe_st(gov, k8, i*S8, v[i]); swe_st(goe, k8, i*S8, e[i]); }
                }
            }
            // ---- the cycle's input across the rims
            SWE_FLOW_PUBLISH(u, v, e, real, c*spc, 2);
        }
#pragma unroll 1
        for (int g = 0; g < spc; g++, s++) {
            const int end_s = q.cell_end[g];
            const bool act = real && k < end_s;
            if (!__any(act)) {                                 // the ranges of a cycle only shrink: nothing left for this block
                if (FX) { s += spc - g; break; }               // ... until the next cycle
                c = ncyc; break;
            }
            const int i3 = g % 3;
            SWE_FT(0);
            // Opaque to the optimiser: without this every stage-invariant quantity (facet lengths, reciprocals, gradients ...) is
            // hoisted out of the stage loop and kept live across it - past the register budget.  The per-stage kernel recomputes
            // them in every stage as well.
#pragma unroll
            for (int i = 0; i < 3; i++) asm volatile("" : "+v"(nx[i]), "+v"(ny[i]), "+v"(h[i]));
            if (WD) {
#pragma unroll
                for (int i = 0; i < 3; i++) asm volatile("" : "+v"(al[i]));
            }
            asm volatile("" : "+v"(bmarkers));
#pragma unroll
            for (int f = 0; f < 3; f++)
#pragma unroll
                for (int c = 0; c < NTR; c++) asm volatile("" : "+v"(tr[f][c]));
            if (i3 == 0) {
#pragma unroll
                for (int i = 0; i < 3; i++) { lu0[i][lane] = u[i]; lu0[3 + i][lane] = v[i]; lu0[6 + i][lane] = e[i]; }
            }
            // ---- the block's stage values for its own lanes
#pragma unroll
            for (int i = 0; i < 3; i++) { lds[SWE_LDSI(i*SWE_BLOCK + lane, SWE_FLOW_LDS_DOUBLES)] = u[i]; lds[SWE_LDSI((3 + i)*SWE_BLOCK + lane, SWE_FLOW_LDS_DOUBLES)] = v[i]; lds[SWE_LDSI((6 + i)*SWE_BLOCK + lane, SWE_FLOW_LDS_DOUBLES)] = e[i]; }
            lact[SWE_LDSI(lane, SWE_BLOCK)] = act ? 1 : 0;
            // ---- traces across the rim: the chunks the neighbour blocks wrote for this block, consecutive lanes on consecutive
            //      granules, re-read until every granule a cell of this stage's range needs carries this stage's tag
            // The cell integrals need no neighbour: they are evaluated BEFORE the wave starts to wait for its rim granules (it would
            // sleep in the polling loop otherwise), which takes them - a quarter of a stage's arithmetic - off the chain
            // "neighbour publishes -> this block sees it -> computes -> publishes" that sets the period of a stage (round 5).
            // Same operations in the same order as before: the same bits.
            double bu[3], bv[3], be[3];
            const double twoA = fma(nx[0], ny[1], -(ny[0]*nx[1]));
            // wetting-drying: e[] holds D; the elevation of the own nodes (three reciprocals) is formed here, before the wait
            double eta[3];
#pragma unroll
            for (int i = 0; i < 3; i++) eta[i] = WD ? swe_wd_eta(e[i], h[i], al[i]) : e[i];
#define SWE_FLOW_CELL_TERMS_HERE swe_flow_rhs_cell<NONLIN, WD>(p, u, v, eta, h, nx, ny, bu, bv, be, e)
            // w = a0*U(0) + a1*U_in: the first stage of a step has no U(0) term (swe_stage_kernel<., ., HASU0 = false>); with
            // wetting-drying the continuity equation advances zeta = D - h (the planes, U(0)'s too, hold D)
#define SWE_FLOW_W_HERE do {                                                                                                          \
                const double a0 = q.a0[i3], a1 = q.a1[i3];                                                                        \
                _Pragma("unroll")                                                                                                 \
                for (int i = 0; i < 3; i++) { wu[i] = a1*u[i]; wv[i] = a1*v[i]; we[i] = WD ? a1*(e[i] - h[i]) : a1*e[i]; }        \
                if (i3 > 0) {                                                                                                     \
                    _Pragma("unroll")                                                                                             \
                    for (int i = 0; i < 3; i++) {                                                                                 \
                        wu[i] = fma(a0, lu0[i][lane], wu[i]);                                                                     \
                        wv[i] = fma(a0, lu0[3 + i][lane], wv[i]);                                                                 \
                        we[i] = fma(a0, WD ? lu0[6 + i][lane] - h[i] : lu0[6 + i][lane], we[i]);                                  \
                    }                                                                                                             \
                }                                                                                                                 \
            } while (0)
            // What is evaluated in front of the wait has to be PINNED there: the compiler otherwise sinks the whole evaluation past the
            // polling loop to its first use (the ISA of the build that introduced "cell integrals before the wait" had every FP64
            // instruction of a stage behind the loop).  An empty asm that "modifies" the results keeps them where they are written.
            // Rank 3 of eight, us per step (profiles/r05zf_flow_pinned_before_wait.txt): nothing pinned 16.46, cell integrals 16.25,
            // + the weights 15.90; with source terms 21.22 / 20.47 / 20.66; wetting-drying 20.15 / 20.24 / 20.68 (registers) - hence:
            constexpr bool PIN_CELL = !WD, PIN_W = !WD && !SRC;
            double wu[3], wv[3], we[3];
            if constexpr (PIN_W) {
                SWE_FLOW_W_HERE;
#pragma unroll
                for (int i = 0; i < 3; i++) asm volatile("" : "+v"(wu[i]), "+v"(wv[i]), "+v"(we[i]));
            }
            if (FX || g > 0) {
                SWE_FLOW_DELAY_AT(1);
                const int pc_in = FX ? c*spc + g : g - 1;      // the publish this stage reads
                const unsigned need = base + (unsigned)pc_in + 1u;
                const unsigned par = ((FX && g == 0) ? 2u : ((unsigned)pc_in & 1u))*q.parity_bytes;       // a cycle's input has a slot set of its own
                __syncthreads();                               // the incoming list / the previous stage's staging reads
                // Where a trip's granules live: the incoming list and the cells' activity come from LDS by UNCONDITIONAL reads of
                // clamped indices, all of a trip issued together (two LDS round trips per trip; inside `if`s the compiler waited for
                // each read - sixteen dependent LDS latencies per polling pass, ~0.7 us of the 1.5 us between a neighbour's publish
                // and this block seeing it, profiles/r05b).
                unsigned poff[POLL];
#define SWE_FLOW_POLL_OFFSETS(c0_) do {                                                                                               \
                    int ent_[POLL], act_[POLL];                                                                                   \
                    _Pragma("unroll")                                                                                             \
                    for (int j = 0; j < POLL; j++) ent_[j] = xsrc[SWE_LDSI(min(SWE_FLOW_DIV6((c0_) + j*SWE_BLOCK + lane), SWE_FLOW_MAX_RIM - 1), SWE_FLOW_MAX_RIM)]; \
                    _Pragma("unroll")                                                                                             \
                    for (int j = 0; j < POLL; j++) act_[j] = lact[SWE_LDSI(ent_[j] & (SWE_BLOCK - 1), SWE_BLOCK)];                \
                    _Pragma("unroll")                                                                                             \
                    for (int j = 0; j < POLL; j++) {                                                                              \
                        const int t = (c0_) + j*SWE_BLOCK + lane;                                                                 \
                        /* a cell outside this stage's range needs nothing (and its neighbour may never have published) */        \
                        poff[j] = (t < 6*nrim && act_[j]) ? (unsigned)(ent_[j] >> 6)*SWE_FLOW_SLOT_BYTES + 16u*(unsigned)(t - 6*SWE_FLOW_DIV6(t)) + par : SWE_FLOW_NOWHERE; \
                    }                                                                                                             \
                } while (0)
                // Computed ONCE per stage, outside the spin loop, where a pass is one trip (no block of the order has more rim facets than
                // a trip covers: the product case): a pass is then the granule loads and nothing before them.  (Rounds 3-5 recomputed them
                // in every pass - eight registers across the loop made the kernel spill 24-48 B/lane; the unit is now compiled without
                // machine LICM, _build.py UNIT_FLAGS, which left 14-20 registers free.  profiles/r05n_flow_no_licm_hoisted_poll.txt.)
                SWE_FLOW_CELL_TERMS_HERE;
                if constexpr (PIN_CELL) {
#pragma unroll
                    for (int i = 0; i < 3; i++) asm volatile("" : "+v"(bu[i]), "+v"(bv[i]), "+v"(be[i]));
                }
                const bool one_trip = !WD && 6*nrim <= POLL*SWE_BLOCK;      // (not with wetting-drying: eight registers too many)
                if constexpr (!WD) SWE_FLOW_POLL_OFFSETS(0);
                for (unsigned spins = 0;; spins++) {
                    bool ok = true;
                    for (int c0 = 0; c0 < 6*nrim; c0 += POLL*SWE_BLOCK) {      // POLL loads per lane in flight (32 rim facets per three loads)
                        swe_u32x4 gr[POLL];
                        if (!one_trip) SWE_FLOW_POLL_OFFSETS(c0);
#pragma unroll
                        for (int j = 0; j < POLL; j++) gr[j] = swe_flow_get(rex, poff[j]);
#pragma unroll
                        for (int j = 0; j < POLL; j++) {
                            const int t = c0 + j*SWE_BLOCK + lane;
                            if (poff[j] != SWE_FLOW_NOWHERE) {
                                ok = ok && swe_flow_arrived(gr[j], need);
                                lds[SWE_LDSI(SWE_FLOW_XG + t, SWE_FLOW_LDS_DOUBLES)] = swe_flow_val(gr[j]);      // staging: [slot][6]
                            }
                        }
                    }
                    if (__all(ok) || late) break;
                    __builtin_amdgcn_s_sleep(SWE_FLOW_POLL_SLEEP);
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
#undef SWE_FLOW_POLL_OFFSETS
            } else {
                SWE_FLOW_CELL_TERMS_HERE;
            }
            __syncthreads();
            SWE_FT(1);
            SWE_FT(2);
            // From here to the publish the wave is on the chain that sets the period of a stage; the block it shares its SIMD with is
            // most likely polling (cheap instructions in a loop, which take issue slots all the same): priority to the one that
            // computes (profiles/r05d_flow_loops_ab.txt)
            __builtin_amdgcn_s_setprio(3);
#undef SWE_FLOW_CELL_TERMS_HERE
            double ou[3], ov[3], oe[3];
            swe_flow_rhs_facets<NONLIN, LF, SRC, NTR, WD>(p, k, u, v, eta, h, lds, tr, bmarkers, nx, ny, twoA, bu, bv, be, e, al);
            if constexpr (!PIN_W) SWE_FLOW_W_HERE;
#undef SWE_FLOW_W_HERE
            // (a lane outside the stage's range has no boundary facets to do: the outermost ghost layer of a partition, which is in no
            //  stage's range, points its missing neighbours at a wall - unmasked, every block that holds such a cell ran the boundary
            //  pass in every stage, +0.8 us for the 300 blocks next to the cuts of a rank of eight)
            swe_flow_finish<NONLIN, LF, SWE_FLOW_WALLFAST_SRC || !SRC, WD>(p, k, q.beta[i3], u, v, eta, h, nx, ny, twoA, act ? bmarkers : 0, bkind1, bu, bv, be, wu, wv, we, ou, ov, oe, e, al);
#ifdef SWE_WAVE_TIMING
            if (ou[0] == 1.2345e300) return;          // the arithmetic has to be finished before the time stamp
            SWE_FT(3);
#endif
            // ---- publish the rim traces of this stage's result (FX: the last stage of a cycle leaves that to the next cycle's input
            //      publish, after the exchange; FX = false: nobody reads the last stage of the launch)
            SWE_FLOW_DELAY_AT(2);
            if (FX ? g + 1 < spc : s + 1 < q.n_stages) SWE_FLOW_PUBLISH(ou, ov, oe, act, FX ? c*spc + g + 1 : s, (FX ? c*spc + g + 1 : s) & 1);
            __builtin_amdgcn_s_setprio(0);
            // ---- the step result (every third stage) goes to state buffer 0: read by later launches only - so only the LAST result
            //      the launch computes for the cell is stored (the cell is in no later step's final range: the ranges never grow; with
            //      the exchange inside, in the last cycle).  Rounds 3-5 stored every step's result: nine store instructions per lane and
            //      step that nothing ever read, between a publish and the next polling pass.
            const bool last_result = (!FX || c == ncyc - 1) && (g + 3 >= spc || k >= q.cell_end[g + 3]);
            if (act && i3 == 2 && last_result) {
                const swe_rsrc_t gou = swe_rsrc(q.buf[0]), gov = swe_rsrc(q.buf[0] + 3*S), goe = swe_rsrc(q.buf[0] + 6*S);
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    swe_st(gou, k8, i*S8, ou[i]);
                    swe_st(gov, k8, i*S8, ov[i]);
                    swe_st(goe, k8, i*S8, oe[i]);
                }
            }
            SWE_FT(4);
#ifdef SWE_WAVE_TIMING
            if (s == SWE_FLOW_TS_STAGE - 1 && lane == 0 && lb < SWE_WT_MAX) {         // when the previous stage's granules left, and from which XCD
                // ... and where the block runs: bits 40-53 = simd_id (2) | cu_id (4) | sh_id (1) | se_id (3) | wave_id (4) of HW_ID, bits
                // 56-59 the XCC - which blocks share a SIMD, and do they compute at the same time?  (tools/flowtiming.py --mates)
                unsigned xcc, hw;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                const unsigned key = ((hw >> 4) & 0x3u) | (((hw >> 8) & 0xfu) << 2) | (((hw >> 12) & 0x1u) << 6) | (((hw >> 13) & 0x7u) << 7)
                                     | ((hw & 0xfu) << 10);
                swe_wave_ts[5][lb] = (wall_clock64() & 0xffffffffffull) | ((unsigned long long)key << 40) | ((unsigned long long)(xcc & 0xf) << 56);
            }
#endif
#pragma unroll
            for (int i = 0; i < 3; i++) { u[i] = ou[i]; v[i] = ov[i]; e[i] = oe[i]; }
        }
        if (FX) SWE_FLOW_DELAY_AT(8);
        if (FX && has_send) {
            // ---- push: the send cells of this block straight into the peers' landing zones as granules tagged with the push number
            //      (the cycle's last stage left the step result in u, v, e).  No drain, no flag.
            const unsigned target = (unsigned)(S0 + (unsigned long long)c + 1ull);
            // a cell's record (nine granules = 144 B) is written by nine consecutive lanes: the send lanes drop their values and
            // the record's place in LDS (send cell after send cell of the block), then the wave stores record after record
#pragma unroll 1
            for (int w = 0; w < 2; w++) {
                const int j = w ? xs2 : xs1;
                const unsigned long long sm = __ballot(j >= 0);
                if (sm == 0ull) break;                         // uniform
                const int ns = (int)__builtin_popcountll(sm);
                const int ci = (int)__builtin_popcountll(sm & ((1ull << lane) - 1ull));
                __syncthreads();
                if (j >= 0) {
                    int pp = 0;
#pragma unroll 1
                    for (int i = 1; i < q.x_n_peers; i++) if (j >= q.x_off[i]) pp = i;          // segments are sorted by offset
                    lrec[SWE_LDSI(ci, SWE_BLOCK)] = (target & 1u)*q.x_rslot[pp] + (unsigned)(j - q.x_off[pp])*144u;
                    lpeer[SWE_LDSI(ci, SWE_BLOCK)] = pp;
#pragma unroll
                    for (int i = 0; i < 3; i++) { lds[SWE_LDSI(SWE_FLOW_XG + 9*ci + i, SWE_FLOW_LDS_DOUBLES)] = u[i]; lds[SWE_LDSI(SWE_FLOW_XG + 9*ci + 3 + i, SWE_FLOW_LDS_DOUBLES)] = v[i]; lds[SWE_LDSI(SWE_FLOW_XG + 9*ci + 6 + i, SWE_FLOW_LDS_DOUBLES)] = e[i]; }
                }
                __syncthreads();
                for (int pp = 0; pp < q.x_n_peers; pp++) {     // uniform: one buffer resource per peer
                    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(q.x_rdata[pp], 0, q.x_rbytes[pp], 0x00020000);
                    for (int c0 = 0; c0 < 9*ns; c0 += 3*SWE_BLOCK) {          // LDS reads of a trip together, then its stores
                        double xv[3];
                        unsigned zoff[3];
                        int pr[3];
#pragma unroll
                        for (int j = 0; j < 3; j++) {
                            const int t = min(c0 + j*SWE_BLOCK + lane, 9*ns - 1);
                            const int cc = (t*7282) >> 16;
                            pr[j] = lpeer[SWE_LDSI(cc, SWE_BLOCK)];
                            zoff[j] = lrec[SWE_LDSI(cc, SWE_BLOCK)] + 16u*(unsigned)(t - 9*cc);
                            xv[j] = lds[SWE_LDSI(SWE_FLOW_XG + t, SWE_FLOW_LDS_DOUBLES)];
                        }
#pragma unroll
                        for (int j = 0; j < 3; j++)
                            if (c0 + j*SWE_BLOCK + lane < 9*ns && pr[j] == pp) swe_flow_put_sys(rp, zoff[j], xv[j], target, SWE_FLOW_TORN_HERE(c, 1));
                    }
                }
            }
        }
    }
#undef SWE_FLOW_PUBLISH
#undef SWE_FLOW_TORN_HERE
    // retired or finished: every block's counter ends the launch at base + n_stages
    if (lane == 0) *myflag = fin;
    if (FX && lane == 0) {
        // the last block to finish advances the epochs (nobody reads them any more in this launch)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t_ = __hip_atomic_fetch_add(q.xtick, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t_ == (unsigned)q.n_blocks - 1u) {
            __hip_atomic_store(q.xtick, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            q.xctr->epoch_send = S0 + (unsigned long long)ncyc;
            q.xctr->epoch_recv = R0 + (unsigned long long)(ncyc - 1) + (pend0 ? 1ull : 0ull);
        }
    }
}

// Receives a pending push (pushes > receives) outside a flow launch: the ghost cells' granules into the state planes.  For the
// end of an advance - the next FX launch would do it itself, but the state may leave the device or other kernels may run first.
static __global__ __launch_bounds__(256) void swe_flow_unpack_kernel(double *planes, size_t stride, const int *recv_cells, int n_recv,
                                                              void *zone, unsigned zbytes, unsigned slot, SweP2pCounters *ctr,
                                                              unsigned *status, unsigned long long timeout_ticks)
{
    const unsigned long long S0 = ctr->epoch_send, R0 = ctr->epoch_recv;
    if (S0 <= R0) return;                                      // nothing pending (uniform over the grid)
    const unsigned target = (unsigned)(R0 + 1ull);
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(zone, 0, zbytes, 0x00020000);
    const unsigned long long w0 = wall_clock64();
    for (int j = blockIdx.x*256 + threadIdx.x; j < n_recv; j += gridDim.x*256) {
        const unsigned zo = (target & 1u)*slot + (unsigned)j*144u;
        double x[9];
        bool timed_out = false;
        for (;;) {
            bool ok = true;
            for (int i = 0; i < 9; i++) {
                const swe_u32x4 g = swe_flow_get_sys(rz, zo + 16u*i);
                x[i] = swe_flow_val(g);
                ok = ok && swe_flow_arrived(g, target);
            }
            if (ok) break;
            if (wall_clock64() - w0 > timeout_ticks) {
                // counted and located like a timeout of the flow kernel ("block" = this kernel's workgroup); the cell keeps its old
                // values - the state is reported invalid either way, never silently patched with half-arrived granules
                if (atomicAdd(status, 1u) == 0u) status[1] = (unsigned)blockIdx.x + 1u;
                timed_out = true;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        if (timed_out) continue;
        const int k = recv_cells[j];
        for (int i = 0; i < 9; i++) planes[(size_t)i*stride + k] = x[i];
    }
    // the epoch: advanced by the last block (every block read R0 before it)
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned *tick = status + 2;
        const unsigned t_ = __hip_atomic_fetch_add(tick, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t_ == gridDim.x - 1u) {
            __hip_atomic_store(tick, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ctr->epoch_recv = R0 + 1ull;
        }
    }
}
