// swe2d_api_flow.hip - host side of the dataflow stage loop (swe2d_flow.h): slot tables, launches, the ABI entry points
#include "swe2d_handle.h"
#include "swe2d_pick.h"

namespace swe2d_impl {

// the last flow launch per device of this process (launch_flow)
struct FlowChain { hipEvent_t ev = nullptr; unsigned long long last_uid = 0ull; };
constexpr int kFlowChainDevices = 64;
FlowChain g_flow_chain[kFlowChainDevices];
std::mutex g_flow_chain_mu;

// ---- dataflow stage loop (swe2d_flow.h): host tables
// The kernel's 64-cell blocks are consecutive positions of a FLOW ORDER of the cells (default: the device numbering; a
// partition passes an order in which its ghost layers - appended layer by layer to the numbering the stage ranges need - sit
// next to the owned cells they touch, swe2d_flow_set_order).  Rim facets = interior facets whose two cells sit in different
// blocks.  A block's exchange slots are contiguous and grouped by the block they face, so the chunk block A writes for block B
// is contiguous and B reads it coalesced.
int flow_build(Handle *h, const int32_t *order)
{
    const int n = h->n_cells;
    const size_t S = h->stride;
    if (h->npc != 3 || h->h_nbr.empty()) return SWE2D_OK;
    const int *nbr = h->h_nbr.data();
    const int nb = (n + SWE_BLOCK - 1)/SWE_BLOCK;
    std::vector<int> fcell((size_t)nb*SWE_BLOCK, -1), fpos((size_t)n, -1);
    for (int pp = 0; pp < n; pp++) {
        const int c = order ? order[pp] : pp;
        if (c < 0 || c >= n || fpos[c] >= 0) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "flow order: not a permutation of the cells");
        fcell[pp] = c;
        fpos[c] = pp;
    }
    for (int pp = n; pp < nb*SWE_BLOCK; pp++) fcell[pp] = -1 - fcell[n - 1];          // padding lanes mimic the last cell
    struct Rim { int nbblock, pos, f; };
    std::vector<int> own((size_t)3*nb*SWE_BLOCK, -1);                                // global slot of (position, f)
    std::vector<int2> blk((size_t)nb, int2{0, 0});
    int n_slots = 0;
    bool too_many = false;
    int max_rim = 0;
    std::vector<Rim> rim;
    for (int b = 0; b < nb; b++) {
        rim.clear();
        for (int pp = b*SWE_BLOCK; pp < std::min(n, (b + 1)*SWE_BLOCK); pp++)
            for (int f = 0; f < 3; f++) {
                const int code = nbr[(size_t)f*S + fcell[pp]];
                if (code >= 0 && fpos[code >> 2]/SWE_BLOCK != b) rim.push_back(Rim{fpos[code >> 2]/SWE_BLOCK, pp, f});
            }
        std::sort(rim.begin(), rim.end(), [](const Rim &x, const Rim &y) {
            return x.nbblock != y.nbblock ? x.nbblock < y.nbblock : (x.pos != y.pos ? x.pos < y.pos : x.f < y.f); });
        if ((int)rim.size() > SWE_FLOW_MAX_RIM) too_many = true;      // the kernel's staging area holds SWE_FLOW_MAX_RIM facets
        max_rim = std::max(max_rim, (int)rim.size());
        blk[b] = int2{n_slots, (int)rim.size()};
        for (const Rim &r : rim) own[(size_t)3*r.pos + r.f] = n_slots++;
    }
    // incoming list of a block: the slots its neighbours write for it, neighbour by neighbour in THEIR slot order;
    // entry = producer's slot << 6 | lane of the consuming cell; xin(position, f) = place of the slot facing (position, f)
    std::vector<int> xsrc((size_t)std::max(n_slots, 1), 0), xin((size_t)3*nb*SWE_BLOCK, -1);
    std::vector<std::pair<int, int>> inc;                                            // (producer's slot, consumer position*4 + f)
    for (int b = 0; b < nb; b++) {
        inc.clear();
        for (int pp = b*SWE_BLOCK; pp < std::min(n, (b + 1)*SWE_BLOCK); pp++)
            for (int f = 0; f < 3; f++)
                if (own[(size_t)3*pp + f] >= 0) {
                    const int code = nbr[(size_t)f*S + fcell[pp]];
                    inc.push_back({own[(size_t)3*fpos[code >> 2] + (code & 3)], (pp << 2) | f});
                }
        std::sort(inc.begin(), inc.end());                                           // by producer's slot = by neighbour block, then its order
        for (size_t i = 0; i < inc.size(); i++) {
            const int pp = inc[i].second >> 2, f = inc[i].second & 3;
            xsrc[(size_t)blk[b].x + i] = (inc[i].first << 6) | (pp & (SWE_BLOCK - 1));
            xin[(size_t)3*pp + f] = (int)i;
        }
    }
    // per position: {my slot of facet 0, 1, 2 counted from the block's first (-1: not a rim facet), w}, {w, w} with w = place of the
    // incoming slot (rim facet) or the lane of the neighbour inside the block (this lane itself for a boundary facet)
    std::vector<int4> p4((size_t)nb*SWE_BLOCK, int4{-1, -1, -1, 0});
    std::vector<int2> p2((size_t)nb*SWE_BLOCK, int2{0, 0});
    for (int pp = 0; pp < nb*SWE_BLOCK; pp++) {
        if (pp >= n) { p4[pp] = int4{-1, -1, -1, pp & (SWE_BLOCK - 1)}; p2[pp] = int2{pp & (SWE_BLOCK - 1), pp & (SWE_BLOCK - 1)}; continue; }
        const int b0 = blk[pp/SWE_BLOCK].x;
        int lo[3], w[3];
        for (int f = 0; f < 3; f++) {
            const int code = nbr[(size_t)f*S + fcell[pp]];
            const int o = own[(size_t)3*pp + f];
            lo[f] = o >= 0 ? o - b0 : -1;
            w[f] = o >= 0 ? xin[(size_t)3*pp + f] : (code >= 0 ? (fpos[code >> 2] & (SWE_BLOCK - 1)) : (pp & (SWE_BLOCK - 1)));
        }
        p4[pp] = int4{lo[0], lo[1], lo[2], w[0]};
        p2[pp] = int2{w[1], w[2]};
    }
    // (slot << 6 must fit an int, the exchange array must stay below SWE_FLOW_NOWHERE)
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (void *ptr : {(void *)h->flow_xblk, (void *)h->flow_xsrc, (void *)h->flow_xo4, (void *)h->flow_xo2, (void *)h->flow_ex, (void *)h->flow_cell})
        if (ptr) (void)hipFree(ptr);
    h->flow_xblk = nullptr; h->flow_xsrc = nullptr; h->flow_xo4 = nullptr; h->flow_xo2 = nullptr; h->flow_ex = nullptr; h->flow_cell = nullptr;
    // no flow kernel for this handle / this order: a block with more rim facets than the staging area holds (cells numbered without
    // locality), or slot numbers that do not fit
    if (too_many || !((size_t)3*n_slots*SWE_FLOW_SLOT_BYTES < ((size_t)1 << 31) && n_slots < (1 << 25))) return SWE2D_OK;
    h->flow_fpos = fpos;
    h->flow_max_rim = max_rim;
    h->flow_x_ready = false;
    h->flow_parity_bytes = (unsigned)((size_t)std::max(n_slots, 1)*SWE_FLOW_SLOT_BYTES);
    h->flow_ex_bytes = (size_t)3*h->flow_parity_bytes;
    HIP_TRY(h, hipMalloc(&h->flow_xblk, blk.size()*sizeof(int2)));
    HIP_TRY(h, hipMemcpy(h->flow_xblk, blk.data(), blk.size()*sizeof(int2), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMalloc(&h->flow_xsrc, xsrc.size()*sizeof(int)));
    HIP_TRY(h, hipMemcpy(h->flow_xsrc, xsrc.data(), xsrc.size()*sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMalloc(&h->flow_cell, fcell.size()*sizeof(int)));
    HIP_TRY(h, hipMemcpy(h->flow_cell, fcell.data(), fcell.size()*sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMalloc(&h->flow_xo4, p4.size()*sizeof(int4)));
    HIP_TRY(h, hipMalloc(&h->flow_xo2, p2.size()*sizeof(int2)));
    HIP_TRY(h, hipMemcpy(h->flow_xo4, p4.data(), p4.size()*sizeof(int4), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->flow_xo2, p2.data(), p2.size()*sizeof(int2), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMalloc(&h->flow_ex, h->flow_ex_bytes));
    HIP_TRY(h, hipMemset(h->flow_ex, 0, h->flow_ex_bytes));
    // the stage counters restart with the slots
    HIP_TRY(h, hipMemset(h->flow_flag, 0, (size_t)h->flow_blocks*SWE_FLOW_FLAG_STRIDE*sizeof(unsigned)));
    return SWE2D_OK;
}

// the configurations the flow kernel covers: triangles, no viscosity; wetting-drying since round 5 (swe_flow_kernel<..., WD>, nonlinear
// equations as in the stage kernels; SWE2D_OPT_FLOW_WD = 0 leaves it to the stage launches)
bool flow_kernel_covers(const Handle *h)
{
    if (h->wd && (!h->par.use_nonlinear_equations || !opt_on(h, SWE2D_OPT_FLOW_WD))) return false;
    return h->npc == 3 && !h->visc && h->idx4 && h->flow_flag && h->flow_ex && h->opt[SWE2D_OPT_BND_INLINE] != 0;
}

// Resident one-wave workgroups of the flow kernel: every block of a launch must be resident (a block waits for its
// neighbours' flags), so the grid must not exceed what the device holds at once.
int flow_capacity(Handle *h)
{
    if (h->flow_capacity >= 0) return h->flow_capacity;
    h->flow_capacity = 0;
    int per_cu = 0, dev_cus = 0;
    flow_kernel_t kern = pick_flow_kernel(true, true, true, true, 9);        // the largest variant
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern), SWE_BLOCK, 0) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess) return 0;
    h->flow_capacity = h->opt[SWE2D_OPT_FLOW_CAPACITY] >= 0 ? h->opt[SWE2D_OPT_FLOW_CAPACITY] : per_cu*dev_cus;    // (tests force the limit)
    return h->flow_capacity;
}

// FX launches: the places of every flow position's cell in the halo lists, the blocks that hold send / ghost cells
int flow_build_exchange(Handle *h)
{
    if (h->flow_x_ready) return SWE2D_OK;
    const int np = h->flow_blocks*SWE_BLOCK;
    if ((int)h->flow_fpos.size() != h->n_cells) return fail(h, SWE2D_ERR_UNSUPPORTED, "flow: no tables");
    std::vector<int2> xs((size_t)np, int2{-1, -1});
    std::vector<int> xr((size_t)np, -1);
    for (int j = 0; j < h->n_send; j++) {
        int2 &e = xs[h->flow_fpos[h->h_send[j]]];
        if (e.x < 0) e.x = j;
        else if (e.y < 0) e.y = j;
        else return fail(h, SWE2D_ERR_UNSUPPORTED, "flow with the exchange inside: a cell is sent to more than two peers");
    }
    for (int j = 0; j < h->n_recv; j++) xr[h->flow_fpos[h->h_recv[j]]] = j;
    h->flow_push_blocks = h->flow_recv_blocks = 0;
    for (int b = 0; b < h->flow_blocks; b++) {
        bool anys = false, anyr = false;
        for (int l = 0; l < SWE_BLOCK; l++) { anys = anys || xs[(size_t)b*SWE_BLOCK + l].x >= 0; anyr = anyr || xr[(size_t)b*SWE_BLOCK + l] >= 0; }
        h->flow_push_blocks += anys; h->flow_recv_blocks += anyr;
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (!h->flow_xsend) HIP_TRY(h, hipMalloc(&h->flow_xsend, (size_t)np*sizeof(int2)));
    if (!h->flow_xrecv) HIP_TRY(h, hipMalloc(&h->flow_xrecv, (size_t)np*sizeof(int)));
    if (!h->flow_xtick) {
        HIP_TRY(h, hipMalloc(&h->flow_xtick, (2*SWE_FLOW_MAX_CYCLES + 32)*sizeof(unsigned)));
        HIP_TRY(h, hipMemset(h->flow_xtick, 0, (2*SWE_FLOW_MAX_CYCLES + 32)*sizeof(unsigned)));
    }
    HIP_TRY(h, hipMemcpy(h->flow_xsend, xs.data(), (size_t)np*sizeof(int2), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->flow_xrecv, xr.data(), (size_t)np*sizeof(int), hipMemcpyHostToDevice));
    h->flow_x_ready = true;
    return SWE2D_OK;
}

// n_stages stages (a multiple of 3) on the ranges [0, cell_end[s]) in ONE launch; n_cycles > 0: n_cycles exchange cycles of
// n_stages stages each with the peer-to-peer halo exchange (channel 0) inside the launch
int launch_flow(Handle *h, int n_stages, const int32_t *cell_end, int n_cycles)
{
    if (!flow_kernel_covers(h)) return fail(h, SWE2D_ERR_UNSUPPORTED, "the flow kernel covers triangles without viscosity (wetting-drying: nonlinear equations)");
    const bool fx = n_cycles > 0;
    const int total = n_stages*(fx ? n_cycles : 1);
    if (n_stages <= 0 || n_stages % 3 != 0 || total > SWE_FLOW_MAX_STAGES || n_cycles > SWE_FLOW_MAX_CYCLES)
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "flow: n_stages must be a multiple of 3, at most 384 stages and 64 cycles per launch");
    for (int s = 0; s < n_stages; s++)
        if (cell_end[s] < 0 || cell_end[s] > h->n_cells || (s > 0 && cell_end[s] > cell_end[s - 1]))
            return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "flow: the stage ranges must shrink and stay inside the mesh");
    const int grid = ((h->flow_blocks + 7)/8)*8;
    if (grid > flow_capacity(h))
        return fail(h, SWE2D_ERR_UNSUPPORTED, "flow: more 64-cell blocks than the device holds resident at once");
    SweFlowArgs q{};
    if (fx) {
        auto &z = h->p2p;
        const int ch = z.n_channels - 1;                     // the granule channel: the last one, nine 16-byte granules per cell
        if (!z.zone || !z.ctr || z.n_peers == 0 || z.n_from == 0 || h->n_send == 0 || h->n_recv == 0)
            return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "flow with the exchange inside: the peer-to-peer halo is not connected");
        if (ch < 0 || z.width[ch] != 18)
            return fail(h, SWE2D_ERR_UNSUPPORTED, "flow with the exchange inside: the last peer-to-peer channel must have width 18 (nine granules per cell)");
        if (int rc = flow_build_exchange(h)) return rc;
        q.n_cycles = n_cycles; q.stages_per_cycle = n_stages;
        q.xsend = h->flow_xsend; q.xrecv = h->flow_xrecv; q.xtick = h->flow_xtick;
        q.xctr = z.ctr + ch;
        q.x_n_peers = z.n_peers;
        for (int i = 0; i < z.n_peers; i++) {                // as swe2d_p2p_push
            if (i > 0 && z.off[i] < z.off[i - 1]) return fail(h, SWE2D_ERR_UNSUPPORTED, "flow with the exchange inside: send segments must be sorted by offset");
            q.x_off[i] = z.off[i];
            char *base = z.remote_base[i];
            q.x_rdata[i] = base + p2p_channel_offset(z.width, ch, z.remote_n_recv[i]) + (size_t)z.remote_off[i]*144;
            q.x_rslot[i] = (unsigned)((size_t)z.remote_n_recv[i]*144);
            // my segment ends cnt cells after its start in both slots: the resource covers slot 0 .. the end of my segment in slot 1
            q.x_rbytes[i] = q.x_rslot[i] + (unsigned)((size_t)z.cnt[i]*144);
        }
        char *mine = static_cast<char *>(z.zone);            // as swe2d_p2p_wait_unpack
        q.x_zone = mine + p2p_channel_offset(z.width, ch, h->n_recv);
        q.x_slot = (unsigned)((size_t)h->n_recv*144);
        q.x_zbytes = 2*q.x_slot;
        q.x_timeout = (unsigned long long)(opt_seconds(h, SWE2D_OPT_P2P_TIMEOUT_MS, 5.0)*1e8);
    }
    fill_stage_args(h, q.st, 0, 0, 1, 0.0, 1.0, 1.0, 0, 0);
    for (int i = 0; i < 3; i++) q.buf[i] = h->state[i];
    q.flag = h->flow_flag; q.status = h->flow_status;
    q.xo4 = h->flow_xo4; q.xo2 = h->flow_xo2; q.ex = h->flow_ex;
    q.xblk = h->flow_xblk; q.xsrc = h->flow_xsrc; q.parity_bytes = h->flow_parity_bytes;
    q.fcell = h->flow_cell;
    q.n_blocks = h->flow_blocks; q.n_stages = total;
    for (int s = 0; s < SWE_FLOW_MAX_STAGES; s++) q.cell_end[s] = s < n_stages ? cell_end[s] : 0;
    for (int s = 0; s < 3; s++) { q.a0[s] = s ? kAlpha0[s] : 0.0; q.a1[s] = s ? kAlphaIn[s] : 1.0; q.beta[s] = kBeta[s]; }
    q.timeout_ticks = (unsigned long long)(opt_seconds(h, SWE2D_OPT_FLOW_TIMEOUT_MS, 2.0)*1e8);
    // granule loads per polling trip: by the blocks' rim facets (a trip covers 10.7 facets per load); a wider instance may be forced
    // (A/B, tests), a narrower one only makes more trips per pass
    int poll = h->flow_max_rim > 64 ? 9 : (h->flow_max_rim > 42 ? 6 : (h->flow_max_rim > 32 ? 4 : 3));
    if (const int o = h->opt[SWE2D_OPT_FLOW_POLL]; o > 0) poll = o > 8 ? 9 : (o > 4 ? 6 : (o > 3 ? 4 : 3));
    flow_kernel_t kern = h->wd ? pick_flow_kernel_wd(h->par.use_lax_friedrichs_velocity != 0, has_sources(h), fx, poll)
                               : pick_flow_kernel(h->par.use_nonlinear_equations != 0, h->par.use_lax_friedrichs_velocity != 0, has_sources(h), fx,
                                                  poll);
    SWE_CHK_SYNC(h->stream);
    // Every block of a flow launch must be resident at once, and flow_capacity counts the whole device: two flow launches of
    // DIFFERENT handles (streams) of this process on one device could each get a part of it and wait for their missing blocks
    // until the timeout.  Launches that do not exchange with a peer are therefore chained per device: a launch waits for the
    // previous flow launch of another handle (an event wait on the stream, no host synchronisation).  FX launches are left alone
    // (peers inside one process must run side by side; across processes DistributedSwe2d does not choose the flow path by itself
    // when ranks share a device), and so are launches under stream capture (one handle per graph).
    bool chained = false;
    if (!fx) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const bool capturing = h->stream && hipStreamIsCapturing(h->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
        chained = !capturing && h->device >= 0 && h->device < kFlowChainDevices;
    }
    std::unique_lock<std::mutex> lock(g_flow_chain_mu, std::defer_lock);
    if (chained) {
        lock.lock();
        FlowChain &fc = g_flow_chain[h->device];
        if (fc.ev && fc.last_uid != h->uid) HIP_TRY(h, hipStreamWaitEvent(h->stream, fc.ev, 0));
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SWE_BLOCK), 0, h->stream, q);
    HIP_TRY(h, hipGetLastError());
    if (chained) {
        FlowChain &fc = g_flow_chain[h->device];
        if (!fc.ev) HIP_TRY(h, hipEventCreateWithFlags(&fc.ev, hipEventDisableTiming));
        HIP_TRY(h, hipEventRecord(fc.ev, h->stream));
        fc.last_uid = h->uid;
    }
    h->flow_used = true;
    h->stage_valid[0] = h->stage_valid[1] = false;           // the stage solutions stay in registers
    return SWE2D_OK;
}

// after a synchronisation of the stream: did a wave of a flow launch give up waiting?  (then the state is wrong)
int flow_check(Handle *h)
{
    if (!h->flow_used || !h->flow_status) return SWE2D_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (h->stream && hipStreamIsCapturing(h->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return SWE2D_OK;
    unsigned st[2] = {0u, 0u};
    HIP_TRY(h, hipMemcpyAsync(st, h->flow_status, sizeof(st), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->flow_used = false;
    if (st[0] == 0u) return SWE2D_OK;
    // leave the handle usable: counters and flags back to a consistent start
    (void)hipMemsetAsync(h->flow_status, 0, 4*sizeof(unsigned), h->stream);
    (void)hipMemsetAsync(h->flow_flag, 0, (size_t)h->flow_blocks*SWE_FLOW_FLAG_STRIDE*sizeof(unsigned), h->stream);
    (void)hipMemsetAsync(h->flow_ex, 0, h->flow_ex_bytes, h->stream);
    (void)hipStreamSynchronize(h->stream);
    char msg[200];
    std::snprintf(msg, sizeof(msg), "flow kernel: %u block waits timed out (first: block %u) - blocks not resident together? The state is invalid",
                  st[0], st[1] - 1u);
    return fail(h, SWE2D_ERR_HIP, msg);
}

}  // namespace swe2d_impl

extern "C" {

int swe2d_solve_flow(swe2d_handle *hh, int32_t n_stages, const int32_t *cell_end)
{
    Handle *h = H(hh);
    if (!h || !cell_end) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    RoctxRange range(h, "swe2d_solve_flow");
    return launch_flow(h, n_stages, cell_end);
}

int swe2d_solve_flow_exchange(swe2d_handle *hh, int32_t n_cycles, int32_t stages_per_cycle, const int32_t *cell_end)
{
    Handle *h = H(hh);
    if (!h || !cell_end || n_cycles < 1) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    RoctxRange range(h, "swe2d_solve_flow_exchange");
    return launch_flow(h, stages_per_cycle, cell_end, n_cycles);
}

int swe2d_flow_unpack_pending(swe2d_handle *hh)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    auto &z = h->p2p;
    const int ch = z.n_channels - 1;
    if (!z.zone || !z.ctr || ch < 0 || z.width[ch] != 18 || !h->flow_status)
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_flow_unpack_pending: no granule channel");
    if (h->n_recv == 0) return SWE2D_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    char *mine = static_cast<char *>(z.zone);
    const unsigned slot = (unsigned)((size_t)h->n_recv*144);
    hipLaunchKernelGGL(swe_flow_unpack_kernel, dim3(std::min(256, grid_for(h->n_recv))), dim3(256), 0, h->stream, h->state[0], h->stride,
                       h->recv_cells, h->n_recv, (void *)(mine + p2p_channel_offset(z.width, ch, h->n_recv)), 2*slot, slot, z.ctr + ch,
                       h->flow_status, (unsigned long long)(opt_seconds(h, SWE2D_OPT_P2P_TIMEOUT_MS, 5.0)*1e8));
    HIP_TRY(h, hipGetLastError());
    h->flow_used = true;
    return SWE2D_OK;
}

int swe2d_flow_prepare_exchange(swe2d_handle *hh)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    return flow_build_exchange(h);
}

int swe2d_flow_set_order(swe2d_handle *hh, const int32_t *cells_in_flow_order)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    if (!h->flow_flag) return fail(h, SWE2D_ERR_UNSUPPORTED, "the flow kernel covers triangles");
    HIP_TRY(h, hipSetDevice(h->device));
    if (int rc = flow_check(h)) return rc;
    return flow_build(h, cells_in_flow_order);
}

int swe2d_flow_supported(swe2d_handle *hh)
{
    Handle *h = H(hh);
    if (!h || !flow_kernel_covers(h)) return 0;
    if (hipSetDevice(h->device) != hipSuccess) return 0;
    return ((h->flow_blocks + 7)/8)*8 <= flow_capacity(h) ? (has_sources(h) ? 1 : 2) : 0;
}

int swe2d_flow_status(swe2d_handle *hh, int32_t *timeouts)
{
    Handle *h = H(hh);
    if (!h || !timeouts) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null argument");
    *timeouts = 0;
    if (!h->flow_status) return SWE2D_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    unsigned st[2] = {0u, 0u};
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(st, h->flow_status, sizeof(st), hipMemcpyDeviceToHost));
    *timeouts = (int32_t)st[0];
    return SWE2D_OK;
}

// test hook: adds `delta` to the stage counter of one block (tests/test_gpu_flow_kernel.py: a block whose neighbours wait for it)
int swe2d_debug_flow_poke(swe2d_handle *hh, int32_t block, int32_t delta)
{
    Handle *h = H(hh);
    if (!h || !h->flow_flag || block < 0 || block >= h->flow_blocks) return SWE2D_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    unsigned v = 0;
    HIP_TRY(h, hipMemcpy(&v, h->flow_flag + (size_t)block*SWE_FLOW_FLAG_STRIDE, sizeof(v), hipMemcpyDeviceToHost));
    v += (unsigned)delta;
    HIP_TRY(h, hipMemcpy(h->flow_flag + (size_t)block*SWE_FLOW_FLAG_STRIDE, &v, sizeof(v), hipMemcpyHostToDevice));
    return SWE2D_OK;
}

// test hook of the -DSWE_FLOW_DELAY build (csrc/swe2d_flow.h): block `block` of every flow launch of this process sleeps
// `microseconds` at the points in `where` of every `every`-th stage; block < 0 switches it off.  SWE2D_ERR_UNSUPPORTED in the product build.
int swe2d_debug_flow_delay(swe2d_handle *hh, int32_t block, int32_t where, int32_t microseconds, int32_t every)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
#ifdef SWE_FLOW_DELAY
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const int cfg[4] = {block, where, microseconds*100, every < 1 ? 1 : every};
    if (swe_flow_debug_config(0, cfg) != 0) return fail(h, SWE2D_ERR_HIP, "swe2d_debug_flow_delay: the switch could not be set");
    return SWE2D_OK;
#else
    (void)block; (void)where; (void)microseconds; (void)every;
    return fail(h, SWE2D_ERR_UNSUPPORTED, "swe2d_debug_flow_delay: this library was built without -DSWE_FLOW_DELAY");
#endif
}

// test hook of the -DSWE_FLOW_TEAR build (csrc/swe2d_flow.h): the granule stores of block `block` (-2: of every block, -1: off) of
// every `every`-th publish are made in two halves - the half with the NEW tag first, the value `microseconds` later; across_ranks:
// also the pushes into the peers' landing zones.  SWE2D_ERR_UNSUPPORTED in the product build.
int swe2d_debug_flow_tear(swe2d_handle *hh, int32_t block, int32_t microseconds, int32_t every, int32_t across_ranks)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
#ifdef SWE_FLOW_TEAR
    if (block == -3) {                       // query: does this build's consumer test the check word?  (1 yes, 0: -DSWE_FLOW_NOCHECK, the negative control)
#ifdef SWE_FLOW_NOCHECK
        return 0;
#else
        return 1;
#endif
    }
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const int cfg[4] = {block, microseconds*100, every < 1 ? 1 : every, across_ranks ? 1 : 0};
    if (swe_flow_debug_config(1, cfg) != 0) return fail(h, SWE2D_ERR_HIP, "swe2d_debug_flow_tear: the switch could not be set");
    return SWE2D_OK;
#else
    (void)block; (void)microseconds; (void)every; (void)across_ranks;
    return fail(h, SWE2D_ERR_UNSUPPORTED, "swe2d_debug_flow_tear: this library was built without -DSWE_FLOW_TEAR");
#endif
}

}  // extern "C"
