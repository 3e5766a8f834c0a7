// swe2d_api_fuse.hip - stages 1 + 2 of a step in one launch by overlapped tiles (swe2d_fuse.h): instances, tile tables, launch
#include "swe2d_handle.h"
#include "swe2d_fuse.h"

namespace swe2d_impl {

namespace {
typedef void (*fuse_kernel_t)(const SweFuseArgs);
fuse_kernel_t pick_fuse_kernel(bool nl, bool lf)
{
    if (nl) return lf ? swe_fuse12_kernel<true, true> : swe_fuse12_kernel<true, false>;
    return lf ? swe_fuse12_kernel<false, true> : swe_fuse12_kernel<false, false>;
}
}  // namespace

// What the kernel covers: triangles, the whole mesh on one device, no source terms, no wetting-drying, no viscosity; taken from
// 250 k cells, where a step streams from memory (same box, us per step, stage launches -> fused pair + stage 3, device numbering in
// 16 x 6-quad tiles: 125 k cells 23.9 -> 24.1, 250 k 38.8 -> 37.0, 500 k 64.9 -> 61.1, 1 M 121.0 -> 107.3, 2 M 264 -> 235, 4 M 525 -> 477;
// profiles/r05zl_fused_stage_pair.txt), and where the numbering gives tiles worth it (mean interior >= 176 of 192 cells: the
// structured tile order, and the Hilbert order of an unstructured mesh - 1 M Delaunay triangles 192.0 + 49.9 cells per tile, 120.8 ->
// 113.3 us per step; an order that does not keeps its stage launches).
// SWE2D_OPT_FUSED_STAGES = 0: never; = 1: on every mesh of at least 768 cells whatever its tiles.  (Not in the range-checked build: the
// LDS index checks of the shared functions know the flow kernel's array only.)
bool fuse12_covers(const Handle *h)
{
#ifdef SWE_RANGE_CHECK
    return false;
#else
    const int mode = h->opt[SWE2D_OPT_FUSED_STAGES];         // -1: by size and tile quality
    if (mode == 0 || h->fuse_state == -1) return false;
    if (h->opt[SWE2D_OPT_BND_INLINE] == 0) return false;     // the epilogue variant was asked for
    return h->npc == 3 && !h->wd && !h->visc && !has_sources(h) && h->n_owned == h->n_cells && !h->h_nbr.empty() && h->idx4
           && h->n_cells >= (mode > 0 ? 4*SWE_FUSE_INNER : 250000);
#endif
}

// Tiles: consecutive cells of the device numbering (compact patches in the tile-Hilbert order) as long as the interior holds at
// most 192 cells and the ring - every cell that shares a facet with an interior cell - at most 64.
int fuse12_build(Handle *h)
{
    if (h->fuse_tile || h->fuse_state == -1) return SWE2D_OK;
    {   // allocations and copies: not inside a stream capture - such a capture keeps the stage launches, the next call outside one builds
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(h->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return SWE2D_OK;
        (void)hipGetLastError();
    }
    const int n = h->n_cells;
    const size_t S = h->stride;
    const int *nbr = h->h_nbr.data();
    std::vector<int2> tl;
    std::vector<int> inner;
    std::vector<int> state((size_t)n, 0), lane_of((size_t)n, -1);      // 0 outside | 1 interior | 2 ring, of the tile being built
    std::vector<int> ring, cells;
    long long n_ring_total = 0;
    for (int pos = 0; pos < n;) {
        cells.clear(); ring.clear();
        int n_ring = 0;
        while (pos < n && (int)cells.size() < SWE_FUSE_INNER) {
            const int kk = pos;
            int delta = state[kk] == 2 ? -1 : 0;
            for (int f = 0; f < 3; f++) {
                const int code = nbr[(size_t)f*S + kk];
                if (code >= 0 && state[code >> 2] == 0) {
                    bool dup = false;                            // (two facets never lead to the same neighbour on a valid mesh; cheap to be safe)
                    for (int g = 0; g < f; g++) dup = dup || (nbr[(size_t)g*S + kk] >= 0 && (nbr[(size_t)g*S + kk] >> 2) == (code >> 2));
                    if (!dup) delta++;
                }
            }
            if (!cells.empty() && n_ring + delta > SWE_FUSE_RING) break;
            if (n_ring + delta > SWE_FUSE_RING) return fail(h, SWE2D_ERR_UNSUPPORTED, "fused stage pair: a cell with more neighbours than a ring holds");
            if (state[kk] == 2) n_ring--;
            state[kk] = 1;
            cells.push_back(kk);
            for (int f = 0; f < 3; f++) {
                const int code = nbr[(size_t)f*S + kk];
                if (code >= 0 && state[code >> 2] == 0) { state[code >> 2] = 2; ring.push_back(code >> 2); n_ring++; }
            }
            pos++;
        }
        const int ni = (int)cells.size();
        // the ring in the order of discovery, without the cells that became interior later
        for (int c : ring) if (state[c] == 2) cells.push_back(c);
        const int nt = (int)cells.size();
        if (nt - ni != n_ring || nt > SWE_FUSE_WG) return fail(h, SWE2D_ERR_UNSUPPORTED, "fused stage pair: tile bookkeeping");
        for (int l = 0; l < nt; l++) lane_of[cells[l]] = l;
        const size_t base = tl.size();
        tl.resize(base + SWE_FUSE_WG, int2{-1, 0});
        int n_out = 0;
        for (int l = 0; l < nt; l++) {
            const int c = cells[l];
            unsigned w = 0u;
            for (int f = 0; f < 3; f++) {
                const int code = nbr[(size_t)f*S + c];
                unsigned field;
                if (code < 0) field = (unsigned)l;                                   // boundary facet: the cell itself
                else if (state[code >> 2] != 0) field = (unsigned)lane_of[code >> 2];
                else {
                    if (l < ni || n_out >= SWE_FUSE_MAX_OUT) return fail(h, SWE2D_ERR_UNSUPPORTED, "fused stage pair: ring bookkeeping");
                    field = 0x200u | (unsigned)n_out++;
                }
                w |= field << (SWE_FUSE_FBITS*f);
            }
            tl[base + l] = int2{c, (int)w};
        }
        inner.push_back(ni);
        n_ring_total += nt - ni;
        for (int c : cells) { state[c] = 0; lane_of[c] = -1; }
    }
    h->fuse_n_tiles = (int)inner.size();
    if (h->opt[SWE2D_OPT_FUSED_STAGES] <= 0 && (double)n/h->fuse_n_tiles < 176.0) { h->fuse_state = -1; h->fuse_n_tiles = 0; return SWE2D_OK; }   // tiles not worth it
    HIP_TRY(h, hipMalloc(&h->fuse_tile, tl.size()*sizeof(int2)));
    HIP_TRY(h, hipMalloc(&h->fuse_inner, inner.size()*sizeof(int)));
    HIP_TRY(h, hipMemcpy(h->fuse_tile, tl.data(), tl.size()*sizeof(int2), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->fuse_inner, inner.data(), inner.size()*sizeof(int), hipMemcpyHostToDevice));
    h->fuse_ring_cells = n_ring_total;
    return SWE2D_OK;
}

// stages 1 and 2 of a step: state buffer A (U(0)) -> state buffer C (U(2)); stage 3 follows as a stage launch
int launch_fuse12(Handle *h)
{
    if (int rc = fuse12_build(h)) return rc;
    SweFuseArgs q;
    fill_stage_args(h, q.st, 0, 0, 2, 0.0, 1.0, kBeta[0], 0, h->n_owned);
    q.st.idxc = h->idxc;                                  // (the 16-B connectivity records where they exist: a streaming kernel)
    q.tile = h->fuse_tile;
    q.n_inner = h->fuse_inner;
    q.n_tiles = h->fuse_n_tiles;
    q.beta1 = kBeta[0];
    q.a0_2 = kAlpha0[1]; q.a1_2 = kAlphaIn[1]; q.beta2 = kBeta[1];
    q.out = h->state[2];
    fuse_kernel_t kern = pick_fuse_kernel(h->par.use_nonlinear_equations != 0, h->par.use_lax_friedrichs_velocity != 0);
    const int grid = ((h->fuse_n_tiles + 7)/8)*8;
    SWE_CHK_SYNC(h->stream);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SWE_FUSE_WG), 0, h->stream, q);
    HIP_TRY(h, hipGetLastError());
    h->stage_valid[0] = false; h->stage_valid[1] = true;    // U(1) never left the chip; buffer C holds U(2)
    return SWE2D_OK;
}

}  // namespace swe2d_impl
