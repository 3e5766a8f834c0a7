// swe2d_api_fuse.hip - stages 1 + 2 of a step in one launch by overlapped tiles (swe2d_fuse.h): instances, tile tables, launch
#include "swe2d_handle.h"
#include "swe2d_fuse.h"

namespace swe2d_impl {

namespace {
typedef void (*fuse_kernel_t)(const SweFuseArgs);
template <bool SRC>
fuse_kernel_t pick_fuse_src(bool nl, bool lf)
{
    if (nl) return lf ? swe_fuse12_kernel<true, true, SRC> : swe_fuse12_kernel<true, false, SRC>;
    return lf ? swe_fuse12_kernel<false, true, SRC> : swe_fuse12_kernel<false, false, SRC>;
}
fuse_kernel_t pick_fuse_kernel(bool nl, bool lf, bool src) { return src ? pick_fuse_src<true>(nl, lf) : pick_fuse_src<false>(nl, lf); }
}  // namespace

// What the kernel covers: triangles, no wetting-drying, no viscosity - with or without source terms (the SRC instances keep the 168
// VGPRs / three workgroups per CU of the plain ones, tools/kres.py), the whole mesh or a partition's owned + ghost cells on its
// shrinking stage ranges (round 6); taken from
// 250 k cells, where a step streams from memory (same box, us per step, stage launches -> fused pair + stage 3, device numbering in
// 16 x 6-quad tiles: 125 k cells 23.9 -> 24.1, 250 k 38.8 -> 37.0, 500 k 64.9 -> 61.1, 1 M 121.0 -> 107.3, 2 M 264 -> 235, 4 M 525 -> 477;
// profiles/r05zl_fused_stage_pair.txt), and where the numbering gives tiles worth it (mean interior >= 176 of 192 cells: the
// structured tile order, and the Hilbert order of an unstructured mesh - 1 M Delaunay triangles 192.0 + 49.9 cells per tile, 120.8 ->
// 113.3 us per step; an order that does not keeps its stage launches).
// SWE2D_OPT_FUSED_STAGES = 0: never; = 1: on every mesh (of at least 64 cells) whatever its tiles.  (In the range-checked build too since
// round 6: the shared functions test their LDS indices against the array they are handed, the tile tables are host-built indices.)
static bool fuse_applies(const Handle *h)                      // what every fused kernel needs, whatever the size
{
    if (h->opt[SWE2D_OPT_FUSED_STAGES] == 0) return false;
    if (h->opt[SWE2D_OPT_BND_INLINE] == 0) return false;     // the epilogue variant was asked for
    return !h->wd && !h->visc && !h->h_nbr.empty();
}

bool fuse12_covers(const Handle *h)
{
    const int mode = h->opt[SWE2D_OPT_FUSED_STAGES];         // -1, 2: by size and tile quality; 1, 3: forced
    if (!fuse_applies(h) || h->fuse_state == -1) return false;
    const bool forced = mode == 1 || mode == 3;
    // triangles: whole meshes and partitions, from 250 k cells
    if (h->npc == 3) return h->idx4 != nullptr && h->n_cells >= (forced ? 64 : 250000);
    // quadrilaterals (swe_fuse12_quad_kernel, round 6): whole meshes, from the size at which the three state buffers (3 x 96 B per cell)
    // leave the Infinity Cache - same box, us per step without -> with: 1 M cells 188.9 -> 172.6, + Manning 221.1 -> 212.0, cfg 4 338.8 ->
    // 329.3; 640 k cells 115.7 -> 114.6, + Manning 136.8 -> 142.7 (profiles/r06g_quads*.txt)
    return h->n_owned == h->n_cells && h->n_cells >= (forced ? 64 : 850000);
}

// Tiles: consecutive cells of the device numbering (compact patches in the tile-Hilbert order) - or of the order handed in with
// swe2d_fused_set_order: a partition's ghost layers are appended to its numbering layer by layer, strips one cell wide whose tiles
// would be all ring - as long as the interior holds at most 192 cells and the ring - every cell that shares a facet with an
// interior cell - at most 64.
int fuseq_build(Handle *h);
int fuse12_build(Handle *h)
{
    if (h->npc == 4) return fuseq_build(h);
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
    const int *order = (int)h->fuse_order.size() == n ? h->fuse_order.data() : nullptr;
    for (int pos = 0; pos < n;) {
        cells.clear(); ring.clear();
        int n_ring = 0;
        while (pos < n && (int)cells.size() < SWE_FUSE_INNER) {
            const int kk = order ? order[pos] : pos;
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
    // tiles not worth it: the mean interior below 176 of 192 cells on a whole mesh, below 150 on a partition (its tiles along the cuts
    // and through the ghost layers are partial by construction)
    const bool forced = h->opt[SWE2D_OPT_FUSED_STAGES] == 1 || h->opt[SWE2D_OPT_FUSED_STAGES] == 3;
    if (!forced && (double)n/h->fuse_n_tiles < (h->n_owned == h->n_cells ? 176.0 : 150.0)) {
        h->fuse_state = -1; h->fuse_n_tiles = 0;
        return SWE2D_OK;
    }
    HIP_TRY(h, hipMalloc(&h->fuse_tile, tl.size()*sizeof(int2)));
    HIP_TRY(h, hipMalloc(&h->fuse_inner, inner.size()*sizeof(int)));
    HIP_TRY(h, hipMemcpy(h->fuse_tile, tl.data(), tl.size()*sizeof(int2), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->fuse_inner, inner.data(), inner.size()*sizeof(int), hipMemcpyHostToDevice));
    h->fuse_ring_cells = n_ring_total;
    return SWE2D_OK;
}

// stages 1 and 2 of a step: state buffer A (U(0)) -> state buffer C (U(2)) on the cells [0, cell_end); stage 3 follows as a stage launch
int launch_fuse12_quad(Handle *h, int cell_end);
int launch_fuse12(Handle *h, int cell_end)
{
    if (h->npc == 4) return launch_fuse12_quad(h, cell_end);
    if (int rc = fuse12_build(h)) return rc;
    if (!h->fuse_tile) return fail(h, SWE2D_ERR_UNSUPPORTED, "fused stage pair: no tile tables");
    SweFuseArgs q;
    fill_stage_args(h, q.st, 0, 0, 2, 0.0, 1.0, kBeta[0], 0, h->n_owned);
    q.st.idxc = h->opt[SWE2D_OPT_COMPACT_IDX] == 0 ? nullptr : h->idxc;      // (the 16-B connectivity records: a streaming kernel)
    q.tile = h->fuse_tile;
    q.n_inner = h->fuse_inner;
    q.n_tiles = h->fuse_n_tiles;
    q.cell_end = cell_end;
    q.beta1 = kBeta[0];
    q.a0_2 = kAlpha0[1]; q.a1_2 = kAlphaIn[1]; q.beta2 = kBeta[1];
    q.out = h->state[2];
    fuse_kernel_t kern = pick_fuse_kernel(h->par.use_nonlinear_equations != 0, h->par.use_lax_friedrichs_velocity != 0, has_sources(h));
    const int grid = ((h->fuse_n_tiles + 7)/8)*8;
    SWE_CHK_SYNC(h->stream);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SWE_FUSE_WG), 0, h->stream, q);
    HIP_TRY(h, hipGetLastError());
    h->stage_valid[0] = false; h->stage_valid[1] = true;    // U(1) never left the chip; buffer C holds U(2)
    return SWE2D_OK;
}

// ---- the stage pair on quadrilaterals (swe_fuse12_quad_kernel): tiles of up to 192 consecutive cells + their ring of at most 64
int fuseq_build(Handle *h)
{
    if (h->fuseq_tile || h->fuse_state == -1) return SWE2D_OK;
    {   // allocations and copies: not inside a stream capture
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(h->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return SWE2D_OK;
        (void)hipGetLastError();
    }
    const int n = h->n_cells;
    const size_t S = h->stride;
    const int *nbr = h->h_nbr.data();
    const int *order = (int)h->fuse_order.size() == n ? h->fuse_order.data() : nullptr;
    std::vector<int4> tl;
    std::vector<int> inner, ring, cells;
    std::vector<int> state((size_t)n, 0), lane_of((size_t)n, -1);      // 0 outside | 1 interior | 2 ring, of the tile being built
    long long n_ring_total = 0;
    for (int pos = 0; pos < n;) {
        cells.clear(); ring.clear();
        int n_ring = 0;
        while (pos < n && (int)cells.size() < SWE_QFUSE_INNER) {
            const int kk = order ? order[pos] : pos;
            int delta = state[kk] == 2 ? -1 : 0;
            for (int f = 0; f < 4; f++) {
                const int code = nbr[(size_t)f*S + kk];
                if (code >= 0 && state[code >> 2] == 0) {
                    bool dup = false;
                    for (int g = 0; g < f; g++) dup = dup || (nbr[(size_t)g*S + kk] >= 0 && (nbr[(size_t)g*S + kk] >> 2) == (code >> 2));
                    if (!dup) delta++;
                }
            }
            if (!cells.empty() && n_ring + delta > SWE_QFUSE_RING) break;
            if (n_ring + delta > SWE_QFUSE_RING) return fail(h, SWE2D_ERR_UNSUPPORTED, "fused stage pair: a cell with more neighbours than a ring holds");
            if (state[kk] == 2) n_ring--;
            state[kk] = 1;
            cells.push_back(kk);
            for (int f = 0; f < 4; f++) {
                const int code = nbr[(size_t)f*S + kk];
                if (code >= 0 && state[code >> 2] == 0) { state[code >> 2] = 2; ring.push_back(code >> 2); n_ring++; }
            }
            pos++;
        }
        const int ni = (int)cells.size();
        for (int c : ring) if (state[c] == 2) cells.push_back(c);
        const int nt = (int)cells.size();
        if (nt - ni != n_ring || nt > SWE_FUSE_WG) return fail(h, SWE2D_ERR_UNSUPPORTED, "fused stage pair: tile bookkeeping");
        for (int l = 0; l < nt; l++) lane_of[cells[l]] = l;
        const size_t base = tl.size();
        tl.resize(base + SWE_FUSE_WG, int4{-1, 0, 0, 0});
        int n_out = 0;
        for (int l = 0; l < nt; l++) {
            const int c = cells[l];
            unsigned w[4];
            for (int f = 0; f < 4; f++) {
                const int code = nbr[(size_t)f*S + c];
                if (code < 0) w[f] = (unsigned)l;                                    // boundary facet: the cell itself
                else if (state[code >> 2] != 0) w[f] = (unsigned)lane_of[code >> 2];
                else {
                    if (l < ni || n_out >= SWE_QFUSE_MAX_OUT) return fail(h, SWE2D_ERR_UNSUPPORTED, "fused stage pair: ring bookkeeping");
                    w[f] = 0x200u | (unsigned)n_out++;
                }
            }
            tl[base + l] = int4{c, (int)(w[0] | (w[1] << SWE_FUSE_FBITS) | (w[2] << (2*SWE_FUSE_FBITS))), (int)w[3], 0};
        }
        inner.push_back(ni);
        n_ring_total += nt - ni;
        for (int c : cells) { state[c] = 0; lane_of[c] = -1; }
    }
    h->fuseq_n_tiles = (int)inner.size();
    const bool forced = h->opt[SWE2D_OPT_FUSED_STAGES] == 1 || h->opt[SWE2D_OPT_FUSED_STAGES] == 3;
    if (!forced && (double)n/h->fuseq_n_tiles < 176.0) { h->fuse_state = -1; h->fuseq_n_tiles = 0; return SWE2D_OK; }   // tiles not worth it
    HIP_TRY(h, hipMalloc(&h->fuseq_tile, tl.size()*sizeof(int4)));
    HIP_TRY(h, hipMalloc(&h->fuseq_inner, inner.size()*sizeof(int)));
    HIP_TRY(h, hipMemcpy(h->fuseq_tile, tl.data(), tl.size()*sizeof(int4), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->fuseq_inner, inner.data(), inner.size()*sizeof(int), hipMemcpyHostToDevice));
    h->fuseq_ring_cells = n_ring_total;
    return SWE2D_OK;
}

namespace {
typedef void (*fuseq_kernel_t)(const SweFuseQuadArgs);
template <bool SRC, bool AFFINE>
fuseq_kernel_t pick_fuseq_2(bool nl, bool lf)
{
    if (nl) return lf ? swe_fuse12_quad_kernel<true, true, SRC, AFFINE> : swe_fuse12_quad_kernel<true, false, SRC, AFFINE>;
    return lf ? swe_fuse12_quad_kernel<false, true, SRC, AFFINE> : swe_fuse12_quad_kernel<false, false, SRC, AFFINE>;
}
fuseq_kernel_t pick_fuseq_kernel(bool nl, bool lf, bool src, bool affine)
{
    if (affine) return src ? pick_fuseq_2<true, true>(nl, lf) : pick_fuseq_2<false, true>(nl, lf);
    return src ? pick_fuseq_2<true, false>(nl, lf) : pick_fuseq_2<false, false>(nl, lf);
}
}  // namespace

int launch_fuse12_quad(Handle *h, int cell_end)
{
    if (int rc = fuseq_build(h)) return rc;
    if (!h->fuseq_tile) return fail(h, SWE2D_ERR_UNSUPPORTED, "fused stage pair: no tile tables");
    SweFuseQuadArgs q;
    fill_stage_args(h, q.st, 0, 0, 2, 0.0, 1.0, kBeta[0], 0, h->n_owned);
    q.tile = h->fuseq_tile;
    q.n_inner = h->fuseq_inner;
    q.n_tiles = h->fuseq_n_tiles;
    q.cell_end = cell_end;
    q.beta1 = kBeta[0];
    q.a0_2 = kAlpha0[1]; q.a1_2 = kAlphaIn[1]; q.beta2 = kBeta[1];
    q.out = h->state[2];
    fuseq_kernel_t kern = pick_fuseq_kernel(h->par.use_nonlinear_equations != 0, h->par.use_lax_friedrichs_velocity != 0, has_sources(h), h->affine);
    const int grid = ((h->fuseq_n_tiles + 7)/8)*8;
    SWE_CHK_SYNC(h->stream);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SWE_FUSE_WG), 0, h->stream, q);
    HIP_TRY(h, hipGetLastError());
    h->stage_valid[0] = false; h->stage_valid[1] = true;
    return SWE2D_OK;
}

// ---- all three stages in one launch (swe_fuse123_kernel): tiles with two rings
namespace {
typedef void (*fuse3_kernel_t)(const SweFuse3Args);
template <bool SRC>
fuse3_kernel_t pick_fuse3_src(bool nl, bool lf)
{
    if (nl) return lf ? swe_fuse123_kernel<true, true, SRC> : swe_fuse123_kernel<true, false, SRC>;
    return lf ? swe_fuse123_kernel<false, true, SRC> : swe_fuse123_kernel<false, false, SRC>;
}
fuse3_kernel_t pick_fuse3_kernel(bool nl, bool lf, bool src) { return src ? pick_fuse3_src<true>(nl, lf) : pick_fuse3_src<false>(nl, lf); }
}  // namespace

// Tiles: consecutive cells of the tile order as long as interior + ring 1 (facet neighbours of the interior) + ring 2 (facet
// neighbours of ring 1) fit the 256 lanes and ring 2's facets towards the outside fit the staging area - and up to the next
// position the caller marked as the start of a tile (swe2d_fused_set_triple_tiles: patches of 12 x 7 quads = 168 triangles + 38 + 42
// fill 248 lanes, where 147 consecutive cells of the 16 x 6 numbering leave ragged patches with rings of 52 + 57).
int fuse123_build(Handle *h)
{
    if (h->fuse3_tile) return SWE2D_OK;
    const int n = h->n_cells;
    const size_t S = h->stride;
    const int *nbr = h->h_nbr.data();
    const int *order = (int)h->fuse3_order.size() == n ? h->fuse3_order.data() : ((int)h->fuse_order.size() == n ? h->fuse_order.data() : nullptr);
    const unsigned char *start = (int)h->fuse3_start.size() == n ? h->fuse3_start.data() : nullptr;
    std::vector<int2> tl, cnt;
    std::vector<unsigned char> state((size_t)n, 0);                // 0 outside | 1 interior | 2 ring 1 | 3 ring 2, of the tile being built
    std::vector<int> lane_of((size_t)n, -1), touched, inner;
    std::vector<std::pair<int, unsigned char>> undo;
    int count[4] = {0, 0, 0, 0};
    long long r1_total = 0, r2_total = 0;
    auto set = [&](int c, unsigned char ns) {
        undo.push_back({c, state[c]});
        if (state[c] == 0) touched.push_back(c);
        count[state[c]]--; state[c] = ns; count[ns]++;
    };
    for (int pos = 0; pos < n;) {
        inner.clear(); touched.clear();
        count[1] = count[2] = count[3] = 0;
        while (pos < n) {
            if (start && start[pos] && !inner.empty()) break;      // the caller's tiles (swe2d_fused_set_triple_tiles): compact patches
            const int kk = order ? order[pos] : pos;
            undo.clear();
            const size_t touched_before = touched.size();
            set(kk, 1);
            for (int f = 0; f < 3; f++) {
                const int code = nbr[(size_t)f*S + kk];
                if (code < 0) continue;
                const int c1 = code >> 2;
                if (state[c1] == 0 || state[c1] == 3) {
                    set(c1, 2);
                    for (int g = 0; g < 3; g++) {
                        const int code2 = nbr[(size_t)g*S + c1];
                        if (code2 >= 0 && state[code2 >> 2] == 0) set(code2 >> 2, 3);
                    }
                }
            }
            if (count[1] + count[2] + count[3] > SWE_FUSE_WG || 2*count[3] > SWE_FUSE3_MAX_OUT) {
                if (inner.empty()) return fail(h, SWE2D_ERR_UNSUPPORTED, "fused stages: a cell whose two rings do not fit a tile");
                for (size_t i = undo.size(); i-- > 0;) { count[state[undo[i].first]]--; state[undo[i].first] = undo[i].second; count[undo[i].second]++; }
                touched.resize(touched_before);
                break;
            }
            inner.push_back(kk);
            pos++;
        }
        // lanes: the interior in the order it was added, ring 1, ring 2 (in the order of discovery)
        std::vector<int> cells(inner);
        for (int want = 2; want <= 3; want++)
            for (int c : touched) if (state[c] == want) cells.push_back(c);
        const int ni = (int)inner.size(), nm = ni + count[2], nt = (int)cells.size();
        if (nt != count[1] + count[2] + count[3] || ni != count[1] || nt > SWE_FUSE_WG) return fail(h, SWE2D_ERR_UNSUPPORTED, "fused stages: tile bookkeeping");
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
                    if (l < nm || n_out >= SWE_FUSE3_MAX_OUT) return fail(h, SWE2D_ERR_UNSUPPORTED, "fused stages: ring bookkeeping");
                    field = 0x200u | (unsigned)n_out++;
                }
                w |= field << (SWE_FUSE_FBITS*f);
            }
            tl[base + l] = int2{c, (int)w};
        }
        cnt.push_back(int2{ni, nm});
        r1_total += count[2]; r2_total += count[3];
        for (int c : cells) { state[c] = 0; lane_of[c] = -1; }
    }
    h->fuse3_n_tiles = (int)cnt.size();
    HIP_TRY(h, hipMalloc(&h->fuse3_tile, tl.size()*sizeof(int2)));
    HIP_TRY(h, hipMalloc(&h->fuse3_cnt, cnt.size()*sizeof(int2)));
    HIP_TRY(h, hipMemcpy(h->fuse3_tile, tl.data(), tl.size()*sizeof(int2), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->fuse3_cnt, cnt.data(), cnt.size()*sizeof(int2), hipMemcpyHostToDevice));
    h->fuse3_ring1 = r1_total; h->fuse3_ring2 = r2_total;
    return SWE2D_OK;
}

// a whole step: state buffer A (U(0)) -> state buffer B (U(3)), then the two change places
int launch_fuse123(Handle *h, int cell_end)
{
    if (int rc = fuse123_build(h)) return rc;
    SweFuse3Args q;
    fill_stage_args(h, q.st, 0, 0, 1, 0.0, 1.0, kBeta[0], 0, h->n_owned);
    q.st.idxc = h->opt[SWE2D_OPT_COMPACT_IDX] == 0 ? nullptr : h->idxc;
    q.tile = h->fuse3_tile;
    q.counts = h->fuse3_cnt;
    q.n_tiles = h->fuse3_n_tiles;
    q.cell_end = cell_end;
    for (int s = 0; s < 3; s++) { q.a0[s] = s ? kAlpha0[s] : 0.0; q.a1[s] = s ? kAlphaIn[s] : 1.0; q.beta[s] = kBeta[s]; }
    q.out = h->state[1];
    fuse3_kernel_t kern = pick_fuse3_kernel(h->par.use_nonlinear_equations != 0, h->par.use_lax_friedrichs_velocity != 0, has_sources(h));
    const int grid = ((h->fuse3_n_tiles + 7)/8)*8;
    SWE_CHK_SYNC(h->stream);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SWE_FUSE_WG), 0, h->stream, q);
    HIP_TRY(h, hipGetLastError());
    std::swap(h->state[0], h->state[1]);
    h->stage_valid[0] = h->stage_valid[1] = false;          // U(1) and U(2) never left the chip
    {   // inside a stream capture the swap is only the host's: a graph that holds an odd number of them ends on the other buffer than
        // it began on and cannot be replayed twice - counted here, reported by capture_parity_check at the next call outside the capture
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(h->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) h->capture_swaps++;
        (void)hipGetLastError();
    }
    return SWE2D_OK;
}

int capture_parity_check(Handle *h)
{
    if (h->capture_swaps == 0) return SWE2D_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(h->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return SWE2D_OK; }
    (void)hipGetLastError();
    const int swaps = h->capture_swaps;
    h->capture_swaps = 0;
    if (swaps & 1)
        return fail(h, SWE2D_ERR_UNSUPPORTED, "a stream capture recorded an odd number of swe2d_solve_step_cells launches: the graph ends on the other "
                    "state buffer than it began on and cannot be replayed; capture an even number per sequence");
    return SWE2D_OK;
}

// All three stages in one launch.  With tiles cut as consecutive cells of the numbering (147 + 52 + 57 per tile, ragged) the second ring
// costs 1.26 x the arithmetic of the pair and only pays where the state no longer fits the Infinity Cache - same box, us per step,
// three stage launches / fused pair + stage 3 / all three fused (profiles/r06b_fused_sizes.txt): 250 k cells 37.0 / 35.2 / 36.1,
// 1 M 110.0 / 103.0 / 110.8, 2 M 272.0 / 233.3 / 229.9, 4 M 527.9 / 475.3 / 452.1: by itself from 2.5 M cells.  With the caller's
// patches (swe2d_fused_set_triple_tiles: 11 x 8 quads of a RectangleMesh = 176 + 38 + 42 cells, every lane of the 256 used) it wins
// wherever the dataflow kernel does not apply (profiles/r06l_triple_tiles.txt, r06m_triple_sizes.txt): 150 k cells 28.3 / 27.5 / 24.1,
// 250 k 36.9 / 34.8 / 31.5, 500 k 62.4 / 57.9 / 53.8, 1 M - / 103.4 / 96.8, 2 M - / 235.4 / 207.5, 4 M - / 476.7 / 406.9: by itself from
// 131 073 cells.  Not with source terms (those instances need 187-199 VGPRs: at three workgroups per CU they spill 80-132 B per lane - 1 M
// cells 159-161 us per step by the pair, 235-239 by this kernel -, at two, as built, 195-196 against 163-164: profiles/r06q_*;
// SWE2D_OPT_FUSED_STAGES = 3 forces it, = 2 keeps the pair at every size).
// Whole meshes; not inside a stream capture (the launch swaps two state buffers on the host).
bool fuse123_wanted(const Handle *h)
{
    const int mode = h->opt[SWE2D_OPT_FUSED_STAGES];
    const bool patches = (int)h->fuse3_start.size() == h->n_cells;
    const int from = patches ? 131073 : 2500000;              // (131 072 cells = 2048 resident blocks: what the dataflow kernel holds)
    if (h->npc != 3 || !fuse_applies(h) || h->idx4 == nullptr || h->n_owned != h->n_cells) return false;
    return mode == 3 ? h->n_cells >= 64 : (mode == -1 && h->n_cells >= from && !has_sources(h));
}

// one SSPRK33 step of the shallow-water state on the whole mesh by the launches swe2d_advance would take when the dataflow kernel
// does not apply: the fused stage pair + stage 3 where it covers the handle, three stage launches otherwise
int step_swe(Handle *h)
{
    if (fuse123_wanted(h)) {
        // all three stages in one launch (whole meshes; the launch swaps two state buffers on the host: not inside a stream capture)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const bool capturing = hipStreamIsCapturing(h->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
        (void)hipGetLastError();
        if (!capturing) return launch_fuse123(h, h->n_owned);
    }
    if (fuse12_covers(h)) { if (int rc = fuse12_build(h)) return rc; }
    if (fuse12_covers(h) && (h->npc == 4 ? h->fuseq_tile != nullptr : h->fuse_tile != nullptr)) {
        if (int rc = launch_fuse12(h, h->n_owned)) return rc;
        return stage_on_range(h, 2, 0, h->n_owned);
    }
    for (int s = 0; s < 3; s++)
        if (int rc = stage_on_range(h, s, 0, h->n_owned)) return rc;
    return SWE2D_OK;
}

}  // namespace swe2d_impl

extern "C" {

int swe2d_fused_set_order(swe2d_handle *hh, const int32_t *cells_in_tile_order)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    std::vector<int> order;
    if (cells_in_tile_order) {
        std::vector<char> seen((size_t)h->n_cells, 0);
        for (int i = 0; i < h->n_cells; i++) {
            const int c = cells_in_tile_order[i];
            if (c < 0 || c >= h->n_cells || seen[c]) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "fused stages: the tile order is not a permutation of the cells");
            seen[c] = 1;
        }
        order.assign(cells_in_tile_order, cells_in_tile_order + h->n_cells);
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->fuse_tile) { (void)hipFree(h->fuse_tile); h->fuse_tile = nullptr; }
    if (h->fuse_inner) { (void)hipFree(h->fuse_inner); h->fuse_inner = nullptr; }
    if (h->fuseq_tile) { (void)hipFree(h->fuseq_tile); h->fuseq_tile = nullptr; }
    if (h->fuseq_inner) { (void)hipFree(h->fuseq_inner); h->fuseq_inner = nullptr; }
    h->fuseq_n_tiles = 0;
    if (h->fuse3_tile) { (void)hipFree(h->fuse3_tile); h->fuse3_tile = nullptr; }
    if (h->fuse3_cnt) { (void)hipFree(h->fuse3_cnt); h->fuse3_cnt = nullptr; }
    h->fuse3_n_tiles = 0;
    h->fuse_n_tiles = 0; h->fuse_ring_cells = 0;
    if (h->fuse_state == -1) h->fuse_state = 0;
    h->fuse_order.swap(order);
    return SWE2D_OK;
}

int swe2d_fused_set_triple_tiles(swe2d_handle *hh, const int32_t *cells_in_tile_order, const int32_t *tile_starts, int32_t n_starts)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    std::vector<int> order;
    std::vector<unsigned char> start;
    if (cells_in_tile_order) {
        std::vector<char> seen((size_t)h->n_cells, 0);
        for (int i = 0; i < h->n_cells; i++) {
            const int c = cells_in_tile_order[i];
            if (c < 0 || c >= h->n_cells || seen[c]) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "fused stages: the tile order is not a permutation of the cells");
            seen[c] = 1;
        }
        order.assign(cells_in_tile_order, cells_in_tile_order + h->n_cells);
    }
    if (n_starts < 0 || (n_starts > 0 && !tile_starts)) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "fused stages: tile starts");
    if (n_starts > 0) {
        start.assign((size_t)h->n_cells, 0);
        for (int i = 0; i < n_starts; i++) {
            if (tile_starts[i] < 0 || tile_starts[i] >= h->n_cells) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "fused stages: a tile start outside the cells");
            start[tile_starts[i]] = 1;
        }
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->fuse3_tile) { (void)hipFree(h->fuse3_tile); h->fuse3_tile = nullptr; }
    if (h->fuse3_cnt) { (void)hipFree(h->fuse3_cnt); h->fuse3_cnt = nullptr; }
    h->fuse3_n_tiles = 0;
    h->fuse3_order.swap(order);
    h->fuse3_start.swap(start);
    return SWE2D_OK;
}

int swe2d_fused_triple_info(swe2d_handle *hh, int32_t out[4])
{
    Handle *h = H(hh);
    if (!h || !out) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null argument");
    out[0] = out[1] = out[2] = out[3] = 0;
    if (!fuse123_wanted(h)) return SWE2D_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    if (int rc = fuse123_build(h)) return rc;
    out[0] = 1; out[1] = h->fuse3_n_tiles; out[2] = (int32_t)h->fuse3_ring1; out[3] = (int32_t)h->fuse3_ring2;
    return SWE2D_OK;
}

// A partition's whole step in one launch: what swe2d_solve_step_cells needs (the rule is the caller's - see swe2d_fused_step_info)
static bool fuse123_partition_ok(const Handle *h)
{
    const int mode = h->opt[SWE2D_OPT_FUSED_STAGES];
    if (h->npc != 3 || !fuse_applies(h) || h->idx4 == nullptr || mode == 2 || mode == 1) return false;
    return mode == 3 || !has_sources(h);
}

int swe2d_fused_step_info(swe2d_handle *hh, int32_t out[4])
{
    Handle *h = H(hh);
    if (!h || !out) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null argument");
    out[0] = out[1] = out[2] = out[3] = 0;
    // by itself where the caller handed in patches (swe2d_fused_set_triple_tiles) and the cell range is beyond the dataflow kernel's
    const int mode = h->opt[SWE2D_OPT_FUSED_STAGES];
    if (!fuse123_partition_ok(h)) return SWE2D_OK;
    if (mode != 3 && ((int)h->fuse3_start.size() != h->n_cells || h->n_cells <= 131072)) return SWE2D_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    if (int rc = fuse123_build(h)) return rc;
    out[0] = 1; out[1] = h->fuse3_n_tiles; out[2] = (int32_t)h->fuse3_ring1; out[3] = (int32_t)h->fuse3_ring2;
    return SWE2D_OK;
}

// All three stages of a step on a partition: stage 3 on [0, cell_end) - the last of the step's three shrinking ranges; the tiles
// evaluate stages 1 and 2 on supersets of theirs (every cell of a tile / interior + first ring), values that never leave the chip.
// U(3) goes to state buffer B and the two change places: cells of the new buffer A beyond cell_end hold what B held before - stale
// ghost values that the next exchange rewrites, and that no later stage of the cycle reads (its ranges end inside cell_end).
int swe2d_solve_step_cells(swe2d_handle *hh, int32_t cell_end)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    if (cell_end < 0 || cell_end > h->n_cells) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad cell range");
    if (!fuse123_partition_ok(h)) return fail(h, SWE2D_ERR_UNSUPPORTED, "swe2d_solve_step_cells: the three-stage kernel does not cover this handle");
    HIP_TRY(h, hipSetDevice(h->device));
    if (int rc = capture_parity_check(h)) return rc;
    if (!h->fuse3_tile) {           // tile tables: allocations and copies, not inside a stream capture
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const bool capturing = hipStreamIsCapturing(h->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
        (void)hipGetLastError();
        if (capturing) return fail(h, SWE2D_ERR_UNSUPPORTED, "swe2d_solve_step_cells: first call inside a stream capture (swe2d_fused_step_info builds the tables)");
    }
    return launch_fuse123(h, cell_end);
}

// stages 0 and 1 of a step on the ranges [0, cell_end_0) and [0, cell_end_1) (cell_end_1 <= cell_end_0, every cell of the second
// range with its facet neighbours inside the first: a partition's stage ranges): one fused launch where the kernel covers the
// handle, else the two stage launches.  Buffer C holds U(2) on [0, cell_end_1) afterwards either way.
int swe2d_solve_stage_pair_cells(swe2d_handle *hh, int32_t cell_end_0, int32_t cell_end_1)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    if (cell_end_1 < 0 || cell_end_1 > cell_end_0 || cell_end_0 > h->n_cells) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad cell ranges");
    HIP_TRY(h, hipSetDevice(h->device));
    if (fuse12_covers(h)) { if (int rc = fuse12_build(h)) return rc; }
    if (fuse12_covers(h) && (h->npc == 4 ? h->fuseq_tile != nullptr : h->fuse_tile != nullptr)) return launch_fuse12(h, cell_end_1);
    if (int rc = stage_on_range(h, 0, 0, cell_end_0)) return rc;
    return stage_on_range(h, 1, 0, cell_end_1);
}

}  // extern "C"
