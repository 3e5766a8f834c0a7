// Compact triangle connectivity: 16 B per cell instead of 24 (round 5).  Plain C++ (no HIP types needed): the host packs, the
// kernels unpack, tests/test_host_logic.py compiles both with g++ and round-trips them.
//
// A stage reads 36 B of static data per triangle (12 B neighbour codes, 12 B vertex ids, half a vertex record) next to 144-216 B
// of state; in a numbering made for locality (tile / Hilbert order of the cells, first-touch order of the vertices) neighbours
// and vertices sit close to the cell, so differences take their place:
//   bits   0..25  vertex id of node 0                       (< 2^26)
//   bits  26..44  vertex id of node 1 - that of node 0      (19 bits, signed)
//   bits  45..63  vertex id of node 2 - that of node 0      (19 bits, signed)
//   bits  64..84, 85..105, 106..126  one 21-bit field per facet: [1:0] = 3 for a boundary facet, then [20:2] = its marker;
//                 otherwise the facet's number in the neighbour, and [20:2] = neighbour cell - this cell (19 bits, signed)
//   bit   127     escape: something did not fit (the seams of the space-filling curve: ~0.2 % of the cells of the bench mesh,
//                 in ~2 % of its waves) - the lane reads the wide records (SweStageArgs::idx4 / idx2) after all.
// The decoded values are the wide records' values: nothing downstream changes, and the results are the same bits
// (tests/test_gpu_parity.py::test_compact_connectivity_gives_the_bits_of_the_wide_records).
#pragma once
#ifndef __HIPCC__
struct int4 { int x, y, z, w; };
#define SWE_CONN_HD
#else
#define SWE_CONN_HD __host__ __device__
#endif
#define SWE_CONN_DBITS 19

SWE_CONN_HD inline bool swe_conn_fits(long long d) { return d >= -(1ll << (SWE_CONN_DBITS - 1)) && d < (1ll << (SWE_CONN_DBITS - 1)); }

// nb[f] = (cell << 2) | facet-in-neighbour or -(marker), vid[i] = vertex ids, of cell k
inline int4 swe_conn_pack(int k, const int nb[3], const int vid[3])
{
    const unsigned long long m19 = (1ull << SWE_CONN_DBITS) - 1ull;
    bool ok = vid[0] >= 0 && vid[0] < (1 << 26) && swe_conn_fits((long long)vid[1] - vid[0]) && swe_conn_fits((long long)vid[2] - vid[0]);
    unsigned long long lo = 0ull, hi = 0ull;
    if (ok) lo = (unsigned long long)vid[0] | (((unsigned long long)(long long)(vid[1] - vid[0]) & m19) << 26) | (((unsigned long long)(long long)(vid[2] - vid[0]) & m19) << 45);
    for (int f = 0; f < 3 && ok; f++) {
        unsigned long long field;
        if (nb[f] < 0) {
            ok = swe_conn_fits(-(long long)nb[f]);
            field = 3ull | (((unsigned long long)(-(long long)nb[f]) & m19) << 2);
        } else {
            const long long d = (long long)(nb[f] >> 2) - k;
            ok = swe_conn_fits(d) && (nb[f] & 3) != 3;
            field = (unsigned long long)(nb[f] & 3) | (((unsigned long long)d & m19) << 2);
        }
        hi |= field << (21*f);
    }
    if (!ok) { lo = 0ull; hi = 1ull << 63; }
    int4 c;
    c.x = (int)(unsigned)lo; c.y = (int)(unsigned)(lo >> 32); c.z = (int)(unsigned)hi; c.w = (int)(unsigned)(hi >> 32);
    return c;
}

// returns true for an escape record (nb, vid then hold nothing of use)
SWE_CONN_HD inline bool swe_conn_unpack(const int4 c, int k, int nb[3], int vid[3])
{
    const unsigned w0 = (unsigned)c.x, w1 = (unsigned)c.y, w2 = (unsigned)c.z, w3 = (unsigned)c.w;
    vid[0] = (int)(w0 & 0x3ffffffu);
    vid[1] = vid[0] + ((int)(((w0 >> 26) | (w1 << 6)) << 13) >> 13);
    vid[2] = vid[0] + ((int)w1 >> 13);
    const unsigned fld[3] = {w2, (w2 >> 21) | (w3 << 11), w3 >> 10};
    for (int f = 0; f < 3; f++) {
        const int val = (int)(fld[f] << 11) >> 13;       // bits 20..2 of the field, sign-extended
        const int code = (int)(fld[f] & 3u);
        nb[f] = code == 3 ? -val : (int)(((unsigned)(k + val) << 2) | (unsigned)code);
    }
    return c.w < 0;
}
