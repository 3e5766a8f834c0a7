// swe2d_p2p.h - peer-to-peer halo exchange through IPC-mapped device memory (gfx950, xGMI or same-device peers).
//
// Replaces the per-par_loop halo exchange of PyOP2 under mpiexec [FD-assumed] (examples/README.md:51-56) for one process per
// GPU.  No host call and no RCCL call sits in the step loop: a rank PUSHES the state of its send cells straight into the
// landing zone of every peer (stores over xGMI) and raises a per-peer epoch flag there; the receiver's WAIT+UNPACK kernel
// spins on its own flags (local, uncached memory) and scatters the landed cells into its ghost cells.  Both are ordinary
// kernels on the handle's stream, so a whole exchange cycle (stage kernels, push, overlapped interior work, wait+unpack) is
// ONE HIP graph.
//
// Landing zone of a rank (one allocation, hipDeviceMallocUncached when available, exported with hipIpcGetMemHandle):
//   [ header: 64-bit epoch flags, one per (channel, sending peer), 64 B apart ]
//   [ channel 0: slot 0 | slot 1 ]  [ channel 1: slot 0 | slot 1 ] ...      slot = [n_recv][width] doubles, in recv-list order
// A channel is one exchanged field set (0: the SWE state, width 3k; 1..: tracers, width k).  Epochs count exchanges per
// channel; exchange e lands in slot e & 1.  Two slots suffice: a peer can push exchange e+1 only after it has received my
// exchange e, which I push after having unpacked exchange e-1 (stream order) - the slot it overwrites is already consumed.
// The epoch counters live in device memory and are advanced by the kernels themselves, so a captured graph replays correctly.
//
// Memory ordering (MI355X_MICROARCH.md, inter-workgroup visibility): payload is written with system-scope (sc0 sc1,
// write-through) stores; every workgroup drains them (s_waitcnt vmcnt(0)) and takes a ticket; the last workgroup to arrive
// stores the flags (system scope).  The consumer polls with system-scope relaxed loads and reads the payload with
// system-scope loads from the uncached zone (no cache holds a line of it).  One workgroup per CU at most: per-workgroup
// fences (buffer_wbl2 / buffer_inv) cost microseconds each and are not needed for write-through payload.
// Every spin is bounded (wall clock); a timeout is recorded in the status word, the kernel carries on and later waits of the
// channel do not spin at all, so a lost peer costs one timeout and can never hang the GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SWE_P2P_MAX_PEERS 8
#define SWE_P2P_MAX_CHANNELS 8
#define SWE_P2P_HEADER_BYTES 8192          // SWE_P2P_MAX_CHANNELS * SWE_P2P_MAX_PEERS flags, 64 B apart, rounded up
#define SWE_P2P_FLAG_STRIDE 8              // in 8-byte words

// per-channel device counters (ordinary device memory of the owning rank)
struct SweP2pCounters {
    unsigned long long epoch_send, epoch_recv;      // completed pushes / unpacks
    unsigned int ticket_send, ticket_recv;
    unsigned int timeouts, pad;
};

struct SweP2pPushArgs {
    const double *planes;          // np planes of the field to send
    size_t stride;
    const int *send_cells;
    int n_send, np;
    int n_peers;
    int off[SWE_P2P_MAX_PEERS], cnt[SWE_P2P_MAX_PEERS];      // per peer: segment of the send list (cells)
    double *rdata[SWE_P2P_MAX_PEERS];                        // peer's landing segment for me, slot 0
    size_t rslot[SWE_P2P_MAX_PEERS];                         // doubles between the peer's slot 0 and slot 1
    unsigned long long *rflag[SWE_P2P_MAX_PEERS];            // my flag in the peer's header
    SweP2pCounters *ctr;
};

struct SweP2pUnpackArgs {
    double *planes;
    size_t stride;
    const int *recv_cells;
    int n_recv, np;
    int n_from;                                              // number of peers that send to me
    const unsigned long long *flag[SWE_P2P_MAX_PEERS];       // their flags in MY header
    const double *zone;                                      // my landing data of this channel, slot 0
    size_t slot;                                             // doubles between slot 0 and slot 1
    unsigned long long timeout_ticks;                        // wall_clock64 ticks (100 MHz)
    int fence;                                               // zone is ordinary device memory: acquire fence after the wait
    SweP2pCounters *ctr;
};

__device__ __forceinline__ void swe_p2p_store(double *p, double x)
{
    __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);          // global_store_dwordx2 sc0 sc1
}
__device__ __forceinline__ double swe_p2p_load(const double *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);       // global_load_dwordx2 sc0 sc1
}

// first-contact probe of a freshly mapped peer zone (swe2d_p2p_open): system-scope store + load of one word
static __global__ void swe_p2p_probe_kernel(unsigned long long *word, unsigned long long pattern, unsigned long long *out)
{
    if (threadIdx.x == 0) {
        __hip_atomic_store(word, pattern, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *out = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

#define SWE_P2P_MAX_BLOCKS 256            // one 256-lane workgroup per CU, grid-stride over the message

__device__ __forceinline__ int swe_p2p_peer_of(const SweP2pPushArgs &a, int j)
{
    int p = 0;
#pragma unroll 1
    for (int i = 1; i < a.n_peers; i++) if (j >= a.off[i]) p = i;                    // segments are sorted by offset
    return p;
}

__device__ __forceinline__ void swe_p2p_push_body(const SweP2pPushArgs &a)
{
    const unsigned long long target = a.ctr->epoch_send + 1ull;      // every workgroup reads it before the last one advances it
    const int total = a.np*a.n_send, step = gridDim.x*256;
    // four elements per lane and trip: the gathers from the planes are all in flight before the first remote store
    for (int t0 = blockIdx.x*256 + threadIdx.x; t0 < total; t0 += 4*step) {
        double x[4];
        int j[4], q[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int t = t0 + r*step;
            j[r] = t/a.np; q[r] = t - a.np*j[r];
            x[r] = t < total ? a.planes[(size_t)q[r]*a.stride + a.send_cells[j[r]]] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (t0 + r*step >= total) continue;
            const int p = swe_p2p_peer_of(a, j[r]);
            swe_p2p_store(a.rdata[p] + (target & 1ull)*a.rslot[p] + (size_t)(j[r] - a.off[p])*a.np + q[r], x[r]);
        }
    }
    // the payload stores are write-through (sc0 sc1): nothing of them stays dirty in the L2, draining them is the release
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int ticket = __hip_atomic_fetch_add(&a.ctr->ticket_send, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == gridDim.x - 1) {                                               // every workgroup has drained its stores
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");                            // belt and braces, once per exchange
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int i = 0; i < a.n_peers; i++)
                __hip_atomic_store(a.rflag[i], target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            a.ctr->ticket_send = 0u;
            a.ctr->epoch_send = target;
        }
    }
}

static __global__ __launch_bounds__(256) void swe_p2p_push_kernel(const SweP2pPushArgs a) { swe_p2p_push_body(a); }

// Several channels (the shallow water state and the tracers of a coupled cycle) in ONE launch: blockIdx.y is the channel, every
// channel keeps its own epoch counters, tickets and flags - exactly the single-channel kernels side by side, two kernel boundaries
// per channel less (6-7 us each on a rank of eight: profiles/r06c_cfg4_rank8_kernel_stats.csv)
#define SWE_P2P_MULTI 4
struct SweP2pPushMulti { SweP2pPushArgs a[SWE_P2P_MULTI]; };
static __global__ __launch_bounds__(256) void swe_p2p_push_multi_kernel(const SweP2pPushMulti m) { swe_p2p_push_body(m.a[blockIdx.y]); }

__device__ __forceinline__ void swe_p2p_unpack_body(const SweP2pUnpackArgs &a)
{
    const unsigned long long target = a.ctr->epoch_recv + 1ull;
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        bool late = a.ctr->timeouts != 0u;              // sticky: after one timeout (a lost peer) no later wait spins again
        for (int i = 0; i < a.n_from && !late; i++) {
            while (__hip_atomic_load(a.flag[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < target) {
                if (wall_clock64() - t0 > a.timeout_ticks) { late = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (late && blockIdx.x == 0) atomicAdd(&a.ctr->timeouts, 1u);
        // an uncached / fine-grained zone is read past the caches by the system-scope loads below; an ordinary allocation
        // (fallback) needs the caches invalidated first
        if (a.fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    __syncthreads();
    asm volatile("" ::: "memory");
    const int total = a.np*a.n_recv, step = gridDim.x*256;
    const double *src = a.zone + (target & 1ull)*a.slot;
    for (int t0 = blockIdx.x*256 + threadIdx.x; t0 < total; t0 += 4*step) {
        double x[4];
#pragma unroll
        for (int r = 0; r < 4; r++) x[r] = (t0 + r*step < total) ? swe_p2p_load(src + t0 + r*step) : 0.0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int t = t0 + r*step;
            if (t >= total) continue;
            const int j = t/a.np, q = t - a.np*j;
            a.planes[(size_t)q*a.stride + a.recv_cells[j]] = x[r];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int ticket = __hip_atomic_fetch_add(&a.ctr->ticket_recv, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == gridDim.x - 1) {
            a.ctr->ticket_recv = 0u;
            a.ctr->epoch_recv = target;
        }
    }
}

static __global__ __launch_bounds__(256) void swe_p2p_unpack_kernel(const SweP2pUnpackArgs a) { swe_p2p_unpack_body(a); }
struct SweP2pUnpackMulti { SweP2pUnpackArgs a[SWE_P2P_MULTI]; };
static __global__ __launch_bounds__(256) void swe_p2p_unpack_multi_kernel(const SweP2pUnpackMulti m) { swe_p2p_unpack_body(m.a[blockIdx.y]); }
