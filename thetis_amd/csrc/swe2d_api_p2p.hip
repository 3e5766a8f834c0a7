// swe2d_api_p2p.hip - halo lists, pack / unpack and the peer-to-peer landing zones (swe2d_p2p.h): ABI entry points
#include "swe2d_handle.h"

namespace swe2d_impl {

// byte offset of channel c's slot 0 in the landing zone of a rank with n_recv halo cells (both sides compute it)
size_t p2p_channel_offset(const int *width, int c, int n_recv)
{
    size_t off = SWE_P2P_HEADER_BYTES;
    for (int i = 0; i < c; i++) off += 2*(size_t)n_recv*width[i]*sizeof(double);
    return off;
}

}  // namespace swe2d_impl

extern "C" {

int swe2d_halo_setup(swe2d_handle *hh, int32_t n_send, const int32_t *send_cells, int32_t n_recv, const int32_t *recv_cells)
{
    Handle *h = H(hh);
    if (!h || n_send < 0 || n_recv < 0 || (n_send > 0 && !send_cells) || (n_recv > 0 && !recv_cells))
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad halo lists");
    for (int i = 0; i < n_send; i++)
        if (send_cells[i] < 0 || send_cells[i] >= h->n_owned)
            return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "send cell is not an owned cell");
    // (a handle WITHOUT ghost cells may copy between its own cells: chunks of one mesh that carry copies of their neighbours' rim,
    // tools/chunkbench.py)
    for (int i = 0; i < n_recv; i++)
        if (recv_cells[i] < (h->n_owned < h->n_cells ? h->n_owned : 0) || recv_cells[i] >= h->n_cells)
            return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "receive cell is not a ghost cell");
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->send_cells) { HIP_TRY(h, hipFree(h->send_cells)); h->send_cells = nullptr; }
    if (h->recv_cells) { HIP_TRY(h, hipFree(h->recv_cells)); h->recv_cells = nullptr; }
    h->n_send = n_send;
    h->n_recv = n_recv;
    h->h_send.assign(send_cells, send_cells + n_send);
    h->h_recv.assign(recv_cells, recv_cells + n_recv);
    h->flow_x_ready = false;
    if (n_send > 0) {
        HIP_TRY(h, hipMalloc(&h->send_cells, (size_t)n_send*sizeof(int)));
        HIP_TRY(h, hipMemcpy(h->send_cells, send_cells, (size_t)n_send*sizeof(int), hipMemcpyHostToDevice));
    }
    if (n_recv > 0) {
        HIP_TRY(h, hipMalloc(&h->recv_cells, (size_t)n_recv*sizeof(int)));
        HIP_TRY(h, hipMemcpy(h->recv_cells, recv_cells, (size_t)n_recv*sizeof(int), hipMemcpyHostToDevice));
    }
    return SWE2D_OK;
}

int swe2d_halo_pack(swe2d_handle *hh, int i_buffer, double *send_buf_dev)
{
    Handle *h = H(hh);
    if (!h || i_buffer < 0 || i_buffer > 2) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad buffer index");
    if (h->n_send == 0) return SWE2D_OK;
    if (!send_buf_dev) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null send buffer");
    HIP_TRY(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(swe_halo_pack, dim3(grid_for(3*h->npc*h->n_send)), dim3(256), 0, h->stream,
                       h->state[i_buffer], h->stride, h->send_cells, h->n_send, send_buf_dev, 3*h->npc);
    HIP_TRY(h, hipGetLastError());
    return SWE2D_OK;
}

int swe2d_halo_unpack(swe2d_handle *hh, int i_buffer, const double *recv_buf_dev)
{
    Handle *h = H(hh);
    if (!h || i_buffer < 0 || i_buffer > 2) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "bad buffer index");
    if (h->n_recv == 0) return SWE2D_OK;
    if (!recv_buf_dev) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "null recv buffer");
    HIP_TRY(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(swe_halo_unpack, dim3(grid_for(3*h->npc*h->n_recv)), dim3(256), 0, h->stream,
                       h->state[i_buffer], h->stride, h->recv_cells, h->n_recv, recv_buf_dev, 3*h->npc);
    HIP_TRY(h, hipGetLastError());
    return SWE2D_OK;
}

}  // extern "C"



// ------------------------------------------------------------------------------------------------------------------
// peer-to-peer halo exchange (swe2d_p2p.h)

extern "C" {

int swe2d_p2p_create(swe2d_handle *hh, int32_t n_channels, const int32_t *widths)
{
    Handle *h = H(hh);
    if (!h || n_channels < 1 || n_channels > SWE_P2P_MAX_CHANNELS || !widths)
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_create: 1..8 channels");
    if (h->p2p.zone) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_create: already created");
    HIP_TRY(h, hipSetDevice(h->device));
    auto &z = h->p2p;
    z.n_channels = n_channels;
    for (int c = 0; c < n_channels; c++) {
        if (widths[c] < 1) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_create: bad channel width");
        z.width[c] = widths[c];
    }
    z.zone_bytes = p2p_channel_offset(z.width, n_channels, h->n_recv);
    z.zone_bytes = (z.zone_bytes + 4095)/4096*4096;
    // remote GPUs write here and local kernels poll it: keep it out of the (non-coherent) L2 when the runtime allows
    const int want = h->opt[SWE2D_OPT_P2P_ZONE] > 0 ? h->opt[SWE2D_OPT_P2P_ZONE] : 0;      // 1 uncached | 2 fine-grained | 3 device (debugging)
    z.zone_kind = 0;
    if ((want == 0 || want == 1) && hipExtMallocWithFlags(&z.zone, z.zone_bytes, hipDeviceMallocUncached) == hipSuccess) z.zone_kind = 1;
    if (!z.zone_kind) (void)hipGetLastError();
    if (!z.zone_kind && (want == 0 || want == 2)
        && hipExtMallocWithFlags(&z.zone, z.zone_bytes, hipDeviceMallocFinegrained) == hipSuccess) z.zone_kind = 2;
    if (!z.zone_kind) {
        (void)hipGetLastError();
        HIP_TRY(h, hipMalloc(&z.zone, z.zone_bytes));
        z.zone_kind = 3;
    }
    HIP_TRY(h, hipMemset(z.zone, 0, z.zone_bytes));
    HIP_TRY(h, hipMalloc(&z.ctr, n_channels*sizeof(SweP2pCounters)));
    HIP_TRY(h, hipMemset(z.ctr, 0, n_channels*sizeof(SweP2pCounters)));
    HIP_TRY(h, hipDeviceSynchronize());
    return SWE2D_OK;
}

int swe2d_p2p_export(swe2d_handle *hh, void *ipc_handle_out, void **local_base, int32_t *zone_kind)
{
    Handle *h = H(hh);
    if (!h || !h->p2p.zone) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_export: no landing zone");
    static_assert(sizeof(hipIpcMemHandle_t) == SWE2D_IPC_HANDLE_BYTES, "IPC handle size");
    HIP_TRY(h, hipSetDevice(h->device));
    if (ipc_handle_out) {
        hipIpcMemHandle_t mh;
        HIP_TRY(h, hipIpcGetMemHandle(&mh, h->p2p.zone));
        std::memcpy(ipc_handle_out, &mh, sizeof(mh));
    }
    if (local_base) *local_base = h->p2p.zone;
    if (zone_kind) *zone_kind = h->p2p.zone_kind;
    return SWE2D_OK;
}

int swe2d_p2p_open(swe2d_handle *hh, const void *ipc_handle, void **remote_base)
{
    Handle *h = H(hh);
    if (!h || !ipc_handle || !remote_base) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_open: null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    hipIpcMemHandle_t mh;
    std::memcpy(&mh, ipc_handle, sizeof(mh));
    void *p = nullptr;
    HIP_TRY(h, hipIpcOpenMemHandle(&p, mh, hipIpcMemLazyEnablePeerAccess));
    h->p2p.opened.push_back(p);
    // First contact with a peer's memory must fail with an error code, never with a memory fault inside the push kernel (which
    // would take the process down): the mapping has to be a device pointer of this process, a host-initiated copy into the
    // unused upper half of the zone header has to round-trip, and so has a store + load from a kernel of this device.
    hipPointerAttribute_t attr;
    HIP_TRY(h, hipPointerGetAttributes(&attr, p));
    unsigned long long *probe = reinterpret_cast<unsigned long long *>(static_cast<char *>(p) + SWE_P2P_HEADER_BYTES/2)
                                + (unsigned)getpid() % (SWE_P2P_HEADER_BYTES/16);
    const unsigned long long pattern = 0x5157453244503250ull ^ ((unsigned long long)getpid() << 20) ^ (unsigned long long)h->device;
    unsigned long long back = 0;
    HIP_TRY(h, hipMemcpy(probe, &pattern, sizeof(pattern), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(&back, probe, sizeof(back), hipMemcpyDeviceToHost));
    if (back != pattern) return fail(h, SWE2D_ERR_HIP, "swe2d_p2p_open: a copy into the peer's landing zone does not read back");
    unsigned long long *dback = nullptr;
    HIP_TRY(h, hipMalloc(&dback, sizeof(*dback)));
    hipLaunchKernelGGL(swe_p2p_probe_kernel, dim3(1), dim3(64), 0, h->stream, probe, ~pattern, dback);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) e = hipMemcpy(&back, dback, sizeof(back), hipMemcpyDeviceToHost);
    (void)hipFree(dback);
    if (e != hipSuccess) return fail(h, SWE2D_ERR_HIP, "swe2d_p2p_open: the probe kernel failed on the peer's landing zone");
    if (back != ~pattern) return fail(h, SWE2D_ERR_HIP, "swe2d_p2p_open: a kernel store into the peer's landing zone does not read back");
    *remote_base = p;
    return SWE2D_OK;
}

int swe2d_p2p_connect(swe2d_handle *hh, int32_t n_peers, void *const *remote_base, const int32_t *send_offset,
                      const int32_t *send_count, const int32_t *remote_recv_offset, const int32_t *remote_flag_index,
                      const int32_t *remote_n_recv, int32_t n_from)
{
    Handle *h = H(hh);
    if (!h || !h->p2p.zone) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_connect: no landing zone");
    if (n_peers < 0 || n_peers > SWE_P2P_MAX_PEERS || n_from < 0 || n_from > SWE_P2P_MAX_PEERS)
        return fail(h, SWE2D_ERR_UNSUPPORTED, "swe2d_p2p_connect: at most 8 peers per rank");
    auto &z = h->p2p;
    int end = 0;
    for (int i = 0; i < n_peers; i++) {
        if (!remote_base[i] || send_offset[i] != end || send_count[i] < 0 || remote_flag_index[i] < 0
            || remote_flag_index[i] >= SWE_P2P_MAX_PEERS || remote_recv_offset[i] < 0
            || remote_recv_offset[i] + send_count[i] > remote_n_recv[i])
            return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_connect: segments must tile the send list in order");
        end += send_count[i];
        z.remote_base[i] = static_cast<char *>(remote_base[i]);
        z.off[i] = send_offset[i]; z.cnt[i] = send_count[i];
        z.remote_off[i] = remote_recv_offset[i]; z.remote_flag[i] = remote_flag_index[i]; z.remote_n_recv[i] = remote_n_recv[i];
    }
    if (end != h->n_send) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_connect: segments do not cover the send list");
    z.n_peers = n_peers;
    z.n_from = n_from;
    return SWE2D_OK;
}

namespace {
int p2p_field(Handle *h, int channel, int i_buffer, double **planes, int *np)
{
    if (channel < 0 || channel >= h->p2p.n_channels || i_buffer < 0 || i_buffer > 2)
        return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "p2p: bad channel or buffer index");
    if (channel == 0) { *planes = h->state[i_buffer]; *np = 3*h->npc; }
    else {
        if (channel - 1 >= (int)h->tracers.size()) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "p2p: channel has no tracer");
        *planes = h->tracers[channel - 1].buf[i_buffer]; *np = h->npc;
    }
    if (*np != h->p2p.width[channel]) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "p2p: channel width mismatch");
    return SWE2D_OK;
}
}  // namespace

namespace {
int p2p_push_args(Handle *h, int channel, int i_buffer, SweP2pPushArgs &a)
{
    double *planes; int np;
    if (int rc = p2p_field(h, channel, i_buffer, &planes, &np)) return rc;
    auto &z = h->p2p;
    a = SweP2pPushArgs{};
    a.planes = planes; a.stride = h->stride; a.send_cells = h->send_cells; a.n_send = h->n_send; a.np = np;
    a.n_peers = z.n_peers;
    for (int i = 0; i < z.n_peers; i++) {
        a.off[i] = z.off[i]; a.cnt[i] = z.cnt[i];
        char *base = z.remote_base[i];
        a.rdata[i] = reinterpret_cast<double *>(base + p2p_channel_offset(z.width, channel, z.remote_n_recv[i]))
                     + (size_t)z.remote_off[i]*np;
        a.rslot[i] = (size_t)z.remote_n_recv[i]*np;
        a.rflag[i] = reinterpret_cast<unsigned long long *>(base)
                     + (size_t)(channel*SWE_P2P_MAX_PEERS + z.remote_flag[i])*SWE_P2P_FLAG_STRIDE;
    }
    a.ctr = z.ctr + channel;
    return SWE2D_OK;
}

int p2p_unpack_args(Handle *h, int channel, int i_buffer, SweP2pUnpackArgs &a)
{
    double *planes; int np;
    if (int rc = p2p_field(h, channel, i_buffer, &planes, &np)) return rc;
    auto &z = h->p2p;
    a = SweP2pUnpackArgs{};
    a.planes = planes; a.stride = h->stride; a.recv_cells = h->recv_cells; a.n_recv = h->n_recv; a.np = np;
    a.n_from = z.n_from;
    char *base = static_cast<char *>(z.zone);
    for (int i = 0; i < z.n_from; i++)
        a.flag[i] = reinterpret_cast<const unsigned long long *>(base) + (size_t)(channel*SWE_P2P_MAX_PEERS + i)*SWE_P2P_FLAG_STRIDE;
    a.zone = reinterpret_cast<const double *>(base + p2p_channel_offset(z.width, channel, h->n_recv));
    a.slot = (size_t)h->n_recv*np;
    a.timeout_ticks = (unsigned long long)(opt_seconds(h, SWE2D_OPT_P2P_TIMEOUT_MS, 5.0)*1e8);
    a.ctr = z.ctr + channel;
    a.fence = z.zone_kind == 3;
    return SWE2D_OK;
}
}  // namespace

int swe2d_p2p_push(swe2d_handle *hh, int channel, int i_buffer)
{
    Handle *h = H(hh);
    if (!h || !h->p2p.zone) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_push: not connected");
    SweP2pPushArgs a;
    if (int rc = p2p_push_args(h, channel, i_buffer, a)) return rc;
    if (h->p2p.n_peers == 0 || h->n_send == 0) return SWE2D_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(swe_p2p_push_kernel, dim3(std::min(SWE_P2P_MAX_BLOCKS, grid_for(a.np*h->n_send))), dim3(256), 0, h->xstream ? h->xstream : h->stream, a);
    HIP_TRY(h, hipGetLastError());
    return SWE2D_OK;
}

// the same for several channels in ONE launch each way (swe_p2p_push_multi_kernel: the single-channel kernels side by side)
int swe2d_p2p_push_multi(swe2d_handle *hh, int n, const int32_t *channels, const int32_t *i_buffers)
{
    Handle *h = H(hh);
    if (!h || !h->p2p.zone) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_push_multi: not connected");
    if (n < 1 || n > SWE_P2P_MULTI || !channels || !i_buffers) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_push_multi: 1 .. 4 channels");
    SweP2pPushMulti m;
    int np_max = 0;
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < i; j++) if (channels[j] == channels[i]) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_push_multi: a channel twice");
        if (int rc = p2p_push_args(h, channels[i], i_buffers[i], m.a[i])) return rc;
        np_max = std::max(np_max, m.a[i].np);
    }
    if (h->p2p.n_peers == 0 || h->n_send == 0) return SWE2D_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(swe_p2p_push_multi_kernel, dim3(std::min(SWE_P2P_MAX_BLOCKS, grid_for(np_max*h->n_send)), n), dim3(256), 0, h->xstream ? h->xstream : h->stream, m);
    HIP_TRY(h, hipGetLastError());
    return SWE2D_OK;
}

int swe2d_p2p_wait_unpack_multi(swe2d_handle *hh, int n, const int32_t *channels, const int32_t *i_buffers)
{
    Handle *h = H(hh);
    if (!h || !h->p2p.zone) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_wait_unpack_multi: not connected");
    if (n < 1 || n > SWE_P2P_MULTI || !channels || !i_buffers) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_wait_unpack_multi: 1 .. 4 channels");
    SweP2pUnpackMulti m;
    int np_max = 0;
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < i; j++) if (channels[j] == channels[i]) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_wait_unpack_multi: a channel twice");
        if (int rc = p2p_unpack_args(h, channels[i], i_buffers[i], m.a[i])) return rc;
        np_max = std::max(np_max, m.a[i].np);
    }
    if (h->p2p.n_from == 0 || h->n_recv == 0) return SWE2D_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(swe_p2p_unpack_multi_kernel, dim3(std::min(SWE_P2P_MAX_BLOCKS, grid_for(np_max*h->n_recv)), n), dim3(256), 0, h->xstream ? h->xstream : h->stream, m);
    HIP_TRY(h, hipGetLastError());
    return SWE2D_OK;
}

int swe2d_p2p_wait_unpack(swe2d_handle *hh, int channel, int i_buffer)
{
    Handle *h = H(hh);
    if (!h || !h->p2p.zone) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_wait_unpack: not connected");
    SweP2pUnpackArgs a;
    if (int rc = p2p_unpack_args(h, channel, i_buffer, a)) return rc;
    if (h->p2p.n_from == 0 || h->n_recv == 0) return SWE2D_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(swe_p2p_unpack_kernel, dim3(std::min(SWE_P2P_MAX_BLOCKS, grid_for(a.np*h->n_recv))), dim3(256), 0, h->xstream ? h->xstream : h->stream, a);
    HIP_TRY(h, hipGetLastError());
    return SWE2D_OK;
}

// The exchange kernels (swe2d_p2p_push / swe2d_p2p_wait_unpack: a few thousand cells, 5-6 us each, mostly latency) on a stream of
// their own: the caller orders it against the handle's stream with events (push after the send cells' stage, the next reader of the
// ghost cells after the unpack) and the stage kernels of the interior run meanwhile.  Null: back on the handle's stream.  No
// synchronisation here (usable around a stream capture that forks into this stream and joins again).
int swe2d_set_exchange_stream(swe2d_handle *hh, void *hip_stream)
{
    Handle *h = H(hh);
    if (!h) return SWE2D_ERR_INVALID_ARGUMENT;
    h->xstream = reinterpret_cast<hipStream_t>(hip_stream);
    return SWE2D_OK;
}

int swe2d_p2p_status(swe2d_handle *hh, int64_t *epochs_sent, int64_t *epochs_received, int32_t *timeouts)
{
    Handle *h = H(hh);
    if (!h || !h->p2p.ctr) return fail(h, SWE2D_ERR_INVALID_ARGUMENT, "swe2d_p2p_status: not created");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->xstream) HIP_TRY(h, hipStreamSynchronize(h->xstream));
    std::vector<SweP2pCounters> c(h->p2p.n_channels);
    HIP_TRY(h, hipMemcpy(c.data(), h->p2p.ctr, c.size()*sizeof(SweP2pCounters), hipMemcpyDeviceToHost));
    int to = 0;
    for (int i = 0; i < h->p2p.n_channels; i++) {
        if (epochs_sent) epochs_sent[i] = (int64_t)c[i].epoch_send;
        if (epochs_received) epochs_received[i] = (int64_t)c[i].epoch_recv;
        to += (int)c[i].timeouts;
    }
    if (timeouts) *timeouts = to;
    return SWE2D_OK;
}

}  // extern "C"
