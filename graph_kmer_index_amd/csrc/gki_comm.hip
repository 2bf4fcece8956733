// Multi-GPU exchange: all-gather(v) of the FlatKmers columns over RCCL (xGMI inside a node).
//
// The enumeration shards by critical-path ranges and needs no exchange; the one collective of the
// build is gathering every rank's finished columns before the table build.  Shards
// have different sizes, so the gather is a group of point-to-point transfers: every rank sends its
// shard to each peer and receives each peer's shard at its offset.  On MI355X's fully connected xGMI
// every pair has its own link, so all 7 sends of a rank run in parallel -- a ring schedule would
// move the same bytes 7 hops over one link each.
//
// librccl.so is loaded lazily (dlopen) so that single-GPU users never touch it.
#include "gki_common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <time.h>

struct gki_comm {
    ncclComm_t comm;
    int world, rank;
    hipStream_t stream;
};

namespace {
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
    if (g_rccl.h) return GKI_OK;
    void *h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return gki_set_error(GKI_ERR_HIP, "cannot load librccl.so: %s", dlerror());
#define SYM(field, name)                                                                     \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));                 \
    if (!g_rccl.field) return gki_set_error(GKI_ERR_HIP, "librccl.so lacks %s", name);
    SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
    SYM(Send, "ncclSend") SYM(Recv, "ncclRecv") SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd")
    SYM(GetErrorString, "ncclGetErrorString") SYM(AllReduce, "ncclAllReduce")
    SYM(CommCount, "ncclCommCount") SYM(CommUserRank, "ncclCommUserRank")
#undef SYM
    g_rccl.h = h;
    return GKI_OK;
}

// The exchanges have never run between more than one real GPU before a user's (or the driver's) first multi-GPU run: rank 0
// says on stderr what it posted and when it completed, so that a stall names its phase.
void comm_trace(const gki_comm *c, const char *phase, const char *what, int sends, int recvs, double bytes, double ms) {
    if (c->rank != 0 || c->world < 2) return;
    if (ms < 0) fprintf(stderr, "[gki comm] %s: %s (%d sends, %d receives, %.3f GB out of rank 0, world %d)\n", phase, what, sends, recvs, bytes / 1e9, c->world);
    else fprintf(stderr, "[gki comm] %s: %s after %.1f ms\n", phase, what, ms);
    fflush(stderr);
}
double now_ms() {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return 1e3 * (double)t.tv_sec + 1e-6 * (double)t.tv_nsec;
}

#define NCCL_TRY(call)                                                                                      \
    do {                                                                                                    \
        ncclResult_t r_ = (call);                                                                           \
        if (r_ != ncclSuccess) return gki_set_error(GKI_ERR_HIP, "%s -> %s", #call, g_rccl.GetErrorString(r_)); \
    } while (0)
}  // namespace

extern "C" {

int gki_comm_get_unique_id(void *h_id) {
    GKI_TRY(load_rccl());
    static_assert(sizeof(ncclUniqueId) == GKI_COMM_ID_BYTES, "ncclUniqueId size");
    NCCL_TRY(g_rccl.GetUniqueId(reinterpret_cast<ncclUniqueId *>(h_id)));
    return GKI_OK;
}

int gki_comm_create(gki_comm **out, int world_size, int rank, const void *h_id) {
    *out = nullptr;
    if (world_size < 1 || rank < 0 || rank >= world_size) return gki_set_error(GKI_ERR_BAD_ARG, "bad rank %d of %d", rank, world_size);
    GKI_TRY(load_rccl());
    ncclUniqueId id;
    memcpy(&id, h_id, sizeof(id));
    hipStream_t stream = nullptr;
    HIP_TRY(hipStreamCreate(&stream));
    ncclComm_t comm = nullptr;
    const ncclResult_t r = g_rccl.CommInitRank(&comm, world_size, id, rank);
    if (r != ncclSuccess) {                              // nothing of a half-built communicator stays behind
        (void)hipStreamDestroy(stream);
        return gki_set_error(GKI_ERR_HIP, "ncclCommInitRank(world %d, rank %d) -> %s", world_size, rank,
                             g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error");
    }
    gki_comm *c = new gki_comm();
    c->world = world_size; c->rank = rank; c->stream = stream; c->comm = comm;
    *out = c;
    return GKI_OK;
}

int gki_comm_info(gki_comm *c, int *rccl_world, int *rccl_rank) {
    *rccl_world = 0; *rccl_rank = -1;
    if (!c) return gki_set_error(GKI_ERR_BAD_ARG, "no communicator");
    NCCL_TRY(g_rccl.CommCount(c->comm, rccl_world));
    NCCL_TRY(g_rccl.CommUserRank(c->comm, rccl_rank));
    return GKI_OK;
}

int gki_comm_destroy(gki_comm *c) {
    if (!c) return GKI_OK;
    (void)hipStreamSynchronize(c->stream);
    if (g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return GKI_OK;
}

int gki_comm_allgather_flat(gki_comm *c, const int64_t *h_counts, const void *d_hashes, const void *d_nodes,
                            const void *d_ref_offsets, const void *d_af32, void *d_out_hashes, void *d_out_nodes,
                            void *d_out_ref_offsets, void *d_out_af32) {
    const int W = c->world, me = c->rank;
    int64_t off = 0, my_off = 0;
    for (int r = 0; r < W; r++) { if (h_counts[r] < 0) return gki_set_error(GKI_ERR_BAD_ARG, "negative count"); if (r == me) my_off = off; off += h_counts[r]; }
    const void *in[4] = {d_hashes, d_nodes, d_ref_offsets, d_af32};
    void *outp[4] = {d_out_hashes, d_out_nodes, d_out_ref_offsets, d_out_af32};
    const size_t esz[4] = {8, 4, 8, 4};
    hipStream_t s = c->stream;
    HIP_TRY(hipDeviceSynchronize());                  // the columns were produced on other streams (finder, partition)
    // own shard: device-to-device copy
    for (int col = 0; col < 4; col++)
        if (h_counts[me] > 0)
            HIP_TRY(hipMemcpyAsync((char *)outp[col] + (size_t)my_off * esz[col], in[col], (size_t)h_counts[me] * esz[col],
                                   hipMemcpyDeviceToDevice, s));
    const double t0 = now_ms();
    if (W > 1) {
        int n_send = 0, n_recv = 0;
        double bytes = 0;
        NCCL_TRY(g_rccl.GroupStart());
        for (int col = 0; col < 4; col++) {
            int64_t o = 0;
            for (int r = 0; r < W; r++) {
                if (r != me) {
                    if (h_counts[me] > 0) {
                        NCCL_TRY(g_rccl.Send(in[col], (size_t)h_counts[me] * esz[col], ncclUint8, r, c->comm, s));
                        n_send++; bytes += (double)h_counts[me] * esz[col];
                    }
                    if (h_counts[r] > 0) {
                        NCCL_TRY(g_rccl.Recv((char *)outp[col] + (size_t)o * esz[col], (size_t)h_counts[r] * esz[col], ncclUint8, r,
                                             c->comm, s));
                        n_recv++;
                    }
                }
                o += h_counts[r];
            }
        }
        comm_trace(c, "all-gather", "group posted, closing it", n_send, n_recv, bytes, -1);
        NCCL_TRY(g_rccl.GroupEnd());
        comm_trace(c, "all-gather", "group closed, waiting for the stream", n_send, n_recv, bytes, -1);
    }
    HIP_TRY(hipStreamSynchronize(s));
    comm_trace(c, "all-gather", "completed", 0, 0, 0, now_ms() - t0);
    return GKI_OK;
}

int gki_comm_alltoall_flat(gki_comm *c, const int64_t *h_send_start, const void *d_hashes, const void *d_nodes,
                           const void *d_ref_offsets, const void *d_af32, const int64_t *h_recv_start, void *d_out_hashes,
                           void *d_out_nodes, void *d_out_ref_offsets, void *d_out_af32) {
    const int W = c->world, me = c->rank;
    for (int r = 0; r < W; r++)
        if (h_send_start[r + 1] < h_send_start[r] || h_recv_start[r + 1] < h_recv_start[r])
            return gki_set_error(GKI_ERR_BAD_ARG, "alltoall: slice tables must be non-decreasing");
    if (h_send_start[me + 1] - h_send_start[me] != h_recv_start[me + 1] - h_recv_start[me])
        return gki_set_error(GKI_ERR_BAD_ARG, "alltoall: own slice differs between the send and receive tables");
    const void *in[4] = {d_hashes, d_nodes, d_ref_offsets, d_af32};
    void *outp[4] = {d_out_hashes, d_out_nodes, d_out_ref_offsets, d_out_af32};
    const size_t esz[4] = {8, 4, 8, 4};
    hipStream_t s = c->stream;
    HIP_TRY(hipDeviceSynchronize());                  // the columns were produced on other streams (finder, partition)
    const int64_t own = h_send_start[me + 1] - h_send_start[me];
    for (int col = 0; col < 4; col++)
        if (own > 0)
            HIP_TRY(hipMemcpyAsync((char *)outp[col] + (size_t)h_recv_start[me] * esz[col],
                                   (const char *)in[col] + (size_t)h_send_start[me] * esz[col], (size_t)own * esz[col],
                                   hipMemcpyDeviceToDevice, s));
    const double t0 = now_ms();
    if (W > 1) {
        // one point-to-point pair per peer and column; xGMI is fully connected, so all pairs move at once
        int n_send = 0, n_recv = 0;
        double bytes = 0;
        NCCL_TRY(g_rccl.GroupStart());
        for (int col = 0; col < 4; col++)
            for (int r = 0; r < W; r++) {
                if (r == me) continue;
                const int64_t ns = h_send_start[r + 1] - h_send_start[r], nr = h_recv_start[r + 1] - h_recv_start[r];
                if (ns > 0) {
                    NCCL_TRY(g_rccl.Send((const char *)in[col] + (size_t)h_send_start[r] * esz[col], (size_t)ns * esz[col], ncclUint8,
                                         r, c->comm, s));
                    n_send++; bytes += (double)ns * esz[col];
                }
                if (nr > 0) {
                    NCCL_TRY(g_rccl.Recv((char *)outp[col] + (size_t)h_recv_start[r] * esz[col], (size_t)nr * esz[col], ncclUint8, r,
                                         c->comm, s));
                    n_recv++;
                }
            }
        comm_trace(c, "all-to-all", "group posted, closing it", n_send, n_recv, bytes, -1);
        NCCL_TRY(g_rccl.GroupEnd());
        comm_trace(c, "all-to-all", "group closed, waiting for the stream", n_send, n_recv, bytes, -1);
    }
    HIP_TRY(hipStreamSynchronize(s));
    comm_trace(c, "all-to-all", "completed", 0, 0, 0, now_ms() - t0);
    return GKI_OK;
}

int gki_comm_allreduce_u32(gki_comm *c, void *d_buf, int64_t n) {
    if (n <= 0) return GKI_OK;
    HIP_TRY(hipDeviceSynchronize());                  // the counts were produced on other streams
    if (c->world > 1) NCCL_TRY(g_rccl.AllReduce(d_buf, d_buf, (size_t)n, ncclUint32, ncclSum, c->comm, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GKI_OK;
}

// ---- ranks that share ONE device (a multi-rank run rehearsed on a single GPU): RCCL refuses a communicator with two
// ranks on the same device ("invalid usage", duplicate GPU), so such ranks exchange through HIP IPC handles instead --
// every rank exports its buffers, the peers open them and copy device-to-device.  The handles travel over the
// control plane (parallel.SharedDeviceComm).
int gki_device_bus_id(char *buf, int len) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetPCIBusId(buf, len, dev));
    return GKI_OK;
}

int gki_ipc_export(const void *d_ptr, void *h_handle, int64_t *offset) {
    static_assert(sizeof(hipIpcMemHandle_t) == GKI_IPC_HANDLE_BYTES, "hipIpcMemHandle_t size");
    void *base = nullptr;
    size_t size = 0;
    HIP_TRY(hipMemGetAddressRange((hipDeviceptr_t *)&base, &size, (hipDeviceptr_t)d_ptr));
    hipIpcMemHandle_t h;
    HIP_TRY(hipIpcGetMemHandle(&h, base));
    memcpy(h_handle, &h, sizeof(h));
    *offset = (int64_t)((const char *)d_ptr - (const char *)base);
    return GKI_OK;
}

int gki_ipc_open(const void *h_handle, void **d_base) {
    hipIpcMemHandle_t h;
    memcpy(&h, h_handle, sizeof(h));
    *d_base = nullptr;
    HIP_TRY(hipIpcOpenMemHandle(d_base, h, hipIpcMemLazyEnablePeerAccess));
    return GKI_OK;
}

int gki_ipc_close(void *d_base) {
    if (d_base) HIP_TRY(hipIpcCloseMemHandle(d_base));
    return GKI_OK;
}

}  // extern "C"
