// gather.cpp -- libvolrend_gather.so: the RGBA8 tile gather of the screen-tile shard over RCCL
// (include/volrend_gather.h).  The only translation unit of the product that talks to RCCL.
#include "volrend_gather.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

struct VrGatherOpaque {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

namespace {
thread_local char g_err[512] = "";
int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}
#define NCCL_TRY(expr)                                                                          \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess) return fail("%s: %s", #expr, ncclGetErrorString(r_));            \
    } while (0)
static_assert(sizeof(ncclUniqueId) == VR_GATHER_ID_BYTES, "VR_GATHER_ID_BYTES");
}  // namespace

extern "C" {

const char* vr_gather_last_error(void) { return g_err; }

int vr_gather_version(void) {
    int v = 0;
    return ncclGetVersion(&v) == ncclSuccess ? v : 0;
}

int vr_gather_unique_id(void* id_out) {
    if (!id_out) return fail("id_out is NULL");
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

int vr_gather_init_rank(const void* id_bytes, int rank, int world, int device, vr_gather_t* out) {
    if (!id_bytes || !out) return fail("NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return fail("rank %d outside world %d", rank, world);
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device) != hipSuccess)
        return fail("hipSetDevice(%d) failed", device);
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    VrGatherOpaque* g = new (std::nothrow) VrGatherOpaque();
    if (!g) return fail("host allocation failed");
    g->rank = rank;
    g->world = world;
    g->device = device;
    const ncclResult_t r = ncclCommInitRank(&g->comm, world, id, rank);
    (void)hipSetDevice(prev);
    if (r != ncclSuccess) {
        delete g;
        return fail("ncclCommInitRank(rank %d of %d, device %d): %s", rank, world, device,
                    ncclGetErrorString(r));
    }
    *out = g;
    return 0;
}

int vr_gather_init_all(int n, const int* devices, vr_gather_t* out) {
    if (n < 1 || !devices || !out) return fail("bad arguments");
    std::vector<ncclComm_t> comms((size_t)n);
    NCCL_TRY(ncclCommInitAll(comms.data(), n, devices));
    for (int r = 0; r < n; ++r) {
        VrGatherOpaque* g = new (std::nothrow) VrGatherOpaque();
        if (!g) return fail("host allocation failed");
        g->comm = comms[(size_t)r];
        g->rank = r;
        g->world = n;
        g->device = devices[r];
        out[r] = g;
    }
    return 0;
}

int vr_gather_free(vr_gather_t g) {
    if (!g) return 0;
    if (g->comm) (void)ncclCommDestroy(g->comm);
    delete g;
    return 0;
}

int vr_gather_rank(vr_gather_t g) { return g ? g->rank : -1; }
int vr_gather_world(vr_gather_t g) { return g ? g->world : 0; }

int vr_gather_group_begin(void) {
    NCCL_TRY(ncclGroupStart());
    return 0;
}
int vr_gather_group_end(void) {
    NCCL_TRY(ncclGroupEnd());
    return 0;
}

int vr_gather_tiles(vr_gather_t g, const void* send, void* recv_base, int64_t rank_stride,
                    int64_t bytes, int self_transfer, void* stream) {
    if (!g) return fail("gather handle is NULL");
    if (bytes < 0) return fail("bytes < 0");
    hipStream_t hs = static_cast<hipStream_t>(stream);
    if (g->rank != 0) {
        if (!send) return fail("rank %d has nothing to send (send is NULL)", g->rank);
        NCCL_TRY(ncclSend(send, (size_t)bytes, ncclUint8, 0, g->comm, hs));
        return 0;
    }
    if (g->world > 1 && !recv_base) return fail("the root has no receive buffer");
    NCCL_TRY(ncclGroupStart());
    ncclResult_t r = ncclSuccess;
    if (self_transfer) {
        if (!send || !recv_base) {
            (void)ncclGroupEnd();
            return fail("self transfer needs both buffers");
        }
        r = ncclSend(send, (size_t)bytes, ncclUint8, 0, g->comm, hs);
        if (r == ncclSuccess) r = ncclRecv(recv_base, (size_t)bytes, ncclUint8, 0, g->comm, hs);
    }
    for (int p = 1; p < g->world && r == ncclSuccess; ++p)
        r = ncclRecv(static_cast<char*>(recv_base) + rank_stride * p, (size_t)bytes, ncclUint8, p, g->comm, hs);
    const ncclResult_t e = ncclGroupEnd();
    if (r != ncclSuccess) return fail("ncclSend / ncclRecv: %s", ncclGetErrorString(r));
    if (e != ncclSuccess) return fail("ncclGroupEnd: %s", ncclGetErrorString(e));
    return 0;
}

}  // extern "C"
