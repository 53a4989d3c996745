// Multi-GPU exchange of interface rows behind the C ABI: one process (or thread) per GPU, RCCL point-to-point transfers
// (ncclSend / ncclRecv) on a side stream next to the assembly launches, the received rows added on the device.
// Replaces, across partitions, the single-address-space scatter of CsrParAssembler::assemble_into_csr (global.rs:314-376);
// rows of interface nodes are the only data that crosses a partition boundary, each interface rides one xGMI
// link, no collective over the matrix.  RCCL is loaded at run time (dlopen) so that the library loads without it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/fenris_hip.h"
#include "group_internal.hpp"

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
    bool load() {
        if (handle) return true;
        // FENRIS_HIP_RCCL_LIB names the library (tests point it at a file that does not exist: the group calls must then report
        // FH_UNSUPPORTED, not crash)
        const char* forced = std::getenv("FENRIS_HIP_RCCL_LIB");
        const char* dflt[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        std::string why = "not found";
        for (const char* name : dflt) {
            if (forced && *forced) name = forced;
            handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
            const char* m = dlerror();   // once: the call clears the message
            if (m) why = m;
            if (forced && *forced) break;
        }
        if (!handle) { error = "dlopen(librccl): " + why; return false; }
#define SYM(field, sym)                                                    \
    field = reinterpret_cast<decltype(field)>(dlsym(handle, #sym));        \
    if (!field) { error = "librccl: missing symbol " #sym; dlclose(handle); handle = nullptr; return false; }
        SYM(GetUniqueId, ncclGetUniqueId)
        SYM(CommInitRank, ncclCommInitRank)
        SYM(CommDestroy, ncclCommDestroy)
        SYM(CommCount, ncclCommCount)
        SYM(Send, ncclSend)
        SYM(Recv, ncclRecv)
        SYM(GroupStart, ncclGroupStart)
        SYM(GroupEnd, ncclGroupEnd)
        SYM(GetErrorString, ncclGetErrorString)
#undef SYM
        return true;
    }
};
Rccl g_rccl;

__global__ void __launch_bounds__(256) k_add_into(double* dst, const double* src, unsigned long long n) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256)
        dst[i] += src[i];
}

}  // namespace

struct fh_group {
    fh_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t side = nullptr;
    hipEvent_t ready = nullptr, done = nullptr;
    int send_peer = -1, recv_peer = -1;
    uint64_t send_first = 0, send_count = 0, recv_first = 0, recv_count = 0;
    double* recv_buf = nullptr;
    uint64_t recv_cap = 0;
    bool in_flight = false;
};

#define G_HIP(g, call)                                                                     \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) return fh_internal_fail((g)->ctx, FH_HIP_ERROR, std::string(#call ": ") + hipGetErrorString(e_)); \
    } while (0)
#define G_NCCL(g, call)                                                                    \
    do {                                                                                   \
        ncclResult_t r_ = (call);                                                          \
        if (r_ != ncclSuccess) return fh_internal_fail((g)->ctx, FH_HIP_ERROR, std::string(#call ": ") + g_rccl.GetErrorString(r_)); \
    } while (0)

extern "C" {

int fh_group_unique_id(uint8_t id[FH_GROUP_ID_BYTES]) {
    static_assert(FH_GROUP_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id is an ncclUniqueId");
    if (!id) return FH_BAD_ARGUMENT;
    if (!g_rccl.load()) return FH_UNSUPPORTED;
    ncclUniqueId u;
    if (g_rccl.GetUniqueId(&u) != ncclSuccess) return FH_HIP_ERROR;
    std::memcpy(id, u.internal, FH_GROUP_ID_BYTES);
    return FH_OK;
}

int fh_group_create(fh_ctx* c, const uint8_t id[FH_GROUP_ID_BYTES], int rank, int world, fh_group** out) {
    if (!c || !id || !out || world < 1 || rank < 0 || rank >= world) return FH_BAD_ARGUMENT;
    if (!g_rccl.load()) return fh_internal_fail(c, FH_UNSUPPORTED, g_rccl.error);
    fh_group* g = new fh_group();
    g->ctx = c;
    g->rank = rank;
    g->world = world;
    g->device = fh_internal_device(c);
    DevGuardExt dev_guard_(g->device);
    if (hipSetDevice(g->device) != hipSuccess || hipStreamCreateWithFlags(&g->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&g->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g->done, hipEventDisableTiming) != hipSuccess) {
        fh_group_destroy(g);
        return fh_internal_fail(c, FH_HIP_ERROR, "fh_group_create: stream / event creation failed");
    }
    ncclUniqueId u;
    std::memcpy(u.internal, id, FH_GROUP_ID_BYTES);
    const ncclResult_t r = g_rccl.CommInitRank(&g->comm, world, u, rank);
    if (r != ncclSuccess) {
        const std::string msg = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r);
        g->comm = nullptr;
        fh_group_destroy(g);
        return fh_internal_fail(c, FH_HIP_ERROR, msg);
    }
    *out = g;
    return FH_OK;
}

int fh_group_size(const fh_group* g, int* ranks) {
    if (!g || !ranks) return FH_BAD_ARGUMENT;
    if (!g->comm || !g_rccl.CommCount) return FH_INVALID_STATE;
    return g_rccl.CommCount(g->comm, ranks) == ncclSuccess ? FH_OK : FH_HIP_ERROR;
}

void fh_group_destroy(fh_group* g) {
    if (!g) return;
    DevGuardExt dev_guard_(g->device);
    if (g->side) (void)hipStreamSynchronize(g->side);
    if (g->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(g->comm);
    if (g->recv_buf) (void)hipFree(g->recv_buf);
    if (g->ready) (void)hipEventDestroy(g->ready);
    if (g->done) (void)hipEventDestroy(g->done);
    if (g->side) (void)hipStreamDestroy(g->side);
    delete g;
}

int fh_group_set_exchange(fh_group* g, int send_peer, uint64_t send_first, uint64_t send_count, int recv_peer, uint64_t recv_first,
                          uint64_t recv_count) {
    if (!g) return FH_BAD_ARGUMENT;
    if (g->in_flight) return fh_internal_fail(g->ctx, FH_INVALID_STATE, "fh_group_set_exchange: an exchange is in flight");
    if (send_peer >= g->world || recv_peer >= g->world || send_peer == g->rank || recv_peer == g->rank)
        return fh_internal_fail(g->ctx, FH_BAD_ARGUMENT, "fh_group_set_exchange: bad peer");
    g->send_peer = (send_peer >= 0 && send_count) ? send_peer : -1;
    g->recv_peer = (recv_peer >= 0 && recv_count) ? recv_peer : -1;
    g->send_first = send_first; g->send_count = send_count;
    g->recv_first = recv_first; g->recv_count = recv_count;
    if (g->recv_peer >= 0 && g->recv_cap < recv_count) {
        DevGuardExt dev_guard_(g->device);
        if (g->recv_buf) (void)hipFree(g->recv_buf);
        g->recv_buf = nullptr;
        g->recv_cap = 0;
        G_HIP(g, hipMalloc(reinterpret_cast<void**>(&g->recv_buf), sizeof(double) * recv_count));
        g->recv_cap = recv_count;
    }
    return FH_OK;
}

int fh_group_exchange_start(fh_group* g, double* values_dev) {
    if (!g || !values_dev) return FH_BAD_ARGUMENT;
    if (g->in_flight) return fh_internal_fail(g->ctx, FH_INVALID_STATE, "fh_group_exchange_start: already started");
    if (g->send_peer < 0 && g->recv_peer < 0) { g->in_flight = true; return FH_OK; }
    DevGuardExt dev_guard_(g->device);
    hipStream_t main = fh_internal_stream(g->ctx);
    // the transfers are ordered after everything enqueued so far on the context's stream (the launch that produced the rows
    // to send) and run beside whatever is enqueued next
    G_HIP(g, hipEventRecord(g->ready, main));
    G_HIP(g, hipStreamWaitEvent(g->side, g->ready, 0));
    // a failure inside the RCCL group must still close it, and must not leave the exchange marked as started
    G_NCCL(g, g_rccl.GroupStart());
    ncclResult_t r = ncclSuccess;
    const char* what = "";
    if (g->send_peer >= 0) { r = g_rccl.Send(values_dev + g->send_first, g->send_count, ncclDouble, g->send_peer, g->comm, g->side); what = "ncclSend: "; }
    if (r == ncclSuccess && g->recv_peer >= 0) { r = g_rccl.Recv(g->recv_buf, g->recv_count, ncclDouble, g->recv_peer, g->comm, g->side); what = "ncclRecv: "; }
    const ncclResult_t r_end = g_rccl.GroupEnd();
    if (r != ncclSuccess) return fh_internal_fail(g->ctx, FH_HIP_ERROR, std::string(what) + g_rccl.GetErrorString(r));
    if (r_end != ncclSuccess) return fh_internal_fail(g->ctx, FH_HIP_ERROR, std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(r_end));
    G_HIP(g, hipEventRecord(g->done, g->side));
    g->in_flight = true;
    return FH_OK;
}

int fh_group_exchange_finish(fh_group* g, double* values_dev) {
    if (!g || !values_dev) return FH_BAD_ARGUMENT;
    if (!g->in_flight) return fh_internal_fail(g->ctx, FH_INVALID_STATE, "fh_group_exchange_finish: nothing started");
    g->in_flight = false;
    if (g->send_peer < 0 && g->recv_peer < 0) return FH_OK;
    DevGuardExt dev_guard_(g->device);
    hipStream_t main = fh_internal_stream(g->ctx);
    G_HIP(g, hipStreamWaitEvent(main, g->done, 0));   // also orders later writes to the sent rows behind the send
    if (g->recv_peer >= 0) {
        const unsigned grid = (unsigned)((g->recv_count + 255) / 256 > 65536 ? 65536 : (g->recv_count + 255) / 256);
        hipLaunchKernelGGL(k_add_into, dim3(grid), dim3(256), 0, main, values_dev + g->recv_first, g->recv_buf, (unsigned long long)g->recv_count);
        G_HIP(g, hipGetLastError());
    }
    return FH_OK;
}

}  // extern "C"
