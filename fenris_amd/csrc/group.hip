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
#include <vector>

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

// list mode (arbitrary partitions): the rows of the listed nodes, node by node, into / out of one packed buffer.  One workgroup per
// node: first value and count from the node-level pattern (S x S values per column block of the node's S rows)
__global__ void __launch_bounds__(128) k_pack_rows(const unsigned* nodes, const unsigned long long* dst, const unsigned* noff, int ss, const double* vals,
                                                   double* buf) {
    const unsigned node = nodes[blockIdx.x];
    const unsigned long long first = (unsigned long long)ss * noff[node], count = (unsigned long long)ss * (noff[node + 1] - noff[node]);
    double* o = buf + dst[blockIdx.x];
    for (unsigned long long i = threadIdx.x; i < count; i += 128) o[i] = vals[first + i];
}
__global__ void __launch_bounds__(128) k_unpack_add_rows(const unsigned* nodes, const unsigned long long* src, const unsigned* noff, int ss,
                                                         const double* buf, double* vals) {
    const unsigned node = nodes[blockIdx.x];
    const unsigned long long first = (unsigned long long)ss * noff[node], count = (unsigned long long)ss * (noff[node + 1] - noff[node]);
    const double* in = buf + src[blockIdx.x];
    for (unsigned long long i = threadIdx.x; i < count; i += 128) vals[first + i] += in[i];
}
__global__ void __launch_bounds__(256) k_row_counts(const unsigned* nodes, unsigned long long n, const unsigned* noff, int ss, unsigned long long* count) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) count[i] = (unsigned long long)ss * (noff[nodes[i] + 1] - noff[nodes[i]]);
}

// node VECTORS (residual, source vector: `comp` values per node) of the listed nodes
__global__ void __launch_bounds__(256) k_pack_nodes(const unsigned* nodes, unsigned long long n, int comp, const double* vec, double* buf) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n * comp) buf[i] = vec[(unsigned long long)nodes[i / comp] * comp + i % comp];
}
__global__ void __launch_bounds__(256) k_unpack_add_nodes(const unsigned* nodes, unsigned long long n, int comp, const double* buf, double* vec) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n * comp) vec[(unsigned long long)nodes[i / comp] * comp + i % comp] += buf[i];
}

struct PeerList {   // one direction of the list-mode exchange
    std::vector<int> peers;
    std::vector<unsigned long long> peer_first, peer_count;   // per peer: first value / values in the packed buffer
    std::vector<unsigned long long> entry_first;              // per peer (+ 1): first entry of its list
    unsigned* nodes = nullptr;               // device: local node per entry
    unsigned long long* offs = nullptr;      // device: first value of every entry's rows in the packed buffer
    double* buf = nullptr;                   // device: the packed buffer
    double* vbuf = nullptr;                  // device: packed node vectors (entries x vcomp), allocated at the first vector exchange
    int vcomp = 0;
    unsigned long long entries = 0, values = 0;
    void release() {
        if (nodes) (void)hipFree(nodes);
        if (offs) (void)hipFree(offs);
        if (buf) (void)hipFree(buf);
        if (vbuf) (void)hipFree(vbuf);
        nodes = nullptr; offs = nullptr; buf = nullptr; vbuf = nullptr; vcomp = 0; entries = values = 0;
        peers.clear(); peer_first.clear(); peer_count.clear(); entry_first.clear();
    }
};

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
    bool vec_in_flight = false;
    bool list_mode = false;          // fh_group_set_exchange_nodes: packed lists, any number of peers
    unsigned long long pattern_gen = 0;   // ... built against this generation of the context's pattern (noff below points into it)
    PeerList snd, rcv;
    const unsigned* noff = nullptr;  // the context's node-level row offsets (device)
    int ss = 0;                      // S x S
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
    g->snd.release();
    g->rcv.release();
    if (g->ready) (void)hipEventDestroy(g->ready);
    if (g->done) (void)hipEventDestroy(g->done);
    if (g->side) (void)hipStreamDestroy(g->side);
    delete g;
}

int fh_group_set_exchange(fh_group* g, int send_peer, uint64_t send_first, uint64_t send_count, int recv_peer, uint64_t recv_first,
                          uint64_t recv_count) {
    if (!g) return FH_BAD_ARGUMENT;
    if (g->in_flight || g->vec_in_flight) return fh_internal_fail(g->ctx, FH_INVALID_STATE, "fh_group_set_exchange: an exchange is in flight");
    if (send_peer >= g->world || recv_peer >= g->world || send_peer == g->rank || recv_peer == g->rank)
        return fh_internal_fail(g->ctx, FH_BAD_ARGUMENT, "fh_group_set_exchange: bad peer");
    g->list_mode = false;
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


// one direction of the list-mode exchange: node lists per peer -> device arrays, packed-buffer offsets (the rows of a node are
// S x S x (its column blocks) values: counted on the device from the context's pattern, summed on the host)
static int build_peer_list(fh_group* g, PeerList& L, int npeers, const int32_t* peers, const uint64_t* offsets, const uint64_t* nodes,
                           uint64_t num_nodes) {
    L.release();
    if (npeers <= 0) return FH_OK;
    const uint64_t n = offsets[npeers];
    std::vector<unsigned> h_nodes((size_t)n);
    for (uint64_t i = 0; i < n; ++i) {
        if (nodes[i] >= num_nodes) return fh_internal_fail(g->ctx, FH_BAD_ARGUMENT, "fh_group_set_exchange_nodes: node index out of range");
        h_nodes[i] = (unsigned)nodes[i];
    }
    for (int p = 0; p < npeers; ++p) {
        if (peers[p] < 0 || peers[p] >= g->world || offsets[p + 1] < offsets[p])
            return fh_internal_fail(g->ctx, FH_BAD_ARGUMENT, "fh_group_set_exchange_nodes: bad peer or offsets");
        L.peers.push_back(peers[p]);
    }
    L.entry_first.assign(offsets, offsets + npeers + 1);
    L.entries = n;
    if (n == 0) { L.peer_first.assign((size_t)npeers, 0); L.peer_count.assign((size_t)npeers, 0); return FH_OK; }
    G_HIP(g, hipMalloc(reinterpret_cast<void**>(&L.nodes), sizeof(unsigned) * n));
    G_HIP(g, hipMalloc(reinterpret_cast<void**>(&L.offs), sizeof(unsigned long long) * n));
    hipStream_t main = fh_internal_stream(g->ctx);
    G_HIP(g, hipMemcpyAsync(L.nodes, h_nodes.data(), sizeof(unsigned) * n, hipMemcpyHostToDevice, main));
    hipLaunchKernelGGL(k_row_counts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, main, L.nodes, (unsigned long long)n, g->noff, g->ss, L.offs);
    G_HIP(g, hipGetLastError());
    std::vector<unsigned long long> cnt((size_t)n);
    G_HIP(g, hipMemcpyAsync(cnt.data(), L.offs, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost, main));
    G_HIP(g, hipStreamSynchronize(main));
    unsigned long long run = 0;
    for (int p = 0; p < npeers; ++p) {
        L.peer_first.push_back(run);
        for (uint64_t i = offsets[p]; i < offsets[p + 1]; ++i) { const unsigned long long c = cnt[i]; cnt[i] = run; run += c; }
        L.peer_count.push_back(run - L.peer_first.back());
    }
    L.values = run;
    G_HIP(g, hipMemcpyAsync(L.offs, cnt.data(), sizeof(unsigned long long) * n, hipMemcpyHostToDevice, main));
    G_HIP(g, hipStreamSynchronize(main));   // cnt / h_nodes are released on return
    if (run) G_HIP(g, hipMalloc(reinterpret_cast<void**>(&L.buf), sizeof(double) * run));
    return FH_OK;
}

int fh_group_set_exchange_nodes(fh_group* g, int num_send_peers, const int32_t* send_peers, const uint64_t* send_offsets, const uint64_t* send_nodes,
                                int num_recv_peers, const int32_t* recv_peers, const uint64_t* recv_offsets, const uint64_t* recv_nodes) {
    if (!g || num_send_peers < 0 || num_recv_peers < 0) return FH_BAD_ARGUMENT;
    if ((num_send_peers && (!send_peers || !send_offsets)) || (num_recv_peers && (!recv_peers || !recv_offsets)))
        return fh_internal_fail(g->ctx, FH_BAD_ARGUMENT, "fh_group_set_exchange_nodes: null list");
    if (g->in_flight || g->vec_in_flight) return fh_internal_fail(g->ctx, FH_INVALID_STATE, "fh_group_set_exchange_nodes: an exchange is in flight");
    g->list_mode = false;   // until both lists are in place (a failure below must not leave the mode set over released lists)
    const unsigned* ncols = nullptr;
    uint64_t N = 0;
    int S = 0;
    if (!fh_internal_pattern(g->ctx, &g->noff, &ncols, &N, &S))
        return fh_internal_fail(g->ctx, FH_INVALID_STATE, "fh_group_set_exchange_nodes: build the pattern first (fh_pattern)");
    g->ss = S * S;
    DevGuardExt dev_guard_(g->device);
    int rc = build_peer_list(g, g->snd, num_send_peers, send_peers, send_offsets, send_nodes, N);
    if (rc) return rc;
    rc = build_peer_list(g, g->rcv, num_recv_peers, recv_peers, recv_offsets, recv_nodes, N);
    if (rc) return rc;
    g->pattern_gen = fh_internal_pattern_gen(g->ctx);
    g->list_mode = true;
    return FH_OK;
}

// the lists hold device pointers into the context's pattern: a pattern rebuilt since (fh_set_mesh, a new mask followed by fh_pattern) frees them
static int list_pattern_current(fh_group* g, const char* who) {
    if (g->pattern_gen != 0 && g->pattern_gen == fh_internal_pattern_gen(g->ctx)) return FH_OK;
    return fh_internal_fail(g->ctx, FH_INVALID_STATE, std::string(who) + ": the context's pattern was rebuilt since fh_group_set_exchange_nodes -- set the node lists again");
}

static int list_exchange_start(fh_group* g, double* values_dev) {
    { const int rc_g = list_pattern_current(g, "fh_group_exchange_start"); if (rc_g) return rc_g; }
    DevGuardExt dev_guard_(g->device);
    hipStream_t main = fh_internal_stream(g->ctx);
    G_HIP(g, hipEventRecord(g->ready, main));
    G_HIP(g, hipStreamWaitEvent(g->side, g->ready, 0));
    if (g->snd.entries) {
        hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)g->snd.entries), dim3(128), 0, g->side, g->snd.nodes, g->snd.offs, g->noff, g->ss, values_dev, g->snd.buf);
        G_HIP(g, hipGetLastError());
    }
    G_NCCL(g, g_rccl.GroupStart());
    ncclResult_t r = ncclSuccess;
    for (size_t p = 0; p < g->snd.peers.size() && r == ncclSuccess; ++p)
        if (g->snd.peer_count[p]) r = g_rccl.Send(g->snd.buf + g->snd.peer_first[p], g->snd.peer_count[p], ncclDouble, g->snd.peers[p], g->comm, g->side);
    for (size_t p = 0; p < g->rcv.peers.size() && r == ncclSuccess; ++p)
        if (g->rcv.peer_count[p]) r = g_rccl.Recv(g->rcv.buf + g->rcv.peer_first[p], g->rcv.peer_count[p], ncclDouble, g->rcv.peers[p], g->comm, g->side);
    const ncclResult_t r_end = g_rccl.GroupEnd();
    if (r != ncclSuccess) return fh_internal_fail(g->ctx, FH_HIP_ERROR, std::string("ncclSend / ncclRecv: ") + g_rccl.GetErrorString(r));
    if (r_end != ncclSuccess) return fh_internal_fail(g->ctx, FH_HIP_ERROR, std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(r_end));
    G_HIP(g, hipEventRecord(g->done, g->side));
    g->in_flight = true;
    return FH_OK;
}

static int list_exchange_finish(fh_group* g, double* values_dev) {
    g->in_flight = false;
    DevGuardExt dev_guard_(g->device);
    hipStream_t main = fh_internal_stream(g->ctx);
    G_HIP(g, hipStreamWaitEvent(main, g->done, 0));
    // one launch per peer, in list order: a node may receive from several peers (the nodes of ONE peer's list are distinct), and the
    // additions happen in the same order every run
    for (size_t p = 0; p < g->rcv.peers.size(); ++p) {
        const unsigned long long e0 = g->rcv.entry_first[p], e1 = g->rcv.entry_first[p + 1];
        if (e1 == e0) continue;
        hipLaunchKernelGGL(k_unpack_add_rows, dim3((unsigned)(e1 - e0)), dim3(128), 0, main, g->rcv.nodes + e0, g->rcv.offs + e0, g->noff, g->ss, g->rcv.buf, values_dev);
        G_HIP(g, hipGetLastError());
    }
    return FH_OK;
}

// node vectors through the same lists: comp values per listed node
static int vec_buffers(fh_group* g, PeerList& L, int comp) {
    if (L.entries == 0 || L.vcomp == comp) return FH_OK;
    if (L.vbuf) (void)hipFree(L.vbuf);
    L.vbuf = nullptr;
    L.vcomp = 0;
    G_HIP(g, hipMalloc(reinterpret_cast<void**>(&L.vbuf), sizeof(double) * L.entries * comp));
    L.vcomp = comp;
    return FH_OK;
}

int fh_group_exchange_vector_start(fh_group* g, double* vec_dev, uint32_t components) {
    if (!g || !vec_dev || components == 0 || components > 16) return FH_BAD_ARGUMENT;
    if (!g->list_mode) return fh_internal_fail(g->ctx, FH_INVALID_STATE, "fh_group_exchange_vector_start: set the node lists first (fh_group_set_exchange_nodes)");
    if (g->vec_in_flight || g->in_flight) return fh_internal_fail(g->ctx, FH_INVALID_STATE, "fh_group_exchange_vector_start: an exchange is in flight");
    { const int rc_g = list_pattern_current(g, "fh_group_exchange_vector_start"); if (rc_g) return rc_g; }
    DevGuardExt dev_guard_(g->device);
    const int comp = (int)components;
    int rc = vec_buffers(g, g->snd, comp);
    if (rc) return rc;
    rc = vec_buffers(g, g->rcv, comp);
    if (rc) return rc;
    hipStream_t main = fh_internal_stream(g->ctx);
    G_HIP(g, hipEventRecord(g->ready, main));
    G_HIP(g, hipStreamWaitEvent(g->side, g->ready, 0));
    if (g->snd.entries) {
        const unsigned long long n = g->snd.entries;
        hipLaunchKernelGGL(k_pack_nodes, dim3((unsigned)((n * comp + 255) / 256)), dim3(256), 0, g->side, g->snd.nodes, n, comp, vec_dev, g->snd.vbuf);
        G_HIP(g, hipGetLastError());
    }
    G_NCCL(g, g_rccl.GroupStart());
    ncclResult_t r = ncclSuccess;
    for (size_t p = 0; p < g->snd.peers.size() && r == ncclSuccess; ++p) {
        const unsigned long long e0 = g->snd.entry_first[p], e1 = g->snd.entry_first[p + 1];
        if (e1 > e0) r = g_rccl.Send(g->snd.vbuf + e0 * comp, (e1 - e0) * comp, ncclDouble, g->snd.peers[p], g->comm, g->side);
    }
    for (size_t p = 0; p < g->rcv.peers.size() && r == ncclSuccess; ++p) {
        const unsigned long long e0 = g->rcv.entry_first[p], e1 = g->rcv.entry_first[p + 1];
        if (e1 > e0) r = g_rccl.Recv(g->rcv.vbuf + e0 * comp, (e1 - e0) * comp, ncclDouble, g->rcv.peers[p], g->comm, g->side);
    }
    const ncclResult_t r_end = g_rccl.GroupEnd();
    if (r != ncclSuccess) return fh_internal_fail(g->ctx, FH_HIP_ERROR, std::string("ncclSend / ncclRecv: ") + g_rccl.GetErrorString(r));
    if (r_end != ncclSuccess) return fh_internal_fail(g->ctx, FH_HIP_ERROR, std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(r_end));
    G_HIP(g, hipEventRecord(g->done, g->side));
    g->vec_in_flight = true;
    return FH_OK;
}

int fh_group_exchange_vector_finish(fh_group* g, double* vec_dev, uint32_t components) {
    if (!g || !vec_dev || components == 0 || components > 16) return FH_BAD_ARGUMENT;
    if (!g->vec_in_flight) return fh_internal_fail(g->ctx, FH_INVALID_STATE, "fh_group_exchange_vector_finish: nothing started");
    g->vec_in_flight = false;
    const int comp = (int)components;
    if ((g->rcv.entries && g->rcv.vcomp != comp)) return fh_internal_fail(g->ctx, FH_BAD_ARGUMENT, "fh_group_exchange_vector_finish: components differ from the start");
    DevGuardExt dev_guard_(g->device);
    hipStream_t main = fh_internal_stream(g->ctx);
    G_HIP(g, hipStreamWaitEvent(main, g->done, 0));
    for (size_t p = 0; p < g->rcv.peers.size(); ++p) {   // peer by peer: a node may receive from several
        const unsigned long long e0 = g->rcv.entry_first[p], e1 = g->rcv.entry_first[p + 1];
        if (e1 == e0) continue;
        hipLaunchKernelGGL(k_unpack_add_nodes, dim3((unsigned)(((e1 - e0) * comp + 255) / 256)), dim3(256), 0, main, g->rcv.nodes + e0, e1 - e0, comp,
                           g->rcv.vbuf + e0 * comp, vec_dev);
        G_HIP(g, hipGetLastError());
    }
    return FH_OK;
}

int fh_group_exchange_start(fh_group* g, double* values_dev) {
    if (!g || !values_dev) return FH_BAD_ARGUMENT;
    if (g->in_flight) return fh_internal_fail(g->ctx, FH_INVALID_STATE, "fh_group_exchange_start: already started");
    if (g->list_mode) return list_exchange_start(g, values_dev);
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
    if (g->list_mode) return list_exchange_finish(g, values_dev);
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
