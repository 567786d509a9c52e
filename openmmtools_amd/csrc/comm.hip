// RCCL behind the C ABI: the one data-path collective of the replica-exchange iteration -- every rank's rows of the reduced
// potential matrix to every rank -- for hosts that are not Python / torch.distributed (the reference's seam is mpiplus:
// multistatesampler.py:1296-1311 distributes replicas over ranks and gathers their results; replicaexchange.py:255 mixes on
// one rank and broadcasts).  Here: one process per GPU, replicas block-partitioned by the host (remd_set_replicas' r_begin /
// R_local), u_kl rows gathered over xGMI into every rank's own matrix, then the SAME deterministic mix on every rank.
//
// librccl is opened at run time (the first remd_comm_* call), not linked: a host that already carries RCCL (PyTorch-ROCm ships
// one) keeps using its copy, and hosts that never shard do not load it at all.
#include "remd_internal.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <string.h>
#include <mutex>

namespace {
struct rccl_api {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};
rccl_api g_rccl;
std::mutex g_rccl_mutex;

const rccl_api* rccl()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib || !g_rccl.err.empty()) return &g_rccl;
    const char* env = getenv("REMD_RCCL_LIB");
    const char* names[] = { env ? env : "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    void* lib = nullptr;
    for (const char* n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!lib) { g_rccl.err = std::string("librccl not found (") + dlerror() + ")"; return &g_rccl; }
    bool ok = true;
    auto sym = [&](const char* name) { void* p = dlsym(lib, name); if (!p) { ok = false; g_rccl.err = std::string("librccl lacks ") + name; } return p; };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
    g_rccl.Broadcast = (decltype(g_rccl.Broadcast))sym("ncclBroadcast");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    if (ok) g_rccl.lib = lib; else dlclose(lib);
    return &g_rccl;
}
}  // namespace

#define RCCL_CHECK(h, api, expr) do { ncclResult_t _e = (expr); if (_e != ncclSuccess) \
    return remd_fail(h, -2, std::string(#expr) + ": " + (api)->GetErrorString(_e)); } while (0)

static_assert(sizeof(ncclUniqueId) == REMD_COMM_ID_BYTES, "REMD_COMM_ID_BYTES follows ncclUniqueId");

extern "C" {

int remd_comm_unique_id(void* id)
{
    if (!id) return remd_fail(nullptr, -1, "remd_comm_unique_id: null pointer");
    const rccl_api* api = rccl();
    if (!api->lib) return remd_fail(nullptr, -2, "remd_comm_unique_id: " + api->err);
    ncclUniqueId u;
    RCCL_CHECK(nullptr, api, api->GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return 0;
}

int remd_comm_init(remd_handle h, int rank, int world, const void* id)
{
    if (!h) return remd_fail(nullptr, -1, "remd_comm_init: null handle");
    if (world < 1 || rank < 0 || rank >= world) return remd_fail(h, -1, "remd_comm_init: rank " + std::to_string(rank) + " of " + std::to_string(world));
    if (!id) return remd_fail(h, -1, "remd_comm_init: null communicator id (remd_comm_unique_id on rank 0, passed on by the host)");
    const rccl_api* api = rccl();
    if (!api->lib) return remd_fail(h, -2, "remd_comm_init: " + api->err);
    remd_comm_release(h);
    REMD_CHECK(h, hipSetDevice(h->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t c = nullptr;
    RCCL_CHECK(h, api, api->CommInitRank(&c, world, u, rank));
    h->comm = c; h->comm_rank = rank; h->comm_world = world; h->comm_part_current = false;
    REMD_CHECK(h, hipMalloc(&h->d_comm_part, sizeof(long long) * 2 * (size_t)(world + 1)));
    return 0;
}

// Every rank's block (r_begin, R_local): exchanged once per remd_set_replicas, then cached.  The blocks must tile
// 0 .. R_global-1 in rank order, each replica on exactly one rank.
static int comm_exchange_blocks(remd_ctx* h, const rccl_api* api)
{
    const int W = h->comm_world;
    const long long mine[2] = { h->r_begin, h->R };
    long long* d_mine = h->d_comm_part + 2 * (size_t)W;
    REMD_CHECK(h, hipMemcpyAsync(d_mine, mine, sizeof(mine), hipMemcpyHostToDevice, h->stream));
    RCCL_CHECK(h, api, api->AllGather(d_mine, h->d_comm_part, 2, ncclInt64, (ncclComm_t)h->comm, h->stream));
    std::vector<long long> all(2 * (size_t)W);
    REMD_CHECK(h, hipMemcpyAsync(all.data(), h->d_comm_part, sizeof(long long) * all.size(), hipMemcpyDeviceToHost, h->stream));
    REMD_CHECK(h, hipStreamSynchronize(h->stream));
    h->comm_begin.assign(W, 0); h->comm_count.assign(W, 0);
    long long next = 0;
    for (int r = 0; r < W; ++r) {
        h->comm_begin[r] = all[2 * r]; h->comm_count[r] = all[2 * r + 1];
        if (h->comm_begin[r] != next || h->comm_count[r] < 0)
            return remd_fail(h, -1, "remd_comm_all_gather_energies: the ranks' replica blocks do not tile 0..R_global-1 in rank order (rank "
                             + std::to_string(r) + " holds " + std::to_string(h->comm_begin[r]) + "+" + std::to_string(h->comm_count[r]) + ")");
        next += h->comm_count[r];
    }
    if (next != h->R_global) return remd_fail(h, -1, "remd_comm_all_gather_energies: the ranks hold " + std::to_string(next) + " of " + std::to_string(h->R_global) + " replicas");
    h->comm_part_current = true;
    return 0;
}

int remd_comm_all_gather_energies(remd_handle h)
{
    if (!h || !h->d_ukl) return remd_fail(h, -1, "remd_comm_all_gather_energies: states/replicas not set");
    if (!h->comm) {
        if (h->R == h->R_global) return 0;                        // one process holds every replica: the matrix is complete
        return remd_fail(h, -1, "remd_comm_all_gather_energies: replicas are sharded but remd_comm_init was not called");
    }
    const rccl_api* api = rccl();
    REMD_CHECK(h, hipSetDevice(h->device));
    if (!h->comm_part_current) { const int rc = comm_exchange_blocks(h, api); if (rc) return rc; }
    // ragged blocks: one broadcast per rank inside one group (a single fused launch over xGMI), each in place in the full matrix
    const size_t K = (size_t)h->K;
    RCCL_CHECK(h, api, api->GroupStart());
    for (int r = 0; r < h->comm_world; ++r) {
        if (h->comm_count[r] == 0) continue;
        double* block = h->d_ukl + (size_t)h->comm_begin[r] * K;
        const ncclResult_t e = api->Broadcast(block, block, (size_t)h->comm_count[r] * K, ncclDouble, r, (ncclComm_t)h->comm, h->stream);
        if (e != ncclSuccess) { api->GroupEnd(); return remd_fail(h, -2, std::string("ncclBroadcast: ") + api->GetErrorString(e)); }
    }
    RCCL_CHECK(h, api, api->GroupEnd());
    return 0;
}

int remd_comm_finalize(remd_handle h)
{
    if (!h) return 0;
    remd_comm_release(h);
    return 0;
}

}  // extern "C"

void remd_comm_release(remd_ctx* h)
{
    if (h->comm) {
        hipStreamSynchronize(h->stream);
        const rccl_api* api = rccl();
        if (api->lib) api->CommDestroy((ncclComm_t)h->comm);
        h->comm = nullptr;
    }
    if (h->d_comm_part) { hipFree(h->d_comm_part); h->d_comm_part = nullptr; }
    h->comm_rank = 0; h->comm_world = 1; h->comm_part_current = false;
}
