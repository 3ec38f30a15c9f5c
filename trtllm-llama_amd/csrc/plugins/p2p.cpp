// Host side of the one-shot peer-to-peer all-reduce (kernels/p2p_allreduce.hip): one uncached inbox region per rank,
// exported with hipIpc, mapped by every peer.  The exchange of the 64-byte handles is the caller's business (bench.py
// and tensorrt_llm/parallel.py use torch.distributed), like the RCCL unique id.
#include "comm.h"
#include "../kernels/kernels.h"
#include <cstring>
#include <mutex>

namespace tllm
{
namespace comm
{
namespace p2p
{
namespace
{
struct State
{
    int world = 0, rank = -1;
    size_t slot_bytes = 0, flag_offset = 0, region_bytes = 0;
    void* local = nullptr;          // my region
    void* peer[8] = {};             // every rank's region as mapped here (peer[rank] == local)
    uint32_t* counters = nullptr;   // [0] epoch, [1] error  (ordinary device memory)
    bool attached = false, enabled = false;
    bool fused = true;          // the fused layer seam (all-reduce + residual + RMSNorm + quantiser) passed its validation
    bool out_of_service = false; // a time-out took the transport out: enable(true) is refused until create + attach
    int max_spins = 0;
};
// Bumped by every change of what a captured step graph may contain (create, enable on / off, fused on / off, time-out): a
// session compares it with the value it captured its graph under and re-captures on a difference (session.cpp).
uint64_t g_generation = 1;
// Bumped by disable_after_error only: a session whose last check saw an older value has work in flight (or in a captured
// graph) that ran against the broken group, and must fail that call (session.cpp check_comm).
uint64_t g_error_generation = 0;
std::mutex g_mu;
State g;

void release_locked()
{
    for (int r = 0; r < g.world; ++r)
        if (g.peer[r] && r != g.rank)
            (void) hipIpcCloseMemHandle(g.peer[r]);
    if (g.local)
        (void) hipFree(g.local);
    if (g.counters)
        (void) hipFree(g.counters);
    g = State();
    ++g_generation;
}
} // namespace

int create(int world, int rank, size_t max_bytes, void* handle64)
{
    std::lock_guard<std::mutex> lk(g_mu);
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t must be 64 bytes");
    if (world < 2 || world > 8 || rank < 0 || rank >= world || max_bytes == 0 || !handle64)
    {
        set_error("p2p: bad arguments (world %d in [2, 8], rank %d)", world, rank);
        return -1;
    }
    release_locked();
    g.world = world;
    g.rank = rank;
    g.slot_bytes = (max_bytes + 255) / 256 * 256;
    g.flag_offset = 2 * (size_t) world * g.slot_bytes;
    g.region_bytes = g.flag_offset + 4096;
    // uncached: peers write it over xGMI and the owner must see those writes without an L2 line in the way
    hipError_t e = hipExtMallocWithFlags(&g.local, g.region_bytes, hipDeviceMallocUncached);
    if (e != hipSuccess)
    {
        set_error("p2p: hipExtMallocWithFlags(uncached, %zu): %s", g.region_bytes, hipGetErrorString(e));
        release_locked();
        return -1;
    }
    if (hipMemset(g.local, 0, g.region_bytes) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&g.counters), 256) != hipSuccess
        || hipMemset(g.counters, 0, 256) != hipSuccess)
    {
        set_error("p2p: allocation of the counters failed");
        release_locked();
        return -1;
    }
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, g.local);
    if (e != hipSuccess)
    {
        set_error("p2p: hipIpcGetMemHandle: %s", hipGetErrorString(e));
        release_locked();
        return -1;
    }
    memcpy(handle64, &h, 64);
    g.peer[rank] = g.local;
    return 0;
}

int attach(const void* handles)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g.local || !handles)
    {
        set_error("p2p: attach before create");
        return -1;
    }
    for (int r = 0; r < g.world; ++r)
    {
        if (r == g.rank)
            continue;
        hipIpcMemHandle_t h;
        memcpy(&h, static_cast<const char*>(handles) + 64 * r, 64);
        hipError_t e = hipIpcOpenMemHandle(&g.peer[r], h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess)
        {
            set_error("p2p: hipIpcOpenMemHandle(rank %d): %s", r, hipGetErrorString(e));
            g.peer[r] = nullptr;
            return -1;
        }
    }
    g.attached = true;
    return 0;
}

int enable(bool on)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (on && g.out_of_service)
    {
        // the epochs of the ranks are no longer in step after a time-out; only a fresh region (create + attach on every rank) is
        set_error("p2p: the transport timed out earlier and is out of service; tllm_comm_p2p_create + attach bring it back");
        return -1;
    }
    const bool v = on && g.attached;
    if (v != g.enabled)
        ++g_generation;
    g.enabled = v;
    return (on && !v) ? -1 : 0;
}

void enable_fused(bool on)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (on != g.fused)
        ++g_generation;
    g.fused = on;
}

bool usable(int world, int64_t bytes)
{
    return g.enabled && g.world == world && bytes > 0 && (size_t) bytes <= g.slot_bytes && bytes % 16 == 0;
}

bool usable_fused(int world, int64_t bytes)
{
    return g.fused && usable(world, bytes);
}

bool usable_fused_flag()
{
    return g.fused;
}

uint64_t generation()
{
    return g_generation;
}

uint64_t error_generation()
{
    return g_error_generation;
}

int64_t slot_capacity(int world)
{
    return (g.enabled && g.world == world) ? (int64_t) g.slot_bytes : 0;
}

bool attached()
{
    return g.attached;
}

bool enabled()
{
    return g.enabled;
}

void set_max_spins(int n)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g.max_spins = n;
}

void disable_after_error()
{
    std::lock_guard<std::mutex> lk(g_mu);
    g.enabled = false;
    g.out_of_service = true;
    ++g_generation;
    ++g_error_generation;
    if (g.counters)
        (void) hipMemset(g.counters + 1, 0, 4);
    if (g.local)
        (void) hipMemset(static_cast<char*>(g.local) + g.flag_offset + kernels::P2P_POISON_OFFSET, 0, 4);
}

namespace
{
void fill_common(kernels::P2PParams& p)
{
    for (int r = 0; r < g.world; ++r)
        p.peer[r] = g.peer[r];
    p.world = g.world;
    p.rank = g.rank;
    p.slot_bytes = g.slot_bytes;
    p.flag_offset = g.flag_offset;
    p.epoch = g.counters;
    p.error = g.counters + 1;
    if (g.max_spins > 0)
        p.max_spins = g.max_spins;
}
} // namespace

int all_reduce_residual_norm(void* partial, int rows, int cols, const FusedTail& t, hipStream_t stream)
{
    if (!g.attached || rows < 1 || cols < 8 || (cols % 8))
    {
        set_error("p2p: fused all-reduce needs attached peers and rows of a multiple of 8 halfs");
        return -1;
    }
    kernels::P2PParams p;
    fill_common(p);
    p.x = partial;
    p.n16 = (int32_t) ((int64_t) rows * cols / 8);
    p.residual = t.x;
    p.x_out = t.x;
    p.norm_out = t.norm_out;
    p.gamma = t.gamma;
    p.eps = t.eps;
    p.rows = rows;
    p.cols = cols;
    p.quant = t.quant;
    p.quant_scale = t.quant_scale;
    p.dyn_scale_out = t.dyn_scale_out;
    return kernels::launch_p2p_allreduce(p, stream);
}

int all_reduce_f16(void* buf, int64_t count, hipStream_t stream)
{
    if (!g.attached || (count % 8))
    {
        set_error("p2p: all-reduce needs attached peers and a multiple of 8 halfs");
        return -1;
    }
    kernels::P2PParams p;
    fill_common(p);
    p.x = buf;
    p.n16 = (int32_t) (count / 8);
    return kernels::launch_p2p_allreduce(p, stream);
}

int all_gather(const void* in, void* out, int64_t bytes, hipStream_t stream)
{
    if (!g.attached || (bytes % 16) || (reinterpret_cast<uintptr_t>(out) & 15))
    {
        set_error("p2p: all-gather needs attached peers, a multiple of 16 bytes per rank and a 16-byte aligned output");
        return -1;
    }
    kernels::P2PParams p;
    fill_common(p);
    p.x = const_cast<void*>(in);
    p.gather_out = out;
    p.n16 = (int32_t) (bytes / 16);
    return kernels::launch_p2p_allreduce(p, stream);
}

int error_flag(uint32_t* out)
{
    *out = 0;
    if (!g.counters)
        return 0;
    // this rank's own time-out, or the word a peer that gave up wrote into this rank's region (a rank that has not launched
    // since then learns of it here: the decision to leave the transport is collective)
    uint32_t own = 0, poison = 0;
    if (hipMemcpy(&own, g.counters + 1, 4, hipMemcpyDeviceToHost) != hipSuccess)
        return -1;
    if (g.local
        && hipMemcpy(&poison, static_cast<char*>(g.local) + g.flag_offset + kernels::P2P_POISON_OFFSET, 4, hipMemcpyDeviceToHost) != hipSuccess)
        return -1;
    *out = own ? own : (poison ? (poison | 0x80000000u) : 0);
    return 0;
}

int destroy()
{
    std::lock_guard<std::mutex> lk(g_mu);
    release_locked();
    return 0;
}

} // namespace p2p
} // namespace comm
} // namespace tllm
