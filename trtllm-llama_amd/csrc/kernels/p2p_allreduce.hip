// One-shot peer-to-peer all-reduce for the decode step's [B, D] fp16 partial sums.
//
// The tensor-parallel decode step needs 2 sum all-reduces per layer of 8 KB (B = 1): 64 per token.  A ring
// (RCCL) is latency-bound at that size; xGMI is point to point, so every rank can write its vector straight into a
// slot of every peer's inbox, raise a flag, wait for its own inbox to fill and add the slots up - one hop:
//
//   inbox[gen][r]  <- rank r's vector   (uncached device memory, mapped into every peer with hipIpc)
//   flag [gen][r]  <- epoch             (after a system-scope fence)
//   out = fp16( sum_r float(inbox[gen][r]) )   in rank order: every rank computes bit-identical sums
//
// Two generations (epoch parity): a rank can be at most one all-reduce ahead of a peer, because it cannot finish
// all-reduce k + 1 without the peer's contribution, which the peer sends only after it has read generation k.
// The epoch lives in device memory and is advanced by the kernel itself, so the launch is graph-replayable.
// Every spin is bounded; a timeout raises `error` instead of hanging the GPU: the launch that timed out and every later one
// leave their buffers alone, and the host fails the call at its next synchronisation point (session.cpp check_comm).
#include "dev_utils.h"
#include "kernels.h"

namespace tllm
{
namespace kernels
{
using namespace dev;

namespace
{

__global__ __launch_bounds__(1024) void p2p_allreduce_kernel(const P2PParams p)
{
    __shared__ uint32_t s_epoch;
    const int tid = threadIdx.x;
    __shared__ uint32_t s_failed;
    if (tid == 0)
    {
        s_epoch = *p.epoch + 1;
        s_failed = *p.error;
    }
    __syncthreads();
    // A time-out is sticky: once a wait has expired the inboxes and the epochs of the ranks can no longer be trusted, so every
    // later launch (the rest of a replayed step graph) returns at once instead of summing stale slots and spinning again.
    // The host sees the flag at its next synchronisation point, fails the call and takes this transport out of service
    // (runtime/session.cpp check_comm; comm::p2p::destroy + create starts afresh).
    if (s_failed)
        return;
    const uint32_t epoch = s_epoch;
    const int gen = epoch & 1, W = p.world;
    const size_t slot16 = p.slot_bytes / 16;
    uint4* x = reinterpret_cast<uint4*>(p.x);
    // 1. my vector -> slot [gen][rank] of every inbox (my own included)
    for (int v = tid; v < p.n16; v += blockDim.x)
    {
        const uint4 val = x[v];
        for (int r = 0; r < W; ++r)
        {
            uint4* dst = reinterpret_cast<uint4*>(p.peer[r]) + ((size_t) gen * W + p.rank) * slot16 + v;
            __builtin_nontemporal_store(val.x, &dst->x);
            __builtin_nontemporal_store(val.y, &dst->y);
            __builtin_nontemporal_store(val.z, &dst->z);
            __builtin_nontemporal_store(val.w, &dst->w);
        }
    }
    __threadfence_system();
    __syncthreads();
    // 2. signal every peer, 3. wait until every peer has signalled me
    if (tid < W)
    {
        uint32_t* pf = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(p.peer[tid]) + p.flag_offset) + gen * W + p.rank;
        __hip_atomic_store(pf, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint32_t* mf = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(p.peer[p.rank]) + p.flag_offset) + gen * W + tid;
        int spins = 0;
        while (__hip_atomic_load(mf, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != epoch)
        {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > p.max_spins)
            {
                atomicExch(p.error, epoch ? epoch : 1u);
                break;
            }
        }
    }
    __syncthreads();
    __threadfence_system();
    if (__hip_atomic_load(p.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        return; // this launch timed out: the inbox may be stale or half written - leave x alone, do not advance the epoch
    const uint4* in = reinterpret_cast<const uint4*>(p.peer[p.rank]) + (size_t) gen * W * slot16;
    if (p.gather_out)
    {
        // 4'. all-gather: copy the slots out in rank order
        uint4* out = reinterpret_cast<uint4*>(p.gather_out);
        for (int v = tid; v < p.n16 * W; v += blockDim.x)
        {
            const int r = v / p.n16, i = v % p.n16;
            const uint4* src = in + (size_t) r * slot16 + i;
            uint4 q;
            q.x = __builtin_nontemporal_load(&src->x);
            q.y = __builtin_nontemporal_load(&src->y);
            q.z = __builtin_nontemporal_load(&src->z);
            q.w = __builtin_nontemporal_load(&src->w);
            out[v] = q;
        }
    }
    else
    {
        // 4. sum the slots in rank order
        for (int v = tid; v < p.n16; v += blockDim.x)
        {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int r = 0; r < W; ++r)
            {
                const uint4* src = in + (size_t) r * slot16 + v;
                uint4 q;
                q.x = __builtin_nontemporal_load(&src->x);
                q.y = __builtin_nontemporal_load(&src->y);
                q.z = __builtin_nontemporal_load(&src->z);
                q.w = __builtin_nontemporal_load(&src->w);
                const uint32_t w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                {
                    acc[2 * j] += h2f((uint16_t) (w4[j] & 0xffffu));
                    acc[2 * j + 1] += h2f((uint16_t) (w4[j] >> 16));
                }
            }
            x[v] = make_uint4(pack_h2(acc[0], acc[1]), pack_h2(acc[2], acc[3]), pack_h2(acc[4], acc[5]), pack_h2(acc[6], acc[7]));
        }
    }
    if (tid == 0)
        *p.epoch = epoch;
}

} // namespace

int launch_p2p_allreduce(const P2PParams& p, hipStream_t stream)
{
    if (p.world < 2 || p.world > 8 || p.n16 <= 0 || (size_t) p.n16 * 16 > p.slot_bytes || (reinterpret_cast<uintptr_t>(p.x) & 15))
    {
        set_error("p2p all-reduce: bad arguments (world %d, %d x 16 B, slot %zu B)", p.world, p.n16, p.slot_bytes);
        return -1;
    }
    const int threads = p.n16 >= 1024 ? 1024 : (p.n16 >= 512 ? 512 : 256);
    hipLaunchKernelGGL(p2p_allreduce_kernel, dim3(1), dim3(threads), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
    {
        set_error("p2p all-reduce launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}

} // namespace kernels
} // namespace tllm
