// One-shot peer-to-peer all-reduce for the decode step's [B, D] fp16 partial sums, with the two
// pointwise stages that frame it in the tensor-parallel layer fused in: the residual add in front of it and the NEXT
// RMSNorm (+ the SmoothQuant activation quantiser) behind it.
//
// The tensor-parallel decode step needs 2 sum all-reduces per layer of 8 KB (B = 1): 64 per token.  A ring
// (RCCL) is latency-bound at that size; xGMI is point to point, so every rank can write its vector straight into a
// slot of every peer's inbox, raise a flag, wait for its own inbox to fill and add the slots up - one hop:
//
//   inbox[gen][r]  <- rank r's vector   (uncached device memory, mapped into every peer with hipIpc)
//   flag [gen][r]  <- epoch             (after a system-scope fence)
//   sum = sum_r float(inbox[gen][r])    in rank order: every rank computes bit-identical sums
//
//   plain   : x        <- fp16(sum)                                         (in place; the AllReduce plugin's contract,
//                                                                            P/ncclPlugin/allreducePlugin.cpp:80-96)
//   fused   : x        <- fp16(residual + sum)                              (Q/llama_model.py:107-108, :117-118: hidden = residual + ...)
//             norm_out <- RMSNorm(x) * gamma  [-> int8, static or per token] (PY/functional.py:3195-3219; the next layer's
//                         input_layernorm / this layer's post_layernorm / ln_f; K/layernormKernels.cu:146-183 quantiser tail)
//
// The reference runs this seam as three graph nodes on every rank (allreduce plugin, elementwise add, rms_norm; with
// SmoothQuant a fourth, PY/quantization/layer.py:215,377); here the one workgroup that holds the whole reduced row does all of
// it, every rank produces bit-identical x / norm_out, no rank is special (the un-fused path lets rank 0 carry the residual into
// the sum), and the consuming GEMV starts from its operand type (PRO_NONE) instead of re-normalising the row in each of its
// ~900 workgroups.  Same rounding points as the GEMV prologue it replaces (gemv_impl.h PK_NORM): fp32 statistics,
// n16 = fp16(x * inv), y = fp16(n16 * gamma), q = sat(rni(y * s)).
//
// Two generations (epoch parity): a rank can be at most one all-reduce ahead of a peer, because it cannot finish
// all-reduce k + 1 without the peer's contribution, which the peer sends only after it has read generation k.
// The epoch lives in device memory and is advanced by the kernel itself, so the launch is graph-replayable.
//
// Failure is bounded and COLLECTIVE.  Every spin is bounded; a wait that expires raises this rank's `error` word and writes a
// poison word into EVERY peer's region.  A launch that finds its own region poisoned (at its start, or while it waits) raises
// its own error word too: a rank that never timed out itself learns within one launch that the group is broken, instead of
// carrying on peer-to-peer while the rank that timed out has gone back to RCCL.  Once the error word is set, this launch and
// every later one (the rest of a replayed step graph) leave their buffers alone and return at once; the host sees the word at its
// next synchronisation point, fails the call and takes the transport out of service on every rank (runtime/session.cpp
// check_comm; comm::p2p::disable_after_error clears the words, destroy + create + attach starts afresh).
#include "dev_utils.h"
#include "kernels.h"

namespace tllm
{
namespace kernels
{
using namespace dev;

namespace
{

__device__ __forceinline__ uint4 ld_slot(const uint4* src)
{
    uint4 q;
    q.x = __builtin_nontemporal_load(&src->x);
    q.y = __builtin_nontemporal_load(&src->y);
    q.z = __builtin_nontemporal_load(&src->z);
    q.w = __builtin_nontemporal_load(&src->w);
    return q;
}

__global__ __launch_bounds__(1024) void p2p_allreduce_kernel(const P2PParams p)
{
    __shared__ uint32_t s_epoch;
    __shared__ uint32_t s_failed;
    __shared__ float s_red[32];
    const int tid = threadIdx.x;
    const int W = p.world;
    volatile uint32_t* my_poison
        = reinterpret_cast<volatile uint32_t*>(reinterpret_cast<char*>(p.peer[p.rank]) + p.flag_offset + P2P_POISON_OFFSET);
    if (tid == 0)
    {
        s_epoch = *p.epoch + 1;
        uint32_t f = *p.error;
        if (!f)
        {
            const uint32_t po = __hip_atomic_load(const_cast<uint32_t*>(my_poison), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (po)
            {
                f = po | 0x80000000u; // a peer gave up: this rank is out as well
                *p.error = f;
            }
        }
        s_failed = f;
    }
    __syncthreads();
    if (s_failed)
        return;
    const uint32_t epoch = s_epoch;
    const int gen = epoch & 1;
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
            ++spins;
            if ((spins & 1023) == 0)
            {
                const uint32_t po = __hip_atomic_load(const_cast<uint32_t*>(my_poison), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (po)
                {
                    atomicExch(p.error, po | 0x80000000u);
                    break;
                }
            }
            if (spins > p.max_spins)
            {
                atomicExch(p.error, epoch ? epoch : 1u);
                break;
            }
        }
    }
    __syncthreads();
    __threadfence_system();
    const uint32_t err = __hip_atomic_load(p.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (err)
    {
        // this launch gave up: the inbox may be stale or half written - leave x alone, do not advance the epoch, and tell
        // every peer (idempotent: a rank that was told passes it on once more)
        if (tid < W && tid != p.rank)
        {
            uint32_t* pp = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(p.peer[tid]) + p.flag_offset + P2P_POISON_OFFSET);
            __hip_atomic_store(pp, (err & 0x7fffffffu) ? (err & 0x7fffffffu) : 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    const uint4* in = reinterpret_cast<const uint4*>(p.peer[p.rank]) + (size_t) gen * W * slot16;
    if (p.gather_out)
    {
        // 4'. all-gather: copy the slots out in rank order
        uint4* out = reinterpret_cast<uint4*>(p.gather_out);
        for (int v = tid; v < p.n16 * W; v += blockDim.x)
        {
            const int r = v / p.n16, i = v % p.n16;
            out[v] = ld_slot(in + (size_t) r * slot16 + i);
        }
    }
    else if (!p.norm_out)
    {
        // 4. sum the slots in rank order (+ the residual, when the caller fuses the add but not the norm)
        const uint4* res = reinterpret_cast<const uint4*>(p.residual);
        uint4* xo = p.x_out ? reinterpret_cast<uint4*>(p.x_out) : x;
        for (int v = tid; v < p.n16; v += blockDim.x)
        {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int r = 0; r < W; ++r)
            {
                const uint4 q = ld_slot(in + (size_t) r * slot16 + v);
                const uint32_t w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                {
                    acc[2 * j] += h2f((uint16_t) (w4[j] & 0xffffu));
                    acc[2 * j + 1] += h2f((uint16_t) (w4[j] >> 16));
                }
            }
            if (res)
            {
                const uint4 q = res[v];
                const uint32_t w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                {
                    acc[2 * j] += h2f((uint16_t) (w4[j] & 0xffffu));
                    acc[2 * j + 1] += h2f((uint16_t) (w4[j] >> 16));
                }
            }
            xo[v] = make_uint4(pack_h2(acc[0], acc[1]), pack_h2(acc[2], acc[3]), pack_h2(acc[4], acc[5]), pack_h2(acc[6], acc[7]));
        }
    }
    else
    {
        // 4''. fused tail, row by row: x <- fp16(residual + sum); norm_out <- RMSNorm(x) * gamma (-> int8)
        const int cols16 = p.cols / 8;
        const uint4* res = reinterpret_cast<const uint4*>(p.residual);
        const uint4* gam = reinterpret_cast<const uint4*>(p.gamma);
        uint4* xo = reinterpret_cast<uint4*>(p.x_out);
        const float qstatic = p.quant == 1 ? p.quant_scale[0] : 1.f;
        for (int row = 0; row < p.rows; ++row)
        {
            // pass A: the reduced row -> x_out, sum of squares of the ROUNDED values (what the GEMV prologue would read back)
            float ss = 0.f;
            for (int c = tid; c < cols16; c += blockDim.x)
            {
                const int v = row * cols16 + c;
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int r = 0; r < W; ++r)
                {
                    const uint4 q = ld_slot(in + (size_t) r * slot16 + v);
                    const uint32_t w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                    {
                        acc[2 * j] += h2f((uint16_t) (w4[j] & 0xffffu));
                        acc[2 * j + 1] += h2f((uint16_t) (w4[j] >> 16));
                    }
                }
                if (res)
                {
                    const uint4 q = res[v];
                    const uint32_t w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                    {
                        acc[2 * j] += h2f((uint16_t) (w4[j] & 0xffffu));
                        acc[2 * j + 1] += h2f((uint16_t) (w4[j] >> 16));
                    }
                }
                uint32_t o4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                {
                    const uint16_t lo = f2h(acc[2 * j]), hi = f2h(acc[2 * j + 1]);
                    const float f0 = h2f(lo), f1 = h2f(hi);
                    ss += f0 * f0 + f1 * f1;
                    o4[j] = (uint32_t) lo | ((uint32_t) hi << 16);
                }
                xo[v] = make_uint4(o4[0], o4[1], o4[2], o4[3]);
            }
            ss = block_sum(ss, s_red);
            const float inv = 1.0f / sqrtf(ss / (float) p.cols + p.eps);
            // pass B: y = fp16(fp16(x * inv) * gamma); the thread re-reads the vectors it wrote itself (program order)
            float amax = 0.f;
            for (int c = tid; c < cols16; c += blockDim.x)
            {
                const int v = row * cols16 + c;
                const uint4 q = xo[v], g = gam[c];
                const uint32_t x4[4] = {q.x, q.y, q.z, q.w}, g4[4] = {g.x, g.y, g.z, g.w};
                uint32_t y4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                {
                    const float n0 = h2f(f2h(h2f((uint16_t) (x4[j] & 0xffffu)) * inv));
                    const float n1 = h2f(f2h(h2f((uint16_t) (x4[j] >> 16)) * inv));
                    const uint16_t y0 = f2h(n0 * h2f((uint16_t) (g4[j] & 0xffffu)));
                    const uint16_t y1 = f2h(n1 * h2f((uint16_t) (g4[j] >> 16)));
                    amax = fmaxf(amax, fmaxf(fabsf(h2f(y0)), fabsf(h2f(y1))));
                    y4[j] = (uint32_t) y0 | ((uint32_t) y1 << 16);
                }
                if (p.quant == 0)
                    reinterpret_cast<uint4*>(p.norm_out)[v] = make_uint4(y4[0], y4[1], y4[2], y4[3]);
                else if (p.quant == 1)
                {
                    uint32_t o[2] = {0, 0};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                    {
                        const uint32_t b0 = (uint8_t) f2i8_rni_sat(h2f((uint16_t) (y4[j] & 0xffffu)) * qstatic);
                        const uint32_t b1 = (uint8_t) f2i8_rni_sat(h2f((uint16_t) (y4[j] >> 16)) * qstatic);
                        o[j >> 1] |= (b0 | (b1 << 8)) << (16 * (j & 1));
                    }
                    reinterpret_cast<uint2*>(p.norm_out)[v] = make_uint2(o[0], o[1]);
                }
            }
            if (p.quant == 2)
            {
                // per token: amax = max(T(1e-6), max|y|), q = sat(rni(y * (127 / amax))), scale = amax / 127 (K/quantization.cu:94-118)
                amax = block_max(amax, s_red);
                amax = fmaxf(amax, h2f(f2h(1e-6f)));
                const float qs = 127.f / amax;
                if (tid == 0)
                    p.dyn_scale_out[row] = amax / 127.f;
                for (int c = tid; c < cols16; c += blockDim.x)
                {
                    const int v = row * cols16 + c;
                    const uint4 q = xo[v], g = gam[c];
                    const uint32_t x4[4] = {q.x, q.y, q.z, q.w}, g4[4] = {g.x, g.y, g.z, g.w};
                    uint32_t o[2] = {0, 0};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                    {
                        const float n0 = h2f(f2h(h2f((uint16_t) (x4[j] & 0xffffu)) * inv));
                        const float n1 = h2f(f2h(h2f((uint16_t) (x4[j] >> 16)) * inv));
                        const float y0 = h2f(f2h(n0 * h2f((uint16_t) (g4[j] & 0xffffu))));
                        const float y1 = h2f(f2h(n1 * h2f((uint16_t) (g4[j] >> 16))));
                        const uint32_t b0 = (uint8_t) f2i8_rni_sat(y0 * qs);
                        const uint32_t b1 = (uint8_t) f2i8_rni_sat(y1 * qs);
                        o[j >> 1] |= (b0 | (b1 << 8)) << (16 * (j & 1));
                    }
                    reinterpret_cast<uint2*>(p.norm_out)[v] = make_uint2(o[0], o[1]);
                }
            }
            __syncthreads(); // s_red is reused by the next row
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
    if (p.norm_out
        && (p.gather_out || !p.x_out || !p.gamma || p.rows < 1 || p.cols < 8 || (p.cols % 8) || (int64_t) p.rows * p.cols != (int64_t) p.n16 * 8
            || p.quant < 0 || p.quant > 2 || (p.quant == 1 && !p.quant_scale) || (p.quant == 2 && !p.dyn_scale_out)
            || (reinterpret_cast<uintptr_t>(p.x_out) & 15) || (reinterpret_cast<uintptr_t>(p.norm_out) & 15)
            || (reinterpret_cast<uintptr_t>(p.gamma) & 15) || (reinterpret_cast<uintptr_t>(p.residual) & 15)))
    {
        set_error("p2p all-reduce: bad arguments of the fused residual + RMSNorm tail (rows %d, cols %d, quant %d)", p.rows, p.cols, p.quant);
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
