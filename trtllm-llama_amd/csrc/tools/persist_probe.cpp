// Feasibility probe (GPU box only): a LLaMA-7B-sized decode layer as (A) five graph-captured streaming kernels vs
// (B) ONE persistent launch with grid barriers between the phases and the first weight tiles of the next phase
// requested BEFORE the barrier.  Each phase "reads W (sizes of the int8 layer GEMVs / KV), depends on a vector the
// previous phase wrote".  No arithmetic of interest - this measures the launch structure only.
//   build/persist_probe [layers=16] [iters=20] [wg_per_cu=1]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                                          \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess)                                                                                          \
        {                                                                                                              \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);                                      \
            return 1;                                                                                                  \
        }                                                                                                              \
    } while (0)

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

struct Phase
{
    const char* w;
    size_t bytes; // multiple of 8 KiB
};
struct Layer
{
    Phase ph[5];
};

__device__ __forceinline__ u4 ldnt(const void* p)
{
    return __builtin_nontemporal_load(reinterpret_cast<const u4*>(p));
}

// one wave streams tiles t = first, first + stride, ... of 8 KiB each (8 x 16 B per lane), double-buffered
struct Stream
{
    u4 buf[8];
    __device__ __forceinline__ void request(const char* w, size_t tile, int lane)
    {
        const char* p = w + tile * 8192 + lane * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            buf[i] = ldnt(p + i * 1024);
    }
    __device__ __forceinline__ uint32_t consume(uint32_t x)
    {
        uint32_t a = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            a += (buf[i].x ^ x) + (buf[i].y & x) + buf[i].z + buf[i].w;
        return a;
    }
};

__device__ __forceinline__ uint32_t run_phase(const Phase& ph, size_t wave, size_t nwaves, int lane, uint32_t x, Stream& s,
    bool first_requested)
{
    const size_t ntiles = ph.bytes / 8192;
    uint32_t acc = 0;
    size_t t = wave;
    if (!first_requested && t < ntiles)
        s.request(ph.w, t, lane);
    while (t < ntiles)
    {
        Stream nxt;
        const size_t tn = t + nwaves;
        if (tn < ntiles)
            nxt.request(ph.w, tn, lane);
        acc += s.consume(x);
        s = nxt;
        t = tn;
    }
    return acc;
}

// ---- (A'') phase kernel with pieces of the real GEMV's structure added one at a time
struct BigArgs
{
    Phase ph;
    char pad[240];
};
template <int MODE>
__global__ __launch_bounds__(256) void phase_kernel_x(BigArgs a, const uint32_t* xin, uint32_t* xout)
{
    extern __shared__ uint32_t lds[];
    const Phase ph = a.ph;
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t) blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t) gridDim.x * 4;
    Stream s;
    if (wave < ph.bytes / 8192)
        s.request(ph.w, wave, lane);
    uint32_t x = xin[threadIdx.x];
    if (MODE & 1)
    {
        // block-wide reduction of x through LDS, then everybody reads a normalised copy (2 barriers)
        uint32_t v = x * x;
        for (int o = 32; o; o >>= 1)
            v += __shfl_xor(v, o, 64);
        if (lane == 0)
            lds[threadIdx.x >> 6] = v;
        __syncthreads();
        v = lds[0] + lds[1] + lds[2] + lds[3];
        lds[16 + threadIdx.x] = x ^ v;
        __syncthreads();
        x = lds[16 + ((threadIdx.x * 7) & 255)];
    }
    uint32_t extra[64];
    if (MODE & 4)
    {
#pragma unroll
        for (int i = 0; i < 64; ++i)
            extra[i] = x * (i + 3);
    }
    uint32_t acc = run_phase(ph, wave, nwaves, lane, x, s, true);
    if (MODE & 4)
    {
#pragma unroll
        for (int i = 0; i < 64; ++i)
            acc += extra[i] ^ (acc >> (i & 7));
    }
    if (acc == 0x12345678u || (blockIdx.x == 0 && threadIdx.x < 256))
        xout[threadIdx.x] = acc + x;
}

// ---- (A) one kernel per phase
__global__ __launch_bounds__(256) void phase_kernel(Phase ph, const uint32_t* xin, uint32_t* xout)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t) blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t) gridDim.x * 4;
    Stream s;
    if (wave < ph.bytes / 8192)
        s.request(ph.w, wave, lane);
    const uint32_t x = xin[threadIdx.x]; // the dependency on the previous kernel
    const uint32_t a = run_phase(ph, wave, nwaves, lane, x, s, true);
    if (a == 0x12345678u || (blockIdx.x == 0 && threadIdx.x < 256))
        xout[threadIdx.x] = a + x;
}

// ---- (B) persistent: grid barrier, XCD-hierarchical, bounded spins
struct Bar
{
    uint32_t* xcc; // [8] arrivals per group
    uint32_t* top; // arrivals of group leaders
    uint32_t* gen; // [8] release generation per group
    uint32_t* err;
};

__device__ __forceinline__ bool grid_barrier(const Bar& b, int grp, int per_grp, int ngrp, uint32_t epoch)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0)
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const uint32_t a = __hip_atomic_fetch_add(&b.xcc[grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a == epoch * per_grp - 1)
        {
            const uint32_t t = __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == epoch * ngrp - 1)
                for (int g = 0; g < ngrp; ++g)
                    __hip_atomic_store(&b.gen[g], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        int spins = 0;
        while (__hip_atomic_load(&b.gen[grp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch)
        {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 2000000)
            {
                ok = false;
                atomicExch(b.err, 1u);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void persistent_kernel(const Layer* layers, int nlayers, uint32_t* xbuf /*[2][256]*/, Bar bar,
    int per_grp, int ngrp, int prefetch)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t) blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6), nwaves = (size_t) gridDim.x * (THREADS / 64);
    const int grp = blockIdx.x % ngrp;
    uint32_t epoch = 0;
    Stream s;
    bool have = false;
    int cur = 0;
    if (wave < layers[0].ph[0].bytes / 8192)
    {
        s.request(layers[0].ph[0].w, wave, lane);
        have = true;
    }
    for (int l = 0; l < nlayers; ++l)
        for (int p = 0; p < 5; ++p)
        {
            const Phase ph = layers[l].ph[p];
            const uint32_t x = __builtin_nontemporal_load(&xbuf[cur * 256 + (threadIdx.x & 255)]);
            const uint32_t a = run_phase(ph, wave, nwaves, lane, x, s, have);
            have = false;
            if (a == 0x12345678u || (blockIdx.x == 0 && threadIdx.x < 256))
                xbuf[(cur ^ 1) * 256 + threadIdx.x] = a + x;
            cur ^= 1;
            // next phase's first tile goes out BEFORE the barrier: the HBM pipe keeps streaming while the grid syncs
            const int np = p == 4 ? 0 : p + 1, nl = p == 4 ? l + 1 : l;
            if (prefetch && nl < nlayers && wave < layers[nl].ph[np].bytes / 8192)
            {
                s.request(layers[nl].ph[np].w, wave, lane);
                have = true;
            }
            if (!grid_barrier(bar, grp, per_grp, ngrp, ++epoch))
                return;
        }
}

int main(int argc, char** argv)
{
    const int L = argc > 1 ? atoi(argv[1]) : 16, iters = argc > 2 ? atoi(argv[2]) : 20, wgpc = argc > 3 ? atoi(argv[3]) : 1;
    const size_t mb[5] = {50800000, 8900000, 17800000, 91200000, 45400000};
    size_t per_layer = 0, sz[5];
    for (int p = 0; p < 5; ++p)
    {
        sz[p] = (mb[p] + 8191) / 8192 * 8192;
        per_layer += sz[p];
    }
    char* pool;
    CK(hipMalloc(&pool, per_layer * L));
    CK(hipMemset(pool, 0x5a, per_layer * L));
    std::vector<Layer> hl(L);
    for (int l = 0; l < L; ++l)
    {
        size_t off = 0;
        for (int p = 0; p < 5; ++p)
        {
            hl[l].ph[p].w = pool + per_layer * l + off;
            hl[l].ph[p].bytes = sz[p];
            off += sz[p];
        }
    }
    Layer* dl;
    CK(hipMalloc(&dl, sizeof(Layer) * L));
    CK(hipMemcpy(dl, hl.data(), sizeof(Layer) * L, hipMemcpyHostToDevice));
    uint32_t *xb, *sync;
    CK(hipMalloc(&xb, 4096));
    CK(hipMemset(xb, 1, 4096));
    CK(hipMalloc(&sync, 4096));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms;

    // (A) graph of 5 L launches, 1024 workgroups each
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int l = 0; l < L; ++l)
        for (int p = 0; p < 5; ++p)
            hipLaunchKernelGGL(phase_kernel, dim3(1024), dim3(256), 0, st, hl[l].ph[p], xb + ((l * 5 + p) & 1) * 256,
                xb + (((l * 5 + p) & 1) ^ 1) * 256);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 2; ++rep)
    {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i)
            CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
    }
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us_a = ms * 1e3 / iters / L;
    printf("A  launches (5 per layer, graph)        : %7.2f us/layer  %6.0f GB/s   (between a HIP event pair)\n", us_a, per_layer / us_a / 1e3);
    {
        // the same replays timed by the host clock between two stream synchronisations, NO event recorded next to them: an
        // event pair around graph replays costs ~0.9 us per kernel node on this runtime (bench.py)
        CK(hipStreamSynchronize(st));
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < iters; ++i)
            CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        const double us_h = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters / L;
        printf("A  launches (5 per layer, graph)        : %7.2f us/layer  %6.0f GB/s   (host clock, no events)\n", us_h, per_layer / us_h / 1e3);
    }

    // (A') each phase size on its own: the pure-streaming floor of one launch of that size inside a graph
    for (int p = 0; p < 5; ++p)
    {
        hipGraph_t g2;
        hipGraphExec_t ge2;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int l = 0; l < L; ++l)
            hipLaunchKernelGGL(phase_kernel, dim3(1024), dim3(256), 0, st, hl[l].ph[p], xb + (l & 1) * 256, xb + ((l & 1) ^ 1) * 256);
        CK(hipStreamEndCapture(st, &g2));
        CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
        for (int rep = 0; rep < 2; ++rep)
        {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i)
                CK(hipGraphLaunch(ge2, st));
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
        }
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters / L;
        printf("A' phase %d alone (%6.1f MB)              : %7.2f us/launch %6.0f GB/s\n", p, sz[p] / 1e6, us, sz[p] / us / 1e3);
    }

    // (A' eager) the same launches issued one by one from the host (what the microbench and tllm_session_time_kernel time)
    for (int p = 0; p < 5; ++p)
    {
        for (int rep = 0; rep < 2; ++rep)
        {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i)
                for (int l = 0; l < L; ++l)
                    hipLaunchKernelGGL(phase_kernel, dim3(1024), dim3(256), 0, st, hl[l].ph[p], xb + (l & 1) * 256, xb + ((l & 1) ^ 1) * 256);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
        }
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters / L;
        printf("A' phase %d alone, eager launches         : %7.2f us/launch %6.0f GB/s\n", p, us, sz[p] / us / 1e3);
    }

    // (A'') structural pieces of the real kernel added to the pure stream, phase sizes 0 (50.8 MB) and 4 (45.4 MB)
    for (int mode : {0, 1, 4, 5})
        for (int p : {0, 4})
        {
            hipGraph_t g2;
            hipGraphExec_t ge2;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int l = 0; l < L; ++l)
            {
                BigArgs ba;
                ba.ph = hl[l].ph[p];
                const uint32_t* xi = xb + (l & 1) * 256;
                uint32_t* xo = xb + ((l & 1) ^ 1) * 256;
                const size_t smem = (mode & 1) ? 8192 : 0;
                if (mode == 0)
                    hipLaunchKernelGGL(phase_kernel_x<0>, dim3(1024), dim3(256), smem, st, ba, xi, xo);
                else if (mode == 1)
                    hipLaunchKernelGGL(phase_kernel_x<1>, dim3(1024), dim3(256), smem, st, ba, xi, xo);
                else if (mode == 4)
                    hipLaunchKernelGGL(phase_kernel_x<4>, dim3(1024), dim3(256), smem, st, ba, xi, xo);
                else
                    hipLaunchKernelGGL(phase_kernel_x<5>, dim3(1024), dim3(256), smem, st, ba, xi, xo);
            }
            CK(hipStreamEndCapture(st, &g2));
            CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
            for (int rep = 0; rep < 2; ++rep)
            {
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < iters; ++i)
                    CK(hipGraphLaunch(ge2, st));
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
            }
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / iters / L;
            printf("A'' mode %d (1 = LDS prologue, 4 = +64 VGPRs) phase %d: %7.2f us/launch %6.0f GB/s\n", mode, p, us, sz[p] / us / 1e3);
        }
    if (getenv("PROBE_SKIP_PERSISTENT"))
        return 0;

    // (B) persistent
    for (int pf = 0; pf < 2; ++pf)
        for (int variant = 0; variant < 2; ++variant)
        {
            const int threads = variant == 0 ? 1024 : 256;
            const int blocks = variant == 0 ? 256 * wgpc : 1024;
            const int ngrp = 8, per_grp = blocks / ngrp;
            Bar bar{sync, sync + 64, sync + 128, sync + 192};
            for (int rep = 0; rep < 2; ++rep)
            {
                CK(hipMemsetAsync(sync, 0, 4096, st));
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < iters; ++i)
                {
                    CK(hipMemsetAsync(sync, 0, 1024, st));
                    if (variant == 0)
                        hipLaunchKernelGGL(persistent_kernel<1024>, dim3(blocks), dim3(1024), 0, st, dl, L, xb, bar, per_grp, ngrp, pf);
                    else
                        hipLaunchKernelGGL(persistent_kernel<256>, dim3(blocks), dim3(256), 0, st, dl, L, xb, bar, per_grp, ngrp, pf);
                }
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
            }
            CK(hipEventElapsedTime(&ms, e0, e1));
            uint32_t err = 0;
            CK(hipMemcpy(&err, sync + 192, 4, hipMemcpyDeviceToHost));
            const double us_b = ms * 1e3 / iters / L;
            printf("B  persistent %4d x %4d thr, prefetch %d : %7.2f us/layer  %6.0f GB/s  (x%.3f of A)%s\n", blocks, threads, pf, us_b,
                per_layer / us_b / 1e3, us_b / us_a, err ? "  BARRIER TIMEOUT" : "");
        }
    return 0;
}
