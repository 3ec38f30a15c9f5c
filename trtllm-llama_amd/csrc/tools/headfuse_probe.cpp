// Feasibility probe (GPU box only): what would fusing the decode step's QKV GEMV with the attention of the SAME head buy?
//
// The only seam of the decode layer that is not all-to-all: head h's 384 QKV rows (1.5 MB of int8 weights) ->
// RoPE -> KV append -> split-KV attention of head h.  Candidate structure B: 32 heads x 8 workgroups = 256 workgroups (one per CU, 8
// waves); a workgroup streams 48 of its head's rows, publishes its 48 outputs as 8-byte {tag, value} granules (R2 of the guide's
// Guideline 16: the data is the flag, sc1 stores, relaxed agent-scope sweep), one wave sweeps the head's 192 granules, then the
// workgroup runs its 1/8 of the head's KV range - whose rows it requested at t = 0, so that round trip hides under the weight
// stream.  Against A = what runs today: the QKV GEMV launch (1024 x 256 threads) followed by the attention launch (224 x 256).
// Both sit in the real chain (O-projection, gate|up, down-projection launches behind them), 32 layers, one hipGraph per variant.
// Nothing of numerical interest is computed: streams are consumed by integer adds, the attention arithmetic is a dependent FMA
// chain sized per lane group (12 cache rows in A, 5 in B) - this measures the launch / hand-off structure only.
//   build/headfuse_probe [layers=32] [replays=30]
#include <hip/hip_runtime.h>
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
typedef __attribute__((address_space(1))) unsigned long long gu64;

__device__ __forceinline__ u4 ldnt(const void* p)
{
    return __builtin_nontemporal_load(reinterpret_cast<const u4*>(p));
}

struct Tile
{
    u4 buf[8]; // 8 KiB per wave
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

// the GEMV prologue's shape: block reduction of x through LDS, normalised copy back (2 barriers)
__device__ __forceinline__ uint32_t prologue(uint32_t x, uint32_t* lds, int nwaves)
{
    const int lane = threadIdx.x & 63;
    uint32_t v = x * x;
    for (int o = 32; o; o >>= 1)
        v += __shfl_xor(v, o, 64);
    if (lane == 0)
        lds[threadIdx.x >> 6] = v;
    __syncthreads();
    v = 0;
    for (int i = 0; i < nwaves; ++i)
        v += lds[i];
    lds[16 + threadIdx.x] = x ^ v;
    __syncthreads();
    return lds[16 + ((threadIdx.x * 7) % blockDim.x)];
}

// stand-in for the scores / softmax / PV arithmetic of `rows` cache rows per lane group: ~115 VALU per row, 4-way ILP
__device__ __forceinline__ float attn_math(const u4* kv, int nkv, int rows, float q)
{
    float a0 = q, a1 = q * 0.5f, a2 = q * 0.25f, a3 = 1.f;
    for (int r = 0; r < rows; ++r)
    {
        const u4 v = kv[r % nkv];
#pragma unroll
        for (int i = 0; i < 7; ++i)
        {
            a0 = __builtin_fmaf(a0, 0.999f, (float) (v.x >> i));
            a1 = __builtin_fmaf(a1, 0.998f, (float) (v.y >> i));
            a2 = __builtin_fmaf(a2, 0.997f, (float) (v.z >> i));
            a3 = __builtin_fmaf(a3, 0.996f, (float) (v.w >> i));
            a0 += __shfl_xor(a1, 1 << (i & 3), 64); // the lane-group dot-product reductions
            a2 += a3 * a0;
            a1 = __builtin_fmaf(a2, 1e-3f, a1);
            a3 = __builtin_fmaf(a0, 1e-3f, a3);
        }
    }
    return a0 + a1 + a2 + a3;
}

// ---- A1: streaming GEMV launch (persistent grid, 256 threads)
__global__ __launch_bounds__(256) void stream_kernel(const char* w, size_t bytes, const uint32_t* xin, uint32_t* xout)
{
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t) blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t) gridDim.x * 4, ntiles = bytes / 8192;
    Tile s;
    size_t t = wave;
    if (t < ntiles)
        s.request(w, t, lane);
    const uint32_t x = prologue(xin[threadIdx.x], lds, 4);
    uint32_t acc = 0;
    while (t < ntiles)
    {
        Tile nxt;
        const size_t tn = t + nwaves;
        if (tn < ntiles)
            nxt.request(w, tn, lane);
        acc += s.consume(x);
        s = nxt;
        t = tn;
    }
    for (int o = 32; o; o >>= 1)
        acc += __shfl_xor(acc, o, 64);
    if (lane == 0)
        xout[wave & 4095] = acc;
}

// ---- A2: the attention launch: 224 workgroups x 256 threads, 12 rows of K and of V per lane group requested at t = 0
__global__ __launch_bounds__(256) void attn_kernel(const char* kv, size_t bytes, const uint32_t* qkv, float* part)
{
    __shared__ float sm[16][132];
    const size_t per_wg = bytes / gridDim.x / 16 * 16;
    const char* base = kv + (size_t) blockIdx.x * per_wg;
    constexpr int NV = 10; // 40 KB per workgroup = 160 B per thread
    u4 r[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i)
        r[i] = *reinterpret_cast<const u4*>(base + ((size_t) i * 256 + threadIdx.x) * 16 % per_wg);
    const float q = (float) qkv[threadIdx.x & 127];
    const float o = attn_math(r, NV, 12, q);
    const int gid = threadIdx.x >> 4;
    sm[gid][threadIdx.x & 15] = o;
    __syncthreads();
    float m = 0.f;
    for (int g = 0; g < 16; ++g)
        m += sm[g][threadIdx.x & 15];
    __syncthreads();
    if (threadIdx.x < 130)
        part[(size_t) blockIdx.x * 130 + threadIdx.x] = m;
}

// ---- B: one launch: 32 heads x 8 workgroups of 8 waves; head = wg % 32 (the group shares an XCD: wg % 8 is the same for all)
struct FusedArgs
{
    const char* w;      // this layer's QKV weights
    size_t wbytes;
    const char* kv;
    size_t kvbytes;
    const uint32_t* xin;
    unsigned long long* xchg; // [32 heads][192 granules]
    const uint32_t* epoch;
    float* part;
    uint32_t* timeout;
};
__global__ __launch_bounds__(512) void fused_kernel(FusedArgs a)
{
    __shared__ uint32_t lds[16 + 512];
    __shared__ float sm[32][20];
    __shared__ uint32_t qkv_s[192];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int head = blockIdx.x & 31, member = blockIdx.x >> 5;
    // t = 0: this workgroup's 1/8 of the head's KV range (34 KB: 5 x 16 B per thread, the last partly redundant) ...
    const size_t kv_per_wg = a.kvbytes / 256 / 16 * 16;
    const char* kvb = a.kv + (size_t) blockIdx.x * kv_per_wg;
    constexpr int NV = 5;
    u4 kvr[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i)
        kvr[i] = *reinterpret_cast<const u4*>(kvb + ((size_t) i * 512 + threadIdx.x) * 16 % kv_per_wg);
    // ... and the first weight tile of every wave: 48 rows x 4 KB = 24 tiles of 8 KB per workgroup, 3 per wave
    const size_t tiles_per_wg = a.wbytes / 8192 / 256;
    const size_t t0 = (size_t) blockIdx.x * tiles_per_wg;
    Tile s;
    size_t t = wv;
    if (t < tiles_per_wg)
        s.request(a.w, t0 + t, lane);
    const uint32_t ep = *a.epoch;
    const uint32_t x = prologue(a.xin[threadIdx.x & 255], lds, 8);
    uint32_t acc = 0;
    while (t < tiles_per_wg)
    {
        Tile nxt;
        const size_t tn = t + 8;
        if (tn < tiles_per_wg)
            nxt.request(a.w, t0 + tn, lane);
        acc += s.consume(x);
        s = nxt;
        t = tn;
    }
    for (int o = 32; o; o >>= 1)
        acc += __shfl_xor(acc, o, 64);
    // publish: 48 outputs per workgroup = 24 granules {tag, 2 x fp16}; wave wv writes 3 of them (lanes 0..2)
    gu64* gx = (gu64*) (a.xchg + (size_t) head * 192);
    if (lane < 3)
        __hip_atomic_store(gx + member * 24 + wv * 3 + lane, ((unsigned long long) ep << 32) | (acc & 0xffffffffu), __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_AGENT);
    // gather: wave 0 sweeps the head's 192 granules until every tag is this step's
    if (wv == 0)
    {
        unsigned spins = 0;
        for (;;)
        {
            bool ok = true;
            uint32_t v[3];
#pragma unroll
            for (int k = 0; k < 3; ++k)
            {
                const unsigned long long g = __hip_atomic_load(gx + k * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v[k] = (uint32_t) g;
                ok &= (uint32_t) (g >> 32) == ep;
            }
            if (__all(ok))
            {
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    qkv_s[k * 64 + lane] = v[k];
                break;
            }
            if (++spins > 2000000u)
            {
                if (lane == 0)
                    atomicExch(a.timeout, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    const float q = (float) qkv_s[threadIdx.x & 127];
    const float o = attn_math(kvr, NV, 5, q);
    const int gid = threadIdx.x >> 4;
    sm[gid][threadIdx.x & 15] = o;
    __syncthreads();
    float m = 0.f;
    for (int g = 0; g < 32; ++g)
        m += sm[g][threadIdx.x & 15];
    if (threadIdx.x < 130)
        a.part[(size_t) blockIdx.x * 130 + threadIdx.x] = m;
}

// ---- B4: the same on 4 waves per workgroup (the real GEMV's block size): 12 rows per wave as 6 double-buffered tiles; the KV range
// split as today (192-token chunks: 6 of the 8 workgroups of a head hold one at a 1073-slot cache), 12 rows per lane group
__global__ __launch_bounds__(256) void fused4_kernel(FusedArgs a)
{
    __shared__ uint32_t lds[16 + 256];
    __shared__ float sm[16][20];
    __shared__ uint32_t qkv_s[192];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int head = blockIdx.x & 31, member = blockIdx.x >> 5;
    const bool has_kv = member < 6;
    const size_t kv_per_wg = a.kvbytes / 192 / 16 * 16; // 32 heads x 6 splits
    const char* kvb = a.kv + (size_t) (head * 6 + (has_kv ? member : 0)) * kv_per_wg;
    constexpr int NV = 12; // 48 KB per workgroup = 192 B per thread
    u4 kvr[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i)
        kvr[i] = *reinterpret_cast<const u4*>(kvb + ((size_t) i * 256 + threadIdx.x) * 16 % kv_per_wg);
    const size_t tiles_per_wg = a.wbytes / 8192 / 256;
    const size_t t0 = (size_t) blockIdx.x * tiles_per_wg;
    Tile s;
    size_t t = wv;
    if (t < tiles_per_wg)
        s.request(a.w, t0 + t, lane);
    const uint32_t ep = *a.epoch;
    const uint32_t x = prologue(a.xin[threadIdx.x], lds, 4);
    uint32_t acc = 0;
    while (t < tiles_per_wg)
    {
        Tile nxt;
        const size_t tn = t + 4;
        if (tn < tiles_per_wg)
            nxt.request(a.w, t0 + tn, lane);
        acc += s.consume(x);
        s = nxt;
        t = tn;
    }
    for (int o = 32; o; o >>= 1)
        acc += __shfl_xor(acc, o, 64);
    gu64* gx = (gu64*) (a.xchg + (size_t) head * 192);
    if (lane < 6)
        __hip_atomic_store(gx + member * 24 + wv * 6 + lane, ((unsigned long long) ep << 32) | (acc & 0xffffffffu), __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_AGENT);
    if (!has_kv) // block-uniform: nothing to attend to
        return;
    if (wv == 0)
    {
        unsigned spins = 0;
        for (;;)
        {
            bool ok = true;
            uint32_t v[3];
#pragma unroll
            for (int k = 0; k < 3; ++k)
            {
                const unsigned long long g = __hip_atomic_load(gx + k * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v[k] = (uint32_t) g;
                ok &= (uint32_t) (g >> 32) == ep;
            }
            if (__all(ok))
            {
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    qkv_s[k * 64 + lane] = v[k];
                break;
            }
            if (++spins > 2000000u)
            {
                if (lane == 0)
                    atomicExch(a.timeout, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    const float q = (float) qkv_s[threadIdx.x & 127];
    const float o = attn_math(kvr, NV, 12, q);
    const int gid = threadIdx.x >> 4;
    sm[gid][threadIdx.x & 15] = o;
    __syncthreads();
    float m = 0.f;
    for (int g = 0; g < 16; ++g)
        m += sm[g][threadIdx.x & 15];
    if (threadIdx.x < 130)
        a.part[(size_t) blockIdx.x * 130 + threadIdx.x] = m;
}

__global__ void bump_kernel(uint32_t* epoch)
{
    *epoch += 1;
}

int main(int argc, char** argv)
{
    const int L = argc > 1 ? atoi(argv[1]) : 32, replays = argc > 2 ? atoi(argv[2]) : 30;
    const size_t sizes[5] = {50331648 + 8192 * 60, 8912896, 17825792, 91226112, 45400064 / 8192 * 8192}; // QKV, KV, O, gate|up, down
    std::vector<char*> bufs((size_t) L * 5);
    for (int l = 0; l < L; ++l)
        for (int k = 0; k < 5; ++k)
        {
            CK(hipMalloc(reinterpret_cast<void**>(&bufs[(size_t) l * 5 + k]), sizes[k] + 65536));
            CK(hipMemset(bufs[(size_t) l * 5 + k], 0x5a + l + k, sizes[k] + 65536));
        }
    uint32_t *x0, *x1, *epoch, *timeout;
    float* part;
    unsigned long long* xchg;
    CK(hipMalloc(reinterpret_cast<void**>(&x0), 65536));
    CK(hipMalloc(reinterpret_cast<void**>(&x1), 65536));
    CK(hipMemset(x0, 1, 65536));
    CK(hipMemset(x1, 2, 65536));
    CK(hipMalloc(reinterpret_cast<void**>(&epoch), 256));
    CK(hipMalloc(reinterpret_cast<void**>(&timeout), 256));
    CK(hipMemset(timeout, 0, 256));
    const uint32_t one = 1;
    CK(hipMemcpy(epoch, &one, 4, hipMemcpyHostToDevice));
    CK(hipMalloc(reinterpret_cast<void**>(&part), 1 << 20));
    CK(hipMalloc(reinterpret_cast<void**>(&xchg), (size_t) L * 32 * 192 * 8));
    CK(hipMemset(xchg, 0, (size_t) L * 32 * 192 * 8));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const size_t lds = (16 + 256) * 4;
    auto rest = [&](int l) { // O-projection, gate|up, down-projection launches (grids of the real kernels)
        hipLaunchKernelGGL(stream_kernel, dim3(768), dim3(256), lds, st, bufs[(size_t) l * 5 + 2], sizes[2], x1, x0);
        hipLaunchKernelGGL(stream_kernel, dim3(918), dim3(256), lds, st, bufs[(size_t) l * 5 + 3], sizes[3], x0, x1);
        hipLaunchKernelGGL(stream_kernel, dim3(1024), dim3(256), lds, st, bufs[(size_t) l * 5 + 4], sizes[4], x1, x0);
    };
    hipGraphExec_t gexec[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int variant = 0; variant < 4; ++variant)
    {
        hipGraph_t g;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int l = 0; l < L; ++l)
        {
            if (variant == 0)
            {
                hipLaunchKernelGGL(stream_kernel, dim3(1024), dim3(256), lds, st, bufs[(size_t) l * 5], sizes[0] / 8192 * 8192, x0, x1);
                hipLaunchKernelGGL(attn_kernel, dim3(224), dim3(256), 0, st, bufs[(size_t) l * 5 + 1], sizes[1], x1, part);
            }
            else if (variant == 1)
            {
                FusedArgs a{bufs[(size_t) l * 5], sizes[0] / (8192 * 256) * (8192 * 256), bufs[(size_t) l * 5 + 1], sizes[1], x0,
                    xchg + (size_t) l * 32 * 192, epoch, part, timeout};
                hipLaunchKernelGGL(fused_kernel, dim3(256), dim3(512), 0, st, a);
            }
            else if (variant == 3)
            {
                FusedArgs a{bufs[(size_t) l * 5], sizes[0] / (8192 * 256) * (8192 * 256), bufs[(size_t) l * 5 + 1], sizes[1], x0,
                    xchg + (size_t) l * 32 * 192, epoch, part, timeout};
                hipLaunchKernelGGL(fused4_kernel, dim3(256), dim3(256), 0, st, a);
            }
            else // the same two launches without anything behind the hand-off: QKV stream only (what the boundary alone costs)
                hipLaunchKernelGGL(stream_kernel, dim3(1024), dim3(256), lds, st, bufs[(size_t) l * 5], sizes[0] / 8192 * 8192, x0, x1);
            rest(l);
        }
        hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(1), 0, st, epoch);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&gexec[variant], g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char* names[4] = {"A: QKV launch + attention launch", "B: one launch per head group (fused, 8 waves)", "C: QKV launch only (no attention at all)",
        "B4: fused on 4 waves, 192-token splits"};
    float best[4] = {1e30f, 1e30f, 1e30f, 1e30f};
    for (int round = 0; round < 4; ++round) // interleaved rounds: one box, one clock state
        for (int variant = 0; variant < 4; ++variant)
        {
            for (int i = 0; i < 3; ++i)
                CK(hipGraphLaunch(gexec[variant], st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < replays; ++i)
                CK(hipGraphLaunch(gexec[variant], st));
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const float us_layer = ms * 1000.f / replays / L;
            if (us_layer < best[variant])
                best[variant] = us_layer;
            printf("round %d  %-45s %8.2f us per layer\n", round, names[variant], us_layer);
        }
    uint32_t to = 0;
    CK(hipMemcpy(&to, timeout, 4, hipMemcpyDeviceToHost));
    printf("\nbest of 4 rounds, per layer (5 / 4 / 4 launches):  A %.2f us   B %.2f us   C %.2f us   ->  B - A = %+.2f us,  A - C = %.2f us "
           "(what the attention launch costs today);  B4 %.2f us -> B4 - A = %+.2f us%s\n", best[0], best[1], best[2], best[1] - best[0], best[0] - best[2],
        best[3], best[3] - best[0], to ? "   [a granule sweep TIMED OUT]" : "");
    return 0;
}
