// Probe: can the weights of a SMALL decode kernel (the O-projection streams 16.8 MB; the chip's L2s hold 8 x 4 MB) be pulled into
// L2 by the kernel that runs before it (the generation attention, which leaves HBM mostly idle), so that the small kernel runs
// out of L2 instead of HBM?  L2 is per XCD and not shared, so the prefetching workgroup must sit on the XCD of the workgroup
// that will consume the rows: part 1 checks how workgroup ids map to XCDs (and whether that depends on the launch history),
// part 2 measures consume-after-prefetch against a cold consume.
//
//   l2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x)                                                                                                          \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess)                                                                                          \
        {                                                                                                              \
            printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                                          \
            exit(1);                                                                                                   \
        }                                                                                                              \
    } while (0)

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t xcc_id()
{
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

__global__ void xcc_map(uint32_t* out)
{
    if (threadIdx.x == 0)
        out[blockIdx.x] = xcc_id();
}

__global__ void spacer(uint32_t* sink)
{
    if (threadIdx.x == 0 && sink[0] == 0x1234567u)
        sink[1] = 1;
}

constexpr int ROW_BYTES = 4096, ROWS = 4096, ROWS_PER_WG = 8; // the 7B O-projection, int8
constexpr int WGS = ROWS / ROWS_PER_WG;                        // 512 consumer workgroups

// consumer: workgroup g streams rows [8g, 8g + 8) = 32 KB contiguous, 256 threads x 16 B x 8
template <bool NT>
__global__ __launch_bounds__(256) void consume(const char* w, uint32_t* sink)
{
    const u4* p = reinterpret_cast<const u4*>(w + (size_t) blockIdx.x * ROWS_PER_WG * ROW_BYTES);
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
    {
        const u4 v = NT ? __builtin_nontemporal_load(p + i * 256 + threadIdx.x) : p[i * 256 + threadIdx.x];
        acc ^= v.x ^ v.w;
    }
    if (acc == 0x12345678u)
        sink[0] = acc;
}

// prefetcher: npf workgroups; workgroup j (on XCD j % 8 if ids map round-robin) touches the blocks of the consumer workgroups
// g = (j % 8) + 8 m that share its XCD, m strided over the npf / 8 prefetchers of that XCD
__global__ __launch_bounds__(256) void prefetch(const char* w, uint32_t* sink, int use_hw_xcc)
{
    const int npf = gridDim.x;
    const int xcd = use_hw_xcc ? (int) xcc_id() : blockIdx.x % 8;
    const int slot = blockIdx.x / 8, per_xcd = npf / 8;
    uint32_t acc = 0;
    for (int m = slot; m < WGS / 8; m += per_xcd)
    {
        const int g = xcd + 8 * m;
        const u4* p = reinterpret_cast<const u4*>(w + (size_t) g * ROWS_PER_WG * ROW_BYTES);
#pragma unroll
        for (int i = 0; i < 8; ++i)
        {
            const u4 v = p[i * 256 + threadIdx.x];
            acc ^= v.x ^ v.w;
        }
    }
    if (acc == 0x12345678u)
        sink[0] = acc;
}

// stand-in for the attention kernel's own traffic between the prefetch and the consumer: reads `bytes` once
__global__ __launch_bounds__(256) void noise(const char* buf, size_t bytes, uint32_t* sink)
{
    const u4* p = reinterpret_cast<const u4*>(buf);
    const size_t n = bytes / 16;
    uint32_t acc = 0;
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256)
        acc ^= p[i].x;
    if (acc == 0x12345678u)
        sink[0] = acc;
}

int main()
{
    uint32_t *out, *sink;
    CK(hipMalloc(&out, 4096 * 4));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(sink, 0, 64));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    // ---- part 1: workgroup id -> XCD
    std::vector<uint32_t> h(512);
    const int spacers[4] = {0, 3, 225, 1001};
    for (int sp : spacers)
    {
        if (sp)
            hipLaunchKernelGGL(spacer, dim3(sp), dim3(64), 0, st, sink);
        hipLaunchKernelGGL(xcc_map, dim3(512), dim3(256), 0, st, out);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(h.data(), out, 512 * 4, hipMemcpyDeviceToHost));
        int match = 0, cnt[16] = {0};
        for (int i = 0; i < 512; ++i)
        {
            match += h[i] == (uint32_t) (i % 8);
            cnt[h[i] & 15]++;
        }
        printf("after a %4d-workgroup kernel: xcc == wg %% 8 for %d / 512 workgroups; first ids:", sp, match);
        for (int i = 0; i < 12; ++i)
            printf(" %u", h[i]);
        printf(" ; per-XCD counts:");
        for (int i = 0; i < 8; ++i)
            printf(" %d", cnt[i]);
        printf("\n");
    }
    // ---- part 2: consume cold vs consume after prefetch
    const int NC = 40; // copies (672 MB: every use is cold with respect to L2 and the 256 MB L3)
    const size_t wbytes = (size_t) ROWS * ROW_BYTES;
    char* pool;
    CK(hipMalloc(&pool, wbytes * NC));
    CK(hipMemset(pool, 0x5a, wbytes * NC));
    char* nz;
    const size_t nzbytes = 9 << 20;
    CK(hipMalloc(&nz, nzbytes * NC));
    CK(hipMemset(nz, 0x3c, nzbytes * NC));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timed = [&](const char* what, auto&& body) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep)
        {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < NC; ++i)
                body(i);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("%-64s %8.2f us per iteration\n", what, best * 1e3 / NC);
        return best * 1e3f / NC;
    };
    for (int nt = 0; nt < 2; ++nt)
    {
        printf("consumer loads: %s\n", nt ? "non-temporal" : "plain");
        auto C = [&](int i) {
            if (nt)
                hipLaunchKernelGGL(consume<true>, dim3(WGS), dim3(256), 0, st, pool + wbytes * i, sink);
            else
                hipLaunchKernelGGL(consume<false>, dim3(WGS), dim3(256), 0, st, pool + wbytes * i, sink);
        };
        timed("consume (cold)", [&](int i) { C(i); });
        timed("consume twice the same copy (second one may hit)", [&](int i) { C(i); C(i); });
        for (int npf : {256, 512})
            for (int hw : {0, 1})
            {
                char name[128];
                snprintf(name, sizeof name, "prefetch(%d wgs, %s xcd) alone", npf, hw ? "hw" : "id%8");
                const float tp = timed(name, [&](int i) { hipLaunchKernelGGL(prefetch, dim3(npf), dim3(256), 0, st, pool + wbytes * i, sink, hw); });
                snprintf(name, sizeof name, "prefetch(%d wgs, %s xcd) + consume", npf, hw ? "hw" : "id%8");
                const float tpc = timed(name, [&](int i) {
                    hipLaunchKernelGGL(prefetch, dim3(npf), dim3(256), 0, st, pool + wbytes * i, sink, hw);
                    C(i);
                });
                printf("    -> consume after prefetch ~ %.2f us\n", tpc - tp);
            }
        const float tn = timed("noise (9 MB read, 224 wgs) alone", [&](int i) { hipLaunchKernelGGL(noise, dim3(224), dim3(256), 0, st, nz + nzbytes * i, nzbytes, sink); });
        const float tpn = timed("prefetch(256, id%8) + noise", [&](int i) {
            hipLaunchKernelGGL(prefetch, dim3(256), dim3(256), 0, st, pool + wbytes * i, sink, 0);
            hipLaunchKernelGGL(noise, dim3(224), dim3(256), 0, st, nz + nzbytes * i, nzbytes, sink);
        });
        const float tpnc = timed("prefetch(256, id%8) + noise + consume", [&](int i) {
            hipLaunchKernelGGL(prefetch, dim3(256), dim3(256), 0, st, pool + wbytes * i, sink, 0);
            hipLaunchKernelGGL(noise, dim3(224), dim3(256), 0, st, nz + nzbytes * i, nzbytes, sink);
            C(i);
        });
        printf("    -> consume after prefetch + noise ~ %.2f us (noise alone %.2f)\n", tpnc - tpn, tn);
    }
    return 0;
}
