// Probe: does a run-ahead prefetch stream (a second kernel on another HIP stream that touches the NEXT kernels' weights)
// turn the decode step's dependent chain of weight-streaming kernels into Infinity-Cache (256 MiB L3) readers?
//
// A decode step is ~160 dependent launches that each stream 17-90 MB of weights exactly once; HBM idles in every
// launch boundary / prologue / tail.  If reads allocate in the L3, one long-lived loader that walks the step's weights in
// order, a bounded distance ahead of the consumers, keeps HBM busy all the time, and the consumers hit the L3.
//
//   mall_probe [layers=32] [lookahead_MB=96] [mode: 0 = consumers only, 1 = + prefetcher, 2 = both back to back]
//
// Consumers mimic the GEMV access pattern (16-byte non-temporal loads, every byte once, 1024 x 256 threads); sizes are the
// four LLaMA-7B int8 layer weights (QKV 50.3 MB, O 16.8, gate|up 90.2, down 45.1).
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

struct Seg
{
    const char* ptr;
    size_t bytes;
};

// consumer: block 0 publishes "kernel idx started", everyone streams the segment once
template <bool NT>
__global__ __launch_bounds__(256) void consume(Seg s, int idx, unsigned* progress, uint32_t* sink)
{
    if (blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(progress, (unsigned) idx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const size_t n16 = s.bytes / 16;
    const u4* p = reinterpret_cast<const u4*>(s.ptr);
    const size_t per_block = (n16 + gridDim.x - 1) / gridDim.x;
    const size_t b0 = (size_t) blockIdx.x * per_block, b1 = b0 + per_block < n16 ? b0 + per_block : n16;
    uint32_t acc = 0;
    size_t i = b0 + threadIdx.x;
    for (; i + 3 * 256 < b1; i += 4 * 256)
    {
        u4 v0, v1, v2, v3;
        if (NT)
        {
            v0 = __builtin_nontemporal_load(p + i);
            v1 = __builtin_nontemporal_load(p + i + 256);
            v2 = __builtin_nontemporal_load(p + i + 512);
            v3 = __builtin_nontemporal_load(p + i + 768);
        }
        else
        {
            v0 = p[i];
            v1 = p[i + 256];
            v2 = p[i + 512];
            v3 = p[i + 768];
        }
        acc ^= v0.x ^ v1.y ^ v2.z ^ v3.w;
    }
    for (; i < b1; i += 256)
        acc ^= p[i].x;
    if (acc == 0x12345678u)
        sink[0] = acc;
}

// prefetcher: one persistent launch, walks all segments in order, at most `ahead` bytes in front of the consumers
__global__ __launch_bounds__(256) void prefetch(const Seg* segs, const size_t* seg_start, int nseg, size_t ahead,
    const unsigned* progress, uint32_t* sink, int first)
{
    uint32_t acc = 0;
    for (int k = first; k < nseg; ++k)
    {
        // wait until the consumers are close enough: bytes before segment k minus bytes before the running consumer
        if (threadIdx.x == 0)
        {
            int spins = 0;
            while (true)
            {
                const unsigned done = __hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // started kernels
                const size_t consumed = done ? seg_start[done - 1] : 0;
                if (seg_start[k] <= consumed + ahead || ++spins > 2000000)
                    break;
                __builtin_amdgcn_s_sleep(8);
            }
        }
        __syncthreads();
        const Seg s = segs[k];
        const size_t n16 = s.bytes / 16;
        const u4* p = reinterpret_cast<const u4*>(s.ptr);
        const size_t per_block = (n16 + gridDim.x - 1) / gridDim.x;
        const size_t b0 = (size_t) blockIdx.x * per_block, b1 = b0 + per_block < n16 ? b0 + per_block : n16;
        size_t i = b0 + threadIdx.x;
        for (; i + 3 * 256 < b1; i += 4 * 256)
        {
            const u4 v0 = p[i], v1 = p[i + 256], v2 = p[i + 512], v3 = p[i + 768];
            acc ^= v0.x ^ v1.y ^ v2.z ^ v3.w;
        }
        for (; i < b1; i += 256)
            acc ^= p[i].x;
    }
    if (acc == 0x12345678u)
        sink[0] = acc;
}

int main(int argc, char** argv)
{
    const int layers = argc > 1 ? atoi(argv[1]) : 32;
    const size_t ahead = (size_t) (argc > 2 ? atoi(argv[2]) : 96) << 20;
    const int mode = argc > 3 ? atoi(argv[3]) : 2;
    const int pf_blocks = argc > 4 ? atoi(argv[4]) : 256;
    const int first = argc > 5 ? atoi(argv[5]) : 1;
    const size_t sizes[4] = {(size_t) 12288 * 4096, (size_t) 4096 * 4096, (size_t) 22016 * 4096, (size_t) 4096 * 11008};
    std::vector<Seg> segs;
    std::vector<size_t> start;
    size_t total = 0;
    for (int l = 0; l < layers; ++l)
        for (int k = 0; k < 4; ++k)
        {
            char* p;
            CK(hipMalloc(&p, sizes[k]));
            CK(hipMemset(p, 0x11 + k, sizes[k]));
            segs.push_back({p, sizes[k]});
            start.push_back(total);
            total += sizes[k];
        }
    const int nseg = (int) segs.size();
    Seg* dsegs;
    size_t* dstart;
    unsigned* progress;
    uint32_t* sink;
    CK(hipMalloc(&dsegs, nseg * sizeof(Seg)));
    CK(hipMalloc(&dstart, nseg * sizeof(size_t)));
    CK(hipMalloc(&progress, 64));
    CK(hipMalloc(&sink, 64));
    CK(hipMemcpy(dsegs, segs.data(), nseg * sizeof(Seg), hipMemcpyHostToDevice));
    CK(hipMemcpy(dstart, start.data(), nseg * sizeof(size_t), hipMemcpyHostToDevice));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t e0, e1, fork;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventCreate(&fork));
    auto run = [&](bool with_pf, bool nt) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep)
        {
            CK(hipMemsetAsync(progress, 0, 4, sa));
            CK(hipStreamSynchronize(sa));
            CK(hipEventRecord(e0, sa));
            if (with_pf)
            {
                CK(hipEventRecord(fork, sa));
                CK(hipStreamWaitEvent(sb, fork, 0));
                hipLaunchKernelGGL(prefetch, dim3(pf_blocks), dim3(256), 0, sb, dsegs, dstart, nseg, ahead, progress, sink, first);
            }
            for (int k = 0; k < nseg; ++k)
            {
                if (nt)
                    hipLaunchKernelGGL(consume<true>, dim3(1024), dim3(256), 0, sa, segs[k], k, progress, sink);
                else
                    hipLaunchKernelGGL(consume<false>, dim3(1024), dim3(256), 0, sa, segs[k], k, progress, sink);
            }
            CK(hipEventRecord(e1, sa));
            CK(hipEventSynchronize(e1));
            CK(hipStreamSynchronize(sb));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        return best;
    };
    printf("layers %d, %.2f GB per pass, lookahead %zu MB, prefetch blocks %d\n", layers, total / 1e9, ahead >> 20, pf_blocks);
    for (int nt = 1; nt >= 0; --nt)
    {
        if (mode == 0 || mode == 2)
        {
            const float ms = run(false, nt);
            printf("consumers only   (%s loads): %.3f ms  %.2f TB/s\n", nt ? "nt" : "plain", ms, total / ms / 1e9);
        }
        if (mode == 1 || mode == 2)
        {
            const float ms = run(true, nt);
            printf("with prefetcher  (%s loads): %.3f ms  %.2f TB/s\n", nt ? "nt" : "plain", ms, total / ms / 1e9);
        }
    }
    return 0;
}
