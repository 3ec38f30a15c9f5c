// What the matrix pipe keeps of its rate when the same waves also pull their operand fragments out of LDS and keep an
// LDS-DMA stream going - the three things a prefill GEMM main loop does at once (kernels/gemm_sqp.hip, gemm_glds.hip).
//
//   build/lds_mfma_probe
//
// One "iteration" = the work of one wave for one 64-byte k-step of a 64 x 96 wave tile: 24 v_mfma_i32_16x16x64_i8 on 24
// independent accumulators, R ds_read_b128 fragment reads spread between them (R = 10 is what the tile needs: 4 + 6
// fragments; 0 = operands stay in registers), and D LDS-DMA instructions (1 KiB each, source L2-resident; 3.5 per k-step is
// what a 256 x 192 tile needs per wave: 0 / 4 here).  No barriers, no waits except the ones the data dependence needs.
// Reported: int8 TOP/s of the MFMAs alone, for one and two waves per SIMD, random int8 operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                                          \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e = (x);                                                                                            \
        if (e != hipSuccess)                                                                                           \
        {                                                                                                              \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__);                               \
            exit(1);                                                                                                   \
        }                                                                                                              \
    } while (0)

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;

__device__ __forceinline__ void glds16(const void* gptr, uint32_t lds_byte)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_byte) : "memory");
}

template <int R, int D>
__global__ __launch_bounds__(512) void probe(const char* __restrict__ src, int* __restrict__ sink, int iters)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 64 KiB of operand bytes in LDS (random), the DMA target is a second 64 KiB region
    for (int i = tid; i < 4096; i += blockDim.x)
        reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(src)[i];
    __syncthreads();
    // conflict-free fragment addresses: lane -> row lane & 15 (128-byte rows), 16-byte piece (lane >> 4) ^ g(row)
    const int row = lane & 15, j = row >> 1, g = ((j & 1) << 1) | (j & 4);
    const int base = wid * 4096 + row * 128 + ((((lane >> 4)) ^ g) << 4);
    constexpr int NF = R > 0 ? (R < 10 ? 10 : R) : 10;
    i32x4 f[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i)
        f[i] = *reinterpret_cast<const i32x4*>(lds + ((base + i * 2048) & 0xffff));
    i32x4 acc[24];
#pragma unroll
    for (int n = 0; n < 24; ++n)
        acc[n] = i32x4{0, 0, 0, 0};
    const uint32_t lds_base = (uint32_t) (uintptr_t) (lds_void_t*) lds + 65536 + wid * 4096;
    const char* gsrc = src + ((blockIdx.x & 63) * 65536) + wid * 4096 + lane * 16;
    for (int it = 0; it < iters; ++it)
    {
        const int off = (it & 7) * 256;
#pragma unroll
        for (int q = 0; q < 24; ++q)
        {
            acc[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f[q % 6], f[6 + q % 4], acc[q], 0, 0, 0);
            constexpr int STEP = R > 0 ? 24 / (R < 24 ? R : 24) : 1;
            if (R > 0 && q % STEP == 0 && q / STEP < R)
            {
                const int r = q / STEP;
                // the fragment the NEXT iteration multiplies with
                f[r % NF] = *reinterpret_cast<const i32x4*>(lds + ((base + r * 2048 + off) & 0xffff));
            }
            if (D > 0 && (q == 3 || q == 9 || q == 15 || q == 21) && q / 6 < D)
                glds16(gsrc + ((it * 4 + q / 6) & 15) * 1024, lds_base + (q / 6) * 1024);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (D > 0)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D * 2) : "memory"); // two iterations of DMA stay in flight
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int s = 0;
#pragma unroll
    for (int n = 0; n < 24; ++n)
        s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    if (s == 0x7fffffff)
        sink[threadIdx.x] = s;
}

template <int R, int D>
double run(const char* src, int* sink, int threads, int blocks, int iters)
{
    auto k = probe<R, D>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t smem = 65536 + 8 * 4096;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w)
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), smem, 0, src, sink, iters);
    CK(hipDeviceSynchronize());
    const int reps = 5;
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), smem, 0, src, sink, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double ops = 2.0 * 16 * 16 * 64 * 24.0 * iters * (threads / 64.0) * blocks * reps;
    return ops / (ms * 1e-3) / 1e12;
}

int main()
{
    int cus = 256;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const size_t bytes = 64 * 65536 + 65536;
    std::vector<int> host(bytes / 4);
    uint32_t x = 12345;
    for (auto& v : host)
    {
        x ^= x << 13;
        x ^= x >> 17;
        x ^= x << 5;
        v = (int) x;
    }
    char* src;
    int* sink;
    CK(hipMalloc(&src, bytes));
    CK(hipMalloc(&sink, 4096));
    CK(hipMemcpy(src, host.data(), bytes, hipMemcpyHostToDevice));
    const int iters = 1024;
    printf("MFMA rate (int8 TOP/s, 16x16x64, random operands, %d CUs) next to fragment reads and LDS-DMA in the same waves\n", cus);
    printf("per 24 MFMAs:                                  4 waves/CU   8 waves/CU\n");
#define ROW(R, D, label)                                                                                               \
    printf("%-46s %10.0f   %10.0f\n", label, run<R, D>(src, sink, 256, cus, iters), run<R, D>(src, sink, 512, cus, iters))
    ROW(0, 0, "MFMAs only");
    ROW(5, 0, "+ 5 ds_read_b128");
    ROW(10, 0, "+ 10 ds_read_b128 (the 64 x 96 wave tile)");
    ROW(20, 0, "+ 20 ds_read_b128");
    ROW(0, 4, "+ 4 LDS-DMA (1 KiB each)");
    ROW(10, 4, "+ 10 ds_read_b128 + 4 LDS-DMA");
    ROW(7, 4, "+ 7 ds_read_b128 + 4 LDS-DMA (a 128 x 96 tile)");
    return 0;
}
