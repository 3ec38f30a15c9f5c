// What the int8 matrix pipe of THIS chip sustains, by instruction, operand data and wave count - the denominator the
// SmoothQuant prefill GEMM (kernels/gemm_glds.hip) is to be read against (BASELINE.json: ">= 60 % of gfx950 int8-MFMA peak").
//
//   build/mfma_probe
//
// One kernel = nothing but MFMAs: every wave keeps NACC independent accumulators and cycles through 4 register-resident
// A / B fragment pairs, `iters` rounds; operands come from a buffer that is either zero, a constant, or random int8 (the chip
// clocks to its power budget - MI355X_MICROARCH.md "DVFS give-back" - and random operands toggle the multiplier array).
// Reported: TOP/s from HIP events over 10 launches, for 1 / 2 waves per SIMD.
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
typedef int i32x16 __attribute__((ext_vector_type(16)));

template <bool BIG, int NACC>
__global__ __launch_bounds__(512) void mfma_only(const i32x4* __restrict__ src, int* __restrict__ sink, int iters)
{
    // 4 A and 4 B fragments per lane (32 VGPRs), loaded once
    i32x4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        a[i] = src[(threadIdx.x + i * 512) & 4095];
        b[i] = src[(threadIdx.x + (i + 4) * 512) & 4095];
    }
    if constexpr (BIG)
    {
        i32x16 acc[NACC];
#pragma unroll
        for (int n = 0; n < NACC; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[n][r] = 0;
        for (int it = 0; it < iters; ++it)
        {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int n = 0; n < NACC; ++n)
                    acc[n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[(k + n) & 3], b[k], acc[n], 0, 0, 0);
        }
        int s = 0;
#pragma unroll
        for (int n = 0; n < NACC; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                s += acc[n][r];
        if (s == 0x7fffffff)
            sink[threadIdx.x] = s;
    }
    else
    {
        i32x4 acc[NACC];
#pragma unroll
        for (int n = 0; n < NACC; ++n)
            acc[n] = i32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
        {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int n = 0; n < NACC; ++n)
                    acc[n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[(k + n) & 3], b[k], acc[n], 0, 0, 0);
        }
        int s = 0;
#pragma unroll
        for (int n = 0; n < NACC; ++n)
            s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
        if (s == 0x7fffffff)
            sink[threadIdx.x] = s;
    }
}

template <bool BIG, int NACC>
double run(const i32x4* src, int* sink, int threads, int blocks, int iters)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w)
        hipLaunchKernelGGL((mfma_only<BIG, NACC>), dim3(blocks), dim3(threads), 0, 0, src, sink, iters);
    CK(hipDeviceSynchronize());
    const int reps = 10;
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((mfma_only<BIG, NACC>), dim3(blocks), dim3(threads), 0, 0, src, sink, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double ops_per_mfma = BIG ? 2.0 * 32 * 32 * 32 : 2.0 * 16 * 16 * 64;
    const double ops = ops_per_mfma * 4.0 * NACC * iters * (threads / 64.0) * blocks * reps;
    return ops / (ms * 1e-3) / 1e12;
}

int main()
{
    int cus = 256;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const size_t n = 4096;
    std::vector<int> host(n * 4);
    i32x4* src;
    int* sink;
    CK(hipMalloc(&src, n * 16));
    CK(hipMalloc(&sink, 4096));
    const char* names[3] = {"zero", "constant 0x01", "random int8"};
    printf("int8 MFMA ceilings on %d CUs (TOP/s; nominal dense peak 5000 at 2.4 GHz)\n", cus);
    printf("%-14s %-12s %10s %10s %10s\n", "operands", "instruction", "4 waves/CU", "8 waves/CU", "16 waves/CU");
    for (int mode = 0; mode < 3; ++mode)
    {
        uint32_t x = 12345;
        for (auto& v : host)
        {
            x ^= x << 13;
            x ^= x >> 17;
            x ^= x << 5;
            v = mode == 0 ? 0 : (mode == 1 ? 0x01010101 : (int) x);
        }
        CK(hipMemcpy(src, host.data(), n * 16, hipMemcpyHostToDevice));
        const int iters = 2048;
        double r[2][3];
        const int threads[3] = {256, 512, 512};
        const int blocks[3] = {cus, cus, cus * 2};
        for (int w = 0; w < 3; ++w)
        {
            r[0][w] = run<true, 6>(src, sink, threads[w], blocks[w], iters);
            r[1][w] = run<false, 8>(src, sink, threads[w], blocks[w], iters);
        }
        printf("%-14s %-12s %10.0f %10.0f %10.0f\n", names[mode], "32x32x32 x6", r[0][0], r[0][1], r[0][2]);
        printf("%-14s %-12s %10.0f %10.0f %10.0f\n", names[mode], "16x16x64 x8", r[1][0], r[1][1], r[1][2]);
    }
    return 0;
}
