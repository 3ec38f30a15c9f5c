// Kernel sweep harness (GPU box only): times the fused skinny GEMM of the generation step on the LLaMA-7B shapes
// through the C ABI, rotating over enough weight copies to defeat the 256 MiB Infinity Cache.
//   build/microbench 0 <blocks_per_cu|0> [<case filter> | gemm [M [cfg_lo [cfg_hi]]]]
#include "../../../include/tllm_runtime_api.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

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

struct Case
{
    const char* name;
    int wtype, pro, epi, out_dtype, N, K;
};

// xorshift-style hash per 32-bit word; mode 1 = two fp16 values with exponent clamped to [2^-3, 2) (sign random)
__global__ void fill_rand(uint32_t* p, size_t n, uint32_t seed, int fp16_mode)
{
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
    {
        uint32_t x = (uint32_t) i * 2654435761u ^ seed;
        x ^= x >> 16;
        x *= 0x7feb352du;
        x ^= x >> 15;
        x *= 0x846ca68bu;
        x ^= x >> 16;
        if (fp16_mode)
            x = (x & 0x83ff83ffu) | 0x30003000u | ((x >> 3) & 0x0c000c00u); // exponent field 12..15 -> 2^-3 .. 2^0
        p[i] = x;
    }
}

int main(int argc, char** argv)
{
    // argv[1] is unused (kept so that older command lines keep their positions)
    if (argc > 2)
        tllm_gemv_set_blocks_per_cu(atoi(argv[2]));
    const char* only = argc > 3 ? argv[3] : nullptr;
    initLibNvInferPlugins(nullptr, "tensorrt_llm");
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const int D = 4096, I = 11008, V = 32000;
    // wtype: 0 fp16 1 woq8 2 woq4 3 sq ; pro: 0 none 1 rms 2 rms+qs 4 qs ; epi: 0 none 1 res 2 swiglu 3 swiglu+q
    std::vector<Case> cases = {
        {"sq   qkv    rms+q      ", 3, 2, 0, 1, 3 * D, D},
        {"sq   o      q  +res    ", 3, 4, 1, 1, D, D},
        {"sq   gateup rms+q swi+q", 3, 2, 3, 2, I, D},
        {"sq   down   none +res  ", 3, 0, 1, 1, D, I},
        {"woq8 qkv    rms        ", 1, 1, 0, 1, 3 * D, D},
        {"woq8 o      +res       ", 1, 0, 1, 1, D, D},
        {"woq8 gateup rms swi    ", 1, 1, 2, 1, I, D},
        {"woq8 down   +res       ", 1, 0, 1, 1, D, I},
        {"woq4 qkv    rms        ", 2, 1, 0, 1, 3 * D, D},
        {"woq4 gateup rms swi    ", 2, 1, 2, 1, I, D},
        {"woq4 down   +res       ", 2, 0, 1, 1, D, I},
        {"fp16 qkv    rms        ", 0, 1, 0, 1, 3 * D, D},
        {"fp16 o      +res       ", 0, 0, 1, 1, D, D},
        {"fp16 gateup rms swi    ", 0, 1, 2, 1, I, D},
        {"fp16 down   +res       ", 0, 0, 1, 1, D, I},
        {"fp16 head   rms f32out ", 0, 1, 0, 0, V, D},
        {"x sq N65536 K4096 rmsq ", 3, 2, 0, 1, 65536, D},
        {"x sq N6144  K8192 rmsq ", 3, 2, 0, 1, 6144, 8192},
        {"x sq N12288 K4096 none ", 3, 0, 0, 1, 3 * D, D},
        {"x sq gateup none swi+q ", 3, 0, 3, 2, I, D}, // (against "sq gateup rms+q swi+q": what the RMSNorm + quantiser prologue costs)
        {"x sq N22016 K4096 rmsq ", 3, 2, 0, 1, 2 * I, D},
        {"x sq N2048  K4096 rmsq ", 3, 2, 0, 1, 2048, D},
        {"x sq N256   K4096 rmsq ", 3, 2, 0, 1, 256, D},
        {"y sq  N256 none none   ", 3, 0, 0, 1, 256, D},
        {"y sq  N264 rmsq none   ", 3, 2, 0, 1, 264, D},
        {"y sq  N272 q    res    ", 3, 4, 1, 1, 272, D},
        {"y fp  N280 none none   ", 0, 0, 0, 1, 280, D},
        {"y fp  N288 rms  none   ", 0, 1, 0, 1, 288, D},
        {"y fp  N296 rms  res    ", 0, 1, 1, 1, 296, D},
        {"y w8  N304 rms  none   ", 1, 1, 0, 1, 304, D},
        {"x fp N12288 K4096 rms  ", 0, 1, 0, 1, 3 * D, D},
        {"x fp N6144  K4096 rms  ", 0, 1, 0, 1, 6144, D},
        // per-rank extents under tensor parallelism: QKV / gate|up split their rows, O / down their K
        {"tp2 sq qkv   rms+q     ", 3, 2, 0, 1, 3 * D / 2, D},
        {"tp2 sq o     q +res    ", 3, 4, 1, 1, D, D / 2},
        {"tp2 sq gateup rmsq swiq", 3, 2, 3, 2, I / 2, D},
        {"tp2 sq down  none +res ", 3, 0, 1, 1, D, I / 2},
        {"tp4 sq qkv   rms+q     ", 3, 2, 0, 1, 3 * D / 4, D},
        {"tp4 sq o     q +res    ", 3, 4, 1, 1, D, D / 4},
        {"tp4 sq gateup rmsq swiq", 3, 2, 3, 2, I / 4, D},
        {"tp4 sq down  none +res ", 3, 0, 1, 1, D, I / 4},
        {"tp8 sq qkv   rms+q     ", 3, 2, 0, 1, 3 * D / 8, D},
        {"tp8 sq o     q +res    ", 3, 4, 1, 1, D, D / 8},
        {"tp8 sq gateup rmsq swiq", 3, 2, 3, 2, I / 8, D},
        {"tp8 sq down  none +res ", 3, 0, 1, 1, D, I / 8},
    };
    void *x, *gamma, *res, *y, *scales, *fs;
    CK(hipMalloc(&x, 65536));
    CK(hipMalloc(&gamma, 65536));
    CK(hipMalloc(&res, 1 << 20));
    CK(hipMalloc(&y, 1 << 20));
    CK(hipMalloc(&scales, 1 << 20));
    CK(hipMalloc(&fs, 256));
    CK(hipMemset(x, 0x11, 65536));
    CK(hipMemset(gamma, 0x3c, 65536));
    CK(hipMemset(res, 0, 1 << 20));
    CK(hipMemset(scales, 0x11, 1 << 20));
    float one[4] = {1.f, 1.f, 1.f, 1.f};
    CK(hipMemcpy(fs, one, 16, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t pool_bytes = (size_t) 1200 << 20;
    char* pool;
    CK(hipMalloc(&pool, pool_bytes));
    CK(hipMemset(pool, 0x37, pool_bytes));
    if (only && !strcmp(only, "gemm"))
    {
        // prefill-shaped GEMMs: M = 1024, the four LLaMA-7B layer shapes
        struct GC { const char* name; int wtype, N, K; };
        std::vector<GC> gcs = {{"sq   qkv ", 3, 3 * D, D}, {"sq   o   ", 3, D, D}, {"sq   fc  ", 3, I, D}, {"sq   fc|g", 3, 2 * I, D},
                               {"sq   down", 3, D, I}, {"fp16 qkv ", 0, 3 * D, D}, {"fp16 fc  ", 0, I, D}, {"fp16 down", 0, D, I},
                               {"woq8 qkv ", 1, 3 * D, D}, {"woq4 qkv ", 2, 3 * D, D}};
        const int Mg = argc > 4 ? atoi(argv[4]) : 1024;
        void *a, *c;
        CK(hipMalloc(&a, (size_t) Mg * I * 2));
        CK(hipMemset(a, 0x3c, (size_t) Mg * I * 2));
        CK(hipMalloc(&c, (size_t) Mg * 2 * I * 4));
        // uniform random operand bytes (power / clocks depend on the data: zero or constant operands run ~19 % faster,
        // MI355X_MICROARCH.md; quote the random-data number).  fp16 operands: random mantissa, exponent kept small.
        hipLaunchKernelGGL(fill_rand, dim3(4096), dim3(256), 0, st, (uint32_t*) pool, pool_bytes / 4, 0x9e3779b9u, 0);
        hipLaunchKernelGGL(fill_rand, dim3(4096), dim3(256), 0, st, (uint32_t*) a, (size_t) Mg * I * 2 / 4, 0x85ebca6bu, 0);
        const int cfg_lo = argc > 5 ? atoi(argv[5]) : 0, cfg_hi = argc > 6 ? atoi(argv[6]) : cfg_lo;
        printf("%-12s %4s %6s %9s %10s %9s\n", "gemm", "cfg", "M", "us", "TOP/s", "frac_peak");
        for (int cfg = cfg_lo; cfg <= cfg_hi; ++cfg)
        for (auto& g : gcs)
        {
            if (cfg > 0 && g.wtype != 3 && g.wtype != 0)
                continue;
            tllm_gemm_set_tile_cfg(cfg | (getenv("MB_GEMM_DBG") ? atoi(getenv("MB_GEMM_DBG")) << 8 : 0));
            if (g.wtype == 0 || g.wtype == 1 || g.wtype == 2) // fp16 activations: keep exponents sane (|x| < 2)
                hipLaunchKernelGGL(fill_rand, dim3(4096), dim3(256), 0, st, (uint32_t*) a, (size_t) Mg * I * 2 / 4, 0x85ebca6bu, 1);
            else
                hipLaunchKernelGGL(fill_rand, dim3(4096), dim3(256), 0, st, (uint32_t*) a, (size_t) Mg * I * 2 / 4, 0x85ebca6bu, 0);
            if (g.wtype == 0)
                hipLaunchKernelGGL(fill_rand, dim3(4096), dim3(256), 0, st, (uint32_t*) pool, (size_t) g.N * g.K * 2 / 4, 0x9e3779b9u, 1);
            else
                hipLaunchKernelGGL(fill_rand, dim3(4096), dim3(256), 0, st, (uint32_t*) pool, (size_t) g.N * g.K / 4, 0x9e3779b9u, 0);
            tllm_gemm_params_t q;
            memset(&q, 0, sizeof(q));
            q.wtype = g.wtype;
            q.out_dtype = 1;
            q.M = Mg;
            q.N = g.N;
            q.K = g.K;
            q.a = a;
            q.lda = g.K;
            q.w = pool;
            q.ldw = g.wtype == 0 ? (int64_t) g.K * 2 : (g.wtype == 2 ? g.K / 2 : g.K);
            q.scale_col = scales;
            q.scale_row = (const float*) fs;
            q.per_channel = 1;
            q.c = c;
            q.ldc = g.N;
            const int iters = 20;
            for (int rep = 0; rep < 2; ++rep)
            {
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < iters; ++i)
                    if (tllm_gemm(&q, (tllm_stream_t) st))
                    {
                        printf("%s: %s\n", g.name, tllm_last_error());
                        return 1;
                    }
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
            }
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / iters;
            const double tops = 2.0 * Mg * g.N * g.K / us / 1e6;
            const double peak = g.wtype == 3 ? 5000.0 : 2500.0; // dense int8 / fp16 MFMA peaks, TOP/s (MI355X_MICROARCH.md)
            printf("%-12s %4d %6d %9.2f %10.1f %9.3f\n", g.name, cfg, Mg, us, tops, tops / peak);
        }
        return 0;
    }
    printf("%-26s %8s %9s %9s %8s\n", "case", "MB", "us", "GB/s", "frac8T");
    for (auto& c : cases)
    {
        if (only && !strstr(c.name, only))
            continue;
        const bool swi = c.epi >= 2;
        const int64_t ldw = c.wtype == 0 ? (int64_t) c.K * 2 : (c.wtype == 2 ? c.K / 2 : c.K);
        const int64_t rows = swi ? 2 * c.N : c.N;
        const size_t wbytes = (size_t) rows * ldw;
        int ncopy = (int) (pool_bytes / wbytes);
        if (getenv("MB_NCOPY"))
            ncopy = std::min(ncopy, atoi(getenv("MB_NCOPY")));
        tllm_gemv_params_t p;
        memset(&p, 0, sizeof(p));
        p.wtype = c.wtype;
        p.pro = c.pro;
        p.epi = c.epi;
        p.out_dtype = c.out_dtype;
        p.M = 1;
        p.N = c.N;
        p.K = c.K;
        p.x = x;
        p.ldx = c.K;
        p.ldw = ldw;
        p.scale_col = scales;
        p.scale_row = (const float*) fs;
        p.per_channel = 1;
        p.gamma = gamma;
        p.eps = 1e-6f;
        p.act_scale = (const float*) fs;
        p.residual = res;
        p.epi_scale = (const float*) fs;
        p.y = y;
        p.ldy = c.N;
        const int iters = 60;
        for (int rep = 0; rep < 2; ++rep)
        {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i)
            {
                p.w = pool + (size_t) (i % ncopy) * wbytes;
                if (tllm_gemv(&p, (tllm_stream_t) st))
                {
                    printf("%s: %s\n", c.name, tllm_last_error());
                    return 1;
                }
            }
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
        }
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters;
        const double gbs = wbytes / us / 1e3;
        printf("%-26s %8.1f %9.2f %9.1f %8.3f\n", c.name, wbytes / 1048576.0, us, gbs, gbs / 8000.0);
    }
    return 0;
}
