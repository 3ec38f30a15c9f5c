// MFMA GEMM for prefill-shaped problems, LDS-DMA staged:  C[m,n] = epi( sum_k A[m,k] * W[n,k] )   (W_INT8_SQ, W_FP16)
//
// Both operands are row-major with K contiguous (activations [M][K], weights [N][K]), so one staging scheme serves
// both: a K-tile is 128 BYTES of every tile row (128 int8 | 64 fp16), brought HBM/L2 -> LDS by
// `global_load_lds_dwordx4` (no VGPR round trip, no ds_write pass), double-buffered, ONE barrier per K-tile: the DMA of
// tile t+1 is in flight while the MFMAs of tile t run.
//
// LDS image: [row][128 B], 16-byte column index XOR-swizzled with (row >> 1) & 7 so that the 16 lanes of a
// ds_read_b128 phase (16 consecutive rows, same k-chunk) hit 16 different 16-byte bank groups.  The DMA writes LDS
// lane-linearly (wave-uniform base + lane * 16), so the swizzle is applied to the per-lane GLOBAL source address: the
// 8 lanes of a row read the 8 pieces of that row's 128-byte line in permuted order - still one full line per row.
//
// MFMA: v_mfma_i32_32x32x32_i8 (exact int32) / v_mfma_f32_32x32x16_f16 - identical byte geometry (a k-step is 32
// bytes, lane l holds row l & 31, bytes [16 * (l >> 5), +16)).  Tile shapes are template parameters (waves WM x WN,
// each wave MT x NT MFMA tiles); the launcher picks the shape that fills the 256 CUs best for the problem
// (M = 1024 prefill shapes are one to three workgroup rounds: tile quantisation matters as much as the inner loop).
//
// Epilogue: per-column x per-row scales exactly as the reference's epilogue_per_row_per_col_scale.h:279-347
// (float(acc) * (s_col * s_row)), transposed through LDS so that every lane stores 16 contiguous bytes of a C row.
#include "dev_utils.h"
#include "kernels.h"
#include <map>

namespace tllm
{
namespace kernels
{
using namespace dev;

int gemm_tune_cfg = 0; // test/bench override of the tile shape (0 = heuristic)

namespace
{

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int BKB = 128; // bytes of K per tile row per K-tile

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

// One LDS-DMA instruction: 64 lanes x 16 bytes from per-lane global addresses to LDS [m0 + lane * 16].
// Issued from inline asm ON PURPOSE: for the builtin hipcc places `s_waitcnt vmcnt(0)` in front of the next ds_read (it
// cannot prove the DMA targets the other buffer), which serialises the DMA of tile t+1 with the MFMAs of tile t.  With
// the asm form the compiler sees no outstanding VMEM operation; the kernel waits by hand (vmcnt(0) before the barrier
// that publishes the tile).
__device__ __forceinline__ void glds16(const void* gptr, uint32_t lds_byte)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_byte) : "memory");
}

__device__ __forceinline__ int swz(int row, int c16)
{
    return row * BKB + ((c16 ^ ((row >> 1) & 7)) << 4);
}

template <int WT, int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(64 * WM * WN) void gemm_glds_kernel(const GemmParams p)
{
    constexpr bool SQ = WT == W_INT8_SQ;
    constexpr int ES = SQ ? 1 : 2; // bytes per A / W element
    constexpr int NW = WM * WN, NTHR = 64 * NW;
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr int ROWS = BM + BN;              // tile rows of [A; W]
    constexpr int CHUNKS = ROWS / 8;           // 1 KiB DMA instructions per K-tile
    static_assert(CHUNKS % NW == 0, "tile rows must split evenly over the waves");
    constexpr int CPW = CHUNKS / NW;           // DMA instructions per wave per K-tile
    constexpr int BUF = ROWS * BKB;            // bytes per LDS buffer
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform, and the compiler knows it (LDS-DMA base)
    const int wm = wid / WN, wn = wid % WN;
    // XCD-aware tile order: consecutive workgroup ids go to different XCDs (round-robin dispatch); give each XCD a
    // contiguous range of tiles so that the tiles sharing an A row-panel / W column-panel share an L2
    const int nwg = gridDim.x;
    int wg = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tiles_m = (p.M + BM - 1) / BM;
    const int tm = wg % tiles_m, tn = wg / tiles_m; // M fastest: the (few) row panels of one W panel run together
    const int m0 = tm * BM, n0 = tn * BN;
    const int M = p.M, N = p.N;
    const int ntile = (p.K * ES) / BKB;

    // ---- DMA source addresses: chunk c (8 tile rows) -> lane l: row c * 8 + (l >> 3), LDS slot l & 7
    const char* a_base = reinterpret_cast<const char*>(p.a);
    const char* w_base = reinterpret_cast<const char*>(p.w);
    const char* src[CPW];
#pragma unroll
    for (int i = 0; i < CPW; ++i)
    {
        const int c = i * NW + wid;
        const int row = c * 8 + (lane >> 3);
        const int col = (lane & 7) ^ ((row >> 1) & 7);
        if (row < BM)
        {
            const int gr = m0 + row < M ? m0 + row : M - 1;
            src[i] = a_base + (int64_t) gr * p.lda * ES + col * 16;
        }
        else
        {
            const int gr = n0 + row - BM < N ? n0 + row - BM : N - 1;
            src[i] = w_base + (int64_t) gr * p.ldw + col * 16;
        }
    }
    const uint32_t lds_base = (uint32_t) (uintptr_t) (lds_void_t*) lds;
    auto issue = [&](int t, int buf) {
#pragma unroll
        for (int i = 0; i < CPW; ++i)
        {
            const int c = i * NW + wid;
            glds16(src[i] + (int64_t) t * BKB, lds_base + buf * BUF + c * 1024);
        }
    };

    using acc_t = typename std::conditional<SQ, i32x16, f32x16>::type;
    acc_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i][j][r] = 0;

    const int fr = lane & 31, fk = lane >> 5;
    issue(0, 0);
    for (int t = 0; t < ntile; ++t)
    {
        const int buf = t & 1;
        // tile t has landed (own DMA: vmcnt(0); everyone's: barrier) and nobody still reads buffer buf ^ 1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < ntile)
            issue(t + 1, buf ^ 1);
        const char* As = lds + buf * BUF;
        const char* Bs = As + BM * BKB;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
        {
            uint4 af[MT], bf[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                af[i] = *reinterpret_cast<const uint4*>(As + swz((wm * MT + i) * 32 + fr, ks * 2 + fk));
#pragma unroll
            for (int j = 0; j < NT; ++j)
                bf[j] = *reinterpret_cast<const uint4*>(Bs + swz((wn * NT + j) * 32 + fr, ks * 2 + fk));
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                {
                    if constexpr (SQ)
                    {
                        const i32x4 a4 = {(int) af[i].x, (int) af[i].y, (int) af[i].z, (int) af[i].w};
                        const i32x4 b4 = {(int) bf[j].x, (int) bf[j].y, (int) bf[j].z, (int) bf[j].w};
                        acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, acc[i][j], 0, 0, 0);
                    }
                    else
                    {
                        f16x8 a8, b8;
                        __builtin_memcpy(&a8, &af[i], 16);
                        __builtin_memcpy(&b8, &bf[j], 16);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc[i][j], 0, 0, 0);
                    }
                }
        }
    }

    // ---- epilogue.  acc[i][j][r]: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const float* s_row = p.scale_row;
    const int wave_n0 = n0 + wn * NT * 32;
    float sc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
    {
        const int col = wave_n0 + j * 32 + (lane & 31);
        sc[j] = 1.f;
        if constexpr (SQ)
            sc[j] = p.per_channel ? reinterpret_cast<const float*>(p.scale_col)[col < N ? col : N - 1]
                                  : reinterpret_cast<const float*>(p.scale_col)[0];
    }
    const bool vec_out = p.out_dtype == DT_HALF && !(p.ldc & 7) && !(N & 7) && !(reinterpret_cast<uintptr_t>(p.c) & 15);
    if (vec_out)
    {
        // fp16 rows through a wave-private LDS scratch: [32 rows][NT * 32 halfs], pitch chosen so that the two lane
        // halves (rows + 4) land 16 banks apart
        constexpr int PITCH = NT * 64 + 16;
        __syncthreads(); // every wave has finished reading the operand buffers
        char* scr = lds + wid * (32 * PITCH);
#pragma unroll
        for (int i = 0; i < MT; ++i)
        {
            const int row_base = m0 + (wm * MT + i) * 32;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                {
                    const int rr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    float v;
                    if constexpr (SQ)
                    {
                        const int grow = row_base + rr;
                        const float sr = p.per_token ? s_row[grow < M ? grow : M - 1] : s_row[0];
                        v = (float) acc[i][j][r] * (sc[j] * sr);
                    }
                    else
                        v = acc[i][j][r];
                    *reinterpret_cast<uint16_t*>(scr + rr * PITCH + (j * 32 + (lane & 31)) * 2) = f2h(v);
                }
            // 32 rows x (NT * 4) 16-byte pieces
            constexpr int PIECES = NT * 4;
#pragma unroll
            for (int s = lane; s < 32 * PIECES; s += 64)
            {
                const int rr = s / PIECES, pc = s % PIECES;
                const uint4 v = *reinterpret_cast<const uint4*>(scr + rr * PITCH + pc * 16);
                const int grow = row_base + rr, gcol = wave_n0 + pc * 8;
                if (grow < M && gcol < N)
                    *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p.c) + (int64_t) grow * p.ldc + gcol) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
        {
            const int col = wave_n0 + j * 32 + (lane & 31);
            if (col >= N)
                continue;
#pragma unroll
            for (int r = 0; r < 16; ++r)
            {
                const int row = m0 + (wm * MT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= M)
                    continue;
                const int64_t o = (int64_t) row * p.ldc + col;
                if constexpr (SQ)
                {
                    const int a = acc[i][j][r];
                    if (p.out_dtype == DT_INT32)
                        reinterpret_cast<int32_t*>(p.c)[o] = a;
                    else
                    {
                        const float sr = p.per_token ? s_row[row] : s_row[0];
                        const float v = (float) a * (sc[j] * sr);
                        if (p.out_dtype == DT_HALF)
                            reinterpret_cast<uint16_t*>(p.c)[o] = f2h(v);
                        else
                            reinterpret_cast<float*>(p.c)[o] = v;
                    }
                }
                else
                {
                    const float v = acc[i][j][r];
                    if (p.out_dtype == DT_HALF)
                        reinterpret_cast<uint16_t*>(p.c)[o] = f2h(v);
                    else
                        reinterpret_cast<float*>(p.c)[o] = v;
                }
            }
        }
}

template <int WT, int WM, int WN, int MT, int NT>
int launch_cfg(const GemmParams& p, hipStream_t stream)
{
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr size_t smem = 2 * (size_t) (BM + BN) * BKB;
    auto kfn = gemm_glds_kernel<WT, WM, WN, MT, NT>;
    static bool attr_done = false;
    if (!attr_done)
    {
        if (smem > 64 * 1024)
            (void) hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
        attr_done = true;
    }
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    hipLaunchKernelGGL(kfn, dim3(tiles), dim3(64 * WM * WN), smem, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
    {
        set_error("gemm_glds launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}

// tile shapes: id -> (BM, BN, workgroups resident per CU)
struct Shape
{
    int id, bm, bn, per_cu;
};
constexpr Shape kShapes[] = {{1, 128, 128, 2}, {2, 256, 256, 1}, {3, 256, 192, 1}, {4, 128, 256, 1}};

template <int WT>
int launch_wt(const GemmParams& p, int cfg, hipStream_t stream)
{
    switch (cfg)
    {
    case 1: return launch_cfg<WT, 2, 2, 2, 2>(p, stream);
    case 2: return launch_cfg<WT, 2, 4, 4, 2>(p, stream);
    case 3: return launch_cfg<WT, 4, 2, 2, 3>(p, stream);
    default: return launch_cfg<WT, 2, 2, 2, 4>(p, stream);
    }
}

} // namespace

// returns 0 on success, -1 on a launch error, 1 when the shape / type is not served by this kernel
int launch_gemm_glds(const GemmParams& p, hipStream_t stream)
{
    const bool sq = p.wtype == W_INT8_SQ;
    if (!sq && p.wtype != W_FP16)
        return 1;
    const int es = sq ? 1 : 2;
    if ((reinterpret_cast<uintptr_t>(p.a) & 15) || ((p.lda * es) & 15) || (reinterpret_cast<uintptr_t>(p.w) & 15)
        || (p.ldw & 15) || ((p.K * es) % BKB) || p.K <= 0 || p.M < 32)
        return 1;
    if (!sq && p.out_dtype == DT_INT32)
        return 1;
    int cfg = gemm_tune_cfg;
    if (cfg <= 0 || cfg > 4)
    {
        // fewest workgroup rounds over 256 CUs, then the largest tile (fewest operand re-reads through L2)
        static int cus = 0;
        if (!cus)
        {
            int dev = 0;
            (void) hipGetDevice(&dev);
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
                cus = 256;
        }
        double best = 1e30;
        for (const Shape& s : kShapes)
        {
            const int64_t tiles = (int64_t) ((p.M + s.bm - 1) / s.bm) * ((p.N + s.bn - 1) / s.bn);
            // time ~ tiles on the busiest CU x tile area; the small tile pays ~15 % for its lower MFMA : LDS ratio
            const double cost = (double) ((tiles + cus - 1) / cus) * s.bm * s.bn * (s.id == 1 ? 1.15 : 1.0);
            if (cost < best)
            {
                best = cost;
                cfg = s.id;
            }
        }
    }
    return sq ? launch_wt<W_INT8_SQ>(p, cfg, stream) : launch_wt<W_FP16>(p, cfg, stream);
}

} // namespace kernels
} // namespace tllm
