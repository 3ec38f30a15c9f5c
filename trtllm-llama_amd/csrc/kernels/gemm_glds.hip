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
#include "launch_util.h"
#include <atomic>
#include <map>

namespace tllm
{
namespace kernels
{
using namespace dev;

int gemm_tune_cfg = 0; // test/bench override of the tile shape (0 = heuristic)
int launch_gemm_sqp(const GemmParams& p, int cfg, hipStream_t stream); // gemm_sqp.hip: phased SmoothQuant kernel, ids 13..
int launch_gemm_f16p(const GemmParams& p, int cfg, hipStream_t stream); // gemm_sqp.hip: the same pipeline on fp16 operands, ids 50..
extern void* gemm_clock_probe;                                              // gemm_sqp.hip (microbench hook)

namespace
{

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));


typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

// One LDS-DMA instruction: 64 lanes x 16 bytes from per-lane global addresses to LDS [m0 + lane * 16].
// Issued from inline asm ON PURPOSE: for the builtin hipcc places `s_waitcnt vmcnt(0)` in front of the next ds_read (it
// cannot prove the DMA targets the other buffer), which serialises the DMA of tile t+1 with the MFMAs of tile t.  With
// the asm form the compiler sees no outstanding VMEM operation; the kernel waits by hand (vmcnt(0) before the barrier
// that publishes the tile).
__device__ __forceinline__ void glds16(const void* gptr, uint32_t lds_byte)
{
    uint32_t m0_keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(m0_keep) : "v"(gptr), "s"(lds_byte) : "memory");
}

// the same from a wave-uniform base + a per-lane 32-bit offset
__device__ __forceinline__ void glds16s(const char* base, uint32_t off, uint32_t lds_byte)
{
    uint32_t m0_keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(m0_keep) : "v"(off), "s"(base), "s"(lds_byte) : "memory");
}

// byte offset of 16-byte piece c16 of tile row `row`: XOR swizzle so that the 16 lanes of a ds_read_b128 phase
// (16 consecutive rows, same k-piece) hit 16 different 16-byte bank groups
template <int BKB>
__device__ __forceinline__ int swz(int row, int c16)
{
    if constexpr (BKB == 128)
        return row * 128 + ((c16 ^ ((row >> 1) & 7)) << 4);
    else
        return row * 64 + ((c16 ^ ((row >> 2) & 3)) << 4);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LW > 0: LW extra waves that do nothing but issue the LDS-DMA.  One wave sustains about one LDS-DMA instruction per ~150
// cycles whatever else it does (the instruction holds its issuer for 100 - 185 cycles), which in the 8-wave 256 x 192 shape is a
// third of every compute wave's issue time; loaders (the shape's 160 VGPRs leave room for a third wave per SIMD) take that
// over and run the ring's look-ahead.
template <int WT, int WM, int WN, int MT, int NT, int KG, int BKB, int S, int LW = 0>
__global__ __launch_bounds__(64 * (WM * WN * KG + LW)) void gemm_glds_kernel(const GemmParams p)
{
    constexpr bool SQ = WT == W_INT8_SQ;
    constexpr int ES = SQ ? 1 : 2; // bytes per A / W element
    constexpr int NWC = WM * WN * KG;             // compute waves: KG groups of WM x WN waves split the k-steps of every stage
    constexpr int NW = LW > 0 ? LW : NWC;         // waves that issue the DMA
    static_assert(LW == 0 || KG == 1, "loader waves: one K-group");
    constexpr int KSTEPS = BKB / 32 / KG;         // k-steps per stage per group
    static_assert(KG == 1 || (KG == 2 && BKB == 128 && MT % 2 == 0), "K-groups: 2, on 128-byte stages");
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr int ROWS = BM + BN;                 // tile rows of [A; W]
    constexpr int RPC = 1024 / BKB;               // tile rows per 1 KiB DMA instruction (8 | 16)
    constexpr int PPR = BKB / 16;                 // 16-byte pieces per row (8 | 4)
    constexpr int CHUNKS = ROWS / RPC;            // DMA instructions per stage
    constexpr int CPW = (CHUNKS + NW - 1) / NW;   // ... per wave (the last one may be missing on the high waves)
    constexpr bool RAGGED = CHUNKS % NW != 0;
    constexpr int STAGE = ROWS * BKB;             // bytes per LDS stage
    constexpr int D = S - 1;                      // stages in flight ahead of the one being computed
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform, and the compiler knows it (LDS-DMA base)
    // clock evidence for the microbench (tllm_gemm_set_clock_probe; GemmParams::clock_probe, set by the launcher only)
    void* const clk_probe = SQ ? p.clock_probe : nullptr;
    const uint64_t clk0 = clk_probe ? __builtin_readcyclecounter() : 0, rt0 = clk_probe ? __builtin_amdgcn_s_memrealtime() : 0;
    const bool loader = LW > 0 && wid >= NWC;     // wave-uniform
    const int iw = LW > 0 ? (wid >= NWC ? wid - NWC : 0) : wid; // index among the issuing waves
    const int kg = wid / (WM * WN), wq = wid % (WM * WN);
    const int wm = wq / WN, wn = wq % WN;
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

    // ---- DMA source addresses: chunk c (RPC tile rows) -> lane l: row c * RPC + l / PPR, LDS piece l % PPR
    const char* a_base = reinterpret_cast<const char*>(p.a);
    const char* w_base = reinterpret_cast<const char*>(p.w);
    const char* src[LW > 0 ? 1 : CPW];
    // loader waves carry twice the chunks of a compute wave: 32-bit offsets from the two (wave-uniform) operand bases instead
    // of 64-bit addresses, or the 168-register budget of three waves per SIMD spills
    uint32_t soff[LW > 0 ? CPW : 1];
    constexpr int ACH = BM / RPC; // chunks of the A part of a stage
    static_assert(LW == 0 || ACH % LW == 0, "loader waves: A / W chunk boundary must not split a loader's turn");
    const bool short_wave = RAGGED && (CPW - 1) * NW + iw >= CHUNKS; // this wave has CPW - 1 DMA instructions
#pragma unroll
    for (int i = 0; i < CPW; ++i)
    {
        int c = i * NW + iw;
        c = c < CHUNKS ? c : CHUNKS - 1;
        const int row = c * RPC + lane / PPR;
        const int col = BKB == 128 ? (lane & 7) ^ ((row >> 1) & 7) : (lane & 3) ^ ((row >> 2) & 3);
        if (row < BM)
        {
            const int gr = m0 + row < M ? m0 + row : M - 1;
            if constexpr (LW > 0)
                soff[i] = (uint32_t) ((int64_t) gr * p.lda * ES + col * 16);
            else
                src[i] = a_base + (int64_t) gr * p.lda * ES + col * 16;
        }
        else
        {
            const int gr = n0 + row - BM < N ? n0 + row - BM : N - 1;
            if constexpr (LW > 0)
                soff[i] = (uint32_t) ((int64_t) gr * p.ldw + col * 16);
            else
                src[i] = w_base + (int64_t) gr * p.ldw + col * 16;
        }
    }
    const uint32_t lds_base = (uint32_t) (uintptr_t) (lds_void_t*) lds;
    auto issue = [&](int t) {
        const int stg = t % S;
#pragma unroll
        for (int i = 0; i < CPW; ++i)
        {
            const int c = i * NW + iw;
            if constexpr (LW > 0)
            {
                if (!RAGGED || i < CPW - 1 || !short_wave) // wave-uniform
                    glds16s((i < ACH / LW ? a_base : w_base) + (int64_t) t * BKB, soff[i], lds_base + stg * STAGE + c * 1024);
            }
            else if (!RAGGED || i < CPW - 1 || !short_wave) // wave-uniform
                glds16(src[i] + (int64_t) t * BKB, lds_base + stg * STAGE + c * 1024);
        }
    };
    // the same, spread over the k-steps of the tile being computed: this wave's DMA instructions number i with
    // i mod PARTS == part.  Issued in one burst right after the barrier they held every wave of the SIMD out of the matrix
    // pipe for the whole burst (an LDS-DMA instruction costs its issuer 60-185 cycles); between the MFMA groups their issue
    // time hides under the MFMAs already queued.
    auto issue_part = [&](int t, int part, int parts) {
        const int stg = t % S;
#pragma unroll
        for (int i = 0; i < CPW; ++i)
        {
            const int c = i * NW + iw;
            if constexpr (LW == 0)
                if (i % parts == part && (!RAGGED || i < CPW - 1 || !short_wave)) // wave-uniform
                    glds16(src[i] + (int64_t) t * BKB, lds_base + stg * STAGE + c * 1024);
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
    const bool vec_out = p.out_dtype == DT_HALF && !(p.ldc & 7) && !(N & 7) && !(reinterpret_cast<uintptr_t>(p.c) & 15)
        && !(reinterpret_cast<uintptr_t>(p.residual) & 15);
    if constexpr (LW > 0)
    {
        // the loaders' whole life: request, wait, barrier, request ... - a loop of its own, so that none of the compute waves'
        // registers (accumulators, fragments) is live in it and vice versa
        if (loader)
        {
#pragma unroll
            for (int t = 0; t < D; ++t)
                if (t < ntile)
                    issue(t);
            for (int t = 0; t < ntile; ++t)
            {
                if (t + D - 1 < ntile)
                {
                    if (short_wave)
                        wait_vmcnt<(D - 1) * (CPW - 1)>();
                    else
                        wait_vmcnt<(D - 1) * CPW>();
                }
                else
                    wait_vmcnt<0>();
                __syncthreads();
                if (t + D < ntile)
                    issue(t + D);
            }
            if (vec_out)
                __syncthreads(); // the epilogue's barrier
            return;
        }
    }
    const bool issuer = LW == 0;
    if (issuer)
    {
#pragma unroll
        for (int t = 0; t < D; ++t)
            if (t < ntile)
                issue(t);
    }
    for (int t = 0; t < ntile; ++t)
    {
        // stage t has landed: own DMA by the counted wait (the D - 1 younger stages stay in flight), everyone's by the
        // barrier; the same barrier says nobody still reads stage t - 1, whose buffer the next DMA overwrites
        if (issuer)
        {
            if (t + D - 1 < ntile)
            {
                if (short_wave)
                    wait_vmcnt<(D - 1) * (CPW - 1)>();
                else
                    wait_vmcnt<(D - 1) * CPW>();
            }
            else
                wait_vmcnt<0>();
        }
        __syncthreads();

        // the DMA of tile t + D is spread over the k-steps of tile t (see issue_part): every k-step with stages to spare, the
        // first two k-steps when only one stage is ahead (it must land before the next barrier).  128 x 128 / 3 ahead (O, down)
        // 49 -> 44 us in the real prefill, 256 x 192 / 1 ahead (QKV, gate, up) 53 -> 51 us stand-alone
        constexpr int SPREAD_PARTS = S >= 3 ? KSTEPS : (KSTEPS >= 2 ? 2 : 1);
        const char* As = lds + (t % S) * STAGE;
        const char* Bs = As + BM * BKB;
#pragma unroll
        for (int k2 = 0; k2 < KSTEPS; ++k2)
        {
            const int ks = kg * KSTEPS + k2;
            uint4 af[MT], bf[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                af[i] = *reinterpret_cast<const uint4*>(As + swz<BKB>((wm * MT + i) * 32 + fr, ks * 2 + fk));
#pragma unroll
            for (int j = 0; j < NT; ++j)
                bf[j] = *reinterpret_cast<const uint4*>(Bs + swz<BKB>((wn * NT + j) * 32 + fr, ks * 2 + fk));
            if (LW == 0 && k2 < SPREAD_PARTS && t + D < ntile)
            {
                __builtin_amdgcn_sched_barrier(0);
                issue_part(t + D, k2, SPREAD_PARTS);
                __builtin_amdgcn_sched_barrier(0);
            }
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

    // ---- K-groups: exchange halves through LDS (int32 sums are exact in any order), then group g finalises the
    // MFMA rows i in [g * MT / 2, (g + 1) * MT / 2)
    int i_lo = 0, i_hi = MT;
    if constexpr (KG == 2)
    {
        constexpr int HALF = MT / 2;
        static_assert((size_t) HALF * NT * 16 * 4 * 64 * WM * WN <= (size_t) S * STAGE, "exchange buffer must fit the stages");
        using elem_t = typename std::conditional<SQ, int, float>::type;
        elem_t* xch = reinterpret_cast<elem_t*>(lds) + wq * 64 + lane; // [reg][wave-in-group][lane]
        constexpr int RS = WM * WN * 64;                                // elements per register slot
#pragma unroll
        for (int g = 0; g < 2; ++g)
        {
            // group 1 - g hands its partial sums of the rows that group g finalises
            __syncthreads();
            if (kg == 1 - g)
            {
#pragma unroll
                for (int i = 0; i < HALF; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            xch[((i * NT + j) * 16 + r) * RS] = acc[g * HALF + i][j][r];
            }
            __syncthreads();
            if (kg == g)
            {
#pragma unroll
                for (int i = 0; i < HALF; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            acc[g * HALF + i][j][r] += xch[((i * NT + j) * 16 + r) * RS];
            }
        }
        i_lo = kg * HALF;
        i_hi = i_lo + HALF;
    }

    if (clk_probe && tid == 0)
    {
        uint64_t* dbg = reinterpret_cast<uint64_t*>(clk_probe) + 2 * blockIdx.x;
        dbg[0] = __builtin_readcyclecounter() - clk0;
        dbg[1] = __builtin_amdgcn_s_memrealtime() - rt0;
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
        if constexpr (!SQ)
        {
            if (p.scale_col) // weight-only: integers expanded to fp16, fp16 scale per output channel
                sc[j] = h2f(reinterpret_cast<const uint16_t*>(p.scale_col)[col < N ? col : N - 1]);
        }
        if constexpr (SQ)
            sc[j] = p.per_channel ? reinterpret_cast<const float*>(p.scale_col)[col < N ? col : N - 1]
                                  : reinterpret_cast<const float*>(p.scale_col)[0];
    }
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
            if (i < i_lo || i >= i_hi) // wave-uniform (K-groups)
                continue;
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
                        v = acc[i][j][r] * sc[j];
                    *reinterpret_cast<uint16_t*>(scr + rr * PITCH + (j * 32 + (lane & 31)) * 2) = f2h(v);
                }
            // 32 rows x (NT * 4) 16-byte pieces
            constexpr int PIECES = NT * 4;
#pragma unroll
            for (int s = lane; s < 32 * PIECES; s += 64)
            {
                const int rr = s / PIECES, pc = s % PIECES;
                uint4 v = *reinterpret_cast<const uint4*>(scr + rr * PITCH + pc * 16);
                const int grow = row_base + rr, gcol = wave_n0 + pc * 8;
                if (grow < M && gcol < N)
                {
                    const int64_t o = (int64_t) grow * p.ldc + gcol;
                    if (p.residual)
                    {
                        const uint4 rv = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(p.residual) + o);
                        const uint32_t a4[4] = {v.x, v.y, v.z, v.w}, b4[4] = {rv.x, rv.y, rv.z, rv.w};
                        uint32_t o4[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            o4[e] = (uint32_t) f2h(h2f((uint16_t) (a4[e] & 0xffffu)) + h2f((uint16_t) (b4[e] & 0xffffu)))
                                | ((uint32_t) f2h(h2f((uint16_t) (a4[e] >> 16)) + h2f((uint16_t) (b4[e] >> 16))) << 16);
                        v = make_uint4(o4[0], o4[1], o4[2], o4[3]);
                    }
                    else if (p.silu_gate)
                        v = epi_silu_gate8(v, *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(p.silu_gate) + o));
                    *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p.c) + o) = v;
                }
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
            if (col >= N || i < i_lo || i >= i_hi)
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
                    {
                        const float sr = p.per_token ? s_row[row] : s_row[0];
                        const float v = (float) a * (sc[j] * sr);
                        if (p.out_dtype == DT_INT32)
                            reinterpret_cast<int32_t*>(p.c)[o] = f2i32_rni_sat(v);
                        else if (p.out_dtype == DT_HALF)
                            reinterpret_cast<uint16_t*>(p.c)[o] = f2h(v);
                        else
                            reinterpret_cast<float*>(p.c)[o] = v;
                    }
                }
                else
                {
                    const float v = acc[i][j][r] * sc[j];
                    if (p.out_dtype == DT_HALF)
                        reinterpret_cast<uint16_t*>(p.c)[o] = f2h(v);
                    else
                        reinterpret_cast<float*>(p.c)[o] = v;
                }
            }
        }
}

template <int WT, int WM, int WN, int MT, int NT, int KG, int BKB, int S, int LW = 0>
int launch_cfg(const GemmParams& p, hipStream_t stream)
{
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr size_t smem = (size_t) S * (BM + BN) * BKB;
    static_assert(smem <= 160 * 1024, "LDS budget");
    static_assert(smem >= (size_t) WM * WN * KG * 32 * (NT * 64 + 16), "epilogue scratch must fit the operand stages");
    auto kfn = gemm_glds_kernel<WT, WM, WN, MT, NT, KG, BKB, S, LW>;
    launch_util::ensure_dynamic_lds(reinterpret_cast<const void*>(kfn), smem);
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    hipLaunchKernelGGL(kfn, dim3(tiles), dim3(64 * (WM * WN * KG + LW)), smem, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
    {
        set_error("gemm_glds launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}

// production tile shapes: id -> (BM, BN); the ids index launch_wt's table, the other ids there are the measured
// alternatives kept for the microbench sweep (prefill GEMM)
struct Shape
{
    int id, bm, bn;
    // measured time per (workgroup round x tile area), relative to the phased 256 x 192 tile of the same operand type.  fp16:
    // profiles/r04_tile256x128.txt.  SmoothQuant (r05): relative to the PERSISTENT 256 x 192 tile (profiles/r05_sqgemm_persistent.txt:
    // O at M = 8192 117 us against 135 for 256 x 256 in two rounds, O at M = 4096 63 us for 256 x 128 against 69)
    double f, f_sq;
};
// id 42 is the phased 256 x 128 tile of gemm_sqp.hip (its fp16 sibling is id 54): it has no lock-step form in this file
constexpr int kPhased256x128 = 42;
constexpr Shape kShapes[] = {{8, 128, 128, 1.40, 1.80}, {6, 256, 192, 1.0, 1.0}, {2, 256, 256, 1.04, 1.30}, {4, 128, 256, 1.39, 1.60},
    {kPhased256x128, 256, 128, 1.08, 1.37}};
constexpr int kNumCfg = 12;

template <int WT>
int launch_wt(const GemmParams& p, int cfg, hipStream_t stream)
{
    switch (cfg)
    {
    // production shapes:        waves  MFMA tiles  BKB stages
    case 1: return launch_cfg<WT, 2, 2, 2, 2, 1, 64, 4>(p, stream);  // 128 x 128, 64 KB  -> 2 workgroups per CU
    case 2: return launch_cfg<WT, 2, 4, 4, 2, 1, 64, 4>(p, stream);  // 256 x 256, 128 KB
    case 3: return launch_cfg<WT, 4, 2, 2, 3, 1, 64, 5>(p, stream);  // 256 x 192, 140 KB
    case 4: return launch_cfg<WT, 2, 2, 2, 4, 1, 64, 4>(p, stream);  // 128 x 256, 96 KB
    // experiments
    case 5: return launch_cfg<WT, 2, 2, 2, 2, 1, 128, 2>(p, stream); // 128 x 128, one stage ahead
    case 6: // 256 x 192, one stage ahead
        return launch_cfg<WT, 4, 2, 2, 3, 1, 128, 2>(p, stream);
    case 7: return launch_cfg<WT, 2, 4, 2, 1, 1, 64, 4>(p, stream);  // 128 x 128 on 8 waves
    case 8: return launch_cfg<WT, 2, 2, 2, 2, 1, 128, 4>(p, stream); // 128 x 128, 128-byte stages, 3 ahead (1 per CU)
    case 9: return launch_cfg<WT, 2, 2, 4, 3, 1, 128, 2>(p, stream); // 256 x 192 on 4 waves (128 x 96 per wave)
    case 10: return launch_cfg<WT, 2, 2, 4, 4, 1, 128, 2>(p, stream); // 256 x 256 on 4 waves (128 x 128 per wave)
    case 11: return launch_cfg<WT, 2, 2, 4, 3, 2, 128, 2>(p, stream); // 256 x 192, 2 K-groups of 4 waves (128 x 96 per wave)
    case 12: return launch_cfg<WT, 2, 2, 2, 2, 2, 128, 2>(p, stream);  // 128 x 128, 2 K-groups of 4 waves (64 x 64 per wave)
    case 36: return launch_cfg<WT, 4, 2, 2, 3, 1, 64, 5, 4>(p, stream);  // 256 x 192, 64-byte stages 4 ahead, 8 compute + 4 loader waves
    case 37: return launch_cfg<WT, 4, 2, 2, 3, 1, 128, 2, 4>(p, stream); // 256 x 192, one 128-byte stage ahead, 8 + 4 waves
    default: return launch_cfg<WT, 2, 2, 2, 2, 1, 128, 4>(p, stream);
    }
}

} // namespace

static bool glds_serves(const GemmParams& p)
{
    const bool sq = p.wtype == W_INT8_SQ;
    if (!sq && p.wtype != W_FP16)
        return false;
    const int es = sq ? 1 : 2;
    if ((reinterpret_cast<uintptr_t>(p.a) & 15) || ((p.lda * es) & 15) || (reinterpret_cast<uintptr_t>(p.w) & 15)
        || (p.ldw & 15) || ((p.K * es) % 128) || p.K <= 0 || p.M < 32)
        return false;
    if (!sq && p.out_dtype == DT_INT32)
        return false;
    if (p.residual
        && (p.out_dtype != DT_HALF || (p.ldc & 7) || (p.N & 7) || (reinterpret_cast<uintptr_t>(p.c) & 15)
            || (reinterpret_cast<uintptr_t>(p.residual) & 15)))
        return false; // the fused residual lives in the vector epilogue (a mis-aligned residual: launch_gemm adds it in a pass)
    if (p.silu_gate
        && (sq || p.residual || p.out_dtype != DT_HALF || (p.ldc & 7) || (p.N & 7) || (reinterpret_cast<uintptr_t>(p.c) & 15)
            || (reinterpret_cast<uintptr_t>(p.silu_gate) & 15)))
        return false; // so does the fused SwiGLU gate (fp16 operands only)
    return true;
}

// The static rule: fewest workgroup rounds over the CUs x the tile's area x what a tile of that kind costs per area (the 128-wide
// lock-step tiles pay ~40 % for their lower MFMA : LDS ratio and their exposed prologue / epilogue).  `phased_ok` = false leaves
// out the tile that exists in gemm_sqp.hip only (its launcher refused the problem: alignment, 32-bit DMA offsets).
static int static_shape_cfg(const GemmParams& p, bool phased_ok = true)
{
    const int cus = launch_util::device_cus();
    double best = 1e30;
    int cfg = 8;
    for (const Shape& s : kShapes)
    {
        if (s.id == kPhased256x128 && !phased_ok)
            continue;
        const int64_t tiles = (int64_t) ((p.M + s.bm - 1) / s.bm) * ((p.N + s.bn - 1) / s.bn);
        // beyond two rounds the 256 x 128 tile loses to 256 x 256 / 256 x 192 (M = 8192: 162 vs 136 us on O, 467 vs 377 on gate / up)
        if (s.id == kPhased256x128 && tiles > 2 * cus)
            continue;
        const double cost = (double) ((tiles + cus - 1) / cus) * s.bm * s.bn * (p.wtype == W_INT8_SQ ? s.f_sq : s.f);
        if (cost < best)
        {
            best = cost;
            cfg = s.id;
        }
    }
    return cfg;
}

// the kernel id launch_gemm_glds runs for this problem when no profile entry exists (the tactic profiler's incumbent)
int gemm_static_cfg(const GemmParams& p)
{
    if (!glds_serves(p))
        return 0;
    const int cfg = static_shape_cfg(p);
    // (SmoothQuant: the persistent forms 62 / 63 of gemm_sqp.hip serve fp16 output on 16-byte rows; launch_gemm_glds falls back
    // to the one-tile-per-workgroup forms 42 / 20 for the rest)
    const bool persist = p.out_dtype == DT_HALF && !(p.ldc & 7) && !(p.N & 7) && !(reinterpret_cast<uintptr_t>(p.c) & 15)
        && p.K >= 256;
    if (cfg == kPhased256x128)
        return p.wtype == W_INT8_SQ ? (persist ? 62 : 42) : (persist ? 56 : 54);
    // the 256 x 192 tile runs its phased sibling (gemm_sqp.hip)
    return cfg == 6 ? (p.wtype == W_INT8_SQ ? (persist ? 63 : 20) : (persist ? 55 : 50)) : cfg;
}

// exactly kernel `cfg`, no fall-back: 0 launched, -1 launch error, 1 this kernel does not serve the problem (the tactic profiler)
int launch_gemm_cfg(const GemmParams& p, int cfg, hipStream_t stream)
{
    if (!glds_serves(p))
        return 1;
    const bool sq = p.wtype == W_INT8_SQ;
    if ((cfg >= 1 && cfg <= kNumCfg) || cfg == 36 || cfg == 37)
    {
        // a tile far larger than the problem only burns time in the sweep
        return sq ? launch_wt<W_INT8_SQ>(p, cfg, stream) : launch_wt<W_FP16>(p, cfg, stream);
    }
    return sq ? launch_gemm_sqp(p, cfg, stream) : launch_gemm_f16p(p, cfg, stream);
}

// returns 0 on success, -1 on a launch error, 1 when the shape / type is not served by this kernel
int launch_gemm_glds(const GemmParams& p, hipStream_t stream)
{
    const bool sq = p.wtype == W_INT8_SQ;
    if (!glds_serves(p))
        return 1;
    int cfg = gemm_tune_cfg;
    bool from_table = false;
    if (cfg <= 0)
    {
        // the kernel the on-device profile found fastest for this shape (gemm_tactics.hip), else the static rule below
        cfg = gemm_tactic_lookup(p.wtype, p.M, p.N, p.K);
        from_table = cfg > 0;
    }
    const bool glds_id = (cfg >= 1 && cfg <= kNumCfg) || cfg == 36 || cfg == 37; // ids served by this file's table
    if (cfg > kNumCfg && !glds_id)
    {
        const int r = sq ? launch_gemm_sqp(p, cfg, stream) : launch_gemm_f16p(p, cfg, stream);
        if (r <= 0)
            return r;
        cfg = 0; // not served there (shape / alignment): the heuristic below picks a lock-step shape
    }
    if (cfg <= 0 || !((cfg >= 1 && cfg <= kNumCfg) || cfg == 36 || cfg == 37))
    {
        cfg = static_shape_cfg(p);
        if (cfg == kPhased256x128)
        {
            // one round of 256 x 128 tiles where 128 x 128 would take two (O / down at M = 2048: 34 vs 44 us, 84 vs 107 us)
            int r = sq ? launch_gemm_sqp(p, 62, stream) : launch_gemm_f16p(p, 56, stream); // the persistent form first (r05)
            if (r > 0)
                r = sq ? launch_gemm_sqp(p, 42, stream) : launch_gemm_f16p(p, 54, stream);
            if (r <= 0)
                return r;
            cfg = static_shape_cfg(p, false);
        }
    }
    if (!sq && cfg == 6 && gemm_tune_cfg <= 0 && !from_table)
    {
        // the same sibling on fp16 operands (r04): 6 - 7 % faster on QKV / gate / up at M = 1024 (profiles/r04_fp16_gemm_sweep.txt)
        int r = launch_gemm_f16p(p, 55, stream); // (r05: its persistent form first)
        if (r > 0)
            r = launch_gemm_f16p(p, 50, stream);
        if (r <= 0)
            return r;
    }
    if (sq && cfg == 6 && gemm_tune_cfg <= 0 && !from_table)
    {
        // the 256 x 192 SmoothQuant tile has a phased sibling (gemm_sqp.hip) that measures 2-5 % faster at the 7B prefill
        // shapes; exact either way
        // ... and that one a persistent form (r05: one workgroup per CU walks its tiles, the next tile's first K-tiles are
        // requested under the epilogue, tiles ordered in bands of 4 row tiles): 3 - 10 % on top (profiles/r05_sqgemm_persistent.txt)
        int r = launch_gemm_sqp(p, 63, stream);
        if (r > 0)
            r = launch_gemm_sqp(p, 20, stream);
        if (r <= 0)
            return r;
    }
    if (sq && gemm_clock_probe)
    {
        GemmParams q = p;
        q.clock_probe = gemm_clock_probe;
        return launch_wt<W_INT8_SQ>(q, cfg, stream);
    }
    return sq ? launch_wt<W_INT8_SQ>(p, cfg, stream) : launch_wt<W_FP16>(p, cfg, stream);
}

} // namespace kernels
} // namespace tllm
