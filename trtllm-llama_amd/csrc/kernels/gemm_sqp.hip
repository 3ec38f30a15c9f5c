// SmoothQuant int8 GEMM for prefill, phased pipeline:  C[m,n] = float(sum_k X[m,k] * W[n,k]) * (s_col[n] * s_row[m])
// (A10: K/cutlass_kernels/int8_gemm/int8_gemm_template.h:56-172, epilogue
//  K/cutlass_extensions/.../epilogue_per_row_per_col_scale.h:279-347 - exact int32 accumulation, one fp32 scale product.)
//
// Why a second kernel next to gemm_glds.hip: that one is a lock-step loop (one barrier per K-tile, the whole next tile
// requested in a burst, fragment reads right in front of the MFMAs that use them, vmcnt(0) every tile) and its matrix
// pipe sits idle ~55 % of the time (profiles/r02_mfma_pmc.txt).  This one is built around the three things that keep the
// pipe busy on CDNA4:
//   1. the tile's operands live in LDS as QUARTER units (X-half 0/1, W-half 0/1, each a full 128-byte K line per row) in
//      two buffers; a unit is re-requested by LDS-DMA the moment its last reader is past it, so 4-6 units (~56 KB per CU)
//      are always in flight and the wait in front of a barrier is a COUNTED vmcnt that never drains the queue;
//   2. a K-tile is four phases, one C quadrant each (X-half i x W-half j); the fragments of the next phase are read into
//      registers while the MFMAs of this phase run, so no MFMA ever waits for an LDS read it has just issued;
//   3. the LDS-DMA instructions (60-185 cycles of issue each) sit in the MIDDLE of a phase's MFMA run, at different
//      positions for the two waves that share a SIMD, so one wave's MFMAs cover the other's DMA issue.
// MFMA: v_mfma_i32_16x16x64_i8 (random-operand ceiling 4.1 POP/s vs 3.5 for 32x32x32, profiles/r02_mfma_ceiling.txt).
// W rows are the MFMA's A operand and X rows its B operand, so a lane ends up with 4 consecutive output COLUMNS of one
// output row (D row = 4 * (lane >> 4) + r -> n, D col = lane & 15 -> m).
//
// LDS image of a unit: [row][128 B], 16-byte piece p of row r stored at piece p ^ g(r), g(r) = ((r>>1 & 1) << 1) | (r>>1 & 4):
// the lane groups of a ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) then touch 16 different 16-byte bank
// slots for the 16x64 fragment layout (lane l: row l & 15, pieces 4 * kstep + (l >> 4)).  The DMA writes LDS lane-linearly,
// so the same involution is applied to each lane's global SOURCE piece.
#include "dev_utils.h"
#include "kernels.h"
#include "launch_util.h"
#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>

namespace tllm
{
namespace kernels
{
using namespace dev;

namespace
{

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) unsigned gu32;
constexpr int kKsplitFlagBytes = 8192; // split-K: two flag words per tile at the head of the workspace (<= 1024 tiles)

// one LDS-DMA instruction: 64 lanes x 16 bytes, global (wave-uniform base + per-lane 32-bit offset) -> LDS [m0 + lane * 16]
__device__ __forceinline__ void glds16s(const char* base, uint32_t off, uint32_t lds_byte)
{
    uint32_t m0_keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(m0_keep) : "v"(off), "s"(base), "s"(lds_byte) : "memory");
}

// 4 bytes per lane from per-lane 64-bit addresses (scales of two different tensors in one chunk)
__device__ __forceinline__ void glds4v(const void* gptr, uint32_t lds_byte)
{
    uint32_t m0_keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(m0_keep) : "v"(gptr), "s"(lds_byte) : "memory");
}

// the same with 4 bytes per lane (scales)
__device__ __forceinline__ void glds4s(const char* base, uint32_t off, uint32_t lds_byte)
{
    uint32_t m0_keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(m0_keep) : "v"(off), "s"(base), "s"(lds_byte) : "memory");
}

__device__ __forceinline__ int swz_g(int row)
{
    const int j = row >> 1;
    return ((j & 1) << 1) | (j & 4);
}

// SCL: the tile's scales are staged in LDS ahead of the operand stream (the epilogue issues no global load).  false: the epilogue
// reads them from global / L2 instead - 1.25 KB of LDS less, which is what lets TWO 128 x 192 workgroups (2 x 80 KB) share a CU.
// F16 (r04): the same pipeline on fp16 operands - v_mfma_f32_16x16x32_f16 has the byte geometry of v_mfma_i32_16x16x64_i8 (a lane
// holds 16 bytes of a 64-byte k-step of one row; 16 x 16 fp32 / int32 results in the same lanes), a 128-byte K line is 64 halfs,
// so the quarter units, the swizzle, the DMA schedule and the counted waits carry over unchanged; fp32 accumulation, optional fp16
// per-channel scale (the weight-only expand path), A7 P/gemmPlugin/gemmPlugin.cpp:121-190.
// PERSIST (r05): one workgroup per CU walks tiles wg, wg + grid, ...; the fp16 tile leaves through a STAGING area of its own
// (quarter-tile rounds) instead of the operand buffers, so the next tile's scales and its first two K-tiles are requested BEFORE
// the epilogue and land under it; the epilogue's stores stay in the vmcnt queue across the next tile's first waits (counted).
// KSPLIT (r06): TWO workgroups per tile, each walks half of the K-tiles; then workgroup h hands the accumulators of X-half 1 - h to
// its partner (write-through 16-byte stores into the tile's slab, a drained flag: guide G16 R1), adds the partner's X-half h sums to its
// own (sc1 loads) and finishes rows [h AH, (h + 1) AH) of the tile - int32 sums are exact, so the SmoothQuant result is bit-identical
// to the one-pass form.  For problems whose 256-row tiles leave half the chip idle (N = 4096 at M = 1024: 128 tiles on 256 CUs);
// the pair sits on one XCD (workgroup ids are remapped to XCD-contiguous), both must be resident: grid <= CUs, checked by the launcher.
template <int WR, int WC, int MTH, int NTH, int DMA_POS0, int DMA_POS1, bool PRIO, int ABL, int RSP, bool DUAL = false, bool SCL = true,
    bool F16 = false, bool PERSIST = false, bool KSPLIT = false>
__global__ __launch_bounds__(64 * WR * WC, (KSPLIT && MTH == 1) ? 4 : 1) void gemm_sqp_kernel(const GemmParams p)
{
    // (split-K of the 128 x 128 tile: TWO workgroups per CU - four waves per SIMD - so that 512 workgroups are resident at once)
    static_assert(!KSPLIT || (!PERSIST && !DUAL), "split-K: the one-tile form, single GEMM");
    static_assert(SCL || !DUAL, "the fused SwiGLU epilogue reads its scales from LDS");
    static_assert(!PERSIST || SCL || F16, "the persistent form: SmoothQuant with staged scales (fp16 out, or the fused SwiGLU int8 out), or fp16 operands");
    static_assert(!F16 || (!DUAL && !SCL), "the fp16 variant has no dual / staged-scale form");
    using acc_t = typename std::conditional<F16, f32x4, i32x4>::type;
    constexpr int ES = F16 ? 2 : 1; // bytes per operand element
    constexpr int NW = WR * WC;
    constexpr int AH = WR * MTH * 16, BH = WC * NTH * 16; // rows of an X-half / W-half unit
    constexpr int BM = 2 * AH, BN = 2 * BH;
    constexpr int ACH = AH / 8, BCH = BH / 8;           // 1 KiB DMA chunks (8 rows) per unit
    constexpr int APW = (ACH + NW - 1) / NW, BPW = (BCH + NW - 1) / NW; // chunk slots per wave
    static_assert(ACH % NW == 0, "X-half chunks must divide evenly over the waves");
    static_assert((2 * BCH) % NW == 0, "the two W-halves together must divide evenly over the waves");
    constexpr int CYC = (2 * ACH + 2 * BCH) / NW;        // DMA instructions per wave per K-tile (= per 4 consecutive units)
    constexpr int UA = AH * 128, UB = BH * 128;          // unit bytes
    constexpr int BUF = 2 * UA + 2 * UB;                 // one buffer: X0 | X1 | W0 | W1
    constexpr int OFF_X0 = 0, OFF_X1 = UA, OFF_W0 = 2 * UA, OFF_W1 = 2 * UA + UB;
    constexpr int QM = MTH * NTH * 2;                    // MFMAs per phase
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // clock evidence (microbench only: GemmParams::clock_probe, set by the launchers): shader cycles (s_memtime) against the
    // constant 100 MHz counter (s_memrealtime) over this workgroup's lifetime -> the clock the chip actually held
    const uint64_t clk0 = p.clock_probe ? __builtin_readcyclecounter() : 0, rt0 = p.clock_probe ? __builtin_amdgcn_s_memrealtime() : 0;
    const int wr = wid / WC, wc = wid % WC;
    const int grp = wid >= NW / 2 ? 1 : 0; // the second-dispatched half: its DMA sits elsewhere in the phase
    const int nwg = gridDim.x;
    int wg = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tiles_m = (p.M + BM - 1) / BM;
    const int M = p.M, N = p.N;
    const int tiles_n = (N + (DUAL ? BH : BN) - 1) / (DUAL ? BH : BN);
    const int total_tiles = tiles_m * tiles_n;
    // split-K: workgroup 2 t + h = K-half h of tile t
    const int khalf = KSPLIT ? (wg & 1) : 0;
    if constexpr (KSPLIT)
        wg >>= 1;
    const int ntile_all = p.K * ES / 128;
    const int kt0 = KSPLIT ? (khalf ? (ntile_all + 1) / 2 : 0) : 0;
    const int ntile = KSPLIT ? (khalf ? ntile_all - (ntile_all + 1) / 2 : (ntile_all + 1) / 2) : ntile_all;
    // DUAL: W-half 0 = rows [n0, n0 + BH) of the first matrix, W-half 1 = the SAME rows of the second one; BH output columns
    int m0, n0;

    // ---- DMA sources.  Unit kinds: 0 = X0, 1 = W0, 2 = W1, 3 = X1 (issue order inside a K-tile).  Chunk c of a unit
    // covers rows [8c, 8c+8); lane l -> row 8c + (l >> 3), LDS piece l & 7 <- global piece (l & 7) ^ g(row).
    // W-halves may have fewer chunks than 2 per wave: W0 hands its surplus to the low waves, W1 to the high waves.
    const char* xb = reinterpret_cast<const char*>(p.a) + (int64_t) kt0 * 128;
    const char* wb = reinterpret_cast<const char*>(p.w) + (int64_t) kt0 * 128;
    const char* wb2 = DUAL ? reinterpret_cast<const char*>(p.w2) : wb;
    uint32_t xo[2][APW], wo[2][BPW];
    const int wrev = NW - 1 - wid;
    auto set_tile = [&](int tw) {
        // tile order: bands of GM row tiles, row-fastest inside a band.  The 32 workgroups an XCD runs side by side (ids are
        // contiguous per XCD, above) then cover GM row tiles x 32 / GM column tiles, and what the XCD's L2 has to fetch per round is
        // GM X-tiles + 32 / GM W-tiles instead of 32 X-tiles + 1 W-tile (M = 8192: 10 MB instead of 33 MB per XCD and round)
        constexpr int GM = 4; // (2 and 8 measure the same, 16 loses 1 - 3 %: r05)
        const int bi = tw / (GM * tiles_n), rem = tw - bi * (GM * tiles_n);
        const int bh = tiles_m - bi * GM < GM ? tiles_m - bi * GM : GM;
        const int tm = bi * GM + rem % bh, tn = rem / bh;
        m0 = tm * BM, n0 = tn * (DUAL ? BH : BN);
        int lane = tid & 63;
        if constexpr (PERSIST) // per-lane terms recomputed per tile: kept live across the K loop they cost main-loop registers
            asm volatile("" : "+v"(lane));
#pragma unroll
        for (int h = 0; h < 2; ++h)
        {
#pragma unroll
            for (int k = 0; k < APW; ++k)
            {
                const int c = wid + k * NW;
                const int row = c * 8 + (lane >> 3);
                int gr = m0 + h * AH + row;
                gr = gr < M ? gr : M - 1;
                xo[h][k] = (uint32_t) (gr * (int) p.lda * ES + (((lane & 7) ^ swz_g(row)) << 4));
            }
#pragma unroll
            for (int k = 0; k < BPW; ++k)
            {
                int c = (h == 0 ? wid : wrev) + k * NW;
                c = c < BCH ? c : BCH - 1;
                const int row = c * 8 + (lane >> 3);
                int gr = n0 + (DUAL ? 0 : h * BH) + row;
                gr = gr < N ? gr : N - 1;
                wo[h][k] = (uint32_t) (gr * (int) p.ldw + (((lane & 7) ^ swz_g(row)) << 4));
            }
        }
    };
    set_tile(wg);
    const uint32_t lds_base = (uint32_t) (uintptr_t) (lds_void_t*) lds;
    // issue unit `kind` of K-tile t into buffer t & 1
    auto dma = [&](int kind, int t) {
        const uint32_t bufb = lds_base + (t & 1) * BUF;
        if (kind == 0 || kind == 3)
        {
            const int h = kind == 3;
#pragma unroll
            for (int k = 0; k < APW; ++k)
                glds16s(xb + (int64_t) t * 128, xo[h][k], bufb + (h ? OFF_X1 : OFF_X0) + (wid + k * NW) * 1024);
        }
        else
        {
            const int h = kind == 2;
            const int w0 = h == 0 ? wid : wrev;
#pragma unroll
            for (int k = 0; k < BPW; ++k)
                if (w0 + k * NW < BCH) // wave-uniform
                    glds16s((h ? wb2 : wb) + (int64_t) t * 128, wo[h][k], bufb + (h ? OFF_W1 : OFF_W0) + (w0 + k * NW) * 1024);
        }
    };

    // ---- fragment read offsets: lane l -> row l & 15, piece 4 * ks + (l >> 4), swizzled; + 2048 per 16-row MFMA tile
    const int row16 = lane & 15, kq = lane >> 4, gl = swz_g(row16);
    int xr[2], wrd[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
    {
        xr[ks] = (wr * MTH * 16 + row16) * 128 + (((ks * 4 + kq) ^ gl) << 4);
        wrd[ks] = (wc * NTH * 16 + row16) * 128 + (((ks * 4 + kq) ^ gl) << 4);
    }

    acc_t acc[2][2][MTH][NTH];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int m = 0; m < MTH; ++m)
#pragma unroll
                    for (int n = 0; n < NTH; ++n)
                        acc[i][j][m][n] = acc_t{0, 0, 0, 0};
    };
    zero_acc();

    i32x4 fa[2][MTH][2];     // X-half fragments [half][m][ks]
    i32x4 fb0[2][NTH][2];    // W-half 0 fragments, two sets: the next tile's are read while this tile's are still in use
    i32x4 fb1[NTH][2];       // W-half 1 fragments

    auto read_x = [&](i32x4 (&dst)[MTH][2], const char* unit) {
#pragma unroll
        for (int m = 0; m < MTH; ++m)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                dst[m][ks] = *reinterpret_cast<const i32x4*>(unit + xr[ks] + m * 2048);
    };
    auto read_w = [&](i32x4 (&dst)[NTH][2], const char* unit) {
#pragma unroll
        for (int n = 0; n < NTH; ++n)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                dst[n][ks] = *reinterpret_cast<const i32x4*>(unit + wrd[ks] + n * 2048);
    };

    // phase boundary: the unit the next reads touch has landed (own chunks by the counted wait, everybody's by the barrier),
    // own fragment reads are done.  The wait leaves the SIX younger units in flight: the DMA latency under load is ~1.1 us
    // = 3-4 phases, and a wait that left only four in flight (one count for every phase) parked the waves for 25 % of the
    // kernel (SQ_WAIT_ANY).  The six-unit window holds 2 X-halves-pairs + the W-halves in between: per phase and wave half
    //   P1: X1 X0 W0 W1 X1 X0 = 8 + (w0 + w1)      P2: X0 W0 W1 X1 X0 W0 = 6 + (w0 + w1) + w0
    //   P3: W0 W1 X1 X0 W0 W1 = 4 + 2 (w0 + w1)    P4: W1 X1 X0 W0 W1 X1 = 6 + (w0 + w1) + w1      (w0 / w1: this wave's chunks
    //   of a W-half 0 / 1 unit, XPW of an X-half)
#define SQP_TOP(counted, N)                                                                                            \
    do                                                                                                                 \
    {                                                                                                                  \
        if (ABL & 32) /* ablation: no barrier (wrong results) */                                                       \
        {                                                                                                              \
            if (counted)                                                                                               \
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(N) : "memory");                       \
            else                                                                                                       \
                asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");                               \
        }                                                                                                              \
        else if (ABL & 64) /* ablation: barrier, no DMA wait (wrong results) */                                        \
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                            \
        else if (counted)                                                                                              \
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");              \
        else                                                                                                           \
            asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                      \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    } while (0)

    // ---- the tile's scales go to LDS by 4-byte LDS-DMA, ahead of the operand stream (older in the vmcnt order): the
    // epilogue then needs no global load at all.  s_col -> [0, BN) floats, s_row -> [BN, BN + BM) floats behind the buffers
    // (PERSIST: two such areas, the next tile's scales land while this tile's are still in use)
    constexpr int SC_OFF = 2 * BUF;
    constexpr int SCB = (BM + BN) * 4;
    // PERSIST: 16-byte output stores every thread issues for a full tile (fp16: 2 MTH rounds of 16 WR rows; DUAL: the int8 tile at once)
    constexpr int STORES_PER_TILE = DUAL ? BM * (BH / 16) / (NW * 64) : 2 * MTH * (WR * 16 * (BN / 8)) / (NW * 64);
    // head of a tile's stream: its scales, then K-tiles 0 and 1
    auto issue_head = [&](int area) {
        if constexpr (SCL)
        {
            constexpr int CCH = (BN + 63) / 64, RCH = (BM + 63) / 64;
            const char* scb = reinterpret_cast<const char*>(p.scale_col);
            const char* srb = reinterpret_cast<const char*>(p.scale_row);
            const uint32_t sbase = lds_base + SC_OFF + area * SCB;
            for (int c = wid; c < CCH + RCH; c += NW) // wave-uniform
            {
                if (c < CCH)
                {
                    const int j = c * 64 + lane; // column slot of the tile: DUAL keeps the second matrix's scales in [BH, 2 BH)
                    int col = n0 + (DUAL ? (j < BH ? j : j - BH) : j);
                    col = col < N ? col : N - 1;
                    const char* base = (DUAL && j >= BH) ? reinterpret_cast<const char*>(p.scale_col2) : scb;
                    glds4v(base + (p.per_channel ? (int64_t) col * 4 : 0), sbase + c * 256);
                }
                else
                {
                    int row = m0 + (c - CCH) * 64 + lane;
                    row = row < M ? row : M - 1;
                    glds4s(srb, p.per_token ? (uint32_t) row * 4u : 0u, sbase + BN * 4 + (c - CCH) * 256);
                }
            }
        }
        dma(0, 0), dma(1, 0), dma(2, 0), dma(3, 0);
        if (ntile > 1)
            dma(0, 1), dma(1, 1), dma(2, 1), dma(3, 1);
    };
    // ---- prologue: tiles 0 and 1 requested, X0(0) / W0(0) fragments in registers, X0(2) requested.  `behind`: vector-memory
    // instructions this wave has issued AFTER the head (PERSIST: the previous tile's output stores) - negative: unknown, drain
    auto finish_head = [&](int behind) {
        if (ntile > 1 && behind == 0)
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(CYC) : "memory");
        else if (PERSIST && ntile > 1 && behind > 0)
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(CYC + STORES_PER_TILE) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        read_x(fa[0], lds + OFF_X0);
        read_w(fb0[0], lds + OFF_W0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (ntile > 2)
            dma(0, 2);
    };
    issue_head(0);
    if constexpr (!PERSIST)
        finish_head(0);

    // one phase: the MFMAs of C quadrant (I, J); the NR fragment reads of the NEXT phase are spread evenly between them (all
    // 8 waves reading in one burst behind the barrier fills the LDS queue and the in-order waves stall in front of their
    // MFMAs: measured 1.17 us per K-tile without any DMA, against 0.64 of MFMA time), and the DMA of one unit sits after
    // MFMA number `dma_pos`.  Every step is pinned: the destination registers are not used before the next phase, so the
    // pinning costs no wait.
    auto phase = [&](acc_t (&c)[MTH][NTH], const i32x4 (&a)[MTH][2], const i32x4 (&b)[NTH][2], auto& rdst, auto rtiles,
                     const char* runit, const int (&roff)[2], bool do_read, int dma_pos, int kind, int t_dma, bool do_dma) {
        constexpr int RT = decltype(rtiles)::value, NR = RT * 2;
        if (PRIO)
            __builtin_amdgcn_s_setprio(1);
        int q = 0, r = 0;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int m = 0; m < MTH; ++m)
#pragma unroll
                for (int n = 0; n < NTH; ++n)
                {
                    if (q == dma_pos)
                    {
                        if (PRIO)
                            __builtin_amdgcn_s_setprio(0);
                        if (do_dma && !(ABL & 1))
                            dma(kind, t_dma);
                        if (PRIO)
                            __builtin_amdgcn_s_setprio(1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (ABL & 2)
                    {
                        asm volatile("" ::"v"(b[n][ks]), "v"(a[m][ks]));
                    }
                    else if constexpr (F16)
                        c[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, b[n][ks]),
                            __builtin_bit_cast(f16x8, a[m][ks]), c[m][n], 0, 0, 0);
                    else
                        c[m][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b[n][ks], a[m][ks], c[m][n], 0, 0, 0);
                    if (r < NR && q == (RSP ? r * RSP : (r * QM) / NR))
                    {
                        // k-step 0 of every MFMA tile first: the next phase starts with those
                        if (do_read && !(ABL & 4))
                            rdst[r % RT][r / RT] = *reinterpret_cast<const i32x4*>(runit + roff[r / RT] + (r % RT) * 2048);
                        ++r;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    ++q;
                }
        if (dma_pos >= QM && do_dma && !(ABL & 1))
            dma(kind, t_dma);
        if (PRIO)
            __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    using TM = std::integral_constant<int, MTH>;
    using TN = std::integral_constant<int, NTH>;

    auto tile = [&](auto parity, auto group, int t) {
        constexpr int b = decltype(parity)::value;
        // DMA position inside the phase: per wave half, or - DMA_POS0 < 0 - per wave PAIR (four code paths), so that the waves'
        // LDS-DMA instructions reach the CU's one address unit spread over the phase instead of in two bunches
        constexpr int GQ = decltype(group)::value; // wave pair (wid / 2) when DMA_POS0 < 0, else wave half
        constexpr int DP = DMA_POS0 < 0 ? (GQ * QM) / 4 + (GQ & 1) : (GQ ? DMA_POS1 : DMA_POS0);
        const char* buf = lds + b * BUF;
        const char* nbuf = lds + (b ^ 1) * BUF;
        constexpr int G = DMA_POS0 < 0 ? decltype(group)::value / 2 : decltype(group)::value;
        // this wave's chunks per unit: X-halves APW; W-half 0: the low waves carry the surplus, W-half 1: the high waves
        constexpr int W0C = BCH % NW == 0 ? BCH / NW : (G == 0 ? BCH / NW + 1 : BCH / NW);
        constexpr int W1C = BCH % NW == 0 ? BCH / NW : (G == 0 ? BCH / NW : BCH / NW + 1);
        constexpr int V1 = 4 * APW + W0C + W1C, V2 = 3 * APW + 2 * W0C + W1C, V3 = 2 * APW + 2 * (W0C + W1C),
                      V4 = 3 * APW + W0C + 2 * W1C;
        const bool steady = t + 2 < ntile;
        const bool more = t + 1 < ntile;
        // P1: quadrant (0,0); W1(t) -> registers; W0(t+2) requested
        SQP_TOP(steady, V1);
        phase(acc[0][0], fa[0], fb0[b], fb1, TN{}, buf + OFF_W1, wrd, true, DP, 1, t + 2, steady);
        // P2: quadrant (0,1); X1(t) -> registers; W1(t+2) requested
        SQP_TOP(steady, V2);
        phase(acc[0][1], fa[0], fb1, fa[1], TM{}, buf + OFF_X1, xr, true, DP, 2, t + 2, steady);
        // P3: quadrant (1,1); X0(t+1) -> registers; X1(t+2) requested
        SQP_TOP(steady, V3);
        phase(acc[1][1], fa[1], fb1, fa[0], TM{}, nbuf + OFF_X0, xr, more, DP, 3, t + 2, steady);
        // P4: quadrant (1,0); W0(t+1) -> the other register set; X0(t+3) requested
        SQP_TOP(steady, V4);
        phase(acc[1][0], fa[1], fb0[b], fb0[b ^ 1], TN{}, nbuf + OFF_W0, wrd, more, DP, 0, t + 3, t + 3 < ntile);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    auto loop = [&](auto group) {
        for (int t = 0; t < ntile; t += 2)
        {
            tile(P0{}, group, t);
            if (t + 1 < ntile)
                tile(P1{}, group, t + 1);
        }
    };
    auto kloop = [&]() {
        if constexpr (DMA_POS0 < 0)
        {
            static_assert(NW == 8, "per-pair DMA slots are laid out for 8 waves");
            switch (wid >> 1) // wave-uniform
            {
            case 0: loop(std::integral_constant<int, 0>{}); break;
            case 1: loop(std::integral_constant<int, 1>{}); break;
            case 2: loop(std::integral_constant<int, 2>{}); break;
            default: loop(std::integral_constant<int, 3>{}); break;
            }
        }
        else if (grp == 0)
            loop(P0{});
        else
            loop(P1{});
    };
    if constexpr (PERSIST)
    {
        // fp16 tile out in 2 * MTH rounds of RR = 16 * WR rows (round (i, m): every wave's 16 rows of MFMA tile row m of X-half i)
        constexpr int PITCH = BN * 2 + 16, RR = WR * 16, PPR = BN / 8, SPT = RR * PPR / (NW * 64); // 16-byte stores per thread per round
        static_assert((RR * PPR) % (NW * 64) == 0, "a round's pieces must divide evenly over the threads");
        char* stg = lds + SC_OFF + (SCL ? 2 * SCB : 0);
        // the wave-group dispatch is the OUTERMOST branch: with the two K-loop forms re-joining once per tile the register
        // allocator spilled fragments inside the loop (scratch traffic shares the vmcnt queue with the DMA)
        auto persist = [&](auto group) {
        int area = 0, behind = 0;
        for (int tw = wg;;)
        {
            finish_head(behind);
            loop(group);
            // the last LDS read of the operand buffers sits in front of the last tile's P3 barrier: both buffers are free here
            const int em0 = m0, en0 = n0;
            const int next = tw + nwg;
            const bool has_next = next < total_tiles; // workgroup-uniform
            if (has_next)
            {
                set_tile(next);
                issue_head(area ^ 1);
            }
            const float* sc_t = reinterpret_cast<const float*>(lds + SC_OFF + area * SCB);
            const float* sr_t = sc_t + BN;
            int lane = threadIdx.x & 63, tid = threadIdx.x; // (fresh copies: see set_tile)
            asm volatile("" : "+v"(lane), "+v"(tid));
            if constexpr (DUAL)
            {
                // SwiGLU + static quantisation, as in the one-tile form below; the int8 tile [BM][BH] has staging rows of its own
                constexpr int QPITCH = BH + 16, QPPR = BH / 16;
                static_assert((BM * QPPR) % (NW * 64) == 0, "the tile's pieces must divide evenly over the threads");
                const float qs = p.swiglu_qscale[0];
                const float sr2 = p.scale_row2 ? p.scale_row2[0] : p.scale_row[0];
#pragma unroll
                for (int n = 0; n < NTH; ++n)
                {
                    const int cl = (wc * NTH + n) * 16 + 4 * (lane >> 4);
                    const float4 sg = *reinterpret_cast<const float4*>(sc_t + cl);
                    const float4 su = *reinterpret_cast<const float4*>(sc_t + BH + cl);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int m = 0; m < MTH; ++m)
                        {
                            const int rl = i * AH + (wr * MTH + m) * 16 + (lane & 15);
                            const float sr = sr_t[rl];
                            const i32x4 ag = acc[i][0][m][n], au = acc[i][1][m][n];
                            const float sgv[4] = {sg.x, sg.y, sg.z, sg.w}, suv[4] = {su.x, su.y, su.z, su.w};
                            uint32_t q4 = 0;
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                            {
                                const float g16 = h2f(f2h((float) ag[r] * (sgv[r] * sr)));
                                const float u16 = h2f(f2h((float) au[r] * (suv[r] * sr2)));
                                const float a16 = h2f(f2h(g16 / (1.f + __expf(-g16))));
                                const float o16 = h2f(f2h(a16 * u16));
                                q4 |= ((uint32_t) (uint8_t) f2i8_rni_sat(o16 * qs)) << (8 * r);
                            }
                            *reinterpret_cast<uint32_t*>(stg + rl * QPITCH + cl) = q4;
                        }
                }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < BM * QPPR / (NW * 64); ++q)
                {
                    const int k = tid + q * NW * 64;
                    const int rl = k / QPPR, pc = k % QPPR;
                    const int grow = em0 + rl, gcol = en0 + pc * 16;
                    if (grow < M && gcol < N)
                        *reinterpret_cast<uint4*>(reinterpret_cast<int8_t*>(p.c) + (int64_t) grow * p.ldc + gcol)
                            = *reinterpret_cast<const uint4*>(stg + rl * QPITCH + pc * 16);
                }
                if (!has_next)
                    break;
                behind = (em0 + BM <= M && en0 + BH <= N) ? 1 : -1;
                zero_acc();
                area ^= 1;
                tw = next;
                continue;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int m = 0; m < MTH; ++m)
                {
                    if (i | m)
                        __syncthreads(); // the previous round has been read out of the staging rows
                    const int rl = i * AH + (wr * MTH + m) * 16 + (lane & 15);
                    const float sr = F16 ? 1.f : sr_t[rl];
                    char* srow = stg + (wr * 16 + (lane & 15)) * PITCH;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int n = 0; n < NTH; ++n)
                        {
                            const int cl = j * BH + (wc * NTH + n) * 16 + 4 * (lane >> 4);
                            float4 sc;
                            if constexpr (F16)
                            {
                                // fp16 operands: no scale at all, or an fp16 factor per output channel
                                const uint16_t* g = reinterpret_cast<const uint16_t*>(p.scale_col);
                                float v[4] = {1.f, 1.f, 1.f, 1.f};
                                if (g)
                                {
#pragma unroll
                                    for (int r = 0; r < 4; ++r)
                                        v[r] = h2f(g[en0 + cl + r < N ? en0 + cl + r : N - 1]);
                                }
                                sc = make_float4(v[0], v[1], v[2], v[3]);
                            }
                            else
                                sc = *reinterpret_cast<const float4*>(sc_t + cl);
                            const acc_t a = acc[i][j][m][n];
                            // the same two products per element, as packed fp32 multiplies (v_pk_mul_f32)
                            typedef float f32x2 __attribute__((ext_vector_type(2)));
                            const f32x2 v01 = f32x2{(float) a[0], (float) a[1]} * (f32x2{sc.x, sc.y} * f32x2{sr, sr});
                            const f32x2 v23 = f32x2{(float) a[2], (float) a[3]} * (f32x2{sc.z, sc.w} * f32x2{sr, sr});
                            const uint32_t lo = (uint32_t) f2h(v01.x) | ((uint32_t) f2h(v01.y) << 16);
                            const uint32_t hi = (uint32_t) f2h(v23.x) | ((uint32_t) f2h(v23.y) << 16);
                            *reinterpret_cast<uint2*>(srow + cl * 2) = make_uint2(lo, hi);
                        }
                    __syncthreads();
#pragma unroll
                    for (int q = 0; q < SPT; ++q)
                    {
                        const int k = tid + q * NW * 64;
                        const int sr16 = k / PPR, pc = k % PPR;
                        const int grow = em0 + i * AH + ((sr16 >> 4) * MTH + m) * 16 + (sr16 & 15), gcol = en0 + pc * 8;
                        if (grow < M && gcol < N)
                        {
                            uint4 v = *reinterpret_cast<const uint4*>(stg + sr16 * PITCH + pc * 16);
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
                            else if (F16 && p.silu_gate)
                                v = epi_silu_gate8(v, *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(p.silu_gate) + o));
                            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                            __builtin_nontemporal_store(u4{v.x, v.y, v.z, v.w}, reinterpret_cast<u4*>(reinterpret_cast<uint16_t*>(p.c) + o));
                        }
                    }
                }
            if (!has_next)
                break;
            // a full tile: every thread has issued exactly 2 * MTH * SPT stores behind the next head; a ragged one: unknown
            behind = (em0 + BM <= M && en0 + BN <= N && !p.residual && !(F16 && (p.silu_gate || p.scale_col))) ? 1 : -1;
            zero_acc();
            area ^= 1;
            tw = next;
        }
        };
        static_assert(DMA_POS0 >= 0, "the persistent form dispatches on the wave half");
        if (grp == 0)
            persist(P0{});
        else
            persist(P1{});
        if (p.clock_probe && tid == 0)
        {
            uint64_t* dbg = reinterpret_cast<uint64_t*>(p.clock_probe) + 2 * blockIdx.x;
            dbg[0] = __builtin_readcyclecounter() - clk0;
            dbg[1] = __builtin_amdgcn_s_memrealtime() - rt0;
        }
        return;
    }
    else
        kloop();
#undef SQP_TOP

    // ---- epilogue.  acc[i][j][m][n][r]: row i*AH + (wr*MTH + m)*16 + (lane & 15),
    //                                    col j*BH + (wc*NTH + n)*16 + 4*(lane >> 4) + r      (inside the tile)
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); // scales landed (ntile == 1 aside, they have long ago)
    if (p.clock_probe && tid == 0)
    {
        uint64_t* dbg = reinterpret_cast<uint64_t*>(p.clock_probe) + 2 * blockIdx.x;
        dbg[0] = __builtin_readcyclecounter() - clk0;
        dbg[1] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
    if constexpr (KSPLIT)
    {
        // slabs: [tile][receiving half][(j, m, n)][thread] 16 bytes; behind them the flags [tile][receiving half] (zero between launches:
        // the reader re-arms its own).  Stores and loads bypass the caches that are not shared (sc1: write-through / L2-served), the
        // flag is stored behind a drained vmcnt - guide G16 R1; no fence, no cache invalidation.
        constexpr int VSTRIDE = NW * 64 * 16, SLAB = 2 * MTH * NTH * VSTRIDE;
        // (the flags sit at a FIXED place, the head of the workspace: behind the slabs their position moved with the tile count, and a
        //  launch with fewer tiles than an earlier one found its flags inside old slab data - wrong sums on the first launch of a shape)
        char* ws = reinterpret_cast<char*>(p.ksplit_ws);
        gu32* flags = (gu32*) ws;
        char* out_slab = ws + kKsplitFlagBytes + ((size_t) wg * 2 + (1 - khalf)) * SLAB + tid * 16;
        const char* in_slab = ws + kKsplitFlagBytes + ((size_t) wg * 2 + khalf) * SLAB + tid * 16;
        auto exchange = [&](auto hc) {
            constexpr int HK = decltype(hc)::value; // the X-half this workgroup keeps and finishes
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int m = 0; m < MTH; ++m)
#pragma unroll
                    for (int n = 0; n < NTH; ++n)
                    {
                        const acc_t v = acc[1 - HK][j][m][n];
                        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(out_slab + ((j * MTH + m) * NTH + n) * VSTRIDE), "v"(v)
                                     : "memory");
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0)
            {
                __hip_atomic_store(flags + wg * 2 + (1 - khalf), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // the partner is resident (the launcher sizes the grid to the chip): this wait is a few microseconds.  Should it never
                // end, the launch is killed loudly rather than left to write half a sum
                long spins = 0;
                while (__hip_atomic_load(flags + wg * 2 + khalf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
                {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > 200000000L)
                        __builtin_trap();
                }
                __hip_atomic_store(flags + wg * 2 + khalf, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            // the partner's sums of this X-half: agent-scope 8-byte loads (sc1: past this CU's L1, whatever the producer's XCD) that
            // hipcc counts itself - any tile shape, no hand-placed wait
            acc_t in[2][MTH][NTH];
            typedef __attribute__((address_space(1))) unsigned long long gu64;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int m = 0; m < MTH; ++m)
#pragma unroll
                    for (int n = 0; n < NTH; ++n)
                    {
                        const gu64* src = (const gu64*) (in_slab + ((j * MTH + m) * NTH + n) * VSTRIDE);
                        const unsigned long long lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned long long hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                        const u32x4 raw = {(unsigned) lo, (unsigned) (lo >> 32), (unsigned) hi, (unsigned) (hi >> 32)};
                        in[j][m][n] = __builtin_bit_cast(acc_t, raw);
                    }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int m = 0; m < MTH; ++m)
#pragma unroll
                    for (int n = 0; n < NTH; ++n)
                        acc[HK][j][m][n] += in[j][m][n];
        };
        if (khalf == 0) // workgroup-uniform
            exchange(P0{});
        else
            exchange(P1{});
    }
    const float* sc_l = reinterpret_cast<const float*>(lds + SC_OFF);
    const float* sr_l = sc_l + BN;
    // the four column scales of output columns [cl, cl + 4) / the row scale of tile row rl
    auto col_scales = [&](int cl) -> float4 {
        if constexpr (SCL)
            return *reinterpret_cast<const float4*>(sc_l + cl);
        else if constexpr (F16)
        {
            // fp16: no scale at all, or - the weight-only expand path - an fp16 factor per output channel
            const uint16_t* g = reinterpret_cast<const uint16_t*>(p.scale_col);
            if (!g)
                return make_float4(1.f, 1.f, 1.f, 1.f);
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
            {
                const int col = n0 + cl + r;
                v[r] = h2f(g[col < N ? col : N - 1]);
            }
            return make_float4(v[0], v[1], v[2], v[3]);
        }
        else
        {
            const float* g = reinterpret_cast<const float*>(p.scale_col);
            if (!p.per_channel)
                return make_float4(g[0], g[0], g[0], g[0]);
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
            {
                const int col = n0 + cl + r;
                v[r] = g[col < N ? col : N - 1];
            }
            return make_float4(v[0], v[1], v[2], v[3]);
        }
    };
    auto row_scale = [&](int rl) -> float {
        if constexpr (SCL)
            return sr_l[rl];
        else if constexpr (F16)
            return 1.f;
        else
        {
            const int row = m0 + rl;
            return p.scale_row[p.per_token ? (row < M ? row : M - 1) : 0];
        }
    };
    const bool vec16 = p.out_dtype == DT_HALF && !(p.ldc & 7) && !(N & 7) && !(reinterpret_cast<uintptr_t>(p.c) & 15)
        && !(reinterpret_cast<uintptr_t>(p.residual) & 15);
    if (ABL & 8)
    {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int m = 0; m < MTH; ++m)
#pragma unroll
                    for (int n = 0; n < NTH; ++n)
                        asm volatile("" ::"v"(acc[i][j][m][n]));
        return;
    }
    if constexpr (DUAL)
    {
        // SwiGLU + static quantisation: quadrant (i, 0) holds fc (through SiLU), (i, 1) gate, same output column.  Rounding
        // points of the un-fused path: each product to fp16, silu to fp16, the product of the two to fp16, then sat(rni(. * qs))
        const float qs = p.swiglu_qscale[0];
        const float sr2 = p.scale_row2 ? p.scale_row2[0] : p.scale_row[0]; // static scales: one value per GEMM
        constexpr int PITCH = BH + 16; // int8 tile [BM][BH] through LDS, then 16-byte row pieces
        static_assert(BM * PITCH <= 2 * BUF, "the int8 tile must fit the operand buffers");
        const bool vec = !(p.ldc & 15) && !(N & 15) && !(reinterpret_cast<uintptr_t>(p.c) & 15);
        char* ot = lds;
#pragma unroll
        for (int n = 0; n < NTH; ++n)
        {
            const int cl = (wc * NTH + n) * 16 + 4 * (lane >> 4);
            const float4 sg = *reinterpret_cast<const float4*>(sc_l + cl);
            const float4 su = *reinterpret_cast<const float4*>(sc_l + BH + cl);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int m = 0; m < MTH; ++m)
                {
                    const int rl = i * AH + (wr * MTH + m) * 16 + (lane & 15);
                    const float sr = sr_l[rl];
                    const i32x4 ag = acc[i][0][m][n], au = acc[i][1][m][n];
                    const float sgv[4] = {sg.x, sg.y, sg.z, sg.w}, suv[4] = {su.x, su.y, su.z, su.w};
                    uint32_t q4 = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                    {
                        const float g16 = h2f(f2h((float) ag[r] * (sgv[r] * sr)));
                        const float u16 = h2f(f2h((float) au[r] * (suv[r] * sr2)));
                        const float a16 = h2f(f2h(g16 / (1.f + __expf(-g16))));
                        const float o16 = h2f(f2h(a16 * u16));
                        q4 |= ((uint32_t) (uint8_t) f2i8_rni_sat(o16 * qs)) << (8 * r);
                    }
                    if (vec)
                        *reinterpret_cast<uint32_t*>(ot + rl * PITCH + cl) = q4;
                    else
                    {
                        const int row = m0 + rl;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (row < M && n0 + cl + r < N)
                                reinterpret_cast<int8_t*>(p.c)[(int64_t) row * p.ldc + n0 + cl + r] = (int8_t) (q4 >> (8 * r));
                    }
                }
        }
        if (!vec)
            return;
        __syncthreads();
        constexpr int PPR = BH / 16;
        for (int k = tid; k < BM * PPR; k += NW * 64)
        {
            const int rl = k / PPR, pc = k % PPR;
            const int grow = m0 + rl, gcol = n0 + pc * 16;
            if (grow < M && gcol < N)
                *reinterpret_cast<uint4*>(reinterpret_cast<int8_t*>(p.c) + (int64_t) grow * p.ldc + gcol)
                    = *reinterpret_cast<const uint4*>(ot + rl * PITCH + pc * 16);
        }
        return;
    }
    if (vec16)
    {
        // fp16 tile through LDS ([BM][BN] halfs, pitch + 16 bytes: the 16 rows of a ds_write_b64 lane group land on 16
        // different bank pairs), then 16-byte stores of whole rows: 3 (BN = 192) full 128-byte lines per row
        constexpr int PITCH = BN * 2 + 16;
        static_assert(BM * PITCH <= 2 * BUF, "the fp16 tile must fit the operand buffers");
        char* ot = lds;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int n = 0; n < NTH; ++n)
            {
                const int cl = j * BH + (wc * NTH + n) * 16 + 4 * (lane >> 4);
                const float4 sc = col_scales(cl);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int m = 0; m < MTH; ++m)
                    {
                        if (KSPLIT && i != khalf) // (uniform: the other X-half is the partner's)
                            continue;
                        const int rl = i * AH + (wr * MTH + m) * 16 + (lane & 15);
                        const float sr = row_scale(rl);
                        const acc_t a = acc[i][j][m][n];
                        const uint32_t lo = (uint32_t) f2h((float) a[0] * (sc.x * sr)) | ((uint32_t) f2h((float) a[1] * (sc.y * sr)) << 16);
                        const uint32_t hi = (uint32_t) f2h((float) a[2] * (sc.z * sr)) | ((uint32_t) f2h((float) a[3] * (sc.w * sr)) << 16);
                        *reinterpret_cast<uint2*>(ot + rl * PITCH + cl * 2) = make_uint2(lo, hi);
                    }
            }
        __syncthreads();
        constexpr int PPR = BN / 8; // 16-byte pieces per row
#pragma unroll 4
        for (int k = tid; k < BM * PPR; k += NW * 64)
        {
            const int rl = k / PPR, pc = k % PPR;
            const int grow = m0 + rl, gcol = n0 + pc * 8;
            if (KSPLIT && rl / AH != khalf)
                continue;
            if (grow < M && gcol < N)
            {
                uint4 v = *reinterpret_cast<const uint4*>(ot + rl * PITCH + pc * 16);
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
                else if (F16 && p.silu_gate)
                    v = epi_silu_gate8(v, *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(p.silu_gate) + o));
                if (ABL & 16)
                {
                    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store(u4{v.x, v.y, v.z, v.w}, reinterpret_cast<u4*>(reinterpret_cast<uint16_t*>(p.c) + o));
                }
                else
                    *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p.c) + o) = v;
            }
        }
        return;
    }
    // every other output form (int32 / fp32, unaligned or ragged fp16): element stores straight from the accumulators
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int n = 0; n < NTH; ++n)
        {
            const int cl = j * BH + (wc * NTH + n) * 16 + 4 * (lane >> 4);
            const int col = n0 + cl;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int m = 0; m < MTH; ++m)
                {
                    const int rl = i * AH + (wr * MTH + m) * 16 + (lane & 15);
                    const int row = m0 + rl;
                    if (row >= M || (KSPLIT && i != khalf))
                        continue;
                    const float sr = row_scale(rl);
                    const float4 sc4 = col_scales(cl);
                    const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w};
                    const acc_t a = acc[i][j][m][n];
                    const int64_t o = (int64_t) row * p.ldc + col;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                    {
                        if (col + r >= N)
                            continue;
                        {
                            const float v = (float) a[r] * (scv[r] * sr);
                            if (p.out_dtype == DT_INT32)
                                reinterpret_cast<int32_t*>(p.c)[o + r] = f2i32_rni_sat(v);
                            else if (p.out_dtype == DT_FLOAT)
                                reinterpret_cast<float*>(p.c)[o + r] = v;
                            else
                            {
                                uint16_t h = f2h(v);
                                if (p.residual)
                                    h = f2h(h2f(h) + h2f(reinterpret_cast<const uint16_t*>(p.residual)[o + r]));
                                reinterpret_cast<uint16_t*>(p.c)[o + r] = h;
                            }
                        }
                    }
                }
        }
}

// split-K workspace of a stream (two GEMMs of one stream never overlap; two streams must not share the slabs), per device
void* ksplit_workspace(hipStream_t stream, size_t bytes)
{
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, std::pair<void*, size_t>> ws;
    int dev = 0;
    (void) hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    auto& e = ws[std::make_pair(dev, stream)];
    if (e.second < bytes)
    {
        // (a hipMalloc: not inside a stream capture - the first use of a shape comes from the tactic profile or an eager prefill)
        // the flags must read zero when the first kernel on this workspace starts: cleared ON THAT STREAM (a hipMemset on the NULL
        // stream is not ordered against a non-blocking stream - a recycled allocation then shows the pair garbage flags: found by
        // tests that ran clean alone and failed behind other tests of the same process)
        void* d = nullptr;
        if (hipMalloc(&d, bytes) != hipSuccess || hipMemsetAsync(d, 0, bytes, stream) != hipSuccess)
            return nullptr;
        if (e.first)
            (void) hipFree(e.first);
        e = std::make_pair(d, bytes);
    }
    return e.first;
}

template <int WR, int WC, int MTH, int NTH, int DP0, int DP1, bool PRIO, int ABL = 0, int RSP = 0, bool DUAL = false, bool SCL = true,
    bool F16 = false, bool PERSIST = false, bool KSPLIT = false>
int launch_sqp(const GemmParams& pin, hipStream_t stream)
{
    GemmParams p = pin;
    constexpr int BM = 2 * WR * MTH * 16, BN = 2 * WC * NTH * 16;
    // operand buffers + the tile's scales (PERSIST: two scale areas + the staging rows of one output round)
    constexpr size_t smem = (size_t) 2 * (BM + BN) * 128 + (SCL ? (BM + BN) * 4 : 0) * (PERSIST ? 2 : 1)
        + (PERSIST ? (DUAL ? BM * (BN / 2 + 16) : WR * 16 * (BN * 2 + 16)) : 0);
    static_assert(smem <= 160 * 1024, "LDS budget");
    auto kfn = gemm_sqp_kernel<WR, WC, MTH, NTH, DP0, DP1, PRIO, ABL, RSP, DUAL, SCL, F16, PERSIST, KSPLIT>;
    launch_util::ensure_dynamic_lds(reinterpret_cast<const void*>(kfn), smem);
    constexpr int BNO = DUAL ? BN / 2 : BN; // output columns per tile
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BNO - 1) / BNO);
    int grid = tiles;
    if (PERSIST)
    {
        // every workgroup the same number of tiles (+-1): ceil(tiles / rounds) workgroups, rounds = ceil(tiles / CUs)
        const int cus = launch_util::device_cus();
        const int rounds = (tiles + cus - 1) / cus;
        grid = (tiles + rounds - 1) / rounds;
    }
    if constexpr (KSPLIT)
    {
        // two workgroups per tile, both resident
        // (both workgroups of every pair must be resident: the occupancy query of THIS instance - registers and LDS - x the CUs)
        const int cus = launch_util::device_cus();
        const int per_cu = launch_util::blocks_per_cu(reinterpret_cast<const void*>(kfn), 64 * WR * WC, smem);
        if (2 * tiles > cus * per_cu || p.K * (F16 ? 2 : 1) / 128 < 4)
            return 1;
        grid = 2 * tiles;
        constexpr size_t SLAB = (size_t) 2 * MTH * NTH * WR * WC * 64 * 16;
        if (tiles * 2 * 4 > kKsplitFlagBytes)
            return 1;
        p.ksplit_ws = ksplit_workspace(stream, (size_t) kKsplitFlagBytes + (size_t) tiles * 2 * SLAB);
        if (!p.ksplit_ws)
        {
            set_error("gemm_sqp: no split-K workspace");
            return -1;
        }
    }
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * WR * WC), smem, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
    {
        set_error("gemm_sqp launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}

} // namespace

// A/B hook (tllm_gemm_set_tile_cfg(-2) / (-3)): the fused SwiGLU GEMM in its one-tile-per-workgroup form
bool gemm_swiglu_one_tile = false;

// microbench hook (tllm_gemm_set_clock_probe): device buffer of 2 x uint64 per workgroup = {shader cycles, 100 MHz ticks}
void* gemm_clock_probe = nullptr;

// Shape ids (tllm_gemm_set_tile_cfg 13..): returns 1 when this kernel does not serve the problem
int launch_gemm_sqp(const GemmParams& pin, int cfg, hipStream_t stream)
{
    GemmParams p = pin;
    p.clock_probe = gemm_clock_probe;
    if (p.wtype != W_INT8_SQ || p.silu_gate)
        return 1;
    if ((reinterpret_cast<uintptr_t>(p.a) & 15) || (p.lda & 15) || (reinterpret_cast<uintptr_t>(p.w) & 15) || (p.ldw & 15)
        || (p.K % 128) || p.K <= 0 || p.M < 32)
        return 1;
    if ((int64_t) p.M * p.lda >= (1ll << 31) || (int64_t) p.N * p.ldw >= (1ll << 31))
        return 1; // 32-bit DMA offsets
    if (p.residual
        && (p.out_dtype != DT_HALF || (p.ldc & 7) || (p.N & 7) || (reinterpret_cast<uintptr_t>(p.c) & 15)
            || (reinterpret_cast<uintptr_t>(p.residual) & 15)))
        return 1; // the fused residual lives in the vector epilogue: a mis-aligned one is not served
    switch (cfg)
    {
    //                              waves  tiles/half  DMA after MFMA # (low / high waves)  setprio
    // (r05: the measured alternatives of the 256 x 192 shape - DMA slots 2/8 with and without setprio, head / tail slots, packed
    //  fragment reads, one slot per wave pair, and the two-workgroups-per-CU 128 x 192 form - were within noise or slower
    //  (profiles/r02_sqgemm_shapes.txt, r03_sqgemm_two_per_cu.txt) and are gone; ids 13, 16, 17, 19, 28 - 30, 40, 41)
    case 15: return launch_sqp<2, 2, 2, 2, 1, 5, true>(p, stream);  // 128 x 128, 4 waves, 2 workgroups per CU
    case 20: return launch_sqp<4, 2, 2, 3, 0, 6, false, 16>(p, stream); // non-temporal output stores
    case 18: return launch_sqp<4, 2, 1, 2, 1, 3, true>(p, stream);  // 128 x 128 on 8 waves
    // 256 x 128 on 8 waves (64 x 64 wave tiles): the tile a split-K-2 pass of the O / down shapes would run (256 workgroups at M = 1024)
    case 42: return launch_sqp<4, 2, 2, 2, 0, 4, false, 16>(p, stream);
    // persistent forms of 20 / 42 (fp16 output on 16-byte rows, K >= 256): one workgroup per CU, the next tile's head under the epilogue
    case 60:
    case 62:
        if (p.out_dtype != DT_HALF || (p.ldc & 7) || (p.N & 7) || (reinterpret_cast<uintptr_t>(p.c) & 15)
            || (reinterpret_cast<uintptr_t>(p.residual) & 15) || p.K < 256)
            return 1;
        return cfg == 60 ? launch_sqp<4, 2, 2, 3, 0, 6, false, 16, 0, false, true, false, true>(p, stream)
                         : launch_sqp<4, 2, 2, 2, 0, 4, false, 16, 0, false, true, false, true>(p, stream);
    case 64: // r06: 256 x 128 as TWO workgroups per tile, each half of K (for N = 4096 at M = 1024: 128 tiles on 256 CUs)
        if (p.out_dtype != DT_HALF || (p.ldc & 7) || (p.N & 7) || (reinterpret_cast<uintptr_t>(p.c) & 15)
            || (reinterpret_cast<uintptr_t>(p.residual) & 15))
            return 1;
        return launch_sqp<4, 2, 2, 2, 0, 4, false, 16, 0, false, true, false, false, true>(p, stream);
    case 65: // split-K-2 of the 128 x 128 tile on 8 waves, two workgroups per CU (N = 4096 at M <= 512: 19 / 36 us for the O / down
             // shapes against 21 / 43 of id 64 and 37 / 89 of the one-pass forms, profiles/r06_sqgemm_wave_tiles.txt)
        if (p.out_dtype != DT_HALF || (p.ldc & 7) || (p.N & 7) || (reinterpret_cast<uintptr_t>(p.c) & 15)
            || (reinterpret_cast<uintptr_t>(p.residual) & 15))
            return 1;
        return launch_sqp<4, 2, 1, 2, 1, 3, true, 0, 0, false, true, false, false, true>(p, stream);
    // (split-K of the 256 x 192 tile: 256 VGPRs with a spill, and slower than id 64 on every shape at M = 256 / 512 - not kept)
    case 63: // the same with the DMA slots of id 13 (after MFMA 2 / 8 of a phase): the production form
        if (p.out_dtype != DT_HALF || (p.ldc & 7) || (p.N & 7) || (reinterpret_cast<uintptr_t>(p.c) & 15)
            || (reinterpret_cast<uintptr_t>(p.residual) & 15) || p.K < 256)
            return 1;
        return launch_sqp<4, 2, 2, 3, 2, 8, false, 16, 0, false, true, false, true>(p, stream);
    // (r06: FOUR waves - one per SIMD, accumulators in AGPRs - with wave tiles of 128 x 96 / 128 x 64 / 64 x 128 measured against the
    //  8-wave forms: 4 - 13 % slower at M = 1024 ... 4096, profiles/r06_sqgemm_wave_tiles.txt; ids 70, 71, 73 are gone)
    // ablations of the 256 x 192 shape (wrong results on purpose; microbench only)
    case 21: return launch_sqp<4, 2, 2, 3, 2, 8, false, 1>(p, stream); // no DMA in the loop
    case 22: return launch_sqp<4, 2, 2, 3, 2, 8, false, 2>(p, stream); // no MFMA
    case 23: return launch_sqp<4, 2, 2, 3, 2, 8, false, 4>(p, stream); // no fragment reads
    case 24: return launch_sqp<4, 2, 2, 3, 2, 8, false, 3>(p, stream); // barriers + reads only
    case 25: return launch_sqp<4, 2, 2, 3, 2, 8, false, 6>(p, stream); // DMA + barriers only
    case 26: return launch_sqp<4, 2, 2, 3, 2, 8, false, 8>(p, stream); // no epilogue
    case 27: return launch_sqp<4, 2, 2, 3, 2, 8, false, 5>(p, stream); // MFMA + barriers only
    case 31: return launch_sqp<4, 2, 2, 3, 0, 6, false, 32>(p, stream); // everything, without the barriers
    case 32: return launch_sqp<4, 2, 2, 3, 0, 6, false, 64>(p, stream); // everything, without the DMA waits
    case 33: return launch_sqp<4, 2, 2, 3, 0, 6, false, 32 + 5>(p, stream); // MFMA only, no barriers
    default: return 1;
    }
}

// fp16 operands on the phased pipeline (ids 50..): returns 1 when this kernel does not serve the problem
int launch_gemm_f16p(const GemmParams& pin, int cfg, hipStream_t stream)
{
    GemmParams p = pin;
    p.clock_probe = gemm_clock_probe;
    if (p.wtype != W_FP16 || p.out_dtype == DT_INT32)
        return 1;
    if ((reinterpret_cast<uintptr_t>(p.a) & 15) || ((p.lda * 2) & 15) || (reinterpret_cast<uintptr_t>(p.w) & 15) || (p.ldw & 15)
        || ((p.K * 2) % 128) || p.K <= 0 || p.M < 32)
        return 1;
    if ((int64_t) p.M * p.lda * 2 >= (1ll << 31) || (int64_t) p.N * p.ldw >= (1ll << 31))
        return 1; // 32-bit DMA offsets
    if (p.residual
        && (p.out_dtype != DT_HALF || (p.ldc & 7) || (p.N & 7) || (reinterpret_cast<uintptr_t>(p.c) & 15)
            || (reinterpret_cast<uintptr_t>(p.residual) & 15)))
        return 1; // the fused residual lives in the vector epilogue: a mis-aligned one is not served
    if (p.silu_gate
        && (p.residual || p.out_dtype != DT_HALF || (p.ldc & 7) || (p.N & 7) || (reinterpret_cast<uintptr_t>(p.c) & 15)
            || (reinterpret_cast<uintptr_t>(p.silu_gate) & 15)))
        return 1; // the fused SwiGLU gate lives in the vector epilogue
    switch (cfg)
    {
    case 50: return launch_sqp<4, 2, 2, 3, 0, 6, false, 16, 0, false, false, true>(p, stream); // 256 x 192, non-temporal stores
    case 51: return launch_sqp<4, 2, 1, 2, 1, 3, true, 0, 0, false, false, true>(p, stream);  // 128 x 128 on 8 waves
    case 52: return launch_sqp<2, 2, 2, 2, 1, 5, true, 0, 0, false, false, true>(p, stream);  // 128 x 128 on 4 waves, 2 per CU
    case 53: return launch_sqp<4, 2, 2, 3, 2, 8, true, 0, 0, false, false, true>(p, stream);  // 256 x 192 with setprio
    case 54: return launch_sqp<4, 2, 2, 2, 0, 4, false, 16, 0, false, false, true>(p, stream); // 256 x 128
    // persistent forms (r05: one workgroup per CU, the next tile's first K-tiles under the epilogue, band tile order)
    case 55:
    case 56:
        if (p.out_dtype != DT_HALF || (p.ldc & 7) || (p.N & 7) || (reinterpret_cast<uintptr_t>(p.c) & 15)
            || (reinterpret_cast<uintptr_t>(p.residual) & 15) || (reinterpret_cast<uintptr_t>(p.silu_gate) & 15) || p.K < 128)
            return 1;
        return cfg == 55 ? launch_sqp<4, 2, 2, 3, 2, 8, false, 16, 0, false, false, true, true>(p, stream)  // 256 x 192
                         : launch_sqp<4, 2, 2, 2, 0, 4, false, 16, 0, false, false, true, true>(p, stream); // 256 x 128
    case 57: // r06: 256 x 128 as two workgroups per tile, each half of K
        if (p.out_dtype != DT_HALF || (p.ldc & 7) || (p.N & 7) || (reinterpret_cast<uintptr_t>(p.c) & 15)
            || (reinterpret_cast<uintptr_t>(p.residual) & 15) || (reinterpret_cast<uintptr_t>(p.silu_gate) & 15))
            return 1;
        return launch_sqp<4, 2, 2, 2, 0, 4, false, 16, 0, false, false, true, false, true>(p, stream);
    case 58: // split-K-2 of the 128 x 128 tile, two workgroups per CU (the fp16 twin of id 65)
        if (p.out_dtype != DT_HALF || (p.ldc & 7) || (p.N & 7) || (reinterpret_cast<uintptr_t>(p.c) & 15)
            || (reinterpret_cast<uintptr_t>(p.residual) & 15) || (reinterpret_cast<uintptr_t>(p.silu_gate) & 15))
            return 1;
        return launch_sqp<4, 2, 1, 2, 1, 3, true, 0, 0, false, false, true, false, true>(p, stream);
    default: return 1;
    }
}

int launch_gemm_swiglu(const GemmParams& p, hipStream_t stream)
{
    if (p.wtype != W_INT8_SQ || !p.w2 || !p.scale_col2 || !p.swiglu_qscale || p.per_token || p.residual)
        return 1;
    if ((reinterpret_cast<uintptr_t>(p.a) & 15) || (p.lda & 15) || (reinterpret_cast<uintptr_t>(p.w) & 15)
        || (reinterpret_cast<uintptr_t>(p.w2) & 15) || (p.ldw & 15) || (p.K % 128) || p.K <= 0 || p.M < 32)
        return 1;
    if ((int64_t) p.M * p.lda >= (1ll << 31) || (int64_t) p.N * p.ldw >= (1ll << 31))
        return 1;
    // 256 rows x 96 columns of both matrices; int8 output on 16-byte rows: the persistent form (r05)
    if (!(p.ldc & 15) && !(p.N & 15) && !(reinterpret_cast<uintptr_t>(p.c) & 15) && p.K >= 256 && !gemm_swiglu_one_tile)
        return launch_sqp<4, 2, 2, 3, 2, 8, false, 0, 0, true, true, false, true>(p, stream);
    return launch_sqp<4, 2, 2, 3, 0, 6, false, 0, 0, true>(p, stream);
}

} // namespace kernels
} // namespace tllm
