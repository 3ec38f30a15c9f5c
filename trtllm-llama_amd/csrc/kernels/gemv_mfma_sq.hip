// SmoothQuant decode GEMM for SEVERAL sequences (5 <= M <= 8 rows by default, static activation scales) on the matrix pipe.
// tllm_gemv_set_mfma_rows moves the threshold (0 = never).
//
//   y[m, n] = epi( float(sum_k x8[m, k] * W[n, k]) * (s_col[n] * s_row) )        x8 = the int8 rows, or sat(rni(RMSNorm(x) * s))
//
// Why: the skinny kernel of gemv_impl.h spends one v_dot4 per row and 16 weight bytes plus a 64-lane reduction per row and output -
// at 8 rows its layer GEMVs take 23.8 us where one row takes 12.4 (profiles/r04_batch_sweep.txt): the vector ALUs, not HBM, bound
// the step.  Here a wave owns 16 weight rows: v_mfma_i32_16x16x64_i8 takes 16 B per lane of W (row = lane & 15, k-bytes
// (lane >> 4) * 16 of a 64-byte k-step) against the activation rows from LDS in the same geometry (rows >= M read a zero row), and
// leaves lane (m = lane & 15) four consecutive outputs n = 4 (lane >> 4) + e: no cross-lane reduction at all, one MFMA per KiB of
// weights.  The four waves of a workgroup split K (256-byte block j of a row group goes to wave j & 3), their int32 partials meet in
// LDS (exact, order-free) and wave 0 finishes the 16 x M outputs while the others already stream the next row group.  Persistent, one
// workgroup per CU; the prologue (the rows into LDS, normalised + quantised when asked) once per workgroup.
//
// The weights come through a private LDS ring per wave, filled by LDS-DMA (see the kernel's comment).  The first version loaded the
// MFMA fragments straight into registers - one 16-byte load per lane, i.e. 16 rows x 64 bytes per instruction: exact, and the weights
// arrived at 2.7 TB/s (the guide's TA-bound fragment-shaped load; 32 KB per wave in flight and 1 / 2 / 3 workgroups per CU changed
// nothing).  With 4-row x 256-byte DMA instructions they arrive at ~3.6 - 4.8 TB/s - still short of the skinny kernel's 7 (a deeper ring
// does not help; 256-byte runs of rows 4 - 11 KB apart are what the memory sees), which is why this kernel only wins where the skinny
// one is ALU-bound: profiles/r04_gemv_mfma.txt - 8 sequences 2208 -> 2407 tokens/s, 6: 1829 -> 1963, 5: 1593 -> 1733, 4: 1604 -> 1462 (so: from 5 on).
//
// Arithmetic = gemv_impl.h's, stage by stage (same RMSNorm summation order, same rounding points, exact integer sums), so the
// results are bit-identical to the skinny kernel's - tests/test_gpu_plugins.py::test_mfma_skinny_gemm_equals_the_valu_kernel.
// Reference semantics: A10 cutlass_extensions/.../epilogue_per_row_per_col_scale.h:279-347; A5 PY/functional.py:3195-3219;
// A6 PY/layers/mlp.py:68-73; A11 K/quantization.cu:31-118.
#include "dev_utils.h"
#include "gemv_args.h"
#include "kernels.h"
#include "launch_util.h"
#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace tllm
{
namespace kernels
{
using namespace dev;

namespace
{
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int kRows = 8; // activation rows in LDS; row kRows is all zeros

__device__ __forceinline__ float silu_mul_fp16_(float g, float u)
{
    // fp16 rounding points of the reference graph (gemv_impl.h silu_mul_fp16; PY/layers/mlp.py:68-73)
    const float g16 = h2f(f2h(g));
    const float u16 = h2f(f2h(u));
    const float a = h2f(f2h(g16 / (1.f + __expf(-g16))));
    return h2f(f2h(a * u16));
}

// one LDS-DMA instruction: 64 lanes x 16 bytes, global (wave-uniform base + per-lane 32-bit offset) -> LDS [lds_byte + lane * 16]
__device__ __forceinline__ void glds16(const char* base, uint32_t off, uint32_t lds_byte)
{
    uint32_t m0_keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0" : "=&s"(m0_keep) : "v"(off), "s"(base), "s"(lds_byte) : "memory");
}

// NXV = 0: int8 activations as given (PRO_NONE); else RMSNorm + static quantiser, a thread keeps NXV 16-byte vectors of a row.
// D = slots of a wave's weight ring.  A slot = one 256-byte K block of the wave's 16 rows (4 KB; SwiGLU: gate + up, 8 KB), filled by
// LDS-DMA instructions of 4 rows x 256 contiguous bytes (whole 128-byte lines - the register form's 16 rows x 64 bytes per instruction
// streamed at 2.7 TB/s); lane l of instruction i fetches piece ((l & 15) + row) & 15 of row = 4 i + (l >> 4): the 16-byte pieces of a row
// are ROTATED by the row index, so that the fragment read of the 16 rows at one k position (piece c: slot (c - row) & 15) falls on
// 16 different bank groups.
template <int NXV, bool SWIGLU, int D>
__global__ __launch_bounds__(256) void gemv_mfma_sq_kernel(const GemvParams p, int pitch, int ngroups)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NACC = SWIGLU ? 2 : 1;
    constexpr int NI = SWIGLU ? 8 : 4;       // DMA instructions per block
    constexpr int SLOT = NI * 1024;          // bytes
    float* red = reinterpret_cast<float*>(smem);
    i32x4* racc = reinterpret_cast<i32x4*>(smem + kRedBytes); // [parity][NACC][4 waves][64 lanes]
    constexpr int RING_OFF = kRedBytes + 2 * NACC * 4 * 64 * 16;
    char* xs = smem + RING_OFF + 4 * D * SLOT;                 // [kRows + 1][pitch] s8
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6); // scalar: M0 and DMA bases
    const int K = p.K, M = p.M;

    // ---- this wave's weight stream: row group g(i) = blockIdx.x + i * gridDim.x; of a group's K / 256 blocks this wave takes w, w + 4, ..
    const int r16 = lane & 15, g4 = lane >> 4;
    const int KB = K / 256;
    const int nblk = (KB - wid + 3) / 4; // may be 0 (K < 1024): the wave still meets the others at the group's barrier
    const int ngroups_mine = (int) blockIdx.x < ngroups ? (ngroups - (int) blockIdx.x + (int) gridDim.x - 1) / (int) gridDim.x : 0;
    const int total = ngroups_mine * nblk;
    const char* wbase = reinterpret_cast<const char*>(p.w);
    const char* ubase = p.w_up ? reinterpret_cast<const char*>(p.w_up) : wbase + (int64_t) p.N * p.ldw;
    const int ring_off = RING_OFF + wid * D * SLOT; // this wave's ring inside the dynamic LDS
    const uint32_t ring = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) void*) (smem + ring_off); // its LDS byte address (M0)
    // DMA source of this lane inside an instruction: row (lane >> 4) of the instruction's four, piece rotated by the row
    uint32_t dma_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        const int row = 4 * i + (lane >> 4);
        dma_off[i] = (uint32_t) row * (uint32_t) p.ldw + (uint32_t) (((lane & 15) + row) & 15) * 16u;
    }
    auto issue = [&](int q) {
        const int gi = q / nblk, j = q - gi * nblk;
        const int64_t goff = (int64_t) ((int) blockIdx.x + gi * (int) gridDim.x) * 16 * p.ldw + (int64_t) (j * 4 + wid) * 256;
        const uint32_t slot = ring + (uint32_t) (q % D) * SLOT;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            glds16(wbase + goff, dma_off[i], slot + i * 1024);
        if constexpr (SWIGLU)
        {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                glds16(ubase + goff, dma_off[i], slot + 4096 + i * 1024);
        }
    };
    for (int q = 0; q < D - 1 && q < total; ++q)
        issue(q); // before the prologue: the first weights do not depend on x

    // ---- prologue: the activation rows (and a zero row) into LDS
    for (int k = tid * 16; k < pitch; k += 256 * 16)
        *reinterpret_cast<uint4*>(xs + (size_t) kRows * pitch + k) = make_uint4(0, 0, 0, 0);
    if constexpr (NXV == 0)
    {
        const int8_t* x8 = reinterpret_cast<const int8_t*>(p.x);
        for (int m = 0; m < M; ++m)
            for (int k = tid * 16; k < K; k += 256 * 16)
                *reinterpret_cast<uint4*>(xs + (size_t) m * pitch + k) = *reinterpret_cast<const uint4*>(x8 + (int64_t) m * p.ldx + k);
    }
    else
    {
        // gemv_impl.h PK_NORM + static quantiser, statement for statement (same per-thread element order, same reductions)
        const float pro_q = p.act_scale[0];
        uint4 gv[NXV];
        const uint16_t* gam = reinterpret_cast<const uint16_t*>(p.gamma);
#pragma unroll
        for (int j = 0; j < NXV; ++j)
        {
            const int k = (tid + j * 256) * 8;
            gv[j] = *reinterpret_cast<const uint4*>(gam + (k < K ? k : K - 8));
        }
        // RB rows at a time: all their loads in ONE round trip and one barrier for their sums of squares (a row at a time cost
        // ~1.3 us per row and workgroup: a dependent L2 round trip + a barrier each).  Per-row arithmetic unchanged.
        constexpr int RB = NXV <= 2 ? 8 : 4;
        for (int m0 = 0; m0 < M; m0 += RB)
        {
            uint4 xv[RB][NXV];
#pragma unroll
            for (int r = 0; r < RB; ++r)
            {
                const int mr = m0 + r < M ? m0 + r : M - 1; // rows beyond M: a valid row again, never used
                const uint16_t* xg = reinterpret_cast<const uint16_t*>(p.x) + (int64_t) mr * p.ldx;
#pragma unroll
                for (int j = 0; j < NXV; ++j)
                {
                    const int k = (tid + j * 256) * 8;
                    xv[r][j] = *reinterpret_cast<const uint4*>(xg + (k < K ? k : K - 8));
                }
            }
#pragma unroll
            for (int r = 0; r < RB; ++r)
            {
#pragma unroll
                for (int j = 0; j < NXV; ++j)
                {
                    const bool ok = (tid + j * 256) * 8 < K;
                    xv[r][j] = make_uint4(ok ? xv[r][j].x : 0u, ok ? xv[r][j].y : 0u, ok ? xv[r][j].z : 0u, ok ? xv[r][j].w : 0u);
                }
                float ss = 0.f;
#pragma unroll
                for (int j = 0; j < NXV; ++j)
                {
                    const uint32_t ws[4] = {xv[r][j].x, xv[r][j].y, xv[r][j].z, xv[r][j].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                    {
                        const h2_t h = u32_as_h2(ws[q]);
                        const float f0 = (float) h.x, f1 = (float) h.y;
                        ss += f0 * f0 + f1 * f1;
                    }
                }
                ss = wave_sum(ss);
                if (lane == 0)
                    red[r * 4 + wid] = ss;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < RB; ++r)
            {
                const int m = m0 + r;
                if (m >= M) // uniform
                    continue;
                const float ss = red[r * 4] + red[r * 4 + 1] + red[r * 4 + 2] + red[r * 4 + 3];
                const float inv = 1.0f / sqrtf(ss / (float) K + p.eps);
#pragma unroll
                for (int j = 0; j < NXV; ++j)
                {
                    const int k = (tid + j * 256) * 8;
                    if (k < K)
                    {
                        const uint32_t xs4[4] = {xv[r][j].x, xv[r][j].y, xv[r][j].z, xv[r][j].w};
                        const uint32_t gs4[4] = {gv[j].x, gv[j].y, gv[j].z, gv[j].w};
                        uint32_t o[2] = {0, 0};
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                        {
                            h2_t h = u32_as_h2(xs4[q]);
                            const h2_t gg = u32_as_h2(gs4[q]);
                            const float n0 = h2f(f2h((float) h.x * inv)), n1 = h2f(f2h((float) h.y * inv));
                            h.x = (_Float16) (n0 * (float) gg.x);
                            h.y = (_Float16) (n1 * (float) gg.y);
                            const uint32_t b0 = (uint8_t) f2i8_rni_sat((float) h.x * pro_q);
                            const uint32_t b1 = (uint8_t) f2i8_rni_sat((float) h.y * pro_q);
                            o[q >> 1] |= (b0 | (b1 << 8)) << (16 * (q & 1));
                        }
                        *reinterpret_cast<uint2*>(xs + (size_t) m * pitch + k) = make_uint2(o[0], o[1]);
                    }
                }
            }
            if (m0 + RB < M)
                __syncthreads(); // `red` is rewritten by the next RB rows
        }
    }
    __syncthreads();

    // ---- main loop
    const int xrow = r16 < M ? r16 : kRows;
    const char* xlane = xs + (size_t) xrow * pitch + g4 * 16;
    i32x4 acc = {0, 0, 0, 0}, accu = {0, 0, 0, 0};
    const float* sc = reinterpret_cast<const float*>(p.scale_col);
    const float* su = reinterpret_cast<const float*>(p.scale_col_up);
    const float rs = p.scale_row ? p.scale_row[0] : 1.f;
    const float rsu = p.scale_row_up ? p.scale_row_up[0] : rs;
    const float epi_q = (SWIGLU && p.epi == EPI_SWIGLU_QSTATIC) ? p.epi_scale[0] : 1.f;
    int par = 0;

    // fragment read of this lane inside a slot: row r16 lives in instruction r16 >> 2 at row-in-instruction r16 & 3; piece c at slot (c - r16) & 15
    const char* wfrag = smem + ring_off + (r16 >> 2) * 1024 + (r16 & 3) * 256;
    auto finish_group = [&](int gi) {
        // hipcc (ROCm 7.2) copies the accumulator out of the AGPRs at the head of the NEXT basic block (v_accvgpr_read, a phi of the
        // batch loop) and, when that block is entered by the branch below, places the wait states its hazard recogniser owes the last
        // MFMA BEHIND the first of those reads: component 0 came back stale - every 4th output wrong whenever a group had more than
        // one batch (found by test_mfma_skinny_gemm_equals_the_valu_kernel; an `s_nop` asm here is moved above the MFMAs).  Taking
        // the accumulator through a VGPR operand makes the copy happen HERE, in the MFMAs' own block, where the hazard is handled.
        asm volatile("; accumulator out of the matrix pipe" : "+v"(acc));
        if constexpr (SWIGLU)
            asm volatile("; accumulator out of the matrix pipe" : "+v"(accu));
        // the four K-quarters of the workgroup meet in LDS; wave 0 finishes the group, the others go on streaming
        racc[((par * NACC + 0) * 4 + wid) * 64 + lane] = acc;
        acc = i32x4{0, 0, 0, 0};
        if constexpr (SWIGLU)
        {
            racc[((par * NACC + 1) * 4 + wid) * 64 + lane] = accu;
            accu = i32x4{0, 0, 0, 0};
        }
        __syncthreads();
        if (wid == 0 && r16 < M)
        {
            const int g = (int) blockIdx.x + gi * (int) gridDim.x;
            const int n0 = g * 16 + g4 * 4;
            i32x4 v = racc[((par * NACC + 0) * 4 + 0) * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w)
                v += racc[((par * NACC + 0) * 4 + w) * 64 + lane];
            i32x4 vu = {0, 0, 0, 0};
            if constexpr (SWIGLU)
            {
                vu = racc[((par * NACC + 1) * 4 + 0) * 64 + lane];
#pragma unroll
                for (int w = 1; w < 4; ++w)
                    vu += racc[((par * NACC + 1) * 4 + w) * 64 + lane];
            }
            const int64_t o = (int64_t) r16 * p.ldy + n0;
            float r0[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                r0[e] = (float) v[e] * (sc[p.per_channel ? n0 + e : 0] * rs);
            if constexpr (SWIGLU)
            {
                float o16[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                {
                    const float s1 = su ? su[p.per_channel ? n0 + e : 0] : sc[p.per_channel ? p.N + n0 + e : 0];
                    o16[e] = silu_mul_fp16_(r0[e], (float) vu[e] * (s1 * rsu));
                }
                if (p.epi == EPI_SWIGLU)
                    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.y) + o) = make_uint2(
                        (uint32_t) f2h(o16[0]) | ((uint32_t) f2h(o16[1]) << 16), (uint32_t) f2h(o16[2]) | ((uint32_t) f2h(o16[3]) << 16));
                else
                {
                    uint32_t q = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        q |= (uint32_t) (uint8_t) f2i8_rni_sat(o16[e] * epi_q) << (8 * e);
                    *reinterpret_cast<uint32_t*>(reinterpret_cast<int8_t*>(p.y) + o) = q;
                }
            }
            else if (p.epi == EPI_RESIDUAL)
            {
                const uint2 rv = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(p.residual) + o);
                const uint16_t rr[4] = {(uint16_t) (rv.x & 0xffffu), (uint16_t) (rv.x >> 16), (uint16_t) (rv.y & 0xffffu), (uint16_t) (rv.y >> 16)};
                uint16_t h[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    h[e] = f2h(h2f(f2h(r0[e])) + h2f(rr[e]));
                *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.y) + o)
                    = make_uint2((uint32_t) h[0] | ((uint32_t) h[1] << 16), (uint32_t) h[2] | ((uint32_t) h[3] << 16));
            }
            else if (p.out_dtype == DT_HALF)
                *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.y) + o) = make_uint2(
                    (uint32_t) f2h(r0[0]) | ((uint32_t) f2h(r0[1]) << 16), (uint32_t) f2h(r0[2]) | ((uint32_t) f2h(r0[3]) << 16));
            else if (p.out_dtype == DT_FLOAT)
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + o) = make_float4(r0[0], r0[1], r0[2], r0[3]);
            else
                *reinterpret_cast<int4*>(reinterpret_cast<int32_t*>(p.y) + o)
                    = make_int4(f2i32_rni_sat(r0[0]), f2i32_rni_sat(r0[1]), f2i32_rni_sat(r0[2]), f2i32_rni_sat(r0[3]));
        }
        par ^= 1;
    };

    int q = 0;
    for (int gi = 0; gi < ngroups_mine; ++gi)
    {
        for (int j = 0; j < nblk; ++j, ++q)
        {
            // keep D - 1 blocks in flight; then block q has landed when at most the younger ones are outstanding (in-order counter;
            // anything the compiler has in flight on top only makes the wait longer)
            if (q + D - 1 < total)
            {
                issue(q + D - 1);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * NI) : "memory");
            }
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const char* slot = wfrag + (q % D) * SLOT;
            const char* xk = xlane + (size_t) (j * 4 + wid) * 256;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4)
            {
                const int c = c4 * 4 + g4; // 16-byte piece of the 256-byte block = this lane's share of k-step c4
                const uint4 xr = *reinterpret_cast<const uint4*>(xk + c4 * 64);
                const uint4 wr = *reinterpret_cast<const uint4*>(slot + ((c - r16) & 15) * 16);
                const i32x4 xa = {(int) xr.x, (int) xr.y, (int) xr.z, (int) xr.w};
                const i32x4 wf = {(int) wr.x, (int) wr.y, (int) wr.z, (int) wr.w};
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf, xa, acc, 0, 0, 0);
                if constexpr (SWIGLU)
                {
                    const uint4 ur = *reinterpret_cast<const uint4*>(slot + 4096 + ((c - r16) & 15) * 16);
                    const i32x4 uf = {(int) ur.x, (int) ur.y, (int) ur.z, (int) ur.w};
                    accu = __builtin_amdgcn_mfma_i32_16x16x64_i8(uf, xa, accu, 0, 0, 0);
                }
            }
            // the slot is free for the DMA of the next iteration only once its fragments are in registers
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        finish_group(gi);
    }
}

template <int NXV, bool SWIGLU, int D>
int launch_inst(const GemvParams& p, int pitch, int ngroups, int grid, size_t smem, hipStream_t stream)
{
    auto kfn = gemv_mfma_sq_kernel<NXV, SWIGLU, D>;
    static std::atomic<size_t> attr_set{0};
    if (smem > 48 * 1024 && attr_set.load() < smem)
    {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem) != hipSuccess)
        {
            set_error("gemv_mfma_sq: cannot get %zu bytes of LDS", smem);
            return -1;
        }
        attr_set.store(smem);
    }
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), smem, stream, p, pitch, ngroups);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess)
    {
        set_error("gemv_mfma_sq launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}

template <int NXV, bool SWIGLU>
int launch_depth(const GemvParams& p, int pitch, int ngroups, int cus, hipStream_t stream)
{
    // the deepest ring the LDS holds next to the activation rows: 4 slots (SwiGLU's double slots: 3), else 3, else 2
    constexpr int SLOT = (SWIGLU ? 8 : 4) * 1024;
    const size_t fixed = kRedBytes + 2 * (SWIGLU ? 2 : 1) * 4 * 64 * 16 + (size_t) (kRows + 1) * pitch;
    auto go = [&](auto d) {
        constexpr int D = decltype(d)::value;
        const size_t smem = fixed + (size_t) 4 * D * SLOT;
        const int grid = ngroups < cus ? ngroups : cus;
        return launch_inst<NXV, SWIGLU, D>(p, pitch, ngroups, grid, smem, stream);
    };
    // (7 slots where they fit were measured too: QKV 15.9 us against 14.5 with 4 - depth is not what bounds the stream)
    if constexpr (!SWIGLU)
    {
        if (fixed + 4 * 4 * SLOT <= 160 * 1024)
            return go(std::integral_constant<int, 4>());
    }
    if (fixed + 4 * 3 * SLOT <= 160 * 1024)
        return go(std::integral_constant<int, 3>());
    if (fixed + 4 * 2 * SLOT <= 160 * 1024)
        return go(std::integral_constant<int, 2>());
    return 1;
}

} // namespace

int gemv_mfma_min_rows = -1; // -1: the default on first use; rows from which launch_gemv takes this kernel (0 = never)

// 0 launched, -1 error, 1 not served (launch_gemv goes on to the skinny kernel)
int launch_gemv_mfma_sq(const GemvParams& p, hipStream_t stream)
{
    if (gemv_mfma_min_rows < 0) // (tllm_gemv_set_mfma_rows moves the threshold: tests, sweeps)
        gemv_mfma_min_rows = 5; // measured: +7 - 9 % tokens/s at 5 - 8 sequences (the skinny kernel runs 5 rows in its 8-row bucket), -9 % at 4 (header)
    if (gemv_mfma_min_rows <= 0 || p.M < gemv_mfma_min_rows || p.M > kRows || p.wtype != W_INT8_SQ)
        return 1;
    const bool swiglu = p.epi == EPI_SWIGLU || p.epi == EPI_SWIGLU_QSTATIC;
    const bool norm = p.pro == PRO_RMSNORM_QSTATIC;
    if (!norm && p.pro != PRO_NONE)
        return 1; // per-token quantisers and the attention-merge prologue stay on the skinny kernel
    if (p.per_token || p.x_pro_out || p.dyn_scale_out || !p.scale_col)
        return 1;
    if ((p.N & 15) || (p.K & 255) || (p.ldw & 15) || (reinterpret_cast<uintptr_t>(p.w) & 15) || (p.ldy & 3))
        return 1;
    if (swiglu && ((p.w_up && (reinterpret_cast<uintptr_t>(p.w_up) & 15)) || (p.epi == EPI_SWIGLU_QSTATIC && !p.epi_scale)))
        return 1;
    if (!swiglu && p.epi != EPI_NONE && p.epi != EPI_RESIDUAL)
        return 1;
    if (p.epi == EPI_RESIDUAL && (!p.residual || p.out_dtype != DT_HALF || (reinterpret_cast<uintptr_t>(p.residual) & 7)))
        return 1;
    const int yes = swiglu ? (p.epi == EPI_SWIGLU ? 2 : 1) : (p.out_dtype == DT_HALF ? 2 : 4);
    if ((reinterpret_cast<uintptr_t>(p.y) & (4 * yes - 1)) || ((p.ldy * yes) & (4 * yes - 1)))
        return 1;
    if (norm)
    {
        if (!p.gamma || !p.act_scale || (reinterpret_cast<uintptr_t>(p.x) & 15) || ((p.ldx * 2) & 15) || (reinterpret_cast<uintptr_t>(p.gamma) & 15)
            || p.K > 256 * 8 * kNXVMax)
            return 1;
    }
    else if ((reinterpret_cast<uintptr_t>(p.x) & 15) || (p.ldx & 15))
        return 1;
    const int pitch = p.K + 16;
    const int cus = launch_util::device_cus();
    if ((int64_t) 16 * p.ldw + 256 >= (1ll << 32))
        return 1; // 32-bit DMA offsets inside a row group
    const int ngroups = p.N / 16;
    if (!norm)
        return swiglu ? launch_depth<0, true>(p, pitch, ngroups, cus, stream) : launch_depth<0, false>(p, pitch, ngroups, cus, stream);
    if (p.K <= 256 * 8 * kNXVSmall)
        return swiglu ? launch_depth<kNXVSmall, true>(p, pitch, ngroups, cus, stream) : launch_depth<kNXVSmall, false>(p, pitch, ngroups, cus, stream);
    return swiglu ? launch_depth<kNXVMax, true>(p, pitch, ngroups, cus, stream) : launch_depth<kNXVMax, false>(p, pitch, ngroups, cus, stream);
}

} // namespace kernels
} // namespace tllm
