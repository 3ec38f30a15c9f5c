// Weight-only (int8 / int4 weights, fp16 activations) MFMA GEMM for prefill-shaped problems with the dequantisation IN the
// main loop:  C[m,n] = fp16( (sum_k A[m,k] * q[n,k]) * s[n] )      (A8; reference: the mixed-input CUTLASS GEMM,
// K/cutlass_kernels/fpA_intB_gemm/fpA_intB_gemm_template.h:60-160, which converts the integer tile to fp16 between shared memory
// and the tensor-core fragments - same idea, this chip's instructions).
//
// Why: the first version of this path expanded the whole weight matrix to an fp16 image in a scratch buffer on EVERY call and
// ran the fp16 LDS-DMA kernel on it (gemm.hip woq_expand_kernel): 3 N K bytes of HBM traffic the algorithm does not have - 1.2 GB
// per layer-stack pass of a 1024-token LLaMA-7B prefill - and weight-only prefill was slower than fp16 prefill (20 - 21 ms against
// 17.3 ms).  Here the u8 / nibble tile goes HBM -> LDS by LDS-DMA as it is (half / a quarter of the fp16 tile's bytes, in LDS as
// well), a lane reads the 16 bytes (8 bytes for int4) that hold its k-elements of TWO consecutive MFMA k-steps, and turns them into
// fp16 fragments in registers with the byte / nibble splices of the decode GEMV (0x6400 | b = 1024 + b exactly; gemv_impl.h) -
// 8 VALU per fragment next to the MFMAs, which run on their own pipe.  Exact integers in fp16, fp32 accumulation, the fp16
// per-channel scale once in the epilogue: bit-for-bit the arithmetic of the expanded path (tests/test_gpu_plugins.py
// test_weight_only_quant_matmul, test_woq_prefill_gemm_equals_the_expanded_path).
//
// Geometry.  A stage is 64 k-elements: 128 bytes of every A row (fp16), 64 (int8) or 32 (int4) bytes of every W row.
// MFMA v_mfma_f32_32x32x16_f16: lane l holds row l & 31 and 8 k-elements of half fk = l >> 5.  Which 8 is free as long as A and W
// agree, so k-step ks of a stage takes  k in [32 (ks >> 1) + 16 fk + 8 (ks & 1), + 8):  the two k-steps 2j, 2j + 1 of a lane
// are one contiguous run of 16 elements - ONE ds_read_b128 of int8 weights (one ds_read_b64 of nibbles) feeds two MFMA k-steps,
// and the A fragment is still one aligned 16-byte piece (piece 4 j + 2 fk + (ks & 1) of the row's 128-byte line).
// LDS images (the LDS-DMA writes lane-linearly, so each XOR is applied to the lane's global SOURCE piece):
//   A   [row][128 B], 16-byte piece p at p ^ ((row >> 1) & 7)                  (as gemm_glds.hip)
//   W8  [row][ 64 B], piece p at p ^ ((row >> 2) & 3): the 16 rows of a ds_read_b128 lane group fall on 16 different bank slots
//   W4  [row][ 32 B], piece p at p ^ ((row >> 3) & 1): ds_read_b64, 2-way conflicts between rows 16 apart (the 8-byte unit
//       inside a 16-byte DMA piece cannot be permuted per row)
// Pipeline: the lock-step ring of gemm_glds.hip - S stages, one barrier per stage, counted vmcnt, the next stage's DMA spread over
// the k-steps of the one being computed.
#include "dev_utils.h"
#include "kernels.h"
#include "launch_util.h"
#include <atomic>

namespace tllm
{
namespace kernels
{
using namespace dev;

namespace
{

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void_t;

__device__ __forceinline__ void glds16(const void* gptr, uint32_t lds_byte)
{
    uint32_t m0_keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(m0_keep) : "v"(gptr), "s"(lds_byte) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// 4 bytes u = q + 128 -> 4 fp16 q (two words), exactly: 0x6400 | u = 1024 + u, minus 1152
__device__ __forceinline__ void deq_u8x4(uint32_t w, uint32_t& lo, uint32_t& hi)
{
    const uint32_t magic = 0x64646464u;
    const h2_t bias = {(_Float16) 1152.f, (_Float16) 1152.f};
    lo = h2_as_u32(u32_as_h2(__builtin_amdgcn_perm(magic, w, 0x04010400u)) - bias);
    hi = h2_as_u32(u32_as_h2(__builtin_amdgcn_perm(magic, w, 0x04030402u)) - bias);
}

// 8 nibbles n = q + 8 of one word (order e0 e2 e4 e6 | e1 e3 e5 e7, weight_layout.h) -> 8 fp16 q in element order
__device__ __forceinline__ uint4 deq_u4x8(uint32_t w)
{
    const uint32_t m = 0x64006400u, w8 = w >> 8;
    const h2_t b0 = {(_Float16) 1032.f, (_Float16) 1032.f};
    const h2_t s1 = {(_Float16) 0.0625f, (_Float16) 0.0625f};
    const h2_t b1 = {(_Float16) -72.f, (_Float16) -72.f};
    return make_uint4(h2_as_u32(u32_as_h2((w & 0x000f000fu) | m) - b0), h2_as_u32(u32_as_h2((w & 0x00f000f0u) | m) * s1 + b1),
        h2_as_u32(u32_as_h2((w8 & 0x000f000fu) | m) - b0), h2_as_u32(u32_as_h2((w8 & 0x00f000f0u) | m) * s1 + b1));
}

template <int BITS, int WM, int WN, int MT, int NT, int S>
__global__ __launch_bounds__(64 * WM * WN) void gemm_woq_kernel(const GemmParams p)
{
    constexpr int NW = WM * WN;
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr int WB = BITS == 8 ? 64 : 32;      // W bytes per row and stage
    constexpr int WRPC = 1024 / WB;              // W rows per 1 KiB DMA instruction (16 | 32)
    constexpr int WPPR = WB / 16;                // 16-byte pieces per W row (4 | 2)
    constexpr int ACH = BM / 8, WCH = BN / WRPC; // DMA instructions per stage
    static_assert(BN % WRPC == 0, "W tile rows must fill whole DMA instructions");
    constexpr int CHUNKS = ACH + WCH;
    constexpr int CPW = (CHUNKS + NW - 1) / NW;
    constexpr bool RAGGED = CHUNKS % NW != 0;
    constexpr int STAGE = BM * 128 + BN * WB;
    constexpr int D = S - 1;
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int nwg = gridDim.x;
    int wg = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tiles_m = (p.M + BM - 1) / BM;
    const int tm = wg % tiles_m, tn = wg / tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int M = p.M, N = p.N;
    const int ntile = p.K / 64;

    // ---- DMA sources
    const char* a_base = reinterpret_cast<const char*>(p.a);
    const char* w_base = reinterpret_cast<const char*>(p.w);
    const char* src[CPW];
    int adv[CPW]; // bytes per stage along K of that chunk's operand
    const bool short_wave = RAGGED && (CPW - 1) * NW + wid >= CHUNKS;
#pragma unroll
    for (int i = 0; i < CPW; ++i)
    {
        int c = i * NW + wid;
        c = c < CHUNKS ? c : CHUNKS - 1;
        if (c < ACH)
        {
            const int row = c * 8 + (lane >> 3);
            const int col = (lane & 7) ^ ((row >> 1) & 7);
            const int gr = m0 + row < M ? m0 + row : M - 1;
            src[i] = a_base + (int64_t) gr * p.lda * 2 + col * 16;
            adv[i] = 128;
        }
        else
        {
            const int row = (c - ACH) * WRPC + lane / WPPR;
            const int col = BITS == 8 ? (lane & 3) ^ ((row >> 2) & 3) : (lane & 1) ^ ((row >> 3) & 1);
            const int gr = n0 + row < N ? n0 + row : N - 1;
            src[i] = w_base + (int64_t) gr * p.ldw + col * 16;
            adv[i] = WB;
        }
    }
    const uint32_t lds_base = (uint32_t) (uintptr_t) (lds_void_t*) lds;
    // chunk c lands at: A chunks [0, ACH) x 1 KiB, then the W chunks
    auto issue_part = [&](int t, int part, int parts) {
        const int stg = t % S;
#pragma unroll
        for (int i = 0; i < CPW; ++i)
        {
            const int c = i * NW + wid;
            if (i % parts == part && (!RAGGED || i < CPW - 1 || !short_wave)) // wave-uniform
                glds16(src[i] + (int64_t) t * adv[i], lds_base + stg * STAGE + c * 1024);
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i][j][r] = 0.f;

    const int fr = lane & 31, fk = lane >> 5;
#pragma unroll
    for (int t = 0; t < D; ++t)
        if (t < ntile)
            issue_part(t, 0, 1);
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
        const char* As = lds + (t % S) * STAGE;
        const char* Ws = As + BM * 128;
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2) // pairs of k-steps: k in [32 j2, 32 j2 + 32)
        {
            // W: the lane's 16 k-elements of this pair -> two fp16 fragments per MFMA tile
            uint4 bf[NT][2];
#pragma unroll
            for (int j = 0; j < NT; ++j)
            {
                const int row = (wn * NT + j) * 32 + fr;
                if constexpr (BITS == 8)
                {
                    const uint4 q = *reinterpret_cast<const uint4*>(Ws + row * 64 + (((j2 * 2 + fk) ^ ((row >> 2) & 3)) << 4));
                    deq_u8x4(q.x, bf[j][0].x, bf[j][0].y);
                    deq_u8x4(q.y, bf[j][0].z, bf[j][0].w);
                    deq_u8x4(q.z, bf[j][1].x, bf[j][1].y);
                    deq_u8x4(q.w, bf[j][1].z, bf[j][1].w);
                }
                else
                {
                    const int unit = j2 * 2 + fk; // 8-byte unit of the 32-byte row
                    const uint2 q = *reinterpret_cast<const uint2*>(Ws + row * 32 + ((((unit >> 1) ^ ((row >> 3) & 1))) << 4) + (unit & 1) * 8);
                    bf[j][0] = deq_u4x8(q.x);
                    bf[j][1] = deq_u4x8(q.y);
                }
            }
            if (t + D < ntile)
            {
                __builtin_amdgcn_sched_barrier(0);
                issue_part(t + D, j2, 2);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) // k-step 2 j2 + h
            {
                uint4 af[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i)
                {
                    const int row = (wm * MT + i) * 32 + fr;
                    af[i] = *reinterpret_cast<const uint4*>(As + row * 128 + (((j2 * 4 + fk * 2 + h) ^ ((row >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                    {
                        f16x8 a8, b8;
                        __builtin_memcpy(&a8, &af[i], 16);
                        __builtin_memcpy(&b8, &bf[j][h], 16);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc[i][j], 0, 0, 0);
                    }
            }
        }
    }

    // ---- epilogue: acc[i][j][r]: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); * fp16 scale of the column
    const int wave_n0 = n0 + wn * NT * 32;
    float sc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
    {
        const int col = wave_n0 + j * 32 + (lane & 31);
        sc[j] = h2f(reinterpret_cast<const uint16_t*>(p.scale_col)[col < N ? col : N - 1]);
    }
    const bool vec_out = p.out_dtype == DT_HALF && !(p.ldc & 7) && !(N & 7) && !(reinterpret_cast<uintptr_t>(p.c) & 15)
        && !(reinterpret_cast<uintptr_t>(p.residual) & 15);
    if (vec_out)
    {
        constexpr int PITCH = NT * 64 + 16;
        __syncthreads(); // every wave has finished reading the operand stages
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
                    *reinterpret_cast<uint16_t*>(scr + rr * PITCH + (j * 32 + (lane & 31)) * 2) = f2h(acc[i][j][r] * sc[j]);
                }
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
            if (col >= N)
                continue;
#pragma unroll
            for (int r = 0; r < 16; ++r)
            {
                const int row = m0 + (wm * MT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= M)
                    continue;
                const int64_t o = (int64_t) row * p.ldc + col;
                const float v = acc[i][j][r] * sc[j];
                if (p.out_dtype == DT_HALF)
                {
                    uint16_t hv = f2h(v);
                    if (p.residual)
                        hv = f2h(h2f(hv) + h2f(reinterpret_cast<const uint16_t*>(p.residual)[o]));
                    reinterpret_cast<uint16_t*>(p.c)[o] = hv;
                }
                else
                    reinterpret_cast<float*>(p.c)[o] = v;
            }
        }
}

template <int BITS, int WM, int WN, int MT, int NT, int S>
int launch_woq_cfg(const GemmParams& p, hipStream_t stream)
{
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr size_t smem = (size_t) S * (BM * 128 + BN * (BITS == 8 ? 64 : 32));
    static_assert(smem <= 160 * 1024, "LDS budget");
    static_assert(smem >= (size_t) WM * WN * 32 * (NT * 64 + 16), "epilogue scratch must fit the operand stages");
    auto kfn = gemm_woq_kernel<BITS, WM, WN, MT, NT, S>;
    launch_util::ensure_dynamic_lds(reinterpret_cast<const void*>(kfn), smem);
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    hipLaunchKernelGGL(kfn, dim3(tiles), dim3(64 * WM * WN), smem, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
    {
        set_error("gemm_woq launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}

template <int BITS>
int launch_woq_bits(const GemmParams& p, int cfg, hipStream_t stream)
{
    switch (cfg)
    {
    case 2: return launch_woq_cfg<BITS, 2, 2, 2, 2, 4>(p, stream); // 128 x 128, 4 waves, 3 stages ahead
    case 3: return launch_woq_cfg<BITS, 4, 2, 2, 3, 3>(p, stream); // 256 x 192, 2 stages ahead
    case 4: return launch_woq_cfg<BITS, 2, 2, 4, 3, 2>(p, stream); // 256 x 192 on 4 waves (128 x 96 per wave): half the dequantisation per MFMA
    case 5: return launch_woq_cfg<BITS, 4, 2, 2, 2, 3>(p, stream); // 256 x 128, 8 waves (64 x 64 per wave), 2 stages ahead
    case 6: return launch_woq_cfg<BITS, 4, 2, 2, 2, 2>(p, stream); // 256 x 128, 1 stage ahead
    default: return launch_woq_cfg<BITS, 4, 2, 2, 3, 2>(p, stream); // 256 x 192, 8 waves, 1 stage ahead
    }
}

} // namespace

int gemm_woq_tune_cfg = 0; // test / bench override (tllm_gemm_set_tile_cfg 101..106 -> 1..6)

// returns 0 on success, -1 on a launch error, 1 when the problem is not served (caller falls back to the expanded path)
int launch_gemm_woq(const GemmParams& p, hipStream_t stream)
{
    const bool w8 = p.wtype == W_INT8_WOQ, w4 = p.wtype == W_INT4_WOQ;
    if (!w8 && !w4)
        return 1;
    if ((reinterpret_cast<uintptr_t>(p.a) & 15) || ((p.lda * 2) & 15) || (reinterpret_cast<uintptr_t>(p.w) & 15) || (p.ldw & 15)
        || (p.K % 64) || p.K <= 0 || p.M < 32 || !p.scale_col)
        return 1;
    if (p.out_dtype != DT_HALF && p.out_dtype != DT_FLOAT)
        return 1;
    if (p.residual && (p.out_dtype != DT_HALF))
        return 1;
    if (p.silu_gate
        && (p.residual || p.out_dtype != DT_HALF || (p.ldc & 7) || (p.N & 7) || (reinterpret_cast<uintptr_t>(p.c) & 15)
            || (reinterpret_cast<uintptr_t>(p.silu_gate) & 15)))
        return 1; // the fused SwiGLU gate lives in the vector epilogue
    int cfg = gemm_woq_tune_cfg;
    if (cfg <= 0)
    {
        // fewest workgroup rounds over the CUs (the rule of gemm_glds.hip): at M = 1024 256 x 192 for QKV / gate / up, 128 x 128 for O / down
        const int cus = launch_util::device_cus();
        // cost = workgroup rounds x tile area x the measured cost of a tile of that kind per area, relative to 256 x 192
        // (profiles/r04_tile256x128.txt): 256 x 128 fills the chip in ONE round for O / down at M = 2048 where 128 x 128 takes two
        struct Cand
        {
            int id, bm, bn;
            double f;
        };
        const Cand cands[] = {{1, 256, 192, 1.0}, {6, 256, 128, 1.08}, {2, 128, 128, w8 ? 1.28 : 1.15}};
        double best = 1e30;
        for (const Cand& c : cands)
        {
            const int64_t t = (int64_t) ((p.M + c.bm - 1) / c.bm) * ((p.N + c.bn - 1) / c.bn);
            if (c.id == 6 && t > 2 * cus)
                continue; // beyond two rounds 256 x 192 wins again (M = 8192: down 737 vs 845 us)
            const double cost = (double) ((t + cus - 1) / cus) * c.bm * c.bn * c.f;
            if (cost < best)
            {
                best = cost;
                cfg = c.id;
            }
        }
    }
    return w8 ? launch_woq_bits<8>(p, cfg, stream) : launch_woq_bits<4>(p, cfg, stream);
}

} // namespace kernels
} // namespace tllm
