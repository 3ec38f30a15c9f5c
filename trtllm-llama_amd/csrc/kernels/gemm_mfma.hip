// MFMA GEMMs for prefill-shaped problems (M > 8):  C[m,n] = epi( sum_k A[m,k] * W[n,k] ).
//
//   W_INT8_SQ   A s8, W s8           v_mfma_i32_32x32x32_i8, exact int32, epilogue float(acc) * (s_col[n] * s_row[m])
//               (A10: K/cutlass_kernels/int8_gemm/int8_gemm_template.h:56-172 + epilogue_per_row_per_col_scale.h:279-347)
//   W_FP16      A fp16, W fp16       v_mfma_f32_32x32x16_f16, fp32 accumulate          (A7: P/gemmPlugin, cuBLAS fp16/fp32)
//   W_INT8_WOQ  A fp16, W u8 = q+128 dequantised to fp16 while staging, fp16 MFMA, * s[n] in the epilogue
//   W_INT4_WOQ  A fp16, W nibbles   (A8: K/cutlass_kernels/fpA_intB_gemm/fpA_intB_gemm_template.h:60-160)
//
// One workgroup of 4 waves computes a 128 x 128 tile of C; every wave owns a 64 x 64 quadrant = 2 x 2 MFMA tiles of
// 32 x 32 (64 accumulator registers).  K advances in slabs of 64 BYTES per operand row (64 int8 or 32 fp16):
// global -> registers (16-byte loads, issued one slab ahead) -> LDS (XOR-swizzled 16-byte chunks, conflict-free
// ds_read_b128 fragment reads) -> MFMA.  Two LDS buffers, one barrier per slab.
// Fragment layouts (gfx950): A/B operand of a 32x32xK MFMA: lane l holds row (l & 31), k-bytes [16*(l>>5), +16) of the
// K-slab's 32-byte half; C/D: col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5).
#include "dev_utils.h"
#include "kernels.h"
#include "weight_layout.h"

namespace tllm
{
namespace kernels
{
using namespace dev;

namespace
{

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BN = 128, BKB = 64; // BKB: bytes of K per operand row per slab (in the LDS image)

// byte offset of 16-byte chunk `c` (0..3) of row `r` inside a [128][64 B] tile, XOR-swizzled so that 16 lanes
// reading the same chunk of 16 consecutive rows hit 16 different 16-byte bank groups
__device__ __forceinline__ int swz(int r, int c)
{
    return r * BKB + ((c ^ ((r >> 2) & 3)) << 4);
}

// u8 (q + 128) x16 -> 16 fp16 (q exactly), two uint4
__device__ __forceinline__ void dequant_u8x16(const uint4& w, uint4& lo, uint4& hi)
{
    const uint32_t magic = 0x64646464u;
    const h2_t bias = {(_Float16) 1152.f, (_Float16) 1152.f};
    const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
    uint32_t o[8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        o[2 * i] = h2_as_u32(u32_as_h2(__builtin_amdgcn_perm(magic, ws[i], 0x04010400u)) - bias);
        o[2 * i + 1] = h2_as_u32(u32_as_h2(__builtin_amdgcn_perm(magic, ws[i], 0x04030402u)) - bias);
    }
    lo = make_uint4(o[0], o[1], o[2], o[3]);
    hi = make_uint4(o[4], o[5], o[6], o[7]);
}

// 8 nibbles of one 32-bit word (layout of weight_layout.h) -> 8 fp16 in natural k order
__device__ __forceinline__ uint4 dequant_u4x8(uint32_t w)
{
    const uint32_t m = 0x64006400u;
    const uint32_t w8 = w >> 8;
    const h2_t b0 = {(_Float16) 1032.f, (_Float16) 1032.f};
    const h2_t s1 = {(_Float16) 0.0625f, (_Float16) 0.0625f};
    const h2_t b1 = {(_Float16) -72.f, (_Float16) -72.f};
    const h2_t e01 = u32_as_h2((w & 0x000f000fu) | m) - b0;
    const h2_t e23 = u32_as_h2((w & 0x00f000f0u) | m) * s1 + b1;
    const h2_t e45 = u32_as_h2((w8 & 0x000f000fu) | m) - b0;
    const h2_t e67 = u32_as_h2((w8 & 0x00f000f0u) | m) * s1 + b1;
    return make_uint4(h2_as_u32(e01), h2_as_u32(e23), h2_as_u32(e45), h2_as_u32(e67));
}

template <int WT>
__global__ __launch_bounds__(256) void gemm_mfma_kernel(const GemmParams p)
{
    constexpr bool SQ = WT == W_INT8_SQ;
    constexpr int A_ES = SQ ? 1 : 2;                 // bytes per A element
    constexpr int KE = BKB / A_ES;                   // K elements per slab (64 int8 | 32 fp16)
    __shared__ __attribute__((aligned(16))) char lds[2][2][BM * BKB]; // [buffer][A|B][tile]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1; // wave quadrant
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int M = p.M, N = p.N, K = p.K;
    const int nslab = (K + KE - 1) / KE;

    // ---- global -> register staging: thread t moves A rows r = t / 4 + {0, 64}, 16-byte chunk c = t % 4
    const int lr = tid >> 2, lc = tid & 3;
    const char* a_base = reinterpret_cast<const char*>(p.a);
    const char* w_base = reinterpret_cast<const char*>(p.w);
    uint4 ra[2], rb[2]; // A: two rows; B: fp16/s8: two rows; woq8: one row-pair worth; see below

    auto load_slab = [&](int s) {
        const int kb = s * BKB + lc * 16; // byte offset along K in A (and in W for s8 / fp16)
#pragma unroll
        for (int h = 0; h < 2; ++h)
        {
            const int r = lr + h * 64;
            ra[h] = make_uint4(0, 0, 0, 0);
            if (m0 + r < M && kb < K * A_ES)
                ra[h] = *reinterpret_cast<const uint4*>(a_base + ((int64_t) (m0 + r) * p.lda) * A_ES + kb);
        }
        if constexpr (WT == W_INT8_SQ || WT == W_FP16)
        {
#pragma unroll
            for (int h = 0; h < 2; ++h)
            {
                const int r = lr + h * 64;
                rb[h] = make_uint4(0, 0, 0, 0);
                if (n0 + r < N && kb < K * A_ES)
                    rb[h] = ld_nt16(w_base + (int64_t) (n0 + r) * p.ldw + kb);
            }
        }
        else if constexpr (WT == W_INT8_WOQ)
        {
            // 32 weights per row per slab = 2 x 16 bytes: thread t -> row t / 2, half t % 2
            const int r = tid >> 1, hf = tid & 1;
            rb[0] = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
            const int kw = s * KE + hf * 16;
            if (n0 + r < N && kw < K)
                rb[0] = ld_nt16(w_base + (int64_t) (n0 + r) * p.ldw + kw);
        }
        else
        {
            // int4: 32 weights per row per slab = 16 bytes: threads 0..127 -> row t
            rb[0] = make_uint4(0x88888888u, 0x88888888u, 0x88888888u, 0x88888888u);
            const int kw = s * KE;
            if (tid < BN && n0 + tid < N && kw < K)
                rb[0] = ld_nt16(w_base + (int64_t) (n0 + tid) * p.ldw + kw / 2);
        }
    };

    auto store_slab = [&](int buf) {
        char* As = lds[buf][0];
        char* Bs = lds[buf][1];
#pragma unroll
        for (int h = 0; h < 2; ++h)
            *reinterpret_cast<uint4*>(As + swz(lr + h * 64, lc)) = ra[h];
        if constexpr (WT == W_INT8_SQ || WT == W_FP16)
        {
#pragma unroll
            for (int h = 0; h < 2; ++h)
                *reinterpret_cast<uint4*>(Bs + swz(lr + h * 64, lc)) = rb[h];
        }
        else if constexpr (WT == W_INT8_WOQ)
        {
            const int r = tid >> 1, hf = tid & 1;
            uint4 lo, hi;
            dequant_u8x16(rb[0], lo, hi);
            *reinterpret_cast<uint4*>(Bs + swz(r, hf * 2)) = lo;
            *reinterpret_cast<uint4*>(Bs + swz(r, hf * 2 + 1)) = hi;
        }
        else
        {
            if (tid < BN)
            {
                *reinterpret_cast<uint4*>(Bs + swz(tid, 0)) = dequant_u4x8(rb[0].x);
                *reinterpret_cast<uint4*>(Bs + swz(tid, 1)) = dequant_u4x8(rb[0].y);
                *reinterpret_cast<uint4*>(Bs + swz(tid, 2)) = dequant_u4x8(rb[0].z);
                *reinterpret_cast<uint4*>(Bs + swz(tid, 3)) = dequant_u4x8(rb[0].w);
            }
        }
    };

    using acc_t = typename std::conditional<SQ, i32x16, f32x16>::type;
    acc_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i][j][r] = 0;

    load_slab(0);
    store_slab(0);
    __syncthreads();
    const int fr = lane & 31, fk = lane >> 5; // fragment row, 16-byte k chunk inside a 32-byte half
    for (int s = 0; s < nslab; ++s)
    {
        const int buf = s & 1;
        if (s + 1 < nslab)
            load_slab(s + 1);
        const char* As = lds[buf][0];
        const char* Bs = lds[buf][1];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) // two 32-byte halves of the slab = two MFMA k-steps
        {
            uint4 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
            {
                af[i] = *reinterpret_cast<const uint4*>(As + swz(wm * 64 + i * 32 + fr, ks * 2 + fk));
                bf[i] = *reinterpret_cast<const uint4*>(Bs + swz(wn * 64 + i * 32 + fr, ks * 2 + fk));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
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
        if (s + 1 < nslab)
        {
            store_slab(buf ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue: C[row][col], col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const float* s_row = p.scale_row;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
        {
            const int col = n0 + wn * 64 + j * 32 + (lane & 31);
            if (col >= N)
                continue;
            float sc = 1.f;
            if constexpr (SQ)
                sc = p.per_channel ? reinterpret_cast<const float*>(p.scale_col)[col] : reinterpret_cast<const float*>(p.scale_col)[0];
            else if constexpr (WT == W_INT8_WOQ || WT == W_INT4_WOQ)
                sc = h2f(reinterpret_cast<const uint16_t*>(p.scale_col)[col]);
#pragma unroll
            for (int r = 0; r < 16; ++r)
            {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= M)
                    continue;
                const int64_t o = (int64_t) row * p.ldc + col;
                if constexpr (SQ)
                {
                    const int a = acc[i][j][r];
                    {
                        const float sr = p.per_token ? s_row[row] : s_row[0];
                        const float v = (float) a * (sc * sr);
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
                    const float v = acc[i][j][r] * sc;
                    if (p.out_dtype == DT_HALF)
                        reinterpret_cast<uint16_t*>(p.c)[o] = f2h(v);
                    else
                        reinterpret_cast<float*>(p.c)[o] = v;
                }
            }
        }
}

template <int WT>
int launch_wt(const GemmParams& p, hipStream_t stream)
{
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM);
    hipLaunchKernelGGL((gemm_mfma_kernel<WT>), grid, dim3(256), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
    {
        set_error("gemm_mfma launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}

} // namespace

// returns 0 on success, -1 on a launch error, 1 when the shape needs the GEMV-slab fallback
int launch_gemm_mfma(const GemmParams& p, hipStream_t stream)
{
    const bool sq = p.wtype == W_INT8_SQ;
    const int a_es = sq ? 1 : 2;
    // 16-byte vector loads: row starts and K extents must be 16-byte multiples
    if ((reinterpret_cast<uintptr_t>(p.a) & 15) || ((p.lda * a_es) & 15) || (reinterpret_cast<uintptr_t>(p.w) & 15)
        || (p.ldw & 15))
        return 1;
    if ((p.K * a_es) % 16)
        return 1;
    if (p.wtype == W_INT8_WOQ && (p.K % 16))
        return 1;
    if (p.wtype == W_INT4_WOQ && (p.K % 32))
        return 1;
    if (!sq && p.out_dtype == DT_INT32)
        return 1;
    switch (p.wtype)
    {
    case W_FP16: return launch_wt<W_FP16>(p, stream);
    case W_INT8_WOQ: return launch_wt<W_INT8_WOQ>(p, stream);
    case W_INT4_WOQ: return launch_wt<W_INT4_WOQ>(p, stream);
    case W_INT8_SQ: return launch_wt<W_INT8_SQ>(p, stream);
    default: return 1;
    }
}

} // namespace kernels
} // namespace tllm
