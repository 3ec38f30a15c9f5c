// GEMM entry point of the Gemm / SmoothQuantGemm / WeightOnlyQuantMatmul plugins.
//   M <= 8  -> the streaming GEMV (gemv.hip), the decode path;
//   M  > 8  -> MFMA tiles: LDS-DMA staged (gemm_glds.hip: SQ / fp16, K-bytes % 128 == 0), else register staged with
//              in-flight dequantisation (gemm_mfma.hip: weight-only types, odd K), else 8-row GEMV slabs.
#include "dev_utils.h"
#include <cstdlib>
#include "kernels.h"
#include "weight_layout.h"

namespace tllm
{
namespace kernels
{

int launch_gemm_mfma(const GemmParams& p, hipStream_t stream); // gemm_mfma.hip; returns 1 when the shape is unsupported
int launch_gemm_woq(const GemmParams& p, hipStream_t stream);  // gemm_woq.hip (weight-only, dequantisation in the main loop); same
int launch_gemm_glds(const GemmParams& p, hipStream_t stream); // gemm_glds.hip (SQ / fp16, LDS-DMA staged); same convention

namespace
{
using namespace dev;

// u8 (q + 128) / nibble (q + 8) weights [N][ldw] -> fp16 q exactly, [N][K]; one thread per 16 bytes of weights
template <int BITS>
__global__ __launch_bounds__(256) void woq_expand_kernel(uint16_t* out, const char* w, int64_t ldw, int32_t N, int32_t K)
{
    constexpr int EPV = BITS == 8 ? 16 : 32; // elements per 16-byte vector
    const int64_t vecs_per_row = K / EPV;
    const int64_t total = (int64_t) N * vecs_per_row;
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t) gridDim.x * blockDim.x)
    {
        const int64_t n = i / vecs_per_row, v = i % vecs_per_row;
        const uint4 q = *reinterpret_cast<const uint4*>(w + n * ldw + v * 16);
        uint16_t* o = out + n * K + v * EPV;
        const uint32_t ws[4] = {q.x, q.y, q.z, q.w};
        if constexpr (BITS == 8)
        {
            const uint32_t magic = 0x64646464u;
            const h2_t bias = {(_Float16) 1152.f, (_Float16) 1152.f};
            uint32_t r[8];
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                r[2 * j] = h2_as_u32(u32_as_h2(__builtin_amdgcn_perm(magic, ws[j], 0x04010400u)) - bias);
                r[2 * j + 1] = h2_as_u32(u32_as_h2(__builtin_amdgcn_perm(magic, ws[j], 0x04030402u)) - bias);
            }
            *reinterpret_cast<uint4*>(o) = make_uint4(r[0], r[1], r[2], r[3]);
            *reinterpret_cast<uint4*>(o + 8) = make_uint4(r[4], r[5], r[6], r[7]);
        }
        else
        {
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                // nibble order of weight_layout.h: e0 e2 e4 e6 | e1 e3 e5 e7
                const uint32_t m = 0x64006400u, w8 = ws[j] >> 8;
                const h2_t b0 = {(_Float16) 1032.f, (_Float16) 1032.f};
                const h2_t s1 = {(_Float16) 0.0625f, (_Float16) 0.0625f};
                const h2_t b1 = {(_Float16) -72.f, (_Float16) -72.f};
                const uint32_t e01 = h2_as_u32(u32_as_h2((ws[j] & 0x000f000fu) | m) - b0);
                const uint32_t e23 = h2_as_u32(u32_as_h2((ws[j] & 0x00f000f0u) | m) * s1 + b1);
                const uint32_t e45 = h2_as_u32(u32_as_h2((w8 & 0x000f000fu) | m) - b0);
                const uint32_t e67 = h2_as_u32(u32_as_h2((w8 & 0x00f000f0u) | m) * s1 + b1);
                *reinterpret_cast<uint4*>(o + 8 * j) = make_uint4(e01, e23, e45, e67);
            }
        }
    }
}
} // namespace

size_t gemm_woq_scratch_bytes(int32_t N, int32_t K)
{
    return (size_t) N * K * 2;
}

static int gemv_slab(const GemmParams& p, int m0, int rows, hipStream_t stream)
{
    GemvParams g;
    g.wtype = p.wtype;
    g.pro = PRO_NONE;
    g.epi = EPI_NONE;
    g.out_dtype = p.out_dtype;
    g.M = rows;
    g.N = p.N;
    g.K = p.K;
    const int a_es = p.wtype == W_INT8_SQ ? 1 : 2;
    g.x = static_cast<const char*>(p.a) + (int64_t) m0 * p.lda * a_es;
    g.ldx = p.lda;
    g.w = p.w;
    g.ldw = p.ldw;
    g.scale_col = p.scale_col;
    g.scale_row = p.scale_row ? (p.per_token ? p.scale_row + m0 : p.scale_row) : nullptr;
    g.per_channel = p.per_channel;
    g.per_token = p.per_token;
    const int c_es = p.out_dtype == DT_HALF ? 2 : 4;
    g.y = static_cast<char*>(p.c) + (int64_t) m0 * p.ldc * c_es;
    g.ldy = p.ldc;
    return launch_gemv(g, stream);
}

int launch_gemm(const GemmParams& pin, hipStream_t stream)
{
    if (pin.M <= 0)
        return 0;
    if (pin.residual && pin.out_dtype != DT_HALF)
    {
        set_error("gemm: the fused residual needs fp16 output");
        return -1;
    }
    if (pin.silu_gate && (pin.residual || pin.out_dtype != DT_HALF))
    {
        set_error("gemm: the fused SwiGLU gate needs fp16 output and excludes the fused residual");
        return -1;
    }
    if (pin.silu_gate && pin.wtype == W_INT8_SQ)
    {
        // SmoothQuant has its own fused form (launch_gemm_swiglu); here: the plain product, then the pointwise pass
        GemmParams q = pin;
        q.silu_gate = nullptr;
        const int rc = launch_gemm(q, stream);
        return rc ? rc : launch_swiglu(pin.c, pin.silu_gate, pin.c, (int64_t) pin.M * pin.N, stream);
    }
    if (pin.M > 8)
    {
        const bool woq = pin.wtype == W_INT8_WOQ || pin.wtype == W_INT4_WOQ;
        static const bool expand_only = getenv("TLLM_WOQ_EXPAND") != nullptr; // A/B switch: the r02 path (fp16 image in a scratch buffer)
        if (woq && !expand_only)
        {
            // weight-only at prefill sizes: the u8 / nibble tile by LDS-DMA, dequantised between LDS and the MFMA fragments -
            // no fp16 image of the weights, a third of the expanded path's HBM traffic
            const int r = launch_gemm_woq(pin, stream);
            if (r <= 0)
                return r;
        }
        if (woq && pin.scratch && pin.M >= 32 && pin.K % (pin.wtype == W_INT8_WOQ ? 16 : 32) == 0 && pin.K % 64 == 0
            && !(reinterpret_cast<uintptr_t>(pin.w) & 15) && !(pin.ldw & 15))
        {
            // weight-only at prefill sizes: integers -> fp16 once (exact), then the LDS-DMA staged fp16 MFMA kernel with
            // the per-channel scale in its epilogue - the same arithmetic as the register-staged kernel, ~1.5x faster
            const int64_t vecs = (int64_t) pin.N * (pin.K / (pin.wtype == W_INT8_WOQ ? 16 : 32));
            const int grid = (int) ((vecs + 255) / 256 < 8192 ? (vecs + 255) / 256 : 8192);
            if (pin.wtype == W_INT8_WOQ)
                hipLaunchKernelGGL(woq_expand_kernel<8>, dim3(grid), dim3(256), 0, stream, static_cast<uint16_t*>(pin.scratch),
                    static_cast<const char*>(pin.w), pin.ldw, pin.N, pin.K);
            else
                hipLaunchKernelGGL(woq_expand_kernel<4>, dim3(grid), dim3(256), 0, stream, static_cast<uint16_t*>(pin.scratch),
                    static_cast<const char*>(pin.w), pin.ldw, pin.N, pin.K);
            GemmParams e = pin;
            e.wtype = W_FP16;
            e.w = pin.scratch;
            e.ldw = (int64_t) pin.K * 2;
            e.scratch = nullptr;
            const int r = launch_gemm_glds(e, stream);
            if (r <= 0)
                return r;
        }
        const int r = launch_gemm_glds(pin, stream); // fuses the residual in its epilogue
        if (r <= 0)
            return r;
    }
    // the other paths write the plain product; the residual (if any) is added by a pointwise pass afterwards
    GemmParams p = pin;
    p.residual = nullptr;
    p.silu_gate = nullptr;
    if (pin.silu_gate && pin.ldc != pin.N)
    {
        set_error("gemm: a strided SwiGLU gate is only supported by the LDS-DMA kernels (M >= 32, aligned operands)");
        return -1;
    }
    if (pin.residual && (pin.ldc != pin.N || pin.residual == pin.c))
    {
        set_error("gemm: a strided or in-place residual is only supported by the LDS-DMA kernel (SQ / fp16, K bytes %% 128 == 0, M >= 32)");
        return -1;
    }
    auto finish = [&](int rc) {
        if (rc == 0 && pin.residual)
            return launch_add(pin.c, pin.c, pin.residual, (int64_t) pin.M * pin.N, stream);
        if (rc == 0 && pin.silu_gate)
            return launch_swiglu(pin.c, pin.silu_gate, pin.c, (int64_t) pin.M * pin.N, stream); // elementwise: in place is fine
        return rc;
    };
    if (p.M > 8)
    {
        const int r = launch_gemm_mfma(p, stream);
        if (r <= 0)
            return finish(r);
    }
    for (int m0 = 0; m0 < p.M; m0 += 8)
    {
        const int rows = p.M - m0 < 8 ? p.M - m0 : 8;
        if (gemv_slab(p, m0, rows, stream))
            return -1;
    }
    return finish(0);
}

} // namespace kernels
} // namespace tllm
