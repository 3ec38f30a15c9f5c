// GEMM entry point of the Gemm / SmoothQuantGemm / WeightOnlyQuantMatmul plugins.
//   M <= 8  -> the streaming GEMV (gemv.hip), the decode path;
//   M  > 8  -> MFMA tiles: LDS-DMA staged (gemm_glds.hip: SQ / fp16, K-bytes % 128 == 0), else register staged with
//              in-flight dequantisation (gemm_mfma.hip: weight-only types, odd K), else 8-row GEMV slabs.
#include "kernels.h"

namespace tllm
{
namespace kernels
{

int launch_gemm_mfma(const GemmParams& p, hipStream_t stream); // gemm_mfma.hip; returns 1 when the shape is unsupported
int launch_gemm_glds(const GemmParams& p, hipStream_t stream); // gemm_glds.hip (SQ / fp16, LDS-DMA staged); same convention

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
    if (pin.M > 8)
    {
        const int r = launch_gemm_glds(pin, stream); // fuses the residual in its epilogue
        if (r <= 0)
            return r;
    }
    // the other paths write the plain product; the residual (if any) is added by a pointwise pass afterwards
    GemmParams p = pin;
    p.residual = nullptr;
    if (pin.residual && (pin.ldc != pin.N || pin.residual == pin.c))
    {
        set_error("gemm: a strided or in-place residual is only supported by the LDS-DMA kernel (SQ / fp16, K bytes %% 128 == 0, M >= 32)");
        return -1;
    }
    auto finish = [&](int rc) {
        if (rc == 0 && pin.residual)
            return launch_add(pin.c, pin.c, pin.residual, (int64_t) pin.M * pin.N, stream);
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
