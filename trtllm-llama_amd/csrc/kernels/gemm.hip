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

using namespace dev;

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
        if (pin.ldc != pin.N) // the pointwise pass runs over M * N contiguous elements
        {
            set_error("gemm: a strided SwiGLU gate is not supported with SmoothQuant weights (ldc %lld != N %d)", (long long) pin.ldc, pin.N);
            return -1;
        }
        GemmParams q = pin;
        q.silu_gate = nullptr;
        const int rc = launch_gemm(q, stream);
        return rc ? rc : launch_swiglu(pin.c, pin.silu_gate, pin.c, (int64_t) pin.M * pin.N, stream);
    }
    if (pin.M > 8)
    {
        const bool woq = pin.wtype == W_INT8_WOQ || pin.wtype == W_INT4_WOQ;
        if (woq)
        {
            // weight-only at prefill sizes: the u8 / nibble tile by LDS-DMA, dequantised between LDS and the MFMA fragments -
            // no fp16 image of the weights (r01 / r02 expanded the matrix to fp16 in a scratch buffer on every call: 3 N K bytes
            // of HBM traffic the algorithm does not have; removed in r05, profiles/r03_woq_gemm_sweep.txt).  Shapes it does not
            // serve (K % 64 != 0, M < 32) take the register-staged kernel below.
            const int r = launch_gemm_woq(pin, stream);
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
