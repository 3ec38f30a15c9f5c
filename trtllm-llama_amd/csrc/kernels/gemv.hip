// Dispatcher of the decode GEMV (kernel: gemv_impl.h; one translation unit per weight type).
#include "gemv_args.h"
#include "weight_layout.h"
#include <cstdint>
#include <cstdlib>

namespace tllm
{
namespace kernels
{

int gemv_tune_blocks_per_cu = 0; // test/bench override: persistent workgroups per CU (0 = occupancy query)

int launch_gemv(const GemvParams& p, hipStream_t stream)
{
    if (p.M < 1 || p.M > 8 || p.N <= 0 || p.K <= 0)
    {
        set_error("gemv: unsupported M=%d N=%d K=%d (1 <= M <= 8)", p.M, p.N, p.K);
        return -1;
    }
    const bool sq = p.wtype == W_INT8_SQ;
    if (sq && p.M >= 2)
    {
        // several sequences: from 5 rows on (tllm_gemv_set_mfma_rows) the matrix-pipe kernel, whose time hardly grows with the rows
        // (gemv_mfma_sq.hip; bit-identical results) - returns 1 at once below its threshold or for a prologue it does not build
        const int r = launch_gemv_mfma_sq(p, stream);
        if (r <= 0)
            return r;
    }
    // The kernel keeps the activation rows of its row bucket (1 / 2 / 4 / 8) in LDS: 8 rows of an fp16 K = 11008 vector are 176 KB -
    // more than a CU has (LLaMA-7B's down-projection at batch 5 .. 8, every LLaMA's at 13B and beyond).  Such a call goes through
    // in slabs of as many rows as fit, each slab streaming the weights again (r04; before, it was refused).
    {
        const int es = sq ? 1 : 2; // bytes per activation element in LDS
        const int64_t kp = (p.K + 31) / 32 * 32;
        int mb = p.M <= 1 ? 1 : (p.M <= 2 ? 2 : (p.M <= 4 ? 4 : 8));
        int fit = mb;
        while (fit > 1 && kRedBytes + (int64_t) fit * kp * es > 160 * 1024)
            fit >>= 1;
        if (fit < mb)
        {
            const bool raw_s8 = sq && p.pro == PRO_NONE;
            for (int m0 = 0; m0 < p.M; m0 += fit)
            {
                GemvParams q = p;
                q.M = p.M - m0 < fit ? p.M - m0 : fit;
                q.x = static_cast<const char*>(p.x) + (int64_t) m0 * p.ldx * (raw_s8 ? 1 : 2);
                const int yes = p.out_dtype == DT_INT8 ? 1 : (p.out_dtype == DT_HALF ? 2 : 4);
                q.y = static_cast<char*>(p.y) + (int64_t) m0 * p.ldy * yes;
                if (p.residual)
                    q.residual = static_cast<const char*>(p.residual) + (int64_t) m0 * p.ldy * 2;
                if (p.per_token && p.scale_row)
                    q.scale_row = p.scale_row + m0;
                if (p.dyn_scale_out)
                    q.dyn_scale_out = p.dyn_scale_out + m0;
                if (p.x_pro_out)
                    q.x_pro_out = static_cast<char*>(p.x_pro_out) + (int64_t) m0 * p.K * es;
                if (launch_gemv(q, stream))
                    return -1;
            }
            return 0;
        }
    }
    const bool swiglu = p.epi == EPI_SWIGLU || p.epi == EPI_SWIGLU_QSTATIC;
    const bool quant_pro = p.pro >= PRO_RMSNORM_QSTATIC && p.pro <= PRO_QDYN;
    if (!sq && quant_pro)
    {
        set_error("gemv: quantising prologue needs W_INT8_SQ");
        return -1;
    }
    int pk;
    switch (p.pro)
    {
    case PRO_NONE: pk = PK_COPY; break;
    case PRO_RMSNORM:
    case PRO_RMSNORM_QSTATIC:
    case PRO_RMSNORM_QDYN: pk = PK_NORM; break;
    case PRO_QSTATIC:
    case PRO_QDYN: pk = PK_QUANT; break;
    default: set_error("gemv: bad prologue %d", p.pro); return -1;
    }
    if (sq && p.pro == PRO_RMSNORM)
    {
        set_error("gemv: W_INT8_SQ needs a quantising prologue (or s8 activations with PRO_NONE)");
        return -1;
    }
    if ((reinterpret_cast<uintptr_t>(p.w) & 15) || (p.ldw & 15))
    {
        set_error("gemv: weight pointer / row stride must be 16-byte aligned (ldw=%lld)", (long long) p.ldw);
        return -1;
    }
    int vec = 8;
    switch (p.wtype)
    {
    case W_FP16: vec = 8; break;
    case W_INT8_WOQ:
    case W_INT8_SQ: vec = 16; break;
    case W_INT4_WOQ: vec = 32; break;
    default: set_error("gemv: bad wtype %d", p.wtype); return -1;
    }
    // activations: 16-byte vectors kept in registers by 256 threads
    const bool raw_s8 = sq && p.pro == PRO_NONE;
    const int xvec = raw_s8 ? 16 : 8;
    const int nxv_lim = ((pk == PK_COPY || pk == PK_QUANT) && !swiglu) ? kNXVLarge : kNXVMax; // (gemv_args.h)
    if ((p.K % xvec) || p.K > 256 * xvec * nxv_lim)
    {
        set_error("gemv: K=%d must be a multiple of %d and <= %d", p.K, xvec, 256 * xvec * nxv_lim);
        return -1;
    }
    if (((reinterpret_cast<uintptr_t>(p.x) & 15) || ((p.ldx * (raw_s8 ? 1 : 2)) & 15)))
    {
        set_error("gemv: activation pointer / row stride must be 16-byte aligned");
        return -1;
    }
    if (pk == PK_NORM && (reinterpret_cast<uintptr_t>(p.gamma) & 15))
    {
        set_error("gemv: RMSNorm weight must be 16-byte aligned");
        return -1;
    }
    GemvArgs a;
    a.p = p;
    a.Kp = (int32_t) layout::round_up(p.K, vec);
    if (layout::row_bytes(p.wtype, p.K) > p.ldw)
    {
        set_error("gemv: ldw=%lld too small for K=%d", (long long) p.ldw, p.K);
        return -1;
    }
    a.nchunks = (a.Kp + 64 * vec - 1) / (64 * vec);
    a.ngroups = swiglu ? p.N : (p.N + R - 1) / R;
    if (gemv_ksplit_applies(a))
        return launch_gemv_ksplit(a, stream);
    switch (p.wtype)
    {
    case W_FP16: return launch_gemv_fp16(a, pk, swiglu, stream);
    case W_INT8_WOQ: return launch_gemv_woq8(a, pk, swiglu, stream);
    case W_INT4_WOQ: return launch_gemv_woq4(a, pk, swiglu, stream);
    default: return launch_gemv_sq(a, pk, swiglu, stream);
    }
}

} // namespace kernels
} // namespace tllm
