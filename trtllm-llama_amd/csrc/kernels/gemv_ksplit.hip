// Decode GEMV for "few output rows, long K" projections (LLaMA's down-projection: N = 4096, K = 11008), single token:
//     y[n] = epi( scale(n) * sum_k x[k] * W[n,k] )          every weight type of the general kernel (gemv_impl.h)
//
// Why a second kernel.  The general kernel walks a row pair in 4 KiB tiles through a two-tile ring; a wave that owns ONE
// row pair of 11 KiB therefore pays the HBM latency three times in sequence (tile 1 is issued when tile 0 is consumed,
// tile 2 when tile 1 is) - measured 12.3 us against 8.3 us for streaming the same 45 MB.  Here a workgroup owns RW (2 - 4) rows and
// its 4 waves split K (wave w takes the 1 KiB chunks w, w + 4, w + 8, ...): every byte the workgroup needs - up to 24 weight
// vectors per lane, the activations that face them, the epilogue's scale and residual - is requested at t = 0, ONE memory
// round trip, then the dot products, a DPP reduction per row, a small exchange through LDS and the epilogue.  The
// activations arrive in the operand type (s8 written by the SwiGLU epilogue of the previous kernel, or fp16), so there is no
// prologue and no LDS staging of x: a lane only ever needs the activations that face its 16 bytes of each row.
//
// SmoothQuant: exact int32 sum and the general kernel's epilogue expression -> bit-identical results
// (tests/test_gpu_plugins.py::test_smooth_quant_gemm_exact covers these shapes).  fp16 / weight-only: fp32 accumulation in a
// different order than the general kernel, inside the same tolerances.
#include "gemv_impl.h"
#include <cstdlib>

namespace tllm
{
namespace kernels
{
namespace
{

template <int WT>
struct KSplit
{
    static constexpr int VEC = WTraits<WT>::VEC;                  // weights per 16-byte vector
    static constexpr bool SQ = WTraits<WT>::IS_SQ;
    static constexpr int XV = SQ ? 1 : VEC / 8;                   // 16-byte activation vectors per weight vector
    // rows per workgroup, measured per weight type on the 7B down-projection (us, rows 8 / 4 / 2): SmoothQuant 9.7 / 9.4 / 9.0,
    // weight-only int8 10.05 / 9.36 / 9.85, int4 8.55 / 7.74 / -, fp16 - / 15.07 / 15.37
    static constexpr int RW = WT == W_INT8_SQ ? 2 : 4;
    static constexpr int NCMAX = WT == W_FP16 ? 6 : (WT == W_INT4_WOQ ? 2 : 3); // chunks per wave: 24 / 16 weight vectors per lane
};

template <int WT, int NC>
__global__ __launch_bounds__(256) void gemv_ksplit_kernel(const GemvParams p, int nchunks)
{
    using T = KSplit<WT>;
    constexpr int VEC = T::VEC, XV = T::XV, RW = T::RW;
    constexpr bool SQ = T::SQ;
    using acc_t = typename std::conditional<SQ, int, float>::type;
    __shared__ acc_t part[4][RW];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int row0 = blockIdx.x * RW;
    const char* wbase = reinterpret_cast<const char*>(p.w);
    const char* xg = reinterpret_cast<const char*>(p.x);
    constexpr int XES = SQ ? 1 : 2; // bytes per activation
    // ---- t = 0: everything this workgroup will ever read
    uint4 w[NC][RW], xv[NC][XV];
#pragma unroll
    for (int i = 0; i < NC; ++i)
    {
        const int c = wid + 4 * i;
        const int k = (c * 64 + lane) * VEC;   // first element of this lane's weight vector
        const bool ok = c < nchunks && k < p.K; // K is a multiple of VEC
        const int kc = ok ? k : 0;
#pragma unroll
        for (int v = 0; v < XV; ++v)
        {
            const uint4 xr = *reinterpret_cast<const uint4*>(xg + (int64_t) kc * XES + v * 16);
            // zero activations beyond K: the weights then need no mask (their neutral-element encodings differ per type)
            xv[i][v] = make_uint4(ok ? xr.x : 0u, ok ? xr.y : 0u, ok ? xr.z : 0u, ok ? xr.w : 0u);
        }
        const int64_t wb = (int64_t) (c < nchunks ? c : 0) * 1024 + lane * 16;
        const int64_t woff = ok ? wb : 0;
#pragma unroll
        for (int r = 0; r < RW; ++r)
        {
            const int row = row0 + r < p.N ? row0 + r : p.N - 1;
            w[i][r] = ld_nt16(wbase + (int64_t) row * p.ldw + woff);
        }
    }
    // epilogue operands of row (tid % RW): every thread loads them (no lane-dependent branch around a load - that would make
    // the compiler fence it), threads 0 .. RW - 1 use them
    float s0 = 1.f, srow = 1.f;
    uint16_t s0_bits = 0, res_bits = 0; // fp16 operands stay raw until the epilogue (a conversion next to the load = a wait)
    const int n = row0 + (tid % RW) < p.N ? row0 + (tid % RW) : p.N - 1;
    if constexpr (SQ)
    {
        s0 = reinterpret_cast<const float*>(p.scale_col)[p.per_channel ? n : 0];
        if (p.scale_row) // uniform
            srow = p.scale_row[0];
    }
    else if constexpr (WT != W_FP16)
        s0_bits = reinterpret_cast<const uint16_t*>(p.scale_col)[n];
    // unconditional: a load inside a conditional block ends in a full s_waitcnt (without a residual: y's own address, ignored)
    res_bits = reinterpret_cast<const uint16_t*>(p.epi == EPI_RESIDUAL ? p.residual : p.y)[n];
    // ---- dots, reduction over the 64 lanes, exchange between the 4 K-slices
#pragma unroll
    for (int r = 0; r < RW; ++r)
    {
        acc_t a = 0;
#pragma unroll
        for (int i = 0; i < NC; ++i)
        {
            if constexpr (SQ)
                a = dot_sq(w[i][r], xv[i][0], a);
            else if constexpr (WT == W_FP16)
                a = dot_fp16(w[i][r], xv[i][0], a);
            else if constexpr (WT == W_INT8_WOQ)
                a = dot_woq8(w[i][r], xv[i][0], xv[i][1], a);
            else
            {
                a = dot_u4x8(w[i][r].x, xv[i][0], a);
                a = dot_u4x8(w[i][r].y, xv[i][1], a);
                a = dot_u4x8(w[i][r].z, xv[i][2], a);
                a = dot_u4x8(w[i][r].w, xv[i][3], a);
            }
        }
        a = wave_sum(a);
        if (lane == 0)
            part[wid][r] = a;
    }
    __syncthreads();
    if (tid < RW && row0 + tid < p.N)
    {
        const acc_t tot = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
        if constexpr (!SQ && WT != W_FP16)
            s0 = h2f(s0_bits);
        const float res = h2f(res_bits);
        const float r0 = (float) tot * (s0 * srow); // the general kernel's expression (srow = 1 unless SmoothQuant)
        if (p.epi == EPI_RESIDUAL)
            reinterpret_cast<uint16_t*>(p.y)[n] = f2h(h2f(f2h(r0)) + res);
        else if (p.out_dtype == DT_HALF)
            reinterpret_cast<uint16_t*>(p.y)[n] = f2h(r0);
        else if (p.out_dtype == DT_FLOAT)
            reinterpret_cast<float*>(p.y)[n] = r0;
        else
            reinterpret_cast<int32_t*>(p.y)[n] = f2i32_rni_sat(r0);
    }
}

template <int WT, int NC>
void launch_nc(const GemvArgs& a, int per_wave, hipStream_t stream)
{
    if constexpr (NC > 1)
        if (per_wave < NC)
            return launch_nc<WT, NC - 1>(a, per_wave, stream);
    constexpr int RW = KSplit<WT>::RW;
    hipLaunchKernelGGL((gemv_ksplit_kernel<WT, NC>), dim3((a.p.N + RW - 1) / RW), dim3(256), 0, stream, a.p, a.nchunks);
}

template <int WT>
bool applies_wt(const GemvArgs& a)
{
    // more than one 4-chunk tile per row (otherwise the general kernel is already one-shot), few enough chunks for the
    // registers, few enough rows that the general kernel would give every wave a single row pair
    return a.nchunks > 4 && a.nchunks <= 4 * KSplit<WT>::NCMAX && (a.p.K % KSplit<WT>::VEC) == 0 && a.p.N <= 8192;
}

} // namespace

bool gemv_ksplit_applies(const GemvArgs& a)
{
    const GemvParams& p = a.p;
    if (p.M != 1 || p.pro != PRO_NONE || !(p.epi == EPI_NONE || p.epi == EPI_RESIDUAL) || p.x_pro_out || p.dyn_scale_out || p.per_token)
        return false;
    if (!(p.out_dtype == DT_HALF || p.out_dtype == DT_FLOAT || (p.out_dtype == DT_INT32 && p.wtype == W_INT8_SQ)))
        return false;
    switch (p.wtype)
    {
    case W_FP16: return applies_wt<W_FP16>(a);
    case W_INT8_WOQ: return applies_wt<W_INT8_WOQ>(a);
    case W_INT4_WOQ: return applies_wt<W_INT4_WOQ>(a);
    case W_INT8_SQ: return applies_wt<W_INT8_SQ>(a);
    default: return false;
    }
}

int launch_gemv_ksplit(const GemvArgs& a, hipStream_t stream)
{
    const int per_wave = (a.nchunks + 3) / 4;
    switch (a.p.wtype)
    {
    case W_FP16: launch_nc<W_FP16, KSplit<W_FP16>::NCMAX>(a, per_wave, stream); break;
    case W_INT8_WOQ: launch_nc<W_INT8_WOQ, KSplit<W_INT8_WOQ>::NCMAX>(a, per_wave, stream); break;
    case W_INT4_WOQ: launch_nc<W_INT4_WOQ, KSplit<W_INT4_WOQ>::NCMAX>(a, per_wave, stream); break;
    default: launch_nc<W_INT8_SQ, KSplit<W_INT8_SQ>::NCMAX>(a, per_wave, stream); break;
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess)
    {
        set_error("gemv (k-split) launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}

} // namespace kernels
} // namespace tllm
