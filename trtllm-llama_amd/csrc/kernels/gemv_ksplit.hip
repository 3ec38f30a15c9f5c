// Decode GEMV for "few output rows, long K" SmoothQuant projections (LLaMA's down-projection: N = 4096, K = 11008 int8),
// single token:  y[n] = epi( float(sum_k x[k] * W[n,k]) * (s_col[n] * s_row) )
//
// Why a second kernel.  The general kernel (gemv_impl.h) walks a row pair in 4 KiB tiles through a two-tile ring; a wave that
// owns ONE row pair of 11 KiB therefore pays the HBM latency three times in sequence (tile 1 is issued when tile 0 is consumed,
// tile 2 when tile 1 is) - measured 12.3 us against 8.3 us for streaming the same 45 MB.  Here a workgroup owns 8 rows and
// its 4 waves split K (wave w takes the 1 KiB chunks w, w + 4, w + 8): every byte the workgroup needs - 24 weight vectors and
// 3 activation vectors per lane, the epilogue's scale and residual - is requested at t = 0, one memory round trip, then
// v_dot4_i32_i8, a DPP reduction per row, a 128-byte exchange through LDS and the epilogue.  The activations are already
// int8 in memory (written by the SwiGLU epilogue of the previous kernel), so no prologue and no LDS staging of x: a lane
// only ever needs the 16 bytes of x that face its 16 bytes of each row.
//
// Arithmetic identical to the general kernel (exact int32 sum, the same float expression in the epilogue): bit-identical
// results; tests/test_gpu_plugins.py::test_smooth_quant_gemm_exact covers this shape.
#include "dev_utils.h"
#include "gemv_args.h"

namespace tllm
{
namespace kernels
{
using namespace dev;

namespace
{
constexpr int RW = 8; // rows per workgroup

template <int NC> // chunks per wave
__global__ __launch_bounds__(256) void gemv_sq_ksplit_kernel(const GemvParams p, int nchunks)
{
    __shared__ int part[4][RW];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int row0 = blockIdx.x * RW;
    const char* wbase = reinterpret_cast<const char*>(p.w);
    const char* xg = reinterpret_cast<const char*>(p.x);
    // ---- t = 0: everything this workgroup will ever read
    uint4 w[NC][RW], xv[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i)
    {
        const int c = wid + 4 * i;
        const int k = (c * 64 + lane) * 16;
        const bool ok = c < nchunks && k < p.K; // K is a multiple of 16
        const int kc = ok ? k : 0;
        const uint4 xr = *reinterpret_cast<const uint4*>(xg + kc);
        xv[i] = make_uint4(ok ? xr.x : 0u, ok ? xr.y : 0u, ok ? xr.z : 0u, ok ? xr.w : 0u); // zero x: the weights need no mask
#pragma unroll
        for (int r = 0; r < RW; ++r)
        {
            const int row = row0 + r < p.N ? row0 + r : p.N - 1;
            w[i][r] = ld_nt16(wbase + (int64_t) row * p.ldw + kc);
        }
    }
    // epilogue operands of row (tid & 7): every thread loads them (no lane-dependent branch around a load - that would make
    // the compiler fence it), threads 0..7 use them
    float srow = 1.f, res = 0.f;
    const int n = row0 + (tid & 7) < p.N ? row0 + (tid & 7) : p.N - 1;
    const float s0 = reinterpret_cast<const float*>(p.scale_col)[p.per_channel ? n : 0];
    if (p.scale_row) // uniform
        srow = p.scale_row[0];
    if (p.epi == EPI_RESIDUAL) // uniform
        res = h2f(reinterpret_cast<const uint16_t*>(p.residual)[n]);
    // ---- dots, reduction over the 64 lanes, exchange between the 4 K-slices
    int acc[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r)
    {
        int a = 0;
#pragma unroll
        for (int i = 0; i < NC; ++i)
        {
            a = sdot4(w[i][r].x, xv[i].x, a);
            a = sdot4(w[i][r].y, xv[i].y, a);
            a = sdot4(w[i][r].z, xv[i].z, a);
            a = sdot4(w[i][r].w, xv[i].w, a);
        }
        acc[r] = wave_sum(a);
    }
    if (lane == 0)
    {
#pragma unroll
        for (int r = 0; r < RW; ++r)
            part[wid][r] = acc[r];
    }
    __syncthreads();
    if (tid < RW && row0 + tid < p.N)
    {
        const int tot = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
        const float r0 = (float) tot * (s0 * srow);
        if (p.epi == EPI_RESIDUAL)
            reinterpret_cast<uint16_t*>(p.y)[n] = f2h(h2f(f2h(r0)) + res);
        else if (p.out_dtype == DT_HALF)
            reinterpret_cast<uint16_t*>(p.y)[n] = f2h(r0);
        else if (p.out_dtype == DT_FLOAT)
            reinterpret_cast<float*>(p.y)[n] = r0;
        else
            reinterpret_cast<int32_t*>(p.y)[n] = tot;
    }
}
} // namespace

bool gemv_sq_ksplit_applies(const GemvArgs& a)
{
    const GemvParams& p = a.p;
    return p.wtype == W_INT8_SQ && p.M == 1 && p.pro == PRO_NONE && (p.epi == EPI_NONE || p.epi == EPI_RESIDUAL)
        && a.nchunks >= 5 && a.nchunks <= 12 && !p.x_pro_out && !p.dyn_scale_out && !p.per_token
        && (p.out_dtype == DT_HALF || p.out_dtype == DT_FLOAT || p.out_dtype == DT_INT32) && (p.K & 15) == 0;
}

int launch_gemv_sq_ksplit(const GemvArgs& a, hipStream_t stream)
{
    const int blocks = (a.p.N + RW - 1) / RW;
    const int nc = (a.nchunks + 3) / 4;
    if (nc == 2)
        hipLaunchKernelGGL(gemv_sq_ksplit_kernel<2>, dim3(blocks), dim3(256), 0, stream, a.p, a.nchunks);
    else
        hipLaunchKernelGGL(gemv_sq_ksplit_kernel<3>, dim3(blocks), dim3(256), 0, stream, a.p, a.nchunks);
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
