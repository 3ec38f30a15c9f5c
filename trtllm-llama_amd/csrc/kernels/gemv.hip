// Skinny GEMM / GEMV for decode (M <= 8): the HBM-bound heart of the per-token path.
//
//   y[m,n] = epi( scale(n,m) * sum_k pro(x)[m,k] * W[n,k] )
//
// One wave owns R weight rows and streams them with 16-byte non-temporal loads (1 KiB per wave
// instruction, U k-chunks in flight per row); pro(x) is built once per workgroup in LDS (RMSNorm and/or
// int8 quantisation fused in, so the normalised / quantised activation never goes to HBM); the dot
// products use v_dot2_f32_f16 (fp16 x fp16 -> fp32) or v_dot4_i32_i8 (SmoothQuant, exact int32);
// reduction over the 64 lanes by cross-lane shuffles; residual-add / SwiGLU / quantising epilogues fused.
//
// Reference semantics: A7 P/gemmPlugin/gemmPlugin.cpp:121-190; A8 K/weightOnlyMatrixVectorMultiplication.cu:136-277
// (y = sum_k x[k] * (q[k,n] * s[n])); A10 cutlass_extensions/.../epilogue_per_row_per_col_scale.h:279-347
// (C = cvt(float(acc_i32) * (alpha_col * alpha_row))); A5 PY/functional.py:3195-3219; A6 PY/layers/mlp.py:68-73;
// A11 K/quantization.cu:31-118, K/layernormKernels.cu:146-183.
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

struct GemvArgs
{
    GemvParams p;
    int32_t Kp;      // K rounded up to the weight vector width
    int32_t nchunks; // ceil(Kp / (64 * VEC))
    int32_t ngroups; // row groups (one per wave-iteration)
    int32_t xh_bytes; // bytes of the fp16 staging region per row m (0 if absent)
};

template <int WT>
struct WTraits;
template <>
struct WTraits<W_FP16>
{
    static constexpr int VEC = 8;
    static constexpr bool IS_SQ = false;
};
template <>
struct WTraits<W_INT8_WOQ>
{
    static constexpr int VEC = 16;
    static constexpr bool IS_SQ = false;
};
template <>
struct WTraits<W_INT4_WOQ>
{
    static constexpr int VEC = 32;
    static constexpr bool IS_SQ = false;
};
template <>
struct WTraits<W_INT8_SQ>
{
    static constexpr int VEC = 16;
    static constexpr bool IS_SQ = true;
};

// ---- per-16-byte dot products -----------------------------------------------------------------

// fp16 weights: 8 halfs vs 8 halfs of x
__device__ __forceinline__ float dot_fp16(const uint4& w, const uint4& x, float acc)
{
    acc = dot2(w.x, x.x, acc);
    acc = dot2(w.y, x.y, acc);
    acc = dot2(w.z, x.z, acc);
    acc = dot2(w.w, x.w, acc);
    return acc;
}

// u8 (q+128) weights: 4 bytes -> two fp16 pairs via the 0x6400 | b splice (1024 + b is exact in fp16)
__device__ __forceinline__ float dot_u8x4(uint32_t w, uint32_t x01, uint32_t x23, float acc)
{
    const uint32_t magic = 0x64646464u;
    uint32_t lo = __builtin_amdgcn_perm(magic, w, 0x04010400u); // {1024+b0, 1024+b1}
    uint32_t hi = __builtin_amdgcn_perm(magic, w, 0x04030402u); // {1024+b2, 1024+b3}
    const h2_t bias = {(_Float16) 1152.f, (_Float16) 1152.f};   // 1024 + 128
    h2_t l = u32_as_h2(lo) - bias;
    h2_t h = u32_as_h2(hi) - bias;
    acc = __builtin_amdgcn_fdot2(l, u32_as_h2(x01), acc, false);
    acc = __builtin_amdgcn_fdot2(h, u32_as_h2(x23), acc, false);
    return acc;
}

__device__ __forceinline__ float dot_woq8(const uint4& w, const uint4& xa, const uint4& xb, float acc)
{
    acc = dot_u8x4(w.x, xa.x, xa.y, acc);
    acc = dot_u8x4(w.y, xa.z, xa.w, acc);
    acc = dot_u8x4(w.z, xb.x, xb.y, acc);
    acc = dot_u8x4(w.w, xb.z, xb.w, acc);
    return acc;
}

// 8 nibbles (layout of weight_layout.h) vs 8 halfs of x (one uint4)
__device__ __forceinline__ float dot_u4x8(uint32_t w, const uint4& x, float acc)
{
    const uint32_t m = 0x64006400u;
    const uint32_t w8 = w >> 8;
    h2_t e01 = u32_as_h2((w & 0x000f000fu) | m);
    h2_t e23 = u32_as_h2((w & 0x00f000f0u) | m);
    h2_t e45 = u32_as_h2((w8 & 0x000f000fu) | m);
    h2_t e67 = u32_as_h2((w8 & 0x00f000f0u) | m);
    const h2_t b0 = {(_Float16) 1032.f, (_Float16) 1032.f}; // 1024 + 8
    const h2_t s1 = {(_Float16) 0.0625f, (_Float16) 0.0625f};
    const h2_t b1 = {(_Float16) -72.f, (_Float16) -72.f}; // (1024 + 16 n) / 16 - 72 = n - 8
    e01 = e01 - b0;
    e45 = e45 - b0;
    e23 = e23 * s1 + b1;
    e67 = e67 * s1 + b1;
    acc = __builtin_amdgcn_fdot2(e01, u32_as_h2(x.x), acc, false);
    acc = __builtin_amdgcn_fdot2(e23, u32_as_h2(x.y), acc, false);
    acc = __builtin_amdgcn_fdot2(e45, u32_as_h2(x.z), acc, false);
    acc = __builtin_amdgcn_fdot2(e67, u32_as_h2(x.w), acc, false);
    return acc;
}

__device__ __forceinline__ int dot_sq(const uint4& w, const uint4& x, int acc)
{
    acc = sdot4(w.x, x.x, acc);
    acc = sdot4(w.y, x.y, acc);
    acc = sdot4(w.z, x.z, acc);
    acc = sdot4(w.w, x.w, acc);
    return acc;
}

__device__ __forceinline__ float silu_mul_fp16(float g, float u)
{
    // fp16 rounding points of the reference graph: inter = fc(x) (fp16) ; a = inter * sigmoid(inter) (fp16) ;
    // out = a * gate(x) (fp16)   (PY/layers/mlp.py:68-73, PY/functional.py:521-532)
    const float g16 = h2f(f2h(g));
    const float u16 = h2f(f2h(u));
    const float a = h2f(f2h(g16 / (1.f + __expf(-g16))));
    return h2f(f2h(a * u16));
}

// ---- the kernel --------------------------------------------------------------------------------
// LDS map: [0,128) reduction scratch | xh: MB rows of Kp fp16 (absent for raw-s8 input) | xq: MB rows of Kp s8 (SQ)
template <int WT, int R, int U, int MB>
__global__ __launch_bounds__(256) void gemv_kernel(const GemvArgs a)
{
    using TR = WTraits<WT>;
    constexpr int VEC = TR::VEC;
    constexpr bool SQ = TR::IS_SQ;
    using acc_t = typename std::conditional<SQ, int, float>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GemvParams& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int K = p.K, Kp = a.Kp;
    float* red = reinterpret_cast<float*>(smem);
    uint16_t* xh = reinterpret_cast<uint16_t*>(smem + 128);
    int8_t* xq = reinterpret_cast<int8_t*>(smem + 128 + (size_t) a.xh_bytes * MB);
    const bool x_is_half = !(SQ && p.pro == PRO_NONE);
    const bool do_norm = p.pro == PRO_RMSNORM || p.pro == PRO_RMSNORM_QSTATIC || p.pro == PRO_RMSNORM_QDYN;
    const bool q_static = p.pro == PRO_RMSNORM_QSTATIC || p.pro == PRO_QSTATIC;
    const bool q_dyn = p.pro == PRO_RMSNORM_QDYN || p.pro == PRO_QDYN;
    float row_scale[MB]; // per-token dequant scale when the prologue quantises dynamically

    // ------------------------------------------------------------------ prologue: build pro(x) in LDS
#pragma unroll
    for (int m = 0; m < MB; ++m)
    {
        row_scale[m] = 1.f;
        if (m >= p.M)
            continue;
        if (x_is_half)
        {
            const uint16_t* xg = reinterpret_cast<const uint16_t*>(p.x) + (int64_t) m * p.ldx;
            uint16_t* xs = xh + (size_t) m * Kp;
            float ss = 0.f;
            const bool vec_ok = ((K & 7) == 0) && ((p.ldx & 7) == 0) && ((reinterpret_cast<uintptr_t>(p.x) & 15) == 0);
            if (vec_ok)
            {
                for (int k = tid * 8; k < Kp; k += 256 * 8)
                {
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (k < K)
                        v = *reinterpret_cast<const uint4*>(xg + k);
                    *reinterpret_cast<uint4*>(xs + k) = v;
                    if (do_norm)
                    {
                        const uint32_t ws[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                        {
                            h2_t h = u32_as_h2(ws[j]);
                            const float f0 = (float) h.x, f1 = (float) h.y;
                            ss += f0 * f0 + f1 * f1;
                        }
                    }
                }
            }
            else
            {
                for (int k = tid; k < Kp; k += 256)
                {
                    const uint16_t b = k < K ? xg[k] : (uint16_t) 0;
                    xs[k] = b;
                    const float f = h2f(b);
                    ss += f * f;
                }
            }
            float amax = 0.f;
            if (do_norm)
            {
                ss = block_sum(ss, red);
                const float inv = 1.0f / sqrtf(ss / (float) K + p.eps);
                const uint16_t* g = reinterpret_cast<const uint16_t*>(p.gamma);
                for (int k = tid; k < K; k += 256)
                {
                    const float n16 = h2f(f2h(h2f(xs[k]) * inv));
                    const uint16_t yb = f2h(n16 * h2f(g[k]));
                    xs[k] = yb;
                    amax = fmaxf(amax, fabsf(h2f(yb)));
                }
            }
            else
            {
                __syncthreads();
                if (q_dyn)
                    for (int k = tid; k < K; k += 256)
                        amax = fmaxf(amax, fabsf(h2f(xs[k])));
            }
            if (SQ && (q_static || q_dyn))
            {
                float qs;
                if (q_dyn)
                {
                    amax = block_max(amax, red);
                    amax = fmaxf(amax, h2f(f2h(1e-6f))); // T localMax = 1e-6f (K/quantization.cu:101)
                    qs = 127.f / amax;
                    row_scale[m] = amax / 127.f;
                    if (blockIdx.x == 0 && tid == 0 && p.dyn_scale_out)
                        p.dyn_scale_out[m] = amax / 127.f;
                }
                else
                {
                    __syncthreads();
                    qs = p.act_scale[0];
                }
                int8_t* qd = xq + (size_t) m * Kp;
                for (int k = tid; k < Kp; k += 256)
                    qd[k] = k < K ? f2i8_rni_sat(h2f(xs[k]) * qs) : (int8_t) 0;
            }
            if (blockIdx.x == 0 && p.x_pro_out && p.pro != PRO_NONE)
            {
                __syncthreads();
                if (SQ)
                {
                    int8_t* o = reinterpret_cast<int8_t*>(p.x_pro_out) + (int64_t) m * K;
                    const int8_t* qd = xq + (size_t) m * Kp;
                    for (int k = tid; k < K; k += 256)
                        o[k] = qd[k];
                }
                else
                {
                    uint16_t* o = reinterpret_cast<uint16_t*>(p.x_pro_out) + (int64_t) m * K;
                    for (int k = tid; k < K; k += 256)
                        o[k] = xs[k];
                }
            }
        }
        else
        {
            // raw s8 activations (SmoothQuantGemm plugin input 0)
            const int8_t* xg = reinterpret_cast<const int8_t*>(p.x) + (int64_t) m * p.ldx;
            int8_t* qd = xq + (size_t) m * Kp;
            const bool vec_ok = ((K & 15) == 0) && ((p.ldx & 15) == 0) && ((reinterpret_cast<uintptr_t>(p.x) & 15) == 0);
            if (vec_ok)
            {
                for (int k = tid * 16; k < Kp; k += 256 * 16)
                    *reinterpret_cast<uint4*>(qd + k) = *reinterpret_cast<const uint4*>(xg + k);
            }
            else
            {
                for (int k = tid; k < Kp; k += 256)
                    qd[k] = k < K ? xg[k] : (int8_t) 0;
            }
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ main loop over row groups
    const bool swiglu = p.epi == EPI_SWIGLU || p.epi == EPI_SWIGLU_QSTATIC;
    constexpr int OUTS = R; // outputs per group when !swiglu; R/2 when swiglu
    const char* wbase = reinterpret_cast<const char*>(p.w);
    const int lane_kbyte = lane * 16; // byte offset of this lane's vector inside a chunk row

    for (int g = blockIdx.x * 4 + wid; g < a.ngroups; g += gridDim.x * 4)
    {
        // weight rows of this group
        int64_t rowoff[R];
        bool rvalid[R];
#pragma unroll
        for (int r = 0; r < R; ++r)
        {
            int row;
            if (swiglu)
            {
                const int o = g * (R / 2) + (r % (R / 2 > 0 ? R / 2 : 1));
                rvalid[r] = o < p.N;
                row = (r < R / 2) ? o : p.N + o;
            }
            else
            {
                row = g * R + r;
                rvalid[r] = row < p.N;
            }
            rowoff[r] = rvalid[r] ? (int64_t) row * p.ldw : 0;
        }

        acc_t acc[R][MB];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int m = 0; m < MB; ++m)
                acc[r][m] = 0;

        for (int c = 0; c < a.nchunks; c += U)
        {
            uint4 wv[U][R];
#pragma unroll
            for (int u = 0; u < U; ++u)
            {
                const int kv = (c + u) * 64 + lane; // vector index along k
                const bool ok = kv * VEC < Kp;
#pragma unroll
                for (int r = 0; r < R; ++r)
                {
                    wv[u][r] = make_uint4(0, 0, 0, 0);
                    if (ok)
                        wv[u][r] = ld_nt16(wbase + rowoff[r] + (int64_t) (c + u) * 1024 + lane_kbyte);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
            {
                const int k0 = ((c + u) * 64 + lane) * VEC;
                if (k0 < Kp)
                {
#pragma unroll
                    for (int m = 0; m < MB; ++m)
                    {
                        if constexpr (WT == W_FP16)
                        {
                            const uint4 xv = *reinterpret_cast<const uint4*>(xh + (size_t) m * Kp + k0);
#pragma unroll
                            for (int r = 0; r < R; ++r)
                                acc[r][m] = dot_fp16(wv[u][r], xv, acc[r][m]);
                        }
                        else if constexpr (WT == W_INT8_WOQ)
                        {
                            const uint4 xa = *reinterpret_cast<const uint4*>(xh + (size_t) m * Kp + k0);
                            const uint4 xb = *reinterpret_cast<const uint4*>(xh + (size_t) m * Kp + k0 + 8);
#pragma unroll
                            for (int r = 0; r < R; ++r)
                                acc[r][m] = dot_woq8(wv[u][r], xa, xb, acc[r][m]);
                        }
                        else if constexpr (WT == W_INT4_WOQ)
                        {
                            const uint4* xp = reinterpret_cast<const uint4*>(xh + (size_t) m * Kp + k0);
                            const uint4 x0 = xp[0], x1 = xp[1], x2 = xp[2], x3 = xp[3];
#pragma unroll
                            for (int r = 0; r < R; ++r)
                            {
                                float t = acc[r][m];
                                t = dot_u4x8(wv[u][r].x, x0, t);
                                t = dot_u4x8(wv[u][r].y, x1, t);
                                t = dot_u4x8(wv[u][r].z, x2, t);
                                t = dot_u4x8(wv[u][r].w, x3, t);
                                acc[r][m] = t;
                            }
                        }
                        else
                        {
                            const uint4 xv = *reinterpret_cast<const uint4*>(xq + (size_t) m * Kp + k0);
#pragma unroll
                            for (int r = 0; r < R; ++r)
                                acc[r][m] = dot_sq(wv[u][r], xv, acc[r][m]);
                        }
                    }
                }
            }
        }

        // ---- cross-lane reduction: every lane ends with every total
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int m = 0; m < MB; ++m)
                acc[r][m] = wave_sum(acc[r][m]);

        // ---- epilogue: lane (o * MB + m) finishes output o of row m
        const int nouts = swiglu ? R / 2 : OUTS;
        float v0 = 0.f, v1 = 0.f; // scaled accumulators picked by this lane (v1: the "up" row for swiglu)
        int ai = 0;
        int my_o = -1, my_m = 0;
#pragma unroll
        for (int o = 0; o < R; ++o)
        {
#pragma unroll
            for (int m = 0; m < MB; ++m)
            {
                if (o < nouts && lane == o * MB + m)
                {
                    my_o = o;
                    my_m = m;
                    ai = (int) acc[o][m];
                    v0 = (float) acc[o][m];
                    if (swiglu)
                        v1 = (float) acc[(o + R / 2) % R][m];
                }
            }
        }
        if (my_o >= 0 && my_m < p.M)
        {
            const int n = swiglu ? g * (R / 2) + my_o : g * R + my_o;
            if (n < p.N)
            {
                // column / row scales
                float s0 = 1.f, s1 = 1.f;
                if constexpr (WT == W_INT8_WOQ || WT == W_INT4_WOQ)
                {
                    const uint16_t* sc = reinterpret_cast<const uint16_t*>(p.scale_col);
                    s0 = h2f(sc[n]);
                    if (swiglu)
                        s1 = h2f(sc[p.N + n]);
                }
                else if constexpr (SQ)
                {
                    const float* sc = reinterpret_cast<const float*>(p.scale_col);
                    const float sr = q_dyn ? row_scale[0] : (p.per_token ? p.scale_row[my_m] : p.scale_row[0]);
                    float srm = sr;
                    if (q_dyn)
                    {
#pragma unroll
                        for (int m = 0; m < MB; ++m)
                            if (m == my_m)
                                srm = row_scale[m];
                    }
                    s0 = (p.per_channel ? sc[n] : sc[0]) * srm;
                    if (swiglu)
                        s1 = (p.per_channel ? sc[p.N + n] : sc[0]) * srm;
                }
                const float r0 = v0 * s0;
                const int64_t oidx = (int64_t) my_m * p.ldy + n;
                if (p.epi == EPI_NONE)
                {
                    if (p.out_dtype == DT_HALF)
                        reinterpret_cast<uint16_t*>(p.y)[oidx] = f2h(r0);
                    else if (p.out_dtype == DT_FLOAT)
                        reinterpret_cast<float*>(p.y)[oidx] = r0;
                    else
                        reinterpret_cast<int32_t*>(p.y)[oidx] = SQ ? ai : (int32_t) r0;
                }
                else if (p.epi == EPI_RESIDUAL)
                {
                    const float res = h2f(reinterpret_cast<const uint16_t*>(p.residual)[oidx]);
                    reinterpret_cast<uint16_t*>(p.y)[oidx] = f2h(h2f(f2h(r0)) + res);
                }
                else
                {
                    const float o16 = silu_mul_fp16(r0, v1 * s1);
                    if (p.epi == EPI_SWIGLU)
                        reinterpret_cast<uint16_t*>(p.y)[oidx] = f2h(o16);
                    else
                        reinterpret_cast<int8_t*>(p.y)[oidx] = f2i8_rni_sat(o16 * p.epi_scale[0]);
                }
            }
        }
    }
}

template <int WT, int R, int U, int MB>
int launch_inst(const GemvArgs& a, int blocks, size_t smem, hipStream_t stream)
{
    auto kfn = gemv_kernel<WT, R, U, MB>;
    if (smem > 64 * 1024)
    {
        static bool attr_done = false;
        if (!attr_done)
        {
            (void) hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_done = true;
        }
    }
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(256), smem, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
    {
        set_error("gemv launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}

template <int WT, int R, int U>
int launch_mb(const GemvArgs& a, int blocks, size_t smem_per_m, hipStream_t stream)
{
    const int M = a.p.M;
    if (M <= 1)
        return launch_inst<WT, R, U, 1>(a, blocks, 128 + smem_per_m, stream);
    if (M <= 2)
        return launch_inst<WT, R, U, 2>(a, blocks, 128 + 2 * smem_per_m, stream);
    if (M <= 4)
        return launch_inst<WT, R, U, 4>(a, blocks, 128 + 4 * smem_per_m, stream);
    return launch_inst<WT, R, U, 8>(a, blocks, 128 + 8 * smem_per_m, stream);
}

} // namespace

int gemv_tune_r = 0; // test/bench override: rows per wave (0 = heuristic)

int launch_gemv(const GemvParams& p, hipStream_t stream)
{
    if (p.M < 1 || p.M > 8 || p.N <= 0 || p.K <= 0)
    {
        set_error("gemv: unsupported M=%d N=%d K=%d (1 <= M <= 8)", p.M, p.N, p.K);
        return -1;
    }
    const bool sq = p.wtype == W_INT8_SQ;
    const bool swiglu = p.epi == EPI_SWIGLU || p.epi == EPI_SWIGLU_QSTATIC;
    if (!sq && (p.pro >= PRO_RMSNORM_QSTATIC))
    {
        set_error("gemv: quantising prologue needs W_INT8_SQ");
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
    if (p.wtype == W_FP16 && (p.K % 8))
    {
        set_error("gemv: fp16 weights need K %% 8 == 0 (K=%d)", p.K);
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
    const bool x_is_half = !(sq && p.pro == PRO_NONE);
    a.xh_bytes = x_is_half ? a.Kp * 2 : 0;
    const size_t smem_per_m = (size_t) a.xh_bytes + (sq ? (size_t) a.Kp : 0);
    const int mb = p.M <= 1 ? 1 : (p.M <= 2 ? 2 : (p.M <= 4 ? 4 : 8));
    if (128 + mb * smem_per_m > 160 * 1024)
    {
        set_error("gemv: K=%d x M=%d does not fit LDS", p.K, p.M);
        return -1;
    }

    // rows per wave: enough 1-KiB loads in flight per wave (R*U >= 8) without starving the grid
    int R = 2;
    if (gemv_tune_r)
        R = gemv_tune_r;
    else if (swiglu)
        R = 2;
    else if (a.nchunks <= 2)
        R = 4;
    if (swiglu && (R & 1))
        R = 2;
    const int outs_per_group = swiglu ? R / 2 : R;
    a.ngroups = (p.N + outs_per_group - 1) / outs_per_group;
    int blocks = (a.ngroups + 3) / 4;
    const int max_blocks = 256 * 8;
    if (blocks > max_blocks)
        blocks = max_blocks;

#define TLLM_GEMV_DISPATCH(WT)                                                                                         \
    if (R == 4)                                                                                                        \
        return launch_mb<WT, 4, 2>(a, blocks, smem_per_m, stream);                                                     \
    else if (R == 1)                                                                                                   \
        return launch_mb<WT, 1, 8>(a, blocks, smem_per_m, stream);                                                     \
    else                                                                                                               \
        return launch_mb<WT, 2, 4>(a, blocks, smem_per_m, stream);

    switch (p.wtype)
    {
    case W_FP16: TLLM_GEMV_DISPATCH(W_FP16)
    case W_INT8_WOQ: TLLM_GEMV_DISPATCH(W_INT8_WOQ)
    case W_INT4_WOQ: TLLM_GEMV_DISPATCH(W_INT4_WOQ)
    case W_INT8_SQ: TLLM_GEMV_DISPATCH(W_INT8_SQ)
    }
#undef TLLM_GEMV_DISPATCH
    return -1;
}

} // namespace kernels
} // namespace tllm
