// Row-wise and element-wise kernels of the LLaMA layer: RMSNorm (+ int8 quant), the activation
// quantisers, SwiGLU, residual add, embedding gather, last-token gather, greedy argmax.
// All HBM/latency-bound: 16-byte accesses, one workgroup per row, wave64 shuffles for reductions.
#include "dev_utils.h"
#include "kernels.h"
#include "launch_util.h"
#include <atomic>
#include <cstdlib>

namespace tllm
{
namespace kernels
{
using namespace dev;

namespace
{

__device__ __forceinline__ void unpack8(const uint4& v, float* f)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
    {
        h2_t h = u32_as_h2(w[j]);
        f[2 * j] = (float) h.x;
        f[2 * j + 1] = (float) h.y;
    }
}

// ---------------------------------------------------------------------------------------------
// RMSNorm (+ residual add, + int8 quant).  A5: PY/functional.py:3195-3219 (fp32 statistics, fp16 io);
// quant tail: K/layernormKernels.cu:146-183 (normalise -> round to fp16 -> quantise; amax floor 1e-6 in T).
// One workgroup per row; the row lives in LDS between the passes.
// ---------------------------------------------------------------------------------------------
// VEC = 8: 16-byte accesses (N % 8 == 0, 16-byte aligned rows); VEC = 1: any N.
template <int VEC>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const RmsnormParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);
    uint16_t* row = reinterpret_cast<uint16_t*>(smem + 128);
    const int m = blockIdx.x, tid = threadIdx.x, N = p.N;
    const uint16_t* x = reinterpret_cast<const uint16_t*>(p.x) + (int64_t) m * N;
    const uint16_t* res = p.residual ? reinterpret_cast<const uint16_t*>(p.residual) + (int64_t) m * N : nullptr;
    uint16_t* so = p.sum_out ? reinterpret_cast<uint16_t*>(p.sum_out) + (int64_t) m * N : nullptr;
    const uint16_t* g = reinterpret_cast<const uint16_t*>(p.gamma);
    const uint16_t* be = reinterpret_cast<const uint16_t*>(p.beta);
    using vec_t = typename std::conditional<VEC == 8, uint4, uint16_t>::type;
    auto unpack = [](const vec_t& v, uint16_t* e) {
        if constexpr (VEC == 8)
        {
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                e[2 * j] = (uint16_t) (w[j] & 0xffffu);
                e[2 * j + 1] = (uint16_t) (w[j] >> 16);
            }
        }
        else
            e[0] = v;
    };
    auto pack = [](const uint16_t* e) {
        if constexpr (VEC == 8)
            return make_uint4(e[0] | ((uint32_t) e[1] << 16), e[2] | ((uint32_t) e[3] << 16), e[4] | ((uint32_t) e[5] << 16),
                e[6] | ((uint32_t) e[7] << 16));
        else
            return e[0];
    };

    float ss = 0.f, sum = 0.f;
    for (int k = tid * VEC; k < N; k += 256 * VEC)
    {
        uint16_t e[VEC], r[VEC];
        unpack(*reinterpret_cast<const vec_t*>(x + k), e);
        if (res)
        {
            unpack(*reinterpret_cast<const vec_t*>(res + k), r);
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                e[j] = f2h(h2f(e[j]) + h2f(r[j]));
            *reinterpret_cast<vec_t*>(so + k) = pack(e);
        }
        *reinterpret_cast<vec_t*>(row + k) = pack(e);
#pragma unroll
        for (int j = 0; j < VEC; ++j)
        {
            const float f = h2f(e[j]);
            ss += f * f;
            sum += f;
        }
    }
    float inv, mean = 0.f;
    if (p.layernorm)
    {
        mean = block_sum(sum, red) / (float) N;
        float var;
        if (p.use_diff_of_squares)
            var = block_sum(ss, red) / (float) N - mean * mean;
        else
        {
            float sd = 0.f;
            for (int k = tid; k < N; k += 256)
            {
                const float d = h2f(row[k]) - mean;
                sd += d * d;
            }
            var = block_sum(sd, red) / (float) N;
        }
        inv = rsqrtf(fmaxf(var, 0.f) + p.eps);
    }
    else
    {
        ss = block_sum(ss, red);
        inv = 1.0f / sqrtf(ss / (float) N + p.eps);
    }

    uint16_t* y = p.y ? reinterpret_cast<uint16_t*>(p.y) + (int64_t) m * N : nullptr;
    int8_t* q = p.q ? p.q + (int64_t) m * N : nullptr;
    float amax = 0.f;
    for (int k = tid * VEC; k < N; k += 256 * VEC)
    {
        uint16_t e[VEC], ge[VEC], bb[VEC];
        unpack(*reinterpret_cast<const vec_t*>(row + k), e);
        unpack(*reinterpret_cast<const vec_t*>(g + k), ge);
        if (p.layernorm)
            unpack(*reinterpret_cast<const vec_t*>(be + k), bb);
#pragma unroll
        for (int j = 0; j < VEC; ++j)
        {
            uint16_t yb;
            if (p.layernorm)
                yb = f2h((h2f(e[j]) - mean) * inv * h2f(ge[j]) + h2f(bb[j])); // one rounding (layernormKernels.cu:146-160)
            else
                yb = f2h(h2f(f2h(h2f(e[j]) * inv)) * h2f(ge[j]));
            e[j] = yb;
            amax = fmaxf(amax, fabsf(h2f(yb)));
        }
        *reinterpret_cast<vec_t*>(row + k) = pack(e);
        if (y)
            *reinterpret_cast<vec_t*>(y + k) = pack(e);
    }
    if (!q)
        return;
    float qs;
    if (p.dyn_scale_out)
    {
        amax = block_max(amax, red);
        amax = fmaxf(amax, h2f(f2h(1e-6f)));
        qs = 127.f / amax;
        if (tid == 0)
            p.dyn_scale_out[m] = amax / 127.f;
    }
    else
    {
        qs = p.static_scale[0];
    }
    if constexpr (VEC == 8)
    {
        for (int k = tid * 8; k < N; k += 256 * 8)
        {
            uint16_t e[8];
            unpack(*reinterpret_cast<const uint4*>(row + k), e);
            uint32_t o[2] = {0, 0};
#pragma unroll
            for (int j = 0; j < 8; ++j)
                o[j >> 2] |= ((uint32_t) (uint8_t) f2i8_rni_sat(h2f(e[j]) * qs)) << (8 * (j & 3));
            *reinterpret_cast<uint2*>(q + k) = make_uint2(o[0], o[1]);
        }
    }
    else
    {
        for (int k = tid; k < N; k += 256)
            q[k] = f2i8_rni_sat(h2f(row[k]) * qs);
    }
}

// The RMSNorm flavour of the kernel above for vector rows of at most 2048 NV elements (LLaMA-7B: NV = 2), the prefill's shape
// ([tokens, hidden], one launch in front of the QKV GEMM and one in front of the MLP GEMMs of every layer).  Same arithmetic in
// the same order - per-thread partial sums over ascending k, the same block reductions, the same rounding points - so the two
// kernels are bit-identical; what differs is the data path: the row stays in registers instead of LDS, and gamma (and the
// static scale) are requested together with the row, not after the reduction (there the gamma load was a second, dependent
// round trip on every row: 7.6 us per launch at [1024, 4096] for 12 MB of traffic).
template <int NV>
__global__ __launch_bounds__(256) void rmsnorm_reg_kernel(const RmsnormParams p)
{
    __shared__ float red[32];
    const int m = blockIdx.x, tid = threadIdx.x, N = p.N;
    const uint16_t* x = reinterpret_cast<const uint16_t*>(p.x) + (int64_t) m * N;
    const uint16_t* res = p.residual ? reinterpret_cast<const uint16_t*>(p.residual) + (int64_t) m * N : nullptr;
    uint16_t* so = p.sum_out ? reinterpret_cast<uint16_t*>(p.sum_out) + (int64_t) m * N : nullptr;
    const uint16_t* g = reinterpret_cast<const uint16_t*>(p.gamma);
    uint4 xv[NV], rv[NV], gv[NV];
    // every request of this thread before the first value is looked at; vectors beyond N read a clamped address and are dropped
#pragma unroll
    for (int j = 0; j < NV; ++j)
    {
        const int k = (tid + j * 256) * 8;
        const int kc = k < N ? k : N - 8;
        xv[j] = *reinterpret_cast<const uint4*>(x + kc);
        rv[j] = res ? *reinterpret_cast<const uint4*>(res + kc) : make_uint4(0, 0, 0, 0);
        gv[j] = *reinterpret_cast<const uint4*>(g + kc);
    }
    const float qs_static = (p.q && !p.dyn_scale_out) ? p.static_scale[0] : 0.f;
    auto unpack = [](const uint4& v, uint16_t* e) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
            e[2 * j] = (uint16_t) (w[j] & 0xffffu);
            e[2 * j + 1] = (uint16_t) (w[j] >> 16);
        }
    };
    auto pack = [](const uint16_t* e) {
        return make_uint4(e[0] | ((uint32_t) e[1] << 16), e[2] | ((uint32_t) e[3] << 16), e[4] | ((uint32_t) e[5] << 16),
            e[6] | ((uint32_t) e[7] << 16));
    };
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
    {
        const int k = (tid + j * 256) * 8;
        if (k < N)
        {
            uint16_t e[8], r[8];
            unpack(xv[j], e);
            if (res)
            {
                unpack(rv[j], r);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    e[i] = f2h(h2f(e[i]) + h2f(r[i]));
                xv[j] = pack(e);
                *reinterpret_cast<uint4*>(so + k) = xv[j];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
            {
                const float f = h2f(e[i]);
                ss += f * f;
            }
        }
    }
    ss = block_sum(ss, red);
    const float inv = 1.0f / sqrtf(ss / (float) N + p.eps);
    uint16_t* y = p.y ? reinterpret_cast<uint16_t*>(p.y) + (int64_t) m * N : nullptr;
    int8_t* q = p.q ? p.q + (int64_t) m * N : nullptr;
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
    {
        const int k = (tid + j * 256) * 8;
        if (k < N)
        {
            uint16_t e[8], ge[8];
            unpack(xv[j], e);
            unpack(gv[j], ge);
#pragma unroll
            for (int i = 0; i < 8; ++i)
            {
                e[i] = f2h(h2f(f2h(h2f(e[i]) * inv)) * h2f(ge[i]));
                amax = fmaxf(amax, fabsf(h2f(e[i])));
            }
            xv[j] = pack(e);
            if (y)
                *reinterpret_cast<uint4*>(y + k) = xv[j];
        }
    }
    if (!q)
        return;
    float qs;
    if (p.dyn_scale_out)
    {
        amax = block_max(amax, red);
        amax = fmaxf(amax, h2f(f2h(1e-6f)));
        qs = 127.f / amax;
        if (tid == 0)
            p.dyn_scale_out[m] = amax / 127.f;
    }
    else
        qs = qs_static;
#pragma unroll
    for (int j = 0; j < NV; ++j)
    {
        const int k = (tid + j * 256) * 8;
        if (k < N)
        {
            uint16_t e[8];
            unpack(xv[j], e);
            uint32_t o[2] = {0, 0};
#pragma unroll
            for (int i = 0; i < 8; ++i)
                o[i >> 2] |= ((uint32_t) (uint8_t) f2i8_rni_sat(h2f(e[i]) * qs)) << (8 * (i & 3));
            *reinterpret_cast<uint2*>(q + k) = make_uint2(o[0], o[1]);
        }
    }
}

// A11 static per-tensor quantiser.  K/quantization.cu:31-64: q = sat(rni(float(x) * scale)).
template <typename T>
__global__ __launch_bounds__(256) void quantize_tensor_kernel(int8_t* dst, const T* src, int64_t size, const float* scale)
{
    const float s = scale[0];
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < size; i += (int64_t) gridDim.x * blockDim.x)
    {
        float f;
        if constexpr (sizeof(T) == 2)
            f = h2f(src[i]);
        else
            f = src[i];
        dst[i] = f2i8_rni_sat(f * s);
    }
}

// A11 per-token quantiser.  K/quantization.cu:94-118: amax in T with floor T(1e-6), scale_out = amax / 127,
// q = sat(rni(float(x) * (127 / amax))).
template <typename T>
__global__ __launch_bounds__(256) void quantize_per_token_kernel(
    int8_t* dst, const T* src, int64_t cols, float* scale_out)
{
    __shared__ float red[32];
    const T* s = src + (int64_t) blockIdx.x * cols;
    int8_t* d = dst + (int64_t) blockIdx.x * cols;
    float amax;
    if constexpr (sizeof(T) == 2)
        amax = h2f(f2h(1e-6f));
    else
        amax = 1e-6f;
    for (int64_t i = threadIdx.x; i < cols; i += blockDim.x)
    {
        float f;
        if constexpr (sizeof(T) == 2)
            f = h2f(s[i]);
        else
            f = s[i];
        amax = fmaxf(amax, fabsf(f));
    }
    amax = block_max(amax, red);
    if (threadIdx.x == 0)
        scale_out[blockIdx.x] = amax / 127.f;
    const float qs = 127.f / amax;
    for (int64_t i = threadIdx.x; i < cols; i += blockDim.x)
    {
        float f;
        if constexpr (sizeof(T) == 2)
            f = h2f(s[i]);
        else
            f = s[i];
        d[i] = f2i8_rni_sat(f * qs);
    }
}

// 8 halfs (16 bytes) per thread per iteration when n % 8 == 0 and the pointers are 16-byte aligned (VEC = 8).
template <int VEC, typename F>
__device__ __forceinline__ void binary_h16(uint16_t* y, const uint16_t* a, const uint16_t* b, int64_t n, F f)
{
    const int64_t stride = (int64_t) gridDim.x * blockDim.x * VEC;
    for (int64_t i = ((int64_t) blockIdx.x * blockDim.x + threadIdx.x) * VEC; i < n; i += stride)
    {
        if constexpr (VEC == 8)
        {
            const uint4 va = *reinterpret_cast<const uint4*>(a + i), vb = *reinterpret_cast<const uint4*>(b + i);
            const uint32_t wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w};
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                o[j] = (uint32_t) f((uint16_t) (wa[j] & 0xffffu), (uint16_t) (wb[j] & 0xffffu))
                    | ((uint32_t) f((uint16_t) (wa[j] >> 16), (uint16_t) (wb[j] >> 16)) << 16);
            *reinterpret_cast<uint4*>(y + i) = make_uint4(o[0], o[1], o[2], o[3]);
        }
        else
            y[i] = f(a[i], b[i]);
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void swiglu_kernel(uint16_t* y, const uint16_t* a, const uint16_t* b, int64_t n)
{
    binary_h16<VEC>(y, a, b, n, [](uint16_t ga, uint16_t ub) {
        const float g = h2f(ga);
        const float s = h2f(f2h(g / (1.f + __expf(-g))));
        return f2h(s * h2f(ub));
    });
}

template <int VEC>
__global__ __launch_bounds__(256) void swiglu_quant_kernel(int8_t* q, const uint16_t* a, const uint16_t* b, int64_t n, const float* scale)
{
    const float qs = scale[0];
    const int64_t stride = (int64_t) gridDim.x * blockDim.x * VEC;
    for (int64_t i = ((int64_t) blockIdx.x * blockDim.x + threadIdx.x) * VEC; i < n; i += stride)
    {
        auto one = [&](uint16_t ga, uint16_t ub) {
            const float g = h2f(ga);
            const float s = h2f(f2h(g / (1.f + __expf(-g))));
            return (uint32_t) (uint8_t) f2i8_rni_sat(h2f(f2h(s * h2f(ub))) * qs);
        };
        if constexpr (VEC == 8)
        {
            const uint4 va = *reinterpret_cast<const uint4*>(a + i), vb = *reinterpret_cast<const uint4*>(b + i);
            const uint32_t wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w};
            uint32_t o[2] = {0, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                o[j >> 1] |= one((uint16_t) (wa[j] & 0xffffu), (uint16_t) (wb[j] & 0xffffu)) << (16 * (j & 1));
                o[j >> 1] |= one((uint16_t) (wa[j] >> 16), (uint16_t) (wb[j] >> 16)) << (16 * (j & 1) + 8);
            }
            *reinterpret_cast<uint2*>(q + i) = make_uint2(o[0], o[1]);
        }
        else
            q[i] = (int8_t) one(a[i], b[i]);
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void add_kernel(uint16_t* y, const uint16_t* a, const uint16_t* b, int64_t n)
{
    binary_h16<VEC>(y, a, b, n, [](uint16_t x, uint16_t z) { return f2h(h2f(x) + h2f(z)); });
}

__global__ __launch_bounds__(256) void embedding_kernel(
    uint16_t* out, const int32_t* ids, const uint16_t* table, int32_t hidden, int32_t vocab)
{
    const int64_t t = blockIdx.x;
    const int32_t id = ids[t];
    const bool ok = id >= 0 && id < vocab;
    const uint16_t* src = table + (int64_t) (ok ? id : 0) * hidden;
    uint16_t* dst = out + t * hidden;
    if ((hidden & 7) == 0)
    {
        for (int k = threadIdx.x * 8; k < hidden; k += blockDim.x * 8)
        {
            uint4 v = ok ? *reinterpret_cast<const uint4*>(src + k) : make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(dst + k) = v;
        }
    }
    else
    {
        for (int k = threadIdx.x; k < hidden; k += blockDim.x)
            dst[k] = ok ? src[k] : (uint16_t) 0;
    }
}

__global__ __launch_bounds__(256) void gather_last_token_kernel(
    uint16_t* out, const uint16_t* hidden, const int32_t* last_token_ids, int32_t seq, int32_t hs)
{
    const int b = blockIdx.x;
    int pos = last_token_ids[b] - 1;
    pos = pos < 0 ? 0 : (pos >= seq ? seq - 1 : pos);
    const uint16_t* src = hidden + ((int64_t) b * seq + pos) * hs;
    uint16_t* dst = out + (int64_t) b * hs;
    for (int k = threadIdx.x; k < hs; k += blockDim.x)
        dst[k] = src[k];
}

// greedy argmax; ties -> lowest index (torch.argmax / top-k=1 of the reference sampler)
__global__ __launch_bounds__(1024) void argmax_kernel(int32_t* out_ids, const float* logits, int32_t vocab)
{
    __shared__ float sv[16];
    __shared__ int si[16];
    const float* l = logits + (int64_t) blockIdx.x * vocab;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < vocab; i += blockDim.x)
    {
        const float v = l[i];
        if (v > best || (v == best && i < bi))
        {
            best = v;
            bi = i;
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
    {
        const float ov = __shfl_xor(best, m, 64);
        const int oi = __shfl_xor(bi, m, 64);
        if (ov > best || (ov == best && oi < bi))
        {
            best = ov;
            bi = oi;
        }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0)
    {
        sv[wid] = best;
        si[wid] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        const int nw = (blockDim.x + 63) >> 6;
        for (int w = 1; w < nw; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi))
            {
                best = sv[w];
                bi = si[w];
            }
        out_ids[blockIdx.x] = bi == 0x7fffffff ? 0 : bi;
    }
}

__global__ __launch_bounds__(1024) void greedy_step_kernel(const GreedyParams p)
{
    __shared__ float sv[16];
    __shared__ int si[16];
    const int b = blockIdx.x;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int part = 0; part < p.nparts; ++part)
    {
        const float* l = p.logits + ((int64_t) part * p.batch + b) * p.vocab_part;
        const int base_id = part * p.vocab_part;
        if ((p.vocab_part & 3) == 0 && (reinterpret_cast<uintptr_t>(l) & 15) == 0)
        {
            // 16-byte loads, all of a thread's requests in flight before the compares (latency-bound otherwise)
            constexpr int UNR = 8;
            const int nvec = p.vocab_part >> 2;
            for (int v0 = threadIdx.x; v0 < nvec; v0 += blockDim.x * UNR)
            {
                float4 vals[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u)
                {
                    const int vi = v0 + u * blockDim.x;
                    vals[u] = vi < nvec ? reinterpret_cast<const float4*>(l)[vi]
                                        : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u)
                {
                    const int id0 = base_id + (v0 + u * blockDim.x) * 4;
                    const float f[4] = {vals[u].x, vals[u].y, vals[u].z, vals[u].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                    {
                        const int id = id0 + e;
                        if (id < p.vocab && (f[e] > best || (f[e] == best && id < bi)))
                        {
                            best = f[e];
                            bi = id;
                        }
                    }
                }
            }
            continue;
        }
        for (int i = threadIdx.x; i < p.vocab_part; i += blockDim.x)
        {
            const int id = base_id + i;
            if (id >= p.vocab)
                break;
            const float v = l[i];
            if (v > best || (v == best && id < bi))
            {
                best = v;
                bi = id;
            }
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
    {
        const float ov = __shfl_xor(best, m, 64);
        const int oi = __shfl_xor(bi, m, 64);
        if (ov > best || (ov == best && oi < bi))
        {
            best = ov;
            bi = oi;
        }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0)
    {
        sv[wid] = best;
        si[wid] = bi;
    }
    if (p.rope_row_out && threadIdx.x >= 64 && threadIdx.x < 64 + p.rope_half)
    {
        // next step's position: (seq_len after this step's advance) - padding of this sequence
        const int j = threadIdx.x - 64;
        int pos = p.seq_len[b] + (p.advance ? 1 : 0) - (p.max_input_len - p.input_lengths[b]);
        pos = pos < 0 ? 0 : (pos >= p.rope_table_len ? p.rope_table_len - 1 : pos);
        reinterpret_cast<float2*>(p.rope_row_out)[(int64_t) b * p.rope_half + j]
            = reinterpret_cast<const float2*>(p.rope_table)[(int64_t) pos * p.rope_half + j];
        if (j == 0 && p.rope_pos_out)
            p.rope_pos_out[b] = pos;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        const int nw = (blockDim.x + 63) >> 6;
        for (int w = 1; w < nw; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi))
            {
                best = sv[w];
                bi = si[w];
            }
        int id = bi == 0x7fffffff ? 0 : bi;
        int sl = p.seq_len[b];
        if (p.advance)
        {
            sl += 1;
            p.seq_len[b] = sl;
        }
        if (p.finished)
        {
            if (p.finished[b])
                id = p.end_id;
            else if (p.end_id >= 0 && id == p.end_id)
                p.finished[b] = 1;
        }
        if (sl < p.out_stride)
            p.out_ids[(int64_t) b * p.out_stride + sl] = id;
        p.cur_ids[b] = id;
        si[0] = id;
        if (b == 0 && p.step_epoch)
            *p.step_epoch += 1;
    }
    if (p.emb_table) // uniform: the next step's input row, gathered here instead of by an embedding launch of its own
    {
        __syncthreads();
        const int id = si[0];
        const bool ok = id >= 0 && id < p.vocab;
        const uint16_t* src = reinterpret_cast<const uint16_t*>(p.emb_table) + (int64_t) (ok ? id : 0) * p.hidden;
        uint16_t* dst = reinterpret_cast<uint16_t*>(p.x_out) + (int64_t) b * p.hidden;
        for (int k = threadIdx.x * 8; k < p.hidden; k += blockDim.x * 8) // hidden % 8 == 0 (checked by the launcher)
        {
            const uint4 v = *reinterpret_cast<const uint4*>(src + k);
            *reinterpret_cast<uint4*>(dst + k) = ok ? v : make_uint4(0, 0, 0, 0);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Beam search step (see BeamParams in kernels.h).  One workgroup of 1024 threads per batch entry.
// Dynamic LDS: the cache-indirection rows being re-parented, [beam][used slots] int32.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void beam_step_kernel(const BeamParams p)
{
    constexpr int MAXW = 8;
    extern __shared__ int32_t ci_stage[];
    __shared__ float red[32];
    __shared__ int redi[32];
    __shared__ float s_lse[MAXW], s_cum[MAXW], s_score[MAXW];
    __shared__ int s_fin[MAXW], s_idx[MAXW];
    const int b = blockIdx.x, W = p.beam, V = p.vocab, tid = threadIdx.x, bb0 = b * W;
    const int nrows = p.logits_per_batch ? p.batch : p.batch * W;
    auto logit = [&](int k, int v) -> float {
        const int part = v / p.vocab_part, vi = v % p.vocab_part;
        return p.logits[((int64_t) part * nrows + (p.logits_per_batch ? b : bb0 + k)) * p.vocab_part + vi];
    };
    if (tid < W)
    {
        s_cum[tid] = p.cum_log_probs[bb0 + tid];
        s_fin[tid] = p.finished ? p.finished[bb0 + tid] : 0;
    }
    __syncthreads();
    // ---- 1. log-sum-exp of every live hypothesis
    for (int k = 0; k < W; ++k)
    {
        if (s_fin[k]) // uniform
            continue;
        float mx = -INFINITY;
        for (int v = tid; v < V; v += blockDim.x)
            mx = fmaxf(mx, logit(k, v));
        mx = block_max(mx, red);
        float sm = 0.f;
        for (int v = tid; v < V; v += blockDim.x)
            sm += __expf(logit(k, v) - mx);
        sm = block_sum(sm, red);
        if (tid == 0)
            s_lse[k] = mx + __logf(sm);
        __syncthreads();
    }
    // ---- 2. the W best (hypothesis, token) pairs, best first; ties -> lowest k * V + v
    for (int j = 0; j < W; ++j)
    {
        float best = -INFINITY;
        int bi = 0x7fffffff;
        auto consider = [&](float sc, int idx) {
            if (sc > best || (sc == best && idx < bi))
            {
                bool taken = false;
                for (int q = 0; q < j; ++q)
                    taken = taken || s_idx[q] == idx;
                if (!taken)
                {
                    best = sc;
                    bi = idx;
                }
            }
        };
        for (int k = 0; k < W; ++k)
        {
            if (s_fin[k])
            {
                // a finished hypothesis stays as it is: one candidate, end_id, at its score
                if (tid == 0 && p.end_id >= 0)
                    consider(s_cum[k], k * V + p.end_id);
                continue;
            }
            const float base = s_cum[k] - s_lse[k];
            for (int v = tid; v < V; v += blockDim.x)
                consider(logit(k, v) + base, k * V + v);
        }
        // block arg-max
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1)
        {
            const float ov = __shfl_xor(best, m, 64);
            const int oi = __shfl_xor(bi, m, 64);
            if (ov > best || (ov == best && oi < bi))
            {
                best = ov;
                bi = oi;
            }
        }
        __syncthreads();
        if ((tid & 63) == 0)
        {
            red[tid >> 6] = best;
            redi[tid >> 6] = bi;
        }
        __syncthreads();
        if (tid == 0)
        {
            for (int w = 1; w < (int) (blockDim.x >> 6); ++w)
                if (red[w] > best || (red[w] == best && redi[w] < bi))
                {
                    best = red[w];
                    bi = redi[w];
                }
            if (bi == 0x7fffffff) // every remaining candidate is -inf (fewer live candidates than beams): repeat the best
            {
                bi = j > 0 ? s_idx[0] : 0;
                best = j > 0 ? s_score[0] : -INFINITY;
            }
            s_idx[j] = bi;
            s_score[j] = best;
        }
        __syncthreads();
    }
    // ---- 3. re-parent the cache indirection: stage the parents' rows, then write them to the children
    const int sl_old = p.seq_len[bb0];
    const int sl_new = sl_old + (p.advance ? 1 : 0);
    const int used = p.advance ? sl_old : sl_new; // slots whose K/V exist before this step's token: [0, used)
    int32_t* ci = p.cache_indirection;
    if (ci)
    {
        for (int i = tid; i < W * used; i += blockDim.x)
        {
            const int j = i / used, sidx = i % used;
            ci_stage[i] = ci[(int64_t) (bb0 + s_idx[j] / V) * p.out_stride + sidx];
        }
        __syncthreads();
        for (int i = tid; i < W * used; i += blockDim.x)
        {
            const int j = i / used, sidx = i % used;
            ci[(int64_t) (bb0 + j) * p.out_stride + sidx] = ci_stage[i];
        }
        // the token consumed by this step put its K/V into the parent's rows at slot sl_old
        if (p.advance && tid < W && sl_old < p.out_stride)
            ci[(int64_t) (bb0 + tid) * p.out_stride + sl_old] = s_idx[tid] / V;
    }
    // ---- 4. bookkeeping of the new hypotheses
    if (tid < W)
    {
        const int j = tid, parent = s_idx[j] / V, tok = s_idx[j] % V;
        const int fin = s_fin[parent] || (p.end_id >= 0 && tok == p.end_id);
        p.cum_log_probs[bb0 + j] = s_score[j];
        if (p.finished)
            p.finished[bb0 + j] = fin;
        p.seq_len[bb0 + j] = sl_new;
        p.cur_ids[bb0 + j] = tok;
        if (sl_new < p.out_stride)
        {
            p.out_ids[(int64_t) (bb0 + j) * p.out_stride + sl_new] = tok;
            p.parent_ids[(int64_t) (bb0 + j) * p.out_stride + sl_new] = parent;
        }
    }
    if (p.rope_row_out)
    {
        // next step's RoPE row, the same position for every hypothesis of this batch entry
        int pos = sl_new - (p.max_input_len - p.input_lengths[bb0]);
        pos = pos < 0 ? 0 : (pos >= p.rope_table_len ? p.rope_table_len - 1 : pos);
        for (int i = tid; i < W * p.rope_half; i += blockDim.x)
            reinterpret_cast<float2*>(p.rope_row_out)[(int64_t) bb0 * p.rope_half + i]
                = reinterpret_cast<const float2*>(p.rope_table)[(int64_t) pos * p.rope_half + i % p.rope_half];
        if (p.rope_pos_out && tid < W)
            p.rope_pos_out[bb0 + tid] = pos;
    }
}

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(256) void fill_random_kernel(void* dst, int dtype, int64_t n, uint32_t seed, float scale)
{
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x)
    {
        const uint32_t r = hash32((uint32_t) i * 2654435761u + seed) ^ hash32((uint32_t) (i >> 32) + seed * 31u);
        if (dtype == DT_HALF)
            reinterpret_cast<uint16_t*>(dst)[i] = f2h(((float) (r >> 8) * (1.f / 8388608.f) - 1.f) * scale);
        else if (dtype == DT_INT8)
            reinterpret_cast<int8_t*>(dst)[i] = (int8_t) ((int) (r % 255u) - 127);
        else
            reinterpret_cast<float*>(dst)[i] = ((float) (r >> 8) * (1.f / 8388608.f) - 1.f) * scale;
    }
}

__global__ void fill_i32_kernel(int32_t* dst, int32_t v, int64_t n)
{
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x)
        dst[i] = v;
}

inline int check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
    {
        set_error("%s launch failed: %s", what, hipGetErrorString(e));
        return -1;
    }
    return 0;
}

inline int grid_for(int64_t n)
{
    int64_t b = (n + 255) / 256;
    return (int) (b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

} // namespace

int launch_rmsnorm(const RmsnormParams& p, hipStream_t stream)
{
    if (p.M <= 0 || p.N <= 0)
        return 0;
    if (p.residual && !p.sum_out)
    {
        set_error("rmsnorm: residual given without sum_out");
        return -1;
    }
    if (p.q && !p.static_scale && !p.dyn_scale_out)
    {
        set_error("rmsnorm: quantised output needs a static scale or a dynamic-scale output");
        return -1;
    }
    const size_t smem = 128 + (size_t) p.N * 2;
    if (smem > 64 * 1024)
    {
        set_error("rmsnorm: N=%d too large", p.N);
        return -1;
    }
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const bool vec = !(p.N & 7) && al16(p.x) && al16(p.gamma) && al16(p.residual) && al16(p.sum_out) && al16(p.y) && al16(p.beta)
        && (reinterpret_cast<uintptr_t>(p.q) & 7) == 0;
    if (p.layernorm && !p.beta)
    {
        set_error("layernorm: bias is required");
        return -1;
    }
    if (vec && !p.layernorm && p.N >= 8 && p.N <= 2048 * 4)
    {
        if (p.N <= 2048)
            hipLaunchKernelGGL(rmsnorm_reg_kernel<1>, dim3(p.M), dim3(256), 0, stream, p);
        else if (p.N <= 4096)
            hipLaunchKernelGGL(rmsnorm_reg_kernel<2>, dim3(p.M), dim3(256), 0, stream, p);
        else
            hipLaunchKernelGGL(rmsnorm_reg_kernel<4>, dim3(p.M), dim3(256), 0, stream, p);
    }
    else if (vec)
        hipLaunchKernelGGL(rmsnorm_kernel<8>, dim3(p.M), dim3(256), smem, stream, p);
    else
        hipLaunchKernelGGL(rmsnorm_kernel<1>, dim3(p.M), dim3(256), smem, stream, p);
    return check_launch("rmsnorm");
}

int launch_quantize_tensor(
    int8_t* dst, const void* src, int32_t src_dtype, int64_t size, const float* scale, hipStream_t stream)
{
    if (size <= 0)
        return 0;
    if (src_dtype == DT_HALF)
        hipLaunchKernelGGL(quantize_tensor_kernel<uint16_t>, dim3(grid_for(size)), dim3(256), 0, stream, dst,
            reinterpret_cast<const uint16_t*>(src), size, scale);
    else if (src_dtype == DT_FLOAT)
        hipLaunchKernelGGL(quantize_tensor_kernel<float>, dim3(grid_for(size)), dim3(256), 0, stream, dst,
            reinterpret_cast<const float*>(src), size, scale);
    else
    {
        set_error("quantize_tensor: unsupported dtype %d", src_dtype);
        return -1;
    }
    return check_launch("quantize_tensor");
}

int launch_quantize_per_token(int8_t* dst, const void* src, int32_t src_dtype, int64_t rows, int64_t cols,
    float* scale_out, hipStream_t stream)
{
    if (rows <= 0 || cols <= 0)
        return 0;
    if (src_dtype == DT_HALF)
        hipLaunchKernelGGL(quantize_per_token_kernel<uint16_t>, dim3((unsigned) rows), dim3(256), 0, stream, dst,
            reinterpret_cast<const uint16_t*>(src), cols, scale_out);
    else if (src_dtype == DT_FLOAT)
        hipLaunchKernelGGL(quantize_per_token_kernel<float>, dim3((unsigned) rows), dim3(256), 0, stream, dst,
            reinterpret_cast<const float*>(src), cols, scale_out);
    else
    {
        set_error("quantize_per_token: unsupported dtype %d", src_dtype);
        return -1;
    }
    return check_launch("quantize_per_token");
}

static bool vec8_ok(int64_t n, const void* y, const void* a, const void* b)
{
    return !(n & 7) && !((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15);
}

int launch_swiglu(void* y, const void* a, const void* b, int64_t n, hipStream_t stream)
{
    if (n <= 0)
        return 0;
    if (vec8_ok(n, y, a, b))
        hipLaunchKernelGGL(swiglu_kernel<8>, dim3(grid_for(n / 8)), dim3(256), 0, stream, reinterpret_cast<uint16_t*>(y),
            reinterpret_cast<const uint16_t*>(a), reinterpret_cast<const uint16_t*>(b), n);
    else
        hipLaunchKernelGGL(swiglu_kernel<1>, dim3(grid_for(n)), dim3(256), 0, stream, reinterpret_cast<uint16_t*>(y),
            reinterpret_cast<const uint16_t*>(a), reinterpret_cast<const uint16_t*>(b), n);
    return check_launch("swiglu");
}

int launch_swiglu_quant(int8_t* q, const void* a, const void* b, int64_t n, const float* scale, hipStream_t stream)
{
    if (n <= 0)
        return 0;
    if (!(n & 7) && !((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) && !(reinterpret_cast<uintptr_t>(q) & 7))
        hipLaunchKernelGGL(swiglu_quant_kernel<8>, dim3(grid_for(n / 8)), dim3(256), 0, stream, q,
            reinterpret_cast<const uint16_t*>(a), reinterpret_cast<const uint16_t*>(b), n, scale);
    else
        hipLaunchKernelGGL(swiglu_quant_kernel<1>, dim3(grid_for(n)), dim3(256), 0, stream, q,
            reinterpret_cast<const uint16_t*>(a), reinterpret_cast<const uint16_t*>(b), n, scale);
    return check_launch("swiglu_quant");
}

int launch_add(void* y, const void* a, const void* b, int64_t n, hipStream_t stream)
{
    if (n <= 0)
        return 0;
    if (vec8_ok(n, y, a, b))
        hipLaunchKernelGGL(add_kernel<8>, dim3(grid_for(n / 8)), dim3(256), 0, stream, reinterpret_cast<uint16_t*>(y),
            reinterpret_cast<const uint16_t*>(a), reinterpret_cast<const uint16_t*>(b), n);
    else
        hipLaunchKernelGGL(add_kernel<1>, dim3(grid_for(n)), dim3(256), 0, stream, reinterpret_cast<uint16_t*>(y),
            reinterpret_cast<const uint16_t*>(a), reinterpret_cast<const uint16_t*>(b), n);
    return check_launch("add");
}

int launch_embedding(void* out, const int32_t* ids, const void* table, int64_t tokens, int32_t hidden, int32_t vocab,
    hipStream_t stream)
{
    if (tokens <= 0)
        return 0;
    hipLaunchKernelGGL(embedding_kernel, dim3((unsigned) tokens), dim3(256), 0, stream,
        reinterpret_cast<uint16_t*>(out), ids, reinterpret_cast<const uint16_t*>(table), hidden, vocab);
    return check_launch("embedding");
}

int launch_gather_last_token(void* out, const void* hidden, const int32_t* last_token_ids, int32_t batch, int32_t seq,
    int32_t hidden_size, hipStream_t stream)
{
    if (batch <= 0)
        return 0;
    hipLaunchKernelGGL(gather_last_token_kernel, dim3(batch), dim3(256), 0, stream, reinterpret_cast<uint16_t*>(out),
        reinterpret_cast<const uint16_t*>(hidden), last_token_ids, seq, hidden_size);
    return check_launch("gather_last_token");
}

__global__ __launch_bounds__(256) void gather_rows_kernel(uint16_t* out, const uint16_t* hidden, const int32_t* rows, int32_t hs)
{
    const uint16_t* src = hidden + (int64_t) rows[blockIdx.x] * hs;
    uint16_t* dst = out + (int64_t) blockIdx.x * hs;
    for (int k = threadIdx.x; k < hs; k += blockDim.x)
        dst[k] = src[k];
}

int launch_gather_rows(void* out, const void* hidden, const int32_t* rows, int32_t batch, int32_t hidden_size, hipStream_t stream)
{
    if (batch <= 0)
        return 0;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(batch), dim3(256), 0, stream, reinterpret_cast<uint16_t*>(out),
        reinterpret_cast<const uint16_t*>(hidden), rows, hidden_size);
    return check_launch("gather_rows");
}

__global__ void exclusive_scan_i32_kernel(int32_t* cu, const int32_t* lens, int32_t n)
{
    if (threadIdx.x == 0)
    {
        int32_t acc = 0;
        cu[0] = 0;
        for (int i = 0; i < n; ++i)
        {
            acc += lens[i];
            cu[i + 1] = acc;
        }
    }
}

int launch_exclusive_scan_i32(int32_t* cu, const int32_t* lens, int32_t n, hipStream_t stream)
{
    hipLaunchKernelGGL(exclusive_scan_i32_kernel, dim3(1), dim3(64), 0, stream, cu, lens, n);
    return check_launch("exclusive_scan");
}

int launch_greedy_step(const GreedyParams& p, hipStream_t stream)
{
    if (p.batch <= 0)
        return 0;
    if (p.emb_table && (!p.x_out || p.hidden <= 0 || (p.hidden & 7)))
    {
        set_error("greedy step: fused embedding gather needs x_out and hidden %% 8 == 0 (got %d)", p.hidden);
        return -1;
    }
    hipLaunchKernelGGL(greedy_step_kernel, dim3(p.batch), dim3(1024), 0, stream, p);
    return check_launch("greedy_step");
}

// Teacher forcing (parity tests): replace the token the sampler just chose by ids[b] - the output slot it wrote, the
// step's input id and the embedding row the next step consumes.  One workgroup per sequence.
__global__ __launch_bounds__(256) void force_token_kernel(const int32_t* ids, int32_t* cur_ids, int32_t* out_ids, int32_t out_stride,
    const int32_t* seq_len, const void* emb, void* x, int32_t hidden, int32_t vocab)
{
    const int b = blockIdx.x, id = ids[b];
    if (threadIdx.x == 0)
    {
        const int sl = seq_len[b];
        if (sl < out_stride)
            out_ids[(int64_t) b * out_stride + sl] = id;
        cur_ids[b] = id;
    }
    if (emb && x)
    {
        const bool ok = id >= 0 && id < vocab;
        const uint16_t* src = reinterpret_cast<const uint16_t*>(emb) + (int64_t) (ok ? id : 0) * hidden;
        uint16_t* dst = reinterpret_cast<uint16_t*>(x) + (int64_t) b * hidden;
        for (int k = threadIdx.x; k < hidden; k += blockDim.x)
            dst[k] = ok ? src[k] : (uint16_t) 0;
    }
}

int launch_force_token(const int32_t* ids_dev, int32_t* cur_ids, int32_t* out_ids, int32_t out_stride, const int32_t* seq_len,
    const void* emb, void* x, int32_t batch, int32_t hidden, int32_t vocab, hipStream_t stream)
{
    if (batch <= 0)
        return 0;
    hipLaunchKernelGGL(force_token_kernel, dim3(batch), dim3(256), 0, stream, ids_dev, cur_ids, out_ids, out_stride, seq_len, emb, x,
        hidden, vocab);
    return check_launch("force_token");
}

int launch_beam_step(const BeamParams& p, hipStream_t stream)
{
    if (p.batch <= 0)
        return 0;
    if (p.beam < 1 || p.beam > 8 || !p.cum_log_probs || !p.parent_ids || !p.cache_indirection)
    {
        set_error("beam step: beam width %d out of [1, 8] or missing state buffers", p.beam);
        return -1;
    }
    const size_t smem = (size_t) p.beam * p.out_stride * sizeof(int32_t);
    if (smem > 96 * 1024)
    {
        set_error("beam step: beam %d x %d slots does not fit the staging buffer", p.beam, p.out_stride);
        return -1;
    }
    launch_util::ensure_dynamic_lds(reinterpret_cast<const void*>(beam_step_kernel), 96 * 1024);
    hipLaunchKernelGGL(beam_step_kernel, dim3(p.batch), dim3(1024), smem, stream, p);
    return check_launch("beam_step");
}

int launch_fill_random(void* dst, int32_t dtype, int64_t n, uint32_t seed, float scale, hipStream_t stream)
{
    if (n <= 0)
        return 0;
    hipLaunchKernelGGL(fill_random_kernel, dim3(grid_for(n)), dim3(256), 0, stream, dst, dtype, n, seed, scale);
    return check_launch("fill_random");
}

int launch_fill_i32(int32_t* dst, int32_t value, int64_t n, hipStream_t stream)
{
    if (n <= 0)
        return 0;
    hipLaunchKernelGGL(fill_i32_kernel, dim3(grid_for(n)), dim3(256), 0, stream, dst, value, n);
    return check_launch("fill_i32");
}

int launch_argmax(int32_t* out_ids, const float* logits, int32_t batch, int32_t vocab, hipStream_t stream)
{
    if (batch <= 0)
        return 0;
    hipLaunchKernelGGL(argmax_kernel, dim3(batch), dim3(1024), 0, stream, out_ids, logits, vocab);
    return check_launch("argmax");
}

} // namespace kernels
} // namespace tllm
