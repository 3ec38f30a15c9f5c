// Context (prompt) phase of the attention plugin.
//
// Reference: enqueueContext, P/gptAttentionCommon/gptAttentionCommon.cpp:362-620 — five kernels and a 134 MB
// fp32 score scratch at S=1024: invokeAddFusedQKVBiasTranspose (RoPE), invokeTranspose4dBatchMajor (KV write,
// int8 quant), cuBLAS QK^T (fp32 out), invokeMaskedSoftmax, cuBLAS PV, invokeTransposeQKV.
// Here: (1) one pass applies RoPE to q,k in place in the packed QKV buffer (the reference rewrites it too,
// K/unfusedAttentionKernels.cu:1401-1403) and writes RoPE'd K and raw V into the cache [B,2,H,Smax,Dh]
// (int8: sat(rni(x * s)); padded rows are written as zero); (2) a flash-style pass — one wave per query row,
// online softmax, no score matrix — reads K/V straight from the packed buffer (L2-resident per head).
// Numerics as the decode kernel: fp32 dot / softmax / accumulation, probabilities rounded to fp16, one
// rounding to fp16 at the end; keys j > i or j >= input_len[b] are excluded (the reference adds -10000, whose
// exp underflows to the same 0).
#include "dev_utils.h"
#include "kernels.h"

namespace tllm
{
namespace kernels
{
using namespace dev;

namespace
{

__device__ __forceinline__ void h8_to_f(const uint4& v, float* f)
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

__device__ __forceinline__ uint4 f_to_h8(const float* f)
{
    return make_uint4(pack_h2(f[0], f[1]), pack_h2(f[2], f[3]), pack_h2(f[4], f[5]), pack_h2(f[6], f[7]));
}

__device__ __forceinline__ uint2 quant8(const uint4& v, float s)
{
    float f[8];
    h8_to_f(v, f);
    uint32_t o[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < 8; ++j)
        o[j >> 2] |= ((uint32_t) (uint8_t) f2i8_rni_sat(f[j] * s)) << (8 * (j & 3));
    return make_uint2(o[0], o[1]);
}

// grid (S, H, B); DH/8 active lanes, each owning 8 consecutive elements of the head.
template <int DH>
__global__ void rope_kv_write_kernel(const ContextAttnParams p)
{
    constexpr int LPR = DH / 8;
    const int s = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int li = threadIdx.x;
    if (li >= LPR)
        return;
    const int H = p.num_heads, S = p.seq;
    uint16_t* row = reinterpret_cast<uint16_t*>(p.qkv) + ((int64_t) b * S + s) * 3 * H * DH;
    uint16_t* qp = row + (int64_t) h * DH + li * 8;
    uint16_t* kp = row + (int64_t) (H + h) * DH + li * 8;
    uint16_t* vp = row + (int64_t) (2 * H + h) * DH + li * 8;
    const bool valid = s < p.input_lengths[b];
    uint4 q4 = *reinterpret_cast<const uint4*>(qp);
    uint4 k4 = *reinterpret_cast<const uint4*>(kp);
    uint4 v4 = *reinterpret_cast<const uint4*>(vp);
    if (!valid)
    {
        // padding rows are zeroed (K/unfusedAttentionKernels.cu:1411-1423)
        q4 = k4 = v4 = make_uint4(0, 0, 0, 0);
    }
    else if (p.rotary_dim > 0)
    {
        float qf[8], kf[8];
        h8_to_f(q4, qf);
        h8_to_f(k4, kf);
        const int half = p.rotary_dim >> 1;
        int pos = s < p.rope_table_len ? s : p.rope_table_len - 1;
        const float2* tab = reinterpret_cast<const float2*>(p.rope_table) + (int64_t) pos * half;
        if (p.neox)
        {
            float qo[8], ko[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
            {
                qo[j] = __shfl_xor(qf[j], LPR / 2, 64);
                ko[j] = __shfl_xor(kf[j], LPR / 2, 64);
            }
            const bool second = li >= LPR / 2;
#pragma unroll
            for (int j = 0; j < 8; ++j)
            {
                const int d = li * 8 + j;
                const float2 cs = tab[second ? d - half : d];
                const float sn = second ? cs.y : -cs.y;
                qf[j] = cs.x * qf[j] + sn * qo[j];
                kf[j] = cs.x * kf[j] + sn * ko[j];
            }
        }
        else
        {
#pragma unroll
            for (int j = 0; j < 8; j += 2)
            {
                const int d = li * 8 + j;
                if (d < p.rotary_dim)
                {
                    const float2 cs = tab[d >> 1];
                    const float q0 = qf[j], q1 = qf[j + 1], k0 = kf[j], k1 = kf[j + 1];
                    qf[j] = cs.x * q0 - cs.y * q1;
                    qf[j + 1] = cs.x * q1 + cs.y * q0;
                    kf[j] = cs.x * k0 - cs.y * k1;
                    kf[j + 1] = cs.x * k1 + cs.y * k0;
                }
            }
        }
        q4 = f_to_h8(qf);
        k4 = f_to_h8(kf);
    }
    *reinterpret_cast<uint4*>(qp) = q4;
    *reinterpret_cast<uint4*>(kp) = k4;
    if (!valid)
        *reinterpret_cast<uint4*>(vp) = v4;
    if (s < p.max_seq_len)
    {
        const int esz = p.int8_kv ? 1 : 2;
        char* kc = reinterpret_cast<char*>(p.kv_cache)
            + (((int64_t) (b * 2 + 0) * H + h) * p.max_seq_len + s) * DH * esz + li * 8 * esz;
        char* vc = reinterpret_cast<char*>(p.kv_cache)
            + (((int64_t) (b * 2 + 1) * H + h) * p.max_seq_len + s) * DH * esz + li * 8 * esz;
        if (p.int8_kv)
        {
            const float sc = p.kv_scale_orig_quant[0];
            *reinterpret_cast<uint2*>(kc) = quant8(k4, sc);
            *reinterpret_cast<uint2*>(vc) = quant8(v4, sc);
        }
        else
        {
            *reinterpret_cast<uint4*>(kc) = k4;
            *reinterpret_cast<uint4*>(vc) = v4;
        }
    }
}

// grid (ceil(S/4), H, B); 4 waves, one query row per wave.
template <int DH>
__global__ __launch_bounds__(256) void context_attn_kernel(const ContextAttnParams p)
{
    constexpr int LPR = DH / 8, RPW = 64 / LPR, NIT = 4;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int qi = blockIdx.x * 4 + wid, h = blockIdx.y, b = blockIdx.z;
    const int li = lane % LPR, grp = lane / LPR;
    const int H = p.num_heads, S = p.seq;
    if (qi >= S)
        return;
    const int len = p.input_lengths[b];
    uint16_t* outp = reinterpret_cast<uint16_t*>(p.out) + (((int64_t) b * S + qi) * H + h) * DH + li * 8;
    if (qi >= len)
    {
        if (grp == 0)
            *reinterpret_cast<uint4*>(outp) = make_uint4(0, 0, 0, 0);
        return;
    }
    const uint16_t* base = reinterpret_cast<const uint16_t*>(p.qkv) + (int64_t) b * S * 3 * H * DH;
    const int64_t rs = (int64_t) 3 * H * DH; // row stride
    const uint4 q16 = *reinterpret_cast<const uint4*>(base + (int64_t) qi * rs + (int64_t) h * DH + li * 8);
    const uint16_t* kb = base + (int64_t) (H + h) * DH + li * 8;
    const uint16_t* vb = base + (int64_t) (2 * H + h) * DH + li * 8;

    float m = -INFINITY, l = 0.f;
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int nkeys = qi + 1; // causal; qi < len so all of them are real tokens
    for (int j0 = 0; j0 < nkeys; j0 += RPW * NIT)
    {
        uint4 kk[NIT], vv[NIT];
        float s[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i)
        {
            const int j = j0 + i * RPW + grp;
            kk[i] = vv[i] = make_uint4(0, 0, 0, 0);
            if (j < nkeys)
            {
                kk[i] = *reinterpret_cast<const uint4*>(kb + (int64_t) j * rs);
                vv[i] = *reinterpret_cast<const uint4*>(vb + (int64_t) j * rs);
            }
        }
        float mt = -INFINITY;
#pragma unroll
        for (int i = 0; i < NIT; ++i)
        {
            const int j = j0 + i * RPW + grp;
            float d = 0.f;
            d = dot2(q16.x, kk[i].x, d);
            d = dot2(q16.y, kk[i].y, d);
            d = dot2(q16.z, kk[i].z, d);
            d = dot2(q16.w, kk[i].w, d);
            d = group_sum<LPR>(d) * p.inv_sqrt_dh;
            s[i] = j < nkeys ? d : -INFINITY;
            mt = fmaxf(mt, s[i]);
        }
#pragma unroll
        for (int mk = 32; mk >= LPR; mk >>= 1)
            mt = fmaxf(mt, __shfl_xor(mt, mk, 64));
        const float mn = fmaxf(m, mt);
        const float alpha = (m == -INFINITY) ? 0.f : __expf(m - mn);
        l *= alpha;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            o[j] *= alpha;
        m = mn;
#pragma unroll
        for (int i = 0; i < NIT; ++i)
        {
            const float pr = (s[i] == -INFINITY) ? 0.f : __expf(s[i] - m);
            l += pr;
            const float p16 = h2f(f2h(pr));
            float vf[8];
            h8_to_f(vv[i], vf);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                o[j] = fmaf(p16, vf[j], o[j]);
        }
    }
    // merge the lane groups (each saw a disjoint subset of keys, same running max)
#pragma unroll
    for (int mk = 32; mk >= LPR; mk >>= 1)
    {
        l += __shfl_xor(l, mk, 64);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            o[j] += __shfl_xor(o[j], mk, 64);
    }
    if (grp == 0)
    {
        const float inv = 1.f / (l + 1.e-6f);
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            r[j] = o[j] * inv;
        *reinterpret_cast<uint4*>(outp) = f_to_h8(r);
    }
}

template <int DH>
int launch_dh(const ContextAttnParams& p, hipStream_t stream)
{
    hipLaunchKernelGGL((rope_kv_write_kernel<DH>), dim3(p.seq, p.num_heads, p.batch), dim3(64), 0, stream, p);
    hipLaunchKernelGGL((context_attn_kernel<DH>), dim3((p.seq + 3) / 4, p.num_heads, p.batch), dim3(256), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
    {
        set_error("context attention launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}

} // namespace

int launch_context_attention(const ContextAttnParams& p, hipStream_t stream)
{
    if (p.rotary_dim > 0 && (!p.rope_table || (p.neox && p.rotary_dim != p.head_size)))
    {
        set_error("context attention: bad rotary configuration");
        return -1;
    }
    if (p.int8_kv && !p.kv_scale_orig_quant)
    {
        set_error("context attention: int8 KV cache needs kv_scale_orig_quant");
        return -1;
    }
    if (p.seq > p.max_seq_len)
    {
        set_error("context attention: seq %d exceeds cache capacity %d", p.seq, p.max_seq_len);
        return -1;
    }
    switch (p.head_size)
    {
    case 32: return launch_dh<32>(p, stream);
    case 64: return launch_dh<64>(p, stream);
    case 128: return launch_dh<128>(p, stream);
    case 256: return launch_dh<256>(p, stream);
    default: set_error("context attention: head_size %d not supported", p.head_size); return -1;
    }
}

} // namespace kernels
} // namespace tllm
