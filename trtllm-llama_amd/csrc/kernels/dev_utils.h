// Device-side helpers shared by the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tllm
{
namespace dev
{

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));

constexpr int kWave = 64;

// int8 = cvt.rni.sat.s8.f32: round-half-even, saturate to [-128, 127], NaN -> 0
// (reference: T/cpp/tensorrt_llm/common/cudaTypeUtils.cuh:361-371).
__device__ __forceinline__ int8_t f2i8_rni_sat(float x)
{
    float r = __builtin_rintf(x);              // v_rndne_f32, ties to even
    r = __builtin_fminf(__builtin_fmaxf(r, -128.f), 127.f); // NaN -> -128 by fmax semantics; fix below
    int v = (int) r;
    return (int8_t) ((x != x) ? 0 : v);
}

// int32 output of the SmoothQuant GEMM: CUTLASS NumericConverter<int32_t, float> = cvt.rni.s32.f32 (round-half-even, saturating,
// NaN -> 0) applied to float(acc) * (scale_col * scale_row)   (epilogue_per_row_per_col_scale.h:296; the reference's known-answer
// formula T/tests/quantization/_utils.py:112-114 rounds the SCALED product for dtype int32 - the raw accumulator never leaves)
__device__ __forceinline__ int32_t f2i32_rni_sat(float x)
{
    const float r = __builtin_rintf(x);
    if (x != x)
        return 0;
    if (r >= 2147483648.f)
        return 2147483647;
    if (r <= -2147483648.f)
        return (int32_t) 0x80000000;
    return (int32_t) r;
}

__device__ __forceinline__ float h2f(uint16_t bits)
{
    _Float16 h;
    __builtin_memcpy(&h, &bits, 2);
    return (float) h;
}

__device__ __forceinline__ uint16_t f2h(float f)
{
    // The fp32 value is the contract's rounding point: without the (empty) asm the backend may fold a preceding
    // fmul into v_fma_mixlo_f16, i.e. round the exact product straight to fp16 - one rounding instead of the
    // reference's two (fp32 then fp16), 1 ulp off on rare ties (seen in the SmoothQuant GEMM epilogue).
    asm("" : "+v"(f));
    _Float16 h = (_Float16) f; // v_cvt_f16_f32, round-to-nearest-even
    uint16_t b;
    __builtin_memcpy(&b, &h, 2);
    return b;
}

// SwiGLU with the gate operand already in memory, fused into the epilogue of the OTHER projection: u = this GEMM's finished fp16
// values, g = the first projection's (8 halfs each).  Same rounding points as the separate pass (pointwise.hip swiglu_kernel;
// PY/layers/mlp.py:68-73): fp16(silu(g)) then fp16(that * u).
__device__ __forceinline__ uint16_t silu_mul_h(uint16_t g, uint16_t u)
{
    const float gf = h2f(g);
    const float s = h2f(f2h(gf / (1.f + __expf(-gf))));
    return f2h(s * h2f(u));
}
__device__ __forceinline__ uint4 epi_silu_gate8(const uint4& u, const uint4& g)
{
    const uint32_t u4[4] = {u.x, u.y, u.z, u.w}, g4[4] = {g.x, g.y, g.z, g.w};
    uint32_t o4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
        o4[e] = (uint32_t) silu_mul_h((uint16_t) (g4[e] & 0xffffu), (uint16_t) (u4[e] & 0xffffu))
            | ((uint32_t) silu_mul_h((uint16_t) (g4[e] >> 16), (uint16_t) (u4[e] >> 16)) << 16);
    return make_uint4(o4[0], o4[1], o4[2], o4[3]);
}

__device__ __forceinline__ h2_t u32_as_h2(uint32_t u)
{
    h2_t r;
    __builtin_memcpy(&r, &u, 4);
    return r;
}

__device__ __forceinline__ uint32_t h2_as_u32(h2_t h)
{
    uint32_t r;
    __builtin_memcpy(&r, &h, 4);
    return r;
}

__device__ __forceinline__ uint32_t pack_h2(float lo, float hi)
{
    h2_t r;
    r.x = (_Float16) lo;
    r.y = (_Float16) hi;
    return h2_as_u32(r);
}

// fp32 += a.lo*b.lo + a.hi*b.hi with fp16 operands (v_dot2_f32_f16: exact products, fp32 accumulate).
__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c)
{
    return __builtin_amdgcn_fdot2(u32_as_h2(a), u32_as_h2(b), c, false);
}

// int32 += sum of 4 signed int8 products (v_dot4_i32_i8).
__device__ __forceinline__ int sdot4(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot4((int) a, (int) b, c, false);
}

// ---- wave64 reductions on the DPP network (no LDS crossbar round trips).
// quad_perm xor1 / xor2 -> row_half_mirror -> row_mirror -> row_bcast:15 (rows 1,3) -> row_bcast:31 (rows 2,3);
// the total lands in lane 63 and is broadcast through an SGPR (v_readlane).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int src, int old)
{
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}

__device__ __forceinline__ float wave_sum(float v)
{
#define TLLM_DPP_F(ctrl, mask) __builtin_bit_cast(float, dpp_i32<ctrl, mask>(__builtin_bit_cast(int, v), 0))
    v += TLLM_DPP_F(0xB1, 0xf);  // quad_perm [1,0,3,2]
    v += TLLM_DPP_F(0x4E, 0xf);  // quad_perm [2,3,0,1]
    v += TLLM_DPP_F(0x141, 0xf); // row_half_mirror
    v += TLLM_DPP_F(0x140, 0xf); // row_mirror
    v += TLLM_DPP_F(0x142, 0xa); // row_bcast:15 -> rows 1, 3
    v += TLLM_DPP_F(0x143, 0xc); // row_bcast:31 -> rows 2, 3
#undef TLLM_DPP_F
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ int wave_sum(int v)
{
    v += dpp_i32<0xB1, 0xf>(v, 0);
    v += dpp_i32<0x4E, 0xf>(v, 0);
    v += dpp_i32<0x141, 0xf>(v, 0);
    v += dpp_i32<0x140, 0xf>(v, 0);
    v += dpp_i32<0x142, 0xa>(v, 0);
    v += dpp_i32<0x143, 0xc>(v, 0);
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ float wave_max(float v)
{
    // `old` = the lane's own value: rows that a row_bcast step does not target keep max(v, v) = v
#define TLLM_DPP_M(ctrl, mask)                                                                                         \
    v = fmaxf(v, __builtin_bit_cast(float, dpp_i32<ctrl, mask>(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v))))
    TLLM_DPP_M(0xB1, 0xf);
    TLLM_DPP_M(0x4E, 0xf);
    TLLM_DPP_M(0x141, 0xf);
    TLLM_DPP_M(0x140, 0xf);
    TLLM_DPP_M(0x142, 0xa);
    TLLM_DPP_M(0x143, 0xc);
#undef TLLM_DPP_M
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Sum over groups of G consecutive lanes (G power of two <= 64); every lane of the group gets the total.
template <int G>
__device__ __forceinline__ float group_sum(float v)
{
    // DPP inside a 16-lane row (xor 1, xor 2, half-mirror, mirror); only G > 16 needs the LDS crossbar
#define TLLM_DPP_G(ctrl) __builtin_bit_cast(float, dpp_i32<ctrl, 0xf>(__builtin_bit_cast(int, v), 0))
    if constexpr (G >= 2)
        v += TLLM_DPP_G(0xB1);
    if constexpr (G >= 4)
        v += TLLM_DPP_G(0x4E);
    if constexpr (G >= 8)
        v += TLLM_DPP_G(0x141);
    if constexpr (G >= 16)
        v += TLLM_DPP_G(0x140);
#undef TLLM_DPP_G
    if constexpr (G >= 32)
        v += __shfl_xor(v, 16, 64);
    if constexpr (G >= 64)
        v += __shfl_xor(v, 32, 64);
    return v;
}

// Block-wide reductions through LDS.  `red` must hold >= 32 floats.  All threads get the result.
__device__ __forceinline__ float block_sum(float v, float* red)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0)
        red[wid] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i)
        t += red[i];
    return t;
}

__device__ __forceinline__ float block_max(float v, float* red)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0)
        red[wid] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i)
        t = fmaxf(t, red[i]);
    return t;
}

// 16-byte streaming load that bypasses temporal caching (weights / KV read once per token).
__device__ __forceinline__ uint4 ld_nt16(const void* p)
{
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ uint2 ld_nt8(const void* p)
{
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    u2 v = __builtin_nontemporal_load(reinterpret_cast<const u2*>(p));
    return make_uint2(v.x, v.y);
}

// The same on the RAW splice (1024 + (q + 128)): the general kernel takes 1152 * sum_k x[k] - one number per activation row -
// off the finished sum instead of 1152 off every weight (2 of 6 VALU instructions per 4 weights)
__device__ __forceinline__ float dot_u8x4_raw(uint32_t w, uint32_t x01, uint32_t x23, float acc)
{
    const uint32_t magic = 0x64646464u;
    acc = __builtin_amdgcn_fdot2(u32_as_h2(__builtin_amdgcn_perm(magic, w, 0x04010400u)), u32_as_h2(x01), acc, false);
    acc = __builtin_amdgcn_fdot2(u32_as_h2(__builtin_amdgcn_perm(magic, w, 0x04030402u)), u32_as_h2(x23), acc, false);
    return acc;
}

__device__ __forceinline__ float dot_woq8_raw(const uint4& w, const uint4& xa, const uint4& xb, float acc)
{
    acc = dot_u8x4_raw(w.x, xa.x, xa.y, acc);
    acc = dot_u8x4_raw(w.y, xa.z, xa.w, acc);
    acc = dot_u8x4_raw(w.z, xb.x, xb.y, acc);
    acc = dot_u8x4_raw(w.w, xb.z, xb.w, acc);
    return acc;
}

// Half raw: the nibbles spliced as 1024 + 16 n (elements 2 3 6 7 of the word) go into the dot product as they are and
// 72 * sum(x_b) - one number per activation row - comes off the finished sum:  sum_k (n_k - 8) x_k = a + b / 16 - 72 sum(x_b).
// The nibbles spliced as 1024 + n (elements 0 1 4 5) are brought to n - 8 exactly before the dot product: taken raw as well
// (r02: 1032 * sum(x_a) off the sum) their bias is 220 x the signal, and with a DC offset in the activations (x = 3 + N(0, 1),
// K = 11008) the fp32 rounding of that sum no longer cancels: 1.5 fp16 ulp of error at the output (ADVICE r2,
// tests/test_gpu_plugins.py::test_weight_only_gemv_with_a_dc_offset_in_the_activations).  The 1024 + 16 n splice has the
// bias : signal ratio of the int8 path (16 x) and stays raw.  2 of 12 VALU per 8 weights saved instead of 4.
__device__ __forceinline__ void dot_u4x8_raw(uint32_t w, const uint4& x, float& a, float& b)
{
    const uint32_t m = 0x64006400u;
    const uint32_t w8 = w >> 8;
    const h2_t b0 = {(_Float16) 1032.f, (_Float16) 1032.f}; // 1024 + 8
    a = __builtin_amdgcn_fdot2(u32_as_h2((w & 0x000f000fu) | m) - b0, u32_as_h2(x.x), a, false);
    b = __builtin_amdgcn_fdot2(u32_as_h2((w & 0x00f000f0u) | m), u32_as_h2(x.y), b, false);
    a = __builtin_amdgcn_fdot2(u32_as_h2((w8 & 0x000f000fu) | m) - b0, u32_as_h2(x.z), a, false);
    b = __builtin_amdgcn_fdot2(u32_as_h2((w8 & 0x00f000f0u) | m), u32_as_h2(x.w), b, false);
}


} // namespace dev
} // namespace tllm
