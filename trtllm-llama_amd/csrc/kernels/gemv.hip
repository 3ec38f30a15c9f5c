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
#include <algorithm>
#include <map>

namespace tllm
{
namespace kernels
{
using namespace dev;

int gemv_tune_r = 0;             // test/bench override: rows per wave (0 = heuristic)
int gemv_tune_blocks_per_cu = 0; // test/bench override: persistent workgroups per CU (0 = occupancy query)

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
// LDS map: [0,256) reduction scratch | xh: MB rows of Kp fp16 (absent for raw-s8 input) | xq: MB rows of Kp s8 (SQ)
//
// Latency structure (the per-launch floor matters: a 7B layer is 4 launches of 17-90 MB, i.e. 3-15 us each at HBM
// speed): 1. the x (and gamma) vectors are requested first, 2. the first weight tile of every wave is requested
// right behind them (it does not depend on x) and streams in while 3. the workgroup builds pro(x) in registers
// (sum of squares -> one barrier -> normalise / quantise) and publishes it to LDS (one barrier); 4. dot products,
// 5. cross-lane reduction + epilogue.  Further tiles (large N) are loaded in the loop; co-resident workgroups
// (up to 8 per CU) overlap each other's phases.
constexpr int kRedBytes = 256;
constexpr int kNXV = 6; // x vectors (8 halfs) a thread keeps in registers: K <= 256 * 8 * 6 = 12288

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
    uint16_t* xh = reinterpret_cast<uint16_t*>(smem + kRedBytes);
    int8_t* xq = reinterpret_cast<int8_t*>(smem + kRedBytes + (size_t) a.xh_bytes * MB);
    const bool x_is_half = !(SQ && p.pro == PRO_NONE);
    const bool do_norm = p.pro == PRO_RMSNORM || p.pro == PRO_RMSNORM_QSTATIC || p.pro == PRO_RMSNORM_QDYN;
    const bool q_static = p.pro == PRO_RMSNORM_QSTATIC || p.pro == PRO_QSTATIC || p.pro == PRO_ATTN_QSTATIC;
    const bool q_dyn = p.pro == PRO_RMSNORM_QDYN || p.pro == PRO_QDYN || p.pro == PRO_ATTN_QDYN;
    const bool attn = p.pro >= PRO_ATTN;
    float row_scale[MB]; // per-token dequant scale when the prologue quantises dynamically
#pragma unroll
    for (int m = 0; m < MB; ++m)
        row_scale[m] = 1.f;

    // ------------------------------------------------------------------ weight-tile helpers
    const bool swiglu = p.epi == EPI_SWIGLU || p.epi == EPI_SWIGLU_QSTATIC;
    const char* wbase = reinterpret_cast<const char*>(p.w);
    const char* wup = p.w_up ? reinterpret_cast<const char*>(p.w_up) : wbase + (int64_t) p.N * p.ldw;
    const int lane_kbyte = lane * 16; // byte offset of this lane's vector inside a chunk row

    auto rows_of_group = [&](int g, const char* (&rowptr)[R]) {
#pragma unroll
        for (int r = 0; r < R; ++r)
        {
            if (swiglu)
            {
                const int o = g * (R / 2) + (r % (R / 2 > 0 ? R / 2 : 1));
                rowptr[r] = ((r < R / 2) ? wbase : wup) + (o < p.N ? (int64_t) o * p.ldw : 0);
            }
            else
            {
                const int row = g * R + r;
                rowptr[r] = wbase + (row < p.N ? (int64_t) row * p.ldw : 0);
            }
        }
    };
    // No branches around loads anywhere in this kernel: a lane-dependent `if` makes the compiler fence every load
    // with s_waitcnt + exec masking, which serialises the memory round trips.  Out-of-range lanes load a clamped,
    // valid address and the value is replaced by a select.
    const int64_t last_vec = p.ldw - 16; // byte offset of the last 16-byte vector of a weight row
    auto load_tile = [&](const char* const (&rowptr)[R], int c, uint4 (&wv)[U][R]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            const bool ok = ((c + u) * 64 + lane) * VEC < Kp;
            int64_t off = (int64_t) (c + u) * 1024 + lane_kbyte;
            off = off < last_vec ? off : last_vec;
#pragma unroll
            for (int r = 0; r < R; ++r)
            {
                // neutral element of the weight encoding: 0 (fp16 / s8), q + 128 = 0x80, nibble q + 8 = 0x8
                constexpr uint32_t kZeroW = WT == W_INT8_WOQ ? 0x80808080u : (WT == W_INT4_WOQ ? 0x88888888u : 0u);
                const uint4 v = ld_nt16(rowptr[r] + off);
                wv[u][r] = make_uint4(ok ? v.x : kZeroW, ok ? v.y : kZeroW, ok ? v.z : kZeroW, ok ? v.w : kZeroW);
            }
        }
    };

    // ------------------------------------------------------------------ prologue
    const uint16_t* gam = reinterpret_cast<const uint16_t*>(p.gamma);
    const bool vec_half = x_is_half && ((K & 7) == 0)
        && (attn || (((p.ldx & 7) == 0) && ((reinterpret_cast<uintptr_t>(p.x) & 15) == 0)))
        && (!do_norm || (reinterpret_cast<uintptr_t>(p.gamma) & 15) == 0);
    const bool reg_path = vec_half && Kp <= 256 * 8 * kNXV;

    // 1. request x (row 0) and gamma
    uint4 xv[kNXV], gv[kNXV];
    // x row m as 8-half vectors: from memory, or (PRO_ATTN*) merged from the split-KV attention partials:
    //   ctx[h, d] = sum_i e_i o_i[d] / (sum_i e_i l_i + 1e-6),  e_i = exp(m_i - max_i m_i)   (MM/...Template.h:1756)
    auto load_x_row = [&](int m) {
        if (!attn)
        {
            const uint16_t* xg = reinterpret_cast<const uint16_t*>(p.x) + (int64_t) m * p.ldx;
#pragma unroll
            for (int j = 0; j < kNXV; ++j)
            {
                const int k = (tid + j * 256) * 8;
                if (j * 2048 < Kp) // uniform: vector index range used by this K
                {
                    const uint4 v = *reinterpret_cast<const uint4*>(xg + (k < K ? k : K - 8));
                    xv[j] = make_uint4(k < K ? v.x : 0u, k < K ? v.y : 0u, k < K ? v.z : 0u, k < K ? v.w : 0u);
                }
                else
                    xv[j] = make_uint4(0, 0, 0, 0);
            }
            return;
        }
        const float2* ml = reinterpret_cast<const float2*>(p.attn_ml);
        int ns = p.attn_seq_len[m] / p.attn_tchunk + 1;
        ns = ns > p.attn_nsmax ? p.attn_nsmax : ns;
#pragma unroll
        for (int j = 0; j < kNXV; ++j)
        {
            const int k = (tid + j * 256) * 8;
            xv[j] = make_uint4(0, 0, 0, 0);
            if (k < K)
            {
                const int hh = k / p.attn_dh, d0 = k % p.attn_dh;
                const int64_t base = ((int64_t) m * p.attn_heads + hh) * p.attn_nsmax;
                float Mx = -INFINITY;
                for (int i = 0; i < ns; ++i)
                    Mx = fmaxf(Mx, ml[base + i].x);
                float L = 0.f;
                float o8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int i = 0; i < ns; ++i)
                {
                    const float2 v = ml[base + i];
                    const float e = (v.x == -INFINITY) ? 0.f : __expf(v.x - Mx);
                    const float4 a0 = *reinterpret_cast<const float4*>(p.attn_o + (base + i) * p.attn_dh + d0);
                    const float4 a1 = *reinterpret_cast<const float4*>(p.attn_o + (base + i) * p.attn_dh + d0 + 4);
                    L += v.y * e;
                    o8[0] += a0.x * e;
                    o8[1] += a0.y * e;
                    o8[2] += a0.z * e;
                    o8[3] += a0.w * e;
                    o8[4] += a1.x * e;
                    o8[5] += a1.y * e;
                    o8[6] += a1.z * e;
                    o8[7] += a1.w * e;
                }
                const float inv = 1.f / (L + 1.e-6f);
                xv[j] = make_uint4(pack_h2(o8[0] * inv, o8[1] * inv), pack_h2(o8[2] * inv, o8[3] * inv),
                    pack_h2(o8[4] * inv, o8[5] * inv), pack_h2(o8[6] * inv, o8[7] * inv));
            }
        }
    };
    if (reg_path && !attn)
    {
        load_x_row(0);
#pragma unroll
        for (int j = 0; j < kNXV; ++j)
        {
            const int k = (tid + j * 256) * 8;
            gv[j] = make_uint4(0, 0, 0, 0);
            if (do_norm && j * 2048 < Kp) // uniform
                gv[j] = *reinterpret_cast<const uint4*>(gam + (k < K ? k : K - 8));
        }
    }
    // 2. request the first weight tile of this wave
    const int g0 = blockIdx.x * 4 + wid;
    const char* rowptr[R];
    uint4 wv[U][R];
    rows_of_group(g0 < a.ngroups ? g0 : 0, rowptr);
    load_tile(rowptr, 0, wv);
    // 2b. ... and everything else that does not depend on x: the epilogue operands (scales, residual) of this
    //     wave's first row group and the launch-constant scales.  (Requested after the tile so that waiting for x
    //     does not wait for them; consumed after the dot products.)
    const int gstride = gridDim.x * 4;
    const int nouts = swiglu ? R / 2 : R;
    const int my_o = lane / MB, my_m = lane % MB; // the (output, row) this lane finishes
    const bool my_active = my_o < nouts && my_m < p.M;
    struct EpiOps
    {
        float s0, s1, res;
    };
    auto load_ops = [&](int g) {
        EpiOps e = {1.f, 1.f, 0.f};
        int n = swiglu ? g * (R / 2) + my_o : g * R + my_o;
        n = n < p.N ? n : p.N - 1; // clamped: inactive lanes load a valid element and ignore it
        if constexpr (WT == W_INT8_WOQ || WT == W_INT4_WOQ)
        {
            const uint16_t* sc = reinterpret_cast<const uint16_t*>(p.scale_col);
            e.s0 = h2f(sc[n]);
            if (swiglu) // uniform
                e.s1 = h2f(p.scale_col_up ? reinterpret_cast<const uint16_t*>(p.scale_col_up)[n] : sc[p.N + n]);
        }
        else if constexpr (SQ)
        {
            const float* sc = reinterpret_cast<const float*>(p.scale_col);
            e.s0 = sc[p.per_channel ? n : 0];
            if (swiglu) // uniform
            {
                const float* su = reinterpret_cast<const float*>(p.scale_col_up);
                e.s1 = su ? su[p.per_channel ? n : 0] : sc[p.per_channel ? p.N + n : 0];
            }
        }
        if (p.epi == EPI_RESIDUAL) // uniform
            e.res = h2f(reinterpret_cast<const uint16_t*>(p.residual)[(int64_t) (my_m < p.M ? my_m : 0) * p.ldy + n]);
        return e;
    };
    EpiOps ops_cur = load_ops(g0);
    float static_row_scale = 1.f, static_row_scale_up = 1.f, epi_q = 1.f, pro_q = 1.f;
    if constexpr (SQ)
    {
        if (!q_dyn && p.scale_row) // uniform
            static_row_scale = p.scale_row[(p.per_token && my_m < p.M) ? my_m : 0];
        static_row_scale_up = (!q_dyn && p.scale_row_up) ? p.scale_row_up[0] : static_row_scale;
        if (q_static)
            pro_q = p.act_scale[0];
    }
    if (p.epi == EPI_SWIGLU_QSTATIC)
        epi_q = p.epi_scale[0];

    // 3. build pro(x) in LDS
    if (reg_path)
    {
#pragma unroll
        for (int m = 0; m < MB; ++m)
        {
            if (m >= p.M)
                continue;
            if (m > 0 || attn)
                load_x_row(m); // (attention partials: requested after the weight tile, which streams meanwhile)
            float inv = 1.f;
            if (do_norm)
            {
                float ss = 0.f;
#pragma unroll
                for (int j = 0; j < kNXV; ++j)
                {
                    const uint32_t ws[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                    {
                        const h2_t h = u32_as_h2(ws[q]);
                        const float f0 = (float) h.x, f1 = (float) h.y;
                        ss += f0 * f0 + f1 * f1;
                    }
                }
                ss = wave_sum(ss);
                if (lane == 0)
                    red[m * 4 + wid] = ss;
                __syncthreads();
                ss = red[m * 4] + red[m * 4 + 1] + red[m * 4 + 2] + red[m * 4 + 3];
                inv = 1.0f / sqrtf(ss / (float) K + p.eps);
            }
            float amax = 0.f;
            if (do_norm || q_dyn)
            {
#pragma unroll
                for (int j = 0; j < kNXV; ++j)
                {
                    uint32_t xs4[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w};
                    const uint32_t gs4[4] = {gv[j].x, gv[j].y, gv[j].z, gv[j].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                    {
                        h2_t h = u32_as_h2(xs4[q]);
                        if (do_norm)
                        {
                            const h2_t gg = u32_as_h2(gs4[q]);
                            const float n0 = h2f(f2h((float) h.x * inv)), n1 = h2f(f2h((float) h.y * inv));
                            h.x = (_Float16) (n0 * (float) gg.x);
                            h.y = (_Float16) (n1 * (float) gg.y);
                            xs4[q] = h2_as_u32(h);
                        }
                        amax = fmaxf(amax, fmaxf(fabsf((float) h.x), fabsf((float) h.y)));
                    }
                    xv[j] = make_uint4(xs4[0], xs4[1], xs4[2], xs4[3]);
                }
            }
            float qs = 1.f;
            if (SQ && q_dyn)
            {
                amax = wave_max(amax);
                if (lane == 0)
                    red[32 + m * 4 + wid] = amax;
                __syncthreads();
                amax = fmaxf(fmaxf(red[32 + m * 4], red[32 + m * 4 + 1]), fmaxf(red[32 + m * 4 + 2], red[32 + m * 4 + 3]));
                amax = fmaxf(amax, h2f(f2h(1e-6f))); // T localMax = 1e-6f (K/quantization.cu:101)
                qs = 127.f / amax;
                row_scale[m] = amax / 127.f;
                if (blockIdx.x == 0 && tid == 0 && p.dyn_scale_out)
                    p.dyn_scale_out[m] = amax / 127.f;
            }
            else if (SQ && q_static)
                qs = pro_q;
#pragma unroll
            for (int j = 0; j < kNXV; ++j)
            {
                const int k = (tid + j * 256) * 8;
                if (k < Kp)
                {
                    if (SQ)
                    {
                        const uint32_t ws[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w};
                        uint32_t o[2] = {0, 0};
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                        {
                            const h2_t h = u32_as_h2(ws[q]);
                            const uint32_t b0 = (uint8_t) f2i8_rni_sat((float) h.x * qs);
                            const uint32_t b1 = (uint8_t) f2i8_rni_sat((float) h.y * qs);
                            o[q >> 1] |= (b0 | (b1 << 8)) << (16 * (q & 1));
                        }
                        *reinterpret_cast<uint2*>(xq + (size_t) m * Kp + k) = make_uint2(o[0], o[1]);
                    }
                    else
                        *reinterpret_cast<uint4*>(xh + (size_t) m * Kp + k) = xv[j];
                }
            }
        }
    }
    else
    {
        // generic path (unaligned / very long x): through LDS, scalar passes
#pragma unroll
        for (int m = 0; m < MB; ++m)
        {
            if (m >= p.M)
                continue;
            if (x_is_half)
            {
                const uint16_t* xg = reinterpret_cast<const uint16_t*>(p.x) + (int64_t) m * p.ldx;
                uint16_t* xs = xh + (size_t) m * Kp;
                float ss = 0.f;
                for (int k = tid; k < Kp; k += 256)
                {
                    const uint16_t b = k < K ? xg[k] : (uint16_t) 0;
                    xs[k] = b;
                    const float f = h2f(b);
                    ss += f * f;
                }
                float amax = 0.f;
                if (do_norm)
                {
                    ss = block_sum(ss, red);
                    const float inv = 1.0f / sqrtf(ss / (float) K + p.eps);
                    for (int k = tid; k < K; k += 256)
                    {
                        const float n16 = h2f(f2h(h2f(xs[k]) * inv));
                        const uint16_t yb = f2h(n16 * h2f(gam[k]));
                        xs[k] = yb;
                        amax = fmaxf(amax, fabsf(h2f(yb)));
                    }
                }
                else if (q_dyn)
                    for (int k = tid; k < K; k += 256)
                        amax = fmaxf(amax, fabsf(h2f(xs[k])));
                if (SQ && (q_static || q_dyn))
                {
                    float qs;
                    if (q_dyn)
                    {
                        amax = block_max(amax, red);
                        amax = fmaxf(amax, h2f(f2h(1e-6f)));
                        qs = 127.f / amax;
                        row_scale[m] = amax / 127.f;
                        if (blockIdx.x == 0 && tid == 0 && p.dyn_scale_out)
                            p.dyn_scale_out[m] = amax / 127.f;
                    }
                    else
                        qs = p.act_scale[0];
                    int8_t* qd = xq + (size_t) m * Kp;
                    for (int k = tid; k < Kp; k += 256)
                        qd[k] = k < K ? f2i8_rni_sat(h2f(xs[k]) * qs) : (int8_t) 0;
                }
                __syncthreads();
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
    }
    __syncthreads();
    if (blockIdx.x == 0 && p.x_pro_out && p.pro != PRO_NONE)
    {
        for (int m = 0; m < MB && m < p.M; ++m)
        {
            if (SQ)
            {
                int8_t* o = reinterpret_cast<int8_t*>(p.x_pro_out) + (int64_t) m * K;
                for (int k = tid; k < K; k += 256)
                    o[k] = xq[(size_t) m * Kp + k];
            }
            else
            {
                uint16_t* o = reinterpret_cast<uint16_t*>(p.x_pro_out) + (int64_t) m * K;
                for (int k = tid; k < K; k += 256)
                    o[k] = xh[(size_t) m * Kp + k];
            }
        }
    }

    // ------------------------------------------------------------------ main loop
    // Persistent waves: wave w of workgroup b owns row groups g0, g0 + stride, ...; its tiles (U chunks x R rows,
    // 16-byte loads) are double-buffered: while tile t is being reduced, tile t+1 is in flight, and the epilogue
    // operands (scales, residual) of t are requested BEFORE t+1 so that waiting for them never drains the stream.
    const int tiles_per_group = (a.nchunks + U - 1) / U;
    const int ngroups_mine = g0 < a.ngroups ? (a.ngroups - g0 + gstride - 1) / gstride : 0;
    const int ntiles = ngroups_mine * tiles_per_group;
    float my_row_scale = static_row_scale, my_row_scale_up = static_row_scale_up;
    if (q_dyn)
    {
#pragma unroll
        for (int m = 0; m < MB; ++m)
            if (m == my_m)
                my_row_scale = my_row_scale_up = row_scale[m];
    }

    acc_t acc[R][MB];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < MB; ++m)
            acc[r][m] = 0;

    uint4 wv2[U][R];
    int t_issue = 1, gi_i = 0, ci_i = 1; // tile 0 is already in flight in `wv`
    if (ci_i == tiles_per_group)
    {
        ci_i = 0;
        gi_i = 1;
    }
    int gi_p = 0, ci_p = 0;

    auto step = [&](uint4 (&cur)[U][R], uint4 (&nxt)[U][R]) {
        const bool last = ci_p == tiles_per_group - 1;
        const int g = g0 + gi_p * gstride;
        const int n = swiglu ? g * (R / 2) + my_o : g * R + my_o;
        const bool fin = last && my_active && n < p.N;
        // (1) epilogue operands: this group's were requested one group ahead (ops_cur); request the next group's
        const int64_t oidx = (int64_t) my_m * p.ldy + n;
        EpiOps ops_nxt = ops_cur;
        if (last)
            ops_nxt = load_ops(g + gstride);
        // (2) next tile into the other buffer
        if (t_issue < ntiles)
        {
            if (ci_i == 0)
                rows_of_group(g0 + gi_i * gstride, rowptr);
            load_tile(rowptr, ci_i * U, nxt);
            ++t_issue;
            if (++ci_i == tiles_per_group)
            {
                ci_i = 0;
                ++gi_i;
            }
        }
        // (3) dot products of the current tile
        const int c = ci_p * U;
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            int k0 = ((c + u) * 64 + lane) * VEC;
            k0 = k0 < Kp ? k0 : Kp - VEC; // out-of-range lanes: the weight vector was zeroed, any x will do
            {
#pragma unroll
                for (int m = 0; m < MB; ++m)
                {
                    if constexpr (WT == W_FP16)
                    {
                        const uint4 xa = *reinterpret_cast<const uint4*>(xh + (size_t) m * Kp + k0);
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            acc[r][m] = dot_fp16(cur[u][r], xa, acc[r][m]);
                    }
                    else if constexpr (WT == W_INT8_WOQ)
                    {
                        const uint4 xa = *reinterpret_cast<const uint4*>(xh + (size_t) m * Kp + k0);
                        const uint4 xb = *reinterpret_cast<const uint4*>(xh + (size_t) m * Kp + k0 + 8);
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            acc[r][m] = dot_woq8(cur[u][r], xa, xb, acc[r][m]);
                    }
                    else if constexpr (WT == W_INT4_WOQ)
                    {
                        const uint4* xp = reinterpret_cast<const uint4*>(xh + (size_t) m * Kp + k0);
                        const uint4 x0 = xp[0], x1 = xp[1], x2 = xp[2], x3 = xp[3];
#pragma unroll
                        for (int r = 0; r < R; ++r)
                        {
                            float tt = acc[r][m];
                            tt = dot_u4x8(cur[u][r].x, x0, tt);
                            tt = dot_u4x8(cur[u][r].y, x1, tt);
                            tt = dot_u4x8(cur[u][r].z, x2, tt);
                            tt = dot_u4x8(cur[u][r].w, x3, tt);
                            acc[r][m] = tt;
                        }
                    }
                    else
                    {
                        const uint4 xa = *reinterpret_cast<const uint4*>(xq + (size_t) m * Kp + k0);
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            acc[r][m] = dot_sq(cur[u][r], xa, acc[r][m]);
                    }
                }
            }
        }
        if (!last)
        {
            ++ci_p;
            return;
        }
        ci_p = 0;
        ++gi_p;
        // (4) cross-lane reduction (every lane gets every total), then lane (o * MB + m) finishes output o of row m
        float v0 = 0.f, v1 = 0.f;
        int ai = 0;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int m = 0; m < MB; ++m)
            {
                const acc_t tot = wave_sum(acc[r][m]);
                acc[r][m] = 0;
                if (r < nouts && lane == r * MB + m)
                {
                    ai = (int) tot;
                    v0 = (float) tot;
                }
                if (swiglu && r >= R / 2 && lane == (r - R / 2) * MB + m)
                    v1 = (float) tot;
            }
        const EpiOps e = ops_cur;
        ops_cur = ops_nxt;
        if (!fin)
            return;
        const float s0 = e.s0 * my_row_scale, s1 = e.s1 * my_row_scale_up, resv = e.res;
        const float r0 = v0 * s0;
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
            reinterpret_cast<uint16_t*>(p.y)[oidx] = f2h(h2f(f2h(r0)) + resv);
        else
        {
            const float o16 = silu_mul_fp16(r0, v1 * s1);
            if (p.epi == EPI_SWIGLU)
                reinterpret_cast<uint16_t*>(p.y)[oidx] = f2h(o16);
            else
                reinterpret_cast<int8_t*>(p.y)[oidx] = f2i8_rni_sat(o16 * epi_q);
        }
    };

    for (int t = 0; t < ntiles;)
    {
        step(wv, wv2);
        if (++t >= ntiles)
            break;
        step(wv2, wv);
        ++t;
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
    // persistent grid: no more workgroups than the chip holds at once
    static int cus = 0;
    static std::map<size_t, int> occ_cache;
    if (!cus)
    {
        int dev = 0;
        (void) hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
    }
    auto it = occ_cache.find(smem);
    if (it == occ_cache.end())
    {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kfn, 256, smem) != hipSuccess || nb < 1)
            nb = 2;
        it = occ_cache.emplace(smem, nb > 8 ? 8 : nb).first;
    }
    const int max_blocks = cus * (gemv_tune_blocks_per_cu > 0 ? gemv_tune_blocks_per_cu : it->second);
    if (blocks > max_blocks)
    {
        // every wave gets the same number of row groups (a ragged last round costs a whole extra tile time)
        const int waves = max_blocks * 4;
        const int groups_per_wave = (a.ngroups + waves - 1) / waves;
        blocks = (a.ngroups + 4 * groups_per_wave - 1) / (4 * groups_per_wave);
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
        return launch_inst<WT, R, U, 1>(a, blocks, kRedBytes + smem_per_m, stream);
    if (M <= 2)
        return launch_inst<WT, R, U, 2>(a, blocks, kRedBytes + 2 * smem_per_m, stream);
    if (M <= 4)
        return launch_inst<WT, R, U, 4>(a, blocks, kRedBytes + 4 * smem_per_m, stream);
    return launch_inst<WT, R, U, 8>(a, blocks, kRedBytes + 8 * smem_per_m, stream);
}

} // namespace


int launch_gemv(const GemvParams& p, hipStream_t stream)
{
    if (p.M < 1 || p.M > 8 || p.N <= 0 || p.K <= 0)
    {
        set_error("gemv: unsupported M=%d N=%d K=%d (1 <= M <= 8)", p.M, p.N, p.K);
        return -1;
    }
    const bool sq = p.wtype == W_INT8_SQ;
    const bool swiglu = p.epi == EPI_SWIGLU || p.epi == EPI_SWIGLU_QSTATIC;
    const bool quant_pro = (p.pro >= PRO_RMSNORM_QSTATIC && p.pro <= PRO_QDYN) || p.pro == PRO_ATTN_QSTATIC
        || p.pro == PRO_ATTN_QDYN;
    if (!sq && quant_pro)
    {
        set_error("gemv: quantising prologue needs W_INT8_SQ");
        return -1;
    }
    if (p.pro >= PRO_ATTN)
    {
        if (!p.attn_ml || !p.attn_o || !p.attn_seq_len || p.attn_heads * p.attn_dh != p.K || (p.attn_dh & 7)
            || p.attn_tchunk <= 0 || p.attn_nsmax <= 0 || p.K > 256 * 8 * 6)
        {
            set_error("gemv: PRO_ATTN needs the split-KV partials (heads * dh == K <= 12288, dh %% 8 == 0)");
            return -1;
        }
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
    if (kRedBytes + mb * smem_per_m > 160 * 1024)
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
    const int max_blocks = 256 * 16;
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
