// Skinny GEMM / GEMV for decode (M <= 8): the HBM-bound heart of the per-token path.
//
//   y[m,n] = epi( scale(n,m) * sum_k pro(x)[m,k] * W[n,k] )
//
// One wave owns 2 weight rows per step and streams them with 16-byte non-temporal loads (1 KiB per wave
// instruction, 4 k-chunks per row = an 8 KiB tile, double-buffered); pro(x) is built once per workgroup in
// registers and published to LDS (RMSNorm and/or int8 quantisation fused in, so the normalised / quantised activation
// never goes to HBM); the dot products use v_dot2_f32_f16
// (fp16 x fp16 -> fp32) or v_dot4_i32_i8 (SmoothQuant, exact int32); reduction over the 64 lanes on the DPP
// network; residual-add / SwiGLU / quantising epilogues fused.  Persistent grid (<= what the chip holds at once).
//
// The kernel is specialised at compile time on (weight type, prologue family, epilogue family, row-count bucket):
// one generic kernel with run-time switches was 14k instructions and its cold instruction fetch showed up as a
// ~2 us floor on every launch.
//
// Reference semantics: A7 P/gemmPlugin/gemmPlugin.cpp:121-190; A8 K/weightOnlyMatrixVectorMultiplication.cu:136-277
// (y = sum_k x[k] * (q[k,n] * s[n])); A10 cutlass_extensions/.../epilogue_per_row_per_col_scale.h:279-347
// (C = cvt(float(acc_i32) * (alpha_col * alpha_row))); A5 PY/functional.py:3195-3219; A6 PY/layers/mlp.py:68-73;
// A11 K/quantization.cu:31-118, K/layernormKernels.cu:146-183.
#pragma once
#include "dev_utils.h"
#include "kernels.h"
#include "launch_util.h"
#include "weight_layout.h"
#include "gemv_args.h"
#include <algorithm>
#include <map>
#include <mutex>

// Included by one translation unit per weight type (gemv_fp16.hip, gemv_woq8.hip, gemv_woq4.hip, gemv_sq.hip) so that
// the ~170 kernel instantiations compile in parallel; everything here has internal linkage.
namespace tllm
{
namespace kernels
{
namespace
{
using namespace dev;

template <int WT>
struct WTraits;
template <>
struct WTraits<W_FP16>
{
    static constexpr int VEC = 8;
    static constexpr bool IS_SQ = false;
    static constexpr uint32_t ZERO = 0u;
};
template <>
struct WTraits<W_INT8_WOQ>
{
    static constexpr int VEC = 16;
    static constexpr bool IS_SQ = false;
    static constexpr uint32_t ZERO = 0x80808080u; // q + 128
};
template <>
struct WTraits<W_INT4_WOQ>
{
    static constexpr int VEC = 32;
    static constexpr bool IS_SQ = false;
    static constexpr uint32_t ZERO = 0x88888888u; // q + 8
};
template <>
struct WTraits<W_INT8_SQ>
{
    static constexpr int VEC = 16;
    static constexpr bool IS_SQ = true;
    static constexpr uint32_t ZERO = 0u;
};

// ---- per-16-byte dot products -----------------------------------------------------------------
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
    const h2_t bias = {(_Float16) 1152.f, (_Float16) 1152.f}; // 1024 + 128
    const h2_t l = u32_as_h2(__builtin_amdgcn_perm(magic, w, 0x04010400u)) - bias;
    const h2_t h = u32_as_h2(__builtin_amdgcn_perm(magic, w, 0x04030402u)) - bias;
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

// (dot_u8x4_raw / dot_woq8_raw - the same on the RAW splice - live in dev_utils.h: qkv_attn_fused.hip restates this path)

// 8 nibbles (layout of weight_layout.h) vs 8 halfs of x (one uint4)
__device__ __forceinline__ float dot_u4x8(uint32_t w, const uint4& x, float acc)
{
    const uint32_t m = 0x64006400u;
    const uint32_t w8 = w >> 8;
    const h2_t b0 = {(_Float16) 1032.f, (_Float16) 1032.f}; // 1024 + 8
    const h2_t s1 = {(_Float16) 0.0625f, (_Float16) 0.0625f};
    const h2_t b1 = {(_Float16) -72.f, (_Float16) -72.f}; // (1024 + 16 n) / 16 - 72 = n - 8
    const h2_t e01 = u32_as_h2((w & 0x000f000fu) | m) - b0;
    const h2_t e23 = u32_as_h2((w & 0x00f000f0u) | m) * s1 + b1;
    const h2_t e45 = u32_as_h2((w8 & 0x000f000fu) | m) - b0;
    const h2_t e67 = u32_as_h2((w8 & 0x00f000f0u) | m) * s1 + b1;
    acc = __builtin_amdgcn_fdot2(e01, u32_as_h2(x.x), acc, false);
    acc = __builtin_amdgcn_fdot2(e23, u32_as_h2(x.y), acc, false);
    acc = __builtin_amdgcn_fdot2(e45, u32_as_h2(x.z), acc, false);
    acc = __builtin_amdgcn_fdot2(e67, u32_as_h2(x.w), acc, false);
    return acc;
}

// (dot_u4x8_raw - the half-raw int4 splice - lives in dev_utils.h next to dot_woq8_raw: qkv_attn_fused.hip restates this path too)

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
// LDS map: [0,256) reduction scratch | MB rows of Kp activations (fp16, or s8 for SmoothQuant)
//
// Latency structure (the per-launch floor matters: a 7B layer is 4 launches of 17-90 MB, i.e. 3-15 us each at HBM
// speed).  Everything that does not depend on x is requested at t = 0, in one memory round trip: x / gamma, the first
// weight tile of every wave, the epilogue operands (scales, residual) of its first row group, the launch-constant
// scales.  No branches around loads (a lane-dependent `if` makes the compiler fence each load with s_waitcnt + exec
// masking): out-of-range lanes load a clamped, valid address and the value is replaced by a select.  Then: pro(x) in
// registers (sum of squares -> ONE barrier -> normalise / quantise) -> LDS -> ONE barrier -> dots -> DPP reduction ->
// epilogue.  Further tiles are double-buffered, with the next group's epilogue operands requested ahead of the next
// tile so that waiting for them never drains the weight stream.
template <int WT, int PK, int EK, int MB, int kNXV, int UU = tllm::kernels::gemv_detail::U>
__global__ __launch_bounds__(256) void gemv_kernel(const GemvArgs a)
{
    // chunks per tile: 4, or 2 for rows of at most 2 KiB (int4 K = 4096): a 4-chunk tile would be half neutral elements -
    // half the load slots, LDS reads and dequantisation arithmetic wasted
    constexpr int U = UU;
    using TR = WTraits<WT>;
    constexpr int VEC = TR::VEC;
    constexpr bool SQ = TR::IS_SQ;
    constexpr bool SWIGLU = EK == EK_SWIGLU;
    constexpr bool X_HALF = !(SQ && PK == PK_COPY); // the input activations are fp16 (else raw s8)
    using acc_t = typename std::conditional<SQ, int, float>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GemvParams& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int K = p.K, Kp = a.Kp;
    float* red = reinterpret_cast<float*>(smem);
    char* xs = smem + kRedBytes; // [MB][Kp] halfs, or [MB][Kp] s8 for SQ
    constexpr int XES = SQ ? 1 : 2;
    const bool q_dyn = SQ && (p.pro == PRO_RMSNORM_QDYN || p.pro == PRO_QDYN);
    const bool q_static = SQ && (p.pro == PRO_RMSNORM_QSTATIC || p.pro == PRO_QSTATIC);

    // ------------------------------------------------------------------ weight-tile helpers
    const char* wbase = reinterpret_cast<const char*>(p.w);
    const char* wup = p.w_up ? reinterpret_cast<const char*>(p.w_up) : wbase + (int64_t) p.N * p.ldw;
    const int lane_kbyte = lane * 16;
    const int64_t last_vec = p.ldw - 16;
    auto rows_of_group = [&](int g, const char* (&rowptr)[R]) {
        if constexpr (SWIGLU)
        {
            const int o = g < p.N ? g : p.N - 1; // one output per group: gate row o, up row o
            rowptr[0] = wbase + (int64_t) o * p.ldw;
            rowptr[1] = wup + (int64_t) o * p.ldw;
        }
        else
        {
#pragma unroll
            for (int r = 0; r < R; ++r)
            {
                const int row = g * R + r;
                rowptr[r] = wbase + (int64_t) (row < p.N ? row : p.N - 1) * p.ldw;
            }
        }
    };
    auto load_tile = [&](const char* const (&rowptr)[R], int c, uint4 (&wv)[U][R]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            // pure loads, nothing consumes them here (an operation on a loaded value inside a conditional block makes the
            // compiler wait for the load inside that block): lanes beyond the row read a clamped, valid address and are
            // neutralised when the tile is consumed - by zeroing the ACTIVATIONS that face them
            int64_t off = (int64_t) (c + u) * 1024 + lane_kbyte;
            off = off < last_vec ? off : last_vec;
#pragma unroll
            for (int r = 0; r < R; ++r)
                wv[u][r] = ld_nt16(rowptr[r] + off);
        }
    };

    // ------------------------------------------------------------------ t = 0: every x-independent request
    uint4 xv[kNXV], gv[kNXV];
    auto load_x_row = [&](int m) {
        if constexpr (X_HALF)
        {
            const uint16_t* xg = reinterpret_cast<const uint16_t*>(p.x) + (int64_t) m * p.ldx;
#pragma unroll
            for (int j = 0; j < kNXV; ++j)
            {
                // a pure load from a clamped address, no branch and nothing consuming the value here: a select on the loaded
                // value inside a conditional block made hipcc wait for each of these loads in turn - two serialised round
                // trips before the first weight tile was even requested.  Vectors beyond K are zeroed by mask_x().
                const int k = (tid + j * 256) * 8;
                xv[j] = *reinterpret_cast<const uint4*>(xg + (k < K ? k : K - 8));
            }
        }
        else
        {
            // raw s8 activations: 16 values per vector
            const int8_t* xg = reinterpret_cast<const int8_t*>(p.x) + (int64_t) m * p.ldx;
#pragma unroll
            for (int j = 0; j < kNXV; ++j)
            {
                const int k = (tid + j * 256) * 16;
                xv[j] = *reinterpret_cast<const uint4*>(xg + (k < K ? k : K - 16));
            }
        }
    };
    // zero the activation vectors beyond K (after the loads have been issued)
    auto mask_x = [&]() {
        constexpr int XVEC = X_HALF ? 8 : 16;
#pragma unroll
        for (int j = 0; j < kNXV; ++j)
        {
            const bool ok = (tid + j * 256) * XVEC < K;
            xv[j] = make_uint4(ok ? xv[j].x : 0u, ok ? xv[j].y : 0u, ok ? xv[j].z : 0u, ok ? xv[j].w : 0u);
        }
    };
    load_x_row(0);
    if constexpr (PK == PK_NORM)
    {
        const uint16_t* gam = reinterpret_cast<const uint16_t*>(p.gamma);
#pragma unroll
        for (int j = 0; j < kNXV; ++j)
        {
            const int k = (tid + j * 256) * 8;
            gv[j] = *reinterpret_cast<const uint4*>(gam + (k < K ? k : K - 8)); // multiplies a zeroed x beyond K
        }
    }
    const int g0 = blockIdx.x * 4 + wid;
    const int gstride = gridDim.x * 4;
    const char* rowptr[R];
    uint4 wv[U][R], wv2[U][R];
    rows_of_group(g0 < a.ngroups ? g0 : 0, rowptr);
    load_tile(rowptr, 0, wv);
    // the tile stream of this wave: (group, chunk) of the next tile to request.  (Requesting the second ring buffer here too,
    // before the prologue, was measured: slower on every shape - QKV 10.3 -> 11.2 us, gate|up 16.5 -> 18.0 us.)
    const int tiles_per_group = (a.nchunks + U - 1) / U;
    const int ngroups_mine = g0 < a.ngroups ? (a.ngroups - g0 + gstride - 1) / gstride : 0;
    const int ntiles = ngroups_mine * tiles_per_group;
    int t_issue = 1, gi_i = 0, ci_i = 1; // tile 0 is in flight in `wv`
    if (ci_i == tiles_per_group)
    {
        ci_i = 0;
        gi_i = 1;
    }
    auto issue_next = [&](uint4 (&buf)[U][R]) {
        if (t_issue < ntiles) // wave-uniform
        {
            if (ci_i == 0)
                rows_of_group(g0 + gi_i * gstride, rowptr);
            load_tile(rowptr, ci_i * U, buf);
            ++t_issue;
            if (++ci_i == tiles_per_group)
            {
                ci_i = 0;
                ++gi_i;
            }
        }
    };

    constexpr int NOUTS = SWIGLU ? 1 : R;         // outputs per row group
    const int my_o = lane / MB, my_m = lane % MB; // the (output, row) this lane finishes
    const bool my_active = my_o < NOUTS && my_m < p.M;
    // raw bits of the epilogue operands: converting a loaded fp16 here, inside the conditional block that loads it, makes
    // hipcc wait for the load (and, the counter being in-order, for every load before it) on the spot
    struct EpiOps
    {
        float s0, s1;
        uint16_t h0, h1, res;
    };
    auto load_ops = [&](int g) {
        EpiOps e = {1.f, 1.f, 0, 0, 0};
        int n = g * NOUTS + (my_o < NOUTS ? my_o : 0);
        n = n < p.N ? n : p.N - 1; // clamped: inactive lanes load a valid element and ignore it
        if constexpr (WT == W_INT8_WOQ || WT == W_INT4_WOQ)
        {
            const uint16_t* sc = reinterpret_cast<const uint16_t*>(p.scale_col);
            e.h0 = sc[n];
            if constexpr (SWIGLU)
                e.h1 = p.scale_col_up ? reinterpret_cast<const uint16_t*>(p.scale_col_up)[n] : sc[p.N + n];
        }
        else if constexpr (SQ)
        {
            const float* sc = reinterpret_cast<const float*>(p.scale_col);
            e.s0 = sc[p.per_channel ? n : 0];
            if constexpr (SWIGLU)
            {
                const float* su = reinterpret_cast<const float*>(p.scale_col_up);
                e.s1 = su ? su[p.per_channel ? n : 0] : sc[p.per_channel ? p.N + n : 0];
            }
        }
        if constexpr (!SWIGLU)
        {
            // unconditional (a load inside a conditional block ends in a full s_waitcnt): without a residual the address is
            // y's own, valid and ignored
            const uint16_t* rp = reinterpret_cast<const uint16_t*>(p.epi == EPI_RESIDUAL ? p.residual : p.y);
            e.res = rp[(int64_t) (my_m < p.M ? my_m : 0) * p.ldy + n];
        }
        return e;
    };
    EpiOps ops_cur = load_ops(g0 < a.ngroups ? g0 : 0);
    float static_row_scale = 1.f, static_row_scale_up = 1.f, epi_q = 1.f, pro_q = 1.f;
    if constexpr (SQ)
    {
        if (!q_dyn && p.scale_row) // uniform
            static_row_scale = p.scale_row[(p.per_token && my_m < p.M) ? my_m : 0];
        static_row_scale_up = (!q_dyn && p.scale_row_up) ? p.scale_row_up[0] : static_row_scale;
        if (q_static)
            pro_q = p.act_scale[0];
    }
    if constexpr (SWIGLU)
    {
        if (p.epi == EPI_SWIGLU_QSTATIC)
            epi_q = p.epi_scale[0];
    }

    // ------------------------------------------------------------------ pro(x) -> LDS
    float row_scale[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
        row_scale[m] = 1.f;
#pragma unroll
    for (int m = 0; m < MB; ++m)
    {
        if (m >= p.M) // uniform
            continue;
        if (m > 0)
            load_x_row(m);
        mask_x();
        if constexpr (!X_HALF)
        {
#pragma unroll
            for (int j = 0; j < kNXV; ++j)
            {
                const int k = (tid + j * 256) * 16;
                if (k < Kp)
                    *reinterpret_cast<uint4*>(xs + (size_t) m * Kp + k) = xv[j];
            }
            continue;
        }
        float inv = 1.f;
        if constexpr (PK == PK_NORM)
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
        if (PK == PK_NORM || q_dyn)
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
                    if constexpr (PK == PK_NORM)
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
        float qs = pro_q;
        if constexpr (SQ)
        {
            if (q_dyn) // uniform
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
        }
        if constexpr (WT == W_INT8_WOQ || WT == W_INT4_WOQ)
        {
            // sums of the activations the dots will see (fp16 values, fp32 sums): the biases of the raw weight splices come off
            // once per row.  int4: separately over the halves that face the two kinds of splice (words x, z / y, w of a vector)
            float sa = 0.f, sb = 0.f;
#pragma unroll
            for (int j = 0; j < kNXV; ++j)
            {
                const uint32_t ws[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                {
                    const h2_t h = u32_as_h2(ws[q]);
                    if (q & 1)
                        sb += (float) h.x + (float) h.y;
                    else
                        sa += (float) h.x + (float) h.y;
                }
            }
            const float bias = WT == W_INT8_WOQ ? 1152.f * (sa + sb) : 72.f * sb; // int4: only the 1024 + 16 n splices stay raw
            const float bsum = wave_sum(bias);
            if (lane == 0)
                red[64 + m * 4 + wid] = bsum; // read behind the barrier that publishes the activations
        }
#pragma unroll
        for (int j = 0; j < kNXV; ++j)
        {
            const int k = (tid + j * 256) * 8;
            if (k < Kp)
            {
                if constexpr (SQ)
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
                    *reinterpret_cast<uint2*>(xs + (size_t) m * Kp + k) = make_uint2(o[0], o[1]);
                }
                else
                    *reinterpret_cast<uint4*>(xs + ((size_t) m * Kp + k) * 2) = xv[j];
            }
        }
    }
    __syncthreads();
    if (blockIdx.x == 0 && p.x_pro_out && PK != PK_COPY)
    {
        for (int m = 0; m < MB && m < p.M; ++m)
            for (int k = tid; k < K * XES; k += 256)
                reinterpret_cast<char*>(p.x_pro_out)[(int64_t) m * K * XES + k] = xs[(size_t) m * Kp * XES + k];
    }

    // ------------------------------------------------------------------ main loop: persistent waves, double buffer
    float my_xbias = 0.f; // weight-only: the splice bias of row my_m (int8: 1152 sum x; int4: 72 sum x_b)
    if constexpr (WT == W_INT8_WOQ || WT == W_INT4_WOQ)
    {
#pragma unroll
        for (int m = 0; m < MB; ++m)
            if (m == my_m)
                my_xbias = red[64 + m * 4] + red[64 + m * 4 + 1] + red[64 + m * 4 + 2] + red[64 + m * 4 + 3];
    }
    float my_row_scale = static_row_scale, my_row_scale_up = static_row_scale_up;
    if (q_dyn)
    {
#pragma unroll
        for (int m = 0; m < MB; ++m)
            if (m == my_m)
                my_row_scale = my_row_scale_up = row_scale[m];
    }

    acc_t acc[R][MB];
    float acc16[WT == W_INT4_WOQ ? R : 1][WT == W_INT4_WOQ ? MB : 1]; // int4: the sum over the 1024 + 16 n splices
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < MB; ++m)
        {
            acc[r][m] = 0;
            if constexpr (WT == W_INT4_WOQ)
                acc16[r][m] = 0.f;
        }

    int gi_p = 0, ci_p = 0;

    auto step = [&](uint4 (&cur)[U][R], uint4 (&nxt)[U][R]) {
        const bool last = ci_p == tiles_per_group - 1;
        const int g = g0 + gi_p * gstride;
        const int n = g * NOUTS + (my_o < NOUTS ? my_o : 0);
        const bool fin = last && my_active && n < p.N;
        const int64_t oidx = (int64_t) my_m * p.ldy + n;
        // (1) the next group's epilogue operands
        EpiOps ops_nxt = ops_cur;
        if (last)
            ops_nxt = load_ops(g + gstride < a.ngroups ? g + gstride : g);
        // (2) the next tile, into the other ring buffer
        issue_next(nxt);
        // (3) dot products of the current tile
        const int c = ci_p * U;
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            int k0 = ((c + u) * 64 + lane) * VEC;
            const bool ok = k0 < Kp; // out-of-range lanes: zero activations (any weight bits are then harmless)
            k0 = ok ? k0 : Kp - VEC;
            auto ldx = [&](const char* ptr) {
                const uint4 v = *reinterpret_cast<const uint4*>(ptr);
                return make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
            };
#pragma unroll
            for (int m = 0; m < MB; ++m)
            {
                const char* xr = xs + ((size_t) m * Kp + k0) * XES;
                if constexpr (WT == W_FP16)
                {
                    const uint4 xa = ldx(xr);
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        acc[r][m] = dot_fp16(cur[u][r], xa, acc[r][m]);
                }
                else if constexpr (WT == W_INT8_WOQ)
                {
                    const uint4 xa = ldx(xr);
                    const uint4 xb = ldx(xr + 16);
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        acc[r][m] = dot_woq8_raw(cur[u][r], xa, xb, acc[r][m]);
                }
                else if constexpr (WT == W_INT4_WOQ)
                {
                    const uint4 x0 = ldx(xr), x1 = ldx(xr + 16), x2 = ldx(xr + 32), x3 = ldx(xr + 48);
#pragma unroll
                    for (int r = 0; r < R; ++r)
                    {
                        float ta = acc[r][m], tb = acc16[r][m];
                        dot_u4x8_raw(cur[u][r].x, x0, ta, tb);
                        dot_u4x8_raw(cur[u][r].y, x1, ta, tb);
                        dot_u4x8_raw(cur[u][r].z, x2, ta, tb);
                        dot_u4x8_raw(cur[u][r].w, x3, ta, tb);
                        acc[r][m] = ta;
                        acc16[r][m] = tb;
                    }
                }
                else
                {
                    const uint4 xa = ldx(xr);
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        acc[r][m] = dot_sq(cur[u][r], xa, acc[r][m]);
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
                acc_t tot = wave_sum(acc[r][m]);
                acc[r][m] = 0;
                if constexpr (WT == W_INT4_WOQ)
                {
                    tot += wave_sum(acc16[r][m]) * 0.0625f;
                    acc16[r][m] = 0.f;
                }
                if (r < NOUTS && lane == r * MB + m)
                {
                    ai = (int) tot;
                    v0 = (float) tot;
                }
                if (SWIGLU && r == 1 && lane == m)
                    v1 = (float) tot;
                if constexpr (WT == W_INT8_WOQ || WT == W_INT4_WOQ)
                {
                    // (my_m == m on the lanes that keep a value)
                    if (r < NOUTS && lane == r * MB + m)
                        v0 -= my_xbias;
                    if (SWIGLU && r == 1 && lane == m)
                        v1 -= my_xbias;
                }
            }
        EpiOps e = ops_cur;
        ops_cur = ops_nxt;
        if (!fin)
            return;
        if constexpr (WT == W_INT8_WOQ || WT == W_INT4_WOQ)
        {
            e.s0 = h2f(e.h0);
            e.s1 = h2f(e.h1);
        }
        const float r0 = v0 * (e.s0 * my_row_scale);
        if constexpr (SWIGLU)
        {
            const float o16 = silu_mul_fp16(r0, v1 * (e.s1 * my_row_scale_up));
            if (p.epi == EPI_SWIGLU)
                reinterpret_cast<uint16_t*>(p.y)[oidx] = f2h(o16);
            else
                reinterpret_cast<int8_t*>(p.y)[oidx] = f2i8_rni_sat(o16 * epi_q);
        }
        else
        {
            if (p.epi == EPI_RESIDUAL)
                reinterpret_cast<uint16_t*>(p.y)[oidx] = f2h(h2f(f2h(r0)) + h2f(e.res));
            else if (p.out_dtype == DT_HALF)
                reinterpret_cast<uint16_t*>(p.y)[oidx] = f2h(r0);
            else if (p.out_dtype == DT_FLOAT)
                reinterpret_cast<float*>(p.y)[oidx] = r0;
            else
                reinterpret_cast<int32_t*>(p.y)[oidx] = f2i32_rni_sat(r0);
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

template <int WT, int PK, int EK, int MB, int NXV, int UU = U>
int launch_inst(const GemvArgs& a, hipStream_t stream)
{
    auto kfn = gemv_kernel<WT, PK, EK, MB, NXV, UU>;
    const size_t smem = kRedBytes + (size_t) MB * a.Kp * (WT == W_INT8_SQ ? 1 : 2);
    if (smem > 160 * 1024)
    {
        set_error("gemv: K=%d x M=%d does not fit LDS", a.p.K, a.p.M);
        return -1;
    }
    // per-instantiation, per-DEVICE launch state (LDS attribute, CU count, occupancy per LDS size) behind launch_util's mutex:
    // plugins may be enqueued from several host threads (one per rank is the rule, but nothing forbids more)
    launch_util::ensure_dynamic_lds(reinterpret_cast<const void*>(kfn), smem > 64 * 1024 ? 160 * 1024 : 0);
    // persistent grid: no more workgroups than the chip holds at once, every wave the same number of row groups
    const int cus = launch_util::device_cus();
    int fit_blocks = launch_util::blocks_per_cu(reinterpret_cast<const void*>(kfn), 256, smem);
    fit_blocks = fit_blocks < 1 ? 2 : (fit_blocks > 8 ? 8 : fit_blocks);
    int blocks = (a.ngroups + 3) / 4;
    // persistent workgroups per CU: what fits (occupancy query) for the SwiGLU kernel (gate|up: 16.4 us with 4 per CU, 17.6 with
    // 2); two for the plain projections (QKV: 9.9 us with 2, 10.4 with 3 - 4: fewer, longer-lived waves amortise the RMSNorm
    // prologue over three row groups instead of two)
    const int fit = fit_blocks;
    const int want = EK == EK_SWIGLU ? fit : (fit < 2 ? fit : 2);
    const int max_blocks = cus * (gemv_tune_blocks_per_cu > 0 ? gemv_tune_blocks_per_cu : want);
    if (blocks > max_blocks)
    {
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

template <int WT, int PK, int EK, int NXV>
int launch_mb(const GemvArgs& a, hipStream_t stream)
{
    if (a.p.M <= 1)
    {
        if constexpr (WT == W_INT4_WOQ)
            if (a.nchunks <= 2)
                return launch_inst<WT, PK, EK, 1, NXV, 2>(a, stream);
        return launch_inst<WT, PK, EK, 1, NXV>(a, stream);
    }
    if (a.p.M <= 2)
        return launch_inst<WT, PK, EK, 2, NXV>(a, stream);
    if (a.p.M <= 4)
        return launch_inst<WT, PK, EK, 4, NXV>(a, stream);
    return launch_inst<WT, PK, EK, 8, NXV>(a, stream);
}

template <int WT, int PK, int EK>
int launch_nxv(const GemvArgs& a, hipStream_t stream)
{
    // activation vectors per thread: 256 threads x 16 bytes each (8 halfs, or 16 raw s8 values)
    const int xvec = (WT == W_INT8_SQ && PK == PK_COPY) ? 16 : 8;
    if (a.p.K <= 256 * xvec * kNXVSmall)
        return launch_mb<WT, PK, EK, kNXVSmall>(a, stream);
    if (a.p.K <= 256 * xvec * kNXVMax)
        return launch_mb<WT, PK, EK, kNXVMax>(a, stream);
    if constexpr ((PK == PK_COPY || PK == PK_QUANT) && EK == EK_PLAIN)
        return launch_mb<WT, PK, EK, kNXVLarge>(a, stream);
    set_error("gemv: K=%d exceeds %d, the limit of the normalising / merging prologues and the SwiGLU epilogue", a.p.K, 256 * xvec * kNXVMax);
    return -1;
}

template <int WT>
int launch_wt(const GemvArgs& a, int pk, bool swiglu, hipStream_t stream)
{
    constexpr bool SQ = WT == W_INT8_SQ;
    if (swiglu)
    {
        switch (pk)
        {
        case PK_COPY: return launch_nxv<WT, PK_COPY, EK_SWIGLU>(a, stream);
        case PK_NORM: return launch_nxv<WT, PK_NORM, EK_SWIGLU>(a, stream);
        default: break;
        }
        set_error("gemv: SwiGLU epilogue is built with the copy / RMSNorm prologues only");
        return -1;
    }
    switch (pk)
    {
    case PK_COPY: return launch_nxv<WT, PK_COPY, EK_PLAIN>(a, stream);
    case PK_NORM: return launch_nxv<WT, PK_NORM, EK_PLAIN>(a, stream);
    case PK_QUANT:
        if constexpr (SQ)
            return launch_nxv<WT, PK_QUANT, EK_PLAIN>(a, stream);
        break;
    default: break;
    }
    set_error("gemv: unsupported prologue for this weight type");
    return -1;
}

} // namespace
} // namespace kernels
} // namespace tllm
