// Decode step, batch 1: the QKV projection of head h, RoPE, the KV-cache append and the masked attention of head h in ONE launch
// (SURVEY §8a rows A5 + A11 + A10 -> A1 / A2 / A3).  gfx950, SmoothQuant int8 weights.
//
// Reference: the generation phase of Attention.forward (PY/layers/attention.py: qkv = self.qkv(hidden) -> gpt_attention plugin) =
// SmoothQuantGemm [1, D] x [3 D, D]^T, then masked_multihead_attention_kernel (MM/decoderMaskedMultiheadAttentionTemplate.h:
// 1352-1389 RoPE, 1493-1549 cache append / current-token score, 1553-1800 the loops over the cache, 2019-2181 the multi-block
// reduction).  The reference runs them as two graph nodes; so did rounds 1 - 4 here (gemv_kernel<.., PK_NORM, EK_PLAIN> 9.8 us +
// mmha_partial_kernel 9.2 us per layer at the 7B shape).  This is the one seam of the decoder layer that is NOT all-to-all:
// head h's attention needs exactly the 3 x 128 projection outputs of head h.
//
// Structure (32 heads x 8 members = 256 workgroups of 8 waves, one per CU; member m of head h is workgroup m * H + h, so the
// eight members of a head share an XCD - for speed only, nothing depends on it):
//   t = 0   every wave requests EVERYTHING that does not depend on another workgroup, in one go: x and gamma, its 6 weight
//           rows (2 q rows, 2 k rows, 2 v rows of the member's 3 x 16: 24 KB per wave, 59 MB on the chip - the whole stream is in
//           flight at once, the memory system drains it at its own rate), the member's 1/8 of the head's cache range (NIT rows of K
//           and of V per lane group, 16 bytes per lane), masks, RoPE row, scales.  The cache round trip that stood alone in its
//           own launch at ~1 TB/s now hides under the weight stream.
//   1       RMSNorm + int8 quantiser of x in registers -> LDS (the arithmetic of gemv_impl.h's PK_NORM prologue, same summation
//           order: the int8 operand is bit-identical to the unfused launch's).
//   2       q rows -> dot (v_dot4_i32_i8, exact) -> per-channel x per-tensor dequantisation -> ONE 8-byte {tag, 2 x fp16} granule
//           per wave, written through (sc1): the data is the flag (guide G16 R2).  Then the k rows, then the v rows.
//   3       wave 0 of every member sweeps the head's 64 q granules (RoPE applied in the sweeping lanes), everyone computes the
//           scores / softmax / P.V of its cache rows against q, merges its 64 lane groups (DPP inside a wave, LDS across) and
//           publishes ONE partial {m, l, o[128]} as tagged granules.
//   4       member 0 of the head sweeps the k and v granules and the eight partials, adds the CURRENT token as a ninth partial
//           (score q.k_new with the un-quantised k, weight exp(.), value v_new: MM/...Template.h:1517-1549), normalises, writes the
//           context row (fp16 + its static int8 image for the O-projection) and appends k_new / v_new to the cache.
// Two in-launch hand-offs (q inside the head; partials + k, v to the head's merger), no ticket, no drain, no kernel boundary
// between the projection and the attention.  Every wait is bounded; a time-out raises the error word and the launch ends.
//
// Tags: tag = step_epoch * tag_mul + tag_add (tag_add = layer + 1), step_epoch a device word the sampler advances once per
// generation step - unique per launch, never 0, valid under graph replay (a kernel argument would be frozen).  The exchange
// buffer is shared by all layers (the kernel boundary between two layers orders reuse) and zeroed at session setup.
//
// Numerics: SURVEY Appendix A.1 as in mmha_decode.hip (RoPE fp32 -> fp16; int8 cache = sat(rni(fp16 * s)); cached K / V as exact
// integers with the scale folded into the sums; p rounded to fp16 before P.V; fp32 partials; 1 / (sum + 1e-6); one fp16 rounding).
// The current token enters as its own partial (p = 1 exactly) instead of inside a lane group's partial: a re-association of the
// fp32 merge, nothing else.
#include "dev_utils.h"
#include "kernels.h"
#include <atomic>
#include <map>
#include <math.h>
#include <mutex>
#include <type_traits>

namespace tllm
{
namespace kernels
{
using namespace dev;

namespace
{

typedef __attribute__((address_space(1))) unsigned long long gu64;

constexpr int kDH = 128;       // head size
constexpr int kMembers = 8;    // workgroups per head
constexpr int kWavesF = 8;     // waves per workgroup
constexpr int kKChunks = 4;    // K = 4096 int8 = 4 x 1 KiB per weight row
constexpr int kPartStride = 136; // granules per published partial: o[128], m, l (+ pad)
constexpr int kHeadGranules = 3 * 64 + kMembers * kPartStride;
constexpr int kCtxGranules = kDH / 4; // the head's context row as its int8 image, four elements per granule (O-projection stage)
constexpr int kCtxGranulesH = kDH / 2; // ... as fp16, two elements per granule (weight-only engines)
constexpr int kORowsMax = 24;         // dense-projection rows a row-worker workgroup takes at most (3 per wave)

__device__ __forceinline__ void st_granule(gu64* g, uint32_t tag, uint32_t value)
{
    __hip_atomic_store(g, ((unsigned long long) tag << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_granule(const gu64* g)
{
    return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one LDS-DMA instruction: 64 lanes x 16 bytes, global (per-lane address) -> LDS [lds_byte + lane * 16].  (M0 cannot go on the
// clobber list - hipcc: "reserved register, may not be preserved" - and __builtin_amdgcn_global_load_lds would hand the waits
// to the compiler's own vmcnt accounting: the statement saves M0 and puts it back itself)
__device__ __forceinline__ void glds16(const void* gptr, uint32_t lds_byte)
{
    uint32_t m0_keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(m0_keep) : "v"(gptr), "s"(lds_byte) : "memory");
}

// RoPE (NeoX pairs (d, d + 64)) of the element pair (2l, 2l + 1) this lane of the sweeping wave holds: the partner pair sits in
// lane l ^ 32.  cs = {cos(2l'), sin(2l'), cos(2l' + 1), sin(2l' + 1)}, l' = l & 31.  Rounding: fp32 -> fp16 (...Utils.h:1517-1531)
__device__ __forceinline__ uint32_t rope_pair(uint32_t raw, const float4& cs, bool second)
{
    const h2_t v = u32_as_h2(raw);
    const float x0 = (float) v.x, x1 = (float) v.y;
    const float p0 = __shfl_xor(x0, 32, 64), p1 = __shfl_xor(x1, 32, 64);
    const float s0 = second ? cs.y : -cs.y, s1 = second ? cs.w : -cs.w;
    const uint32_t lo = f2h(cs.x * x0 + s0 * p0), hi = f2h(cs.z * x1 + s1 * p1);
    return lo | (hi << 16);
}

// optional stage clock (FusedQkvAttnParams::timing, tools/fused_timeline.py): the 100 MHz constant counter, one row per workgroup
#define TLLM_STAMP(slot)                                                                                               \
    do                                                                                                                 \
    {                                                                                                                  \
        if (p.timing && lane == 0 && wid == 0)                                                                         \
            p.timing[(size_t) blockIdx.x * 16 + (slot)] = wall_clock64();                                              \
    } while (0)

// v + (the value 8 / 16 / 32 lanes away), on the VALU: row rotate inside a 16-lane row, v_permlane16_swap / v_permlane32_swap of
// the value with itself (rows 1 <-> 0 and 3 <-> 2; halves) - no LDS crossbar round trip, no wait
__device__ __forceinline__ float add_xor8(float v)
{
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
}
// (inline asm, pads inside the string: with the builtin hipcc (ROCm 7.2) loses the swap's SECOND result when both operands
// carry the same value - `v_permlane16_swap v1, v2 ; v_add_f32 v1, v1, v1` for r[0] + r[1], even behind an opaque copy)
__device__ __forceinline__ void swap16(float v, float& a, float& b)
{
    a = v;
    b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap32(float v, float& a, float& b)
{
    a = v;
    b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
// TWO values per swap: with a = X, b = Y the swap leaves X's row pairs in a's even rows / b's even rows and Y's in the odd ones, so
// a' + b' holds (X + X^16) in rows 0, 2 and (Y + Y^16) in rows 1, 3 - one swap and one add reduce two values, no copies
__device__ __forceinline__ float pair_xor16(float x, float y)
{
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}
// likewise over the wave's halves: the result holds (X + X^32) in lanes 0 - 31 and (Y + Y^32) in lanes 32 - 63
__device__ __forceinline__ float pair_xor32(float x, float y)
{
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}
__device__ __forceinline__ float add_xor16(float v)
{
    float a, b;
    swap16(v, a, b);
    return a + b;
}
__device__ __forceinline__ float add_xor32(float v)
{
    float a, b;
    swap32(v, a, b);
    return a + b;
}
__device__ __forceinline__ float max_xor8(float v)
{
    return fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false)));
}
__device__ __forceinline__ float max_xor16(float v)
{
    float a, b;
    swap16(v, a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float max_xor32(float v)
{
    float a, b;
    swap32(v, a, b);
    return fmaxf(a, b);
}
// over the lane groups of a wave (lanes with the same lane % LPR), LPR = 8 or 16
template <int LPR>
__device__ __forceinline__ float groups_sum(float v)
{
    if constexpr (LPR == 8)
        v = add_xor8(v);
    return add_xor32(add_xor16(v));
}
template <int LPR>
__device__ __forceinline__ float groups_max(float v)
{
    if constexpr (LPR == 8)
        v = max_xor8(v);
    return max_xor32(max_xor16(v));
}


// WOQ (r05): weight-only int8 projection weights (u8 = q + 128, fp16 per-channel scales) against the NORMALISED fp16 row - no
// quantiser in the prologue, raw byte splices in the dots (1024 + u), 1152 * sum(x) taken off once per row: the arithmetic and
// the summation order of gemv_impl.h's W_INT8_WOQ path, so the projection is bit-identical to the unfused GEMV.
// WK = 2 (r06): fp16 projection weights (BASELINE.json configs[1]) - rows of 8 KB = TWO 8 KB tiles per row pair, the same two-tile
// ring (q.lo q.hi k.lo k.hi v.lo v.hi alternate through the two register buffers), the normalised fp16 row in LDS, v_dot2_f32_f16 in
// the per-lane chunk order of gemv_impl.h's W_FP16 path, no scales: bit-identical projection.  Two-stage form (no O-projection stage:
// 19 rows of 8 KB do not fit the row worker's LDS).
// WK = 3 (r06): weight-only int4 (nibble = q + 8 in the even / odd word order of weight_layout.h) - rows of 2 KB = HALF a tile per row
// pair (two 1 KiB chunks), the half-raw splices of gemv_impl.h's W_INT4_WOQ path (two running sums per row, 72 * sum(x_b) off once per
// row): bit-identical projection.  Two-stage form.
constexpr int WK_SQ = 0, WK_WOQ8 = 1, WK_FP16 = 2, WK_WOQ4 = 3;
template <int NIT, bool INT8KV, int WK = WK_SQ>
__global__ __launch_bounds__(512) void qkv_attn_fused_kernel(const FusedQkvAttnParams p)
{
    constexpr bool WOQ = WK == WK_WOQ8, F16W = WK == WK_FP16, WOQ4 = WK == WK_WOQ4;
    constexpr bool HALFX = WK != WK_SQ; // the projection's operand row stays fp16 (no quantiser)
    constexpr int KH = F16W ? 2 : 1;    // tiles per row pair
    constexpr int KCT = WOQ4 ? 2 : kKChunks; // 1 KiB chunks per tile and row (int4: a row is 2 KiB)
    constexpr bool WOQX = WOQ || WOQ4;       // weight-only: fp16 scales, the context row travels to the row workers as fp16
    constexpr int NT = 3 * KH;          // tiles per wave: q, k, v
    constexpr int EPL = INT8KV ? 16 : 8; // cache elements per lane (16 bytes)
    constexpr int LPR = kDH / EPL;       // lanes per cache row
    constexpr int RPW = 64 / LPR;        // rows per wave instruction = lane groups per wave
    constexpr int NGRP = kWavesF * RPW;  // lane groups per workgroup
    constexpr int TCHUNK = NGRP * NIT;   // cache slots per member
    constexpr int ESZ = INT8KV ? 1 : 2;
    constexpr int NQW = EPL / 2;         // 32-bit words of q per lane

    constexpr int XSB = HALFX ? 8192 : 4096; // the projection's operand row: fp16 (weight-only / fp16 weights) or int8
    __shared__ __attribute__((aligned(16))) char smem[XSB /* x */ + 256 /* red */ + 3 * 256 /* q', k', v as fp16 */
        + 3 * 256 /* raw q, k, v */ + kWavesF * (kDH + 8) * 4 /* wave partials */ + kMembers * (kDH + 8) * 4 /* head partials */ + 64];
    char* xs = smem;
    float* red = reinterpret_cast<float*>(smem + XSB);
    uint32_t* rot = reinterpret_cast<uint32_t*>(smem + XSB + 256);         // [3][64]: q', k' (RoPE applied), v
    uint32_t* raw = rot + 3 * 64;                                          // [3][64]: q, k, v as projected
    float* wpart = reinterpret_cast<float*>(raw + 3 * 64);                 // [8][136]: o[128], m, l per wave
    float* hpart = wpart + kWavesF * (kDH + 8);                            // [8][136]: the head's partials (member 0)
    float* misc = hpart + kMembers * (kDH + 8);                            // [0] score of the current token, [1] give-up flag
    uint32_t* qflag = reinterpret_cast<uint32_t*>(misc + 2);               // set by wave 0 once q' is in `rot`

    const int H = p.num_heads;
    const int h = blockIdx.x % H, mem = blockIdx.x / H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid) >> 6; // scalar: the role branches below are s_cbranch, not exec masks
    const int K = p.K;
    const uint32_t ep = p.epoch[0];
    // step tags live in the lower 31 bits (never 0: the exchange buffer starts zeroed), host tags carry the top bit: the two
    // numberings cannot meet, and a wrap of the 32-bit product after ~2^32 / (layers + 1) steps lands on a valid tag again
    const uint32_t ep_tag = (ep * p.tag_mul + p.tag_add) & 0x7fffffffu;
    const uint32_t tag = p.tag_host ? (p.tag_host | 0x80000000u) : (ep_tag ? ep_tag : 0x7fffffffu);
    // a bounded wait of an EARLIER launch expired: the session has not looked yet (it reads the word at its next synchronisation),
    // everything behind that launch is invalid anyway - do not spin again, layer after layer, step after step
    if (p.error[0] != 0u) // uniform (scalar load)
        return;
    const int tl = p.sequence_length[0]; // slots in use; the current token goes to slot tl
    const int Smax = p.max_seq_len;
    const bool q_dyn = p.act_quant_scale == nullptr;
    gu64* gx = (gu64*) p.xchg + (size_t) h * kHeadGranules;
    gu64* gp = gx + 3 * 64;
    gu64* gc = (gu64*) p.xchg + (size_t) H * kHeadGranules; // [H][kCtxGranules]: the context rows (O-projection stage)
    // O-projection stage: members 1 .. 7 of every head are the row workers - worker j takes rows [j n / W, (j + 1) n / W) of the dense
    // projection, wave `wid` rows r0 + wid, + 8, + 16.  Their per-channel scales and residual elements are launch constants too
    extern __shared__ __attribute__((aligned(16))) char wo_lds[]; // [rows of this worker][K bytes], filled by LDS-DMA
    const bool o_stage = p.o_w != nullptr; // uniform
    const int o_workers = (kMembers - 1) * H;
    const int o_j = (mem - 1) * H + h;
    const int o_r0 = o_stage && mem ? (int) ((int64_t) o_j * p.o_n / o_workers) : 0;
    const int o_r1 = o_stage && mem ? (int) ((int64_t) (o_j + 1) * p.o_n / o_workers) : 0;
    float o_cs[3], o_res[3];
    float o_rs = 1.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
    {
        const int row = o_r0 + wid + 8 * i;
        const bool on = row < o_r1; // wave-uniform
        if constexpr (WOQX)
            o_cs[i] = on ? h2f(reinterpret_cast<const uint16_t*>(p.o_scale_col)[row]) : 0.f;
        else
            o_cs[i] = on ? reinterpret_cast<const float*>(p.o_scale_col)[p.o_per_channel ? row : 0] : 0.f;
        o_res[i] = on ? h2f(reinterpret_cast<const uint16_t*>(p.x)[row]) : 0.f;
    }
    if (o_stage && WK == WK_SQ)
        o_rs = p.o_scale_row[0];
    // launch constants, requested before anything else (and before the kernel's first store: behind one hipcc no longer uses the
    // scalar path for them, and as vector loads behind the q rows they held the prologue until the q rows had arrived)
    float pro_q = 1.f, deq = 1.f;
    if (!q_dyn) // uniform
    {
        pro_q = p.act_quant_scale[0];
        deq = p.act_dequant_scale[0];
    }
    float s_oq = 1.f, s_qo = 1.f;
    if constexpr (INT8KV)
    {
        s_oq = p.kv_scale_orig_quant[0];
        s_qo = p.kv_scale_quant_orig[0];
    }

    // ------------------------------------------------------------------ t = 0: x, gamma, the scales and the q rows
    // (a) x: vectors t and t + 256 of the 256-thread prologue this one restates (threads >= 256 repeat the first half's work so
    //     that no load sits behind a branch); gamma for the vector this thread normalises
    const int t2 = tid & 255;
    const uint16_t* xg = reinterpret_cast<const uint16_t*>(p.x);
    const uint4 xa = *reinterpret_cast<const uint4*>(xg + t2 * 8);
    const uint4 xb = *reinterpret_cast<const uint4*>(xg + (t2 + 256) * 8);
    const uint4 gv = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(p.gamma) + tid * 8);
    // the per-channel scales of the wave's 6 rows (wave-uniform addresses: scalar loads)
    int wrow[3];
    float cscale[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
    {
        wrow[i] = (i * H + h) * kDH + mem * 16 + 2 * wid;
#pragma unroll
        for (int r = 0; r < 2; ++r)
        {
            if constexpr (WOQ || WOQ4)
                cscale[i][r] = h2f(reinterpret_cast<const uint16_t*>(p.scale_col)[wrow[i] + r]);
            else if constexpr (F16W)
                cscale[i][r] = 1.f;
            else
                cscale[i][r] = reinterpret_cast<const float*>(p.scale_col)[p.per_channel ? wrow[i] + r : 0];
        }
    }
    const float4 cs = reinterpret_cast<const float4*>(p.rope_row)[lane & 31]; // RoPE coefficients of elements 2 l', 2 l' + 1
    __builtin_amdgcn_sched_barrier(0); // issue order = consumption order (hipcc sank the gamma load below the weight loads)
    // (b) the wave's q rows: rows 2 wid, 2 wid + 1 of the member's 16.  The k rows follow behind the prologue, the v rows into
    //     the q rows' registers behind their dots: two 8 KB tiles per wave = 128 KB per CU in flight.  NOT the whole stream at
    //     t = 0 (r05 first form, 230 KB per CU): a CU drains its load queue in order at ~25 GB/s, and what has to pass through it
    //     later - the instruction fetch of the code below, x itself - waited: the prologue finished 6.6 us after the launch
    //     instead of 3.4 (tools/fused_timeline.py).
    const char* wbase = reinterpret_cast<const char*>(p.w);
    uint4 wa[kKChunks][2], wb[kKChunks][2];
    // tile t of the wave's stream: matrix t / KH (q, k, v), chunks [4 (t % KH), + 4) of its two rows
    auto load_tile = [&](int t, uint4 (&wt)[kKChunks][2]) {
#pragma unroll
        for (int u = 0; u < KCT; ++u)
#pragma unroll
            for (int r = 0; r < 2; ++r)
                wt[u][r] = ld_nt16(wbase + (int64_t) (wrow[t / KH] + r) * p.ldw + ((t % KH) * kKChunks + u) * 1024 + lane * 16);
    };
    load_tile(0, wa);
    __builtin_amdgcn_sched_barrier(0);
    TLLM_STAMP(0);

    // ------------------------------------------------------------------ 1. RMSNorm + quantiser -> LDS (gemv_impl.h PK_NORM, MB = 1)
    {
        float ss = 0.f;
        const uint32_t w8[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
        for (int q = 0; q < 8; ++q)
        {
            const h2_t hh = u32_as_h2(w8[q]);
            const float f0 = (float) hh.x, f1 = (float) hh.y;
            ss += f0 * f0 + f1 * f1;
        }
        ss = wave_sum(ss);
        if (lane == 0 && wid < 4)
            red[wid] = ss;
        if (tid == 0)
        {
            misc[1] = 0.f;
            *qflag = 0u;
        }
        __syncthreads();
        ss = red[0] + red[1] + red[2] + red[3];
        const float inv = 1.0f / sqrtf(ss / (float) K + p.eps);
        // this thread's own vector: tid * 8 (threads < 256: vector a, the others: vector b)
        uint32_t xs4[4] = {tid < 256 ? xa.x : xb.x, tid < 256 ? xa.y : xb.y, tid < 256 ? xa.z : xb.z, tid < 256 ? xa.w : xb.w};
        const uint32_t gs4[4] = {gv.x, gv.y, gv.z, gv.w};
        float amax = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
        {
            h2_t hh = u32_as_h2(xs4[q]);
            const h2_t gg = u32_as_h2(gs4[q]);
            const float n0 = h2f(f2h((float) hh.x * inv)), n1 = h2f(f2h((float) hh.y * inv));
            hh.x = (_Float16) (n0 * (float) gg.x);
            hh.y = (_Float16) (n1 * (float) gg.y);
            xs4[q] = h2_as_u32(hh);
            amax = fmaxf(amax, fmaxf(fabsf((float) hh.x), fabsf((float) hh.y)));
        }
        if constexpr (WOQ || WOQ4)
        {
            // the splice bias 1152 * sum of the normalised row (int4: 72 * the sum over the halves that face the 1024 + 16 n splices) (fp16 values, fp32 sums) in the unfused prologue's order: thread
            // t < 256 of that kernel owns vectors t and t + 256 - both are in this thread's registers (t2 = tid & 255), so waves
            // 0 - 3 restate its sums and waves 4 - 7 repeat them
            const uint4 ga = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(p.gamma) + t2 * 8);
            const uint4 gb = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(p.gamma) + (t2 + 256) * 8);
            const uint32_t xv[2][4] = {{xa.x, xa.y, xa.z, xa.w}, {xb.x, xb.y, xb.z, xb.w}};
            const uint32_t gg[2][4] = {{ga.x, ga.y, ga.z, ga.w}, {gb.x, gb.y, gb.z, gb.w}};
            float sa = 0.f, sb = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                {
                    const h2_t hh = u32_as_h2(xv[j][q]);
                    const h2_t g2 = u32_as_h2(gg[j][q]);
                    const float n0 = h2f(f2h((float) hh.x * inv)), n1 = h2f(f2h((float) hh.y * inv));
                    const float y0 = (float) (_Float16) (n0 * (float) g2.x), y1 = (float) (_Float16) (n1 * (float) g2.y);
                    if (q & 1)
                        sb += y0 + y1;
                    else
                        sa += y0 + y1;
                }
            const float bsum = wave_sum(WOQ4 ? 72.f * sb : 1152.f * (sa + sb));
            if (lane == 0 && wid < 4)
                red[8 + wid] = bsum;
        }
        if constexpr (HALFX)
            *reinterpret_cast<uint4*>(xs + tid * 16) = make_uint4(xs4[0], xs4[1], xs4[2], xs4[3]);
        float qs = pro_q;
        if (!HALFX && q_dyn) // uniform: per-token scale amax / 127 (K/quantization.cu:94-118)
        {
            amax = wave_max(amax);
            if (lane == 0)
                red[32 + wid] = amax;
            __syncthreads();
            amax = red[32];
#pragma unroll
            for (int w = 1; w < kWavesF; ++w)
                amax = fmaxf(amax, red[32 + w]);
            amax = fmaxf(amax, h2f(f2h(1e-6f)));
            qs = 127.f / amax;
            deq = amax / 127.f;
        }
        uint32_t o[2] = {0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q)
        {
            const h2_t hh = u32_as_h2(xs4[q]);
            const uint32_t b0 = (uint8_t) f2i8_rni_sat((float) hh.x * qs);
            const uint32_t b1 = (uint8_t) f2i8_rni_sat((float) hh.y * qs);
            o[q >> 1] |= (b0 | (b1 << 8)) << (16 * (q & 1));
        }
        if constexpr (!HALFX)
            *reinterpret_cast<uint2*>(xs + tid * 8) = make_uint2(o[0], o[1]);
    }
    __syncthreads();
    float xbias = 0.f;
    if constexpr (WOQ || WOQ4)
        xbias = red[8] + red[9] + red[10] + red[11];
    TLLM_STAMP(1);
    // (b2) the k rows, (c) the member's cache rows and masks
    __builtin_amdgcn_sched_barrier(0);
    load_tile(1, wb);
    __builtin_amdgcn_sched_barrier(0);
    const int li = lane % LPR, grp = lane / LPR, gid = wid * RPW + grp;
    const int t0 = mem * TCHUNK;
    char* kbase = reinterpret_cast<char*>(p.kv_cache) + ((int64_t) 0 * H + h) * Smax * kDH * ESZ;
    char* vbase = reinterpret_cast<char*>(p.kv_cache) + ((int64_t) 1 * H + h) * Smax * kDH * ESZ;
    uint4 kreg[NIT], vreg[NIT];
    int mk[NIT];
    const bool has_mask = p.masked_tokens != nullptr;
    const int32_t* mask_ptr = has_mask ? p.masked_tokens : p.sequence_length;
#pragma unroll
    for (int i = 0; i < NIT; ++i)
    {
        const int t = min(t0 + i * NGRP + gid, Smax - 1); // clamped, dropped by the validity mask: no branch around a load
        const int64_t off = ((int64_t) t * kDH + li * EPL) * ESZ;
        kreg[i] = *reinterpret_cast<const uint4*>(kbase + off);
        vreg[i] = *reinterpret_cast<const uint4*>(vbase + off);
        // unconditional (a load inside a conditional block ends in a full s_waitcnt - here: behind the whole weight stream):
        // without a mask the word read is sequence_length[0], valid and ignored
        mk[i] = mask_ptr[has_mask ? t : 0];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (blockIdx.x == 0 && p.x_pro_out) // tap: the int8 operand exactly as the projection consumes it
        for (int k = tid; k < XSB / 4; k += 512)
            reinterpret_cast<uint32_t*>(p.x_pro_out)[k] = reinterpret_cast<const uint32_t*>(xs)[k];

    // ------------------------------------------------------------------ 2. the projections, one granule per wave and matrix
    auto project = [&](const uint4 (&wt)[kKChunks][2], int i) {
        if constexpr (WOQ4)
        {
            // lane l, chunk u: 32 nibbles k = (u * 64 + l) * 32 .. + 32 against the 32 halfs of x at the same k (four LDS vectors)
            float fa0 = 0.f, fb0 = 0.f, fa1 = 0.f, fb1 = 0.f;
#pragma unroll
            for (int u = 0; u < KCT; ++u)
            {
                const char* xr = xs + (u * 64 + lane) * 64;
                const uint4 x0 = *reinterpret_cast<const uint4*>(xr), x1 = *reinterpret_cast<const uint4*>(xr + 16);
                const uint4 x2 = *reinterpret_cast<const uint4*>(xr + 32), x3 = *reinterpret_cast<const uint4*>(xr + 48);
                dot_u4x8_raw(wt[u][0].x, x0, fa0, fb0);
                dot_u4x8_raw(wt[u][0].y, x1, fa0, fb0);
                dot_u4x8_raw(wt[u][0].z, x2, fa0, fb0);
                dot_u4x8_raw(wt[u][0].w, x3, fa0, fb0);
                dot_u4x8_raw(wt[u][1].x, x0, fa1, fb1);
                dot_u4x8_raw(wt[u][1].y, x1, fa1, fb1);
                dot_u4x8_raw(wt[u][1].z, x2, fa1, fb1);
                dot_u4x8_raw(wt[u][1].w, x3, fa1, fb1);
            }
            // gemv_impl.h: tot = sum(a) + sum(b) / 16, then the bias, then fp16(. * scale)
            float t0 = wave_sum(fa0), t1 = wave_sum(fa1);
            t0 += wave_sum(fb0) * 0.0625f;
            t1 += wave_sum(fb1) * 0.0625f;
            t0 -= xbias;
            t1 -= xbias;
            const uint32_t v = (uint32_t) f2h(t0 * (cscale[i][0] * 1.f)) | ((uint32_t) f2h(t1 * (cscale[i][1] * 1.f)) << 16);
            if (lane == 0)
                st_granule(gx + i * 64 + mem * 8 + wid, tag, v);
            return;
        }
        if constexpr (WOQ)
        {
            // lane l, chunk u: weights k = (u * 64 + l) * 16 .. + 16 against the 16 halfs of x at the same k (two LDS vectors)
            float f0 = 0.f, f1 = 0.f;
#pragma unroll
            for (int u = 0; u < kKChunks; ++u)
            {
                const uint4 xlo = *reinterpret_cast<const uint4*>(xs + (u * 64 + lane) * 32);
                const uint4 xhi = *reinterpret_cast<const uint4*>(xs + (u * 64 + lane) * 32 + 16);
                f0 = dot_woq8_raw(wt[u][0], xlo, xhi, f0);
                f1 = dot_woq8_raw(wt[u][1], xlo, xhi, f1);
            }
            f0 = wave_sum(f0) - xbias;
            f1 = wave_sum(f1) - xbias;
            // epilogue of the weight-only GEMV: fp16((sum - bias) * scale)
            const uint32_t v = (uint32_t) f2h(f0 * (cscale[i][0] * 1.f)) | ((uint32_t) f2h(f1 * (cscale[i][1] * 1.f)) << 16);
            if (lane == 0)
                st_granule(gx + i * 64 + mem * 8 + wid, tag, v);
            return;
        }
        int a0 = 0, a1 = 0;
#pragma unroll
        for (int u = 0; u < kKChunks; ++u)
        {
            const uint4 xr = *reinterpret_cast<const uint4*>(xs + (u * 64 + lane) * 16);
            a0 = sdot4(wt[u][0].x, xr.x, a0);
            a0 = sdot4(wt[u][0].y, xr.y, a0);
            a0 = sdot4(wt[u][0].z, xr.z, a0);
            a0 = sdot4(wt[u][0].w, xr.w, a0);
            a1 = sdot4(wt[u][1].x, xr.x, a1);
            a1 = sdot4(wt[u][1].y, xr.y, a1);
            a1 = sdot4(wt[u][1].z, xr.z, a1);
            a1 = sdot4(wt[u][1].w, xr.w, a1);
        }
        a0 = wave_sum(a0);
        a1 = wave_sum(a1);
        // epilogue of the SmoothQuant GEMM: fp16(float(acc) * (scale_col * scale_row))
        const uint32_t v = (uint32_t) f2h((float) a0 * (cscale[i][0] * deq)) | ((uint32_t) f2h((float) a1 * (cscale[i][1] * deq)) << 16);
        if (lane == 0)
            st_granule(gx + i * 64 + mem * 8 + wid, tag, v);
    };
    // fp16 weights: half a row pair per tile - the per-lane sums run on across the two tiles of a matrix (chunk order 0 .. 7, the
    // order of gemv_impl.h's two-tile row groups), one cross-lane sum and one granule per matrix
    float h0 = 0.f, h1 = 0.f;
    auto accum16 = [&](const uint4 (&wt)[kKChunks][2], int half) {
#pragma unroll
        for (int u = 0; u < kKChunks; ++u)
        {
            const uint4 xr = *reinterpret_cast<const uint4*>(xs + ((half * kKChunks + u) * 64 + lane) * 16);
            h0 = dot2(wt[u][0].x, xr.x, h0);
            h0 = dot2(wt[u][0].y, xr.y, h0);
            h0 = dot2(wt[u][0].z, xr.z, h0);
            h0 = dot2(wt[u][0].w, xr.w, h0);
            h1 = dot2(wt[u][1].x, xr.x, h1);
            h1 = dot2(wt[u][1].y, xr.y, h1);
            h1 = dot2(wt[u][1].z, xr.z, h1);
            h1 = dot2(wt[u][1].w, xr.w, h1);
        }
    };
    auto finish16 = [&](int i) {
        // epilogue of the fp16 GEMV: fp16(sum) (gemv_impl.h: v0 * (1 * 1))
        const uint32_t v = (uint32_t) f2h(wave_sum(h0) * (cscale[i][0] * 1.f)) | ((uint32_t) f2h(wave_sum(h1) * (cscale[i][1] * 1.f)) << 16);
        if (lane == 0)
            st_granule(gx + i * 64 + mem * 8 + wid, tag, v);
        h0 = h1 = 0.f;
    };
    // The first look at the head's q (and, member 0, k) granules is requested behind the LAST tile in this wave's in-order
    // load queue: the siblings published them microseconds ago, the answer comes back with the v rows - when it is needed - and
    // costs nothing.  (Polling for them while the weights stream does: a poll takes ~2.2 us under load and slows the stream,
    // measured with a ninth "gather" wave: the q hand-off took 4.8 - 6.3 us and the launch 17 - 20 us.)
    // (Every wave sweeping for itself - no workgroup barrier behind the sweep - was measured: 64 waves per head polling the same
    // four lines with write-through loads made every poll slower, the launch 16.9 -> 18.7 us.  One sweeping wave per workgroup,
    // the others wait on an LDS flag.)
    unsigned long long gq = 0, gk = 0;
    auto first_look = [&]() {
        if (wid == 0)
        {
            // (not at once: the load EXECUTES soon after it is issued - only its return is ordered behind the v rows - and the slowest
            // sibling's q granule is ~1 us behind this wave's k rows; the v rows are ~2 us away, so the pause costs nothing.  Without
            // it the first look missed often and the second took 1.8 us)
            __builtin_amdgcn_s_sleep(64);
            gq = ld_granule(gx + lane);
            gk = ld_granule(gx + 64 + lane);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (!F16W)
    {
        project(wa, 0);
        TLLM_STAMP(2);
        __builtin_amdgcn_sched_barrier(0);
        load_tile(2, wa); // the v rows, into the registers the q rows have left
        __builtin_amdgcn_sched_barrier(0);
        project(wb, 1);
        TLLM_STAMP(3);
        first_look();
        // -------------------------------------------------------------- 3. the v rows: the end of the weight stream
        project(wa, 2);
        TLLM_STAMP(6);
    }
    else
    {
        // tiles: q.lo (wa, in flight since t = 0), q.hi (wb), k.lo, k.hi, v.lo, v.hi - each into the buffer the tile two back has left
        accum16(wa, 0);
        __builtin_amdgcn_sched_barrier(0);
        load_tile(2, wa);
        __builtin_amdgcn_sched_barrier(0);
        accum16(wb, 1);
        finish16(0);
        TLLM_STAMP(2);
        __builtin_amdgcn_sched_barrier(0);
        load_tile(3, wb);
        __builtin_amdgcn_sched_barrier(0);
        accum16(wa, 0);
        __builtin_amdgcn_sched_barrier(0);
        load_tile(4, wa);
        __builtin_amdgcn_sched_barrier(0);
        accum16(wb, 1);
        finish16(1);
        TLLM_STAMP(3);
        __builtin_amdgcn_sched_barrier(0);
        load_tile(5, wb);
        __builtin_amdgcn_sched_barrier(0);
        first_look();
        accum16(wa, 0);
        accum16(wb, 1);
        finish16(2);
        TLLM_STAMP(6);
    }

    // ------------------------------------------------------------------ 4. q' of the whole head -> LDS (wave 0; RoPE in the sweep).
    //          No workgroup barrier: the other waves wait on an LDS flag, each from the moment ITS v rows are done (the eight waves'
    //          v rows arrive up to ~2 us apart; a barrier would hold all of them for the last)
    uint32_t qrot = 0;
    if (wid == 0)
    {
        int spins = 0;
        while (!__all((uint32_t) (gq >> 32) == tag))
        {
            if (++spins > p.max_spins) // a sibling never published (a workgroup that is not resident: see the launcher's residency rule)
            {
                if (lane == 0)
                    misc[1] = 1.f;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            gq = ld_granule(gx + lane);
        }
        if (p.timing && lane == 0)
            p.timing[(size_t) blockIdx.x * 16 + 12] = (uint64_t) spins;
        const uint32_t q2 = (uint32_t) gq;
        qrot = rope_pair(q2, cs, lane >= 32);
        raw[lane] = q2;
        rot[lane] = qrot;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0)
            __hip_atomic_store(qflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    else
    {
        while (__hip_atomic_load(qflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u)
            __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    TLLM_STAMP(4);

    // ------------------------------------------------------------------ 5. scores, softmax partial and P.V of this member's rows
    uint32_t q16[NQW];
#pragma unroll
    for (int j = 0; j < NQW; ++j)
        q16[j] = rot[li * NQW + j];
    const float kscale = INT8KV ? s_qo * p.inv_sqrt_dh : p.inv_sqrt_dh;
    // int8 cache: the bytes are spliced into fp16 1024 + u (u = value + 128) and enter the dot product as they are; 1152 x (the sum
    // of this lane's q) comes off once per row instead of 1152 off every element (gemv_impl.h dot_u8x4_raw: the same 16 : 1 ratio
    // of bias to signal, fp32 sums)
    float qsum1152 = 0.f;
    if constexpr (INT8KV)
    {
        const h2_t ones = {(_Float16) 1.f, (_Float16) 1.f};
#pragma unroll
        for (int j = 0; j < NQW; ++j)
            qsum1152 = __builtin_amdgcn_fdot2(u32_as_h2(q16[j]), ones, qsum1152, false);
        qsum1152 *= 1152.f;
    }
    float s[NIT];
    float m_g = -INFINITY;
#pragma unroll
    for (int i = 0; i < NIT; ++i)
    {
        const int t = t0 + i * NGRP + gid;
        float d = 0.f;
        if constexpr (INT8KV)
        {
            const uint32_t magic = 0x64646464u;
            const uint32_t kw[4] = {kreg[i].x ^ 0x80808080u, kreg[i].y ^ 0x80808080u, kreg[i].z ^ 0x80808080u, kreg[i].w ^ 0x80808080u};
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                d = __builtin_amdgcn_fdot2(u32_as_h2(__builtin_amdgcn_perm(magic, kw[j], 0x04010400u)), u32_as_h2(q16[2 * j]), d, false);
                d = __builtin_amdgcn_fdot2(u32_as_h2(__builtin_amdgcn_perm(magic, kw[j], 0x04030402u)), u32_as_h2(q16[2 * j + 1]), d, false);
            }
            d -= qsum1152;
        }
        else
        {
            d = dot2(q16[0], kreg[i].x, d);
            d = dot2(q16[1], kreg[i].y, d);
            d = dot2(q16[2], kreg[i].z, d);
            d = dot2(q16[3], kreg[i].w, d);
        }
        d = group_sum<LPR>(d) * kscale;
        const bool valid = t < tl && t < Smax && (mk[i] == 0 || !has_mask); // the current token (slot tl) is member 0's ninth partial
        s[i] = valid ? d : -INFINITY;
        m_g = fmaxf(m_g, s[i]);
    }
    float l_g = 0.f, l16 = 0.f;
    float o[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j)
        o[j] = 0.f;
#pragma unroll
    for (int i = 0; i < NIT; ++i)
    {
        const float pr = (s[i] == -INFINITY) ? 0.f : __expf(s[i] - m_g);
        l_g += pr;
        const float p16 = h2f(f2h(pr));
        if constexpr (INT8KV)
        {
            const uint32_t vw[4] = {vreg[i].x ^ 0x80808080u, vreg[i].y ^ 0x80808080u, vreg[i].z ^ 0x80808080u, vreg[i].w ^ 0x80808080u};
            l16 += p16;
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    o[4 * w + j] = fmaf(p16, (float) ((vw[w] >> (8 * j)) & 0xffu), o[4 * w + j]);
        }
        else
        {
            const uint32_t vw[4] = {vreg[i].x, vreg[i].y, vreg[i].z, vreg[i].w};
#pragma unroll
            for (int w = 0; w < 4; ++w)
            {
                const h2_t hv = u32_as_h2(vw[w]);
                o[2 * w] = fmaf(p16, (float) hv.x, o[2 * w]);
                o[2 * w + 1] = fmaf(p16, (float) hv.y, o[2 * w + 1]);
            }
        }
    }
    // the lane groups of a wave (lanes li, li + LPR, ... hold the same elements of different rows): one weight per group, the
    // int8 scale and offset folded into it; the sums on the VALU cross-lane network, two values per swap: after the 16-lane and
    // the 32-lane level the 16-lane row r of the wave holds the totals of elements 4 m + r (m = 0 ..), which is where it stores them
    {
        const float m_w = groups_max<LPR>(m_g);
        const float e = (m_g == -INFINITY) ? 0.f : __expf(m_g - m_w);
        const float l_w = groups_sum<LPR>(l_g * e);
        const float es = INT8KV ? e * s_qo : e;
        const float off = INT8KV ? -128.f * l16 : 0.f; // sum p (u - 128) = sum p u - 128 sum p
#pragma unroll
        for (int j = 0; j < EPL; ++j)
        {
            o[j] = (o[j] + off) * es;
            if constexpr (LPR == 8)
                o[j] = add_xor8(o[j]);
        }
        float y[EPL / 4];
#pragma unroll
        for (int m = 0; m < EPL / 4; ++m)
            y[m] = pair_xor32(pair_xor16(o[4 * m], o[4 * m + 1]), pair_xor16(o[4 * m + 2], o[4 * m + 3]));
        float* wp = wpart + wid * (kDH + 8);
        const int row = lane >> 4;
        if (LPR == 16 || (lane & 8) == 0)
        {
#pragma unroll
            for (int m = 0; m < EPL / 4; ++m)
                wp[li * EPL + 4 * m + row] = y[m];
        }
        if (lane == 0)
        {
            wp[kDH] = m_w;
            wp[kDH + 1] = l_w;
        }
    }
    __syncthreads(); // D
    TLLM_STAMP(5);
    if (misc[1] != 0.f) // uniform
    {
        if (tid == 0)
            atomicOr(p.error, 1u);
        return;
    }
    // O-projection stage, row workers: waves 3 - 7 (no part in the publication below) request the worker's rows of the dense
    // projection by LDS-DMA NOW - the CU's load queue is empty, and the 3 - 4 us until the context rows arrive are what the
    // 16 MB of weights take (row slot s of the worker -> wave 3 + s % 5)
    if (o_stage && mem != 0 && wid >= 3)
    {
        const uint32_t wo_base = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) void*) wo_lds;
        const char* owb = reinterpret_cast<const char*>(p.o_w);
#pragma unroll
        for (int i = 0; i < (kORowsMax + 4) / 5; ++i)
        {
            const int slot = (wid - 3) + 5 * i;
            if (o_r0 + slot < o_r1) // wave-uniform
            {
#pragma unroll
                for (int u = 0; u < KCT; ++u)
                    glds16(owb + (int64_t) (o_r0 + slot) * p.o_ldw + u * 1024 + lane * 16, wo_base + (slot * KCT + u) * 1024);
            }
        }
    }
    // the eight waves -> the member's partial, published as tagged granules: o[d] by thread d, m by thread 128, l by thread 129
    if (tid < kDH + 2)
    {
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < kWavesF; ++w)
            M = fmaxf(M, wpart[w * (kDH + 8) + kDH]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int w = 0; w < kWavesF; ++w)
        {
            const float mw = wpart[w * (kDH + 8) + kDH];
            const float e = (mw == -INFINITY) ? 0.f : __expf(mw - M);
            L += wpart[w * (kDH + 8) + kDH + 1] * e;
            O += wpart[w * (kDH + 8) + (tid < kDH ? tid : 0)] * e;
        }
        const float val = tid < kDH ? O : (tid == kDH ? M : L);
        st_granule(gp + mem * kPartStride + tid, tag, __float_as_uint(val));
    }
    if (mem != 0)
    {
        if (!o_stage)
            return;
        // -------------------------------------------------------------- 7. row workers: x_out[n] = x[n] + O(ctx)[n]
        TLLM_STAMP(8);
        // every head's context row - int8, four elements per granule: 1024 granules, two per thread; weight-only: fp16, two per
        // granule: 2048, four per thread - behind the DMA in the queue
        {
            constexpr int NG = WOQX ? 4 : 2;
            const gu64* gbase = gc + tid;
            unsigned long long g[NG];
            int spins = 0;
            for (;;)
            {
#pragma unroll
                for (int i = 0; i < NG; ++i)
                    g[i] = ld_granule(gbase + 512 * i);
                __builtin_amdgcn_sched_barrier(0);
                uint32_t bad = 0;
#pragma unroll
                for (int i = 0; i < NG; ++i)
                    bad |= (uint32_t) (g[i] >> 32) ^ tag;
                if (__all(bad == 0))
                    break;
                if (++spins > p.max_spins)
                {
                    if (lane == 0)
                        misc[1] = 1.f;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int i = 0; i < NG; ++i)
                reinterpret_cast<uint32_t*>(xs)[tid + 512 * i] = (uint32_t) g[i];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the rows this wave requested are in LDS
        __syncthreads();                                  // ... everybody's are, and so is the context row
        TLLM_STAMP(9);
        if (misc[1] != 0.f) // uniform
        {
            if (tid == 0)
                atomicOr(p.error, 4u);
            return;
        }
        float obias = 0.f;
        if constexpr (WOQX)
        {
            // the splice bias 1152 * sum(ctx) in the unfused GEMV's order (thread t < 256 owns vectors t and t + 256 of the row)
            if (wid < 4)
            {
                const uint4 va = *reinterpret_cast<const uint4*>(xs + tid * 16);
                const uint4 vb = *reinterpret_cast<const uint4*>(xs + (tid + 256) * 16);
                const uint32_t xv[2][4] = {{va.x, va.y, va.z, va.w}, {vb.x, vb.y, vb.z, vb.w}};
                float sa = 0.f, sb = 0.f;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                    {
                        const h2_t hh = u32_as_h2(xv[j][q]);
                        if (q & 1)
                            sb += (float) hh.x + (float) hh.y;
                        else
                            sa += (float) hh.x + (float) hh.y;
                    }
                const float bsum = wave_sum(WOQ4 ? 72.f * sb : 1152.f * (sa + sb));
                if (lane == 0)
                    red[8 + wid] = bsum;
            }
            __syncthreads();
            obias = red[8] + red[9] + red[10] + red[11];
        }
        // rows r0 + wid, + 8, + 16 of the worker (any wave reads any row: the barrier above is behind every wave's vmcnt(0)); the
        // three dot products and their cross-lane sums side by side
        using oacc_t = typename std::conditional<WOQX, float, int>::type;
        oacc_t acc[3] = {0, 0, 0};
        float accb[3] = {0.f, 0.f, 0.f}; // int4: the running sums over the 1024 + 16 n splices
#pragma unroll
        for (int u = 0; u < KCT; ++u)
        {
            uint4 xr, xr2, xr3, xr4;
            if constexpr (WOQ4)
            {
                const char* xp = xs + (u * 64 + lane) * 64;
                xr = *reinterpret_cast<const uint4*>(xp);
                xr2 = *reinterpret_cast<const uint4*>(xp + 16);
                xr3 = *reinterpret_cast<const uint4*>(xp + 32);
                xr4 = *reinterpret_cast<const uint4*>(xp + 48);
            }
            else if constexpr (WOQ)
            {
                xr = *reinterpret_cast<const uint4*>(xs + (u * 64 + lane) * 32);
                xr2 = *reinterpret_cast<const uint4*>(xs + (u * 64 + lane) * 32 + 16);
            }
            else
                xr = *reinterpret_cast<const uint4*>(xs + (u * 64 + lane) * 16);
#pragma unroll
            for (int i = 0; i < 3; ++i)
            {
                const int slot = wid + 8 * i < o_r1 - o_r0 ? wid + 8 * i : 0; // (a row that does not exist: slot 0, result dropped)
                const uint4 wv = *reinterpret_cast<const uint4*>(wo_lds + (slot * KCT + u) * 1024 + lane * 16);
                if constexpr (WOQ4)
                {
                    dot_u4x8_raw(wv.x, xr, acc[i], accb[i]);
                    dot_u4x8_raw(wv.y, xr2, acc[i], accb[i]);
                    dot_u4x8_raw(wv.z, xr3, acc[i], accb[i]);
                    dot_u4x8_raw(wv.w, xr4, acc[i], accb[i]);
                }
                else if constexpr (WOQ)
                    acc[i] = dot_woq8_raw(wv, xr, xr2, acc[i]);
                else
                {
                    acc[i] = sdot4(wv.x, xr.x, acc[i]);
                    acc[i] = sdot4(wv.y, xr.y, acc[i]);
                    acc[i] = sdot4(wv.z, xr.z, acc[i]);
                    acc[i] = sdot4(wv.w, xr.w, acc[i]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
        {
            acc[i] = wave_sum(acc[i]);
            if constexpr (WOQ4)
                acc[i] += wave_sum(accb[i]) * 0.0625f; // (gemv_impl.h: tot = sum(a) + sum(b) / 16)
        }
        // epilogue of the unfused GEMV (gemv_impl.h, EPI_RESIDUAL): fp16(fp16(float(acc) * (scale_col * scale_row)) + residual)
        // (weight-only: (sum - bias) * scale, scale_row = 1)
        if (lane < 3)
        {
            const int i = lane;
            const int row = o_r0 + wid + 8 * i;
            const float a = (float) (i == 0 ? acc[0] : (i == 1 ? acc[1] : acc[2])) - obias;
            const float cs_i = i == 0 ? o_cs[0] : (i == 1 ? o_cs[1] : o_cs[2]);
            const float rs_i = i == 0 ? o_res[0] : (i == 1 ? o_res[1] : o_res[2]);
            if (row < o_r1)
                reinterpret_cast<uint16_t*>(p.x_out)[row] = f2h(h2f(f2h(a * (cs_i * o_rs))) + rs_i);
        }
        TLLM_STAMP(10);
        return;
    }

    // ------------------------------------------------------------------ 6. member 0: k, v, the eight partials -> context row
    if (wid == 0)
    {
        bool gave_up = false;
        int spins = 0;
        unsigned long long gvv = ld_granule(gx + 128 + lane);
        while (!__all((uint32_t) (gk >> 32) == tag && (uint32_t) (gvv >> 32) == tag))
        {
            if (++spins > p.max_spins)
            {
                gave_up = true;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            gk = ld_granule(gx + 64 + lane);
            gvv = ld_granule(gx + 128 + lane);
        }
        const uint32_t k2 = (uint32_t) gk, v2 = (uint32_t) gvv;
        const uint32_t krot = rope_pair(k2, cs, lane >= 32);
        raw[64 + lane] = k2;
        raw[128 + lane] = v2;
        rot[64 + lane] = krot;
        rot[128 + lane] = v2;
        // score of the current token: the un-quantised q'.k' (MM/...Template.h:1517-1549)
        const float dn = wave_sum(dot2(qrot, krot, 0.f)) * p.inv_sqrt_dh;
        if (lane == 0)
        {
            misc[0] = dn;
            if (gave_up)
                misc[1] = 1.f;
        }
    }
    else if (wid <= 3 && tid - 64 < kDH + 2) // waves 1 - 3: column c of the eight partials
    {
        const int c = tid - 64;
        unsigned long long g[kMembers];
        int spins = 0;
        for (;;)
        {
#pragma unroll
            for (int i = 0; i < kMembers; ++i)
                g[i] = ld_granule(gp + i * kPartStride + c);
            __builtin_amdgcn_sched_barrier(0);
            uint32_t bad = 0;
#pragma unroll
            for (int i = 0; i < kMembers; ++i)
                bad |= (uint32_t) (g[i] >> 32) ^ tag;
            if (__all(bad == 0))
                break;
            if (++spins > p.max_spins)
            {
                misc[1] = 1.f;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int i = 0; i < kMembers; ++i)
            hpart[i * (kDH + 8) + c] = __uint_as_float((uint32_t) g[i]);
    }
    __syncthreads(); // E
    TLLM_STAMP(11);
    if (misc[1] != 0.f)
    {
        if (tid == 0)
            atomicOr(p.error, 2u);
        return;
    }
    const bool has_new = tl < Smax;
    if (tid < kDH)
    {
        const int d = tid;
        const float s_new = misc[0];
        float Mx = has_new ? s_new : -INFINITY;
#pragma unroll
        for (int i = 0; i < kMembers; ++i)
            Mx = fmaxf(Mx, hpart[i * (kDH + 8) + kDH]);
        float Lt = 0.f, Ot = 0.f;
#pragma unroll
        for (int i = 0; i < kMembers; ++i)
        {
            const float mi = hpart[i * (kDH + 8) + kDH];
            const float e = (mi == -INFINITY) ? 0.f : __expf(mi - Mx);
            Lt += hpart[i * (kDH + 8) + kDH + 1] * e;
            Ot += hpart[i * (kDH + 8) + d] * e;
        }
        if (has_new)
        {
            const float e = __expf(s_new - Mx);
            const uint32_t vw = rot[128 + (d >> 1)];
            Lt += e; // p = exp(0) = 1, exactly representable in fp16
            Ot += e * h2f((uint16_t) ((d & 1) ? (vw >> 16) : (vw & 0xffffu)));
        }
        const uint16_t h16 = f2h(Ot * (1.f / (Lt + 1.e-6f)));
        const int64_t oi = (int64_t) h * kDH + d;
        reinterpret_cast<uint16_t*>(p.out)[oi] = h16;
        if (p.out_q8)
        {
            const int8_t q8 = f2i8_rni_sat(h2f(h16) * p.out_quant_scale[0]);
            reinterpret_cast<int8_t*>(p.out_q8)[oi] = q8;
            reinterpret_cast<int8_t*>(red)[d] = q8; // (the prologue's scratch: free since the first barrier)
        }
        if constexpr (WOQX)
            reinterpret_cast<uint16_t*>(red)[d] = h16; // weight-only: the row travels as fp16
    }
    else if (tid < kDH + 16)
    {
        // append the current token: 8 elements per thread, K row then V row of slot tl (bit-exact row A3)
        const int j = tid - kDH;
        if (has_new)
        {
            const uint4 k8 = *reinterpret_cast<const uint4*>(rot + 64 + j * 4);
            const uint4 v8 = *reinterpret_cast<const uint4*>(rot + 128 + j * 4);
            const int64_t off = ((int64_t) tl * kDH + j * 8) * ESZ;
            if constexpr (INT8KV)
            {
                const uint32_t kw[4] = {k8.x, k8.y, k8.z, k8.w}, vw[4] = {v8.x, v8.y, v8.z, v8.w};
                uint32_t ko[2] = {0, 0}, vo[2] = {0, 0};
#pragma unroll
                for (int e = 0; e < 8; ++e)
                {
                    const uint16_t kh = (uint16_t) ((e & 1) ? (kw[e >> 1] >> 16) : (kw[e >> 1] & 0xffffu));
                    const uint16_t vh = (uint16_t) ((e & 1) ? (vw[e >> 1] >> 16) : (vw[e >> 1] & 0xffffu));
                    ko[e >> 2] |= ((uint32_t) (uint8_t) f2i8_rni_sat(h2f(kh) * s_oq)) << (8 * (e & 3));
                    vo[e >> 2] |= ((uint32_t) (uint8_t) f2i8_rni_sat(h2f(vh) * s_oq)) << (8 * (e & 3));
                }
                *reinterpret_cast<uint2*>(kbase + off) = make_uint2(ko[0], ko[1]);
                *reinterpret_cast<uint2*>(vbase + off) = make_uint2(vo[0], vo[1]);
            }
            else
            {
                *reinterpret_cast<uint4*>(kbase + off) = k8;
                *reinterpret_cast<uint4*>(vbase + off) = v8;
            }
        }
    }
    else if (tid >= 192 && tid < 192 + 192 && p.qkv_out)
    {
        // the projection's output as the unfused launch leaves it (q | k | v, before RoPE): parity tests read it back
        const int c = tid - 192, i = c >> 6, g = c & 63;
        reinterpret_cast<uint32_t*>(p.qkv_out)[((int64_t) (i * H + h) * kDH) / 2 + g] = raw[c];
    }
    if (o_stage) // uniform: the context row to the row workers, four int8 per granule
    {
        __syncthreads(); // F
        constexpr int CG = WOQX ? kCtxGranulesH : kCtxGranules;
        if (tid < CG)
            st_granule(gc + h * CG + tid, tag, reinterpret_cast<const uint32_t*>(red)[tid]);
    }
    TLLM_STAMP(7);
}
#undef TLLM_STAMP

// the instance that serves (cache rows per lane group, cache type, weight type)
template <bool INT8KV, int WK>
const void* fused_kernel_of(int nit)
{
    switch (nit)
    {
    case 1: return reinterpret_cast<const void*>(qkv_attn_fused_kernel<1, INT8KV, WK>);
    case 2: return reinterpret_cast<const void*>(qkv_attn_fused_kernel<2, INT8KV, WK>);
    case 3: return reinterpret_cast<const void*>(qkv_attn_fused_kernel<3, INT8KV, WK>);
    case 4: return reinterpret_cast<const void*>(qkv_attn_fused_kernel<4, INT8KV, WK>);
    case 6: return reinterpret_cast<const void*>(qkv_attn_fused_kernel<6, INT8KV, WK>);
    case 8: return reinterpret_cast<const void*>(qkv_attn_fused_kernel<8, INT8KV, WK>);
    default: return nullptr;
    }
}

const void* fused_kernel(int nit, bool int8_kv, int wk)
{
    switch (wk)
    {
    case WK_WOQ8: return int8_kv ? fused_kernel_of<true, WK_WOQ8>(nit) : fused_kernel_of<false, WK_WOQ8>(nit);
    case WK_FP16: return int8_kv ? fused_kernel_of<true, WK_FP16>(nit) : fused_kernel_of<false, WK_FP16>(nit);
    case WK_WOQ4: return int8_kv ? fused_kernel_of<true, WK_WOQ4>(nit) : fused_kernel_of<false, WK_WOQ4>(nit);
    default: return int8_kv ? fused_kernel_of<true, WK_SQ>(nit) : fused_kernel_of<false, WK_SQ>(nit);
    }
}

// Per-DEVICE launch state (a process may hold sessions on devices with different CU counts): the CU count, which
// instances have had their dynamic-LDS attribute raised, and how many workgroups of an instance one CU admits.
struct FusedDevState
{
    int cus = 0;
    std::map<const void*, bool> attr_done;
    std::map<std::pair<const void*, size_t>, int> per_cu;
};
std::mutex fused_mu;
std::map<int, FusedDevState> fused_dev;

FusedDevState& dev_state_locked()
{
    int dev = 0;
    (void) hipGetDevice(&dev);
    FusedDevState& d = fused_dev[dev];
    if (!d.cus && (hipDeviceGetAttribute(&d.cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || d.cus <= 0))
        d.cus = 1;
    return d;
}

// workgroups of this instance the whole chip admits at once (the occupancy query x CUs; 0 when the query fails)
int resident_capacity(const void* kfn, size_t dyn)
{
    std::lock_guard<std::mutex> lock(fused_mu);
    FusedDevState& d = dev_state_locked();
    if (dyn && !d.attr_done[kfn])
    {
        (void) hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) dyn);
        d.attr_done[kfn] = true;
    }
    auto key = std::make_pair(kfn, dyn);
    auto it = d.per_cu.find(key);
    if (it == d.per_cu.end())
    {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kfn, 64 * kWavesF, dyn) != hipSuccess)
            nb = 0;
        it = d.per_cu.emplace(key, nb).first;
    }
    return it->second * d.cus;
}

int device_cus()
{
    std::lock_guard<std::mutex> lock(fused_mu);
    return dev_state_locked().cus;
}

size_t fused_dyn_lds(bool o_stage)
{
    // O-projection stage: the row worker's rows of the dense projection live in dynamic LDS (<= 24 rows x K bytes)
    return o_stage ? (size_t) kORowsMax * kKChunks * 1024 : 0;
}

int pick_nit(int max_seq_len, bool int8_kv)
{
    const int ngrp = kWavesF * (int8_kv ? 8 : 4);
    const int need = (max_seq_len + kMembers * ngrp - 1) / (kMembers * ngrp);
    for (int n : {1, 2, 3, 4, 6, 8})
        if (need <= n)
            return n;
    return 0;
}

} // namespace

size_t qkv_attn_fused_xchg_bytes(int32_t num_heads)
{
    return (size_t) num_heads * (kHeadGranules + kCtxGranulesH) * sizeof(uint64_t);
}

// the O-projection stage: K = H * Dh = 4 KiB rows (the context row is swept by 512 threads x two granules), every row worker's
// share of the rows fits its LDS area
bool qkv_attn_fused_serves_o(int32_t num_heads, int32_t head_size, int32_t o_n, int32_t o_k, int64_t o_ldw, int32_t weight_kind)
{
    const int workers = (kMembers - 1) * num_heads;
    const int64_t row_bytes = weight_kind == WK_WOQ4 ? o_k / 2 : o_k; // (int8 kinds; fp16 has no such stage)
    return head_size == kDH && o_k == num_heads * head_size && o_k == kKChunks * 1024 && num_heads * kCtxGranules == 1024
        && o_ldw % 16 == 0 && o_ldw >= row_bytes && o_n > 0 && (o_n + workers - 1) / workers <= kORowsMax && weight_kind != WK_FP16;
}

bool qkv_attn_fused_serves(int32_t K, int32_t num_heads, int32_t head_size, int32_t max_seq_len, int32_t int8_kv, int32_t weight_kind,
    int32_t o_stage)
{
    const int nit = pick_nit(max_seq_len, int8_kv != 0);
    // (K = 4096 elements: rows of 4 KiB int8 / 8 KiB fp16 = one / two tiles of 4 x 1 KiB chunks x 2 rows)
    if (K != kKChunks * 1024 || head_size != kDH || nit == 0 || weight_kind < WK_SQ || weight_kind > WK_WOQ4)
        return false;
    if (weight_kind == WK_FP16 && o_stage)
        return false; // the row worker's share of an fp16 dense projection does not fit its LDS
    const void* kfn = fused_kernel(nit, int8_kv != 0, weight_kind);
    if (!kfn)
        return false;
    // every workgroup of a head waits for its siblings (and the row workers for every head's merger): the whole grid must be
    // RESIDENT AT ONCE.  The occupancy query (this instance's registers, its dynamic LDS) x the CU count of THIS device must cover
    // the grid; and the launch only pays when it fills most of the chip.  (What the query cannot see - another queue's kernels
    // holding CUs - is what the bounded waits and the session's fall-back + retry are for.)
    const int cus = device_cus();
    const int grid = num_heads * kMembers;
    if (grid > cus || grid * 4 < cus * 3)
        return false;
    return resident_capacity(kfn, fused_dyn_lds(o_stage != 0)) >= grid;
}

int launch_qkv_attn_fused(const FusedQkvAttnParams& p, hipStream_t stream)
{
    const bool o_stage = p.o_w != nullptr;
    const int wk = p.fp16_w ? WK_FP16 : (p.woq4 ? WK_WOQ4 : (p.woq8 ? WK_WOQ8 : WK_SQ));
    if (!qkv_attn_fused_serves(p.K, p.num_heads, p.head_size, p.max_seq_len, p.int8_kv, wk, o_stage ? 1 : 0))
    {
        set_error("fused QKV + attention: shape not served or grid not resident (K %d, heads %d x %d, cache %d, O stage %d)", p.K,
            p.num_heads, p.head_size, p.max_seq_len, o_stage ? 1 : 0);
        return -1;
    }
    if (!p.x || !p.gamma || !p.w || (!p.scale_col && !p.fp16_w) || !p.kv_cache || !p.sequence_length || !p.rope_row || !p.xchg || !p.error || !p.out
        || !p.epoch || (p.act_quant_scale && !p.act_dequant_scale) || (p.out_q8 && !p.out_quant_scale)
        || (p.int8_kv && (!p.kv_scale_orig_quant || !p.kv_scale_quant_orig)) || p.ldw % 16)
    {
        set_error("fused QKV + attention: missing operand");
        return -1;
    }
    if (p.o_w
        && ((!p.woq8 && !p.woq4 && (!p.out_q8 || !p.o_scale_row)) || !p.o_scale_col || !p.x_out
            || !qkv_attn_fused_serves_o(p.num_heads, p.head_size, p.o_n, p.num_heads * p.head_size, p.o_ldw, wk)))
    {
        set_error("fused QKV + attention: the O-projection stage needs the static int8 context row and a dense projection of %d x %d",
            p.num_heads * p.head_size, p.num_heads * p.head_size);
        return -1;
    }
    if ((p.woq8 || p.fp16_w || p.woq4) && (p.out_q8 || p.act_quant_scale))
    {
        set_error("fused QKV + attention: the weight-only / fp16 forms have no quantiser");
        return -1;
    }
    if (p.fp16_w && (p.woq8 || p.woq4 || p.ldw != (int64_t) p.K * 2))
    {
        set_error("fused QKV + attention: fp16 weights are dense rows of K halfs");
        return -1;
    }
    if (p.woq4 && (p.woq8 || p.ldw != (int64_t) p.K / 2))
    {
        set_error("fused QKV + attention: int4 weights are dense rows of K / 2 bytes");
        return -1;
    }
    const void* kfn = fused_kernel(pick_nit(p.max_seq_len, p.int8_kv != 0), p.int8_kv != 0, wk);
    const dim3 grid(p.num_heads * kMembers), block(64 * kWavesF);
    FusedQkvAttnParams q = p;
    void* args[] = {&q};
    const hipError_t e = hipLaunchKernel(kfn, grid, block, args, fused_dyn_lds(o_stage), stream);
    if (e != hipSuccess)
    {
        set_error("fused QKV + attention launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}

} // namespace kernels
} // namespace tllm
