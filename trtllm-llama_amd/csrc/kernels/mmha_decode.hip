// Decode-step masked multi-head attention for gfx950.
//
// Reference: masked_multihead_attention_kernel, MM/decoderMaskedMultiheadAttentionTemplate.h:1195-2183
// (one CTA per (head, batch); 32 CTAs at B=1 cannot feed 256 CUs; its multi-block mode, :2019-2181, splits the
// KV range over gridDim.z CTAs and lets the last CTA to arrive — an atomic counter — rescale and reduce).
// Here the KV range of every (batch, head) is always split over workgroups of 4 waves (grid = splits x H x B;
// 256-token splits at the 7B decode shape = 160 workgroups at L ~ 1100).
// A group of Dh/8 lanes owns one cache row per load instruction (16 B per lane for fp16, 8 B for int8) and
// keeps NIT rows of K and of V in flight, all requested before anything else is computed and with NO branch
// around any load (out-of-range rows load a clamped address and are dropped by a select: a lane-dependent `if`
// makes hipcc fence every load with s_waitcnt vmcnt(0) + exec masking); scores with v_dot2_f32_f16 + a DPP sum
// inside the lane group; the int8 cache is widened with a byte splice to exact integers and the scale is folded
// into the dot product; an independent softmax partial {max, sum, out[Dh]} per lane group (no cross-group
// shuffles), merged per workgroup through LDS and published per split as (m, l) + un-normalised fp32 o.
// The splits of a (batch, head) are merged INSIDE this launch by the last split of the head to arrive (step 6: write-through
// partials, one ticket per workgroup, agent-scope loads; MmhaParams::tail_tickets) - plugin and session alike - which leaves the
// O-projection a plain fp16 / int8 vector; caches that need more than 16 splits run the finest split and mmha_combine_kernel.
// (r01 - r03 merged in the prologue of every O-projection workgroup: ~60 MB of L2 reads per launch for 116 KB of distinct data;
// removed in r05, profiles/r04_attn_tail_merge_ab.txt.)  (r01's first ticket merge used __threadfence() - an L2 write-back + invalidate per workgroup - and
// measured 23 us; a one-workgroup-per-head variant 22 - 35 us.  Two forms without the store drain - data-tagged granules polled by
// a designated split, and epoch-tagged granules behind an un-drained ticket - were built in r04, bit-identical, and not faster:
// profiles/r04_attn_granule_ab.txt.)  RoPE coefficients come from a 512-byte row the sampler
// prepared for this step (GreedyParams::rope_row_out), so nothing chases length -> position -> table.
//
// Numerics follow SURVEY Appendix A.1: RoPE in fp32 -> fp16; int8 cache store = sat(rni(float(k16) * s)),
// load = fp16(float(q8) * s^-1); q.k products fp32-accumulated, * inv_sqrt_dh in fp32; masked positions are
// dropped (weight 0, excluded from the max); probabilities rounded to fp16 before P.V; fp32 accumulation;
// normaliser (sum + 1e-6); one final rounding to fp16.  The split only reorders fp32 additions and rounds the
// un-normalised probability instead of the normalised one.
//
// Cache layout (bit-exact row A3, K/kvCacheUtils.h:114-170): [B, 2, H, Smax, Dh];
//   element (b, kv, h, t, d) at ((b*2 + kv)*H + h)*Smax*Dh + t*Dh + d.
#include "dev_utils.h"
#include "kernels.h"
#include <math.h>

namespace tllm
{
namespace kernels
{
using namespace dev;

namespace
{

constexpr int kWaves = 4; // 256-thread workgroups
constexpr int kTailSlots = 16; // partials the in-launch merge takes (MmhaParams::tail_tickets)

template <int DH, int NIT>
struct MmhaGeom
{
    static constexpr int LPR = DH / 8;        // lanes per cache row (8 elements per lane)
    static constexpr int RPW = 64 / LPR;      // rows per wave instruction = lane groups per wave
    static constexpr int NGRP = kWaves * RPW; // lane groups per workgroup
    static constexpr int TCHUNK = NGRP * NIT; // timesteps per workgroup (NIT rows of K and of V per lane group)
};

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

// 8 cached int8 -> 8 fp16: fp16(float(q) * s)   (…Utils.h:2358-2365)
__device__ __forceinline__ uint4 dequant8(const uint2& q, float s)
{
    const float nb = -128.f * s; // float(u) * s - 128 s == (u - 128) * s exactly (one rounding)
    const uint32_t a = q.x ^ 0x80808080u, b = q.y ^ 0x80808080u;
    float f[8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
    {
        f[j] = fmaf((float) ((a >> (8 * j)) & 0xffu), s, nb);
        f[4 + j] = fmaf((float) ((b >> (8 * j)) & 0xffu), s, nb);
    }
    return f_to_h8(f);
}

// 8 fp16 -> 8 int8: sat(rni(float(x16) * s))   (…Utils.h:2383-2390, 2276-2286)
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

// ---- kernel 1: one workgroup per (split, head, batch) -> partial {max, sum, out[DH]} in the workspace
// CACHE: how a (sequence, time step) finds its K/V row -
//   CACHE_LINEAR  [B, 2, H, Smax, DH], the sequence's own rows
//   CACHE_BEAM    the same buffer, rows of the sibling hypothesis cache_indirection names
//   CACHE_PAGED   block table [B, 2, max_blocks] of pointers to [H, tokens_per_block, DH] blocks (K/kvCacheUtils.h:34-112),
//                 through the cache indirection too when beam_width > 1
enum
{
    CACHE_LINEAR = 0,
    CACHE_BEAM = 1,
    CACHE_PAGED = 2
};

template <int DH, int NIT, bool INT8KV, int CACHE>
__global__ __launch_bounds__(256) void mmha_partial_kernel(const MmhaParams p, float2* ws_ml, float* ws_o, int nsplit_max)
{
    using G = MmhaGeom<DH, NIT>;
    constexpr int LPR = G::LPR, RPW = G::RPW, NGRP = G::NGRP;
    const int c = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane % LPR, grp = lane / LPR;
    const int gid = wid * RPW + grp; // lane group inside the workgroup
    const int H = p.num_heads, Smax = p.max_seq_len;
    const int t0 = c * G::TCHUNK;

    constexpr int ESZ = INT8KV ? 1 : 2;
    char* kbase = reinterpret_cast<char*>(p.kv_cache) + ((int64_t) (b * 2 + 0) * H + h) * Smax * DH * ESZ;
    char* vbase = reinterpret_cast<char*>(p.kv_cache) + ((int64_t) (b * 2 + 1) * H + h) * Smax * DH * ESZ;

    // ---- 1. request the cache rows: nothing below is needed to form their addresses (rows at or beyond the
    //         current length are loaded too - the buffer has Smax rows - and dropped by the validity mask)
    uint4 kreg[NIT], vreg[NIT];
    constexpr bool BEAM = CACHE == CACHE_BEAM, PAGED = CACHE == CACHE_PAGED;
    // beam search: timestep t of this hypothesis lives in the cache rows of a sibling (cache_indirection); the row
    // strides between siblings are whole (batch, 2, H, Smax, DH) sequences
    int64_t sib[NIT];
    // paged: the block holding time step t, from the sequence's (or the sibling's) row of the block table
    const char* kblk[NIT];
    const char* vblk[NIT];
    const int tpb_mask = p.tokens_per_block - 1;
    if constexpr (PAGED)
    {
        const int lg = 31 - __builtin_clz(p.tokens_per_block);
        const int k_own = p.beam_width > 1 ? b % p.beam_width : 0;
        const int32_t* ci = p.cache_indirection + (int64_t) b * Smax;
#pragma unroll
        for (int i = 0; i < NIT; ++i)
        {
            const int t = min(t0 + i * NGRP + gid, Smax - 1);
            int seq = b;
            if (p.beam_width > 1) // uniform
                seq = b - k_own + ci[t];
            const int64_t* row = p.block_pointers + (int64_t) seq * 2 * p.max_blocks_per_seq + (t >> lg);
            // a table entry of 0 = a logical block the cache manager has not handed out yet (KVCacheManager grows the table
            // on demand, PY/runtime/kv_cache_manager.py).  Such a block lies beyond the current length, so its rows are
            // dropped by the validity mask below - but the loads themselves are unconditional (no branch around a load),
            // so they are pointed at the pool instead of at address 0 + offset.  A select, not another dependent load.
            const int64_t kb = row[0], vb = row[p.max_blocks_per_seq];
            kblk[i] = kb ? reinterpret_cast<const char*>(kb) : reinterpret_cast<const char*>(p.kv_cache);
            vblk[i] = vb ? reinterpret_cast<const char*>(vb) : reinterpret_cast<const char*>(p.kv_cache);
        }
    }
    if constexpr (BEAM)
    {
        const int32_t* ci = p.cache_indirection + (int64_t) b * Smax;
        const int64_t seq_bytes = (int64_t) 2 * H * Smax * DH * ESZ;
        const int k_own = b % p.beam_width;
#pragma unroll
        for (int i = 0; i < NIT; ++i)
            sib[i] = (int64_t) (ci[min(t0 + i * NGRP + gid, Smax - 1)] - k_own) * seq_bytes;
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i)
    {
        // no branches around the loads (a divergent `if` makes the compiler drain the queue at every one):
        // out-of-range rows read the last row of the buffer and are masked out by `valid` below
        const int t = min(t0 + i * NGRP + gid, Smax - 1);
        int64_t off = ((int64_t) t * DH + li * 8) * ESZ;
        if constexpr (BEAM)
            off += sib[i];
        const char *kp = kbase + off, *vp = vbase + off;
        if constexpr (PAGED)
        {
            off = (((int64_t) h * p.tokens_per_block + (t & tpb_mask)) * DH + li * 8) * ESZ;
            kp = kblk[i] + off;
            vp = vblk[i] + off;
        }
        if constexpr (INT8KV)
        {
            const uint2 k8 = *reinterpret_cast<const uint2*>(kp);
            const uint2 v8 = *reinterpret_cast<const uint2*>(vp);
            kreg[i] = make_uint4(k8.x, k8.y, 0, 0);
            vreg[i] = make_uint4(v8.x, v8.y, 0, 0);
        }
        else
        {
            kreg[i] = *reinterpret_cast<const uint4*>(kp);
            vreg[i] = *reinterpret_cast<const uint4*>(vp);
        }
    }
    // ---- 2. the new token's q, k, v, the step scalars, the padding-mask words and the RoPE row: all requested
    //         here, before anything is consumed (one memory round trip for the whole prologue)
    const uint16_t* qkv = reinterpret_cast<const uint16_t*>(p.qkv) + (int64_t) b * 3 * H * DH;
    const uint4 q_raw = *reinterpret_cast<const uint4*>(qkv + (int64_t) h * DH + li * 8);
    const uint4 k_raw = *reinterpret_cast<const uint4*>(qkv + (int64_t) (H + h) * DH + li * 8);
    const uint4 v_new = *reinterpret_cast<const uint4*>(qkv + (int64_t) (2 * H + h) * DH + li * 8);
    const int tl = p.sequence_length[b]; // slots already used; the new token goes to slot tl
    int mk[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i)
        mk[i] = 0;
    if (p.masked_tokens) // uniform
    {
        const int32_t* mask = p.masked_tokens + (int64_t) b * Smax;
#pragma unroll
        for (int i = 0; i < NIT; ++i)
            mk[i] = mask[min(t0 + i * NGRP + gid, Smax - 1)];
    }
    // RoPE coefficients (cos, sin) of this lane's 8 elements: one row of the table, as 16-byte loads.
    // NeoX pairs (d, d + rot/2) share coefficient d mod rot/2; GPT-J pairs (2i, 2i+1) share coefficient i.
    float4 c4[4] = {make_float4(1.f, 0.f, 1.f, 0.f), make_float4(1.f, 0.f, 1.f, 0.f), make_float4(1.f, 0.f, 1.f, 0.f),
        make_float4(1.f, 0.f, 1.f, 0.f)};
    if (p.rotary_dim > 0) // uniform
    {
        const int half = p.rotary_dim >> 1;
        const float2* row;
        if (p.rope_row) // uniform: prepared by the sampler for this step
            row = reinterpret_cast<const float2*>(p.rope_row) + (int64_t) b * half;
        else
        {
            const int ts = p.timestep_host >= 0 ? p.timestep_host : tl;
            int pos = ts - (p.max_input_len - p.input_lengths[b]);
            pos = pos < 0 ? 0 : (pos >= p.rope_table_len ? p.rope_table_len - 1 : pos);
            row = reinterpret_cast<const float2*>(p.rope_table) + (int64_t) pos * half;
        }
        const int d0 = li * 8;
        int idx0 = p.neox ? (d0 >= half ? d0 - half : d0) : (d0 >> 1);
        const int span = p.neox ? 8 : 4;
        idx0 = idx0 + span <= half ? idx0 : (half >= span ? half - span : 0);
        const float4* r4 = reinterpret_cast<const float4*>(row + idx0);
        c4[0] = r4[0];
        c4[1] = r4[1];
        if (p.neox)
        {
            c4[2] = r4[2];
            c4[3] = r4[3];
        }
    }
    // splits that hold at least one used slot (or the slot the new token goes to): ONE expression for the early return here and
    // for the ticket count of the merge below - a split that returns never takes a ticket, and the last ticket is nact - 1
    const int nact = min(tl / G::TCHUNK + 1, nsplit_max);
    if (c >= nact)
        return; // uniform: this split lies entirely beyond the sequence
    float qf[8], kf[8];
    h8_to_f(q_raw, qf);
    h8_to_f(k_raw, kf);
    if (p.rotary_dim > 0)
    {
        const float cs[16] = {c4[0].x, c4[0].y, c4[0].z, c4[0].w, c4[1].x, c4[1].y, c4[1].z, c4[1].w,
            c4[2].x, c4[2].y, c4[2].z, c4[2].w, c4[3].x, c4[3].y, c4[3].z, c4[3].w};
        if (p.neox)
        {
            // pair (j, j + rot/2); rot == DH: the partner sits in lane li ^ (LPR/2)
            const bool second = li >= LPR / 2;
#pragma unroll
            for (int j = 0; j < 8; ++j)
            {
                const float qp = __shfl_xor(qf[j], LPR / 2, 64);
                const float kp = __shfl_xor(kf[j], LPR / 2, 64);
                // first half:  x' = x cos - y sin ; second half: y' = y cos + x sin
                // fp16 rounding of the rotated value (...Utils.h:1517-1531)
                const float c = cs[2 * j], sn = second ? cs[2 * j + 1] : -cs[2 * j + 1];
                qf[j] = h2f(f2h(c * qf[j] + sn * qp));
                kf[j] = h2f(f2h(c * kf[j] + sn * kp));
            }
        }
        else
        {
            // GPT-J style: pair (2i, 2i+1), both in this lane
#pragma unroll
            for (int j = 0; j < 8; j += 2)
            {
                const float c = (li * 8 + j < p.rotary_dim) ? cs[j] : 1.f;
                const float sn = (li * 8 + j < p.rotary_dim) ? cs[j + 1] : 0.f;
                const float q0 = qf[j], q1 = qf[j + 1], k0 = kf[j], k1 = kf[j + 1];
                qf[j] = h2f(f2h(c * q0 - sn * q1));
                qf[j + 1] = h2f(f2h(c * q1 + sn * q0));
                kf[j] = h2f(f2h(c * k0 - sn * k1));
                kf[j + 1] = h2f(f2h(c * k1 + sn * k0));
            }
        }
    }
    const uint4 q16 = f_to_h8(qf);
    const uint4 k_new = f_to_h8(kf);

    float s_oq = 1.f, s_qo = 1.f;
    if constexpr (INT8KV)
    {
        s_oq = p.kv_scale_orig_quant[0];
        s_qo = p.kv_scale_quant_orig[0];
    }

    // ---- 3. the workgroup that owns slot tl appends the new token to the cache
    if (tl / G::TCHUNK == c && gid == 0 && tl < Smax)
    {
        int64_t off = ((int64_t) tl * DH + li * 8) * ESZ;
        char *kw = kbase + off, *vw = vbase + off;
        bool mapped = true;
        if constexpr (PAGED)
        {
            // the sequence's OWN block (never a sibling's: a new token belongs to the hypothesis that consumed it)
            const int lg = 31 - __builtin_clz(p.tokens_per_block);
            const int64_t* row = p.block_pointers + (int64_t) b * 2 * p.max_blocks_per_seq + (tl >> lg);
            off = (((int64_t) h * p.tokens_per_block + (tl & tpb_mask)) * DH + li * 8) * ESZ;
            // the manager must have mapped the block of slot tl before this step (KVCacheManager.step); a missing block is
            // the caller's error and must not become a wild store
            mapped = row[0] != 0 && row[p.max_blocks_per_seq] != 0;
            kw = reinterpret_cast<char*>(row[0]) + off;
            vw = reinterpret_cast<char*>(row[p.max_blocks_per_seq]) + off;
        }
        if (mapped)
        {
            if constexpr (INT8KV)
            {
                *reinterpret_cast<uint2*>(kw) = quant8(k_new, s_oq);
                *reinterpret_cast<uint2*>(vw) = quant8(v_new, s_oq);
            }
            else
            {
                *reinterpret_cast<uint4*>(kw) = k_new;
                *reinterpret_cast<uint4*>(vw) = v_new;
            }
        }
    }

    // ---- 4. scores of this lane group's rows, its own softmax partial.
    // int8 cache: the bytes are turned into exact small integers (q + 128 spliced into fp16 1024 + u, minus 1152)
    // and the dequantisation scale is applied once to the dot product / to the P.V sum instead of to every element
    // (the reference rounds every dequantised element to fp16 first, ...Utils.h:2358-2365 - this is the same value
    // without that rounding; well inside the 2e-3 tolerance of the reference's own plugin test).
    float dnew = 0.f;
    dnew = dot2(q16.x, k_new.x, dnew);
    dnew = dot2(q16.y, k_new.y, dnew);
    dnew = dot2(q16.z, k_new.z, dnew);
    dnew = dot2(q16.w, k_new.w, dnew);
    dnew = group_sum<LPR>(dnew) * p.inv_sqrt_dh;
    const float kscale = INT8KV ? s_qo * p.inv_sqrt_dh : p.inv_sqrt_dh;
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
            const h2_t bias = {(_Float16) 1152.f, (_Float16) 1152.f};
            const uint32_t ka = kreg[i].x ^ 0x80808080u, kb = kreg[i].y ^ 0x80808080u;
            d = __builtin_amdgcn_fdot2(u32_as_h2(__builtin_amdgcn_perm(magic, ka, 0x04010400u)) - bias, u32_as_h2(q16.x), d, false);
            d = __builtin_amdgcn_fdot2(u32_as_h2(__builtin_amdgcn_perm(magic, ka, 0x04030402u)) - bias, u32_as_h2(q16.y), d, false);
            d = __builtin_amdgcn_fdot2(u32_as_h2(__builtin_amdgcn_perm(magic, kb, 0x04010400u)) - bias, u32_as_h2(q16.z), d, false);
            d = __builtin_amdgcn_fdot2(u32_as_h2(__builtin_amdgcn_perm(magic, kb, 0x04030402u)) - bias, u32_as_h2(q16.w), d, false);
        }
        else
        {
            d = dot2(q16.x, kreg[i].x, d);
            d = dot2(q16.y, kreg[i].y, d);
            d = dot2(q16.z, kreg[i].z, d);
            d = dot2(q16.w, kreg[i].w, d);
        }
        d = group_sum<LPR>(d) * kscale;
        if (t == tl)
            d = dnew; // the current token uses the un-quantised k (MM/...Template.h:1517-1549)
        const bool valid = t <= tl && t < Smax && (t == tl || mk[i] == 0);
        s[i] = valid ? d : -INFINITY;
        m_g = fmaxf(m_g, s[i]);
    }
    float l_g = 0.f, l16 = 0.f;
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float p_new = 0.f;
#pragma unroll
    for (int i = 0; i < NIT; ++i)
    {
        const int t = t0 + i * NGRP + gid;
        const float pr = (s[i] == -INFINITY) ? 0.f : __expf(s[i] - m_g);
        l_g += pr;
        const float p16 = h2f(f2h(pr));
        if (t == tl)
        {
            p_new = p16;
            continue;
        }
        if constexpr (INT8KV)
        {
            const uint32_t va = vreg[i].x ^ 0x80808080u, vb = vreg[i].y ^ 0x80808080u;
            l16 += p16;
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                o[j] = fmaf(p16, (float) ((va >> (8 * j)) & 0xffu), o[j]);
                o[4 + j] = fmaf(p16, (float) ((vb >> (8 * j)) & 0xffu), o[4 + j]);
            }
        }
        else
        {
            float vf[8];
            h8_to_f(vreg[i], vf);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                o[j] = fmaf(p16, vf[j], o[j]);
        }
    }
    {
        float vf[8];
        h8_to_f(v_new, vf);
#pragma unroll
        for (int j = 0; j < 8; ++j)
        {
            if constexpr (INT8KV)
                o[j] = s_qo * (o[j] - 128.f * l16); // sum p (u - 128) = sum p u - 128 sum p
            o[j] = fmaf(p_new, vf[j], o[j]);
        }
    }

    // ---- 5. merge the lane groups of the workgroup through LDS, publish the split partial
    __shared__ float sm_m[NGRP], sm_l[NGRP], sm_w[NGRP];
    __shared__ __attribute__((aligned(16))) float sm_o[NGRP][DH];
    *reinterpret_cast<float4*>(&sm_o[gid][li * 8]) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(&sm_o[gid][li * 8 + 4]) = make_float4(o[4], o[5], o[6], o[7]);
    if (li == 0)
    {
        sm_m[gid] = m_g;
        sm_l[gid] = l_g;
    }
    __syncthreads();
    float M = -INFINITY;
    for (int g = 0; g < NGRP; ++g)
        M = fmaxf(M, sm_m[g]);
    if (tid < NGRP)
        sm_w[tid] = (sm_m[tid] == -INFINITY) ? 0.f : __expf(sm_m[tid] - M);
    __syncthreads();
    const int64_t pi = ((int64_t) b * H + h) * nsplit_max + c;
    const bool tail = p.tail_tickets != nullptr; // uniform
    __shared__ __attribute__((aligned(16))) float sm_out[DH];
    for (int d = tid; d < DH; d += 256)
    {
        float L = 0.f, O = 0.f;
        for (int g = 0; g < NGRP; ++g)
        {
            L += sm_l[g] * sm_w[g];
            O += sm_o[g][d] * sm_w[g];
        }
        if (tail)
        {
            sm_out[d] = O;
            if (d == 0) // (m, l) as ONE 8-byte write-through store
                __hip_atomic_store(reinterpret_cast<uint64_t*>(ws_ml + pi),
                    (uint64_t) __float_as_uint(M) | ((uint64_t) __float_as_uint(L) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        else
        {
            ws_o[pi * DH + d] = O;
            if (d == 0)
                ws_ml[pi] = make_float2(M, L);
        }
    }
    if (!tail)
        return;
    // write-through (sc1) 16-byte stores of the partial: the merging workgroup may sit on another XCD, whose L2 never sees this
    // one's; scalar write-through stores are one fabric write each (guide: dword ~6 x the dwordx4 time per byte)
    __syncthreads();
    if (tid < DH / 4)
    {
        typedef float f32x4_t __attribute__((ext_vector_type(4)));
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(&sm_out[tid * 4]);
        float* dst = ws_o + pi * DH + tid * 4;
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
    }
    // ---- 6. (tail merge) the partial is written through -> one ticket per workgroup -> the last arriver of the (sequence, head)
    //         merges.  Protocol = the guide's R1 form (cdna_hip_programming.md G16: "sc1 payload -> asm vmcnt(0) -> relaxed agent flag;
    //         sc1 loads may replace the acquire when the producer stored sc1"): write-through payload, drained by every storing wave
    //         (vmcnt(0) in asm: the compiler may drop its own wait), workgroup barrier, relaxed agent-scope ticket; the consumer reads
    //         the payload with agent-scope (L1-bypassing) loads behind the ticket it took.  The ticket word re-arms itself for the
    //         next launch; the session also zeroes it whenever a prompt is loaded (a launch that died mid-way must not poison the next).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ uint32_t sm_ticket;
    if (tid == 0)
        sm_ticket = __hip_atomic_fetch_add(p.tail_tickets + b * H + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if ((int) sm_ticket != nact - 1)
        return;
    if (tid == 0) // re-arm for the next launch (ordered by the kernel boundary)
        __hip_atomic_store(p.tail_tickets + b * H + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int64_t base = ((int64_t) b * H + h) * nsplit_max;
    for (int d = tid; d < DH; d += 256)
    {
        // the arithmetic of the O-projection's merge prologue (gemv_impl.h PK_ATTN), slot order, fp32: bit-identical results.
        // Up to kTailSlots partials (the prologue form stops at 8: there every slot costs each of ~500 workgroups a load)
        float ms[kTailSlots], ls[kTailSlots], os[kTailSlots];
#pragma unroll
        for (int i = 0; i < kTailSlots; ++i)
        {
            const int ic = i < nact ? i : 0;
            const uint64_t mlb = __hip_atomic_load(reinterpret_cast<const uint64_t*>(ws_ml + base + ic), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ms[i] = __uint_as_float((uint32_t) mlb);
            ls[i] = __uint_as_float((uint32_t) (mlb >> 32));
            os[i] = __hip_atomic_load(ws_o + (base + ic) * DH + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        float Mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < kTailSlots; ++i)
            Mx = i < nact ? fmaxf(Mx, ms[i]) : Mx;
        float Lt = 0.f, Ot = 0.f;
#pragma unroll
        for (int i = 0; i < kTailSlots; ++i)
        {
            const bool act = i < nact && ms[i] != -INFINITY;
            const float e = act ? __expf(ms[i] - Mx) : 0.f;
            Lt += act ? ls[i] * e : 0.f;
            Ot += act ? os[i] * e : 0.f;
        }
        const uint16_t h16 = f2h(Ot * (1.f / (Lt + 1.e-6f)));
        const int64_t o = ((int64_t) b * H + h) * DH + d;
        reinterpret_cast<uint16_t*>(p.out)[o] = h16;
        if (p.tail_out_q8)
            reinterpret_cast<int8_t*>(p.tail_out_q8)[o] = f2i8_rni_sat(h2f(h16) * p.tail_quant_scale[0]);
    }
}

// ---- kernel 2: merge the splits of every (batch, head): 256 threads = (d, part), all loads independent
template <int DH, int NIT>
__global__ __launch_bounds__(256) void mmha_combine_kernel(const MmhaParams p, const float2* ws_ml, const float* ws_o, int nsplit_max)
{
    using G = MmhaGeom<DH, NIT>;
    constexpr int PARTS = 256 / DH > 0 ? 256 / DH : 1;
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int tl = p.sequence_length[b];
    int ns = tl / G::TCHUNK + 1;
    ns = ns > nsplit_max ? nsplit_max : ns;
    const int64_t base = ((int64_t) b * p.num_heads + h) * nsplit_max;
    __shared__ float sm_w[256];
    __shared__ float sm_l[PARTS];
    __shared__ float sm_o[PARTS][DH];
    // weights exp(m_i - M) of the splits (ns <= 256)
    float mine = -INFINITY, lmine = 0.f;
    if (tid < ns)
    {
        const float2 ml = ws_ml[base + tid];
        mine = ml.x;
        lmine = ml.y;
    }
    float M = mine;
#pragma unroll
    for (int mk = 32; mk >= 1; mk >>= 1)
        M = fmaxf(M, __shfl_xor(M, mk, 64));
    __shared__ float sm_mw[4];
    if ((tid & 63) == 0)
        sm_mw[tid >> 6] = M;
    __syncthreads();
    M = fmaxf(fmaxf(sm_mw[0], sm_mw[1]), fmaxf(sm_mw[2], sm_mw[3]));
    const float w = (mine == -INFINITY) ? 0.f : __expf(mine - M);
    sm_w[tid] = w;
    float lw = wave_sum(lmine * w);
    __syncthreads();
    if ((tid & 63) == 0)
        sm_mw[tid >> 6] = lw;
    const int d = tid % DH, part = tid / DH;
    float O = 0.f;
    if (part < PARTS)
        for (int i = part; i < ns; i += PARTS)
            O += ws_o[(base + i) * DH + d] * sm_w[i];
    if (part < PARTS)
        sm_o[part][d] = O;
    __syncthreads();
    if (tid < DH)
    {
        const float L = sm_mw[0] + sm_mw[1] + sm_mw[2] + sm_mw[3];
        float Ot = 0.f;
#pragma unroll
        for (int q = 0; q < PARTS; ++q)
            Ot += sm_o[q][tid];
        // inv_sum = 1 / (sum + 1e-6)  (MM/...Template.h:1756)
        reinterpret_cast<uint16_t*>(p.out)[((int64_t) b * p.num_heads + h) * DH + tid] = f2h(Ot * (1.f / (L + 1.e-6f)));
    }
}

template <int DH, int NIT>
int launch_nit(const MmhaParams& p, hipStream_t stream)
{
    using G = MmhaGeom<DH, NIT>;
    const int ns = (p.max_seq_len + G::TCHUNK - 1) / G::TCHUNK;
    if (ns > 256)
    {
        set_error("mmha: max_seq_len %d needs more than 256 splits", p.max_seq_len);
        return -1;
    }
    if (p.tail_tickets && (ns > kTailSlots || (p.tail_out_q8 && !p.tail_quant_scale)))
    {
        set_error("mmha: the in-launch merge takes at most %d splits (got %d) and a scale with its int8 output", kTailSlots, ns);
        return -1;
    }
    char* ws = reinterpret_cast<char*>(p.workspace);
    float2* ws_ml = reinterpret_cast<float2*>(ws);
    const size_t ml_bytes = ((size_t) p.batch * p.num_heads * ns * sizeof(float2) + 255) / 256 * 256;
    float* ws_o = reinterpret_cast<float*>(ws + ml_bytes);
    dim3 grid(ns, p.num_heads, p.batch);
    const bool beam = p.beam_width > 1 && p.cache_indirection;
    const int mode = p.block_pointers ? CACHE_PAGED : (beam ? CACHE_BEAM : CACHE_LINEAR);
#define TLLM_MMHA_LAUNCH(I8, MODE)                                                                                     \
    hipLaunchKernelGGL((mmha_partial_kernel<DH, NIT, I8, MODE>), grid, dim3(256), 0, stream, p, ws_ml, ws_o, ns)
    if (p.int8_kv)
    {
        if (mode == CACHE_PAGED)
            TLLM_MMHA_LAUNCH(true, CACHE_PAGED);
        else if (mode == CACHE_BEAM)
            TLLM_MMHA_LAUNCH(true, CACHE_BEAM);
        else
            TLLM_MMHA_LAUNCH(true, CACHE_LINEAR);
    }
    else
    {
        if (mode == CACHE_PAGED)
            TLLM_MMHA_LAUNCH(false, CACHE_PAGED);
        else if (mode == CACHE_BEAM)
            TLLM_MMHA_LAUNCH(false, CACHE_BEAM);
        else
            TLLM_MMHA_LAUNCH(false, CACHE_LINEAR);
    }
#undef TLLM_MMHA_LAUNCH
    if (!p.tail_tickets)
        hipLaunchKernelGGL((mmha_combine_kernel<DH, NIT>), dim3(p.num_heads, p.batch), dim3(256), 0, stream, p, ws_ml, ws_o, ns);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
    {
        set_error("mmha launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}

template <int DH>
int launch_dh(const MmhaParams& p, hipStream_t stream)
{
    if (p.rows_per_group == 16)
        return launch_nit<DH, 16>(p, stream);
    if (p.rows_per_group == 12)
        return launch_nit<DH, 12>(p, stream);
    return launch_nit<DH, 4>(p, stream);
}

} // namespace

size_t mmha_ticket_bytes(int32_t batch, int32_t num_heads)
{
    return ((size_t) batch * num_heads * sizeof(uint32_t) + 255) / 256 * 256;
}

size_t mmha_workspace_size(int32_t batch, int32_t num_heads, int32_t head_size, int32_t max_seq_len)
{
    // the merge tickets, then the partials sized for the finest split (NIT = 4)
    const int lpr = head_size / 8;
    if (lpr <= 0 || 64 % lpr)
        return 0;
    const int tchunk = kWaves * (64 / lpr) * 4;
    const int ns = (max_seq_len + tchunk - 1) / tchunk;
    const size_t ml = ((size_t) batch * num_heads * ns * sizeof(float2) + 255) / 256 * 256;
    return mmha_ticket_bytes(batch, num_heads) + ml + (size_t) batch * num_heads * ns * head_size * sizeof(float);
}

int mmha_split_layout(int32_t head_size, int32_t max_seq_len, int32_t rows_per_group, int32_t batch, int32_t num_heads,
    int32_t* tchunk, int32_t* nsplit, size_t* out_offset)
{
    const int lpr = head_size / 8;
    if (lpr <= 0 || 64 % lpr)
        return -1;
    const int nit = rows_per_group == 16 ? 16 : (rows_per_group == 12 ? 12 : 4);
    const int tc = kWaves * (64 / lpr) * nit;
    const int ns = (max_seq_len + tc - 1) / tc;
    if (tchunk)
        *tchunk = tc;
    if (nsplit)
        *nsplit = ns;
    if (out_offset)
        *out_offset = ((size_t) batch * num_heads * ns * sizeof(float2) + 255) / 256 * 256;
    return 0;
}

int mmha_reset_workspace(void* workspace, int32_t batch, int32_t num_heads, hipStream_t stream)
{
    if (hipMemsetAsync(workspace, 0, (size_t) batch * num_heads * sizeof(unsigned), stream) != hipSuccess)
    {
        set_error("mmha: hipMemsetAsync(tickets) failed");
        return -1;
    }
    return 0;
}

void fill_rope_table_host(float* table, int32_t max_pos, int32_t rotary_dim)
{
    const int half = rotary_dim / 2;
    for (int pos = 0; pos < max_pos; ++pos)
        for (int j = 0; j < half; ++j)
        {
            // rotary_embedding_coefficient(zid = 2j, rot, t): inv_freq = t / pow(10000, zid / rot)  (fp32)
            const float inv_freq = (float) pos / powf(10000.0f, (float) (2 * j) / (float) rotary_dim);
            table[((int64_t) pos * half + j) * 2 + 0] = cosf(inv_freq);
            table[((int64_t) pos * half + j) * 2 + 1] = sinf(inv_freq);
        }
}

int launch_mmha(const MmhaParams& p, hipStream_t stream)
{
    if (p.rotary_dim > 0)
    {
        if (!p.rope_table || p.rope_table_len <= 0)
        {
            set_error("mmha: rotary embedding requested without a RoPE table");
            return -1;
        }
        if (p.neox && p.rotary_dim != p.head_size)
        {
            set_error("mmha: NeoX rotary needs rotary_dim == head_size (got %d vs %d)", p.rotary_dim, p.head_size);
            return -1;
        }
    }
    if (p.int8_kv && (!p.kv_scale_orig_quant || !p.kv_scale_quant_orig))
    {
        set_error("mmha: int8 KV cache needs both scales");
        return -1;
    }
    if (p.timestep_host >= p.max_seq_len)
    {
        set_error("mmha: timestep %d exceeds cache capacity %d (circular cache not supported)", p.timestep_host,
            p.max_seq_len);
        return -1;
    }
    if (p.block_pointers)
    {
        const int t = p.tokens_per_block;
        if (t < 1 || (t & (t - 1)) || (int64_t) p.max_blocks_per_seq * t < p.max_seq_len)
        {
            set_error("mmha: paged KV cache needs tokens_per_block a power of two and max_blocks_per_seq * tokens_per_block >= "
                      "max_seq_len (got %d x %d for %d)", p.max_blocks_per_seq, t, p.max_seq_len);
            return -1;
        }
        if (p.beam_width > 1 && !p.cache_indirection)
        {
            set_error("mmha: beam width %d without a cache indirection", p.beam_width);
            return -1;
        }
    }
    switch (p.head_size)
    {
    case 32: return launch_dh<32>(p, stream);
    case 64: return launch_dh<64>(p, stream);
    case 128: return launch_dh<128>(p, stream);
    case 256: return launch_dh<256>(p, stream);
    default: set_error("mmha: head_size %d not supported (32, 64, 128, 256)", p.head_size); return -1;
    }
}

} // namespace kernels
} // namespace tllm
