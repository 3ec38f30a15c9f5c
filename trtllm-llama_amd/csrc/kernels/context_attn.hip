// Context (prompt) phase of the attention plugin.
//
// Reference: enqueueContext, P/gptAttentionCommon/gptAttentionCommon.cpp:362-620 — five kernels and a 134 MB
// fp32 score scratch at S=1024: invokeAddFusedQKVBiasTranspose (RoPE), invokeTranspose4dBatchMajor (KV write,
// int8 quant), cuBLAS QK^T (fp32 out), invokeMaskedSoftmax, cuBLAS PV, invokeTransposeQKV.
// Here: (1) one pass applies RoPE to q,k in place in the packed QKV buffer (the reference rewrites it too,
// K/unfusedAttentionKernels.cu:1401-1403) and writes RoPE'd K and raw V into the cache [B,2,H,Smax,Dh]
// (int8: sat(rni(x * s)); padded rows are written as zero); (2) a flash-style pass, online softmax, no score matrix:
// MFMA tiles for head sizes 64 / 128 when the caller provides the V^T scratch (see context_attn_mfma_ks_kernel), else
// one wave per query row reading K/V straight from the packed buffer.
// Numerics as the decode kernel: fp32 dot / softmax / accumulation, probabilities rounded to fp16, one
// rounding to fp16 at the end; keys j > i or j >= input_len[b] are excluded (the reference adds -10000, whose
// exp underflows to the same 0).
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

// first row of sequence b in the (padded or packed) token-major buffers
__device__ __forceinline__ int64_t seq_row0(const ContextAttnParams& p, int b)
{
    return p.cu_seqlens ? (int64_t) p.cu_seqlens[b] : (int64_t) b * p.seq;
}

// One (token s, head h) of sequence b, handled by a group of DH / 8 consecutive lanes (li = lane in the group): RoPE on q and k in
// place, padding rows zeroed, k / v appended to the cache (rope_kv_load + rope_kv_finish).  rope_kv_finish returns this lane's 8 v
// elements as attention will see them (zero for padding positions); every lane of the group must call it (shuffles inside the group).
struct RopeRow
{
    uint4 q4, k4, v4;
    uint16_t *qp, *kp, *vp;
    bool valid, has_row;
};

// the loads of one (token, head) - split from the rest so that a caller with several rows can have all of them in flight
template <int DH>
__device__ __forceinline__ RopeRow rope_kv_load(const ContextAttnParams& p, int s, int h, int b, int li)
{
    RopeRow r;
    const int H = p.num_heads;
    r.valid = s < p.input_lengths[b];
    r.has_row = r.valid || !p.cu_seqlens; // packed buffers hold no padding rows
    uint16_t* row = reinterpret_cast<uint16_t*>(p.qkv) + (seq_row0(p, b) + (r.has_row ? s : 0)) * 3 * H * DH;
    r.qp = row + (int64_t) h * DH + li * 8;
    r.kp = row + (int64_t) (H + h) * DH + li * 8;
    r.vp = row + (int64_t) (2 * H + h) * DH + li * 8;
    r.q4 = *reinterpret_cast<const uint4*>(r.qp);
    r.k4 = *reinterpret_cast<const uint4*>(r.kp);
    r.v4 = *reinterpret_cast<const uint4*>(r.vp);
    return r;
}

template <int DH>
__device__ __forceinline__ uint4 rope_kv_finish(const ContextAttnParams& p, int s, int h, int b, int li, const RopeRow& r)
{
    constexpr int LPR = DH / 8;
    const int H = p.num_heads;
    const bool valid = r.valid, has_row = r.has_row;
    uint16_t *qp = r.qp, *kp = r.kp, *vp = r.vp;
    uint4 q4 = r.q4, k4 = r.k4, v4 = r.v4;
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
    if (has_row)
    {
        *reinterpret_cast<uint4*>(qp) = q4;
        *reinterpret_cast<uint4*>(kp) = k4;
        if (!valid)
            *reinterpret_cast<uint4*>(vp) = v4;
    }
    if (s < p.max_seq_len)
    {
        const int esz = p.int8_kv ? 1 : 2;
        char* kc = reinterpret_cast<char*>(p.kv_cache)
            + (((int64_t) (b * p.cache_seq_stride * 2 + 0) * H + h) * p.max_seq_len + s) * DH * esz + li * 8 * esz;
        char* vc = reinterpret_cast<char*>(p.kv_cache)
            + (((int64_t) (b * p.cache_seq_stride * 2 + 1) * H + h) * p.max_seq_len + s) * DH * esz + li * 8 * esz;
        bool mapped = true;
        if (p.block_pointers) // uniform: paged cache, block s / tokens_per_block of the sequence, row s % tokens_per_block
        {
            const int lg = 31 - __builtin_clz(p.tokens_per_block);
            const int64_t* row = p.block_pointers + (int64_t) b * p.cache_seq_stride * 2 * p.max_blocks_per_seq + (s >> lg);
            const int64_t off = (((int64_t) h * p.tokens_per_block + (s & (p.tokens_per_block - 1))) * DH + li * 8) * esz;
            // a table entry of 0 is a logical block the cache manager has not handed out yet (KVCacheManager.add_sequence
            // allocates ceil((len + 1) / tokens_per_block) blocks and grows on demand, PY/runtime/kv_cache_manager.py): such a
            // block holds only padding positions, which the reference never touches (K/kvCacheUtils.h:34-112) - skip them
            mapped = row[0] != 0 && row[p.max_blocks_per_seq] != 0;
            kc = reinterpret_cast<char*>(row[0]) + off;
            vc = reinterpret_cast<char*>(row[p.max_blocks_per_seq]) + off;
        }
        if (!mapped)
            return v4;
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
    return v4;
}

// grid (S, ceil(H / HG), B), 256 threads: DH/8 lanes own the 8-element pieces of one head of one token, HG = 256 / (DH/8)
// heads per workgroup (one (token, head) per 64-thread workgroup with 16 active lanes took 19 us per layer at S = 1024).
template <int DH>
__global__ __launch_bounds__(256) void rope_kv_write_kernel(const ContextAttnParams p)
{
    constexpr int LPR = DH / 8, HG = 256 / LPR;
    const int s = blockIdx.x, h = blockIdx.y * HG + threadIdx.x / LPR, b = blockIdx.z;
    if (h >= p.num_heads) // whole lane groups (the shuffles stay inside a group)
        return;
    const int li = threadIdx.x % LPR;
    const RopeRow r = rope_kv_load<DH>(p, s, h, b, li);
    (void) rope_kv_finish<DH>(p, s, h, b, li, r);
}

// The same per-row work for a tile of 64 tokens of ONE head, plus the V^T image the MFMA kernel stages its PV operand from
// ([B, H, DH, spad] fp16, keys contiguous; keys beyond the sequence are zero): the v values pass through an LDS tile and leave
// transposed, so V is read once instead of by a transpose launch of its own (14.1 + 6.7 us per layer at S = 1024 before).
// grid (spad / 64, H, B), 256 threads.
template <int DH>
__global__ __launch_bounds__(256) void rope_kv_vt_kernel(const ContextAttnParams p, int spad)
{
    constexpr int LPR = DH / 8, RPP = 256 / LPR; // token rows per pass
    constexpr int PITCH = DH + 2;                // halfs; odd number of dwords -> the column reads below are conflict-free
    __shared__ uint16_t tile[64 * PITCH];
    const int kv0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const int li = threadIdx.x % LPR;
    constexpr int NP = 64 / RPP;
    RopeRow rr[NP];
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) // every row's q / k / v requested before the first one is touched
    {
        const int s = kv0 + pass * RPP + threadIdx.x / LPR;
        rr[pass] = rope_kv_load<DH>(p, s < p.seq ? s : p.seq - 1, h, b, li);
    }
#pragma unroll
    for (int pass = 0; pass < NP; ++pass)
    {
        const int key = pass * RPP + threadIdx.x / LPR, s = kv0 + key;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (s < p.seq) // group-uniform
            v = rope_kv_finish<DH>(p, s, h, b, li, rr[pass]);
        uint32_t* d = reinterpret_cast<uint32_t*>(tile + key * PITCH + li * 8);
        d[0] = v.x;
        d[1] = v.y;
        d[2] = v.z;
        d[3] = v.w;
    }
    __syncthreads();
    uint16_t* vt = reinterpret_cast<uint16_t*>(p.workspace) + ((int64_t) (b * p.num_heads + h) * DH) * spad + kv0;
    for (int i = threadIdx.x; i < DH * 8; i += 256)
    {
        const int g = i % 8, d = i / 8; // 8 keys g * 8 .. + 8 of column d: eight consecutive lanes fill one 128-byte line of V^T
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            w[j] = (uint32_t) tile[(g * 8 + 2 * j) * PITCH + d] | ((uint32_t) tile[(g * 8 + 2 * j + 1) * PITCH + d] << 16);
        *reinterpret_cast<uint4*>(vt + (int64_t) d * spad + g * 8) = make_uint4(w[0], w[1], w[2], w[3]);
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
    uint16_t* outp = reinterpret_cast<uint16_t*>(p.out) + ((seq_row0(p, b) + qi) * H + h) * DH + li * 8;
    if (qi >= len)
    {
        if (grp == 0 && !p.cu_seqlens) // packed outputs have no padding rows
            *reinterpret_cast<uint4*>(outp) = make_uint4(0, 0, 0, 0);
        return;
    }
    const uint16_t* base = reinterpret_cast<const uint16_t*>(p.qkv) + seq_row0(p, b) * 3 * H * DH;
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

// ------------------------------------------------------------------------------------------------------------------
// MFMA (flash-style) context attention, DH in {64, 128}  (SURVEY 8f rank 3).
//
// Workgroup = 4 waves = 128 consecutive queries of one (batch, head); wave w owns 32 of them.  Keys / values advance in
// blocks of 64.  Operand roles are chosen so that everything the softmax needs is lane-local:
//   S^T[key, q]  = K[key, :] . Q^T[:, q]     MFMA 32x32x16 f16:  A = K rows (LDS), B = Q (registers, loaded once)
//       -> lane (q = lane & 31, half = lane >> 5) holds 2 x 16 scores of ITS query: row max / sum are in-register plus
//          one exchange with lane ^ 32; the running max, the rescale factor and 1/l are per-lane scalars;
//   O^T[d, q]   += V^T[d, keys] . P^T[keys, q]:                  A = V^T rows (LDS), B = P^T = the score registers
//          converted to fp16 in place - no LDS round trip for P.
// The C-layout of S^T puts MFMA row m = 8a + 4*half + c in register 4a + c; feeding K row pi(m) (pi swaps bits 2 and 3)
// to MFMA row m makes register r of half h hold key 16 (r >> 3) + 8 h + (r & 7): exactly the 8 consecutive keys the
// B operand of the PV product wants, and the V^T fragment is one 16-byte LDS read.
// V^T comes from a scratch the transpose kernel below fills ([B, H, DH, Spad] fp16, keys contiguous), so both tiles
// are 128-byte-row images staged by LDS-DMA exactly like the GEMM's (gemm_glds.hip), double-buffered.
// Numerics: fp32 scores / softmax / accumulation, probabilities rounded to fp16 for the PV product, normaliser
// 1 / (l + 1e-6), one rounding of the result to fp16 - the same rounding points as the wave-per-query kernel above
// (online softmax over 64-key blocks instead of 16-key ones).
// ------------------------------------------------------------------------------------------------------------------
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16(const void* gptr, uint32_t lds_byte)
{
    // inline asm on purpose: see gemm_glds.hip (the builtin makes hipcc drain the DMA before the next ds_read)
    uint32_t m0_keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(m0_keep) : "v"(gptr), "s"(lds_byte) : "memory");
}

__device__ __forceinline__ int swz128(int row, int c16)
{
    return row * 128 + ((c16 ^ ((row >> 1) & 7)) << 4);
}


// The MFMA kernel launch_dh uses (key-split; r01's 4 compute + 4 loader wave form was removed in r05): 2 NQ compute waves (NQ = 4: two per SIMD), no loader waves.
// Waves w and w + NQ serve the same 32 queries and split every 64-key block between them (keys 32 kh .. 32 kh + 31, kh = w / NQ):
// per block each wave runs half the QK^T MFMAs, the softmax of 16 scores per lane and half the PV MFMAs on its own running
// (m, l, O), so that one wave's MFMAs run under the other's softmax arithmetic on the same SIMD - with one compute wave per SIMD
// the three stages were strictly serial (r02 ablation at S = 1024: 40 us = 23 us of VALU-only + 20 us of MFMA-only work on the
// longest workgroup).  Each wave issues an eighth of the LDS-DMA pieces of the next block right after the barrier; the issue
// stall is covered by the sibling wave.  The two partial results are merged once, after the last block, through LDS
// (flash-decoding style: O = O0 a0 + O1 a1, l likewise, a = exp2((m_i - max) c2)).
// Arithmetic: scores stay unscaled, p = exp2(s c2 - m c2) with c2 = scale log2(e) is one FMA + v_exp per score and the running
// max is kept in score units (scale > 0, checked by the launcher); exp2(-inf) = 0 removes the masked scores without a select; the
// mask runs only on the half-blocks that cross the diagonal; the accumulator rescale is skipped while no lane's maximum moved.
// grid (H, ceil(S / (32 NQ)), B), 128 NQ threads; heads fastest, heavy (late) query blocks first: the dispatcher sees the longest
// workgroups first and every query block of head h runs on XCD h mod 8, whose L2 serves that head's K / V^T tiles to all of them.
//
// PAIR (NQ = 4 only): the workgroup serves TWO 64-query blocks instead of one 128-query block - the heaviest remaining one
// (block nblk - 1 - y: nblk - y key blocks) on query slices 0 / 1 and the lightest (block y: y + 1 key blocks) on slices 2 / 3 -
// over the same K / V^T stages.  With one 128-query workgroup per CU (256 workgroups at S = 1024 and 32 heads: exactly one
// round) the launch lasted as long as its last query block (16 key blocks) while the CU of the first one was done after 2;
// paired, every workgroup runs nblk + 1 half-blocks of work.  The light slices fall out of the loop body through the causal
// skip that was there already (they still arrive at the barriers), and the waves are dealt so that every SIMD hosts one
// heavy and one light wave: wave kh * 4 + s takes slice s (kh = 0) or slice (s + 2) % 4 (kh = 1).
//
// RING: three operand stages instead of two - block t + 2 is requested at step t, so TWO blocks are in flight and the wait in
// front of the barrier is a counted one (the pieces this wave issued for block t + 1 stay outstanding).  With two stages the
// launch could not be shorter than its chain of request -> landing latencies: the paired kernel with its arithmetic removed
// (DMA, waits and barriers only) still took 22 of its 27 us at S = 1024 (profiles/r02_ctx_attn_pairing.txt, box D).
template <int DH, int NQ, bool PAIR = false, int NST = 2>
__global__ __launch_bounds__(128 * NQ) void context_attn_mfma_ks_kernel(const ContextAttnParams p, int spad)
{
    static_assert(!PAIR || NQ == 4, "pairing deals four query slices");
    constexpr int NSUB = DH / 64;              // 128-byte sub-tiles of a K row
    constexpr int KST = DH / 16;               // k-steps of the QK product
    constexpr int DT = DH / 32;                // 32-row tiles of O^T / V^T
    constexpr int K_BYTES = 64 * DH * 2;       // K tile  [NSUB][64][128 B]
    constexpr int STAGE = K_BYTES + DH * 128;  // + V^T tile [DH][128 B]
    constexpr int NWV = 2 * NQ;                // waves: NQ query slices of 32 x 2 key halves
    constexpr int QB = 32 * NQ;                // queries per workgroup
    constexpr int CHUNKS = STAGE / 1024, CPW = CHUNKS / NWV;
    constexpr int SLAB = (2 + 16 * DT) * 64 * 4; // merge record of one wave: m, l, O per lane
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kh = wave / NQ;                                     // key half
    const int qs = (PAIR && kh) ? (wave + 2) % NQ : wave % NQ;    // query slice
    const int h = blockIdx.x, b = blockIdx.z;
    const int H = p.num_heads, S = p.seq;
    // PAIR: 64-query blocks, heavy = the y-th from the end (slices 0 / 1), light = the y-th from the start (slices 2 / 3; nobody
    // when the two meet in the middle block of an odd count)
    const int nblk64 = (S + 63) / 64;
    const int heavy = PAIR ? nblk64 - 1 - (int) blockIdx.y : 0, light = PAIR ? (int) blockIdx.y : 0;
    const int qb = PAIR ? heavy : gridDim.y - 1 - blockIdx.y; // the block that sets this workgroup's key range
    const int q0 = PAIR ? (qs < 2 ? heavy : light) * 64 + (qs & 1) * 32 : qb * QB + qs * 32; // first query of this wave
    const int ql = lane & 31, hf = lane >> 5;
    const int q = q0 + ql;
    const int len = p.input_lengths[b];
    const int64_t rs = (int64_t) 3 * H * DH * 2; // bytes per token row of the fused QKV buffer
    const char* qkv = reinterpret_cast<const char*>(p.qkv) + seq_row0(p, b) * rs;
    const int nrows = p.cu_seqlens ? len : S; // token rows of this sequence that exist in the buffers
    const char* vt = reinterpret_cast<const char*>(p.workspace) + ((int64_t) (b * H + h) * DH) * spad * 2;

    // Q fragments (B operand): lane (q, half) holds d = 16 s + 8 half .. + 8 for every k-step s
    uint4 qf[KST];
    {
        const char* qrow = qkv + (int64_t) (q < nrows ? q : nrows - 1) * rs + (int64_t) h * DH * 2;
#pragma unroll
        for (int s = 0; s < KST; ++s)
            qf[s] = *reinterpret_cast<const uint4*>(qrow + (16 * s + 8 * hf) * 2);
    }

    bool valid = true; // this wave's slice exists (PAIR: wave-uniform; else block-uniform and handled by the return below)
    int kv_end;
    if constexpr (PAIR)
    {
        // packed inputs: blocks beyond this sequence do not exist - the key range is that of the heaviest block that does
        const int top = heavy * 64 < nrows ? heavy : (light * 64 < nrows ? light : -1);
        if (top < 0) // block-uniform
            return;
        valid = q0 < nrows && !(qs >= 2 && light == heavy);
        kv_end = top * 64 + 64 < nrows ? top * 64 + 64 : nrows;
    }
    else
    {
        if (qb * QB >= nrows) // packed inputs: this query block lies entirely beyond the sequence (block-uniform)
            return;
        kv_end = (qb * QB + QB < nrows ? qb * QB + QB : nrows); // causal: keys <= the block's last query
    }
    const int nkb = (kv_end + 63) / 64;
    const uint32_t lds_base = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) void*) lds;
    // PAIR: once the light block's last key block is behind (t > light: all four light waves skip the body from there on) its
    // waves take over the whole LDS-DMA issue - an instruction holds its issuer for 100 - 185 cycles, and from that point the
    // heavy waves are alone on their SIMDs with nothing to cover it
    const bool is_light = PAIR && qs >= 2;
    const int li = (qs - 2) * 2 + kh; // 0 .. 3 among the light waves
    constexpr bool RING = NST > 2;
    constexpr int AHEAD = NST - 1; // block t is requested at step t - AHEAD: NST - 1 blocks in flight
    auto stage_of = [&](int t) { return NST == 2 ? (t & 1) : t % NST; };
    auto handed_at = [&](int t) { return PAIR && (t - AHEAD > light || light == heavy); }; // workgroup-uniform
    // LDS-DMA pieces THIS wave issues for block t
    auto pieces_of = [&](int t) { return t >= nkb ? 0 : (handed_at(t) ? (is_light ? 2 * CPW : 0) : CPW); };
    auto issue = [&](int t) {
        const int kv0 = t * 64;
        const bool handed = handed_at(t);
        if (handed && !is_light)
            return;
        constexpr int NI = PAIR ? 2 * CPW : CPW;
#pragma unroll
        for (int i = 0; i < NI; ++i)
        {
            if (i >= CPW && !handed)
                break;
            const int c = handed ? i * 4 + li : i * NWV + wave;
            const int r8 = lane >> 3;
            const char* src;
            if (c < 8 * NSUB)
            {
                const int sub = c / 8, row = (c % 8) * 8 + r8;
                const int col = (lane & 7) ^ ((row >> 1) & 7);
                const int key = kv0 + row < nrows ? kv0 + row : nrows - 1;
                src = qkv + (int64_t) key * rs + (int64_t) (H + h) * DH * 2 + sub * 128 + col * 16;
            }
            else
            {
                const int d = (c - 8 * NSUB) * 8 + r8;
                const int col = (lane & 7) ^ ((d >> 1) & 7);
                src = vt + ((int64_t) d * spad + kv0) * 2 + col * 16;
            }
            glds16(src, lds_base + stage_of(t) * STAGE + c * 1024);
        }
    };
    f32x16_t oacc[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            oacc[i][r] = 0.f;
    float m = -INFINITY, l = 0.f; // m in units of the unscaled score
    const float c2 = p.inv_sqrt_dh * 1.4426950408889634f;
    const int krow = (ql & ~12) | ((ql & 4) << 1) | ((ql & 8) >> 1); // pi(ql): bits 2 and 3 swapped

    // The Q fragments are consumed here, in front of the first LDS-DMA: hipcc does not see the hand-written vmcnt wait at the top of
    // the loop, so it kept its own `s_waitcnt vmcnt(7) .. vmcnt(0)` for these eight loads in front of the eight QK MFMAs of EVERY
    // iteration - where the only memory operations still in flight are this wave's LDS-DMA pieces of the NEXT block, which the
    // last MFMA of the block then waited for (r02, found in the ISA: the DMA latency sat inside the compute chain of every step).
#pragma unroll
    for (int s = 0; s < KST; ++s)
        asm volatile("" : "+v"(qf[s].x), "+v"(qf[s].y), "+v"(qf[s].z), "+v"(qf[s].w));
    issue(0);
#pragma unroll
    for (int a = 1; a < AHEAD; ++a)
        if (a < nkb)
            issue(a);
    for (int t = 0; t < nkb; ++t)
    {
        // this wave's pieces of block t have landed (RING: those of block t + 1 may still be on their way - loads retire in order)
        if constexpr (RING)
        {
            int keep = 0; // wave-uniform: the younger pieces of this wave, blocks t + 1 .. t + AHEAD - 1
#pragma unroll
            for (int a = 1; a < AHEAD; ++a)
                keep += pieces_of(t + a);
            switch (keep / CPW)
            {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CPW) : "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * CPW) : "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * CPW) : "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * CPW) : "memory"); break;
            }
        }
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads(); // block t is in LDS for everybody; everybody is done with block t - 1
        const bool more = t + AHEAD < nkb; // block t + AHEAD goes into the stage block t - 1 occupied
        const int kb0 = t * 64 + 32 * kh; // first key of this wave's half-block
        if (more)
            issue(t + AHEAD); // right behind the barrier: between the MFMAs (+4.5 us), behind them (+2.8) or inside the softmax (+1) all lost
        if (kb0 > q0 + 31 || !valid) // every key of it lies in the future of every query of this wave (wave-uniform)
            continue;
        const char* Ks = lds + stage_of(t) * STAGE;
        const char* Vs = Ks + K_BYTES;
        // two accumulators (even / odd k-steps): eight MFMAs into one are a chain of eight result latencies, and once the light
        // waves have dropped out (PAIR) or the sibling sits at the barrier nothing else runs on this SIMD
        f32x16_t sacc, sacc1;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            sacc[r] = 0.f, sacc1[r] = 0.f;
        // all K fragments requested before the first MFMA (left alone hipcc reused one register quad: read, wait, MFMA, eight times)
        uint4 kfr[KST];
#pragma unroll
        for (int s = 0; s < KST; ++s)
            kfr[s] = *reinterpret_cast<const uint4*>(Ks + (s >> 2) * (64 * 128) + swz128(kh * 32 + krow, (2 * s + hf) & 7));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KST; ++s)
        {
            f16x8_t ak, bq;
            __builtin_memcpy(&ak, &kfr[s], 16);
            __builtin_memcpy(&bq, &qf[s], 16);
            if (s & 1)
                sacc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ak, bq, sacc1, 0, 0, 0);
            else
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ak, bq, sacc, 0, 0, 0);
        }
        // the V^T fragments of both key octets are requested now, under the softmax arithmetic
        uint4 vfr[2][DT];
#pragma unroll
        for (int s2l = 0; s2l < 2; ++s2l)
#pragma unroll
            for (int i = 0; i < DT; ++i)
                vfr[s2l][i] = *reinterpret_cast<const uint4*>(Vs + swz128(i * 32 + ql, 2 * (2 * kh + s2l) + hf));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            sacc[r] += sacc1[r];
        if (kb0 + 31 > q0) // wave-uniform: the half-block crosses the diagonal for some query of this wave
        {
#pragma unroll
            for (int r = 0; r < 16; ++r)
            {
                const int key = kb0 + 16 * (r >> 3) + 8 * hf + (r & 7);
                sacc[r] = key <= q ? sacc[r] : -INFINITY;
            }
        }
        float mt = sacc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r)
            mt = fmaxf(mt, sacc[r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        // a query whose keys in this half-block are all masked (the first half-blocks of the diagonal) keeps m = -inf: its p
        // must be 0, not exp2(-inf + inf)
        const float mn = fmaxf(m, mt);
        const float alpha = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m - mn) * c2);
        m = mn;
        const float mc = (mn == -INFINITY) ? 0.f : -mn * c2;
        l *= alpha;
        if (!__all(alpha == 1.f)) // wave-uniform
        {
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    oacc[i][r] *= alpha;
        }
        // probabilities -> fp16 B fragments (8 consecutive keys per register octet), PV product over this wave's 32 keys
#pragma unroll
        for (int s2l = 0; s2l < 2; ++s2l)
        {
            f16x8_t bp;
#pragma unroll
            for (int j = 0; j < 8; ++j)
            {
                const float pr = __builtin_amdgcn_exp2f(fmaf(sacc[8 * s2l + j], c2, mc));
                l += pr;
                bp[j] = (_Float16) pr;
            }
#pragma unroll
            for (int i = 0; i < DT; ++i)
            {
                f16x8_t av;
                __builtin_memcpy(&av, &vfr[s2l][i], 16);
                oacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bp, oacc[i], 0, 0, 0);
            }
        }
    }

    // ---- merge the key halves: wave (qs, 1) hands (m, l, O) to wave (qs, 0), lane to lane
    __syncthreads(); // every wave is done with the operand stages
    float* slab = reinterpret_cast<float*>(lds + qs * SLAB);
    if (kh == 1)
    {
        slab[lane] = m;
        slab[64 + lane] = l;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                slab[(2 + 16 * i + r) * 64 + lane] = oacc[i][r];
    }
    __syncthreads();
    if (kh == 1 || !valid)
        return;
    {
        const float m1 = slab[lane], l1 = slab[64 + lane];
        const float mn = fmaxf(m, m1); // m is finite: block 0 holds key 0 of every query
        const float a0 = __builtin_amdgcn_exp2f((m - mn) * c2);
        const float a1 = (m1 == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m1 - mn) * c2);
        l = l * a0 + l1 * a1;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                oacc[i][r] = oacc[i][r] * a0 + slab[(2 + 16 * i + r) * 64 + lane] * a1;
    }

    // ---- normalise, transpose through LDS (inside this wave's own slab, read completely above), store whole rows
    l += __shfl_xor(l, 32, 64);
    const float inv = (q < len) ? 1.f / (l + 1.e-6f) : 0.f; // padding queries produce zero rows
    constexpr int PITCH = DH * 2 + 16;
    static_assert(32 * PITCH <= SLAB, "the output scratch must fit the wave's slab");
    char* scr = lds + qs * SLAB;
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
        {
            const uint32_t w0 = pack_h2(oacc[i][4 * g] * inv, oacc[i][4 * g + 1] * inv);
            const uint32_t w1 = pack_h2(oacc[i][4 * g + 2] * inv, oacc[i][4 * g + 3] * inv);
            *reinterpret_cast<uint2*>(scr + ql * PITCH + (32 * i + 8 * g + 4 * hf) * 2) = make_uint2(w0, w1);
        }
    constexpr int PPR = DH / 8; // 16-byte pieces per output row
    uint16_t* outp = reinterpret_cast<uint16_t*>(p.out) + seq_row0(p, b) * H * DH + (int64_t) h * DH;
#pragma unroll
    for (int i = lane; i < 32 * PPR; i += 64)
    {
        const int row = i / PPR, pc = i % PPR;
        const uint4 v = *reinterpret_cast<const uint4*>(scr + row * PITCH + pc * 16);
        if (q0 + row < nrows)
        {
            if (p.out_q8) // uniform: the static quantiser of the O-projection's input, on the fp16-rounded values
            {
                const float qsc = p.out_q_scale[0];
                const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
                uint32_t o2[2] = {0, 0};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                {
                    const uint32_t b0 = (uint8_t) f2i8_rni_sat(h2f((uint16_t) (w4[e] & 0xffffu)) * qsc);
                    const uint32_t b1 = (uint8_t) f2i8_rni_sat(h2f((uint16_t) (w4[e] >> 16)) * qsc);
                    o2[e >> 1] |= (b0 | (b1 << 8)) << (16 * (e & 1));
                }
                int8_t* qp = p.out_q8 + (seq_row0(p, b) + q0 + row) * (int64_t) H * DH + (int64_t) h * DH + pc * 8;
                *reinterpret_cast<uint2*>(qp) = make_uint2(o2[0], o2[1]);
            }
            else
                *reinterpret_cast<uint4*>(outp + (int64_t) (q0 + row) * H * DH + pc * 8) = v;
        }
    }
}

template <int DH>
int launch_dh(const ContextAttnParams& p, hipStream_t stream)
{
    constexpr int HG = 256 / (DH / 8);
    bool mfma = false;
    if constexpr (DH == 64 || DH == 128)
    {
        if (p.workspace && p.seq >= 64 && p.inv_sqrt_dh > 0.f)
        {
            mfma = true;
            const int spad = (p.seq + 63) / 64 * 64;
            // RoPE + KV write (+ int8 quantisation) + the V^T image in one launch
            hipLaunchKernelGGL((rope_kv_vt_kernel<DH>), dim3(spad / 64, p.num_heads, p.batch), dim3(256), 0, stream, p, spad);
            {
                // 64-query workgroups while 128-query ones would not fill the chip
                bool narrow = (int64_t) ((p.seq + 127) / 128) * p.num_heads * p.batch < 256;
                constexpr size_t stages = 2 * (size_t) (64 * DH * 2 + DH * 128), slab = (size_t) (2 + 16 * (DH / 32)) * 64 * 4;
                // Paired 64-query blocks while the launch is at most ONE round of 128-query workgroups (32 heads, r02,
                // profiles/r02_ctx_attn_pairing.txt: S = 1024, 256 workgroups on 256 CUs, 31.9 -> 26.8 us; S = 896 / 768 / 512 / 384 /
                // 256, where the unpaired choice is the 64-query kernel, 28.0 -> 24.1, 24.5 -> 21.0, 18.5 -> 16.4, 15.8 -> 14.3,
                // 12.5 -> 12.0): with more workgroups than CUs the dispatcher already back-fills the CUs of the short blocks and
                // pairing only makes every workgroup long (S = 1536: 44 -> 58 us, S = 2048: 61.6 -> 72.4).
                const int cus = launch_util::device_cus();
                const int64_t wg128 = (int64_t) ((p.seq + 127) / 128) * p.num_heads * p.batch;
                const bool pair = wg128 <= cus && 4 * wg128 >= cus;
                if (pair)
                    narrow = false;
                // three operand stages only under the paired kernel (S = 1024: 28.1 -> 27.4 us; one block per workgroup loses with
                // them: 30.4 -> 31.2 at S = 1024, 61.6 -> 63.0 at 2048 - the operand stream is bound by its rate, not by the latency
                // of one block in flight; profiles/r02_ctx_attn_pairing.txt)
                const bool ring = pair;
                // (a FOURTH stage - three blocks in flight, 128 KB - measured 25.6 against 24.5 us at S = 1024 in r05: more bytes in
                //  flight do not raise the rate of the operand stream)
                const int nst = ring ? 3 : 2;
                const size_t stg = stages / 2 * nst;
                const size_t smem = stg > (narrow ? 2 : 4) * slab ? stg : 4 * slab;
                auto kfn = narrow ? context_attn_mfma_ks_kernel<DH, 2>
                                  : (pair ? context_attn_mfma_ks_kernel<DH, 4, true, 3> : context_attn_mfma_ks_kernel<DH, 4>);
                if (stages > 64 * 1024 || 4 * slab > 64 * 1024)
                    launch_util::ensure_dynamic_lds(reinterpret_cast<const void*>(kfn), 96 * 1024);
                const int qbw = narrow ? 64 : 128;
                // paired: ceil(nblk64 / 2) workgroups per head - the same count as 128-query blocks
                hipLaunchKernelGGL(kfn, dim3(p.num_heads, (p.seq + qbw - 1) / qbw, p.batch), dim3(narrow ? 256 : 512), smem, stream, p, spad);
            }
        }
    }
    if (!mfma)
    {
        hipLaunchKernelGGL((rope_kv_write_kernel<DH>), dim3(p.seq, (p.num_heads + HG - 1) / HG, p.batch), dim3(256), 0, stream, p);
        hipLaunchKernelGGL((context_attn_kernel<DH>), dim3((p.seq + 3) / 4, p.num_heads, p.batch), dim3(256), 0, stream, p);
        if (p.out_q8) // the quantiser as a pass of its own behind this kernel
        {
            const int64_t rows = p.cu_seqlens ? -1 : (int64_t) p.batch * p.seq;
            if (rows < 0)
            {
                set_error("context attention: the fused output quantiser needs padded inputs on this path");
                return -1;
            }
            if (launch_quantize_tensor(p.out_q8, p.out, DT_HALF, rows * p.num_heads * DH, p.out_q_scale, stream))
                return -1;
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
    {
        set_error("context attention launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}

} // namespace

size_t context_attention_workspace_size(int batch, int num_heads, int head_size, int seq)
{
    // V^T scratch of the MFMA path: [B, H, Dh, roundup(S, 64)] fp16
    return (size_t) batch * num_heads * head_size * ((seq + 63) / 64 * 64) * 2;
}

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
    if (p.block_pointers)
    {
        const int t = p.tokens_per_block;
        if (t < 1 || (t & (t - 1)) || (int64_t) p.max_blocks_per_seq * t < p.max_seq_len)
        {
            set_error("context attention: paged KV cache needs tokens_per_block a power of two and max_blocks_per_seq * "
                      "tokens_per_block >= max_seq_len (got %d x %d for %d)", p.max_blocks_per_seq, t, p.max_seq_len);
            return -1;
        }
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
