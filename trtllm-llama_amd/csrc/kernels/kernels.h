// Host-side launch interface of the gfx950 kernels.  Every launcher enqueues on `stream`, never
// synchronises, returns 0 on success or a negative code (and sets tllm::set_error) on a shape it
// cannot run.  SURVEY.md §8a row ids (A2, A5, ...) are cited per kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <stddef.h>
#include <stdint.h>

namespace tllm
{

void set_error(const char* fmt, ...);
const char* last_error();

namespace kernels
{

enum DType : int32_t
{
    DT_FLOAT = 0,
    DT_HALF = 1,
    DT_INT8 = 2,
    DT_INT32 = 3
};

// ---------------------------------------------------------------------------------------------
// Skinny GEMM (M <= 8 rows; decode).  y[m,n] = epi( scale * sum_k x'[m,k] * W[n,k] ),  x' = pro(x).
//   A7 (fp16 Gemm), A8 (weight-only int8/int4), A10 (SmoothQuant int8xint8->int32), with the
//   A5 RMSNorm / A11 activation quantisers as prologues and A6 SwiGLU / residual add as epilogues.
// HBM-bound: streams W once with 16-byte non-temporal loads, x' staged in LDS once per workgroup.
// ---------------------------------------------------------------------------------------------
enum WType : int32_t
{
    W_FP16 = 0,     // W fp16 [N, K] row-major
    W_INT8_WOQ = 1, // W u8 = q + 128, [N, ldw] (k contiguous), fp16 per-column scales; x fp16
    W_INT4_WOQ = 2, // W packed nibbles n = q + 8, see weight_layout.h; fp16 per-column scales; x fp16
    W_INT8_SQ = 3   // W s8 [N, ldw]; x s8; int32 accumulate; fp32 per-row x per-col scales
};

enum Prologue : int32_t
{
    PRO_NONE = 0,            // x already in the operand type (fp16, or s8 for W_INT8_SQ)
    PRO_RMSNORM = 1,         // x fp16 -> rmsnorm(x) * gamma (fp16)
    PRO_RMSNORM_QSTATIC = 2, // ... -> s8 with static scale act_scale[0]          (W_INT8_SQ)
    PRO_RMSNORM_QDYN = 3,    // ... -> s8 with per-token scale amax/127            (W_INT8_SQ)
    PRO_QSTATIC = 4,         // x fp16 -> s8 static                                (W_INT8_SQ)
    PRO_QDYN = 5             // x fp16 -> s8 per-token                             (W_INT8_SQ)
};

enum Epilogue : int32_t
{
    EPI_NONE = 0,
    EPI_RESIDUAL = 1,      // y = fp16(fp16(v) + residual)
    EPI_SWIGLU = 2,        // W rows [0,N) gate (mlp.fc), [N,2N) up (mlp.gate): y = fp16(silu(g) * u)
    EPI_SWIGLU_QSTATIC = 3 // ... then s8 with static scale epi_scale[0]
};

struct GemvParams
{
    int32_t wtype = W_FP16, pro = PRO_NONE, epi = EPI_NONE;
    int32_t out_dtype = DT_HALF; // DT_HALF | DT_FLOAT | DT_INT32 (SQ only) | DT_INT8 (EPI_SWIGLU_QSTATIC)
    int32_t M = 1, N = 0, K = 0; // N = output columns (for SWIGLU the weight has 2N rows)
    const void* x = nullptr;     // [M, ldx] fp16 or s8
    int64_t ldx = 0;             // elements
    const void* w = nullptr;
    int64_t ldw = 0;                  // bytes per weight row, multiple of 16
    const void* w_up = nullptr;       // SWIGLU: the "up" matrix (mlp.gate) [N, ldw]; null => rows [N, 2N) of w
    const void* scale_col = nullptr;  // WOQ: fp16 [N]; SQ: f32 [N] or [1]
    const void* scale_col_up = nullptr; // SWIGLU: scales of w_up; null => scale_col + N (per-channel) / scale_col
    const float* scale_row_up = nullptr; // SWIGLU + SQ static: act_scale of the up GEMM [1]; null => scale_row
    const float* scale_row = nullptr; // SQ: f32 [M] or [1] (ignored when the prologue quantises per token)
    int32_t per_channel = 0, per_token = 0;
    const void* gamma = nullptr; // fp16 [K]
    float eps = 1e-6f;
    const float* act_scale = nullptr; // PRO_*QSTATIC: f32 [1] (scale_to_int)
    float* dyn_scale_out = nullptr;   // optional f32 [M]: per-token scales computed by the prologue
    void* x_pro_out = nullptr;        // optional [M, K]: the prologue's result (fp16 or s8), written by workgroup 0
    const void* residual = nullptr;   // fp16 [M, ldy]
    const float* epi_scale = nullptr; // EPI_SWIGLU_QSTATIC: f32 [1]
    void* y = nullptr;
    int64_t ldy = 0; // elements
};

int launch_gemv(const GemvParams& p, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// RMSNorm (A5) with optional fused int8 quantisation (A11 pattern of layernormKernels.cu:162-183).
//   y = fp16( fp16(x * rsqrt(mean(x^2) + eps)) * gamma ); q = sat(rni(y * s)).
// ---------------------------------------------------------------------------------------------
struct RmsnormParams
{
    int32_t M = 0, N = 0;
    const void* x = nullptr;        // fp16 [M, N]
    const void* residual = nullptr; // optional fp16 [M, N]: x <- x + residual first, sum written to sum_out
    void* sum_out = nullptr;        // fp16 [M, N] (required when residual != nullptr)
    const void* gamma = nullptr;    // fp16 [N]
    float eps = 1e-6f;
    void* y = nullptr;                // fp16 [M, N] (may be null when only q is wanted)
    int8_t* q = nullptr;              // optional s8 [M, N]
    const float* static_scale = nullptr; // f32 [1] -> static quant
    float* dyn_scale_out = nullptr;      // f32 [M] -> per-token dynamic quant (amax / 127)
    // LayerNorm flavour (LayernormQuantization plugin, K/layernormKernels.cu:61-205): y = fp16((x - mean) * rstd * gamma + beta)
    int32_t layernorm = 0, use_diff_of_squares = 1;
    const void* beta = nullptr; // fp16 [N]
};
int launch_rmsnorm(const RmsnormParams& p, hipStream_t stream);

// A11: quantize_tensor (static per-tensor) and quantize_per_token.  K/quantization.cu:31-128.
int launch_quantize_tensor(int8_t* dst, const void* src, int32_t src_dtype, int64_t size, const float* scale,
    hipStream_t stream);
int launch_quantize_per_token(int8_t* dst, const void* src, int32_t src_dtype, int64_t rows, int64_t cols,
    float* scale_out, hipStream_t stream);

// A6 standalone SwiGLU: y = fp16(silu(a) * b), a,b fp16 [n].
int launch_swiglu(void* y, const void* a, const void* b, int64_t n, hipStream_t stream);
// the same, quantised on the way out: q = sat(rni(float(fp16(silu(a) * b)) * scale[0])) (static per-tensor SmoothQuant)
int launch_swiglu_quant(int8_t* q, const void* a, const void* b, int64_t n, const float* scale, hipStream_t stream);
// elementwise fp16 add: y = a + b
int launch_add(void* y, const void* a, const void* b, int64_t n, hipStream_t stream);
// A16 embedding gather: out[t, :] = table[ids[t], :] (fp16), ids int32; out-of-range id -> zeros.
int launch_embedding(void* out, const int32_t* ids, const void* table, int64_t tokens, int32_t hidden, int32_t vocab,
    hipStream_t stream);
// A16 gather_last_token_logits input: out[b,:] = hidden[b, last_token_ids[b]-1, :] (padded layout)
int launch_gather_last_token(void* out, const void* hidden, const int32_t* last_token_ids, int32_t batch, int32_t seq,
    int32_t hidden_size, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// Decode attention (A2/A3): masked multi-head attention for one new token per sequence, with RoPE,
// KV-cache write (fp16 or int8) and split-KV across workgroups.
// ---------------------------------------------------------------------------------------------
struct MmhaParams
{
    int32_t batch = 0, num_heads = 0, head_size = 0;
    int32_t rotary_dim = 0, neox = 1;
    float inv_sqrt_dh = 1.f;
    int32_t int8_kv = 0;
    int32_t max_seq_len = 0;   // cache capacity Smax
    int32_t max_input_len = 0; // padded prompt length (RoPE position = timestep - (max_input_len - input_len[b]))
    const void* qkv = nullptr; // fp16 [B, 3*H*Dh] (q | k | v)
    void* kv_cache = nullptr;  // [B, 2, H, Smax, Dh] fp16 or s8
    const int32_t* sequence_length = nullptr; // [B] device: tlength (slots already used)
    const int32_t* input_lengths = nullptr;   // [B] device
    const int32_t* masked_tokens = nullptr;   // [B, Smax] device or null
    int32_t timestep_host = -1;               // past_kv_len from the host scalar; -1 => use sequence_length[0]
    const float* kv_scale_orig_quant = nullptr; // f32 [1]
    const float* kv_scale_quant_orig = nullptr; // f32 [1]
    const float* rope_table = nullptr;          // f32 [max_pos, rotary_dim/2, 2] (cos, sin)
    int32_t rope_table_len = 0;
    const float* rope_row = nullptr; // optional f32 [B, rotary_dim/2, 2]: this step's cos/sin row, prepared on the
                                     // device by the sampler (removes the length -> position -> table dependency)
    int32_t rows_per_group = 0;      // cache rows per lane group and split: 4 (finest split), 12 or 16; 0 = chosen by the launcher
    // beam search (MM/...Template.h:1137-1146, :1624-1631): `batch` counts batch x beam sequences; sequence bb = b * beam_width
    // + k reads the K/V of timestep t from the cache rows of sequence b * beam_width + cache_indirection[bb, t]
    const int32_t* cache_indirection = nullptr; // int32 [batch, max_seq_len]
    int32_t beam_width = 1;
    // paged KV cache (K/kvCacheUtils.h:34-112 KVBlockArray): instead of kv_cache, a table int64 [batch, 2, max_blocks_per_seq]
    // of device pointers to blocks laid out [H, tokens_per_block, Dh]; time step t is row t % tokens_per_block of block
    // t / tokens_per_block (tokens_per_block a power of two)
    const int64_t* block_pointers = nullptr;
    int32_t tokens_per_block = 0, max_blocks_per_seq = 0;
    // The split merge: when set (uint32 [B * H], zero before the first launch, self-resetting), the LAST split of a (sequence,
    // head) to arrive merges all partials (at most 16) inside this launch and writes the normalised context to `out` (and, with
    // tail_quant_scale, its static int8 image to tail_out_q8); null: a combine launch follows the partial launch.
    uint32_t* tail_tickets = nullptr;
    const float* tail_quant_scale = nullptr; // f32 [1]: SmoothQuant static activation scale of the O-projection's input
    void* tail_out_q8 = nullptr;             // s8 [B, H*Dh]
    void* out = nullptr;       // fp16 [B, H*Dh]
    void* workspace = nullptr; // mmha_workspace_size bytes
};
// Split geometry for a given rows_per_group: timesteps per split, number of splits for max_seq_len, and the byte
// offset of the partial outputs inside the workspace ({max,sum} float2 [B,H,ns] first, then out f32 [B,H,ns,Dh]).
int mmha_split_layout(int32_t head_size, int32_t max_seq_len, int32_t rows_per_group, int32_t batch, int32_t num_heads,
    int32_t* tchunk, int32_t* nsplit, size_t* out_offset);
size_t mmha_workspace_size(int32_t batch, int32_t num_heads, int32_t head_size, int32_t max_seq_len);
size_t mmha_ticket_bytes(int32_t batch, int32_t num_heads); // the merge tickets at the head of a plugin's workspace
int launch_mmha(const MmhaParams& p, hipStream_t stream);
// The split-merge tickets at the head of the workspace must be zero before the FIRST launch on it (every launch
// re-arms them): plugins call this per enqueue, the session once per setup.
int mmha_reset_workspace(void* workspace, int32_t batch, int32_t num_heads, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// Decode step, batch 1: the QKV projection (SmoothQuant int8, RMSNorm + quantiser prologue) of head h, RoPE, the cache append
// and the masked attention of head h in ONE launch (kernels/qkv_attn_fused.hip).  8 workgroups per head; the projection outputs
// and the split partials travel between them as 8-byte {tag, value} granules through `xchg`.
// ---------------------------------------------------------------------------------------------
struct FusedQkvAttnParams
{
    int32_t K = 0, num_heads = 0, head_size = 0;
    const void* x = nullptr;     // fp16 [K]: the layer's input row
    const void* gamma = nullptr; // fp16 [K]: input_layernorm
    float eps = 1e-6f;
    const void* w = nullptr; // s8 [3 * num_heads * head_size, ldw]: q | k | v rows
    int64_t ldw = 0;         // bytes
    const void* scale_col = nullptr; // SmoothQuant: f32 [3 * H * Dh] (per_channel) or [1]; weight-only: fp16 [3 * H * Dh]
    int32_t per_channel = 0;
    int32_t woq8 = 0; // 1: weight-only int8 rows (u8 = q + 128) against the normalised fp16 row; no quantiser
    int32_t woq4 = 0;   // 1 (r06): weight-only int4 rows (ldw = K / 2 bytes, fp16 scales) against the normalised fp16 row; two-stage form
    int32_t fp16_w = 0; // 1 (r06): fp16 rows (ldw = 2 K bytes, no scales) against the normalised fp16 row; no quantiser, no O-projection stage
    const float* act_quant_scale = nullptr;   // f32 [1]: static quantiser of the normalised row; null -> per-token (amax / 127)
    const float* act_dequant_scale = nullptr; // f32 [1]: the static activation scale of the dequantisation
    int32_t int8_kv = 0, max_seq_len = 0;
    float inv_sqrt_dh = 1.f;
    void* kv_cache = nullptr;                 // [2, H, Smax, Dh] fp16 or s8 (batch 1)
    const int32_t* sequence_length = nullptr; // [1] device: slots in use
    const int32_t* masked_tokens = nullptr;   // [Smax] device or null
    const float* kv_scale_orig_quant = nullptr;
    const float* kv_scale_quant_orig = nullptr;
    const float* rope_row = nullptr; // f32 [Dh / 2, 2]: this step's (cos, sin) row (NeoX pairs), prepared by the sampler
    uint64_t* xchg = nullptr;        // qkv_attn_fused_xchg_bytes(), zero before the first launch
    const uint32_t* epoch = nullptr; // device word advanced once per generation step
    uint32_t tag_mul = 1, tag_add = 1; // tag of this launch = epoch * tag_mul + tag_add (never 0, unique per launch)
    uint32_t tag_host = 0;             // non-zero: the tag itself (eager launches outside a step, e.g. kernel timing)
    uint32_t* error = nullptr;         // device word: non-zero after a bounded wait expired (a launch that finds it set returns at once)
    // polls (s_sleep + one L2 round trip, ~1 us each) a wait may take: ~50 ms against hand-offs of 1 - 5 us.  Generous enough for a
    // grid that is delayed by another queue's kernels, short enough that ONE expired wait - later launches return at entry - costs a
    // request less than a second
    int32_t max_spins = 50000;
    void* qkv_out = nullptr; // optional fp16 [3 * H * Dh]: the projection's output (before RoPE)
    void* out = nullptr;     // fp16 [H * Dh]: the attention context
    void* out_q8 = nullptr;  // optional s8 [H * Dh] = sat(rni(float(out) * out_quant_scale[0])) (the O-projection's static quantiser)
    const float* out_quant_scale = nullptr;
    void* x_pro_out = nullptr; // optional s8 [K]: the quantised operand (tap)
    uint64_t* timing = nullptr; // optional [workgroups][16] stage clock (100 MHz ticks): tools/fused_timeline.py
    // optional third stage (r05): the O-projection + residual of the same layer in the same launch,  x_out[n] = x[n] + O(ctx)[n]
    // (needs out_q8: the context row reaches the row workers as its static int8 image).  o_w null: the launch ends with the context.
    const void* o_w = nullptr;             // s8 [o_n, o_ldw]: rows of the dense projection, K = num_heads * head_size
    int64_t o_ldw = 0;                     // bytes
    int32_t o_n = 0, o_per_channel = 0;
    const void* o_scale_col = nullptr;     // SmoothQuant: f32 [o_n] (per_channel) or [1]; weight-only: fp16 [o_n]
    const float* o_scale_row = nullptr;    // f32 [1]: the static activation scale of the dequantisation
    void* x_out = nullptr;                 // fp16 [o_n]: may be x itself (every workgroup has consumed x long before)
};
size_t qkv_attn_fused_xchg_bytes(int32_t num_heads);
bool qkv_attn_fused_serves_o(int32_t num_heads, int32_t head_size, int32_t o_n, int32_t o_k, int64_t o_ldw, int32_t weight_kind = 0);
// (woq8 / o_stage decide the instance and its dynamic LDS: the residency check - occupancy query x CUs of the current device >= grid
// - is made for exactly the launch that will run)
// weight_kind: 0 SmoothQuant int8, 1 weight-only int8, 2 fp16, 3 weight-only int4
bool qkv_attn_fused_serves(int32_t K, int32_t num_heads, int32_t head_size, int32_t max_seq_len, int32_t int8_kv, int32_t weight_kind = 0,
    int32_t o_stage = 0);
int launch_qkv_attn_fused(const FusedQkvAttnParams& p, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// Decode step, batch 1, SmoothQuant static: RMSNorm + gate|up + SwiGLU + quantiser + down-projection + residual in ONE launch
// (kernels/mlp_fused.hip, r06): 256 workgroups, the intermediate row handed over inside the launch (write-through bytes, a flag
// byte per workgroup polled through the scalar path), the down-projection's rows requested ahead of it.
// ---------------------------------------------------------------------------------------------
struct FusedMlpParams
{
    int32_t K = 0, I = 0, N = 0; // hidden size (gate|up K), intermediate size (gate|up rows, down K), down rows (= K for LLaMA)
    const void* x = nullptr;     // fp16 [K]: the layer's residual stream behind the attention block (also the residual of the output)
    void* x_out = nullptr;       // fp16 [N]: may be x itself (every workgroup has consumed x before any workgroup writes)
    const void* gamma = nullptr; // fp16 [K]: post_layernorm
    float eps = 1e-6f;
    const float* act_quant = nullptr; // f32 [1]: static quantiser of the normalised row
    const void* w_fc = nullptr;       // s8 [I, ldw]
    const void* w_gate = nullptr;     // s8 [I, ldw]
    int64_t ldw = 0;
    const void* scale_fc = nullptr;   // f32 [I] (per_channel) or [1]
    const void* scale_gate = nullptr;
    int32_t per_channel = 0;
    const float* row_fc = nullptr;    // f32 [1]: activation scale of the dequantisation (fc); row_gate null -> the same
    const float* row_gate = nullptr;
    const float* out_quant = nullptr; // f32 [1]: static quantiser of silu(fc) * gate
    void* inter = nullptr;            // optional s8 [I]: the intermediate row as the two-launch form leaves it (taps)
    const void* w_proj = nullptr;     // s8 [N, ldw_proj]
    int64_t ldw_proj = 0;
    const void* scale_proj = nullptr; // f32 [N] (per_channel_proj) or [1]
    int32_t per_channel_proj = 0;
    const float* row_proj = nullptr;  // f32 [1]
    uint8_t* flags = nullptr;         // the exchange area (a 64-byte line per workgroup: its int8 outputs + the tag; a line per group
                                      // of 16): mlp_fused_flag_bytes(), zero before the first launch
    uint32_t* error = nullptr;        // device word: bit 16 after a bounded wait expired (a launch that finds it set returns at once)
    int32_t max_spins = 50000;
    void* x_pro_out = nullptr;  // optional s8 [K]: the quantised operand (tap)
    uint64_t* timing = nullptr; // optional [256][16] stage clock
};
size_t mlp_fused_flag_bytes();
bool mlp_fused_serves(int32_t K, int32_t I, int32_t N);
int launch_mlp_fused(const FusedMlpParams& p, hipStream_t stream);

// RoPE table builder (host -> device buffer owned by caller): cos/sin(pos / 10000^(2j/rot)) in fp32,
// formula of K/decoderMaskedMultiheadAttentionUtils.h:1511-1515.
void fill_rope_table_host(float* table, int32_t max_pos, int32_t rotary_dim);

// ---------------------------------------------------------------------------------------------
// Context attention (A4): RoPE + KV-cache write + causal/padding-masked softmax(QK^T)V for the prompt.
// ---------------------------------------------------------------------------------------------
struct ContextAttnParams
{
    int32_t batch = 0, seq = 0, num_heads = 0, head_size = 0;
    int32_t rotary_dim = 0, neox = 1;
    float inv_sqrt_dh = 1.f;
    int32_t int8_kv = 0, max_seq_len = 0;
    void* qkv = nullptr;      // fp16 [B, S, 3*H*Dh]; q,k rewritten in place with RoPE applied (reference :1401-1403)
    void* kv_cache = nullptr; // [B, 2, H, Smax, Dh]
    const int32_t* input_lengths = nullptr; // [B]
    const float* kv_scale_orig_quant = nullptr;
    const float* rope_table = nullptr;
    int32_t rope_table_len = 0;
    void* out = nullptr; // fp16 [B, S, H*Dh]; rows >= input_len[b] are zero
    // optional static quantiser behind the attention (SmoothQuant's O-projection input, K/quantization.cu): when set the result
    // goes to out_q8 = sat(rni(float(fp16(ctx)) * out_q_scale[0])), int8 [B, S, H*Dh], and the MFMA kernel does not write `out`
    // at all (the other kernels write `out` and a quantiser pass follows - same values either way)
    int8_t* out_q8 = nullptr;
    const float* out_q_scale = nullptr;
    void* workspace = nullptr; // context_attention_workspace_size() bytes; NULL -> the (slow) wave-per-query kernel
    // packed inputs (remove_input_padding, P/gptAttentionPlugin/gptAttentionPlugin.cpp:344-356): qkv / out hold only the real
    // tokens, [sum(len), ...]; token s of sequence b is row cu_seqlens[b] + s.  `seq` stays the longest sequence (grid bound).
    const int32_t* cu_seqlens = nullptr; // device int32 [B + 1], exclusive prefix sum of input_lengths
    // beam search: prompt b fills the cache rows of sequence b * cache_seq_stride (hypothesis 0 of its beam group; the
    // siblings reach those rows through the cache indirection, so the prompt's K/V is stored once)
    int32_t cache_seq_stride = 1;
    // paged KV cache: see MmhaParams (kv_cache unused when set)
    const int64_t* block_pointers = nullptr;
    int32_t tokens_per_block = 0, max_blocks_per_seq = 0;
};
size_t context_attention_workspace_size(int batch, int num_heads, int head_size, int seq);
// cu[0] = 0, cu[b + 1] = cu[b] + lens[b]  (device, one tiny launch; packed-input bookkeeping)
int launch_exclusive_scan_i32(int32_t* cu, const int32_t* lens, int32_t n, hipStream_t stream);
// out[b, :] = hidden[rows[b], :]  (fp16 rows; the packed-input flavour of gather_last_token)
int launch_gather_rows(void* out, const void* hidden, const int32_t* rows, int32_t batch, int32_t hidden_size, hipStream_t stream);
int launch_context_attention(const ContextAttnParams& p, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// MFMA GEMMs for M > 8 (prefill): C[m,n] = sum_k A[m,k] * B[n,k].
// ---------------------------------------------------------------------------------------------
struct GemmParams
{
    int32_t wtype = W_FP16; // same weight layouts as GemvParams
    int32_t out_dtype = DT_HALF;
    int32_t M = 0, N = 0, K = 0;
    const void* a = nullptr; // fp16 [M, lda] or s8 [M, lda]
    int64_t lda = 0;
    const void* w = nullptr;
    int64_t ldw = 0; // bytes
    const void* scale_col = nullptr;
    const float* scale_row = nullptr;
    int32_t per_channel = 0, per_token = 0;
    void* c = nullptr;
    int64_t ldc = 0;
    const void* residual = nullptr; // optional fp16 [M, ldc]: C = fp16(fp16(gemm) + residual) (may alias c); fp16 output only
    // optional fp16 [M, ldc] (fp16 / weight-only weights, fp16 output, not together with residual): C = fp16(fp16(silu(gate)) * fp16(gemm)) -
    // the SwiGLU pass of the prefill MLP folded into the second projection (may alias c)
    const void* silu_gate = nullptr;
    // SmoothQuant dual GEMM (prefill MLP, static activation scales): w2 / scale_col2 = the second weight matrix [N, ldw] and
    // its per-channel scales; c = int8 [M, ldc] = sat(rni(fp16(silu16(A W^T) * fp16(A W2^T)) * swiglu_qscale[0])) with the
    // fp16 rounding points of the un-fused path (two fp16 GEMM outputs, SwiGLU in fp16, static quantiser); launch_gemm_swiglu
    const void* w2 = nullptr;
    const void* scale_col2 = nullptr;
    const float* scale_row2 = nullptr; // static dequantisation scale of the second GEMM [1] (null: scale_row)
    const float* swiglu_qscale = nullptr;
    // microbench hook (tllm_gemm_set_clock_probe), set by the launchers only: 2 x uint64 per workgroup {shader cycles, 100 MHz ticks}
    void* clock_probe = nullptr;
    // split-K-2 forms of gemm_sqp.hip (r06), set by the launcher only: per tile two slabs of AH x BN accumulators + two flag words
    void* ksplit_ws = nullptr;
};
int launch_gemm(const GemmParams& p, hipStream_t stream);
// fc and gate projections of the SmoothQuant MLP in one kernel with SwiGLU + static int8 quantisation in its epilogue
// (gemm_sqp.hip); returns 1 when the problem is not served (caller runs the two GEMMs + launch_swiglu_quant instead)
int launch_gemm_swiglu(const GemmParams& p, hipStream_t stream);
// On-device tactic selection for the prefill GEMMs (gemm_tactics.hip; reference: int8_gemm_template.h:372-457 profileGemm +
// smoothQuantGemmPlugin.cpp:253-282 mMNKProfileMap).  lookup: kernel id for the shape, 0 = nothing profiled (static rule).
int gemm_tactic_lookup(int wtype, int M, int N, int K);
bool gemm_tactic_known(int wtype, int M, int N, int K);
int gemm_profile(int wtype, int M, int N, int K, int* best_cfg, float* best_us, hipStream_t stream);
void gemm_tactics_clear();
int gemm_tactics_import(const char* text); // entries taken, -1 on a parse error
std::string gemm_tactics_export();       // "wtype:M:N:K:cfg:us;..."

// One-shot peer-to-peer sum all-reduce of an fp16 vector, in place (kernels/p2p_allreduce.hip; plugins/p2p.cpp owns the
// inboxes).  peer[r]: base of rank r's region as mapped in THIS process: [2 gens][world][slot_bytes] data, then flags.
struct P2PParams
{
    void* peer[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int32_t world = 0, rank = 0;
    size_t slot_bytes = 0, flag_offset = 0;
    void* x = nullptr; // fp16, n16 * 8 elements, 16-byte aligned (sum: in place; gather: this rank's contribution)
    void* gather_out = nullptr; // non-null: all-gather instead of sum, out = [world][n16 * 16 bytes]
    int32_t n16 = 0;
    uint32_t* epoch = nullptr; // device counter, advanced by the kernel
    uint32_t* error = nullptr; // device flag: non-zero after a spin timed out (bit 31: a PEER reported the time-out)
    int32_t max_spins = 4000000;
    // optional fused tail of the sum (never with gather_out):
    //   residual != null           : the sum starts from residual[v]                       (hidden = residual + all_reduce(...))
    //   x_out    != null           : the fp16 result goes there instead of back into x     (may alias residual)
    //   norm_out != null           : + RMSNorm over rows of `cols` elements: norm_out = fp16(fp16(x_out * inv) * gamma), or its
    //                                int8 quantisation (quant 1: static scale quant_scale[0]; 2: per token, scales to dyn_scale_out)
    const void* residual = nullptr;
    void* x_out = nullptr;
    void* norm_out = nullptr;
    const void* gamma = nullptr;
    float eps = 1e-6f;
    int32_t rows = 0, cols = 0, quant = 0;
    const float* quant_scale = nullptr;
    float* dyn_scale_out = nullptr;
};
constexpr size_t P2P_POISON_OFFSET = 2048; // byte offset behind flag_offset of the "a peer gave up" word in every region
int launch_p2p_allreduce(const P2PParams& p, hipStream_t stream);

// Greedy sampler: argmax over fp32 logits [B, V] -> ids; ties -> lowest index.
int launch_argmax(int32_t* out_ids, const float* logits, int32_t batch, int32_t vocab, hipStream_t stream);

// One greedy sampling step with the bookkeeping of GenerationSession.decode kept on the device
// (PY/runtime/generation.py:943-983: DynamicDecodeOp top-k=1 + sequence-length update + stop criteria):
//   id = argmax_v logits[b, v]   (logits may be the all-gather of `nparts` vocabulary shards [nparts, B, vocab_part])
//   if advance: seq_len[b] += 1 ;  finished sequences keep emitting end_id ;  out_ids[b, seq_len[b]] = id ;
//   cur_ids[b] = id ;  finished[b] |= (id == end_id)
struct GreedyParams
{
    const float* logits = nullptr;
    int32_t batch = 0, vocab_part = 0, nparts = 1, vocab = 0;
    int32_t* cur_ids = nullptr;
    int32_t* out_ids = nullptr;
    int32_t out_stride = 0;
    int32_t* seq_len = nullptr;
    int32_t* finished = nullptr;
    int32_t end_id = -1, advance = 0;
    // optional: prepare the RoPE cos/sin row of the NEXT generation step, position = seq_len[b] - (max_input_len -
    // input_lengths[b])  (MM/...Template.h:1425-1426), so that the attention kernel does not chase
    // length -> position -> table
    float* rope_row_out = nullptr;     // f32 [B, rope_half, 2]
    int32_t* rope_pos_out = nullptr;   // optional int32 [B]: the position that row was taken at (parity tests read it back)
    const float* rope_table = nullptr; // f32 [rope_table_len, rope_half, 2]
    int32_t rope_half = 0, rope_table_len = 0;
    const int32_t* input_lengths = nullptr;
    int32_t max_input_len = 0;
    // optional: gather the chosen token's embedding row into the next step's input (fp16 [B, hidden]) - one launch and one
    // dependent round trip less per generation step than a separate embedding kernel
    const void* emb_table = nullptr; // fp16 [vocab, hidden]
    void* x_out = nullptr;
    int32_t hidden = 0;
    uint32_t* step_epoch = nullptr; // optional device word, += 1 per call (tags of the in-launch hand-offs of the next step)
};
int launch_greedy_step(const GreedyParams& p, hipStream_t stream);
// teacher forcing for parity tests: overwrite the sampler's last choice (output slot seq_len[b], step input id, next input row)
int launch_force_token(const int32_t* ids_dev, int32_t* cur_ids, int32_t* out_ids, int32_t out_stride, const int32_t* seq_len,
    const void* emb, void* x, int32_t batch, int32_t hidden, int32_t vocab, hipStream_t stream);

// One beam-search step on the device (K/onlineSoftmaxBeamsearchKernels.cu, layers/onlineBeamSearchLayer.cu; called at
// PY/runtime/generation.py:949-961 with beam_width > 1).  Per batch entry, over its `beam` hypotheses:
//   score(k, v) = cum_log_probs[k] + log_softmax(logits[k])[v]      (a finished hypothesis only continues with end_id, score kept)
//   the `beam` best (k, v) in score order (ties: lowest k * vocab + v) become the new hypotheses j: token v, parent k;
//   cache_indirection[j, s] <- cache_indirection[parent, s] for every used slot s, and the slot the consumed token's K/V
//   went to <- parent;  out_ids / parent_ids[j, slot] record the step for the final back-track (gather_tree).
// Bookkeeping (sequence length, current ids, next RoPE row) as launch_greedy_step.
struct BeamParams
{
    const float* logits = nullptr; // [nparts, batch * beam, vocab_part]  ([nparts, batch, vocab_part] if logits_per_batch)
    int32_t logits_per_batch = 0;  // first step after the prompt: one logits row per batch entry, shared by its hypotheses
    int32_t batch = 0, beam = 1, vocab_part = 0, nparts = 1, vocab = 0;
    float* cum_log_probs = nullptr; // [batch * beam]
    int32_t* cur_ids = nullptr;
    int32_t* out_ids = nullptr;    // [batch * beam, out_stride]
    int32_t* parent_ids = nullptr; // [batch * beam, out_stride]
    int32_t out_stride = 0;
    int32_t* seq_len = nullptr;
    int32_t* finished = nullptr;
    int32_t end_id = -1, advance = 0;
    int32_t* cache_indirection = nullptr; // [batch * beam, out_stride]
    float* rope_row_out = nullptr;
    int32_t* rope_pos_out = nullptr; // as GreedyParams
    const float* rope_table = nullptr;
    int32_t rope_half = 0, rope_table_len = 0;
    const int32_t* input_lengths = nullptr;
    int32_t max_input_len = 0;
};
int launch_beam_step(const BeamParams& p, hipStream_t stream);

// deterministic pseudo-random fill (fp16 ~U(-scale, scale) or int8 ~U[-127,127]) for synthetic KV caches
int launch_fill_random(void* dst, int32_t dtype, int64_t n, uint32_t seed, float scale, hipStream_t stream);
// int32 fill
int launch_fill_i32(int32_t* dst, int32_t value, int64_t n, hipStream_t stream);

} // namespace kernels
} // namespace tllm
