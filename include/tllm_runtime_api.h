/*
 * tllm_runtime_api.h — host decode loop of the MI355X LLaMA path behind a C ABI.
 *
 * Takes the place of the TensorRT engine + execution context that the reference's runtime drives
 * (T/tensorrt_llm/runtime/generation.py:43-100 `_Runtime`, :413-488 `setup`, :782-997 `decode`) and of the
 * C++ GptSession host loop (T/cpp/tensorrt_llm/runtime/gptSession.cpp:252,700).  The Python front-end
 * (tensorrt_llm.runtime.GenerationSession) keeps its reference signature and calls these entry points;
 * PyTorch is only used to own device buffers that are handed in as raw pointers.
 *
 * An "engine" here is: a text configuration (key=value lines, the builder_config / plugin_config of
 * config.json, T/tensorrt_llm/builder.py:259-267) plus named weight tensors (module paths of the reference's
 * LLaMAForCausalLM, T/examples/llama_quant/llama_model.py:122-251).  Execution = the plugin kernels of
 * tllm_plugin_api.h enqueued layer by layer on one HIP stream; the generation step is fused
 * (RMSNorm / quantisation / SwiGLU / residual folded into the GEMV launches) and replayed as a hipGraph.
 */
#ifndef TLLM_RUNTIME_API_H
#define TLLM_RUNTIME_API_H

#include "tllm_plugin_api.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tllm_session* tllm_session_t;

/* Create a session from a configuration text.  Recognised keys (defaults in parentheses):
 *   num_layers, num_heads, hidden_size, inter_size, vocab_size, max_position_embeddings (2048),
 *   rms_norm_eps (1e-6), tp_size (1), tp_rank (0), quant_mode (0: QuantMode bits,
 *   T/tensorrt_llm/quantization/mode.py:6-21), weight_only_precision (int8|int4),
 *   use_gpt_attention_plugin/use_gemm_plugin (informational), neox_rotary_style (1),
 *   force_comm (0; tests: issue the tensor-parallel collectives on a 1-rank communicator as well),
 *   remove_input_padding (0; 1: the context phase runs on the packed real tokens only),
 *   paged_kv_cache (0; 1: every layer's cache is a pool of [Hr, tokens_per_block, Dh] blocks reached through a per-sequence
 *   table of block pointers - K/kvCacheUtils.h:34-112, PY/runtime/kv_cache_manager.py; the table is filled at setup),
 *   tokens_per_block (64; a power of two),
 *   debug_taps (0; 1: keep per-layer intermediates of the generation step for tllm_session_get_tap),
 *   gemm_tactics (optional; the table tllm_gemm_tactics_export wrote, as the builder stores it in the engine).
 * Returns NULL on error (tllm_last_error()). */
tllm_session_t tllm_session_create(const char* config_text);

/* Register one weight tensor by module path, e.g. "layers.3.attention.qkv.weight", "vocab_embedding.weight",
 * "ln_f.weight", "lm_head.weight" (reference attribute names, T/examples/llama_quant/weight_quant.py:172-446).
 *   location 0: `data` is a HOST pointer, the bytes are copied to device memory owned by the session;
 *   location 1: `data` is a DEVICE pointer that outlives the session (tensor handoff from torch). */
int32_t tllm_session_set_tensor(tllm_session_t s, const char* name, int32_t dtype, const int64_t* dims, int32_t nbDims,
    const void* data, int32_t location);

/* Resolve the weights against the configuration; fails listing the first missing / mis-shaped tensor. */
int32_t tllm_session_finalize(tllm_session_t s);

/* Parse a serialized engine (format written by tensorrt_llm.Builder.build_engine: "TLLMENG1" header, config
 * text, tensor table, data) = tllm_engine_verify + create + set_tensor(location 0) + finalize. */
tllm_session_t tllm_session_load_engine(const void* engine, size_t nbytes);

/* The check tllm_session_load_engine applies before it accepts an engine, on its own (host only, no GPU needed): the traced
 * network the engine carries (`network_json`: plugin nodes in order, their inputs, fields and the weights feeding them, the
 * I/O tensor names of T/tensorrt_llm/runtime/generation.py:188-208) must be exactly the LLaMA schedule the session executes
 * for the engine's configuration - "the engine is what was defined" (T/tensorrt_llm/builder.py:259-267).  A model
 * definition that was edited (another plugin, another field value, another weight on a port, a node more or less) is
 * refused with an error naming the first node that differs.  0 = accepted. */
int32_t tllm_engine_verify(const void* engine, size_t nbytes);

/* GenerationSession.setup (generation.py:413-488): allocates the per-layer KV cache
 * [B, 2, H/tp, max_input_len + max_new_tokens, Dh] (fp16 or int8), activations and the RoPE table. */
int32_t tllm_session_setup(tllm_session_t s, int32_t batch_size, int32_t max_input_len, int32_t max_new_tokens);

/* The same with beam search (SamplingConfig.num_beams > 1, generation.py:365-411, 823-866): batch_size prompts, beam_width
 * hypotheses each (1 <= beam_width <= 8; batch_size * beam_width is not limited: the generation GEMVs take 8 rows per launch and
 * run more sequences in slabs of 8, each slab streaming the weights again).  The KV cache holds
 * batch_size * beam_width sequences; the prompt's K/V is stored once per batch entry (in hypothesis 0's rows) and reached by
 * the others through the cache indirection [batch_size * beam_width, max_seq_len] (value = sibling hypothesis whose rows
 * hold that time step), which the device-side beam step re-parents every step - the reference's src/tgt
 * cache_indirection pair (gptAttention plugin input, decoderMaskedMultiheadAttentionTemplate.h:1137-1146) in one buffer.
 * Selection rule of the reference's dynamic decoder without beam_hyps (onlineSoftmaxBeamsearchKernels.cu): the
 * beam_width best cum_log_prob + log_softmax(logits) over (hypothesis, token); a finished hypothesis continues with
 * end_id at unchanged score.  tllm_session_generate then returns [batch_size, beam_width, max_seq_len]. */
int32_t tllm_session_setup_beam(tllm_session_t s, int32_t batch_size, int32_t beam_width, int32_t max_input_len,
    int32_t max_new_tokens);

/* gather_tree (decodingKernels.cu:30-171; generation.py:990-994): ids HOST int32 [batch_size, beam_width, max_seq_len], the
 * back-tracked hypotheses, best first; cum_log_probs HOST float [batch_size, beam_width] or NULL.  After the first end_id
 * and beyond the current length the row is filled with end_id.  With beam_width 1 it is tllm_session_get_output_ids. */
int32_t tllm_session_get_beam_output(tllm_session_t s, int32_t* ids, float* cum_log_probs, tllm_stream_t stream);

/* The device-side beam bookkeeping, for tests and debugging (any pointer may be NULL): parent_ids and cache_indirection
 * HOST int32 [batch_size * beam_width, max_seq_len] (generation.py:387-391, 823-837), finished and sequence_lengths
 * HOST int32 [batch_size * beam_width]. */
int32_t tllm_session_get_beam_state(tllm_session_t s, int32_t* parent_ids, int32_t* cache_indirection, int32_t* finished,
    int32_t* sequence_lengths, tllm_stream_t stream);

/* Rows the last logits hold (tllm_session_get_logits): batch_size after the prompt, batch_size * beam_width after a step. */
int32_t tllm_session_logit_rows(tllm_session_t s);

/* Vocabulary size of the model the session holds (the width of a logits row; the reference reads it from the engine's `logits`
 * binding, runtime/session.py:116-145 infer_shapes). */
int32_t tllm_session_vocab_size(tllm_session_t s);

/* GenerationSession.decode (generation.py:782-997), greedy (top-k = 1): context step on the padded prompts,
 * then up to max_new_tokens generation steps with the sampler on device and no per-step host sync
 * (the reference syncs every step, generation.py:963).
 *   input_ids     HOST int32 [B, max_input_len] (padded with pad_id), input_lengths HOST int32 [B];
 *   output_ids    HOST int32 [B, max_input_len + max_new_tokens]: the prompt followed by the generated ids;
 *   end_id < 0 disables early stopping (benchmarks).  Blocks until the ids are on the host. */
int32_t tllm_session_generate(tllm_session_t s, const int32_t* input_ids, const int32_t* input_lengths,
    int32_t max_new_tokens, int32_t end_id, int32_t pad_id, int32_t* output_ids, tllm_stream_t stream);

/* Step-wise interface (parity tests, benchmarks):
 *   context: runs the prompt, leaves the fp32 logits of the last real token of every sequence in the session;
 *   step:    runs `n_steps` generation steps (greedy feedback on device); timing is the caller's business.
 *   use_graph != 0 replays the captured generation step. */
int32_t tllm_session_context(tllm_session_t s, const int32_t* input_ids, const int32_t* input_lengths,
    tllm_stream_t stream);
int32_t tllm_session_step(tllm_session_t s, int32_t n_steps, int32_t use_graph, tllm_stream_t stream);
/* Fill the KV cache for `length` positions with pseudo-random content (as a prompt of that length would)
 * without running a prefill; for decode-rate benchmarks at a given context length. */
int32_t tllm_session_fake_context(tllm_session_t s, int32_t length, uint32_t seed, tllm_stream_t stream);
/* Copy state to HOST buffers (synchronises the stream). */
int32_t tllm_session_get_logits(tllm_session_t s, float* logits /* [B, vocab] */, tllm_stream_t stream);
int32_t tllm_session_get_output_ids(tllm_session_t s, int32_t* ids /* [B, max_in + max_new] */, tllm_stream_t stream);
/* Device pointer of a layer's KV cache (layout [B,2,H/tp,Smax,Dh]) for inspection. */
void* tllm_session_kv_cache_ptr(tllm_session_t s, int32_t layer);
/* Diagnostic: with the session key fused_timeline = 1, the device buffer [heads * 8 workgroups][16] of 100 MHz clock ticks the
 * last fused QKV-projection + attention launch stamped at its stages (kernels/qkv_attn_fused.hip; tools/fused_timeline.py);
 * NULL when the session does not run that launch. */
void* tllm_session_fused_timeline_ptr(tllm_session_t s);
/* ... and of the last one-launch MLP ([256 workgroups][16]; kernels/mlp_fused.hip). */
void* tllm_session_mlp_timeline_ptr(tllm_session_t s);
/* The step-dependent tensors the reference's Python loop builds on the host every step (PY/runtime/generation.py:556-579,
 * :686-689, :735-750, :812-821) live in device memory here and are advanced by the sampler kernel; this copies them out so
 * that tests can pin them to the reference's values (any pointer may be NULL):
 *   sequence_length [B]        cache slots in use = the reference's `sequence_length` / `past_key_value_length[0]` input of
 *                              the NEXT generation step (max_input_len + step);
 *   next_position   [B]        rotary position of the token the next step consumes = the reference's `position_ids` of that
 *                              step (input_lengths + step);
 *   masked_tokens   [B, Smax]  1 on the padding slots [input_lengths[b], max_input_len);
 *   input_lengths   [B]. */
int32_t tllm_session_get_step_state(tllm_session_t s, int32_t* sequence_length, int32_t* next_position, int32_t* masked_tokens,
    int32_t* input_lengths, tllm_stream_t stream);

/* Teacher forcing for parity tests (greedy sessions): replace the token the last context / generation step chose by ids[b]
 * (host, [B]) - in the output buffer, as the next step's input id and as its input embedding row - so that two
 * implementations can be compared step by step on the SAME prefix (the reference's tests feed HF's tokens the same way,
 * T/tests/model/test_llama.py:300-354).  Does not touch the KV cache or the step counters. */
int32_t tllm_session_force_tokens(tllm_session_t s, const int32_t* ids, tllm_stream_t stream);

/* Parity-test tap (sessions created with debug_taps=1 only): the input of layer `layer`'s O-projection GEMM as the last
 * generation step computed it - the attention context after the split-KV merge, [B, H/tp * Dh] fp16, or int8 when the
 * O-projection's prologue quantises it (SmoothQuant: sat(rni(ctx * attention.quantization_scaling_factor)), or the
 * per-token flavour).  This is the tensor the reference's attention tests compare at atol 2e-3
 * (T/tests/attention/test_gpt_attention.py:828-831).  HOST buffer of exactly that many bytes. */
int32_t tllm_session_get_tap(tllm_session_t s, int32_t layer, void* host, size_t nbytes, tllm_stream_t stream);
/* The same for every GEMM input of the layer, i.e. all four quantisers of the SmoothQuant block
 * (PY/quantization/layer.py:223-265 norm + quant, :385-439 MLP, :596-852 attention; K/quantization.cu:31-118):
 *   which 0  QKV input        [B, hidden]     behind input_layernorm (+ its quantiser)
 *         1  O-projection in  [B, H/tp * Dh]  (= tllm_session_get_tap)
 *         2  fc | gate input  [B, hidden]     behind post_layernorm (+ its quantiser)
 *         3  proj input       [B, inter/tp]   SwiGLU output (+ its quantiser)
 *         4  the layer's input row of the residual stream [B, hidden], always fp16
 * 0..3: fp16, or int8 with SmoothQuant - parity tests compare these against the oracle in LSBs, and check every stage of the layer
 * as a function of the engine's OWN previous tap (no compounding across stages).  HOST buffer of exactly that size. */
int32_t tllm_session_get_tap_ex(tllm_session_t s, int32_t layer, int32_t which, void* host, size_t nbytes, tllm_stream_t stream);
/* Bytes a generation step must move from HBM at context length L (weights + KV read + KV write): the
 * algorithmic-bytes model of SURVEY.md §8(d), evaluated for this session's configuration. */
int64_t tllm_session_step_bytes(tllm_session_t s, int32_t context_len);

/* Live per-kernel timing for bench.py's roofline: launches ONLY kernel K<which> of every layer (1 = RMSNorm+QKV GEMV - or, when
 * the session runs the one-launch projection + attention, that launch WITHOUT its later stages; 2 = decode attention,
 * 4 = O-projection GEMV, 5 = RMSNorm+gate|up GEMV+SwiGLU, 6 = down GEMV, 7 = the one-launch projection + attention exactly
 * as the generation step runs it, with every stage tllm_session_decode_form reports, 8 = the one-launch MLP - RMSNorm + gate|up +
 * SwiGLU + down + residual, kernels/mlp_fused.hip - where the session runs it; 5 / 6 then time the two GEMVs it replaces) back to back,
 * `sweeps` passes over the layers (each layer has its own weights, so every launch streams cold HBM exactly as in a
 * real step), bracketed by one HIP event pair on the session's stream.  avg_us = elapsed / launches (inter-launch
 * gaps included).  Activations are whatever the buffers hold: timing only, call tllm_session_fake_context or
 * tllm_session_context again before generating. */
int32_t tllm_session_time_kernel(tllm_session_t s, int32_t which, int32_t sweeps, float* avg_us, int64_t* launches,
    tllm_stream_t stream);

/* Requests tllm_session_generate ran a SECOND time because a bounded in-launch wait of the one-launch projection + attention
 * expired (the session falls back to separate launches and repeats the request: greedy generation is a function of the prompt). */
int32_t tllm_session_fused_retries(tllm_session_t s);

/* Which launches a generation step of this session is made of (decided at setup; a time-out of the one-launch form clears it):
 * bit 0 = QKV projection + RoPE + cache append + attention in one launch (kernels/qkv_attn_fused.hip), bit 1 = the O-projection +
 * residual as a stage of that launch, bit 2 = the gated MLP (gate|up GEMV + down GEMV) in one launch (kernels/mlp_fused.hip, r06;
 * opt-in with the session key fuse_mlp = 1: bit-identical, measured slower than the two GEMVs).  -1 before setup. */
int32_t tllm_session_decode_form(tllm_session_t s);

/* Instrumented generation steps (eager, a hipEvent pair around every launch) for the roofline report:
 * elapsed milliseconds and launch counts per class over `n_steps` steps.
 * Classes: 0 layer GEMVs (QKV, O, gate|up, down), 1 lm_head GEMV, 2 attention (split-KV + combine),
 *          3 embedding + sampler, 4 tensor-parallel collectives.  Arrays of TLLM_PROFILE_CLASSES entries. */
#define TLLM_PROFILE_CLASSES 5
int32_t tllm_session_profile(tllm_session_t s, int32_t n_steps, float* ms_per_class, int64_t* launches_per_class,
    tllm_stream_t stream);

void tllm_session_destroy(tllm_session_t s);

/* ------------------------------------------------------------------------------------------------
 * Kernel-level entry for the fused skinny GEMM that the generation step is built from (kernels/gemv.hip);
 * exposed so that parity tests and the micro-benchmark can drive every prologue / epilogue directly.
 * Field meanings: trtllm-llama_amd/csrc/kernels/kernels.h (GemvParams).
 * ---------------------------------------------------------------------------------------------- */
typedef struct
{
    int32_t wtype, pro, epi, out_dtype;
    int32_t M, N, K;
    const void* x;
    int64_t ldx;
    const void* w;
    int64_t ldw;
    const void* scale_col;
    const float* scale_row;
    int32_t per_channel, per_token;
    const void* gamma;
    float eps;
    const float* act_scale;
    float* dyn_scale_out;
    void* x_pro_out;
    const void* residual;
    const float* epi_scale;
    void* y;
    int64_t ldy;
} tllm_gemv_params_t;

int32_t tllm_gemv(const tllm_gemv_params_t* p, tllm_stream_t stream);
/* Test/bench knob: persistent workgroups per CU (0 = occupancy query). */
void tllm_gemv_set_blocks_per_cu(int32_t n);
/* Test/bench knob: number of rows (sequences) from which the SmoothQuant decode GEMM runs on the matrix pipe
 * (kernels/gemv_mfma_sq.hip, static activation scales) instead of the skinny vector-ALU kernel: 0 = never, -1 = the default
 * (5).  Both kernels give bit-identical results. */
void tllm_gemv_set_mfma_rows(int32_t n);
/* Test/bench knob: kernel id of the prefill GEMM (0 = tactic table, else the static rule).  1..12: lock-step tile shapes of
 * kernels/gemm_glds.hip (8 = 128x128, 6 = 256x192, 2 = 256x256, 4 = 128x256 are the production ones); 15..63: the phased
 * SmoothQuant pipeline of kernels/gemm_sqp.hip (20 = 256x192, 42 = 256x128, one tile per workgroup; 63 / 62 their persistent
 * forms, r05; 64 / 65 = 256x128 / 128x128 as two workgroups per tile splitting K, r06; 21..33 are ablations with wrong results on purpose);
 * 50..58: the same pipeline on fp16 operands (50 = 256x192, 54 = 256x128, 55 / 56 persistent, 57 / 58 split-K); 101..106: the weight-only kernel of kernels/gemm_woq.hip (101 = 256x192, 102 = 128x128,
 * 106 = 256x128); -2: the fused SwiGLU GEMM in its one-tile form.  The ids are what tllm_gemm_profile reports. */
void tllm_gemm_set_tile_cfg(int32_t cfg);

/* Kernel-level entry for the prefill GEMMs (kernels/gemm_mfma.hip; M <= 8 goes to the skinny GEMM):
 * C[m,n] = epi(sum_k A[m,k] W[n,k]); wtype / layouts / scales as tllm_gemv_params_t.  Used by the MFMA-utilisation
 * report (SmoothQuant GEMM at M = 1024) and by parity tests. */
typedef struct
{
    int32_t wtype, out_dtype;
    int32_t M, N, K;
    const void* a;
    int64_t lda;
    const void* w;
    int64_t ldw;
    const void* scale_col;
    const float* scale_row;
    int32_t per_channel, per_token;
    void* c;
    int64_t ldc;
} tllm_gemm_params_t;

int32_t tllm_gemm(const tllm_gemm_params_t* p, tllm_stream_t stream);
/* The same with the decoder layer's residual add in the epilogue: c = fp16(fp16(epi(A W^T)) + residual), residual fp16 [M, ldc]
 * (the `hidden_states = residual + attention_output` / `+ mlp_output` of Q/llama_model.py:78-86, which the prefill runs inside its
 * O- and down-projection GEMMs).  fp16 output only; parity tests of the multi-tile (persistent) kernels' residual path. */
int32_t tllm_gemm_residual(const tllm_gemm_params_t* p, const void* residual, tllm_stream_t stream);
/* The SmoothQuant MLP's fc and gate projections in one launch (static activation scales): c = int8 [M, ldc] =
 * sat(rni(fp16(silu16(fp16(A W^T s)) * fp16(A W_up^T s_up)) * quant_scale[0])) - the rounding points of GEMM + GEMM + SwiGLU +
 * quantiser run separately (PY/layers/mlp.py:68-73 with K/quantization.cu's static quantiser), which it replaces in the
 * prefill.  p->w / p->scale_col: the matrix that goes through SiLU; p->c: int8 output. */
int32_t tllm_gemm_swiglu_quant(const tllm_gemm_params_t* p, const void* w_up, const void* scale_col_up, const float* quant_scale,
    tllm_stream_t stream);
/* On-device tactic selection for the prefill GEMMs (M >= 32; SmoothQuant int8 = wtype 3, fp16 = wtype 0 - also what the weight-only
 * prefill runs on its expanded weights).  Takes the place of the profile the reference's SmoothQuant GEMM plugin runs when an
 * engine is built - every CUTLASS tile configuration timed on the device per M bucket
 * (K/cutlass_kernels/int8_gemm/int8_gemm_template.h:372-457), the winners kept in the plugin's serialisation
 * (P/smoothQuantGemmPlugin/smoothQuantGemmPlugin.cpp:253-282):
 *   profile : times every candidate kernel (tile shape x lock-step / phased pipeline) for C[M, N] = A[M, K] W[N, K]^T on random
 *             operands of its own and records the fastest in the process-wide table; best_cfg = its id (as tllm_gemm_set_tile_cfg
 *             takes them), 0 when no MFMA kernel serves the shape.  A session does this for its own four GEMM shapes at
 *             tllm_session_setup (unless the engine brought the table, or TLLM_GEMM_TACTICS=off);
 *   export  : the table as text "wtype:M:N:K:cfg:us;..." - returns the bytes needed including the terminator; the builder stores
 *             it in the engine file (header line `gemm_tactics=`), tllm_session_load_engine / tllm_session_create import it;
 *   lookup  : the kernel id tllm_gemm will use for the shape (exact M, else the nearest profiled M of the same power-of-two
 *             bucket), 0 = the static "fewest workgroup rounds" rule. */
int32_t tllm_gemm_profile(int32_t wtype, int32_t M, int32_t N, int32_t K, int32_t* best_cfg, float* best_us, tllm_stream_t stream);
int64_t tllm_gemm_tactics_export(char* buf, int64_t capacity);
int32_t tllm_gemm_tactics_import(const char* text);
void tllm_gemm_tactics_clear(void);
int32_t tllm_gemm_tactic_lookup(int32_t wtype, int32_t M, int32_t N, int32_t K);
/* Microbenchmark hook: while set (non-NULL), every workgroup of the phased SmoothQuant GEMM writes {shader cycles, ticks of
 * the constant 100 MHz counter} over its lifetime to device_buffer[2 * workgroup] (uint64) - the clock the chip held. */
void tllm_gemm_set_clock_probe(void* device_buffer);

#ifdef __cplusplus
}
#endif

#endif /* TLLM_RUNTIME_API_H */
