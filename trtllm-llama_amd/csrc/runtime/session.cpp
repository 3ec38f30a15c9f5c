// Host decode loop of the LLaMA path (include/tllm_runtime_api.h).
//
// What the reference does with a TensorRT engine + two execution contexts + a Python loop that syncs with the
// device every step (PY/runtime/generation.py:852-983), this does with: named weight tensors, the plugin
// kernels enqueued layer by layer on one HIP stream, a generation step whose step-dependent scalars
// (sequence length, current token, finished flags) live in device memory — so one captured hipGraph is
// replayed for every step and the host never waits inside the loop.
//
// Layer graph = Q/llama_model.py:78-119 (LLaMADecoderLayer.forward); model head/tail = :159-207, :253-287.
// Generation-step fusion (what TensorRT/Myelin fuses out of pointwise layers, done here by construction):
//   K1  RMSNorm(+quant)  ->  QKV GEMV                       (gemv prologue)
//   K2  RoPE + KV append + split-KV attention, K3 combine   (mmha_decode)
//   K4  (quant) -> O GEMV -> + residual                     (gemv prologue/epilogue)
//   K5  RMSNorm(+quant) -> gate|up GEMV -> silu*mul(+quant) (gemv prologue/epilogue)
//   K6  (quant) -> down GEMV -> + residual
// SmoothQuant block template: SURVEY Appendix A.4 (the reference's SmoothQuant-LLaMA never ran; designed by
// analogy to PY/quantization/layer.py:385-439,596-852).
#include "../../../include/tllm_runtime_api.h"
#include "../kernels/kernels.h"
#include "../kernels/weight_layout.h"
#include "../plugins/comm.h"
#include "../plugins/plugin_base.h"
#include "engine_check.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

using namespace tllm;
using namespace tllm::kernels;

namespace tllm
{
namespace kernels
{
extern int gemv_tune_blocks_per_cu;
extern int gemv_mfma_min_rows;
extern int gemm_tune_cfg;
extern int gemm_woq_tune_cfg;
extern bool gemm_swiglu_one_tile; // gemm_sqp.hip: A/B hook
extern void* gemm_clock_probe;
}
} // namespace tllm

namespace
{

enum QuantBits
{
    QM_INT4_WEIGHTS = 1,
    QM_INT8_WEIGHTS = 2,
    QM_ACTIVATIONS = 4,
    QM_PER_CHANNEL = 8,
    QM_PER_TOKEN = 16,
    QM_INT8_KV = 32
};

struct TensorRec
{
    int32_t dtype = 0;
    std::vector<int64_t> dims;
    void* dev = nullptr;
    bool owned = false;
    size_t bytes = 0;
    int64_t numel() const
    {
        int64_t n = 1;
        for (auto d : dims)
            n *= d;
        return n;
    }
};

struct Linear
{
    int wtype = W_FP16;
    const void* w = nullptr;
    int64_t ldw = 0;
    int N = 0, K = 0;
    const void* scale_col = nullptr; // fp16 [N] (weight-only) | f32 [N] or [1] (SmoothQuant)
    int per_channel = 0;
    const float* act_scale = nullptr; // SmoothQuant static: dequant scale of the GEMM [1,1]
};

struct Layer
{
    const void* ln1 = nullptr;
    const void* ln2 = nullptr;
    const float* ln1_scale = nullptr;  // input_layernorm.scale_to_int (SQ static)
    const float* ln2_scale = nullptr;  // post_layernorm.scale_to_int
    const float* attn_qscale = nullptr; // attention.quantization_scaling_factor (ctx -> int8, SQ static)
    const float* mlp_qscale = nullptr;  // mlp.quantization_scaling_factor (silu*mul -> int8, SQ static)
    const float* kv_oq = nullptr;
    const float* kv_qo = nullptr;
    Linear qkv, dense, fc, gate, proj;
    void* kv = nullptr;                // linear cache [B, 2, Hr, Smax, Dh], or the block pool [2, blocks, Hr, tokens_per_block, Dh]
    const int64_t* kv_table = nullptr; // paged: device table int64 [B, 2, max_blocks] of block pointers into `kv`
};

size_t dtype_bytes(int32_t t)
{
    switch (t)
    {
    case TLLM_FLOAT:
    case TLLM_INT32: return 4;
    case TLLM_HALF: return 2;
    default: return 1;
    }
}

#define HIP_OK(expr)                                                                                                   \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t _e = (expr);                                                                                        \
        if (_e != hipSuccess)                                                                                          \
        {                                                                                                              \
            set_error("%s failed: %s", #expr, hipGetErrorString(_e));                                                  \
            return 1;                                                                                                  \
        }                                                                                                              \
    } while (0)

#define RUN(expr)                                                                                                      \
    do                                                                                                                 \
    {                                                                                                                  \
        const int _rc = (expr); /* (the callee's code travels up: kFusedTimedOut is told apart from a plain failure) */   \
        if (_rc != 0)                                                                                                  \
            return _rc;                                                                                                \
    } while (0)

} // namespace

struct tllm_session
{
    // ---- configuration
    int num_layers = 0, num_heads = 0, hidden = 0, inter = 0, vocab = 0, max_pos = 2048;
    int tp = 1, rank = 0;
    int quant_mode = 0;
    int neox = 1;
    float eps = 1e-6f;
    std::string wo_precision = "int8";
    std::string network_json; // the traced network an engine file carries (Builder.build_engine), verified by load_engine
    // derived
    int Hr = 0, Dh = 0, Dr = 0, Ir = 0, Vr = 0;
    bool sq = false, woq = false, int8_kv = false, per_token = false, per_channel = false;
    int wtype = W_FP16;

    std::map<std::string, TensorRec> tensors;
    std::vector<Layer> layers;
    const void* emb = nullptr;
    const void* lnf = nullptr;
    Linear head;
    bool finalized = false;
    std::vector<int32_t> group;
    bool packed = false;       // remove_input_padding: the context phase runs on the real tokens only
    int ctx_tokens = 0;        // ... their number in the current prompt batch
    int32_t* cu_dev = nullptr;    // [B + 1] exclusive prefix sum of the input lengths
    int32_t* last_rows = nullptr; // [B] packed row of every sequence's last prompt token
    // paged KV cache (plugin field paged_kv_cache; K/kvCacheUtils.h KVBlockArray, PY/runtime/kv_cache_manager.py): the session
    // owns the pool and hands every sequence its blocks at setup - the whole table is known then, so the generation graph
    // needs no host-side block allocation between steps
    bool paged_kv = false;
    int tokens_per_block = 64, max_blocks = 0;
    size_t kv_elems = 0; // elements of one layer's cache / pool
    bool force_comm = false; // tests: run the TP collectives on a 1-rank communicator too (RCCL inside the captured graph)
    // session key no_comm = 1: a rank's launches WITHOUT its collectives (all-reduces and the logits all-gather are skipped, nothing
    // else changes) - the per-rank step time of a tensor-parallel shard on one GPU, bench.py's prediction for the first multi-GPU
    // run.  TIMING ONLY: hidden states are one rank's partial sums.
    bool no_comm = false;
    bool debug_taps = false; // tests: keep every layer's GEMV inputs of the last generation step (tllm_session_get_tap[_ex])
    // tap w of layer li: the activation exactly as GEMV w consumes it, behind its prologue (RMSNorm / split merge / quantiser):
    //   0 QKV input [B, D]   1 O-projection input [B, Dr]   2 gate|up input [B, D]   3 down-projection input [B, Ir]
    // fp16, or s8 where the path quantises (SmoothQuant) - the four quantisers of the SmoothQuant layer;
    //   4 the layer's input row of the residual stream [B, D], always fp16 (tap 4 of layer num_layers - 1 + 1 does not exist:
    //     the last layer's output is what the head consumes)
    static constexpr int kTaps = 5;
    char* tap_buf[kTaps] = {nullptr, nullptr, nullptr, nullptr, nullptr}; // each [num_layers][B][width] x 2 bytes
    char* tap_dst = nullptr; // where the launch being issued leaves its prologue's result (gemv() routes x_pro_out there)
    int tap_width(int which) const { return which == 1 ? Dr : (which == 3 ? Ir : hidden); }
    char* tap_ptr(int which, int li) const { return tap_buf[which] + (size_t) li * B * tap_width(which) * 2; }

    // ---- runtime state (setup)
    int B = 0, max_in = 0, max_new = 0, Smax = 0;
    // beam search: Bc prompts, `beam` hypotheses each; B = Bc * beam sequences in the generation phase (B == Bc otherwise)
    int Bc = 0, beam = 1;
    int logit_rows = 0; // rows of the last head launch (Bc after the prompt, B after a generation step)
    float* cum_log_probs = nullptr; // [B]
    int32_t *parent_ids = nullptr, *cache_ind = nullptr, *in_len_ctx = nullptr; // [B, Smax], [B, Smax], [Bc]
    std::vector<void*> allocs;
    void *x = nullptr, *qkv = nullptr, *ctx = nullptr, *g = nullptr, *u = nullptr, *inter_buf = nullptr, *tmp = nullptr;
    int8_t* q8 = nullptr;   // quantised activations (context path) [B*S, max(D, I)]
    float* qscale = nullptr; // per-token scales [B*S]
    float* logits = nullptr; // [B, V] (or gathered [tp, B, Vr])
    float* logits_local = nullptr;
    void* last_hidden = nullptr;
    // tensor-parallel decode with the fused peer-to-peer seam (kernels/p2p_allreduce.hip): this rank's partial projection
    // output, the normalised (+ quantised) row the next GEMV consumes, its per-token scales
    void* ar_partial = nullptr; // [B, D] fp16
    void* ar_norm = nullptr;    // [B, D] fp16 | s8
    float* ar_scale = nullptr;  // [B]
    void* mmha_ws = nullptr;
    void* ctx_ws = nullptr; // V^T scratch of the MFMA context attention
    int32_t *ids_in = nullptr, *cur_ids = nullptr, *out_ids = nullptr, *seq_len = nullptr, *in_len = nullptr,
            *masked = nullptr, *finished = nullptr, *last_tok = nullptr;
    const float* rope = nullptr;
    int rope_len = 0;
    float* rope_row = nullptr; // [B, Dh/2, 2]: cos/sin row of the next generation step (written by the sampler)
    int32_t* rope_pos = nullptr; // [B]: the position that row belongs to (tllm_session_get_step_state)
    int attn_nit = 4, attn_tchunk = 0, attn_ns = 0;
    size_t attn_o_off = 0;
    // the last split of a head to arrive merges inside the attention launch (mmha_decode.hip step 6); beyond 16 partials the finest
    // split runs with its own combine launch
    bool attn_tail = false;
    uint32_t* attn_tickets = nullptr;
    // r05: batch-1 greedy decode of a SmoothQuant engine runs the QKV projection, RoPE, the cache append and the attention of a head
    // in ONE launch (kernels/qkv_attn_fused.hip); session key fuse_qkv_attention = 0 keeps the two launches (A/B, parity tests)
    int fuse_qkv_cfg = -1;          // -1 auto, 0 off
    bool qkv_attn_fused = false;    // decided at setup
    // ... and the O-projection + residual of the layer as a third stage of that launch (static SmoothQuant: the context row
    // travels as its int8 image); session key fuse_o_projection = 0 keeps the GEMV launch
    int fuse_o_cfg = -1;
    int fused_retries = 0;          // requests tllm_session_generate ran a second time behind an expired in-launch wait
    int dual_mlp_cfg = -1;          // session key dual_mlp_gemm = 0: prefill fc / gate as two GEMMs + the SwiGLU-quantiser pass (A/B)
    int fused_max_spins = -1;       // session key fused_max_spins: bound of the in-launch waits (tests: 0 = the first miss times out)
    bool o_fused = false;
    uint64_t* fused_xchg = nullptr; // granule exchange, shared by all layers
    uint32_t* step_epoch = nullptr; // advanced by the sampler once per generation step (the granule tags derive from it)
    uint32_t* fused_err = nullptr;  // raised by a bounded wait that expired
    uint32_t timing_tag = 0;        // explicit tags of eager launches outside a step (tllm_session_time_kernel)
    // r06: the gated MLP of a decode step (gate|up GEMV + down GEMV) in ONE launch (kernels/mlp_fused.hip): batch 1, tp 1, static
    // SmoothQuant, the 7B extents.  Bit-identical to the two GEMV launches but measured 1 us per layer SLOWER (26.4 against
    // 16.4 + 9.0 us, profiles/r06_mlp_one_launch.txt), so it runs only when asked for: session key fuse_mlp = 1
    int fuse_mlp_cfg = 0;
    bool mlp_fused_dec = false;  // decided at setup
    uint8_t* mlp_flags = nullptr; // one byte per workgroup (shared by all layers), zero before the first launch and after a failed one
    uint64_t* mlp_timing = nullptr;
    uint64_t* fused_timing = nullptr; // session key fused_timeline = 1: stage clock of the fused launch, [Hr * 8][16] ticks
    bool fused_timeline = false;
    void* ctx_q8 = nullptr;
    int end_id = -1;
    hipGraphExec_t graph = nullptr;
    hipStream_t graph_stream = nullptr;
    uint64_t graph_comm_gen = 0;   // comm::p2p::generation() the step graph was captured under
    uint64_t comm_err_seen = 0;    // comm::p2p::error_generation() at this session's last check_comm
    hipStream_t own_stream = nullptr; // used when the caller passes the NULL stream (it cannot be captured)

    // ---- optional per-launch instrumentation (tllm_session_profile): event pairs around every launch class
    enum ProfClass
    {
        PC_GEMV_LAYER = 0,
        PC_GEMV_HEAD = 1,
        PC_ATTENTION = 2,
        PC_OTHER = 3,
        PC_COMM = 4,
        PC_COUNT = 5
    };
    bool profiling = false;
    int only_kernel = -1; // tllm_session_time_kernel: launch just K<only_kernel> of every layer (1,2,4,5,6)
    int gemv_cls = PC_GEMV_LAYER;
    struct ProfRec
    {
        hipEvent_t a, b;
        int cls;
    };
    std::vector<ProfRec> prof;

    hipStream_t pick(tllm_stream_t stream)
    {
        if (stream)
            return reinterpret_cast<hipStream_t>(stream);
        if (!own_stream)
            (void) hipStreamCreate(&own_stream);
        return own_stream;
    }

    template <typename F>
    int timed(int cls, hipStream_t st, F&& f)
    {
        if (!profiling)
            return f();
        ProfRec r;
        r.cls = cls;
        (void) hipEventCreate(&r.a);
        (void) hipEventCreate(&r.b);
        (void) hipEventRecord(r.a, st);
        const int rc = f();
        (void) hipEventRecord(r.b, st);
        prof.push_back(r);
        return rc;
    }

    ~tllm_session()
    {
        if (own_stream)
            (void) hipStreamDestroy(own_stream);
        if (graph)
            (void) hipGraphExecDestroy(graph);
        for (auto p : allocs)
            (void) hipFree(p);
        for (auto& kv : tensors)
            if (kv.second.owned && kv.second.dev)
                (void) hipFree(kv.second.dev);
    }

    // ------------------------------------------------------------------------------------------ helpers
    template <typename T>
    int dalloc(T** p, size_t bytes)
    {
        void* d = nullptr;
        if (hipMalloc(&d, bytes ? bytes : 16) != hipSuccess)
        {
            set_error("session: hipMalloc(%zu) failed", bytes);
            return 1;
        }
        allocs.push_back(d);
        *p = static_cast<T*>(d);
        return 0;
    }

    void free_runtime()
    {
        if (graph)
        {
            (void) hipGraphExecDestroy(graph);
            graph = nullptr;
        }
        for (auto p : allocs)
            (void) hipFree(p);
        allocs.clear();
    }

    const TensorRec* find(const std::string& name, bool required = true)
    {
        auto it = tensors.find(name);
        if (it == tensors.end())
        {
            if (required)
                set_error("session: missing tensor '%s'", name.c_str());
            return nullptr;
        }
        return &it->second;
    }

    int want(const TensorRec* t, const std::string& name, int32_t dtype, int64_t numel)
    {
        if (!t)
            return 1;
        if (t->dtype != dtype || t->numel() != numel)
        {
            set_error("session: tensor '%s' has dtype %d / %lld elements, expected dtype %d / %lld", name.c_str(),
                t->dtype, (long long) t->numel(), dtype, (long long) numel);
            return 1;
        }
        return 0;
    }

    // resolve one linear layer "prefix" with logical shape [N, K] for this session's quantisation mode
    int resolve_linear(const std::string& prefix, int N, int K, Linear& L, bool force_fp16 = false)
    {
        L.N = N;
        L.K = K;
        const std::string wn = prefix + ".weight";
        const TensorRec* w = find(wn);
        if (!w)
            return 1;
        if (force_fp16 || (!sq && !woq))
        {
            RUN(want(w, wn, TLLM_HALF, (int64_t) N * K));
            L.wtype = W_FP16;
            L.w = w->dev;
            L.ldw = (int64_t) K * 2;
            return 0;
        }
        if (woq)
        {
            // processed bytes, declared fp32 [K, N/4 | N/8] (reference view) or int8 [N, ldw]
            L.wtype = wtype;
            L.ldw = layout::row_bytes(wtype, K);
            if ((int64_t) w->bytes != (int64_t) N * L.ldw)
            {
                set_error("session: tensor '%s' has %zu bytes, expected %lld (processed weight-only layout)",
                    wn.c_str(), w->bytes, (long long) N * L.ldw);
                return 1;
            }
            L.w = w->dev;
            const TensorRec* s = find(prefix + ".per_channel_scale");
            RUN(want(s, prefix + ".per_channel_scale", TLLM_HALF, N));
            L.scale_col = s->dev;
            return 0;
        }
        // SmoothQuant: int8 [N, K] (or fp32 view [N, K/4])
        L.wtype = W_INT8_SQ;
        L.ldw = K;
        if ((int64_t) w->bytes != (int64_t) N * K || (K % 16))
        {
            set_error("session: tensor '%s' must hold N*K = %lld int8 values with K %% 16 == 0", wn.c_str(),
                (long long) N * K);
            return 1;
        }
        L.w = w->dev;
        const TensorRec* s = find(prefix + ".per_channel_scale");
        if (!s)
            return 1;
        L.per_channel = s->numel() == N ? 1 : 0;
        if (s->dtype != TLLM_FLOAT || (s->numel() != N && s->numel() != 1))
        {
            set_error("session: '%s.per_channel_scale' must be f32 [1,%d] or [1,1]", prefix.c_str(), N);
            return 1;
        }
        L.scale_col = s->dev;
        if (!per_token)
        {
            const TensorRec* a = find(prefix + ".act_scale");
            RUN(want(a, prefix + ".act_scale", TLLM_FLOAT, 1));
            L.act_scale = static_cast<const float*>(a->dev);
        }
        return 0;
    }

    int scalar_f32(const std::string& name, const float** out)
    {
        const TensorRec* t = find(name);
        RUN(want(t, name, TLLM_FLOAT, 1));
        *out = static_cast<const float*>(t->dev);
        return 0;
    }

    // ------------------------------------------------------------------------------------------ GEMM wrappers
    // decode: fused skinny GEMM
    int gemv(const Linear& L, int M, int pro, int epi, const void* xin, int64_t ldx, const void* gamma,
        const float* in_qscale, const void* residual, const float* epi_scale, void* y, int64_t ldy, int out_dtype,
        const Linear* up, hipStream_t st, const float* row_scales = nullptr)
    {
        GemvParams p;
        p.wtype = L.wtype;
        p.pro = pro;
        p.epi = epi;
        p.out_dtype = out_dtype;
        p.M = M;
        p.N = L.N;
        p.K = L.K;
        p.x = xin;
        p.ldx = ldx;
        p.w = L.w;
        p.ldw = L.ldw;
        p.scale_col = L.scale_col;
        p.scale_row = L.act_scale;
        p.per_channel = L.per_channel;
        p.per_token = 0;
        if (row_scales) // activations quantised per token upstream (the fused all-reduce tail): one dequantisation scale per row
        {
            p.scale_row = row_scales;
            p.per_token = 1;
        }
        p.gamma = gamma;
        p.eps = eps;
        p.act_scale = in_qscale;
        p.residual = residual;
        p.epi_scale = epi_scale;
        p.y = y;
        p.ldy = ldy;
        if (tap_dst)
            p.x_pro_out = tap_dst;
        if (up)
        {
            p.w_up = up->w;
            p.scale_col_up = up->scale_col;
            p.scale_row_up = up->act_scale;
        }
        if (M <= 8)
            return timed(gemv_cls, st, [&] { return launch_gemv(p, st) ? 1 : 0; });
        // more than 8 sequences (batch x beam width): the skinny kernel takes 8 rows per launch, so the rows go through it in
        // slabs of 8 - every operand that is indexed by the row moves along.  Each slab streams the weights again: beyond 8
        // sequences a step costs ceil(B / 8) weight passes (the throughput per sequence of B = 8), but nothing is refused -
        // build.py's default --max_batch_size 8 with any beam width > 1 asks for exactly this (Q/build.py:73-76).
        const int xes = (L.wtype == W_INT8_SQ && pro == PRO_NONE) ? 1 : 2; // raw s8 activations, else fp16
        const int yes = out_dtype == DT_INT8 ? 1 : (out_dtype == DT_HALF ? 2 : 4);
        const int tap_es = L.wtype == W_INT8_SQ ? 1 : 2;
        return timed(gemv_cls, st, [&] {
            for (int m0 = 0; m0 < M; m0 += 8)
            {
                GemvParams q = p;
                q.M = M - m0 < 8 ? M - m0 : 8;
                q.x = static_cast<const char*>(xin) + (int64_t) m0 * ldx * xes;
                q.y = static_cast<char*>(y) + (int64_t) m0 * ldy * yes;
                if (residual)
                    q.residual = static_cast<const char*>(residual) + (int64_t) m0 * ldy * 2;
                if (p.per_token && p.scale_row)
                    q.scale_row = p.scale_row + m0;
                if (p.x_pro_out)
                    q.x_pro_out = static_cast<char*>(p.x_pro_out) + (int64_t) m0 * L.K * tap_es;
                if (launch_gemv(q, st))
                    return 1;
            }
            return 0;
        });
    }

    // context: plain GEMM on M rows (activation already in the operand type)
    int gemm(const Linear& L, int M, const void* a, const float* scale_row, int per_tok, void* c, int out_dtype,
        hipStream_t st, const void* residual = nullptr, const void* silu_gate = nullptr)
    {
        GemmParams g;
        g.residual = residual;
        g.silu_gate = silu_gate;
        g.wtype = L.wtype;
        g.out_dtype = out_dtype;
        g.M = M;
        g.N = L.N;
        g.K = L.K;
        g.a = a;
        g.lda = L.K;
        g.w = L.w;
        g.ldw = L.ldw;
        g.scale_col = L.scale_col;
        g.scale_row = scale_row ? scale_row : L.act_scale;
        g.per_channel = L.per_channel;
        g.per_token = per_tok;
        g.c = c;
        g.ldc = L.N;
        return launch_gemm(g, st) ? 1 : 0;
    }

    // On-device tactic selection (kernels/gemm_tactics.hip; reference: int8_gemm_template.h:372-457, stored per M bucket in the
    // plugin, smoothQuantGemmPlugin.cpp:253-282): the MFMA kernel of each prefill GEMM shape of this model at M rows is the one
    // that measured fastest on THIS device - timed once per process and shape unless the engine file brought the choice along.
    // TLLM_GEMM_TACTICS=off keeps the static rule.
    int profile_prefill_gemms(int M)
    {
        static const bool off = [] {
            const char* e = getenv("TLLM_GEMM_TACTICS");
            return e && (!strcmp(e, "off") || !strcmp(e, "0"));
        }();
        if (off || layers.empty() || packed) // packed inputs: M varies with the prompt batch, the nearest bucket entry serves
            return 0;
        const Layer& L = layers[0];
        // weight-only prefill runs gemm_woq.hip (one kernel per shape class, no tactic table)
        if (L.qkv.wtype == W_INT8_WOQ || L.qkv.wtype == W_INT4_WOQ)
            return 0;
        for (const Linear* l : {&L.qkv, &L.dense, &L.fc, &L.proj})
        {
            const int wt = l->wtype == W_INT8_SQ ? W_INT8_SQ : W_FP16;
            // an entry of the same power-of-two M bucket (what the engine file brought along, Builder._profile_gemm_tactics)
            // serves: the launcher would use it for this M anyway
            if (gemm_tactic_lookup(wt, M, l->N, l->K) > 0)
                continue;
            int cfg = 0;
            float us = 0.f;
            // best effort: a profile that cannot run (no memory left for its operands next to a large session) leaves the shape
            // to the static rule - it must not fail the set-up, nor leave its message behind for a later, unrelated failure
            if (gemm_profile(wt, M, l->N, l->K, &cfg, &us, nullptr))
            {
                set_error("%s", "");
                break;
            }
        }
        return 0;
    }

    int allreduce(void* buf, int64_t n, hipStream_t st)
    {
        if ((tp == 1 && !force_comm) || no_comm)
            return 0;
        if (comm::p2p::usable(tp, n * 2))
            return timed(PC_COMM, st, [&] { return comm::p2p::all_reduce_f16(buf, n, st) ? 1 : 0; });
        // a vector longer than one inbox slot (the prefill's [tokens, D] partial sums) goes through the peer-to-peer path in
        // slot-sized pieces - the same kernel, one epoch per piece - instead of silently needing a second transport
        // (only when no RCCL communicator exists for the group: the one-workgroup exchange kernel is built for the decode
        // step's 8 KB vectors, a ring is the better transport for the prefill's megabytes)
        const int64_t cap = comm::p2p::slot_capacity(tp) / 2 / 8 * 8; // fp16 elements per exchange, whole 16-byte vectors
        if (cap > 0 && n % 8 == 0 && !comm::has_comm(group))
        {
            static bool warned = false;
            if (!warned && n > 8 * cap)
            {
                warned = true;
                fprintf(stderr, "[tllm] warning: a %lld-element all-reduce goes through the peer-to-peer inbox in %lld pieces of %lld "
                                "(no RCCL communicator is registered for this group): register one (tllm_comm_init_rank) or size the inbox "
                                "(tllm_comm_p2p_create max_bytes) for the prefill\n",
                    (long long) n, (long long) ((n + cap - 1) / cap), (long long) cap);
            }
            return timed(PC_COMM, st, [&] {
                for (int64_t off = 0; off < n; off += cap)
                    if (comm::p2p::all_reduce_f16(static_cast<char*>(buf) + off * 2, n - off < cap ? n - off : cap, st))
                        return 1;
                return 0;
            });
        }
        return timed(PC_COMM, st, [&] { return comm::all_reduce_sum(group, buf, buf, n, TLLM_HALF, st) ? 1 : 0; });
    }

    // ------------------------------------------------------------------------------------------ context step
    int run_context(hipStream_t st)
    {
        const int S = max_in, M = packed ? ctx_tokens : Bc * S, D = hidden;
        RUN(launch_embedding(x, ids_in, emb, M, D, vocab, st));
        for (int li = 0; li < num_layers; ++li)
        {
            Layer& L = layers[li];
            // --- attention block
            const void* a_in = tmp;
            RmsnormParams r;
            r.M = M;
            r.N = D;
            r.x = x;
            r.gamma = L.ln1;
            r.eps = eps;
            if (sq)
            {
                r.q = q8;
                if (per_token)
                    r.dyn_scale_out = qscale;
                else
                    r.static_scale = L.ln1_scale;
                a_in = q8;
            }
            else
                r.y = tmp;
            RUN(launch_rmsnorm(r, st));
            RUN(gemm(L.qkv, M, a_in, sq && per_token ? qscale : nullptr, sq && per_token, qkv, DT_HALF, st));
            ContextAttnParams c;
            c.batch = Bc;
            c.seq = S;
            c.num_heads = Hr;
            c.head_size = Dh;
            c.rotary_dim = Dh;
            c.neox = neox;
            c.inv_sqrt_dh = 1.f / sqrtf((float) Dh);
            c.int8_kv = int8_kv;
            c.max_seq_len = Smax;
            c.qkv = qkv;
            c.kv_cache = L.kv;
            c.input_lengths = in_len_ctx;
            c.cache_seq_stride = beam;
            c.block_pointers = L.kv_table;
            c.tokens_per_block = tokens_per_block;
            c.max_blocks_per_seq = max_blocks;
            c.kv_scale_orig_quant = L.kv_oq;
            c.rope_table = rope;
            c.rope_table_len = rope_len;
            c.out = ctx;
            c.workspace = ctx_ws;
            c.cu_seqlens = packed ? cu_dev : nullptr;
            // SmoothQuant static: the O-projection's input quantiser rides in the attention's epilogue (padded inputs; with
            // packed inputs M counts real tokens only and the pass below covers exactly those)
            const bool q_in_attn = sq && !per_token && !packed;
            if (q_in_attn)
            {
                c.out_q8 = q8;
                c.out_q_scale = L.attn_qscale;
            }
            RUN(launch_context_attention(c, st));
            const void* d_in = ctx;
            if (sq)
            {
                if (per_token)
                    RUN(launch_quantize_per_token(q8, ctx, DT_HALF, M, Dr, qscale, st));
                else if (!q_in_attn)
                    RUN(launch_quantize_tensor(q8, ctx, DT_HALF, (int64_t) M * Dr, L.attn_qscale, st));
                d_in = q8;
            }
            // (weight-only: gemm_woq.hip adds the residual in its epilogue too - same two roundings; its own serve conditions)
            const bool woq_w = L.dense.wtype == W_INT8_WOQ || L.dense.wtype == W_INT4_WOQ;
            const bool fuse_res = tp == 1 && !force_comm && M >= 32 && D % 8 == 0
                && (woq_w ? (L.dense.K % 64 == 0 && L.proj.K % 64 == 0)
                          : ((L.dense.wtype == W_INT8_SQ || L.dense.wtype == W_FP16)
                              && (L.dense.K * (L.dense.wtype == W_FP16 ? 2 : 1)) % 128 == 0
                              && (L.proj.K * (L.proj.wtype == W_FP16 ? 2 : 1)) % 128 == 0));
            if (fuse_res)
            {
                // x <- x + O(ctx): residual fused into the GEMM epilogue (same rounding: fp16(gemm) then fp16(sum))
                RUN(gemm(L.dense, M, d_in, sq && per_token ? qscale : nullptr, sq && per_token, x, DT_HALF, st, x));
            }
            else
            {
                RUN(gemm(L.dense, M, d_in, sq && per_token ? qscale : nullptr, sq && per_token, tmp, DT_HALF, st));
                RUN(allreduce(tmp, (int64_t) M * D, st));
                RUN(launch_add(x, x, tmp, (int64_t) M * D, st));
            }
            // --- MLP block
            r = RmsnormParams();
            r.M = M;
            r.N = D;
            r.x = x;
            r.gamma = L.ln2;
            r.eps = eps;
            a_in = tmp;
            if (sq)
            {
                r.q = q8;
                if (per_token)
                    r.dyn_scale_out = qscale;
                else
                    r.static_scale = L.ln2_scale;
                a_in = q8;
            }
            else
                r.y = tmp;
            RUN(launch_rmsnorm(r, st));
            const void* p_in = inter_buf;
            bool mlp_fused = false;
            if (sq && !per_token && M >= 32 && dual_mlp_cfg != 0)
            {
                // fc and gate in one kernel with SwiGLU + the static quantiser in its epilogue (gemm_sqp.hip, DUAL): the two fp16
                // [M, Ir] intermediates and the pointwise pass between the GEMMs disappear.  The int8 result goes to inter_buf
                // (q8 is this kernel's INPUT)
                GemmParams d;
                d.wtype = L.fc.wtype;
                d.out_dtype = DT_INT8;
                d.M = M;
                d.N = L.fc.N;
                d.K = L.fc.K;
                d.a = a_in;
                d.lda = L.fc.K;
                d.w = L.fc.w;
                d.ldw = L.fc.ldw;
                d.scale_col = L.fc.scale_col;
                d.scale_row = L.fc.act_scale;
                d.per_channel = L.fc.per_channel;
                d.per_token = 0;
                d.c = inter_buf;
                d.ldc = L.fc.N;
                d.w2 = L.gate.w;
                d.scale_col2 = L.gate.scale_col;
                d.scale_row2 = L.gate.act_scale;
                d.swiglu_qscale = L.mlp_qscale;
                if (L.gate.ldw == L.fc.ldw && L.gate.per_channel == L.fc.per_channel && L.gate.N == L.fc.N && L.gate.K == L.fc.K)
                {
                    const int rc = launch_gemm_swiglu(d, st);
                    if (rc < 0)
                        return 1;
                    mlp_fused = rc == 0;
                }
            }
            if (mlp_fused)
                p_in = inter_buf;
            else
            {
            RUN(gemm(L.fc, M, a_in, sq && per_token ? qscale : nullptr, sq && per_token, g, DT_HALF, st));
            if (!sq)
            {
                // fp16 / weight-only: SwiGLU folded into the second projection's epilogue (g is read there instead of in a pass of
                // its own; same rounding points) - one launch and a [M, Ir] write + read fewer per layer
                RUN(gemm(L.gate, M, a_in, nullptr, 0, inter_buf, DT_HALF, st, nullptr, g));
            }
            else
            {
            RUN(gemm(L.gate, M, a_in, sq && per_token ? qscale : nullptr, sq && per_token, u, DT_HALF, st));
            if (sq && !per_token)
            {
                RUN(launch_swiglu_quant(q8, g, u, (int64_t) M * Ir, L.mlp_qscale, st)); // SwiGLU and its quantiser in one pass
                p_in = q8;
            }
            else
            {
                RUN(launch_swiglu(inter_buf, g, u, (int64_t) M * Ir, st));
                if (sq)
                {
                    RUN(launch_quantize_per_token(q8, inter_buf, DT_HALF, M, Ir, qscale, st));
                    p_in = q8;
                }
            }
            }
            }
            if (fuse_res)
            {
                RUN(gemm(L.proj, M, p_in, sq && per_token ? qscale : nullptr, sq && per_token, x, DT_HALF, st, x));
            }
            else
            {
                RUN(gemm(L.proj, M, p_in, sq && per_token ? qscale : nullptr, sq && per_token, tmp, DT_HALF, st));
                RUN(allreduce(tmp, (int64_t) M * D, st));
                RUN(launch_add(x, x, tmp, (int64_t) M * D, st));
            }
        }
        // head: last real token of every sequence -> ln_f -> lm_head -> fp32 logits  (Q/llama_model.py:272-279)
        if (packed)
            RUN(launch_gather_rows(last_hidden, x, last_rows, Bc, D, st));
        else
            RUN(launch_gather_last_token(last_hidden, x, last_tok, Bc, S, D, st));
        RUN(run_head(last_hidden, Bc, st));
        return 0;
    }

    int run_head(const void* h, int rows, hipStream_t st, bool normalised = false)
    {
        logit_rows = rows;
        gemv_cls = PC_GEMV_HEAD;
        const int head_rc = gemv(head, rows, normalised ? PRO_NONE : PRO_RMSNORM, EPI_NONE, h, hidden, lnf, nullptr, nullptr, nullptr,
            logits_local, Vr, DT_FLOAT, nullptr, st);
        gemv_cls = PC_GEMV_LAYER;
        RUN(head_rc);
        if ((tp > 1 || force_comm) && !no_comm)
        {
            const int64_t bytes = (int64_t) rows * Vr * 4;
            if (comm::p2p::usable(tp, bytes))
            {
                if (comm::p2p::all_gather(logits_local, logits, bytes, st))
                    return 1;
            }
            else if (comm::all_gather(group, logits_local, logits, (int64_t) rows * Vr, TLLM_FLOAT, st))
                return 1;
        }
        return 0;
    }

    int run_sampler(int advance, hipStream_t st)
    {
        if (beam > 1)
        {
            BeamParams bp;
            bp.logits = (tp > 1 || force_comm) ? logits : logits_local;
            bp.logits_per_batch = advance ? 0 : 1;
            bp.batch = Bc;
            bp.beam = beam;
            bp.vocab_part = Vr;
            bp.nparts = tp;
            bp.vocab = vocab;
            bp.cum_log_probs = cum_log_probs;
            bp.cur_ids = cur_ids;
            bp.out_ids = out_ids;
            bp.parent_ids = parent_ids;
            bp.out_stride = Smax;
            bp.seq_len = seq_len;
            bp.finished = finished;
            bp.end_id = end_id;
            bp.advance = advance;
            bp.cache_indirection = cache_ind;
            bp.rope_row_out = rope_row;
            bp.rope_pos_out = rope_pos;
            bp.rope_table = rope;
            bp.rope_half = Dh / 2;
            bp.rope_table_len = rope_len;
            bp.input_lengths = in_len;
            bp.max_input_len = max_in;
            return timed(PC_OTHER, st, [&] { return launch_beam_step(bp, st) ? 1 : 0; });
        }
        GreedyParams gp;
        gp.logits = (tp > 1 || force_comm) ? logits : logits_local;
        gp.batch = B;
        gp.vocab_part = Vr;
        gp.nparts = tp;
        gp.vocab = vocab;
        gp.cur_ids = cur_ids;
        gp.out_ids = out_ids;
        gp.out_stride = Smax;
        gp.seq_len = seq_len;
        gp.finished = finished;
        gp.end_id = end_id;
        gp.advance = advance;
        gp.rope_row_out = rope_row;
        gp.rope_pos_out = rope_pos;
        gp.step_epoch = step_epoch;
        gp.rope_table = rope;
        gp.rope_half = Dh / 2;
        gp.rope_table_len = rope_len;
        gp.input_lengths = in_len;
        gp.max_input_len = max_in;
        if (hidden % 8 == 0)
        {
            gp.emb_table = emb; // the sampler leaves the next step's input row in x (run_decode_step skips its embedding launch)
            gp.x_out = x;
            gp.hidden = hidden;
        }
        return timed(PC_OTHER, st, [&] { return launch_greedy_step(gp, st) ? 1 : 0; });
    }

    // ------------------------------------------------------------------------------------------ generation step
    int run_decode_step(hipStream_t st)
    {
        const int D = hidden;
        const int ok = only_kernel;
        if (ok < 0 && (beam > 1 || D % 8 != 0)) // greedy: the sampler gathered the row already
            RUN(timed(PC_OTHER, st, [&] { return launch_embedding(x, cur_ids, emb, B, D, vocab, st); }));
        const bool r0 = rank == 0;
        // Tensor parallel over the peer-to-peer transport: the layer seam  all-reduce -> residual add -> next RMSNorm (-> quantiser)
        // is ONE launch (kernels/p2p_allreduce.hip fused tail; reference seam: PY/quantization/layer.py:215,377 allreduce,
        // Q/llama_model.py:107-118 adds, PY/layers/normalization.py:33-54).  The row-parallel GEMVs then write their bare partial
        // sums, every rank adds the residual itself (no rank is special), and the consuming GEMV starts from its operand type.
        // RCCL (or TLLM_NO_FUSED_ALLREDUCE=1) keeps the three-stage path: rank 0 carries the residual, the consumers normalise.
        // decided ONCE per step (layers and head must take the same branch) from the transport's own state: the verdict of the
        // caller's validation lives in comm::p2p (tllm_comm_p2p_enable_fused), not in a process-wide cached getenv;
        // TLLM_NO_FUSED_ALLREDUCE=1 stays as the user's A/B switch and is read per step
        const bool fused_ar = tp > 1 && !getenv("TLLM_NO_FUSED_ALLREDUCE") && D % 8 == 0 && comm::p2p::usable_fused(tp, (int64_t) B * D * 2);
        const int ar_quant = !sq ? 0 : (per_token ? 2 : 1);
        const float* ar_rows = (sq && per_token) ? ar_scale : nullptr; // per-token scales behind the fused quantiser
        auto fused_seam = [&](const void* gamma, const float* qscale, int quant) {
            comm::p2p::FusedTail t;
            t.x = x;
            t.gamma = gamma;
            t.eps = eps;
            t.norm_out = ar_norm;
            t.quant = quant;
            t.quant_scale = qscale;
            t.dyn_scale_out = ar_scale;
            return timed(PC_COMM, st, [&] { return comm::p2p::all_reduce_residual_norm(ar_partial, B, D, t, st) ? 1 : 0; });
        };
        for (int li = 0; li < num_layers; ++li)
        {
            Layer& L = layers[li];
            const int pro_norm = !sq ? PRO_RMSNORM : (per_token ? PRO_RMSNORM_QDYN : PRO_RMSNORM_QSTATIC);
            const int pro_q = !sq ? PRO_NONE : (per_token ? PRO_QDYN : PRO_QSTATIC);
            // K1
            const bool taps = debug_taps && ok < 0;
            if (taps)
                HIP_OK(hipMemcpyAsync(tap_ptr(4, li), x, (size_t) B * D * 2, hipMemcpyDeviceToDevice, st));
            if (qkv_attn_fused)
            {
                // K1 + K2 + K3 in one launch (kernels/qkv_attn_fused.hip)
                if (ok < 0 || ok == 1 || ok == 7)
                {
                    FusedQkvAttnParams f;
                    f.K = D;
                    f.num_heads = Hr;
                    f.head_size = Dh;
                    f.x = x;
                    f.gamma = L.ln1;
                    f.eps = eps;
                    f.w = L.qkv.w;
                    f.ldw = L.qkv.ldw;
                    f.scale_col = L.qkv.scale_col;
                    f.per_channel = L.qkv.per_channel;
                    f.woq8 = L.qkv.wtype == W_INT8_WOQ ? 1 : 0;
                    f.fp16_w = L.qkv.wtype == W_FP16 ? 1 : 0;
                    f.woq4 = L.qkv.wtype == W_INT4_WOQ ? 1 : 0;
                    f.act_quant_scale = (per_token || !sq) ? nullptr : L.ln1_scale;
                    f.act_dequant_scale = (per_token || !sq) ? nullptr : L.qkv.act_scale;
                    f.int8_kv = int8_kv;
                    f.max_seq_len = Smax;
                    f.inv_sqrt_dh = 1.f / sqrtf((float) Dh);
                    f.kv_cache = L.kv;
                    f.sequence_length = seq_len;
                    f.masked_tokens = masked;
                    f.kv_scale_orig_quant = L.kv_oq;
                    f.kv_scale_quant_orig = L.kv_qo;
                    f.rope_row = rope_row;
                    f.xchg = fused_xchg;
                    f.epoch = step_epoch;
                    f.tag_mul = (uint32_t) num_layers + 1;
                    f.tag_add = (uint32_t) li + 1;
                    if (ok >= 0) // eager launches outside a step: the epoch does not advance between them
                        f.tag_host = 1u + (++timing_tag & 0x3fffffffu); // (the kernel sets the top bit: host tags never meet step tags)
                    f.error = fused_err;
                    if (fused_max_spins >= 0)
                        f.max_spins = fused_max_spins;
                    f.qkv_out = qkv;
                    f.out = ctx;
                    if (sq && !per_token)
                    {
                        f.out_q8 = ctx_q8;
                        f.out_quant_scale = L.attn_qscale;
                    }
                    f.x_pro_out = taps ? tap_ptr(0, li) : nullptr;
                    f.timing = fused_timing;
                    // (kernel-timing id 1 measures the two-stage form, id 7 the launch exactly as the step runs it; with the stage clock
                    //  on, the three-stage form is what it looks at - x is overwritten by every timed launch: the timer restores it)
                    if (o_fused && (ok < 0 || ok == 7 || fused_timing))
                    {
                        f.o_w = L.dense.w;
                        f.o_ldw = L.dense.ldw;
                        f.o_n = L.dense.N;
                        f.o_per_channel = L.dense.per_channel;
                        f.o_scale_col = L.dense.scale_col;
                        f.o_scale_row = L.dense.act_scale;
                        f.x_out = x;
                    }
                    RUN(timed(PC_ATTENTION, st, [&] { return launch_qkv_attn_fused(f, st) ? 1 : 0; }));
                }
            }
            else if (ok < 0 || ok == 1)
            {
                int rc1;
                if (fused_ar && li > 0)
                {
                    // the previous layer's seam left input_layernorm's output (quantised for SmoothQuant) in ar_norm
                    if (taps)
                        HIP_OK(hipMemcpyAsync(tap_ptr(0, li), ar_norm, (size_t) B * D * (sq ? 1 : 2), hipMemcpyDeviceToDevice, st));
                    rc1 = gemv(L.qkv, B, PRO_NONE, EPI_NONE, ar_norm, D, nullptr, nullptr, nullptr, nullptr, qkv, 3 * Dr, DT_HALF,
                        nullptr, st, ar_rows);
                }
                else
                {
                    tap_dst = taps ? tap_ptr(0, li) : nullptr;
                    rc1 = gemv(L.qkv, B, pro_norm, EPI_NONE, x, D, L.ln1, L.ln1_scale, nullptr, nullptr, qkv, 3 * Dr, DT_HALF,
                        nullptr, st);
                    tap_dst = nullptr;
                }
                RUN(rc1);
            }
            // K2/K3
            MmhaParams m;
            m.batch = B;
            m.num_heads = Hr;
            m.head_size = Dh;
            m.rotary_dim = Dh;
            m.neox = neox;
            m.inv_sqrt_dh = 1.f / sqrtf((float) Dh);
            m.int8_kv = int8_kv;
            m.max_seq_len = Smax;
            m.max_input_len = max_in;
            m.qkv = qkv;
            m.kv_cache = L.kv;
            m.sequence_length = seq_len;
            m.input_lengths = in_len;
            m.masked_tokens = masked;
            m.timestep_host = -1; // device-resident step state: one graph serves every step
            m.kv_scale_orig_quant = L.kv_oq;
            m.kv_scale_quant_orig = L.kv_qo;
            m.rope_table = rope;
            m.rope_table_len = rope_len;
            m.rope_row = rope_row;
            m.cache_indirection = beam > 1 ? cache_ind : nullptr;
            m.beam_width = beam;
            m.block_pointers = L.kv_table;
            m.tokens_per_block = tokens_per_block;
            m.max_blocks_per_seq = max_blocks;
            m.rows_per_group = attn_nit;
            const bool tail_q8 = attn_tail && sq && !per_token;
            if (attn_tail)
            {
                m.tail_tickets = attn_tickets;
                if (tail_q8)
                {
                    m.tail_out_q8 = ctx_q8;
                    m.tail_quant_scale = L.attn_qscale;
                }
            }
            m.out = ctx;
            m.workspace = mmha_ws;
            if (!qkv_attn_fused && (ok < 0 || ok == 2))
                RUN(timed(PC_ATTENTION, st, [&] { return launch_mmha(m, st); }));
            // K4: x <- x + O(ctx)     (TP: rank 0 carries the residual into the all-reduce)
            const int pro_o = pro_q;
            if (qkv_attn_fused && o_fused && ok < 0)
            {
                // K4 ran as the third stage of the fused launch: x already holds x + O(ctx)
                if (taps)
                    HIP_OK(hipMemcpyAsync(tap_ptr(1, li), sq ? (const void*) ctx_q8 : ctx, (size_t) B * Dr * (sq ? 1 : 2), hipMemcpyDeviceToDevice,
                        st));
            }
            else if (ok < 0 || ok == 4)
            {
                if (taps)
                {
                    if (tail_q8) // the attention launch left the quantised operand itself
                        HIP_OK(hipMemcpyAsync(tap_ptr(1, li), ctx_q8, (size_t) B * Dr, hipMemcpyDeviceToDevice, st));
                    else if (pro_o == PRO_NONE) // nothing is transformed in the prologue: the input itself is the tap
                        HIP_OK(hipMemcpyAsync(tap_ptr(1, li), ctx, (size_t) B * Dr * 2, hipMemcpyDeviceToDevice, st));
                    else
                        tap_dst = tap_ptr(1, li);
                }
                // (tail merge + static SmoothQuant: the attention launch left the int8 operand itself - no prologue at all)
                const void* o_in = tail_q8 ? ctx_q8 : ctx;
                const int pro_o2 = tail_q8 ? (int) PRO_NONE : pro_o;
                const int rc4 = fused_ar
                    ? gemv(L.dense, B, pro_o2, EPI_NONE, o_in, Dr, nullptr, L.attn_qscale, nullptr, nullptr, ar_partial, D, DT_HALF, nullptr, st)
                    : gemv(L.dense, B, pro_o2, (tp == 1 || r0) ? EPI_RESIDUAL : EPI_NONE, o_in, Dr, nullptr, L.attn_qscale, x, nullptr, x, D,
                          DT_HALF, nullptr, st);
                tap_dst = nullptr;
                RUN(rc4);
            }
            if (ok < 0)
            {
                if (fused_ar) // x <- x + sum_r O_r(ctx_r);  ar_norm <- post_layernorm(x) [-> int8]
                    RUN(fused_seam(L.ln2, L.ln2_scale, ar_quant));
                else
                    RUN(allreduce(x, (int64_t) B * D, st));
            }
            // K5
            const bool q_inter = sq && !per_token;
            const bool mlp_one = mlp_fused_dec && !fused_ar && (ok < 0 || ok == 8);
            if (mlp_one)
            {
                // K5 + K6 in one launch (kernels/mlp_fused.hip)
                FusedMlpParams f;
                f.K = D;
                f.I = Ir;
                f.N = L.proj.N;
                f.x = x;
                f.x_out = x;
                f.gamma = L.ln2;
                f.eps = eps;
                f.act_quant = L.ln2_scale;
                f.w_fc = L.fc.w;
                f.w_gate = L.gate.w;
                f.ldw = L.fc.ldw;
                f.scale_fc = L.fc.scale_col;
                f.scale_gate = L.gate.scale_col;
                f.per_channel = L.fc.per_channel;
                f.row_fc = L.fc.act_scale;
                f.row_gate = L.gate.act_scale ? L.gate.act_scale : L.fc.act_scale;
                f.out_quant = L.mlp_qscale;
                f.inter = taps ? q8 : nullptr; // (the compact row: only a tap reads it)
                f.w_proj = L.proj.w;
                f.ldw_proj = L.proj.ldw;
                f.scale_proj = L.proj.scale_col;
                f.per_channel_proj = L.proj.per_channel;
                f.row_proj = L.proj.act_scale;
                f.flags = mlp_flags;
                f.error = fused_err;
                if (fused_max_spins >= 0)
                    f.max_spins = fused_max_spins;
                f.x_pro_out = taps ? tap_ptr(2, li) : nullptr;
                f.timing = mlp_timing;
                RUN(timed(PC_GEMV_LAYER, st, [&] { return launch_mlp_fused(f, st) ? 1 : 0; }));
                if (taps)
                    HIP_OK(hipMemcpyAsync(tap_ptr(3, li), q8, (size_t) B * Ir, hipMemcpyDeviceToDevice, st));
            }
            if (!mlp_one && (ok < 0 || ok == 5))
            {
                int rc5;
                if (fused_ar)
                {
                    if (taps)
                        HIP_OK(hipMemcpyAsync(tap_ptr(2, li), ar_norm, (size_t) B * D * (sq ? 1 : 2), hipMemcpyDeviceToDevice, st));
                    rc5 = gemv(L.fc, B, PRO_NONE, q_inter ? EPI_SWIGLU_QSTATIC : EPI_SWIGLU, ar_norm, D, nullptr, nullptr, nullptr,
                        L.mlp_qscale, q_inter ? (void*) q8 : inter_buf, Ir, q_inter ? DT_INT8 : DT_HALF, &L.gate, st, ar_rows);
                }
                else
                {
                    tap_dst = taps ? tap_ptr(2, li) : nullptr;
                    rc5 = gemv(L.fc, B, pro_norm, q_inter ? EPI_SWIGLU_QSTATIC : EPI_SWIGLU, x, D, L.ln2, L.ln2_scale, nullptr,
                        L.mlp_qscale, q_inter ? (void*) q8 : inter_buf, Ir, q_inter ? DT_INT8 : DT_HALF, &L.gate, st);
                    tap_dst = nullptr;
                }
                RUN(rc5);
            }
            // K6
            if (!mlp_one && (ok < 0 || ok == 6))
            {
                const int pro6 = q_inter ? PRO_NONE : pro_q;
                if (taps)
                {
                    if (pro6 == PRO_NONE) // the SwiGLU epilogue left the operand (s8 behind the static quantiser, fp16 otherwise)
                        HIP_OK(hipMemcpyAsync(tap_ptr(3, li), q_inter ? (const void*) q8 : inter_buf, (size_t) B * Ir * (q_inter ? 1 : 2),
                            hipMemcpyDeviceToDevice, st));
                    else
                        tap_dst = tap_ptr(3, li);
                }
                const int rc6 = fused_ar
                    ? gemv(L.proj, B, pro6, EPI_NONE, q_inter ? (const void*) q8 : inter_buf, Ir, nullptr, L.mlp_qscale, nullptr, nullptr,
                          ar_partial, D, DT_HALF, nullptr, st)
                    : gemv(L.proj, B, pro6, (tp == 1 || r0) ? EPI_RESIDUAL : EPI_NONE, q_inter ? (const void*) q8 : inter_buf, Ir, nullptr,
                          L.mlp_qscale, x, nullptr, x, D, DT_HALF, nullptr, st);
                tap_dst = nullptr;
                RUN(rc6);
            }
            if (ok < 0)
            {
                if (fused_ar)
                {
                    // x <- x + sum_r proj_r(...);  ar_norm <- the NEXT layer's input_layernorm(x) [-> int8], or ln_f(x) for the head
                    const bool last = li + 1 == num_layers;
                    RUN(fused_seam(last ? lnf : layers[li + 1].ln1, last ? nullptr : layers[li + 1].ln1_scale, last ? 0 : ar_quant));
                }
                else
                    RUN(allreduce(x, (int64_t) B * D, st));
            }
        }
        if (ok >= 0)
            return 0;
        RUN(run_head(fused_ar ? ar_norm : x, B, st, fused_ar));
        RUN(run_sampler(1, st));
        return 0;
    }
};

// ================================================================================================
// C API
// ================================================================================================
extern "C" {

tllm_session_t tllm_session_create(const char* config_text)
{
    if (!config_text)
    {
        set_error("tllm_session_create: null config");
        return nullptr;
    }
    auto s = std::make_unique<tllm_session>();
    std::istringstream in(config_text);
    std::string line;
    std::map<std::string, std::string> kv;
    while (std::getline(in, line))
    {
        const size_t eq = line.find('=');
        if (eq == std::string::npos)
            continue;
        auto trim = [](std::string v) {
            const size_t a = v.find_first_not_of(" \t\r");
            const size_t b = v.find_last_not_of(" \t\r");
            return a == std::string::npos ? std::string() : v.substr(a, b - a + 1);
        };
        kv[trim(line.substr(0, eq))] = trim(line.substr(eq + 1));
    }
    auto geti = [&](const char* k, int def) { return kv.count(k) ? atoi(kv[k].c_str()) : def; };
    s->num_layers = geti("num_layers", 0);
    s->num_heads = geti("num_heads", 0);
    s->hidden = geti("hidden_size", 0);
    s->inter = geti("inter_size", 0);
    s->vocab = geti("vocab_size", 0);
    s->max_pos = geti("max_position_embeddings", 2048);
    s->tp = geti("tp_size", 1);
    s->rank = geti("tp_rank", 0);
    s->quant_mode = geti("quant_mode", 0);
    s->neox = geti("neox_rotary_style", 1);
    s->force_comm = geti("force_comm", 0) != 0;
    s->no_comm = geti("no_comm", 0) != 0;
    s->debug_taps = geti("debug_taps", 0) != 0;
    s->fuse_qkv_cfg = geti("fuse_qkv_attention", -1);
    s->fuse_o_cfg = geti("fuse_o_projection", -1);
    s->fuse_mlp_cfg = geti("fuse_mlp", 0);
    s->fused_max_spins = geti("fused_max_spins", -1);
    s->dual_mlp_cfg = geti("dual_mlp_gemm", -1);
    s->fused_timeline = geti("fused_timeline", 0) != 0;
    if (kv.count("gemm_tactics") && !kv["gemm_tactics"].empty())
    {
        // the prefill GEMM kernels the builder's on-device profile chose (engine header; Builder.build_engine)
        if (gemm_tactics_import(kv["gemm_tactics"].c_str()) < 0)
            return nullptr;
    }
    s->packed = geti("remove_input_padding", 0) != 0;
    s->paged_kv = geti("paged_kv_cache", 0) != 0;
    s->tokens_per_block = geti("tokens_per_block", 64);
    if (s->paged_kv && (s->tokens_per_block < 1 || (s->tokens_per_block & (s->tokens_per_block - 1))))
    {
        set_error("tllm_session_create: tokens_per_block must be a power of two (got %d)", s->tokens_per_block);
        return nullptr;
    }
    if (kv.count("rms_norm_eps"))
        s->eps = (float) atof(kv["rms_norm_eps"].c_str());
    if (kv.count("weight_only_precision"))
        s->wo_precision = kv["weight_only_precision"];
    if (kv.count("network_json"))
        s->network_json = kv["network_json"];
    if (s->num_layers <= 0 || s->num_heads <= 0 || s->hidden <= 0 || s->inter <= 0 || s->vocab <= 0 || s->tp < 1
        || s->rank < 0 || s->rank >= s->tp)
    {
        set_error("tllm_session_create: num_layers/num_heads/hidden_size/inter_size/vocab_size/tp_size/tp_rank invalid");
        return nullptr;
    }
    if (s->hidden % s->num_heads || s->num_heads % s->tp || s->inter % s->tp)
    {
        set_error("tllm_session_create: heads must divide hidden, tp must divide heads and inter_size");
        return nullptr;
    }
    s->Dh = s->hidden / s->num_heads;
    s->Hr = s->num_heads / s->tp;
    s->Dr = s->Hr * s->Dh;
    s->Ir = s->inter / s->tp;
    // vocab padded to a multiple of tp (PY/_utils.py:194-195, Q/llama_model.py:244)
    s->Vr = (s->vocab + s->tp - 1) / s->tp;
    const int qm = s->quant_mode;
    s->sq = (qm & QM_ACTIVATIONS) && (qm & QM_INT8_WEIGHTS);
    s->woq = !s->sq && (qm & (QM_INT8_WEIGHTS | QM_INT4_WEIGHTS));
    s->int8_kv = qm & QM_INT8_KV;
    s->per_token = qm & QM_PER_TOKEN;
    s->per_channel = qm & QM_PER_CHANNEL;
    if (s->woq)
        s->wtype = (qm & QM_INT4_WEIGHTS) ? W_INT4_WOQ : W_INT8_WOQ;
    else if (s->sq)
        s->wtype = W_INT8_SQ;
    for (int i = 0; i < s->tp; ++i)
        s->group.push_back(i);
    return s.release();
}

int32_t tllm_session_set_tensor(tllm_session_t s, const char* name, int32_t dtype, const int64_t* dims, int32_t nbDims,
    const void* data, int32_t location)
{
    if (!s || !name || !dims || !data || nbDims < 0 || nbDims > 8)
    {
        set_error("tllm_session_set_tensor: bad arguments");
        return 1;
    }
    TensorRec t;
    t.dtype = dtype;
    t.dims.assign(dims, dims + nbDims);
    t.bytes = (size_t) t.numel() * dtype_bytes(dtype);
    if (location == 0)
    {
        HIP_OK(hipMalloc(&t.dev, t.bytes ? t.bytes : 16));
        t.owned = true;
        HIP_OK(hipMemcpy(t.dev, data, t.bytes, hipMemcpyHostToDevice));
    }
    else
        t.dev = const_cast<void*>(data);
    auto it = s->tensors.find(name);
    if (it != s->tensors.end() && it->second.owned && it->second.dev)
        (void) hipFree(it->second.dev);
    s->tensors[name] = t;
    s->finalized = false;
    return 0;
}

int32_t tllm_session_finalize(tllm_session_t s)
{
    if (!s)
        return 1;
    const int D = s->hidden;
    {
        const TensorRec* t = s->find("vocab_embedding.weight");
        RUN(s->want(t, "vocab_embedding.weight", TLLM_HALF, (int64_t) s->vocab * D));
        s->emb = t->dev;
        t = s->find("ln_f.weight");
        RUN(s->want(t, "ln_f.weight", TLLM_HALF, D));
        s->lnf = t->dev;
        // lm_head stays fp16 in every quantisation mode (Q/quant.py:58)
        RUN(s->resolve_linear("lm_head", s->Vr, D, s->head, true));
    }
    s->layers.assign(s->num_layers, Layer());
    for (int i = 0; i < s->num_layers; ++i)
    {
        Layer& L = s->layers[i];
        const std::string p = "layers." + std::to_string(i) + ".";
        const TensorRec* t = s->find(p + "input_layernorm.weight");
        RUN(s->want(t, p + "input_layernorm.weight", TLLM_HALF, D));
        L.ln1 = t->dev;
        t = s->find(p + "post_layernorm.weight");
        RUN(s->want(t, p + "post_layernorm.weight", TLLM_HALF, D));
        L.ln2 = t->dev;
        RUN(s->resolve_linear(p + "attention.qkv", 3 * s->Dr, D, L.qkv));
        RUN(s->resolve_linear(p + "attention.dense", D, s->Dr, L.dense));
        RUN(s->resolve_linear(p + "mlp.fc", s->Ir, D, L.fc));
        RUN(s->resolve_linear(p + "mlp.gate", s->Ir, D, L.gate));
        RUN(s->resolve_linear(p + "mlp.proj", D, s->Ir, L.proj));
        if (s->sq && !s->per_token)
        {
            RUN(s->scalar_f32(p + "input_layernorm.scale_to_int", &L.ln1_scale));
            RUN(s->scalar_f32(p + "post_layernorm.scale_to_int", &L.ln2_scale));
            RUN(s->scalar_f32(p + "attention.quantization_scaling_factor", &L.attn_qscale));
            RUN(s->scalar_f32(p + "mlp.quantization_scaling_factor", &L.mlp_qscale));
        }
        if (s->int8_kv)
        {
            RUN(s->scalar_f32(p + "attention.kv_orig_quant_scale", &L.kv_oq));
            RUN(s->scalar_f32(p + "attention.kv_quant_orig_scale", &L.kv_qo));
        }
    }
    if (s->tp > 1 && !s->no_comm && !comm::has_comm(s->group) && !comm::p2p::attached())
    {
        set_error("session: tp_size=%d but no communicator registered (tllm_comm_init_rank / tllm_comm_p2p_attach)", s->tp);
        return 1;
    }
    s->finalized = true;
    return 0;
}

namespace
{
struct EngineEntry
{
    std::string name;
    int32_t dtype, nd;
    int64_t dims[8];
    uint64_t nbytes, offset;
};

// "TLLMENG1" | u64 header length | header text | u64 tensor count | table | 64-byte aligned data (tensorrt_llm/builder.py)
int parse_engine(const void* engine, size_t nbytes, std::string& cfg, std::vector<EngineEntry>& ents, size_t& data0)
{
    const char* p = static_cast<const char*>(engine);
    auto fail = [](const char* why) {
        set_error("engine: %s", why);
        return 1;
    };
    if (!p || nbytes < 24 || std::memcmp(p, "TLLMENG1", 8) != 0)
        return fail("not a TLLMENG1 engine");
    size_t off = 8;
    auto rd64 = [&](uint64_t* v) {
        if (off + 8 > nbytes)
            return false;
        std::memcpy(v, p + off, 8);
        off += 8;
        return true;
    };
    uint64_t hlen = 0, nt = 0;
    if (!rd64(&hlen) || hlen > nbytes || off + hlen > nbytes)
        return fail("truncated header");
    cfg.assign(p + off, p + off + hlen);
    off += hlen;
    if (!rd64(&nt) || nt > nbytes / 24)
        return fail("truncated tensor table");
    ents.assign(nt, EngineEntry());
    for (auto& e : ents)
    {
        uint32_t nl = 0;
        if (off + 4 > nbytes)
            return fail("truncated tensor table");
        std::memcpy(&nl, p + off, 4);
        off += 4;
        if (nl > nbytes || off + nl + 8 > nbytes)
            return fail("truncated tensor table");
        e.name.assign(p + off, p + off + nl);
        off += nl;
        std::memcpy(&e.dtype, p + off, 4);
        std::memcpy(&e.nd, p + off + 4, 4);
        off += 8;
        if (e.nd < 0 || e.nd > 8 || off + 8 * (size_t) e.nd + 16 > nbytes)
            return fail("bad tensor entry");
        std::memcpy(e.dims, p + off, 8 * (size_t) e.nd);
        off += 8 * (size_t) e.nd;
        std::memcpy(&e.nbytes, p + off, 8);
        std::memcpy(&e.offset, p + off + 8, 8);
        off += 16;
    }
    data0 = (off + 63) / 64 * 64;
    for (auto& e : ents)
        if (e.offset > nbytes || e.nbytes > nbytes || data0 + e.offset + e.nbytes > nbytes)
            return fail("tensor data out of range");
    return 0;
}

// The engine's traced network against the schedule a session of this configuration executes (runtime/engine_check.h).
int verify_engine_network(tllm_session_t s, const std::vector<EngineEntry>& ents)
{
    if (s->network_json.empty())
    {
        set_error("engine: no network_json in the header - not an engine built by tensorrt_llm.Builder.build_engine");
        return 1;
    }
    runtime::ScheduleDesc d;
    d.num_layers = s->num_layers;
    d.heads_per_rank = s->Hr;
    d.head_size = s->Dh;
    d.tp = s->tp;
    d.eps = s->eps;
    d.sq = s->sq;
    d.per_token = s->per_token;
    d.woq = s->woq;
    d.int4 = s->wtype == W_INT4_WOQ;
    d.int8_kv = s->int8_kv;
    d.paged = s->paged_kv;
    d.packed = s->packed;
    d.neox = s->neox != 0;
    // has_per_channel_scaling of every SmoothQuant GEMM = what the scale tensor the engine carries implies (a "per tensor"
    // QKV scale is stored as one factor per channel: examples/llama_quant/weight.py)
    for (const char* n : {"attention.qkv", "attention.dense", "mlp.fc", "mlp.gate", "mlp.proj"})
    {
        int pc = s->per_channel ? 1 : 0;
        const std::string want = std::string("layers.0.") + n + ".per_channel_scale";
        for (auto& e : ents)
            if (e.name == want)
            {
                int64_t numel = 1;
                for (int i = 0; i < e.nd; ++i)
                    numel *= e.dims[i];
                pc = numel > 1 ? 1 : 0;
            }
        d.per_channel.push_back(pc);
    }
    std::string why;
    if (runtime::verify_network(s->network_json, d, why))
    {
        set_error("engine: %s", why.c_str());
        return 1;
    }
    return 0;
}
} // namespace

int32_t tllm_engine_verify(const void* engine, size_t nbytes)
{
    std::string cfg;
    std::vector<EngineEntry> ents;
    size_t data0 = 0;
    RUN(parse_engine(engine, nbytes, cfg, ents, data0));
    tllm_session_t s = tllm_session_create(cfg.c_str());
    if (!s)
        return 1;
    const int rc = verify_engine_network(s, ents);
    tllm_session_destroy(s);
    return rc;
}

tllm_session_t tllm_session_load_engine(const void* engine, size_t nbytes)
{
    std::string cfg;
    std::vector<EngineEntry> ents;
    size_t data0 = 0;
    if (parse_engine(engine, nbytes, cfg, ents, data0))
        return nullptr;
    tllm_session_t s = tllm_session_create(cfg.c_str());
    if (!s)
        return nullptr;
    // the engine is what was defined: refuse a traced network that is not the schedule this session would run
    if (verify_engine_network(s, ents))
    {
        tllm_session_destroy(s);
        return nullptr;
    }
    const char* p = static_cast<const char*>(engine);
    for (auto& e : ents)
    {
        if (tllm_session_set_tensor(s, e.name.c_str(), e.dtype, e.dims, e.nd, p + data0 + e.offset, 0))
        {
            tllm_session_destroy(s);
            return nullptr;
        }
    }
    if (tllm_session_finalize(s))
    {
        tllm_session_destroy(s);
        return nullptr;
    }
    return s;
}

int32_t tllm_session_setup(tllm_session_t s, int32_t batch_size, int32_t max_input_len, int32_t max_new_tokens)
{
    return tllm_session_setup_beam(s, batch_size, 1, max_input_len, max_new_tokens);
}

int32_t tllm_session_setup_beam(tllm_session_t s, int32_t batch_size, int32_t beam_width, int32_t max_input_len,
    int32_t max_new_tokens)
{
    if (!s || !s->finalized)
    {
        set_error("tllm_session_setup: session not finalized");
        return 1;
    }
    if (batch_size < 1 || max_input_len < 1 || max_new_tokens < 0 || beam_width < 1 || beam_width > 8)
    {
        set_error("tllm_session_setup: bad sizes (batch %d, beam width %d in [1, 8], input %d, new %d)", batch_size, beam_width,
            max_input_len, max_new_tokens);
        return 1;
    }
    if (beam_width > 1 && (size_t) beam_width * (max_input_len + max_new_tokens) * sizeof(int32_t) > 96 * 1024)
    {
        // the device-side beam step stages the cache-indirection rows it re-parents in LDS (pointwise.hip beam_step_kernel)
        set_error("tllm_session_setup: beam_width %d x max_seq_len %d exceeds the beam step's LDS staging (24576 int32)", beam_width,
            max_input_len + max_new_tokens);
        return 1;
    }
    s->free_runtime();
    s->Bc = batch_size;
    s->beam = beam_width;
    s->B = batch_size * beam_width;
    s->max_in = max_input_len;
    s->max_new = max_new_tokens;
    s->Smax = max_input_len + max_new_tokens; // generation.py:450-461
    const int B = s->B, S = s->max_in, D = s->hidden, Smax = s->Smax;
    const int Bc = s->Bc;
    const size_t M = std::max((size_t) Bc * S, (size_t) B); // prompt rows; the generation phase needs B
    const size_t kv_esz = s->int8_kv ? 1 : 2;
    size_t kv_bytes = (size_t) B * 2 * s->Hr * Smax * s->Dh * kv_esz;
    const int T = s->tokens_per_block;
    s->max_blocks = s->paged_kv ? (Smax + T - 1) / T : 0;
    const size_t nblocks = (size_t) B * s->max_blocks, blk_bytes = (size_t) s->Hr * T * s->Dh * kv_esz;
    if (s->paged_kv)
        kv_bytes = 2 * nblocks * blk_bytes;
    s->kv_elems = kv_bytes / kv_esz;
    for (auto& L : s->layers)
    {
        RUN(s->dalloc(&L.kv, kv_bytes));
        HIP_OK(hipMemset(L.kv, 0, kv_bytes));
        L.kv_table = nullptr;
        if (s->paged_kv)
        {
            // pool [2, blocks, Hr, T, Dh] (K half then V half, like the reference's BlocksManager:
            // kv_cache_manager.py:84-96); logical block j of sequence bb is pool block j * B + bb, so consecutive time
            // blocks of one sequence are NOT contiguous
            std::vector<int64_t> table((size_t) B * 2 * s->max_blocks);
            const int64_t base = reinterpret_cast<int64_t>(L.kv);
            for (int bb = 0; bb < B; ++bb)
                for (int j = 0; j < s->max_blocks; ++j)
                {
                    const int64_t blk = (int64_t) j * B + bb;
                    table[((size_t) bb * 2 + 0) * s->max_blocks + j] = base + blk * (int64_t) blk_bytes;
                    table[((size_t) bb * 2 + 1) * s->max_blocks + j] = base + ((int64_t) nblocks + blk) * (int64_t) blk_bytes;
                }
            int64_t* dev = nullptr;
            RUN(s->dalloc(&dev, table.size() * 8));
            HIP_OK(hipMemcpy(dev, table.data(), table.size() * 8, hipMemcpyHostToDevice));
            L.kv_table = dev;
        }
    }
    RUN(s->dalloc(&s->x, M * D * 2));
    RUN(s->dalloc(&s->tmp, M * D * 2));
    RUN(s->dalloc(&s->qkv, M * 3 * s->Dr * 2));
    RUN(s->dalloc(&s->ctx, M * s->Dr * 2));
    RUN(s->dalloc(&s->g, M * s->Ir * 2));
    RUN(s->dalloc(&s->u, M * s->Ir * 2));
    RUN(s->dalloc(&s->inter_buf, M * s->Ir * 2));
    RUN(s->dalloc(&s->q8, M * (size_t) std::max(D, s->Ir)));
    RUN(s->dalloc(&s->qscale, M * 4));
    RUN(s->dalloc(&s->logits_local, (size_t) B * s->Vr * 4));
    RUN(s->dalloc(&s->logits, (size_t) B * s->Vr * s->tp * 4));
    RUN(s->dalloc(&s->last_hidden, (size_t) B * D * 2));
    RUN(s->dalloc(&s->ctx_ws, context_attention_workspace_size(Bc, s->Hr, s->Dh, S) + 256));
    if ((size_t) Bc * S >= 32)
        RUN(s->profile_prefill_gemms(Bc * S));
    RUN(s->dalloc(&s->ar_partial, (size_t) B * D * 2));
    RUN(s->dalloc(&s->ar_norm, (size_t) B * D * 2));
    RUN(s->dalloc(&s->ar_scale, (size_t) B * 4));
    for (int w = 0; w < tllm_session::kTaps; ++w)
    {
        s->tap_buf[w] = nullptr;
        if (s->debug_taps)
        {
            const size_t nb = (size_t) s->num_layers * B * s->tap_width(w) * 2;
            RUN(s->dalloc(&s->tap_buf[w], nb));
            HIP_OK(hipMemset(s->tap_buf[w], 0, nb));
        }
    }
    RUN(s->dalloc(&s->cu_dev, (size_t) (Bc + 1) * 4));
    RUN(s->dalloc(&s->last_rows, (size_t) Bc * 4));
    RUN(s->dalloc(&s->mmha_ws, mmha_workspace_size(B, s->Hr, s->Dh, Smax) + 256));
    HIP_OK(hipMemset(s->mmha_ws, 0, mmha_workspace_size(B, s->Hr, s->Dh, Smax) + 256));
    RUN(s->dalloc(&s->ids_in, M * 4));
    RUN(s->dalloc(&s->cur_ids, (size_t) B * 4));
    RUN(s->dalloc(&s->out_ids, (size_t) B * Smax * 4));
    RUN(s->dalloc(&s->seq_len, (size_t) B * 4));
    RUN(s->dalloc(&s->in_len, (size_t) B * 4));
    RUN(s->dalloc(&s->last_tok, (size_t) Bc * 4));
    RUN(s->dalloc(&s->in_len_ctx, (size_t) Bc * 4));
    s->cum_log_probs = nullptr;
    s->parent_ids = s->cache_ind = nullptr;
    if (s->beam > 1)
    {
        RUN(s->dalloc(&s->cum_log_probs, (size_t) B * 4));
        RUN(s->dalloc(&s->parent_ids, (size_t) B * Smax * 4));
        RUN(s->dalloc(&s->cache_ind, (size_t) B * Smax * 4));
    }
    RUN(s->dalloc(&s->finished, (size_t) B * 4));
    RUN(s->dalloc(&s->masked, (size_t) B * Smax * 4));
    s->rope = plugins::rope_table(s->Dh, Smax > s->max_pos ? Smax : s->max_pos, &s->rope_len);
    if (!s->rope)
        return 1;
    RUN(s->dalloc(&s->rope_row, (size_t) B * s->Dh * sizeof(float)));
    HIP_OK(hipMemset(s->rope_row, 0, (size_t) B * s->Dh * sizeof(float)));
    RUN(s->dalloc(&s->rope_pos, (size_t) B * 4));
    HIP_OK(hipMemset(s->rope_pos, 0, (size_t) B * 4));
    // The split-KV merge runs inside the attention launch, by the last split of a head to arrive (mmha_decode.hip step 6; up to
    // 16 partials: 16-row splits cover 4096 cache slots at head size 128); 12 rows per lane group while that needs <= 8 partials
    // (7 x 32 instead of 5 x 32 workgroups at the 1024-token bench context).  Beyond 16 partials: the finest split with its own
    // combine launch.
    {
        int tc = 0, ns = 0;
        size_t off = 0;
        s->attn_tail = false;
        s->attn_nit = 4;
        s->attn_tickets = nullptr;
        s->ctx_q8 = nullptr;
        if (mmha_split_layout(s->Dh, Smax, 16, B, s->Hr, &tc, &ns, &off) == 0 && ns <= 16)
        {
            s->attn_nit = 16;
            if (mmha_split_layout(s->Dh, Smax, 12, B, s->Hr, &tc, &ns, &off) == 0 && ns <= 8)
                s->attn_nit = 12;
            s->attn_tail = true;
            RUN(s->dalloc(reinterpret_cast<void**>(&s->attn_tickets), (size_t) B * s->Hr * 4));
            HIP_OK(hipMemset(s->attn_tickets, 0, (size_t) B * s->Hr * 4));
            RUN(s->dalloc(&s->ctx_q8, (size_t) B * s->Dr));
        }
        if (mmha_split_layout(s->Dh, Smax, s->attn_nit, B, s->Hr, &tc, &ns, &off))
        {
            set_error("session: unsupported head size %d", s->Dh);
            return 1;
        }
        s->attn_tchunk = tc;
        s->attn_ns = ns;
        s->attn_o_off = off;
    }
    // the one-launch QKV projection + attention: batch 1, greedy, linear cache, SmoothQuant weights, the geometry the kernel is
    // built for (K = 4096, head size 128, heads x 8 workgroups = one per CU), NeoX rotary over the whole head
    RUN(s->dalloc(&s->step_epoch, 64));
    HIP_OK(hipMemset(s->step_epoch, 0, 64));
    s->fused_err = s->step_epoch + 8;
    s->qkv_attn_fused = false;
    s->o_fused = false;
    s->fused_xchg = nullptr;
    s->mlp_fused_dec = false;
    s->mlp_flags = nullptr;
    s->mlp_timing = nullptr;
    if (s->fuse_mlp_cfg > 0 && B == 1 && s->beam == 1 && s->tp == 1 && s->sq && !s->per_token && !s->layers.empty()
        && mlp_fused_serves(D, s->Ir, D))
    {
        bool ok = true;
        for (auto& L : s->layers)
            ok = ok && L.fc.wtype == W_INT8_SQ && L.gate.wtype == W_INT8_SQ && L.proj.wtype == W_INT8_SQ && L.fc.K == D && L.gate.K == D
                && L.fc.N == s->Ir && L.gate.N == s->Ir && L.fc.ldw == L.gate.ldw && L.fc.per_channel == L.gate.per_channel
                && L.proj.N == D && L.proj.K == s->Ir && L.fc.scale_col && L.gate.scale_col && L.proj.scale_col && L.fc.act_scale
                && L.proj.act_scale && L.ln2 && L.ln2_scale && L.mlp_qscale && L.fc.ldw % 16 == 0 && L.proj.ldw % 16 == 0;
        if (ok)
        {
            RUN(s->dalloc(&s->mlp_flags, mlp_fused_flag_bytes()));
            HIP_OK(hipMemset(s->mlp_flags, 0, mlp_fused_flag_bytes()));
            s->mlp_fused_dec = true;
            if (s->fused_timeline)
            {
                RUN(s->dalloc(&s->mlp_timing, (size_t) 256 * 16 * 8));
                HIP_OK(hipMemset(s->mlp_timing, 0, (size_t) 256 * 16 * 8));
            }
        }
    }
    // (SmoothQuant, or - r05 - weight-only int8: the same 4 KB weight rows against the normalised fp16 row)
    // (... or - r06 - fp16: rows of 8 KB, two tiles per row pair; BASELINE.json configs[1])
    const bool woq8_all = !s->sq && !s->layers.empty() && s->layers[0].qkv.wtype == W_INT8_WOQ;
    const bool fp16_all = !s->sq && !s->layers.empty() && s->layers[0].qkv.wtype == W_FP16;
    const bool woq4_all = !s->sq && !s->layers.empty() && s->layers[0].qkv.wtype == W_INT4_WOQ;
    const int wkind = s->sq ? 0 : (woq8_all ? 1 : (fp16_all ? 2 : 3)); // qkv_attn_fused_serves' weight_kind
    if (s->fuse_qkv_cfg != 0 && s->attn_tail && B == 1 && s->beam == 1 && !s->paged_kv && s->tp == 1
        && (s->sq || woq8_all || fp16_all || woq4_all)
        && s->neox && qkv_attn_fused_serves(D, s->Hr, s->Dh, Smax, s->int8_kv ? 1 : 0, wkind, 0))
    {
        bool ok = true;
        for (auto& L : s->layers)
            ok = ok && L.qkv.wtype == (s->sq ? W_INT8_SQ : (woq8_all ? W_INT8_WOQ : (fp16_all ? W_FP16 : W_INT4_WOQ))) && L.qkv.K == D
                && L.qkv.ldw == (fp16_all ? 2 * D : (woq4_all ? D / 2 : D)) && L.qkv.N == 3 * s->Dr && (L.qkv.scale_col || fp16_all);
        if (ok)
        {
            const size_t xb = qkv_attn_fused_xchg_bytes(s->Hr);
            RUN(s->dalloc(&s->fused_xchg, xb));
            HIP_OK(hipMemset(s->fused_xchg, 0, xb));
            s->qkv_attn_fused = true;
            // (tp == 1 here: no all-reduce behind the projection.  Weight-only int8: the context row travels as fp16)
            // (int4: the stage is built and bit-identical but measures at par with the GEMV launch it replaces - 724 vs 721 - 735 tokens/s -
            //  so it is on only when asked for: fuse_o_projection = 1)
            s->o_fused = s->fuse_o_cfg != 0 && !fp16_all && (woq4_all ? s->fuse_o_cfg > 0 : true) && (s->sq ? !s->per_token : true);
            for (auto& L : s->layers)
                s->o_fused = s->o_fused && L.dense.N == D && L.dense.scale_col
                    && (s->sq ? (L.dense.wtype == W_INT8_SQ && L.dense.act_scale && L.attn_qscale)
                              : L.dense.wtype == (woq4_all ? W_INT4_WOQ : W_INT8_WOQ))
                    && qkv_attn_fused_serves_o(s->Hr, s->Dh, L.dense.N, L.dense.K, L.dense.ldw, wkind);
            // ... and the instance that will run must be resident as a whole (occupancy query x CUs of this device >= its grid)
            if (!qkv_attn_fused_serves(D, s->Hr, s->Dh, Smax, s->int8_kv ? 1 : 0, wkind, s->o_fused ? 1 : 0))
            {
                s->o_fused = false;
                s->qkv_attn_fused = qkv_attn_fused_serves(D, s->Hr, s->Dh, Smax, s->int8_kv ? 1 : 0, wkind, 0);
            }
            s->fused_timing = nullptr;
            if (s->fused_timeline)
            {
                RUN(s->dalloc(&s->fused_timing, (size_t) s->Hr * 8 * 16 * 8));
                HIP_OK(hipMemset(s->fused_timing, 0, (size_t) s->Hr * 8 * 16 * 8));
            }
        }
    }
    return 0;
}

// After a stream synchronisation: did a peer-to-peer collective of this tensor-parallel session time out?  (The kernels
// never hang: a bounded wait that expires raises a device flag and every later launch backs off, p2p_allreduce.hip.)  Then
// the hidden states / logits behind this point are not sums over all ranks: fail the call and take the transport out of
// service, so that later sessions of this process fall back to RCCL.
// (distinct from 1: tllm_session_generate re-runs the request on the launches the session has fallen back to)
constexpr int kFusedTimedOut = 2;

static int check_comm(tllm_session_t s)
{
    if ((s->qkv_attn_fused || s->mlp_fused_dec) && s->fused_err)
    {
        // the fused projection + attention launch waits (bounded) for sibling workgroups of the same launch; an expired wait means
        // the grid was not resident at once - the rows behind it are not attention outputs
        uint32_t e = 0;
        if (hipMemcpy(&e, s->fused_err, 4, hipMemcpyDeviceToHost) != hipSuccess)
        {
            set_error("session: cannot read the fused-attention error word");
            return 1;
        }
        if (e)
        {
            (void) hipMemset(s->fused_err, 0, 4);
            s->qkv_attn_fused = false; // later steps take the two-launch path
            s->o_fused = false;
            s->mlp_fused_dec = false;  // ... and the two GEMV launches of the MLP
            if (s->graph)
            {
                (void) hipGraphExecDestroy(s->graph);
                s->graph = nullptr;
            }
            set_error("session: a fused decode launch (QKV + attention: codes 1 - 8, MLP: 16) timed out waiting for a sibling workgroup "
                      "(code %u); the results of this call are invalid, later steps run the unfused launches", e);
            return kFusedTimedOut;
        }
    }
    // only while the transport is IN SERVICE: once a time-out has taken it out (below), later calls run over RCCL and the
    // words that recorded the failure must not fail them (disable_after_error clears them as well)
    if (s->tp == 1 && !s->force_comm)
        return 0;
    // A time-out detected through ANOTHER live session of this process took the transport out of service (and cleared the
    // words) since this session last looked: the launches this call replayed from its captured graph ran against the broken
    // group - they spun to their time-out and left x / ar_norm untouched.  Fail this call too and drop the graph.
    const uint64_t eg = comm::p2p::error_generation();
    if (eg != s->comm_err_seen)
    {
        s->comm_err_seen = eg;
        if (s->graph && s->graph_comm_gen != comm::p2p::generation())
        {
            (void) hipGraphExecDestroy(s->graph);
            s->graph = nullptr;
            set_error("session: the peer-to-peer transport timed out (seen by another session of this process) while this "
                      "session's captured step still used it; the results of this call are invalid, later calls use RCCL");
            return 1;
        }
    }
    if (!comm::p2p::enabled())
        return 0;
    uint32_t e = 0;
    if (comm::p2p::error_flag(&e) != 0)
    {
        set_error("session: cannot read the peer-to-peer error flag");
        return 1;
    }
    if (e)
    {
        comm::p2p::disable_after_error();
        s->comm_err_seen = comm::p2p::error_generation();
        if (s->graph) // the captured step holds the peer-to-peer launches
        {
            (void) hipGraphExecDestroy(s->graph);
            s->graph = nullptr;
        }
        set_error("session: a peer-to-peer all-reduce timed out waiting for a rank (%s, epoch %u); the results of this call are "
                  "invalid and the peer-to-peer transport is out of service on every rank of the group (later calls use RCCL)",
            (e & 0x80000000u) ? "reported by a peer" : "on this rank", e & 0x7fffffffu);
        return 1;
    }
    return 0;
}

static int upload_prompt(tllm_session_t s, const int32_t* input_ids, const int32_t* input_lengths, hipStream_t st)
{
    // input_ids / input_lengths describe the Bc prompts; the per-sequence generation state is tiled over the beam
    // (generation.py:898-915 _tile_beam_width) - the prompt itself runs once per batch entry
    const int B = s->B, Bc = s->Bc, W = s->beam, S = s->max_in, Smax = s->Smax;
    std::vector<int32_t> lens_c(input_lengths, input_lengths + Bc), lens(B), seq(B, S), zeros(B, 0);
    std::vector<int32_t> mask((size_t) B * Smax, 0), out((size_t) B * Smax, 0);
    for (int bb = 0; bb < B; ++bb)
    {
        const int b = bb / W;
        lens[bb] = lens_c[b];
        if (lens[bb] < 1 || lens[bb] > S)
        {
            set_error("session: input_lengths[%d]=%d out of range [1, %d]", b, lens[bb], S);
            return 1;
        }
        // masked_tokens[b, len_b:max_in] = 1 (generation.py:812-821)
        for (int t = lens[bb]; t < S; ++t)
            mask[(size_t) bb * Smax + t] = 1;
        for (int t = 0; t < S; ++t)
            out[(size_t) bb * Smax + t] = input_ids[(size_t) b * S + t];
    }
    std::vector<int32_t> packed_ids, cu(Bc + 1, 0), last(Bc, 0);
    if (s->packed)
    {
        // the real tokens back to back; generation keeps the padded cache layout (slots [len, max_in) masked), so only
        // the context phase changes shape (generation.py:556-568 with remove_input_padding)
        for (int b = 0; b < Bc; ++b)
        {
            cu[b + 1] = cu[b] + lens_c[b];
            last[b] = cu[b + 1] - 1;
            packed_ids.insert(packed_ids.end(), input_ids + (size_t) b * S, input_ids + (size_t) b * S + lens_c[b]);
        }
        s->ctx_tokens = cu[Bc];
        HIP_OK(hipMemcpyAsync(s->ids_in, packed_ids.data(), packed_ids.size() * 4, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(s->cu_dev, cu.data(), cu.size() * 4, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(s->last_rows, last.data(), last.size() * 4, hipMemcpyHostToDevice, st));
    }
    else
        HIP_OK(hipMemcpyAsync(s->ids_in, input_ids, (size_t) Bc * S * 4, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(s->in_len, lens.data(), B * 4, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(s->in_len_ctx, lens_c.data(), Bc * 4, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(s->last_tok, lens_c.data(), Bc * 4, hipMemcpyHostToDevice, st));
    std::vector<float> cum;
    if (W > 1)
    {
        // only hypothesis 0 of every beam group is live before the first step (generation.py:392-397)
        cum.assign(B, -1e20f);
        for (int b = 0; b < Bc; ++b)
            cum[(size_t) b * W] = 0.f;
        HIP_OK(hipMemcpyAsync(s->cum_log_probs, cum.data(), B * 4, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemsetAsync(s->parent_ids, 0, (size_t) B * Smax * 4, st));
        HIP_OK(hipMemsetAsync(s->cache_ind, 0, (size_t) B * Smax * 4, st)); // every slot -> hypothesis 0's rows
    }
    HIP_OK(hipMemcpyAsync(s->seq_len, seq.data(), B * 4, hipMemcpyHostToDevice, st));
    if (s->attn_tickets) // re-arm the in-launch merge (it re-arms itself per launch; this covers a step that never finished)
        HIP_OK(hipMemsetAsync(s->attn_tickets, 0, (size_t) B * s->Hr * 4, st));
    HIP_OK(hipMemcpyAsync(s->finished, zeros.data(), B * 4, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(s->masked, mask.data(), mask.size() * 4, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(s->out_ids, out.data(), out.size() * 4, hipMemcpyHostToDevice, st));
    HIP_OK(hipStreamSynchronize(st)); // the host vectors above go out of scope
    return 0;
}

int32_t tllm_session_context(tllm_session_t s, const int32_t* input_ids, const int32_t* input_lengths,
    tllm_stream_t stream)
{
    if (!s || !s->B)
    {
        set_error("tllm_session_context: call tllm_session_setup first");
        return 1;
    }
    hipStream_t st = s->pick(stream);
    RUN(upload_prompt(s, input_ids, input_lengths, st));
    RUN(s->run_context(st));
    RUN(s->run_sampler(0, st));
    return 0;
}

int32_t tllm_session_step(tllm_session_t s, int32_t n_steps, int32_t use_graph, tllm_stream_t stream)
{
    if (!s || !s->B)
    {
        set_error("tllm_session_step: call tllm_session_setup first");
        return 1;
    }
    hipStream_t st = s->pick(stream);
    // a graph captured while the transport was in another state (enabled / fused seam / a re-created region) holds launches
    // that no longer match what an eager step would issue: capture again
    if (use_graph && (!s->graph || s->graph_stream != st || (s->tp > 1 && s->graph_comm_gen != comm::p2p::generation())))
    {
        if (s->graph)
        {
            (void) hipGraphExecDestroy(s->graph);
            s->graph = nullptr;
        }
        hipGraph_t g = nullptr;
        HIP_OK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        const int rc = s->run_decode_step(st);
        const hipError_t ce = hipStreamEndCapture(st, &g);
        if (rc || ce != hipSuccess || !g)
        {
            if (!rc)
                set_error("session: hipStreamEndCapture failed: %s", hipGetErrorString(ce));
            return 1;
        }
        const hipError_t ie = hipGraphInstantiate(&s->graph, g, nullptr, nullptr, 0);
        (void) hipGraphDestroy(g);
        if (ie != hipSuccess)
        {
            set_error("session: hipGraphInstantiate failed: %s", hipGetErrorString(ie));
            s->graph = nullptr;
            return 1;
        }
        s->graph_stream = st;
        s->graph_comm_gen = comm::p2p::generation();
    }
    for (int i = 0; i < n_steps; ++i)
    {
        if (use_graph)
            HIP_OK(hipGraphLaunch(s->graph, st));
        else
            RUN(s->run_decode_step(st));
    }
    return 0;
}

int32_t tllm_session_fake_context(tllm_session_t s, int32_t length, uint32_t seed, tllm_stream_t stream)
{
    if (!s || !s->B || length < 1 || length > s->max_in)
    {
        set_error("tllm_session_fake_context: bad length");
        return 1;
    }
    hipStream_t st = s->pick(stream);
    const int B = s->B, S = s->max_in;
    std::vector<int32_t> ids((size_t) s->Bc * S, 3), lens(s->Bc, length);
    // a padded prompt of `length` real tokens: slots [length, max_in) are masked
    RUN(upload_prompt(s, ids.data(), lens.data(), st));
    for (int i = 0; i < s->num_layers; ++i)
        RUN(launch_fill_random(s->layers[i].kv, s->int8_kv ? DT_INT8 : DT_HALF, s->kv_elems, seed + 7919u * i, 1.0f, st));
    RUN(s->run_sampler(0, st)); // prepares the RoPE row of the first generation step (the ids are overwritten next)
    RUN(launch_fill_i32(s->cur_ids, 3, B, st));
    RUN(launch_embedding(s->x, s->cur_ids, s->emb, B, s->hidden, s->vocab, st)); // what the sampler would have left in x
    return 0;
}

static int32_t generate_once(tllm_session_t s, const int32_t* input_ids, const int32_t* input_lengths, int32_t max_new_tokens,
    int32_t end_id, int32_t pad_id, int32_t* output_ids, tllm_stream_t stream);

int32_t tllm_session_generate(tllm_session_t s, const int32_t* input_ids, const int32_t* input_lengths,
    int32_t max_new_tokens, int32_t end_id, int32_t pad_id, int32_t* output_ids, tllm_stream_t stream)
{
    // A bounded wait of the one-launch projection + attention expired (its grid was not resident at once - another queue's kernels
    // held CUs): the session has fallen back to the two-launch path, the tokens behind the expired wait are invalid.  Greedy
    // generation is a function of the prompt alone, every cache slot is rewritten before it is read: run the request again.
    const int32_t rc = generate_once(s, input_ids, input_lengths, max_new_tokens, end_id, pad_id, output_ids, stream);
    if (rc != kFusedTimedOut)
        return rc;
    s->fused_retries += 1;
    return generate_once(s, input_ids, input_lengths, max_new_tokens, end_id, pad_id, output_ids, stream) ? 1 : 0;
}

int32_t tllm_session_fused_retries(tllm_session_t s)
{
    return s ? s->fused_retries : 0;
}

static int32_t generate_once(tllm_session_t s, const int32_t* input_ids, const int32_t* input_lengths, int32_t max_new_tokens,
    int32_t end_id, int32_t pad_id, int32_t* output_ids, tllm_stream_t stream)
{
    if (!s || !s->B || !input_ids || !input_lengths || !output_ids)
    {
        set_error("tllm_session_generate: bad arguments / setup not called");
        return 1;
    }
    if (max_new_tokens > s->max_new)
    {
        set_error("tllm_session_generate: max_new_tokens %d exceeds setup's %d", max_new_tokens, s->max_new);
        return 1;
    }
    hipStream_t st = s->pick(stream);
    s->end_id = end_id;
    if (s->graph)
    {
        // end_id is baked into the captured sampler node
        (void) hipGraphExecDestroy(s->graph);
        s->graph = nullptr;
    }
    RUN(tllm_session_context(s, input_ids, input_lengths, stream));
    int produced = max_new_tokens > 0 ? 1 : 0; // tokens generated per sequence (the prompt pass yields the first)
    if (max_new_tokens > 1)
    {
        // first generation step eagerly (also warms lazily-initialised state), the rest from the graph
        RUN(tllm_session_step(s, 1, 0, stream));
        produced = 2;
        if (max_new_tokens > 2)
        {
            const int chunk = 32; // poll the finished flags every `chunk` steps instead of every step
            int done = 2;
            std::vector<int32_t> fin(s->B);
            while (done < max_new_tokens)
            {
                const int n = std::min(chunk, max_new_tokens - done);
                RUN(tllm_session_step(s, n, 1, stream));
                done += n;
                produced = done;
                if (end_id >= 0 && done < max_new_tokens)
                {
                    HIP_OK(hipMemcpyAsync(fin.data(), s->finished, s->B * 4, hipMemcpyDeviceToHost, st));
                    HIP_OK(hipStreamSynchronize(st));
                    RUN(check_comm(s));
                    bool all = true;
                    for (auto f : fin)
                        all = all && f;
                    if (all)
                        break;
                }
            }
        }
    }
    if (s->beam > 1)
        return tllm_session_get_beam_output(s, output_ids, nullptr, stream);
    RUN(tllm_session_get_output_ids(s, output_ids, stream));
    // The reference runs gather_tree for beam_width 1 too (generation.py:990-994; K/decodingKernels.cu:130-156): everything
    // after a sequence's first end token, and the tail no step wrote because every sequence had finished, is end_id - not the
    // 0 (<unk>) the buffer was initialised with.  Without an end token (benchmarks) the unused tail is pad_id.
    const int32_t fill = end_id >= 0 ? end_id : pad_id;
    for (int b = 0; b < s->B; ++b)
    {
        int32_t* o = output_ids + (size_t) b * s->Smax;
        bool done = false;
        for (int t = s->max_in; t < s->Smax; ++t)
        {
            if (done || t >= s->max_in + produced)
                o[t] = fill;
            else if (end_id >= 0 && o[t] == end_id)
                done = true;
        }
    }
    return 0;
}

int32_t tllm_session_get_logits(tllm_session_t s, float* logits, tllm_stream_t stream)
{
    if (!s || !s->B || !logits)
        return 1;
    hipStream_t st = s->pick(stream);
    const int rows = s->logit_rows;
    if (s->tp == 1)
    {
        HIP_OK(hipMemcpyAsync(logits, s->logits_local, (size_t) rows * s->vocab * 4, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        return check_comm(s);
    }
    std::vector<float> g((size_t) s->tp * rows * s->Vr);
    HIP_OK(hipMemcpyAsync(g.data(), s->logits, g.size() * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    RUN(check_comm(s));
    for (int b = 0; b < rows; ++b)
        for (int v = 0; v < s->vocab; ++v)
            logits[(size_t) b * s->vocab + v] = g[((size_t) (v / s->Vr) * rows + b) * s->Vr + v % s->Vr];
    return 0;
}

int32_t tllm_session_get_output_ids(tllm_session_t s, int32_t* ids, tllm_stream_t stream)
{
    if (!s || !s->B || !ids)
        return 1;
    hipStream_t st = s->pick(stream);
    HIP_OK(hipMemcpyAsync(ids, s->out_ids, (size_t) s->B * s->Smax * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    return check_comm(s);
}

int32_t tllm_session_logit_rows(tllm_session_t s)
{
    return s ? s->logit_rows : 0;
}

int32_t tllm_session_vocab_size(tllm_session_t s)
{
    return s ? s->vocab : 0;
}

// Back-track the beams (K/decodingKernels.cu:30-171 gatherTree, called at PY/runtime/generation.py:990-994): hypothesis j of
// batch entry b ends with the token recorded for it at the last slot; its earlier tokens are those of its ancestors.
int32_t tllm_session_get_beam_output(tllm_session_t s, int32_t* ids, float* cum_log_probs, tllm_stream_t stream)
{
    if (!s || !s->B || !ids)
    {
        set_error("tllm_session_get_beam_output: bad arguments / setup not called");
        return 1;
    }
    hipStream_t st = s->pick(stream);
    const int B = s->B, W = s->beam, Smax = s->Smax, S = s->max_in;
    if (W == 1)
    {
        if (cum_log_probs)
        {
            set_error("tllm_session_get_beam_output: cum_log_probs are only kept with beam_width > 1");
            return 1;
        }
        return tllm_session_get_output_ids(s, ids, stream);
    }
    std::vector<int32_t> step_ids((size_t) B * Smax), parents((size_t) B * Smax), len(B);
    HIP_OK(hipMemcpyAsync(step_ids.data(), s->out_ids, step_ids.size() * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(parents.data(), s->parent_ids, parents.size() * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(len.data(), s->seq_len, B * 4, hipMemcpyDeviceToHost, st));
    if (cum_log_probs)
        HIP_OK(hipMemcpyAsync(cum_log_probs, s->cum_log_probs, B * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    RUN(check_comm(s));
    const int32_t fill = s->end_id >= 0 ? s->end_id : 0;
    for (int bb = 0; bb < B; ++bb)
    {
        const int b0 = bb / W * W;
        int32_t* o = ids + (size_t) bb * Smax;
        const int last = std::min(len[bb], Smax - 1); // slot of the newest token
        for (int t = 0; t < S; ++t)
            o[t] = step_ids[(size_t) bb * Smax + t]; // the (padded) prompt, shared by the beam group
        int j = bb - b0;
        for (int t = last; t >= S; --t)
        {
            o[t] = step_ids[(size_t) (b0 + j) * Smax + t];
            j = parents[(size_t) (b0 + j) * Smax + t];
            if (j < 0 || j >= W)
                j = 0;
        }
        // everything after the first end token, and the unused tail, is the end token (:130-156)
        bool done = false;
        for (int t = S; t < Smax; ++t)
        {
            if (t > last || done)
                o[t] = fill;
            else if (s->end_id >= 0 && o[t] == s->end_id)
                done = true;
        }
    }
    return 0;
}

int32_t tllm_session_get_beam_state(tllm_session_t s, int32_t* parent_ids, int32_t* cache_indirection, int32_t* finished,
    int32_t* sequence_lengths, tllm_stream_t stream)
{
    if (!s || !s->B || s->beam < 2)
    {
        set_error("tllm_session_get_beam_state: no beam search set up");
        return 1;
    }
    hipStream_t st = s->pick(stream);
    const size_t n = (size_t) s->B * s->Smax * 4;
    if (parent_ids)
        HIP_OK(hipMemcpyAsync(parent_ids, s->parent_ids, n, hipMemcpyDeviceToHost, st));
    if (cache_indirection)
        HIP_OK(hipMemcpyAsync(cache_indirection, s->cache_ind, n, hipMemcpyDeviceToHost, st));
    if (finished)
        HIP_OK(hipMemcpyAsync(finished, s->finished, (size_t) s->B * 4, hipMemcpyDeviceToHost, st));
    if (sequence_lengths)
        HIP_OK(hipMemcpyAsync(sequence_lengths, s->seq_len, (size_t) s->B * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    return 0;
}

int32_t tllm_session_get_step_state(tllm_session_t s, int32_t* sequence_length, int32_t* next_position, int32_t* masked_tokens,
    int32_t* input_lengths, tllm_stream_t stream)
{
    if (!s || !s->B)
    {
        set_error("tllm_session_get_step_state: setup not called");
        return 1;
    }
    hipStream_t st = s->pick(stream);
    if (sequence_length)
        HIP_OK(hipMemcpyAsync(sequence_length, s->seq_len, (size_t) s->B * 4, hipMemcpyDeviceToHost, st));
    if (next_position)
        HIP_OK(hipMemcpyAsync(next_position, s->rope_pos, (size_t) s->B * 4, hipMemcpyDeviceToHost, st));
    if (masked_tokens)
        HIP_OK(hipMemcpyAsync(masked_tokens, s->masked, (size_t) s->B * s->Smax * 4, hipMemcpyDeviceToHost, st));
    if (input_lengths)
        HIP_OK(hipMemcpyAsync(input_lengths, s->in_len, (size_t) s->B * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    return 0;
}

int32_t tllm_session_force_tokens(tllm_session_t s, const int32_t* ids, tllm_stream_t stream)
{
    if (!s || !s->B || !ids)
    {
        set_error("tllm_session_force_tokens: bad arguments / setup not called");
        return 1;
    }
    if (s->beam > 1)
    {
        set_error("tllm_session_force_tokens: greedy sessions only");
        return 1;
    }
    hipStream_t st = s->pick(stream);
    // staged through the (idle between steps) prompt-id buffer; the greedy sampler leaves the next input row in x only when
    // hidden % 8 == 0, otherwise the step gathers it from cur_ids itself
    HIP_OK(hipMemcpyAsync(s->ids_in, ids, (size_t) s->B * 4, hipMemcpyHostToDevice, st));
    const bool gather = s->hidden % 8 == 0;
    if (tllm::kernels::launch_force_token(s->ids_in, s->cur_ids, s->out_ids, s->Smax, s->seq_len, gather ? s->emb : nullptr,
            gather ? s->x : nullptr, s->B, s->hidden, s->vocab, st))
        return 1;
    HIP_OK(hipStreamSynchronize(st));
    return 0;
}

int32_t tllm_session_get_tap_ex(tllm_session_t s, int32_t layer, int32_t which, void* host, size_t nbytes, tllm_stream_t stream)
{
    if (!s || !s->B || !host || layer < 0 || layer >= s->num_layers || which < 0 || which >= tllm_session::kTaps)
    {
        set_error("tllm_session_get_tap: bad arguments / setup not called");
        return 1;
    }
    if (!s->tap_buf[which])
    {
        set_error("tllm_session_get_tap: the session was not created with debug_taps=1");
        return 1;
    }
    const size_t row = (size_t) s->B * s->tap_width(which) * ((s->sq && which != 4) ? 1 : 2);
    if (nbytes != row)
    {
        set_error("tllm_session_get_tap: buffer of %zu bytes, the tap holds %zu", nbytes, row);
        return 1;
    }
    hipStream_t st = s->pick(stream);
    HIP_OK(hipMemcpyAsync(host, s->tap_ptr(which, layer), row, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    return 0;
}

int32_t tllm_session_get_tap(tllm_session_t s, int32_t layer, void* host, size_t nbytes, tllm_stream_t stream)
{
    return tllm_session_get_tap_ex(s, layer, 1, host, nbytes, stream);
}

void* tllm_session_fused_timeline_ptr(tllm_session_t s)
{
    return s ? s->fused_timing : nullptr;
}

void* tllm_session_mlp_timeline_ptr(tllm_session_t s)
{
    return s ? s->mlp_timing : nullptr;
}

void* tllm_session_kv_cache_ptr(tllm_session_t s, int32_t layer)
{
    if (!s || layer < 0 || layer >= (int) s->layers.size())
        return nullptr;
    return s->layers[layer].kv;
}

int64_t tllm_session_step_bytes(tllm_session_t s, int32_t context_len)
{
    if (!s)
        return 0;
    // weights once + KV read of `context_len` positions + KV write of one position, per rank
    auto lin = [&](const Linear& L) {
        int64_t b = (int64_t) L.N * L.ldw;
        if (L.wtype == W_INT8_WOQ || L.wtype == W_INT4_WOQ)
            b += (int64_t) L.N * 2;
        else if (L.wtype == W_INT8_SQ && L.per_channel)
            b += (int64_t) L.N * 4;
        return b;
    };
    int64_t bytes = lin(s->head);
    const int64_t kv_row = (int64_t) 2 * s->Hr * s->Dh * (s->int8_kv ? 1 : 2);
    for (auto& L : s->layers)
        bytes += lin(L.qkv) + lin(L.dense) + lin(L.fc) + lin(L.gate) + lin(L.proj) + kv_row * s->B * (context_len + 1);
    return bytes;
}

int32_t tllm_session_time_kernel(tllm_session_t s, int32_t which, int32_t sweeps, float* avg_us, int64_t* launches,
    tllm_stream_t stream)
{
    if (!s || !s->B || sweeps < 1 || !avg_us || !launches
        || !(which == 1 || which == 2 || which == 4 || which == 5 || which == 6 || which == 7 || which == 8))
    {
        set_error("tllm_session_time_kernel: bad arguments (which in {1,2,4,5,6,7,8}) / setup not called");
        return 1;
    }
    if (which == 8 && !s->mlp_fused_dec)
    {
        set_error("tllm_session_time_kernel: kernel 8 is the one-launch MLP, which this session does not run");
        return 1;
    }
    if (which == 7 && !s->qkv_attn_fused)
    {
        set_error("tllm_session_time_kernel: kernel 7 is the one-launch projection + attention, which this session does not run");
        return 1;
    }
    hipStream_t st = s->pick(stream);
    hipEvent_t a, b;
    (void) hipEventCreate(&a);
    (void) hipEventCreate(&b);
    // (id 7 with the O-projection stage adds O(ctx) to x on every launch: the residual row is put back afterwards)
    std::vector<char> x_keep;
    if (which == 7 || which == 8)
    {
        x_keep.resize((size_t) s->B * s->hidden * 2);
        HIP_OK(hipMemcpyAsync(x_keep.data(), s->x, x_keep.size(), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
    }
    s->only_kernel = which;
    int rc = s->run_decode_step(st); // untimed sweep
    (void) hipEventRecord(a, st);
    for (int i = 0; i < sweeps && !rc; ++i)
        rc = s->run_decode_step(st);
    (void) hipEventRecord(b, st);
    s->only_kernel = -1;
    (void) hipStreamSynchronize(st);
    float ms = 0.f;
    if (!rc && hipEventElapsedTime(&ms, a, b) != hipSuccess)
    {
        set_error("tllm_session_time_kernel: event timing failed");
        rc = 1;
    }
    (void) hipEventDestroy(a);
    (void) hipEventDestroy(b);
    if (!x_keep.empty())
    {
        HIP_OK(hipMemcpyAsync(s->x, x_keep.data(), x_keep.size(), hipMemcpyHostToDevice, st));
        HIP_OK(hipStreamSynchronize(st));
    }
    *launches = (int64_t) sweeps * s->num_layers;
    *avg_us = ms * 1000.f / (float) *launches;
    return rc;
}

int32_t tllm_session_decode_form(tllm_session_t s)
{
    if (!s || !s->B)
        return -1;
    return (s->qkv_attn_fused ? 1 : 0) | (s->qkv_attn_fused && s->o_fused ? 2 : 0) | (s->mlp_fused_dec ? 4 : 0);
}

int32_t tllm_session_profile(tllm_session_t s, int32_t n_steps, float* ms_per_class, int64_t* launches_per_class,
    tllm_stream_t stream)
{
    if (!s || !s->B || n_steps < 1 || !ms_per_class || !launches_per_class)
    {
        set_error("tllm_session_profile: bad arguments / setup not called");
        return 1;
    }
    hipStream_t st = s->pick(stream);
    s->profiling = true;
    s->prof.clear();
    int rc = 0;
    for (int i = 0; i < n_steps && !rc; ++i)
        rc = s->run_decode_step(st);
    s->profiling = false;
    (void) hipStreamSynchronize(st);
    for (int c = 0; c < tllm_session::PC_COUNT; ++c)
    {
        ms_per_class[c] = 0.f;
        launches_per_class[c] = 0;
    }
    for (auto& r : s->prof)
    {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess)
        {
            ms_per_class[r.cls] += ms;
            launches_per_class[r.cls] += 1;
        }
        (void) hipEventDestroy(r.a);
        (void) hipEventDestroy(r.b);
    }
    s->prof.clear();
    return rc;
}

void tllm_session_destroy(tllm_session_t s)
{
    delete s;
}

int32_t tllm_gemv(const tllm_gemv_params_t* q, tllm_stream_t stream)
{
    if (!q)
        return 1;
    GemvParams p;
    p.wtype = q->wtype;
    p.pro = q->pro;
    p.epi = q->epi;
    p.out_dtype = q->out_dtype;
    p.M = q->M;
    p.N = q->N;
    p.K = q->K;
    p.x = q->x;
    p.ldx = q->ldx;
    p.w = q->w;
    p.ldw = q->ldw;
    p.scale_col = q->scale_col;
    p.scale_row = q->scale_row;
    p.per_channel = q->per_channel;
    p.per_token = q->per_token;
    p.gamma = q->gamma;
    p.eps = q->eps;
    p.act_scale = q->act_scale;
    p.dyn_scale_out = q->dyn_scale_out;
    p.x_pro_out = q->x_pro_out;
    p.residual = q->residual;
    p.epi_scale = q->epi_scale;
    p.y = q->y;
    p.ldy = q->ldy;
    return launch_gemv(p, reinterpret_cast<hipStream_t>(stream)) ? 1 : 0;
}

int32_t tllm_gemm(const tllm_gemm_params_t* q, tllm_stream_t stream)
{
    if (!q)
        return 1;
    GemmParams g;
    g.wtype = q->wtype;
    g.out_dtype = q->out_dtype;
    g.M = q->M;
    g.N = q->N;
    g.K = q->K;
    g.a = q->a;
    g.lda = q->lda;
    g.w = q->w;
    g.ldw = q->ldw;
    g.scale_col = q->scale_col;
    g.scale_row = q->scale_row;
    g.per_channel = q->per_channel;
    g.per_token = q->per_token;
    g.c = q->c;
    g.ldc = q->ldc;
    return launch_gemm(g, reinterpret_cast<hipStream_t>(stream)) ? 1 : 0;
}

int32_t tllm_gemm_residual(const tllm_gemm_params_t* q, const void* residual, tllm_stream_t stream)
{
    if (!q || !residual || q->out_dtype != DT_HALF)
    {
        set_error("tllm_gemm_residual: needs a residual and fp16 output");
        return 1;
    }
    GemmParams g;
    g.wtype = q->wtype;
    g.out_dtype = q->out_dtype;
    g.M = q->M;
    g.N = q->N;
    g.K = q->K;
    g.a = q->a;
    g.lda = q->lda;
    g.w = q->w;
    g.ldw = q->ldw;
    g.scale_col = q->scale_col;
    g.scale_row = q->scale_row;
    g.per_channel = q->per_channel;
    g.per_token = q->per_token;
    g.c = q->c;
    g.ldc = q->ldc;
    g.residual = residual;
    return launch_gemm(g, reinterpret_cast<hipStream_t>(stream)) ? 1 : 0;
}

int32_t tllm_gemm_profile(int32_t wtype, int32_t M, int32_t N, int32_t K, int32_t* best_cfg, float* best_us, tllm_stream_t stream)
{
    int cfg = 0;
    float us = 0.f;
    if (gemm_profile(wtype, M, N, K, &cfg, &us, reinterpret_cast<hipStream_t>(stream)))
        return 1;
    if (best_cfg)
        *best_cfg = cfg;
    if (best_us)
        *best_us = us;
    return 0;
}

int64_t tllm_gemm_tactics_export(char* buf, int64_t capacity)
{
    const std::string t = gemm_tactics_export();
    if (buf && capacity > 0)
    {
        const size_t n = std::min((size_t) capacity - 1, t.size());
        memcpy(buf, t.data(), n);
        buf[n] = 0;
    }
    return (int64_t) t.size() + 1;
}

int32_t tllm_gemm_tactics_import(const char* text)
{
    return gemm_tactics_import(text) < 0 ? 1 : 0;
}

void tllm_gemm_tactics_clear(void)
{
    gemm_tactics_clear();
}

int32_t tllm_gemm_tactic_lookup(int32_t wtype, int32_t M, int32_t N, int32_t K)
{
    return gemm_tactic_lookup(wtype, M, N, K);
}

void tllm_gemv_set_blocks_per_cu(int32_t n)
{
    tllm::kernels::gemv_tune_blocks_per_cu = n;
}

void tllm_gemv_set_mfma_rows(int32_t n)
{
    tllm::kernels::gemv_mfma_min_rows = n;
}

int32_t tllm_gemm_swiglu_quant(const tllm_gemm_params_t* q, const void* w_up, const void* scale_col_up, const float* quant_scale,
    tllm_stream_t stream)
{
    if (!q || !w_up || !scale_col_up || !quant_scale)
    {
        set_error("tllm_gemm_swiglu_quant: null argument");
        return 1;
    }
    GemmParams g;
    g.wtype = q->wtype;
    g.out_dtype = DT_INT8;
    g.M = q->M;
    g.N = q->N;
    g.K = q->K;
    g.a = q->a;
    g.lda = q->lda;
    g.w = q->w;
    g.ldw = q->ldw;
    g.scale_col = q->scale_col;
    g.scale_row = q->scale_row;
    g.per_channel = q->per_channel;
    g.per_token = q->per_token;
    g.c = q->c;
    g.ldc = q->ldc;
    g.w2 = w_up;
    g.scale_col2 = scale_col_up;
    g.swiglu_qscale = quant_scale;
    const int rc = tllm::kernels::launch_gemm_swiglu(g, reinterpret_cast<hipStream_t>(stream));
    if (rc == 1)
        set_error("tllm_gemm_swiglu_quant: problem not served by the fused kernel (SmoothQuant static, K %% 128 == 0, M >= 32, 16-byte aligned operands)");
    return rc ? 1 : 0;
}

void tllm_gemm_set_clock_probe(void* device_buffer)
{
    tllm::kernels::gemm_clock_probe = device_buffer;
}

void tllm_gemm_set_tile_cfg(int32_t cfg)
{
    // 0 resets both tables; 101.. select the tile shape of the weight-only main-loop-dequantising GEMM (gemm_woq.hip: 101 = 256 x 192,
    // 102 = 128 x 128, 103 = 256 x 192 two stages ahead, 104 = 256 x 192 on 4 waves)
    // -2: the fused SwiGLU SmoothQuant GEMM in its one-tile-per-workgroup form (A/B against the persistent one; 0 resets)
    if (cfg == 0 || cfg == -2)
        tllm::kernels::gemm_swiglu_one_tile = cfg == -2;
    if (cfg == -2)
        return;
    if (cfg == 0 || cfg > 100)
        tllm::kernels::gemm_woq_tune_cfg = cfg > 100 ? cfg - 100 : 0;
    if (cfg > 100)
        return;
    tllm::kernels::gemm_tune_cfg = cfg;
}

} // extern "C"
