// The plugin classes behind include/tllm_plugin_api.h — one per reference plugin on the LLaMA hot path
// (SURVEY.md §2.2), same field names, same input/output order, same shape rules.
#include "plugin_base.h"
#include "../kernels/weight_layout.h"
#include "comm.h"
#include <mutex>

namespace tllm
{
namespace plugins
{
using namespace kernels;

namespace
{

int dtype_size(int32_t t)
{
    switch (t)
    {
    case TLLM_FLOAT:
    case TLLM_INT32: return 4;
    case TLLM_HALF: return 2;
    default: return 1;
    }
}

bool linear_fmt(const Desc& d)
{
    return d.format == 0;
}

} // namespace

// ================================================================================================
// RoPE table cache
// ================================================================================================
const float* rope_table(int rotary_dim, int min_len, int* len_out)
{
    struct Entry
    {
        float* dev = nullptr;
        int len = 0;
    };
    static std::mutex mu;
    static std::map<int, Entry> tables;
    std::lock_guard<std::mutex> lk(mu);
    Entry& e = tables[rotary_dim];
    if (e.len < min_len)
    {
        int len = 2048;
        while (len < min_len)
            len *= 2;
        std::vector<float> host((size_t) len * (rotary_dim / 2) * 2);
        fill_rope_table_host(host.data(), len, rotary_dim);
        float* dev = nullptr;
        if (hipMalloc(&dev, host.size() * sizeof(float)) != hipSuccess)
        {
            set_error("rope_table: hipMalloc failed (is a stream capture active? reserve the table in setup)");
            return nullptr;
        }
        if (hipMemcpy(dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        {
            set_error("rope_table: upload failed");
            (void) hipFree(dev);
            return nullptr;
        }
        // the old table (if any) may still be referenced by in-flight kernels: leak it deliberately (tiny).
        e.dev = dev;
        e.len = len;
    }
    if (len_out)
        *len_out = e.len;
    return e.dev;
}

// ================================================================================================
// GPTAttention   (P/gptAttentionPlugin/gptAttentionPlugin.cpp, P/gptAttentionCommon/gptAttentionCommon.cpp)
// ================================================================================================
class GPTAttentionPlugin : public Plugin
{
public:
    static constexpr size_t kCuBytes = 4096; // packed inputs: cu_seqlens at the head of the workspace (batch <= 1023)
    struct Cfg
    {
        int32_t num_heads, head_size, unidirectional;
        float q_scaling;
        int32_t rotary_embedding_dim;
        int8_t neox_rotary_style, context_fmha_type, multi_block_mode, multi_query_mode;
        int32_t int8_kv_cache, fp8_kv_cache;
        int8_t remove_input_padding;
        int32_t mask_type, paged_kv_cache, type_id, in_flight_batching;
    } c;

    static Plugin* create(const Fields& f)
    {
        f.expect_only({"num_heads", "head_size", "unidirectional", "q_scaling", "rotary_embedding_dim",
            "neox_rotary_style", "context_fmha_type", "multi_block_mode", "multi_query_mode", "int8_kv_cache",
            "fp8_kv_cache", "remove_input_padding", "mask_type", "paged_kv_cache", "type_id", "in_flight_batching"});
        auto* p = new GPTAttentionPlugin;
        Cfg& c = p->c;
        c.num_heads = f.i32("num_heads");
        c.head_size = f.i32("head_size");
        c.unidirectional = f.i32("unidirectional");
        c.q_scaling = f.f32("q_scaling");
        c.rotary_embedding_dim = f.i32("rotary_embedding_dim");
        c.neox_rotary_style = f.i8("neox_rotary_style");
        c.context_fmha_type = f.i8("context_fmha_type");
        c.multi_block_mode = f.i8("multi_block_mode");
        c.multi_query_mode = f.i8("multi_query_mode");
        c.int8_kv_cache = f.i32("int8_kv_cache");
        c.fp8_kv_cache = f.i32("fp8_kv_cache");
        c.remove_input_padding = f.i8("remove_input_padding");
        c.mask_type = f.i32("mask_type");
        c.paged_kv_cache = f.i32("paged_kv_cache");
        c.type_id = f.i32("type_id");
        c.in_flight_batching = f.i32("in_flight_batching");
        try
        {
            p->validate();
        }
        catch (...)
        {
            delete p;
            throw;
        }
        return p;
    }

    void validate() const
    {
        if (c.type_id != TLLM_HALF)
            throw std::runtime_error("GPTAttention: only type_id=float16 is built for MI355X");
        if (c.num_heads <= 0 || (c.head_size != 32 && c.head_size != 64 && c.head_size != 128 && c.head_size != 256))
            throw std::runtime_error("GPTAttention: head_size must be one of 32, 64, 128, 256");
        // reserved plugin features, rejected with a clear error until built
        if (c.multi_query_mode)
            throw std::runtime_error("GPTAttention: multi_query_mode not built (LLaMA-7B is MHA)");
        if (c.fp8_kv_cache)
            throw std::runtime_error("GPTAttention: fp8_kv_cache not built");
        if (c.in_flight_batching)
            throw std::runtime_error("GPTAttention: in_flight_batching not built");
        if (!c.unidirectional)
            throw std::runtime_error("GPTAttention: only causal (unidirectional) attention");
    }

    static Plugin* deserialize(Reader& r)
    {
        auto* p = new GPTAttentionPlugin;
        p->c = r.get<Cfg>();
        r.done();
        return p;
    }

    const char* type() const override { return "GPTAttention"; }
    int nbOutputs() const override { return 2; }

    int outputDims(int idx, const Dims* in, int nin, Dims* out) const override
    {
        if (nin < 8)
            return -1;
        if (idx == 0)
        {
            // [B, S, 3*Dr] -> [B, S, Dr]   (gptAttentionPlugin.cpp getOutputDimensions)
            *out = in[0];
            out->d[out->nbDims - 1] = c.num_heads * c.head_size;
        }
        else
            *out = in[1];
        return 0;
    }

    int outputDtype(int idx, const int32_t* t, int nin) const override { return idx == 0 ? t[0] : t[1]; }

    bool supportsFormat(int pos, const Desc* io, int nin, int nout) const override
    {
        const Desc& d = io[pos];
        if (!linear_fmt(d))
            return false;
        const int nin_expected = 8 + (c.int8_kv_cache ? 2 : 0) + (c.paged_kv_cache ? 1 : 0);
        if (nin != nin_expected)
            return false;
        if (pos == 0 || pos == nin)
            return d.type == c.type_id;
        if (pos == 1 || pos == nin + 1)
            return d.type == (c.int8_kv_cache ? TLLM_INT8 : c.type_id);
        if (pos >= 2 && pos <= 7)
            return d.type == TLLM_INT32;
        if (c.paged_kv_cache && pos == block_pointers_idx())
            return d.type == TLLM_INT32; // int64 pointers carried as int32 pairs (gptAttentionPlugin.cpp:106-110)
        return d.type == TLLM_FLOAT; // kv scales
    }

    int block_pointers_idx() const { return c.int8_kv_cache ? 10 : 8; } // gptAttentionPlugin.h:156-159

    size_t workspaceSize(const Desc* in, int nin, const Desc* out, int nout) const override
    {
        // packed inputs: qkv is [1, tokens, 3 D]; the batch is input_lengths' extent, the longest sequence input 6's
        const int B = c.remove_input_padding ? in[5].dims.d[0] : in[0].dims.d[0];
        const int S = c.remove_input_padding ? in[6].dims.d[0] : in[0].dims.d[1];
        const int Smax = c.paged_kv_cache ? in[7].dims.d[2] : in[1].dims.d[3];
        // max(context, generation), like the reference (gptAttentionPlugin.cpp:132-144)
        const size_t gen = mmha_workspace_size(B, c.num_heads, c.head_size, Smax) + 256;
        const size_t ctx = context_attention_workspace_size(B, c.num_heads, c.head_size, S) + 256 + kCuBytes;
        return gen > ctx ? gen : ctx;
    }

    int enqueue(const Desc* inDesc, const Desc* outDesc, const void* const* in, void* const* out, void* ws,
        hipStream_t stream) override
    {
        const int nin_expected = 8 + (c.int8_kv_cache ? 2 : 0);
        (void) nin_expected;
        const bool packed = c.remove_input_padding != 0; // tokens of all sequences back to back (gptAttentionPlugin.cpp:344-356)
        const int B = packed ? inDesc[5].dims.d[0] : inDesc[0].dims.d[0];
        int S = packed ? 1 : inDesc[0].dims.d[1];
        const int Smax = inDesc[7].dims.d[2]; // max_seq_len from cache_indirection.shape[2] (gptAttentionPlugin.cpp:335)
        // paged KV cache (gptAttentionPlugin.cpp:313-325): input 1 is the block pool [blocks, 2, H, tokens_per_block, Dh],
        // reached only through the pointer table int64 [B, beam, 2, max_blocks] that travels as int32 [.., 2 * max_blocks]
        const bool paged = c.paged_kv_cache != 0;
        const int64_t* block_pointers = nullptr;
        int tokens_per_block = 0, max_blocks = 0;
        if (paged)
        {
            const Desc& bp = inDesc[block_pointers_idx()];
            if (inDesc[1].dims.nbDims != 5 || inDesc[1].dims.d[1] != 2 || inDesc[1].dims.d[2] != c.num_heads
                || inDesc[1].dims.d[4] != c.head_size || bp.dims.nbDims != 4 || bp.dims.d[2] != 2 || bp.dims.d[3] % 2)
            {
                set_error("GPTAttention: paged KV cache needs the pool [blocks,2,H,tokens_per_block,Dh] and block pointers "
                          "int32 [B,beam,2,2*max_blocks]");
                return 1;
            }
            tokens_per_block = inDesc[1].dims.d[3];
            max_blocks = bp.dims.d[3] / 2;
            block_pointers = static_cast<const int64_t*>(in[block_pointers_idx()]);
        }
        else if (inDesc[1].dims.nbDims != 5 || inDesc[1].dims.d[3] != Smax || inDesc[1].dims.d[2] != c.num_heads
            || inDesc[1].dims.d[4] != c.head_size || inDesc[1].dims.d[1] != 2)
        {
            set_error("GPTAttention: past_key_value must be [B,2,H,Smax,Dh] with Smax = cache_indirection.shape[2]");
            return 1;
        }
        // cache_indirection [batch, beam_width, max_seq_len] (gptAttentionPlugin.cpp:330-336): in the generation phase the
        // B = batch * beam_width sequences read time step t from sibling cache_indirection[b, k, t]'s cache rows
        const int beam_width = inDesc[7].dims.d[1];
        if (beam_width < 1 || beam_width > 8)
        {
            set_error("GPTAttention: beam width %d (cache_indirection.shape[1]) outside [1, 8]", beam_width);
            return 1;
        }
        if (in[1] != out[1])
        {
            set_error("GPTAttention: present_key_value must alias past_key_value (in-place cache update)");
            return 1;
        }
        const int32_t* host_scalars = static_cast<const int32_t*>(in[3]); // HOST tensor [past_len, is_context]
        const int past_len = host_scalars[0];
        const bool is_context = host_scalars[1] != 0;
        const int max_input_len = inDesc[6].dims.d[0]; // value carried by the shape (gptAttentionPlugin.cpp:283-284)
        const int rot = c.rotary_embedding_dim;
        const float* table = nullptr;
        int table_len = 0;
        if (rot > 0)
        {
            table = rope_table(rot, Smax, &table_len);
            if (!table)
                return 1;
        }
        const float inv_sqrt_dh = 1.f / (sqrtf((float) c.head_size) * c.q_scaling);
        if (is_context)
        {
            ContextAttnParams p;
            if (packed)
            {
                // row of (b, s) = cu[b] + s: prefix sum of the device-resident input_lengths at the head of the workspace
                if (!ws || B + 1 > (int) (kCuBytes / 4))
                {
                    set_error("GPTAttention: packed inputs need the workspace (batch <= %d)", (int) (kCuBytes / 4) - 1);
                    return 1;
                }
                S = max_input_len;
                if (launch_exclusive_scan_i32(static_cast<int32_t*>(ws), static_cast<const int32_t*>(in[5]), B, stream))
                    return 1;
                p.cu_seqlens = static_cast<const int32_t*>(ws);
                ws = static_cast<char*>(ws) + kCuBytes;
            }
            p.batch = B;
            p.seq = S;
            p.num_heads = c.num_heads;
            p.head_size = c.head_size;
            p.rotary_dim = rot;
            p.neox = c.neox_rotary_style;
            p.inv_sqrt_dh = inv_sqrt_dh;
            p.int8_kv = c.int8_kv_cache;
            p.max_seq_len = Smax;
            p.qkv = const_cast<void*>(in[0]);
            p.kv_cache = out[1];
            if (paged)
            {
                // the prompt of batch entry b fills the blocks of table row (b, hypothesis 0): with beam search the table has
                // beam_width rows per entry while the context phase runs one sequence per entry
                p.block_pointers = block_pointers;
                p.tokens_per_block = tokens_per_block;
                p.max_blocks_per_seq = max_blocks;
                const int rows = inDesc[block_pointers_idx()].dims.d[0] * inDesc[block_pointers_idx()].dims.d[1];
                p.cache_seq_stride = rows == B ? 1 : beam_width;
            }
            p.input_lengths = static_cast<const int32_t*>(in[5]);
            p.kv_scale_orig_quant = c.int8_kv_cache ? static_cast<const float*>(in[8]) : nullptr;
            p.rope_table = table;
            p.rope_table_len = table_len;
            p.out = out[0];
            p.workspace = ws; // V^T scratch of the MFMA path (NULL -> wave-per-query kernel)
            return launch_context_attention(p, stream) ? 1 : 0;
        }
        if (S != 1)
        {
            set_error("GPTAttention: generation step expects seq_len == 1, got %d", S);
            return 1;
        }
        MmhaParams p;
        p.batch = B;
        p.num_heads = c.num_heads;
        p.head_size = c.head_size;
        p.rotary_dim = rot;
        p.neox = c.neox_rotary_style;
        p.inv_sqrt_dh = inv_sqrt_dh;
        p.int8_kv = c.int8_kv_cache;
        p.max_seq_len = Smax;
        p.max_input_len = max_input_len;
        p.qkv = in[0];
        p.kv_cache = out[1];
        p.sequence_length = static_cast<const int32_t*>(in[2]);
        p.input_lengths = static_cast<const int32_t*>(in[5]);
        p.masked_tokens = static_cast<const int32_t*>(in[4]);
        p.timestep_host = past_len;
        if (beam_width > 1)
        {
            if (inDesc[7].dims.d[0] * beam_width != B)
            {
                set_error("GPTAttention: %d sequences but cache_indirection is [%d, %d, ...]", B, inDesc[7].dims.d[0], beam_width);
                return 1;
            }
            p.cache_indirection = static_cast<const int32_t*>(in[7]);
            p.beam_width = beam_width;
        }
        if (paged)
        {
            p.block_pointers = block_pointers;
            p.tokens_per_block = tokens_per_block;
            p.max_blocks_per_seq = max_blocks;
        }
        if (c.int8_kv_cache)
        {
            p.kv_scale_orig_quant = static_cast<const float*>(in[8]);
            p.kv_scale_quant_orig = static_cast<const float*>(in[9]);
        }
        p.rope_table = table;
        p.rope_table_len = table_len;
        p.out = out[0];
        p.workspace = ws;
        if (!ws)
        {
            set_error("GPTAttention: workspace is null");
            return 1;
        }
        // the split merge runs inside the attention launch (the last split of a head to arrive: mmha_decode.hip step 6) - the same
        // code path as the session's decode step; its tickets sit at the head of the workspace and are zeroed per enqueue.  Caches
        // that need more than 16 splits take the finest split with its combine launch.
        int tc = 0, ns = 0;
        const size_t tick = mmha_ticket_bytes(B, c.num_heads);
        if (mmha_split_layout(c.head_size, Smax, 16, B, c.num_heads, &tc, &ns, nullptr) == 0 && ns <= 16)
        {
            p.rows_per_group = 16;
            if (mmha_split_layout(c.head_size, Smax, 12, B, c.num_heads, &tc, &ns, nullptr) == 0 && ns <= 8)
                p.rows_per_group = 12;
            if (mmha_reset_workspace(ws, B, c.num_heads, stream))
                return 1;
            p.tail_tickets = static_cast<uint32_t*>(ws);
        }
        p.workspace = static_cast<char*>(ws) + tick;
        return launch_mmha(p, stream) ? 1 : 0;
    }

    void serialize(Writer& w) const override { w.put(c); }
    Plugin* clone() const override { return new GPTAttentionPlugin(*this); }
};

// ================================================================================================
// Gemm   (P/gemmPlugin/gemmPlugin.cpp:121-190): C = op(A) op(B); LLaMA uses transa=0, transb=1.
// ================================================================================================
class GemmPlugin : public Plugin
{
public:
    int32_t transa = 0, transb = 0, type_id = TLLM_HALF;

    static Plugin* create(const Fields& f)
    {
        f.expect_only({"transa", "transb", "type_id"});
        auto* p = new GemmPlugin;
        p->transa = f.i32("transa");
        p->transb = f.i32("transb");
        p->type_id = f.i32("type_id");
        if (p->type_id != TLLM_HALF || p->transa != 0 || p->transb != 1)
        {
            delete p;
            throw std::runtime_error("Gemm: built for float16, transa=0, transb=1 (the Linear layers' call)");
        }
        return p;
    }
    static Plugin* deserialize(Reader& r)
    {
        auto* p = new GemmPlugin;
        p->transa = r.get<int32_t>();
        p->transb = r.get<int32_t>();
        p->type_id = r.get<int32_t>();
        r.done();
        return p;
    }
    const char* type() const override { return "Gemm"; }
    int nbOutputs() const override { return 1; }
    int outputDims(int idx, const Dims* in, int nin, Dims* out) const override
    {
        if (nin != 2 || idx != 0)
            return -1;
        *out = in[0];
        out->d[out->nbDims - 1] = in[1].d[0]; // B is [N, K]
        return 0;
    }
    int outputDtype(int, const int32_t* t, int) const override { return t[0]; }
    bool supportsFormat(int pos, const Desc* io, int nin, int nout) const override
    {
        return nin == 2 && io[pos].type == type_id && linear_fmt(io[pos]);
    }
    int enqueue(const Desc* inDesc, const Desc* outDesc, const void* const* in, void* const* out, void* ws,
        hipStream_t stream) override
    {
        const int64_t M = rows_of(inDesc[0].dims);
        const int K = inDesc[0].dims.d[inDesc[0].dims.nbDims - 1];
        const int N = inDesc[1].dims.d[0];
        if (inDesc[1].dims.nbDims != 2 || inDesc[1].dims.d[1] != K)
        {
            set_error("Gemm: B must be [N, K] with K = A's last dim");
            return 1;
        }
        GemmParams g;
        g.wtype = W_FP16;
        g.out_dtype = DT_HALF;
        g.M = (int) M;
        g.N = N;
        g.K = K;
        g.a = in[0];
        g.lda = K;
        g.w = in[1];
        g.ldw = (int64_t) K * 2;
        g.c = out[0];
        g.ldc = N;
        return launch_gemm(g, stream) ? 1 : 0;
    }
    void serialize(Writer& w) const override
    {
        w.put(transa);
        w.put(transb);
        w.put(type_id);
    }
    Plugin* clone() const override { return new GemmPlugin(*this); }
};

// ================================================================================================
// SmoothQuantGemm   (P/smoothQuantGemmPlugin/smoothQuantGemmPlugin.cpp:193-218)
//   inputs: A s8 [M.., K]; W s8 [N, K] (the reference smuggles it as fp32 [N, K/4]: both accepted);
//           scales_a f32 [M,1] | [1,1]; scales_b f32 [1,N] | [1,1]   -> [M.., N] fp16 | fp32 | int32
// ================================================================================================
class SmoothQuantGemmPlugin : public Plugin
{
public:
    int32_t per_channel = 0, per_token = 0, type_id = TLLM_HALF;
    static Plugin* create(const Fields& f)
    {
        f.expect_only({"has_per_channel_scaling", "has_per_token_scaling", "type_id"});
        auto* p = new SmoothQuantGemmPlugin;
        p->per_channel = f.i32("has_per_channel_scaling");
        p->per_token = f.i32("has_per_token_scaling");
        p->type_id = f.i32("type_id");
        if (p->type_id != TLLM_HALF && p->type_id != TLLM_FLOAT && p->type_id != TLLM_INT32)
        {
            delete p;
            throw std::runtime_error("SmoothQuantGemm: type_id must be float16, float32 or int32");
        }
        return p;
    }
    static Plugin* deserialize(Reader& r)
    {
        auto* p = new SmoothQuantGemmPlugin;
        p->per_channel = r.get<int32_t>();
        p->per_token = r.get<int32_t>();
        p->type_id = r.get<int32_t>();
        r.done();
        return p;
    }
    const char* type() const override { return "SmoothQuantGemm"; }
    int nbOutputs() const override { return 1; }
    static void weight_shape(const Desc& w, int& N, int& K)
    {
        N = w.dims.d[0];
        K = w.dims.d[1] * (w.type == TLLM_FLOAT ? 4 : 1);
    }
    int outputDims(int idx, const Dims* in, int nin, Dims* out) const override
    {
        if (nin != 4 || idx != 0)
            return -1;
        *out = in[0];
        out->d[out->nbDims - 1] = in[1].d[0];
        return 0;
    }
    int outputDtype(int, const int32_t*, int) const override { return type_id; }
    bool supportsFormat(int pos, const Desc* io, int nin, int nout) const override
    {
        if (nin != 4 || !linear_fmt(io[pos]))
            return false;
        switch (pos)
        {
        case 0: return io[pos].type == TLLM_INT8;
        case 1: return io[pos].type == TLLM_INT8 || io[pos].type == TLLM_FLOAT;
        case 2:
        case 3: return io[pos].type == TLLM_FLOAT;
        default: return io[pos].type == type_id;
        }
    }
    int enqueue(const Desc* inDesc, const Desc* outDesc, const void* const* in, void* const* out, void* ws,
        hipStream_t stream) override
    {
        const int64_t M = rows_of(inDesc[0].dims);
        const int K = inDesc[0].dims.d[inDesc[0].dims.nbDims - 1];
        int N, Kw;
        weight_shape(inDesc[1], N, Kw);
        if (Kw != K)
        {
            set_error("SmoothQuantGemm: weight K=%d does not match activation K=%d", Kw, K);
            return 1;
        }
        // the flags say how many scales the kernel reads; a tensor of a different extent is a mis-built network (e.g. a QKV
        // scale expanded to one factor per channel behind has_per_channel_scaling = 0: K and V would silently get Q's factor)
        const int64_t nb = rows_of(inDesc[3].dims) * inDesc[3].dims.d[inDesc[3].dims.nbDims - 1];
        const int64_t na = rows_of(inDesc[2].dims) * inDesc[2].dims.d[inDesc[2].dims.nbDims - 1];
        if (nb != (per_channel ? (int64_t) N : 1) || na != (per_token ? M : 1))
        {
            set_error("SmoothQuantGemm: scales_b holds %lld values and scales_a %lld, but has_per_channel_scaling = %d / "
                      "has_per_token_scaling = %d ask for %lld and %lld", (long long) nb, (long long) na, per_channel, per_token,
                (long long) (per_channel ? N : 1), (long long) (per_token ? M : 1));
            return 1;
        }
        GemmParams g;
        g.wtype = W_INT8_SQ;
        g.out_dtype = type_id == TLLM_HALF ? DT_HALF : (type_id == TLLM_FLOAT ? DT_FLOAT : DT_INT32);
        g.M = (int) M;
        g.N = N;
        g.K = K;
        g.a = in[0];
        g.lda = K;
        g.w = in[1];
        g.ldw = K;
        g.scale_row = static_cast<const float*>(in[2]);
        g.scale_col = in[3];
        g.per_channel = per_channel;
        g.per_token = per_token;
        g.c = out[0];
        g.ldc = N;
        return launch_gemm(g, stream) ? 1 : 0;
    }
    void serialize(Writer& w) const override
    {
        w.put(per_channel);
        w.put(per_token);
        w.put(type_id);
    }
    Plugin* clone() const override { return new SmoothQuantGemmPlugin(*this); }
};

// ================================================================================================
// WeightOnlyQuantMatmul   (P/weightOnlyQuantMatmulPlugin/weightOnlyQuantMatmulPlugin.cpp:162-222)
//   inputs: A fp16 [M.., K]; W: fp32-typed [K, N/4] (int8) | [K, N/8] (int4) holding this library's processed
//           bytes (layout: kernels/weight_layout.h), or int8-typed [N, ldw]; scales fp16 [N] -> fp16 [M.., N]
// ================================================================================================
class WeightOnlyQuantMatmulPlugin : public Plugin
{
public:
    int32_t type_id = TLLM_HALF, weight_type_id = 1;
    static Plugin* create(const Fields& f)
    {
        f.expect_only({"type_id", "weight_type_id"});
        auto* p = new WeightOnlyQuantMatmulPlugin;
        p->type_id = f.i32("type_id");
        p->weight_type_id = f.i32("weight_type_id");
        if (p->type_id != TLLM_HALF || (p->weight_type_id != 1 && p->weight_type_id != 2))
        {
            delete p;
            throw std::runtime_error("WeightOnlyQuantMatmul: type_id float16, weight_type_id 1 (int8) or 2 (int4)");
        }
        return p;
    }
    static Plugin* deserialize(Reader& r)
    {
        auto* p = new WeightOnlyQuantMatmulPlugin;
        p->type_id = r.get<int32_t>();
        p->weight_type_id = r.get<int32_t>();
        r.done();
        return p;
    }
    const char* type() const override { return "WeightOnlyQuantMatmul"; }
    int nbOutputs() const override { return 1; }
    int n_of(const Dims& w, int32_t wtype) const
    {
        if (wtype == TLLM_FLOAT)
            return w.d[1] * (weight_type_id == 1 ? 4 : 8); // plugin multiplies N back (…Plugin.cpp:189,206)
        return w.d[0];                                      // int8-typed [N, ldw]
    }
    int outputDims(int idx, const Dims* in, int nin, Dims* out) const override
    {
        if (nin != 3 || idx != 0)
            return -1;
        *out = in[0];
        out->d[out->nbDims - 1] = in[2].d[in[2].nbDims - 1]; // scales [N]
        return 0;
    }
    int outputDtype(int, const int32_t*, int) const override { return type_id; }
    bool supportsFormat(int pos, const Desc* io, int nin, int nout) const override
    {
        if (nin != 3 || !linear_fmt(io[pos]))
            return false;
        if (pos == 1)
            return io[pos].type == TLLM_FLOAT || io[pos].type == TLLM_INT8;
        return io[pos].type == type_id;
    }
    size_t workspaceSize(const Desc* in, int nin, const Desc* out, int nout) const override
    {
        return 0; // the prefill GEMM dequantises in its main loop (kernels/gemm_woq.hip): no fp16 image of the weights
    }
    int enqueue(const Desc* inDesc, const Desc* outDesc, const void* const* in, void* const* out, void* ws,
        hipStream_t stream) override
    {
        const int64_t M = rows_of(inDesc[0].dims);
        const int K = inDesc[0].dims.d[inDesc[0].dims.nbDims - 1];
        const int N = n_of(inDesc[1].dims, inDesc[1].type);
        const int wt = weight_type_id == 1 ? W_INT8_WOQ : W_INT4_WOQ;
        if (inDesc[1].type == TLLM_FLOAT && (inDesc[1].dims.d[0] != K || K % (weight_type_id == 1 ? 16 : 32)))
        {
            set_error("WeightOnlyQuantMatmul: fp32-view weight needs dims [K, N/%d] and K %% %d == 0",
                weight_type_id == 1 ? 4 : 8, weight_type_id == 1 ? 16 : 32);
            return 1;
        }
        if (volume(inDesc[2].dims) != N)
        {
            set_error("WeightOnlyQuantMatmul: scales must have N=%d elements", N);
            return 1;
        }
        GemmParams g;
        g.wtype = wt;
        g.out_dtype = DT_HALF;
        g.M = (int) M;
        g.N = N;
        g.K = K;
        g.a = in[0];
        g.lda = K;
        g.w = in[1];
        g.ldw = inDesc[1].type == TLLM_FLOAT ? layout::row_bytes(wt, K) : inDesc[1].dims.d[1];
        g.scale_col = in[2];
        g.c = out[0];
        g.ldc = N;
        return launch_gemm(g, stream) ? 1 : 0;
    }
    void serialize(Writer& w) const override
    {
        w.put(type_id);
        w.put(weight_type_id);
    }
    Plugin* clone() const override { return new WeightOnlyQuantMatmulPlugin(*this); }
};

// ================================================================================================
// QuantizeTensor / QuantizePerToken   (P/quantizeTensorPlugin:95-123, P/quantizePerTokenPlugin:110-139)
// ================================================================================================
class QuantizeTensorPlugin : public Plugin
{
public:
    static Plugin* create(const Fields& f)
    {
        f.expect_only({});
        return new QuantizeTensorPlugin;
    }
    static Plugin* deserialize(Reader& r)
    {
        r.done();
        return new QuantizeTensorPlugin;
    }
    const char* type() const override { return "QuantizeTensor"; }
    int nbOutputs() const override { return 1; }
    int outputDims(int idx, const Dims* in, int nin, Dims* out) const override
    {
        if (nin != 2 || idx != 0)
            return -1;
        *out = in[0];
        return 0;
    }
    int outputDtype(int, const int32_t*, int) const override { return TLLM_INT8; }
    bool supportsFormat(int pos, const Desc* io, int nin, int nout) const override
    {
        if (nin != 2 || !linear_fmt(io[pos]))
            return false;
        if (pos == 0)
            return io[pos].type == TLLM_HALF || io[pos].type == TLLM_FLOAT;
        if (pos == 1)
            return io[pos].type == TLLM_FLOAT;
        return io[pos].type == TLLM_INT8;
    }
    int enqueue(const Desc* inDesc, const Desc*, const void* const* in, void* const* out, void*, hipStream_t stream) override
    {
        const int dt = inDesc[0].type == TLLM_HALF ? DT_HALF : DT_FLOAT;
        return launch_quantize_tensor(static_cast<int8_t*>(out[0]), in[0], dt, volume(inDesc[0].dims),
                   static_cast<const float*>(in[1]), stream)
            ? 1
            : 0;
    }
    void serialize(Writer&) const override {}
    Plugin* clone() const override { return new QuantizeTensorPlugin(*this); }
};

class QuantizePerTokenPlugin : public Plugin
{
public:
    static Plugin* create(const Fields& f)
    {
        f.expect_only({});
        return new QuantizePerTokenPlugin;
    }
    static Plugin* deserialize(Reader& r)
    {
        r.done();
        return new QuantizePerTokenPlugin;
    }
    const char* type() const override { return "QuantizePerToken"; }
    int nbOutputs() const override { return 2; }
    int outputDims(int idx, const Dims* in, int nin, Dims* out) const override
    {
        if (nin != 1)
            return -1;
        *out = in[0];
        if (idx == 1)
            out->d[out->nbDims - 1] = 1; // [M.., 1]
        return 0;
    }
    int outputDtype(int idx, const int32_t*, int) const override { return idx == 0 ? TLLM_INT8 : TLLM_FLOAT; }
    bool supportsFormat(int pos, const Desc* io, int nin, int nout) const override
    {
        if (nin != 1 || !linear_fmt(io[pos]))
            return false;
        if (pos == 0)
            return io[pos].type == TLLM_HALF || io[pos].type == TLLM_FLOAT;
        return io[pos].type == (pos == 1 ? TLLM_INT8 : TLLM_FLOAT);
    }
    int enqueue(const Desc* inDesc, const Desc*, const void* const* in, void* const* out, void*, hipStream_t stream) override
    {
        const int dt = inDesc[0].type == TLLM_HALF ? DT_HALF : DT_FLOAT;
        const int64_t cols = inDesc[0].dims.d[inDesc[0].dims.nbDims - 1];
        return launch_quantize_per_token(static_cast<int8_t*>(out[0]), in[0], dt, rows_of(inDesc[0].dims), cols,
                   static_cast<float*>(out[1]), stream)
            ? 1
            : 0;
    }
    void serialize(Writer&) const override {}
    Plugin* clone() const override { return new QuantizePerTokenPlugin(*this); }
};

// ================================================================================================
// Rmsnorm / RmsnormQuantization / LayernormQuantization — the reference's plugin plus its RMSNorm analogue
// (P/layernormQuantizationPlugin/layernormQuantizationPlugin.cpp:124-166; SURVEY "fact 1": LLaMA needs RMSNorm).
//   Rmsnorm:              inputs x fp16 [M.., N], weight fp16 [N]                    -> y fp16
//   RmsnormQuantization:  inputs x, weight, scale f32 [1] (ignored when dyn_act_scaling)
//                         -> q s8 [M.., N] (+ f32 [M.., 1] dynamic scales when dyn_act_scaling)
//   LayernormQuantization: inputs x, weight, bias, scale f32 [1]; fields eps, use_diff_of_squares, dyn_act_scaling, type_id
// fields: eps f32, (dyn_act_scaling i32,) type_id i32
// ================================================================================================
class RmsnormPlugin : public Plugin
{
public:
    float eps = 1e-6f;
    int32_t quant = 0, dyn = 0, type_id = TLLM_HALF;
    int32_t layernorm = 0, diff_of_squares = 1; // LayernormQuantization: inputs x, weight, bias, scale
    static Plugin* create_layernorm_quant(const Fields& f)
    {
        // fields of PY/quantization/functional.py:92-108
        f.expect_only({"eps", "use_diff_of_squares", "dyn_act_scaling", "type_id"});
        auto* p = new RmsnormPlugin;
        p->quant = 1;
        p->layernorm = 1;
        p->eps = f.f32("eps");
        p->diff_of_squares = f.i32("use_diff_of_squares");
        p->dyn = f.i32("dyn_act_scaling");
        p->type_id = f.i32("type_id");
        return check(p);
    }
    static Plugin* create_plain(const Fields& f)
    {
        f.expect_only({"eps", "type_id"});
        auto* p = new RmsnormPlugin;
        p->eps = f.f32("eps");
        p->type_id = f.i32("type_id");
        return check(p);
    }
    static Plugin* create_quant(const Fields& f)
    {
        f.expect_only({"eps", "dyn_act_scaling", "type_id"});
        auto* p = new RmsnormPlugin;
        p->quant = 1;
        p->eps = f.f32("eps");
        p->dyn = f.i32("dyn_act_scaling");
        p->type_id = f.i32("type_id");
        return check(p);
    }
    static Plugin* check(RmsnormPlugin* p)
    {
        if (p->type_id != TLLM_HALF)
        {
            delete p;
            throw std::runtime_error("Rmsnorm: only float16");
        }
        return p;
    }
    static Plugin* deserialize_plain(Reader& r) { return deser(r, 0); }
    static Plugin* deserialize_quant(Reader& r) { return deser(r, 1); }
    static Plugin* deserialize_layernorm_quant(Reader& r) { return deser(r, 1, 1); }
    static Plugin* deser(Reader& r, int quant, int layernorm = 0)
    {
        auto* p = new RmsnormPlugin;
        p->eps = r.get<float>();
        p->quant = r.get<int32_t>();
        p->dyn = r.get<int32_t>();
        p->type_id = r.get<int32_t>();
        p->layernorm = r.get<int32_t>();
        p->diff_of_squares = r.get<int32_t>();
        r.done();
        if (p->quant != quant || p->layernorm != layernorm)
        {
            delete p;
            throw std::runtime_error("Rmsnorm: serialised kind mismatch");
        }
        return p;
    }
    const char* type() const override { return layernorm ? "LayernormQuantization" : (quant ? "RmsnormQuantization" : "Rmsnorm"); }
    int nbOutputs() const override { return quant && dyn ? 2 : 1; }
    int outputDims(int idx, const Dims* in, int nin, Dims* out) const override
    {
        *out = in[0];
        if (idx == 1)
            out->d[out->nbDims - 1] = 1;
        return 0;
    }
    int outputDtype(int idx, const int32_t*, int) const override
    {
        return quant ? (idx == 0 ? TLLM_INT8 : TLLM_FLOAT) : type_id;
    }
    bool supportsFormat(int pos, const Desc* io, int nin, int nout) const override
    {
        if (!linear_fmt(io[pos]))
            return false;
        const int nparam = layernorm ? 3 : 2; // x, weight(, bias)
        if (pos < nparam)
            return io[pos].type == type_id;
        if (quant && pos == nparam)
            return io[pos].type == TLLM_FLOAT;
        return io[pos].type == outputDtype(pos - nin, nullptr, 0);
    }
    int enqueue(const Desc* inDesc, const Desc*, const void* const* in, void* const* out, void*, hipStream_t stream) override
    {
        RmsnormParams p;
        p.M = (int) rows_of(inDesc[0].dims);
        p.N = inDesc[0].dims.d[inDesc[0].dims.nbDims - 1];
        p.x = in[0];
        p.gamma = in[1];
        p.eps = eps;
        p.layernorm = layernorm;
        p.use_diff_of_squares = diff_of_squares;
        if (layernorm)
            p.beta = in[2];
        if (!quant)
            p.y = out[0];
        else
        {
            p.q = static_cast<int8_t*>(out[0]);
            if (dyn)
                p.dyn_scale_out = static_cast<float*>(out[1]);
            else
                p.static_scale = static_cast<const float*>(in[layernorm ? 3 : 2]);
        }
        return launch_rmsnorm(p, stream) ? 1 : 0;
    }
    void serialize(Writer& w) const override
    {
        w.put(eps);
        w.put(quant);
        w.put(dyn);
        w.put(type_id);
        w.put(layernorm);
        w.put(diff_of_squares);
    }
    Plugin* clone() const override { return new RmsnormPlugin(*this); }
};

// ================================================================================================
// SwiGLU: y = silu(a) * b  (what TensorRT fuses out of PY/layers/mlp.py:68-73).  inputs a, b fp16.
// ================================================================================================
class SwiGLUPlugin : public Plugin
{
public:
    int32_t type_id = TLLM_HALF;
    static Plugin* create(const Fields& f)
    {
        f.expect_only({"type_id"});
        auto* p = new SwiGLUPlugin;
        p->type_id = f.i32("type_id");
        if (p->type_id != TLLM_HALF)
        {
            delete p;
            throw std::runtime_error("SwiGLU: only float16");
        }
        return p;
    }
    static Plugin* deserialize(Reader& r)
    {
        auto* p = new SwiGLUPlugin;
        p->type_id = r.get<int32_t>();
        r.done();
        return p;
    }
    const char* type() const override { return "SwiGLU"; }
    int nbOutputs() const override { return 1; }
    int outputDims(int, const Dims* in, int, Dims* out) const override
    {
        *out = in[0];
        return 0;
    }
    int outputDtype(int, const int32_t*, int) const override { return type_id; }
    bool supportsFormat(int pos, const Desc* io, int nin, int) const override
    {
        return nin == 2 && io[pos].type == type_id && linear_fmt(io[pos]);
    }
    int enqueue(const Desc* inDesc, const Desc*, const void* const* in, void* const* out, void*, hipStream_t stream) override
    {
        return launch_swiglu(out[0], in[0], in[1], volume(inDesc[0].dims), stream) ? 1 : 0;
    }
    void serialize(Writer& w) const override { w.put(type_id); }
    Plugin* clone() const override { return new SwiGLUPlugin(*this); }
};

// ================================================================================================
// AllReduce / AllGather   (P/ncclPlugin/allreducePlugin.cpp:80-96, allgatherPlugin.cpp:84-100)
// ================================================================================================
class CollectivePlugin : public Plugin
{
public:
    bool gather = false;
    std::vector<int32_t> group;
    int32_t type_id = TLLM_HALF;
    static Plugin* make(const Fields& f, bool gather)
    {
        f.expect_only({"group", "type_id"});
        auto* p = new CollectivePlugin;
        p->gather = gather;
        p->group = f.i32s("group");
        p->type_id = f.i32("type_id");
        return p;
    }
    static Plugin* create_ar(const Fields& f) { return make(f, false); }
    static Plugin* create_ag(const Fields& f) { return make(f, true); }
    static Plugin* deser(Reader& r, bool gather)
    {
        auto* p = new CollectivePlugin;
        p->gather = gather;
        p->type_id = r.get<int32_t>();
        const int32_t n = r.get<int32_t>();
        for (int i = 0; i < n; ++i)
            p->group.push_back(r.get<int32_t>());
        r.done();
        return p;
    }
    static Plugin* deserialize_ar(Reader& r) { return deser(r, false); }
    static Plugin* deserialize_ag(Reader& r) { return deser(r, true); }
    const char* type() const override { return gather ? "AllGather" : "AllReduce"; }
    int nbOutputs() const override { return 1; }
    int outputDims(int, const Dims* in, int, Dims* out) const override
    {
        *out = in[0];
        if (gather)
            out->d[0] *= (int32_t) group.size();
        return 0;
    }
    int outputDtype(int, const int32_t* t, int) const override { return t[0]; }
    bool supportsFormat(int pos, const Desc* io, int, int) const override
    {
        return io[pos].type == type_id && linear_fmt(io[pos]);
    }
    int enqueue(const Desc* inDesc, const Desc*, const void* const* in, void* const* out, void*, hipStream_t stream) override
    {
        // build-time no-op (IS_BUILDING, P/common/plugin.h:145-157)
        const char* b = getenv("IS_BUILDING");
        if (b && b[0] == '1')
            return 0;
        const int64_t n = volume(inDesc[0].dims);
        if (gather)
            return comm::all_gather(group, in[0], out[0], n, type_id, stream) ? 1 : 0;
        if (type_id == TLLM_HALF && in[0] == out[0] && comm::p2p::usable((int) group.size(), n * 2))
            return comm::p2p::all_reduce_f16(out[0], n, stream) ? 1 : 0; // one-shot peer-to-peer path (opt-in, validated by the caller)
        return comm::all_reduce_sum(group, in[0], out[0], n, type_id, stream) ? 1 : 0;
    }
    void serialize(Writer& w) const override
    {
        w.put(type_id);
        w.put((int32_t) group.size());
        for (auto g : group)
            w.put(g);
    }
    Plugin* clone() const override { return new CollectivePlugin(*this); }
};

const std::vector<Creator>& registry()
{
    static const std::vector<Creator> r = {
        {"GPTAttention", &GPTAttentionPlugin::create, &GPTAttentionPlugin::deserialize},
        {"Gemm", &GemmPlugin::create, &GemmPlugin::deserialize},
        {"SmoothQuantGemm", &SmoothQuantGemmPlugin::create, &SmoothQuantGemmPlugin::deserialize},
        {"WeightOnlyQuantMatmul", &WeightOnlyQuantMatmulPlugin::create, &WeightOnlyQuantMatmulPlugin::deserialize},
        {"QuantizeTensor", &QuantizeTensorPlugin::create, &QuantizeTensorPlugin::deserialize},
        {"QuantizePerToken", &QuantizePerTokenPlugin::create, &QuantizePerTokenPlugin::deserialize},
        {"Rmsnorm", &RmsnormPlugin::create_plain, &RmsnormPlugin::deserialize_plain},
        {"RmsnormQuantization", &RmsnormPlugin::create_quant, &RmsnormPlugin::deserialize_quant},
        {"LayernormQuantization", &RmsnormPlugin::create_layernorm_quant, &RmsnormPlugin::deserialize_layernorm_quant},
        {"SwiGLU", &SwiGLUPlugin::create, &SwiGLUPlugin::deserialize},
        {"AllReduce", &CollectivePlugin::create_ar, &CollectivePlugin::deserialize_ar},
        {"AllGather", &CollectivePlugin::create_ag, &CollectivePlugin::deserialize_ag},
    };
    return r;
}

} // namespace plugins
} // namespace tllm
