// See engine_check.h.  Method: both the traced graph and the schedule the session hard-codes are turned into the same
// hash-consed expression DAG (a node's identity = its operator, its attributes and the identities of its inputs; leaves are
// network inputs and weights by module path), so "same computation" is integer equality of the `logits` /
// `present_key_value_i` expressions, and the first traced node whose expression the schedule does not contain is the one to
// report.
#include "engine_check.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <set>

namespace tllm
{
namespace runtime
{
namespace
{

// ---------------------------------------------------------------------------------------------- a small JSON reader
struct JVal
{
    enum Type
    {
        NUL,
        BOOL,
        NUM,
        STR,
        ARR,
        OBJ
    } type
        = NUL;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj; // insertion order kept
    const JVal* get(const char* key) const
    {
        for (auto& kv : obj)
            if (kv.first == key)
                return &kv.second;
        return nullptr;
    }
};

struct JParser
{
    const char* p;
    const char* end;
    std::string err;
    void ws()
    {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r'))
            ++p;
    }
    bool fail(const char* what)
    {
        if (err.empty())
            err = what;
        return false;
    }
    bool parse_string(std::string& out)
    {
        if (p >= end || *p != '"')
            return fail("expected a string");
        ++p;
        while (p < end && *p != '"')
        {
            if (*p == '\\')
            {
                if (++p >= end)
                    return fail("bad escape");
                switch (*p)
                {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u':
                    if (end - p < 5)
                        return fail("bad \\u escape");
                    out += '?'; // names in a traced network are ASCII; anything else cannot match a schedule name anyway
                    p += 4;
                    break;
                default: out += *p; break;
                }
                ++p;
            }
            else
                out += *p++;
        }
        if (p >= end)
            return fail("unterminated string");
        ++p;
        return true;
    }
    bool parse(JVal& v, int depth = 0)
    {
        if (depth > 64)
            return fail("nesting too deep");
        ws();
        if (p >= end)
            return fail("unexpected end");
        if (*p == '{')
        {
            v.type = JVal::OBJ;
            ++p;
            ws();
            if (p < end && *p == '}')
            {
                ++p;
                return true;
            }
            for (;;)
            {
                ws();
                std::string k;
                if (!parse_string(k))
                    return false;
                ws();
                if (p >= end || *p != ':')
                    return fail("expected ':'");
                ++p;
                v.obj.emplace_back(k, JVal());
                if (!parse(v.obj.back().second, depth + 1))
                    return false;
                ws();
                if (p < end && *p == ',')
                {
                    ++p;
                    continue;
                }
                if (p < end && *p == '}')
                {
                    ++p;
                    return true;
                }
                return fail("expected ',' or '}'");
            }
        }
        if (*p == '[')
        {
            v.type = JVal::ARR;
            ++p;
            ws();
            if (p < end && *p == ']')
            {
                ++p;
                return true;
            }
            for (;;)
            {
                v.arr.emplace_back();
                if (!parse(v.arr.back(), depth + 1))
                    return false;
                ws();
                if (p < end && *p == ',')
                {
                    ++p;
                    continue;
                }
                if (p < end && *p == ']')
                {
                    ++p;
                    return true;
                }
                return fail("expected ',' or ']'");
            }
        }
        if (*p == '"')
        {
            v.type = JVal::STR;
            return parse_string(v.str);
        }
        if (end - p >= 4 && !std::strncmp(p, "true", 4))
        {
            v.type = JVal::BOOL;
            v.b = true;
            p += 4;
            return true;
        }
        if (end - p >= 5 && !std::strncmp(p, "false", 5))
        {
            v.type = JVal::BOOL;
            p += 5;
            return true;
        }
        if (end - p >= 4 && !std::strncmp(p, "null", 4))
        {
            p += 4;
            return true;
        }
        char* e = nullptr;
        v.num = std::strtod(p, &e);
        if (e == p)
            return fail("unexpected character");
        v.type = JVal::NUM;
        p = e;
        return true;
    }
};

std::string num(double x)
{
    char buf[40];
    std::snprintf(buf, sizeof buf, "%.6g", x);
    return buf;
}

using Fields = std::map<std::string, std::vector<double>>;

std::string fields_text(const Fields& f)
{
    std::string s;
    for (auto& kv : f)
    {
        s += kv.first + "=";
        for (size_t i = 0; i < kv.second.size(); ++i)
            s += (i ? "," : "") + num(kv.second[i]);
        s += ";";
    }
    return s;
}

// ---------------------------------------------------------------------------------------------- the expression DAG
struct Dag
{
    std::map<std::string, int> ids;
    std::vector<std::string> text; // id -> readable form (for messages)
    int intern(const std::string& key)
    {
        auto it = ids.find(key);
        if (it != ids.end())
            return it->second;
        const int id = (int) text.size();
        ids.emplace(key, id);
        text.push_back(key);
        return id;
    }
    int leaf_input(const std::string& name) { return intern("in:" + name); }
    int leaf_param(const std::string& path) { return intern("param:" + path); }
    // identity of output `k` of an operator application
    int apply(const std::string& op, const std::vector<int>& in, const std::string& attrs, int k = 0)
    {
        std::string key = op + "(";
        for (size_t i = 0; i < in.size(); ++i)
            key += (i ? "," : "") + std::to_string(in[i]);
        key += ")[" + attrs + "]#" + std::to_string(k);
        return intern(key);
    }
    // a one-line description of node `id` with its inputs spelled out one level deep
    std::string describe(int id) const
    {
        if (id < 0 || id >= (int) text.size())
            return "?";
        const std::string& t = text[id];
        const size_t lp = t.find('('), rp = t.find(")[");
        if (lp == std::string::npos || rp == std::string::npos)
            return t;
        std::string out = t.substr(0, lp) + "(";
        size_t pos = lp + 1;
        bool first = true;
        while (pos < rp)
        {
            size_t c = t.find(',', pos);
            if (c == std::string::npos || c > rp)
                c = rp;
            const int in = std::atoi(t.substr(pos, c - pos).c_str());
            std::string name = in >= 0 && in < (int) text.size() ? text[in] : "?";
            const size_t q = name.find('(');
            if (q != std::string::npos)
                name = name.substr(0, q) + "(..)";
            out += (first ? "" : ", ") + name;
            first = false;
            pos = c + 1;
        }
        return out + ")" + t.substr(rp + 1);
    }
};

struct Expected
{
    Dag& g;
    std::vector<int> order; // output-0 ids in schedule order
    std::set<int> all;      // every output id
    int node(const std::string& op, const std::vector<int>& in, const std::string& attrs, int nout = 1)
    {
        const int id0 = g.apply(op, in, attrs, 0);
        order.push_back(id0);
        for (int k = 0; k < nout; ++k)
            all.insert(g.apply(op, in, attrs, k));
        return id0;
    }
    int plugin(const std::string& type, const std::vector<int>& in, const Fields& f, int nout = 1)
    {
        return node("plugin:" + type, in, fields_text(f), nout);
    }
};

} // namespace

int verify_network(const std::string& network_json, const ScheduleDesc& d, std::string& err)
{
    JVal root;
    JParser jp{network_json.data(), network_json.data() + network_json.size(), {}};
    if (!jp.parse(root) || root.type != JVal::OBJ)
    {
        err = "network_json does not parse: " + (jp.err.empty() ? std::string("not an object") : jp.err);
        return 1;
    }
    const JVal* jin = root.get("inputs");
    const JVal* jout = root.get("outputs");
    const JVal* jnodes = root.get("nodes");
    const JVal* jconst = root.get("constants");
    if (jin && jout && jnodes && !jconst)
    {
        // engines written before the builder recorded the traced constants (same TLLMENG1 container): say what to do
        err = "network_json has no 'constants' map: this engine was built by an older builder - rebuild it (build.py) with the current one";
        return 1;
    }
    if (!jin || jin->type != JVal::ARR || !jout || jout->type != JVal::ARR || !jnodes || jnodes->type != JVal::ARR || !jconst
        || jconst->type != JVal::OBJ)
    {
        err = "network_json lacks inputs / outputs / nodes / constants";
        return 1;
    }

    // ---- I/O tensor names (PY/runtime/generation.py:188-208)
    std::set<std::string> want_in = {"input_ids", "position_ids", "sequence_length", "past_key_value_length", "masked_tokens",
        "input_lengths", "max_input_length", "last_token_ids", "cache_indirection"};
    std::set<std::string> want_out = {"logits"};
    for (int i = 0; i < d.num_layers; ++i)
    {
        want_in.insert("past_key_value_" + std::to_string(i));
        want_out.insert("present_key_value_" + std::to_string(i));
        if (d.paged)
            want_in.insert("kv_cache_block_pointers_" + std::to_string(i));
    }
    std::set<std::string> have_in, have_out;
    for (auto& v : jin->arr)
        have_in.insert(v.str);
    for (auto& v : jout->arr)
        have_out.insert(v.str);
    for (auto& n : want_in)
        if (!have_in.count(n))
        {
            err = "the engine's network has no input tensor '" + n + "'";
            return 1;
        }
    for (auto& n : have_in)
        if (!want_in.count(n))
        {
            err = "the engine's network takes an input tensor '" + n + "' this runtime does not feed";
            return 1;
        }
    if (have_out != want_out)
    {
        for (auto& n : want_out)
            if (!have_out.count(n))
            {
                err = "the engine's network has no output tensor '" + n + "'";
                return 1;
            }
        for (auto& n : have_out)
            if (!want_out.count(n))
            {
                err = "the engine's network marks an output '" + n + "' this runtime does not produce";
                return 1;
            }
    }

    // ---- the schedule this runtime executes (runtime/session.cpp run_context / run_decode_step), as expressions
    Dag g;
    Expected E{g, {}, {}};
    auto in = [&](const std::string& n) { return g.leaf_input(n); };
    auto P = [&](const std::string& n) { return g.leaf_param(n); };
    const double HALF = 1; // nvinfer1::DataType::kHALF
    std::vector<double> group;
    for (int r = 0; r < d.tp; ++r)
        group.push_back(r);
    const Fields gemm_f = {{"transa", {0}}, {"transb", {1}}, {"type_id", {HALF}}};
    const Fields woq_f = {{"type_id", {HALF}}, {"weight_type_id", {d.int4 ? 2.0 : 1.0}}};
    const Fields coll_f = {{"group", group}, {"type_id", {HALF}}};
    auto sq_f = [&](int gemm_index) {
        return Fields{{"has_per_channel_scaling", {(double) d.per_channel[gemm_index]}},
            {"has_per_token_scaling", {d.per_token ? 1.0 : 0.0}}, {"type_id", {HALF}}};
    };
    const Fields nq_f = {{"eps", {(double) d.eps}}, {"dyn_act_scaling", {d.per_token ? 1.0 : 0.0}}, {"type_id", {HALF}}};
    const std::string norm_attr = "eps=" + num(d.eps);
    // one linear layer of the model in this quantisation mode: x is fp16 (quantised here with `static_scale` when SmoothQuant
    // needs a quantiser in front), or already (int8, scales) from a fused norm + quantiser
    auto linear = [&](const std::string& prefix, int gemm_index, int x16, int xq, int xs, const std::string& static_scale) {
        if (d.sq)
        {
            if (xq < 0)
            {
                if (d.per_token)
                {
                    xq = E.plugin("QuantizePerToken", {x16}, {}, 2);
                    xs = g.apply("plugin:QuantizePerToken", {x16}, "", 1);
                }
                else
                    xq = E.plugin("QuantizeTensor", {x16, P(static_scale)}, {});
            }
            const int sa = d.per_token ? xs : P(prefix + ".act_scale");
            return E.plugin("SmoothQuantGemm", {xq, P(prefix + ".weight"), sa, P(prefix + ".per_channel_scale")}, sq_f(gemm_index));
        }
        if (d.woq)
            return E.plugin("WeightOnlyQuantMatmul", {x16, P(prefix + ".weight"), P(prefix + ".per_channel_scale")}, woq_f);
        return E.plugin("Gemm", {x16, P(prefix + ".weight")}, gemm_f);
    };
    // RMSNorm in front of a column-parallel GEMM: (fp16) or (int8, per-token scales | static)
    auto norm = [&](int x, const std::string& prefix, int& h16, int& hq, int& hs) {
        h16 = hq = hs = -1;
        if (!d.sq)
        {
            h16 = E.node("rms_norm", {x, P(prefix + ".weight")}, norm_attr);
            return;
        }
        if (d.per_token)
        {
            const std::vector<int> ins = {x, P(prefix + ".weight"), P(prefix + ".weight")}; // the scale port is unused
            hq = E.plugin("RmsnormQuantization", ins, nq_f, 2);
            hs = g.apply("plugin:RmsnormQuantization", ins, fields_text(nq_f), 1);
        }
        else
            hq = E.plugin("RmsnormQuantization", {x, P(prefix + ".weight"), P(prefix + ".scale_to_int")}, nq_f);
    };
    int x = E.node("embedding", {in("input_ids"), P("vocab_embedding.weight")}, "");
    std::vector<int> present(d.num_layers, -1);
    for (int i = 0; i < d.num_layers; ++i)
    {
        const std::string p = "layers." + std::to_string(i) + ".";
        int h16, hq, hs;
        norm(x, p + "input_layernorm", h16, hq, hs);
        const int qkv = linear(p + "attention.qkv", 0, h16, hq, hs, "");
        std::vector<int> ain = {qkv, in("past_key_value_" + std::to_string(i)), in("sequence_length"), in("past_key_value_length"),
            in("masked_tokens"), in("input_lengths"), in("max_input_length"), in("cache_indirection")};
        if (d.int8_kv)
        {
            ain.push_back(P(p + "attention.kv_orig_quant_scale"));
            ain.push_back(P(p + "attention.kv_quant_orig_scale"));
        }
        if (d.paged)
            ain.push_back(in("kv_cache_block_pointers_" + std::to_string(i)));
        const Fields af = {{"num_heads", {(double) d.heads_per_rank}}, {"head_size", {(double) d.head_size}}, {"unidirectional", {1}},
            {"q_scaling", {1}}, {"rotary_embedding_dim", {(double) d.head_size}}, {"neox_rotary_style", {d.neox ? 1.0 : 0.0}},
            {"context_fmha_type", {0}}, {"multi_block_mode", {0}}, {"multi_query_mode", {0}},
            {"int8_kv_cache", {d.int8_kv ? 1.0 : 0.0}}, {"fp8_kv_cache", {0}}, {"remove_input_padding", {d.packed ? 1.0 : 0.0}},
            {"mask_type", {1}}, {"paged_kv_cache", {d.paged ? 1.0 : 0.0}}, {"type_id", {HALF}}, {"in_flight_batching", {0}}};
        const int ctx = E.plugin("GPTAttention", ain, af, 2);
        present[i] = g.apply("plugin:GPTAttention", ain, fields_text(af), 1);
        int o = linear(p + "attention.dense", 1, ctx, -1, -1, p + "attention.quantization_scaling_factor");
        if (d.tp > 1)
            o = E.plugin("AllReduce", {o}, coll_f);
        x = E.node("add", {x, o}, "");
        norm(x, p + "post_layernorm", h16, hq, hs);
        const int fc = linear(p + "mlp.fc", 2, h16, hq, hs, "");
        const int act = E.node("silu", {fc}, "");
        const int gate = linear(p + "mlp.gate", 3, h16, hq, hs, "");
        const int inter = E.node("mul", {act, gate}, "");
        int m = linear(p + "mlp.proj", 4, inter, -1, -1, p + "mlp.quantization_scaling_factor");
        if (d.tp > 1)
            m = E.plugin("AllReduce", {m}, coll_f);
        x = E.node("add", {x, m}, "");
    }
    const int hf = E.node("rms_norm", {x, P("ln_f.weight")}, norm_attr);
    const int last = E.node("gather_last_token_logits", {hf, in("last_token_ids")}, d.packed ? "rip=1" : "rip=0");
    int logits = E.plugin("Gemm", {last, P("lm_head.weight")}, gemm_f); // lm_head stays fp16 in every mode (Q/quant.py:58)
    if (d.tp > 1)
        logits = E.plugin("AllGather", {logits}, coll_f);

    // ---- the traced network, through the same interning
    std::map<std::string, int> tensor; // traced tensor name -> expression id
    for (auto& v : jin->arr)
        tensor[v.str] = g.leaf_input(v.str);
    for (auto& kv : jconst->obj)
        tensor[kv.first] = kv.second.type == JVal::STR && !kv.second.str.empty() ? g.leaf_param(kv.second.str)
                                                                                 : g.intern("const:" + kv.first);
    std::map<std::string, int> marked;
    int ordinal = 0;
    for (size_t ni = 0; ni < jnodes->arr.size(); ++ni)
    {
        const JVal& n = jnodes->arr[ni];
        const JVal *jop = n.get("op"), *ji = n.get("inputs"), *jo = n.get("outputs"), *ja = n.get("attrs");
        if (!jop || jop->type != JVal::STR || !ji || ji->type != JVal::ARR || !jo || jo->type != JVal::ARR)
        {
            err = "network_json: malformed node #" + std::to_string(ni);
            return 1;
        }
        const std::string& op = jop->str;
        if (op == "shape" || op == "assertion" || op == "constant")
            continue; // build-time shape checks carry no computation
        std::vector<int> ins;
        for (auto& t : ji->arr)
        {
            auto it = tensor.find(t.str);
            if (it == tensor.end())
            {
                err = "network_json: node #" + std::to_string(ni) + " (" + op + ") reads tensor '" + t.str + "' that nothing produces";
                return 1;
            }
            ins.push_back(it->second);
        }
        if (op == "mark_output")
        {
            if (ins.size() != 1 || jo->arr.size() != 1)
            {
                err = "network_json: malformed mark_output";
                return 1;
            }
            marked[jo->arr[0].str] = ins[0];
            continue;
        }
        std::string name = op, attrs;
        if (op == "plugin")
        {
            const JVal* pt = ja ? ja->get("plugin_type") : nullptr;
            const JVal* pf = ja ? ja->get("fields") : nullptr;
            if (!pt || pt->type != JVal::STR || !pf || pf->type != JVal::OBJ)
            {
                err = "network_json: plugin node #" + std::to_string(ni) + " without plugin_type / fields";
                return 1;
            }
            name = "plugin:" + pt->str;
            Fields f;
            for (auto& kv : pf->obj)
            {
                std::vector<double> vals;
                if (kv.second.type == JVal::ARR)
                    for (auto& e : kv.second.arr)
                        vals.push_back(e.type == JVal::BOOL ? (e.b ? 1.0 : 0.0) : e.num);
                else
                    vals.push_back(kv.second.type == JVal::BOOL ? (kv.second.b ? 1.0 : 0.0) : kv.second.num);
                f[kv.first] = vals;
            }
            attrs = fields_text(f);
        }
        else if (op == "rms_norm")
        {
            const JVal* e = ja ? ja->get("eps") : nullptr;
            attrs = "eps=" + num(e ? e->num : -1.0);
        }
        else if (op == "gather_last_token_logits")
        {
            const JVal* r = ja ? ja->get("remove_input_padding") : nullptr;
            attrs = (r && ((r->type == JVal::BOOL && r->b) || (r->type == JVal::NUM && r->num != 0))) ? "rip=1" : "rip=0";
        }
        const int id0 = g.apply(name, ins, attrs, 0);
        for (size_t k = 0; k < jo->arr.size(); ++k)
            tensor[jo->arr[k].str] = g.apply(name, ins, attrs, (int) k);
        if (!E.all.count(id0))
        {
            err = "the engine's network is not the LLaMA schedule this runtime executes: node #" + std::to_string(ni) + " ";
            const std::string got = g.describe(id0);
            const std::string want = ordinal < (int) E.order.size() ? g.describe(E.order[ordinal]) : std::string();
            // same operator on the same inputs: only name the attributes that differ
            const size_t gb = got.find(")["), wb = want.find(")[");
            if (!want.empty() && gb != std::string::npos && wb != std::string::npos && got.substr(0, gb) == want.substr(0, wb))
            {
                err += "(" + got.substr(0, got.find('(')) + ") differs from step " + std::to_string(ordinal) + " of the schedule in:";
                auto items = [](const std::string& t, size_t from) {
                    std::map<std::string, std::string> m;
                    size_t p0 = from + 2;
                    const size_t end = t.rfind(']');
                    while (p0 < end)
                    {
                        size_t semi = t.find(';', p0);
                        if (semi == std::string::npos || semi > end)
                            semi = end;
                        const std::string kv = t.substr(p0, semi - p0);
                        const size_t eq = kv.find('=');
                        if (eq != std::string::npos)
                            m[kv.substr(0, eq)] = kv.substr(eq + 1);
                        p0 = semi + 1;
                    }
                    return m;
                };
                auto a = items(got, gb), b = items(want, wb);
                for (auto& kv : a)
                    if (!b.count(kv.first))
                        err += " " + kv.first + " = " + kv.second + " (the runtime has no such field)";
                    else if (b[kv.first] != kv.second)
                        err += " " + kv.first + " = " + kv.second + " (the runtime executes " + b[kv.first] + ")";
                for (auto& kv : b)
                    if (!a.count(kv.first))
                        err += " " + kv.first + " missing (the runtime executes " + kv.second + ")";
            }
            else
            {
                err += "= " + got;
                err += want.empty() ? "  -- the runtime's schedule has no further step here"
                                    : "  -- at this point (step " + std::to_string(ordinal) + ") the runtime executes " + want;
            }
            return 1;
        }
        ++ordinal;
    }
    auto check_out = [&](const std::string& name, int want) {
        auto it = marked.find(name);
        if (it == marked.end())
        {
            err = "the engine's network never marks output '" + name + "'";
            return 1;
        }
        if (it->second != want)
        {
            err = "output '" + name + "' of the engine's network is " + g.describe(it->second) + ", the runtime produces "
                + g.describe(want);
            return 1;
        }
        return 0;
    };
    if (check_out("logits", logits))
        return 1;
    for (int i = 0; i < d.num_layers; ++i)
        if (check_out("present_key_value_" + std::to_string(i), present[i]))
            return 1;
    if (ordinal != (int) E.order.size())
    {
        err = "the engine's network has " + std::to_string(ordinal) + " computing nodes, the runtime's schedule "
            + std::to_string(E.order.size());
        return 1;
    }
    return 0;
}

} // namespace runtime
} // namespace tllm
