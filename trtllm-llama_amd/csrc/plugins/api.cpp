// extern "C" surface of include/tllm_plugin_api.h.
#include "comm.h"
#include "plugin_base.h"
#include "../kernels/weight_layout.h"
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <mutex>

namespace tllm
{

static thread_local char g_err[2048] = {0};

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const char* last_error()
{
    return g_err;
}

} // namespace tllm

using namespace tllm;
using namespace tllm::plugins;

struct tllm_plugin
{
    Plugin* impl;
};

namespace
{
std::once_flag g_init_once;
bool g_inited = false;

inline float half_bits_to_float(uint16_t h)
{
    const uint32_t sign = (h & 0x8000u) << 16;
    const uint32_t exp = (h >> 10) & 0x1f;
    const uint32_t man = h & 0x3ffu;
    uint32_t f;
    if (exp == 0)
    {
        if (man == 0)
            f = sign;
        else
        {
            int e = -1;
            uint32_t m = man;
            do
            {
                ++e;
                m <<= 1;
            } while ((m & 0x400u) == 0);
            f = sign | ((uint32_t) (127 - 15 - e) << 23) | ((m & 0x3ffu) << 13);
        }
    }
    else if (exp == 31)
        f = sign | 0x7f800000u | (man << 13);
    else
        f = sign | ((exp + 127 - 15) << 23) | (man << 13);
    float r;
    std::memcpy(&r, &f, 4);
    return r;
}

inline uint16_t float_to_half_bits(float x)
{
    // round-to-nearest-even, IEEE binary16
    uint32_t f;
    std::memcpy(&f, &x, 4);
    const uint32_t sign = (f >> 16) & 0x8000u;
    f &= 0x7fffffffu;
    if (f >= 0x7f800000u)
        return (uint16_t) (sign | 0x7c00u | ((f > 0x7f800000u) ? 0x200u : 0));
    if (f >= 0x477ff000u) // >= 65520 -> inf
        return (uint16_t) (sign | 0x7c00u);
    if (f < 0x38800000u) // subnormal half
    {
        if (f < 0x33000000u)
            return (uint16_t) sign;
        const int e = (int) (f >> 23);
        uint32_t m = (f & 0x7fffffu) | 0x800000u;
        const int shift = 126 - e; // 14..24
        const uint32_t r = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1);
        const uint32_t halfway = 1u << (shift - 1);
        uint32_t h = r;
        if (rem > halfway || (rem == halfway && (r & 1)))
            ++h;
        return (uint16_t) (sign | h);
    }
    uint32_t h = ((f - 0x38000000u) >> 13);
    const uint32_t rem = f & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1)))
        ++h;
    return (uint16_t) (sign | h);
}

} // namespace

extern "C" {

bool initLibNvInferPlugins(void* logger, const char* libNamespace)
{
    (void) logger;
    (void) libNamespace;
    std::call_once(g_init_once, []() {
        (void) registry();
        g_inited = true;
    });
    return g_inited;
}

int32_t getInferLibVersion(void)
{
    return 103; // 0.1.3, the vendored tensorrt_llm package version (T/setup.py:22)
}

const char* tllm_last_error(void)
{
    return last_error();
}

int32_t tllm_plugin_registry_size(void)
{
    return (int32_t) registry().size();
}

const char* tllm_plugin_registry_name(int32_t i)
{
    if (i < 0 || i >= (int32_t) registry().size())
        return nullptr;
    return registry()[i].name;
}

tllm_plugin_t tllm_plugin_create(
    const char* name, const char* version, const char* ns, const tllm_plugin_field_t* fields, int32_t nbFields)
{
    if (!name || !version || !ns || std::strcmp(version, "1") != 0 || std::strcmp(ns, "tensorrt_llm") != 0)
    {
        set_error("plugin creator lookup failed: expected (name, '1', 'tensorrt_llm')");
        return nullptr;
    }
    for (auto& c : registry())
    {
        if (std::strcmp(c.name, name) == 0)
        {
            try
            {
                Fields f(fields, nbFields);
                Plugin* p = c.create(f);
                return new tllm_plugin{p};
            }
            catch (const std::exception& e)
            {
                set_error("%s: %s", name, e.what());
                return nullptr;
            }
        }
    }
    set_error("no plugin creator named '%s'", name);
    return nullptr;
}

const char* tllm_plugin_type(tllm_plugin_t p)
{
    return p ? p->impl->type() : nullptr;
}

const char* tllm_plugin_version(tllm_plugin_t p)
{
    return p ? "1" : nullptr;
}

int32_t tllm_plugin_nb_outputs(tllm_plugin_t p)
{
    return p ? p->impl->nbOutputs() : -1;
}

int32_t tllm_plugin_output_dims(
    tllm_plugin_t p, int32_t outputIndex, const tllm_dims_t* inputs, int32_t nbInputs, tllm_dims_t* out)
{
    if (!p || !inputs || !out || outputIndex < 0 || outputIndex >= p->impl->nbOutputs())
    {
        set_error("tllm_plugin_output_dims: bad arguments");
        return 1;
    }
    return p->impl->outputDims(outputIndex, inputs, nbInputs, out) ? 1 : 0;
}

int32_t tllm_plugin_output_dtype(tllm_plugin_t p, int32_t outputIndex, const int32_t* inputTypes, int32_t nbInputs)
{
    if (!p)
        return -1;
    return p->impl->outputDtype(outputIndex, inputTypes, nbInputs);
}

int32_t tllm_plugin_supports_format(
    tllm_plugin_t p, int32_t pos, const tllm_tensor_desc_t* inOut, int32_t nbInputs, int32_t nbOutputs)
{
    if (!p || !inOut || pos < 0 || pos >= nbInputs + nbOutputs)
        return 0;
    return p->impl->supportsFormat(pos, inOut, nbInputs, nbOutputs) ? 1 : 0;
}

size_t tllm_plugin_workspace_size(tllm_plugin_t p, const tllm_tensor_desc_t* inputs, int32_t nbInputs,
    const tllm_tensor_desc_t* outputs, int32_t nbOutputs)
{
    if (!p)
        return 0;
    return p->impl->workspaceSize(inputs, nbInputs, outputs, nbOutputs);
}

int32_t tllm_plugin_enqueue(tllm_plugin_t p, const tllm_tensor_desc_t* inputDesc, const tllm_tensor_desc_t* outputDesc,
    const void* const* inputs, void* const* outputs, void* workspace, tllm_stream_t stream)
{
    if (!p || !inputDesc || !outputDesc || !inputs || !outputs)
    {
        set_error("tllm_plugin_enqueue: null argument");
        return 1;
    }
    try
    {
        return p->impl->enqueue(inputDesc, outputDesc, inputs, outputs, workspace, reinterpret_cast<hipStream_t>(stream));
    }
    catch (const std::exception& e)
    {
        set_error("%s::enqueue: %s", p->impl->type(), e.what());
        return 1;
    }
}

size_t tllm_plugin_serialization_size(tllm_plugin_t p)
{
    if (!p)
        return 0;
    Writer w;
    p->impl->serialize(w);
    return w.buf.size();
}

int32_t tllm_plugin_serialize(tllm_plugin_t p, void* buffer)
{
    if (!p || !buffer)
        return 1;
    Writer w;
    p->impl->serialize(w);
    if (!w.buf.empty())
        std::memcpy(buffer, w.buf.data(), w.buf.size());
    return 0;
}

tllm_plugin_t tllm_plugin_deserialize(const char* name, const void* data, size_t length)
{
    if (!name)
        return nullptr;
    for (auto& c : registry())
    {
        if (std::strcmp(c.name, name) == 0)
        {
            try
            {
                Reader r(data, length);
                return new tllm_plugin{c.deserialize(r)};
            }
            catch (const std::exception& e)
            {
                set_error("%s: %s", name, e.what());
                return nullptr;
            }
        }
    }
    set_error("no plugin creator named '%s'", name);
    return nullptr;
}

tllm_plugin_t tllm_plugin_clone(tllm_plugin_t p)
{
    return p ? new tllm_plugin{p->impl->clone()} : nullptr;
}

void tllm_plugin_destroy(tllm_plugin_t p)
{
    if (p)
    {
        delete p->impl;
        delete p;
    }
}

int32_t tllm_comm_get_unique_id(void* id128)
{
    return comm::get_unique_id(id128) ? 1 : 0;
}

int32_t tllm_comm_init_rank(const int32_t* group, int32_t groupSize, int32_t rank, const void* id128)
{
    if (!group || groupSize < 1 || !id128)
    {
        set_error("tllm_comm_init_rank: bad arguments");
        return 1;
    }
    return comm::init_rank(std::vector<int32_t>(group, group + groupSize), rank, id128) ? 1 : 0;
}

int32_t tllm_comm_destroy_all(void)
{
    return comm::destroy_all();
}

int32_t tllm_comm_p2p_create(int32_t world, int32_t rank, int64_t max_bytes, void* handle64)
{
    return comm::p2p::create(world, rank, (size_t) max_bytes, handle64) ? 1 : 0;
}

int32_t tllm_comm_p2p_attach(const void* handles)
{
    return comm::p2p::attach(handles) ? 1 : 0;
}

int32_t tllm_comm_p2p_enable(int32_t on)
{
    return comm::p2p::enable(on != 0) ? 1 : 0;
}

void tllm_comm_p2p_enable_fused(int32_t on)
{
    comm::p2p::enable_fused(on != 0);
}

int32_t tllm_comm_p2p_state(void)
{
    return (comm::p2p::attached() ? 1 : 0) | (comm::p2p::enabled() ? 2 : 0)
        | ((comm::p2p::enabled() && comm::p2p::usable_fused_flag()) ? 4 : 0);
}

int32_t tllm_comm_p2p_all_reduce(void* buf, int64_t count, tllm_stream_t stream)
{
    return comm::p2p::all_reduce_f16(buf, count, reinterpret_cast<hipStream_t>(stream)) ? 1 : 0;
}

int32_t tllm_comm_p2p_all_reduce_residual_norm(const void* partial, void* x, const void* gamma, float eps, int32_t rows, int32_t cols,
    void* norm_out, int32_t quant, const float* quant_scale, float* dyn_scale_out, tllm_stream_t stream)
{
    if (!partial || !x || !gamma || !norm_out)
    {
        set_error("tllm_comm_p2p_all_reduce_residual_norm: null argument");
        return 1;
    }
    comm::p2p::FusedTail t;
    t.x = x;
    t.gamma = gamma;
    t.eps = eps;
    t.norm_out = norm_out;
    t.quant = quant;
    t.quant_scale = quant_scale;
    t.dyn_scale_out = dyn_scale_out;
    return comm::p2p::all_reduce_residual_norm(const_cast<void*>(partial), rows, cols, t, reinterpret_cast<hipStream_t>(stream)) ? 1 : 0;
}

void tllm_comm_p2p_set_max_spins(int32_t n)
{
    comm::p2p::set_max_spins(n);
}

int32_t tllm_comm_group_info(const int32_t* group, int32_t groupSize, int32_t* nranks, int32_t* my_index)
{
    if (!group || groupSize < 1)
    {
        set_error("tllm_comm_group_info: bad arguments");
        return 1;
    }
    int n = 0, idx = -1;
    if (comm::group_info(std::vector<int32_t>(group, group + groupSize), &n, &idx))
        return 1;
    if (nranks)
        *nranks = n;
    if (my_index)
        *my_index = idx;
    return 0;
}

int32_t tllm_comm_p2p_error(void)
{
    uint32_t e = 0;
    if (comm::p2p::error_flag(&e))
        return -1;
    return (int32_t) e;
}

// ------------------------------------------------------------------------------------------------
// weight-only quantiser + layout (host).  Arithmetic of
// K/cutlass_kernels/cutlass_preprocessors.cpp:615-721 (symmetric_quantize): per column n of W[k,n]:
//   scale = max_k |w| / 2^(bits-1)  (fp32; stored as fp16),  q = clip(round_half_away(w / scale), -2^(bits-1), 2^(bits-1)-1)
// ------------------------------------------------------------------------------------------------
static void relayout(const int8_t* q_kn_bytes, int64_t k, int64_t n, int32_t bits, int8_t* out)
{
    // q_kn_bytes: int8 [k, n] (bits == 8) or packed int4 [k, n/2] low nibble first (bits == 4)
    if (bits == 8)
    {
        const int64_t ldw = layout::row_bytes(1, k);
        std::memset(out, 128, (size_t) (n * ldw));
        for (int64_t kk = 0; kk < k; ++kk)
            for (int64_t nn = 0; nn < n; ++nn)
                reinterpret_cast<uint8_t*>(out)[nn * ldw + kk] = (uint8_t) ((int) q_kn_bytes[kk * n + nn] + 128);
    }
    else
    {
        const int64_t ldw = layout::row_bytes(2, k);
        std::memset(out, 0x88, (size_t) (n * ldw));
        uint8_t* o = reinterpret_cast<uint8_t*>(out);
        for (int64_t kk = 0; kk < k; ++kk)
            for (int64_t nn = 0; nn < n; ++nn)
            {
                const uint8_t byte = (uint8_t) q_kn_bytes[kk * (n / 2) + nn / 2];
                int q = (nn & 1) ? (byte >> 4) : (byte & 0xf);
                if (q >= 8)
                    q -= 16; // sign-extend the nibble
                const uint32_t nib = (uint32_t) (q + 8);
                const int64_t word = kk / 8;
                const int pos = layout::kElemToNibble[kk % 8];
                uint8_t* w = o + nn * ldw + word * 4 + pos / 2;
                if (pos & 1)
                    *w = (uint8_t) ((*w & 0x0f) | (nib << 4));
                else
                    *w = (uint8_t) ((*w & 0xf0) | nib);
            }
    }
}

int32_t tllm_preprocess_weights_for_mixed_gemm(
    const int8_t* quantized_kn, int64_t k, int64_t n, int32_t bits, int8_t* processed_out)
{
    if (!quantized_kn || !processed_out || (bits != 8 && bits != 4) || k <= 0 || n <= 0 || (bits == 4 && (n & 1)))
    {
        set_error("tllm_preprocess_weights_for_mixed_gemm: bad arguments");
        return 1;
    }
    relayout(quantized_kn, k, n, bits, processed_out);
    return 0;
}

int32_t tllm_symmetric_quantize_last_axis(const uint16_t* weight_kn, int64_t k, int64_t n, int32_t bits,
    int8_t* processed_out, uint16_t* scales_out, int8_t* unprocessed_out)
{
    if (!weight_kn || !processed_out || !scales_out || (bits != 8 && bits != 4) || k <= 0 || n <= 0
        || (bits == 4 && (n & 1)))
    {
        set_error("tllm_symmetric_quantize_last_axis: bad arguments");
        return 1;
    }
    const float quant_range = (float) (1 << (bits - 1)); // 128 or 8
    std::vector<float> colmax((size_t) n, 0.f);
    for (int64_t kk = 0; kk < k; ++kk)
        for (int64_t nn = 0; nn < n; ++nn)
        {
            const float a = std::fabs(half_bits_to_float(weight_kn[kk * n + nn]));
            if (a > colmax[nn])
                colmax[nn] = a;
        }
    std::vector<float> scale((size_t) n);
    for (int64_t nn = 0; nn < n; ++nn)
    {
        scale[nn] = colmax[nn] / quant_range;
        scales_out[nn] = float_to_half_bits(scale[nn]);
    }
    const int64_t bytes_per_row = bits == 8 ? n : n / 2;
    std::vector<int8_t> q((size_t) (k * bytes_per_row), 0);
    const int lo = -(1 << (bits - 1)), hi = (1 << (bits - 1)) - 1;
    for (int64_t kk = 0; kk < k; ++kk)
        for (int64_t nn = 0; nn < n; ++nn)
        {
            const float w = half_bits_to_float(weight_kn[kk * n + nn]);
            const float s = scale[nn];
            float scaled = s != 0.f ? w / s : 0.f;
            int v = (int) std::round(scaled); // half away from zero (cutlass_preprocessors.cpp:683)
            v = v < lo ? lo : (v > hi ? hi : v);
            if (bits == 8)
                q[kk * n + nn] = (int8_t) v;
            else
            {
                uint8_t& b = reinterpret_cast<uint8_t&>(q[kk * (n / 2) + nn / 2]);
                const uint8_t nib = (uint8_t) (v & 0xf);
                b = (nn & 1) ? (uint8_t) ((b & 0x0f) | (nib << 4)) : (uint8_t) ((b & 0xf0) | nib);
            }
        }
    if (unprocessed_out)
        std::memcpy(unprocessed_out, q.data(), q.size());
    relayout(q.data(), k, n, bits, processed_out);
    return 0;
}

} // extern "C"
