// Internal C++ side of the flat plugin ABI (include/tllm_plugin_api.h).
// One class per reference plugin; the vtable shape is IPluginV2DynamicExt's.
#pragma once
#include "../../../include/tllm_plugin_api.h"
#include "../kernels/kernels.h"
#include <cstring>
#include <hip/hip_runtime.h>
#include <map>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

namespace tllm
{
namespace plugins
{

using Desc = tllm_tensor_desc_t;
using Dims = tllm_dims_t;

// Parsed PluginFieldCollection.  get<T>() throws when the field is missing or has the wrong type — the
// creator catches and returns NULL, as the reference's std::optional::value() does
// (P/gptAttentionPlugin/gptAttentionPlugin.cpp:483-511).
class Fields
{
public:
    Fields(const tllm_plugin_field_t* f, int32_t n)
    {
        for (int i = 0; i < n; ++i)
        {
            if (!f[i].name)
                throw std::runtime_error("plugin field with null name");
            mFields[f[i].name] = f[i];
        }
    }

    template <typename T>
    T get(const char* name, int32_t type) const
    {
        auto it = mFields.find(name);
        if (it == mFields.end())
            throw std::runtime_error(std::string("missing plugin field '") + name + "'");
        if (it->second.type != type || !it->second.data || it->second.length < 1)
            throw std::runtime_error(std::string("plugin field '") + name + "' has wrong type/length");
        T v;
        std::memcpy(&v, it->second.data, sizeof(T));
        return v;
    }

    int32_t i32(const char* n) const { return get<int32_t>(n, TLLM_FIELD_INT32); }
    int8_t i8(const char* n) const { return get<int8_t>(n, TLLM_FIELD_INT8); }
    float f32(const char* n) const { return get<float>(n, TLLM_FIELD_FLOAT32); }

    std::vector<int32_t> i32s(const char* name) const
    {
        auto it = mFields.find(name);
        if (it == mFields.end() || it->second.type != TLLM_FIELD_INT32)
            throw std::runtime_error(std::string("missing plugin field '") + name + "'");
        const int32_t* d = static_cast<const int32_t*>(it->second.data);
        return std::vector<int32_t>(d, d + it->second.length);
    }

    void expect_only(std::initializer_list<const char*> known) const
    {
        for (auto& kv : mFields)
        {
            bool ok = false;
            for (auto k : known)
                ok = ok || kv.first == k;
            if (!ok)
                throw std::runtime_error("unknown plugin field '" + kv.first + "'");
        }
    }

private:
    std::map<std::string, tllm_plugin_field_t> mFields;
};

// POD (de)serialisation helpers: raw memcpy in declaration order, length asserted on read (P/common/plugin.h:90-101).
struct Writer
{
    std::vector<char> buf;
    template <typename T>
    void put(const T& v)
    {
        const char* p = reinterpret_cast<const char*>(&v);
        buf.insert(buf.end(), p, p + sizeof(T));
    }
};

struct Reader
{
    const char* p;
    size_t left;
    Reader(const void* d, size_t n)
        : p(static_cast<const char*>(d))
        , left(n)
    {
    }
    template <typename T>
    T get()
    {
        if (left < sizeof(T))
            throw std::runtime_error("plugin deserialisation: buffer too short");
        T v;
        std::memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        left -= sizeof(T);
        return v;
    }
    void done() const
    {
        if (left != 0)
            throw std::runtime_error("plugin deserialisation: trailing bytes");
    }
};

class Plugin
{
public:
    virtual ~Plugin() = default;
    virtual const char* type() const = 0;
    virtual int nbOutputs() const = 0;
    virtual int outputDims(int idx, const Dims* in, int nin, Dims* out) const = 0;
    virtual int outputDtype(int idx, const int32_t* inTypes, int nin) const = 0;
    virtual bool supportsFormat(int pos, const Desc* io, int nin, int nout) const = 0;
    virtual size_t workspaceSize(const Desc* in, int nin, const Desc* out, int nout) const { return 0; }
    virtual int enqueue(const Desc* inDesc, const Desc* outDesc, const void* const* in, void* const* out, void* ws,
        hipStream_t stream)
        = 0;
    virtual void serialize(Writer& w) const = 0;
    virtual Plugin* clone() const = 0;
};

inline int64_t volume(const Dims& d)
{
    int64_t v = 1;
    for (int i = 0; i < d.nbDims; ++i)
        v *= d.d[i];
    return v;
}

inline int64_t rows_of(const Dims& d) // product of all dims but the last
{
    int64_t v = 1;
    for (int i = 0; i + 1 < d.nbDims; ++i)
        v *= d.d[i];
    return v;
}

// RoPE cos/sin table shared by all attention plugin instances (device memory, grown on demand,
// never inside a stream capture: sessions reserve it in setup()).
const float* rope_table(int rotary_dim, int min_len, int* len_out);

// registry
using CreateFn = Plugin* (*) (const Fields&);
using DeserializeFn = Plugin* (*) (Reader&);
struct Creator
{
    const char* name;
    CreateFn create;
    DeserializeFn deserialize;
};
const std::vector<Creator>& registry();

} // namespace plugins
} // namespace tllm
