// "The engine is what was defined" (T/tensorrt_llm/builder.py:259-267): an engine file carries the traced network
// (Builder.build_engine writes it as `network_json`), and the C++ host loop executes a fixed LLaMA schedule chosen from
// the configuration.  verify_network() proves the two are the same computation before the session accepts the engine -
// plugin order, data flow, which weight feeds which plugin, every plugin field, and the I/O tensor names of
// PY/runtime/generation.py:188-208 - and names the first node that differs otherwise.
#pragma once
#include <string>
#include <vector>

namespace tllm
{
namespace runtime
{

// What the session will execute, as far as the traced graph can tell it apart.
struct ScheduleDesc
{
    int num_layers = 0, heads_per_rank = 0, head_size = 0, tp = 1;
    float eps = 1e-6f;
    bool sq = false, per_token = false, woq = false, int4 = false, int8_kv = false, paged = false, packed = false;
    bool neox = true; // GPTAttention neox_rotary_style (the session's config key of the same name)
    // SmoothQuant: has_per_channel_scaling of each GEMM as the loaded scale tensors imply it (order: qkv, dense, fc, gate, proj)
    std::vector<int> per_channel;
};

// 0 when `network_json` (the text after "network_json=") describes exactly the schedule of `d`; otherwise 1 and `err` says
// which node / field / tensor differs.
int verify_network(const std::string& network_json, const ScheduleDesc& d, std::string& err);

} // namespace runtime
} // namespace tllm
