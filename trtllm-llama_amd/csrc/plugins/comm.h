// Tensor-parallel communicator registry over RCCL/xGMI (replaces the NCCL+MPI bootstrap of
// P/ncclPlugin/allreducePlugin.cpp:124-162).  librccl is dlopen'ed on first use so that the plugin library
// loads (and its CPU-side entry points work) on a host without a GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

namespace tllm
{
namespace comm
{
int get_unique_id(void* id128);
int init_rank(const std::vector<int32_t>& group, int32_t rank, const void* id128);
int all_reduce_sum(const std::vector<int32_t>& group, const void* in, void* out, int64_t count, int32_t dtype,
    hipStream_t stream);
int all_gather(const std::vector<int32_t>& group, const void* in, void* out, int64_t count, int32_t dtype,
    hipStream_t stream);
int destroy_all();
bool has_comm(const std::vector<int32_t>& group);
} // namespace comm
} // namespace tllm
