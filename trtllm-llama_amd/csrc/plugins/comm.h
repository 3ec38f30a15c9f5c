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
int group_info(const std::vector<int32_t>& group, int* nranks, int* my_index); // ncclCommCount / ncclCommUserRank

// One-shot peer-to-peer all-reduce for small fp16 vectors (kernels/p2p_allreduce.hip, plugins/p2p.cpp)
namespace p2p
{
int create(int world, int rank, size_t max_bytes, void* handle64);
int attach(const void* handles);
int enable(bool on); // -1: refused (not attached, or out of service after a time-out until create + attach)
// the verdict of the caller's validation of the fused layer seam (default on); off keeps the all-reduce peer-to-peer but the
// residual add / RMSNorm / quantiser in the consuming kernels
void enable_fused(bool on);
bool attached();
bool usable(int world, int64_t bytes);
bool usable_fused(int world, int64_t bytes);
bool usable_fused_flag();
// what a captured step graph depends on (enable / fused / create / time-out all bump it) and the count of time-outs
uint64_t generation();
uint64_t error_generation();
int64_t slot_capacity(int world); // bytes one exchange can carry when the path is enabled for this world size, else 0
bool enabled();
// after a time-out: takes the transport out of service and clears the error / poison words, so that the sticky flag does not
// fail later calls that no longer use this transport (destroy + create + attach brings it back)
void disable_after_error();
void set_max_spins(int n); // tests: how long a flag wait may spin before it gives up (0 = default)
int all_reduce_f16(void* buf, int64_t count, hipStream_t stream);
// the tensor-parallel layer seam in one launch (kernels/p2p_allreduce.hip): x <- fp16(x + sum_r partial_r), then
// norm_out <- RMSNorm(x) * gamma as fp16 (quant 0) or int8 (1: static scale, 2: per token -> dyn_scale_out[rows])
struct FusedTail
{
    void* x = nullptr;            // [rows, cols] fp16 residual stream, updated in place
    const void* gamma = nullptr;  // [cols] fp16
    float eps = 1e-6f;
    void* norm_out = nullptr;     // [rows, cols] fp16 | int8
    int quant = 0;
    const float* quant_scale = nullptr;
    float* dyn_scale_out = nullptr;
};
int all_reduce_residual_norm(void* partial, int rows, int cols, const FusedTail& t, hipStream_t stream);
int all_gather(const void* in, void* out, int64_t bytes_per_rank, hipStream_t stream);
int error_flag(uint32_t* out);
int destroy();
} // namespace p2p
} // namespace comm
} // namespace tllm
