// Decode GEMV kernels for W_FP16 weights (see gemv_impl.h).
#include "gemv_impl.h"

namespace tllm
{
namespace kernels
{
int launch_gemv_fp16(const GemvArgs& a, int pk, bool swiglu, hipStream_t stream)
{
    return launch_wt<W_FP16>(a, pk, swiglu, stream);
}
} // namespace kernels
} // namespace tllm
