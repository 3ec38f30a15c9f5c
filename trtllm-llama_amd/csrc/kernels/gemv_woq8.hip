// Decode GEMV kernels for W_INT8_WOQ weights (see gemv_impl.h).
#include "gemv_impl.h"

namespace tllm
{
namespace kernels
{
int launch_gemv_woq8(const GemvArgs& a, int pk, bool swiglu, hipStream_t stream)
{
    return launch_wt<W_INT8_WOQ>(a, pk, swiglu, stream);
}
} // namespace kernels
} // namespace tllm
