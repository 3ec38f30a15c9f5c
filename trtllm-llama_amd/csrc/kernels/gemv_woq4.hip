// Decode GEMV kernels for W_INT4_WOQ weights (see gemv_impl.h).
#include "gemv_impl.h"

namespace tllm
{
namespace kernels
{
int launch_gemv_woq4(const GemvArgs& a, int pk, bool swiglu, hipStream_t stream)
{
    return launch_wt<W_INT4_WOQ>(a, pk, swiglu, stream);
}
} // namespace kernels
} // namespace tllm
