// Decode GEMV kernels for W_INT8_SQ weights (see gemv_impl.h).
#include "gemv_impl.h"

namespace tllm
{
namespace kernels
{
int launch_gemv_sq(const GemvArgs& a, int pk, bool swiglu, hipStream_t stream)
{
    return launch_wt<W_INT8_SQ>(a, pk, swiglu, stream);
}
} // namespace kernels
} // namespace tllm
