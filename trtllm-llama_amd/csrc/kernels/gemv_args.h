// Shared between the GEMV dispatcher (gemv.hip) and the per-weight-type kernel translation units (gemv_impl.h).
#pragma once
#include "kernels.h"
#include <hip/hip_runtime.h>

namespace tllm
{
namespace kernels
{
namespace gemv_detail
{
constexpr int R = 2, U = 4; // weight tile: 2 rows x 4 chunks of 1 KiB (int4 rows of <= 2 KiB use 2 chunks: kernel template)
constexpr int kRedBytes = 384; // reduction scratch: 32 floats sum of squares, 32 amax, 32 activation sums (weight-only int8)
constexpr int kNXVMax = 6; // 16-byte x vectors a thread keeps in registers: K <= 256 * 8 * 6 = 12288 halfs
constexpr int kNXVSmall = 2; // bucket for K <= 4096 halfs (every 7B hidden-size GEMV): 32 fewer VGPRs -> one more wave/SIMD
// third bucket (r04): K <= 256 * 8 * 12 = 24576 halfs, for the prologues that do not normalise (plain input, or the SmoothQuant
// quantiser) - the down-projection of LLaMA-13B / 30B / 65B (K = 13824 / 17920 / 22016) with fp16 activations, which the 12288 of
// the middle bucket refused (found by the 13B-dimension parity test)
constexpr int kNXVLarge = 12;

enum ProKind
{
    PK_COPY = 0,  // x already in the operand type
    PK_NORM = 1,  // RMSNorm (+ quant for SQ)
    PK_QUANT = 2  // fp16 -> s8 (SQ)
};
enum EpiKind
{
    EK_PLAIN = 0, // none | residual
    EK_SWIGLU = 1 // swiglu (+ static quant)
};

struct GemvArgs
{
    GemvParams p;
    int32_t Kp;      // K rounded up to the weight vector width
    int32_t nchunks; // ceil(Kp / (64 * VEC))
    int32_t ngroups; // row groups (one per wave-step)
};

} // namespace gemv_detail
using namespace gemv_detail;

extern int gemv_tune_blocks_per_cu;

int launch_gemv_fp16(const GemvArgs& a, int pk, bool swiglu, hipStream_t stream);
int launch_gemv_woq8(const GemvArgs& a, int pk, bool swiglu, hipStream_t stream);
int launch_gemv_woq4(const GemvArgs& a, int pk, bool swiglu, hipStream_t stream);
int launch_gemv_sq(const GemvArgs& a, int pk, bool swiglu, hipStream_t stream);
// SmoothQuant, several rows (2 <= M <= 8), static activation scales: the matrix-pipe kernel (gemv_mfma_sq.hip); 1 = not served
extern int gemv_mfma_min_rows; // rows from which launch_gemv tries it (0 = never; -1 = the default 5)
int launch_gemv_mfma_sq(const GemvParams& p, hipStream_t stream);
// "few rows, long K" single-token projections, every weight type (gemv_ksplit.hip)
bool gemv_ksplit_applies(const GemvArgs& a);
int launch_gemv_ksplit(const GemvArgs& a, hipStream_t stream);

} // namespace kernels
} // namespace tllm
