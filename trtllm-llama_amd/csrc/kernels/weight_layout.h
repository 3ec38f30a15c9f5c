// Weight storage layouts of the MI355X GEMV/GEMM kernels (host + device view).
//
// The reference re-lays weight-only matrices for SM80 tensor cores (row permutation, sub-byte transpose,
// ColumnMajorTileInterleave, +128/+8 bias: K/cutlass_kernels/cutlass_preprocessors.cpp:158-535).  None of
// that maps to CDNA4; the layouts here are chosen so that one lane's 16-byte load is one contiguous k-run of
// one output column, which is what a wave64 streaming GEMV and the MFMA B-operand both want.
//
//   W_FP16      fp16  [N][K]                       (the Gemm plugin's own [out, in] operand, transb=1)
//   W_INT8_SQ   s8    [N][ldw]  ldw = roundup(K,16)  pad = 0
//   W_INT8_WOQ  u8    [N][ldw]  u = q + 128          pad = 128   (bias kept from the reference: unsigned
//                                                                bytes splice into fp16 1024+u with one v_perm)
//   W_INT4_WOQ  u4x2  [N][ldw]  ldw = roundup(K,32)/2, nibble n = q + 8, pad nibble = 8.  Within every
//               32-bit word (8 consecutive k: e0..e7) the nibble order is  e0 e2 e4 e6 | e1 e3 e5 e7
//               (nibble i of the word holds e[kNibbleToElem[i]]), so that
//                   (w      & 0x000f000f) -> (e0,e1)     (w      & 0x00f000f0) -> (e2,e3)*16
//                   (w >> 8 & 0x000f000f) -> (e4,e5)     (w >> 8 & 0x00f000f0) -> (e6,e7)*16
//               come out as fp16 pairs in natural k order with two v_and_or per pair.
#pragma once
#include <stdint.h>

namespace tllm
{
namespace layout
{

constexpr int kNibbleToElem[8] = {0, 2, 4, 6, 1, 3, 5, 7};
constexpr int kElemToNibble[8] = {0, 4, 1, 5, 2, 6, 3, 7};

inline int64_t round_up(int64_t v, int64_t m)
{
    return (v + m - 1) / m * m;
}

// bytes per weight row for a given weight type (kernels::WType numbering) and K
inline int64_t row_bytes(int wtype, int64_t K)
{
    switch (wtype)
    {
    case 0: return K * 2;
    case 1:
    case 3: return round_up(K, 16);
    case 2: return round_up(K, 32) / 2;
    default: return 0;
    }
}

} // namespace layout
} // namespace tllm
