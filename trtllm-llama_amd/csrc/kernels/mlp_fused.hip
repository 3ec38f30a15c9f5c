// Decode step, batch 1, SmoothQuant static: the whole gated MLP of a decoder layer in ONE launch (r06)
//     x_out = x + proj( quant( silu(fc(q)) * gate(q) ) ),   q = quant(RMSNorm(x) * gamma)
// = the two launches gemv_kernel<W_INT8_SQ, PK_NORM, EK_SWIGLU> (gate|up, 90 MB) and gemv_ksplit_kernel<W_INT8_SQ> (down, 45 MB)
// restated value for value: the prologue's summation order, exact integer dots, the same epilogue expressions - every int8 operand
// and every output half is bit-identical (tests/test_gpu_mlp_fused.py).
// Reference: GatedMLP.forward (PY/layers/mlp.py:43-73) behind RmsNorm (PY/layers/normalization.py:33-54), the static quantisers
// (K/quantization.cu:31-59), the SmoothQuant GEMM epilogue (cutlass_extensions/.../epilogue_per_row_per_col_scale.h:279-347).
//
// Why: the kernel boundary between the two GEMVs is ~2.3 us of launch-to-launch + ~1 us of ramp with the HBM pipe idle, per layer.
// The seam is an ALL-TO-ALL (every down-projection row needs the whole intermediate row), which inside a launch costs MORE than
// the boundary (3 - 5 us, profiles/r06_gate_up_workgroups.txt) - unless nothing waits for it: here every workgroup requests ALL of
// its down-projection rows (16 rows x 11 KB = 176 KB per CU, into registers) the moment its last gate|up output is stored, so the
// weight stream never stops, and the intermediate row is only needed when those rows have landed ~7 us later.  What made that
// possible is the scalar memory path: "have all 256 workgroups published?" is polled with s_load ... glc, which does not queue
// behind the CU's in-flight vector loads (0.4 us per look under load, tools/scalar_poll_probe.cpp, profiles/r06_scalar_poll.txt);
// the intermediate row itself is then requested with vector loads BEHIND the weight rows - it arrives when they have.
//
// Structure: 256 workgroups x 8 waves, one per CU, all resident (the launcher checks the occupancy query x CUs).
//   1. t = 0: x, gamma, the wave's first three gate|up tiles (fc row n | gate row n, 2 x 4 KB), the epilogue constants.
//   2. RMSNorm + static quantiser -> LDS (gemv_impl.h PK_NORM, MB = 1, same summation order).
//   3. the workgroup's ceil(I / 256) row pairs, wave w takes pairs w, w + 8, ..: two tiles in flight per wave, exact v_dot4,
//      EPI_SWIGLU_QSTATIC epilogue, one int8 per pair into LDS.
//   4. the wave's two down-projection rows (2 x 11 chunks of 1 KB) requested; workgroup barrier; wave 0 writes the workgroup's
//      LINE of the exchange area - its 43 bytes, zero padding and the launch's tag in the last word - as ONE full 64-byte
//      write-through store: the data is the flag (guide G16 R2), and no line has a second writer.  (First form: bytes written one by
//      one + a flag byte per workgroup, 64 flags to a line: byte-granular write-through stores to a shared line are read-modify-write
//      at the memory side and serialise at ~0.45 us each - the last flag was seen 27 us after it was written, launch 95 us.)
//      tag = the line's previous tag + 1, read by the workgroup at its start: no host bookkeeping, graph-replay safe.
//   5. two hops, all polls through the scalar path (s_load_dword glc x 16, one per line): the leader of every 16 workgroups waits
//      for its members' tags and writes the group's line; everybody waits for the 16 group lines; then requests the 256 data
//      lines (16 KB, agent-scope loads, BEHIND the weight rows in the queue) -> compacted into LDS; dots against the rows in
//      registers; residual epilogue.
// Every wait is bounded (max_spins); a time-out raises bit 16 of the error word and the launch ends (the session falls back to
// the two-launch form and repeats the request, as for the fused attention launch).
#include "dev_utils.h"
#include "kernels.h"
#include "launch_util.h"

namespace tllm
{
namespace kernels
{
using namespace dev;

namespace
{

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) int8_t gs8;
typedef __attribute__((address_space(1))) uint32_t gu32;
constexpr int kLine = 64;       // bytes of a workgroup's line of the exchange area: <= 60 data bytes, the tag in the last word
constexpr int kGroup = 16;      // workgroups per group (one leader each); 16 groups

constexpr int kWaves = 16;       // 1024 threads, four waves per SIMD (8 waves of 256 VGPRs: the per-pair epilogue's latency shows, the gate|up rows run at 4.9 TB/s)
constexpr int kThreads = 64 * kWaves;
constexpr int kKC = 4;         // K = 4096 int8 = 4 x 1 KiB chunks per gate|up row
constexpr int kRowsPerWg = 16; // down-projection rows per workgroup (one per wave)
constexpr int kMaxPairs = 4;   // gate|up row pairs per wave at most (2 or 3 here)

__device__ __forceinline__ float silu_mul_fp16_m(float g, float u)
{
    // fp16 rounding points of the reference graph (gemv_impl.h silu_mul_fp16: PY/layers/mlp.py:68-73, PY/functional.py:521-532)
    const float g16 = h2f(f2h(g));
    const float u16 = h2f(f2h(u));
    const float a = h2f(f2h(g16 / (1.f + __expf(-g16))));
    return h2f(f2h(a * u16));
}

// Bounded wait until word 0 of all 16 lines at g carries `tag` (false: gave up).  Agent-scope VECTOR loads, 16 lanes, one line
// each: they return in this wave's own order (behind ITS rows, ~1 us), not behind the whole CU's.  (The scalar path - s_load_dword
// glc x 16 - answers in ~1.5 us in a probe, tools/scalar_poll_probe.cpp, but here made the launch 50 - 70 us: the leader saw its
// last member 16 - 22 us after its line was written; with vector polls 1 - 2 us.  profiles/r06_mlp_one_launch.txt)
__device__ __forceinline__ bool wait16(const void* g, uint32_t tag, int max_spins, int& spins)
{
    const int lane = threadIdx.x & 63;
    for (;;)
    {
        const uint32_t v = __hip_atomic_load((const gu32*) g + (lane & 15) * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(v == tag))
            return true;
        if (++spins > max_spins)
            return false;
        __builtin_amdgcn_s_sleep(2);
    }
}
__device__ __forceinline__ uint32_t sld_word(const void* g)
{
    uint32_t v;
    asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(g) : "memory");
    return v;
}

#define MLP_STAMP(slot)                                                                                                \
    do                                                                                                                 \
    {                                                                                                                  \
        if (p.timing && lane == 0 && wid == 0)                                                                         \
            p.timing[(size_t) blockIdx.x * 16 + (slot)] = wall_clock64();                                              \
    } while (0)

// NC2: 1 KiB chunks per down-projection row (I <= NC2 * 1024)
template <int NC2>
__global__ __launch_bounds__(kThreads) void mlp_fused_kernel(const FusedMlpParams p)
{
    __shared__ __attribute__((aligned(16))) char smem[4096 /* quantised operand */ + 256 /* red */ + NC2 * 1024 /* intermediate row */
        + kWaves * 2 * kMaxPairs * 4 /* the waves' per-channel scales */ + kLine /* the workgroup's line */];
    char* xs = smem;
    float* red = reinterpret_cast<float*>(smem + 4096);
    char* act = smem + 4096 + 256;
    float* give_up = red + 40;
    float* wsc = reinterpret_cast<float*>(smem + 4096 + 256 + NC2 * 1024); // [wave][2 kMaxPairs]
    char* obuf = smem + 4096 + 256 + NC2 * 1024 + kWaves * 2 * kMaxPairs * 4; // [kLine]: this workgroup's int8 outputs, zero padded

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int b = blockIdx.x, nb = gridDim.x;
    const int K = p.K, I = p.I;
    if (p.error[0] != 0u) // an earlier launch's bounded wait expired: everything behind it is invalid, do not spin again
        return;
    // launch constants: scalar loads, requested before anything else (behind the kernel's first store - or an asm statement that
    // clobbers memory - hipcc no longer uses the scalar path, and as vector loads in front of the tiles they would hold them back)
    const float pro_q = p.act_quant[0];
    const float rs_fc = p.row_fc[0];
    const float rs_gate = p.row_gate[0];
    const float epi_q = p.out_quant[0];
    const float rs_proj = p.row_proj[0];
    const float* sc_fc = reinterpret_cast<const float*>(p.scale_fc);
    const float* sc_gate = reinterpret_cast<const float*>(p.scale_gate);
    // the wave's two output rows of the down-projection: scales and residuals (wave-uniform: scalar loads; n0 is even)
    const int n0 = b * kRowsPerWg + wid;
    const float ps0 = reinterpret_cast<const float*>(p.scale_proj)[p.per_channel_proj ? n0 : 0];
    const uint32_t res2 = reinterpret_cast<const uint32_t*>(p.x)[n0 >> 1]; // (the pair of halves that holds x[n0])

    // ------------------------------------------------------------------ t = 0: x, gamma, the first tiles
    const int t2 = tid & 255;
    const uint16_t* xg = reinterpret_cast<const uint16_t*>(p.x);
    const uint4 xa = *reinterpret_cast<const uint4*>(xg + t2 * 8);
    const uint4 xb = *reinterpret_cast<const uint4*>(xg + (t2 + 256) * 8);
    const int t5 = tid & 511; // the prologue is written for 512 threads x 8 elements: waves 8 - 15 repeat the work of waves 0 - 7
    const uint4 gv = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(p.gamma) + t5 * 8);
    // row pairs of this workgroup: [b tpc, min((b + 1) tpc, I)); wave w: pairs w, w + 8, ...
    const int tpc = (I + nb - 1) / nb;
    const int pair0 = b * tpc;
    const int pair1 = pair0 + tpc < I ? pair0 + tpc : I;
    const int npair = pair1 > pair0 ? pair1 - pair0 : 0;
    const int ng = wid < npair ? (npair - wid + kWaves - 1) / kWaves : 0;
    const char* wfc = reinterpret_cast<const char*>(p.w_fc);
    const char* wgt = reinterpret_cast<const char*>(p.w_gate);
    auto pair_of = [&](int i) { return pair0 + wid + i * kWaves; };
    auto load_tile = [&](int i, uint4 (&wv)[kKC][2]) {
        int n = pair_of(i);
        n = n < I ? n : I - 1;
#pragma unroll
        for (int u = 0; u < kKC; ++u)
        {
            wv[u][0] = ld_nt16(wfc + (int64_t) n * p.ldw + u * 1024 + lane * 16);
            wv[u][1] = ld_nt16(wgt + (int64_t) n * p.ldw + u * 1024 + lane * 16);
        }
    };
    // A STATIC schedule of three tiles per wave in three register tiles (every wave has 2 or 3 row pairs): a wave with only 2 reads,
    // as its third tile, 16 bytes that every lane shares (one line, nobody consumes the result) - no load sits inside a conditional
    // block (hipcc ends such a block with a full s_waitcnt).
    // the per-channel scales of the wave's (at most kMaxPairs) row pairs: ONE vector load now (lane j: fc scale of pair j, lane
    // kMaxPairs + j: gate scale), parked in LDS - a load next to the epilogue of every pair would be a full wait on the weight stream
    float my_scale;
    {
        const int j = lane < kMaxPairs ? lane : (lane < 2 * kMaxPairs ? lane - kMaxPairs : 0);
        int n = pair_of(j);
        n = n < I ? n : I - 1;
        my_scale = (lane < kMaxPairs ? sc_fc : sc_gate)[p.per_channel ? n : 0];
    }
    uint4 tl0[kKC][2], tl1[kKC][2], tl2[kKC][2];
    auto load_tile_or_dummy = [&](int i, uint4 (&wv)[kKC][2]) {
        const bool real = i < ng; // wave-uniform
        int n = pair_of(i);
        n = n < I ? n : I - 1;
#pragma unroll
        for (int u = 0; u < kKC; ++u)
        {
            wv[u][0] = ld_nt16(real ? wfc + (int64_t) n * p.ldw + u * 1024 + lane * 16 : wfc);
            wv[u][1] = ld_nt16(real ? wgt + (int64_t) n * p.ldw + u * 1024 + lane * 16 : wgt);
        }
    };
    __builtin_amdgcn_sched_barrier(0);
    load_tile(0, tl0);
    // (tiles 1 and 2 go out behind the prologue: 192 KB per CU in flight at t = 0 held the prologue - x itself, the instruction
    //  fetch - until 13 us into the launch; a CU answers its loads in order at ~25 GB/s)
    __builtin_amdgcn_sched_barrier(0);
    MLP_STAMP(0);
    if (tid == 0)
        *give_up = 0.f;
    if (lane < 2 * kMaxPairs)
        wsc[wid * 2 * kMaxPairs + lane] = my_scale;
    if (tid < kLine / 4)
        reinterpret_cast<uint32_t*>(obuf)[tid] = 0u;
    // the tag of this launch: the last word of this workgroup's own line (nobody else writes it) + 1, never 0
    const char* xlines = reinterpret_cast<const char*>(p.flags);
    const char* glines = xlines + (size_t) nb * kLine;
    const uint32_t prev = sld_word(xlines + (size_t) b * kLine + (kLine - 4));
    const uint32_t tag = prev + 1u ? prev + 1u : 1u;

    // ------------------------------------------------------------------ RMSNorm + static quantiser -> LDS (gemv_impl.h PK_NORM, MB = 1)
    {
        float ss = 0.f;
        const uint32_t w8[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
        for (int q = 0; q < 8; ++q)
        {
            const h2_t hh = u32_as_h2(w8[q]);
            const float f0 = (float) hh.x, f1 = (float) hh.y;
            ss += f0 * f0 + f1 * f1;
        }
        ss = wave_sum(ss);
        if (lane == 0 && wid < 4)
            red[wid] = ss;
        __syncthreads();
        ss = red[0] + red[1] + red[2] + red[3];
        const float inv = 1.0f / sqrtf(ss / (float) K + p.eps);
        uint32_t xs4[4] = {t5 < 256 ? xa.x : xb.x, t5 < 256 ? xa.y : xb.y, t5 < 256 ? xa.z : xb.z, t5 < 256 ? xa.w : xb.w};
        const uint32_t gs4[4] = {gv.x, gv.y, gv.z, gv.w};
        uint32_t o[2] = {0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q)
        {
            h2_t hh = u32_as_h2(xs4[q]);
            const h2_t gg = u32_as_h2(gs4[q]);
            const float n0f = h2f(f2h((float) hh.x * inv)), n1f = h2f(f2h((float) hh.y * inv));
            hh.x = (_Float16) (n0f * (float) gg.x);
            hh.y = (_Float16) (n1f * (float) gg.y);
            const uint32_t b0 = (uint8_t) f2i8_rni_sat((float) hh.x * pro_q);
            const uint32_t b1 = (uint8_t) f2i8_rni_sat((float) hh.y * pro_q);
            o[q >> 1] |= (b0 | (b1 << 8)) << (16 * (q & 1));
        }
        if (tid < 512)
            *reinterpret_cast<uint2*>(xs + tid * 8) = make_uint2(o[0], o[1]);
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    load_tile(1, tl1);
    load_tile_or_dummy(2, tl2);
    __builtin_amdgcn_sched_barrier(0);
    MLP_STAMP(1);
    if (b == 0 && p.x_pro_out) // tap: the int8 operand exactly as the dots consume it
        for (int k = tid; k < K / 4; k += kThreads)
            reinterpret_cast<uint32_t*>(p.x_pro_out)[k] = reinterpret_cast<const uint32_t*>(xs)[k];

    // ------------------------------------------------------------------ the row pairs
    auto finish = [&](int i, int a0, int a1) {
        const int n = pair_of(i);
        const float s0 = wsc[wid * 2 * kMaxPairs + i], s1 = wsc[wid * 2 * kMaxPairs + kMaxPairs + i];
        a0 = wave_sum(a0);
        a1 = wave_sum(a1);
        // epilogue of gemv_impl.h (EPI_SWIGLU_QSTATIC): fp16 rounding points of the reference graph, then the static quantiser
        const float r0 = (float) a0 * (s0 * rs_fc);
        const float o16 = silu_mul_fp16_m(r0, (float) a1 * (s1 * rs_gate));
        if (lane == 0)
            obuf[n - pair0] = (char) f2i8_rni_sat(o16 * epi_q);
    };
    auto dots = [&](const uint4 (&cur)[kKC][2], int& a0, int& a1) {
        a0 = a1 = 0;
#pragma unroll
        for (int u = 0; u < kKC; ++u)
        {
            const uint4 xr = *reinterpret_cast<const uint4*>(xs + (u * 64 + lane) * 16);
            a0 = sdot4(cur[u][0].x, xr.x, a0);
            a0 = sdot4(cur[u][0].y, xr.y, a0);
            a0 = sdot4(cur[u][0].z, xr.z, a0);
            a0 = sdot4(cur[u][0].w, xr.w, a0);
            a1 = sdot4(cur[u][1].x, xr.x, a1);
            a1 = sdot4(cur[u][1].y, xr.y, a1);
            a1 = sdot4(cur[u][1].z, xr.z, a1);
            a1 = sdot4(cur[u][1].w, xr.w, a1);
        }
    };
    // the wave's down-projection row: lane l, chunk c: bytes [(c * 64 + l) * 16, + 16) of row n0 (beyond the row: a clamped, valid
    // address; the activations there are zero).  Requested as soon as tile 0's registers are free, so that the weight stream never
    // stops: the row is consumed ~7 us later.  (Behind tile 1 as well - all gate|up tiles ahead of all rows in the queue, the
    // hand-off under the rows' stream - needs 140 VGPRs: with the 128 that four waves per SIMD leave it spills, 30 us per launch.)
    const char* wpj = reinterpret_cast<const char*>(p.w_proj);
    const int nvec = (I + 15) / 16; // 16-byte vectors per row
    uint4 w2[NC2];
    {
        int a0, a1;
        dots(tl0, a0, a1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NC2; ++c)
        {
            const int v = c * 64 + lane < nvec ? c * 64 + lane : nvec - 1;
            w2[c] = ld_nt16(wpj + (int64_t) n0 * p.ldw_proj + (int64_t) v * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
        finish(0, a0, a1);
        dots(tl1, a0, a1);
        finish(1, a0, a1);
        if (ng > 2) // wave-uniform; no load inside
        {
            dots(tl2, a0, a1);
            finish(2, a0, a1);
        }
    }
    MLP_STAMP(2);

    __syncthreads(); // every wave's outputs are in obuf (LDS: no wait on the rows just requested)
    MLP_STAMP(7);
    if (wid == 0)
    {
        // the line: 15 words of data + the tag, ONE 64-byte write-through store
        if (lane < kLine / 4)
            __hip_atomic_store((gu32*) (p.flags + (size_t) b * kLine) + lane, lane == kLine / 4 - 1 ? tag : reinterpret_cast<const uint32_t*>(obuf)[lane],
                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p.inter && lane < npair) // the compact row (taps; the two-launch form's buffer)
            reinterpret_cast<char*>(p.inter)[pair0 + lane] = obuf[lane];
        MLP_STAMP(3);
        // ---------------------------------------------------------------- everybody published?  (scalar path: not behind the rows)
        int spins = 0;
        bool ok = true;
        if ((b & (kGroup - 1)) == 0) // the group's leader: its 16 members' tags (last word of their lines), then the group's line
        {
            ok = wait16(xlines + (size_t) b * kLine + (kLine - 4), tag, p.max_spins, spins);
            if (ok && lane < kLine / 4)
                __hip_atomic_store((gu32*) (p.flags + (size_t) nb * kLine + (size_t) (b / kGroup) * kLine) + lane, tag, __ATOMIC_RELAXED,
                    __HIP_MEMORY_SCOPE_AGENT);
        }
        MLP_STAMP(8);
        ok = ok && wait16(glines, tag, p.max_spins, spins);
        if (!ok && lane == 0)
            *give_up = 1.f;
        if (p.timing && lane == 0)
            p.timing[(size_t) blockIdx.x * 16 + 12] = (uint64_t) spins;
    }
    __syncthreads();
    MLP_STAMP(4);
    if (*give_up != 0.f) // uniform
    {
        if (tid == 0)
            atomicOr(p.error, 16u);
        return;
    }
    // the 256 lines -> the intermediate row in LDS: 8 bytes per load (agent scope: written through by the other XCDs in this
    // launch), behind the weight rows in the queue; line l's bytes [0, pairs of l) go to [l tpc, ..)
    {
        const gu64* src = (const gu64*) p.flags;
        constexpr int PER = 2048 / kThreads; // 256 lines x 8 units
        unsigned long long g[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k)
            g[k] = __hip_atomic_load(src + tid + kThreads * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < (NC2 * 1024 - I + 3) / 4 && I % 4 == 0) // zero beyond I
            reinterpret_cast<uint32_t*>(act + I)[tid] = 0u;
#pragma unroll
        for (int k = 0; k < PER; ++k)
        {
            const int u = tid + kThreads * k, line = u >> 3, off = (u & 7) * 8;
            const int first = line * tpc;
            const int cnt = first + tpc < I ? tpc : (I > first ? I - first : 0);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (off + j < cnt)
                    act[first + off + j] = (char) (g[k] >> (8 * j));
        }
    }
    __syncthreads();
    MLP_STAMP(5);
    // ------------------------------------------------------------------ dots, residual epilogue (gemv_ksplit.hip / gemv_impl.h EPI_RESIDUAL)
    int d0 = 0;
#pragma unroll
    for (int c = 0; c < NC2; ++c)
    {
        const uint4 xr = *reinterpret_cast<const uint4*>(act + (c * 64 + lane) * 16);
        d0 = sdot4(w2[c].x, xr.x, d0);
        d0 = sdot4(w2[c].y, xr.y, d0);
        d0 = sdot4(w2[c].z, xr.z, d0);
        d0 = sdot4(w2[c].w, xr.w, d0);
    }
    d0 = wave_sum(d0);
    if (lane == 0 && n0 < p.N)
    {
        const float r0 = (float) d0 * (ps0 * rs_proj);
        const float res = h2f((uint16_t) ((n0 & 1) ? (res2 >> 16) : (res2 & 0xffffu)));
        reinterpret_cast<uint16_t*>(p.x_out)[n0] = f2h(h2f(f2h(r0)) + res);
    }
    MLP_STAMP(6);
}
#undef MLP_STAMP

const void* mlp_kernel_of(int nc2)
{
    switch (nc2)
    {
    case 11: return reinterpret_cast<const void*>(mlp_fused_kernel<11>);
    default: return nullptr;
    }
}

} // namespace

size_t mlp_fused_flag_bytes()
{
    return (size_t) (256 + 256 / kGroup) * kLine; // a line per workgroup + a line per group
}

// K = 4096 (four 1 KiB chunks per gate|up row, 512 threads x 8 elements in the prologue), N = 16 rows per workgroup x 256
// workgroups, every wave at least two row pairs, a down-projection row of at most 11 KiB - LLaMA-7B; one workgroup per CU, all resident
bool mlp_fused_serves(int32_t K, int32_t I, int32_t N)
{
    const int nb = 256;
    if (K != kKC * 1024 || N != kRowsPerWg * nb || I % 16 != 0)
        return false;
    const int tpc = (I + nb - 1) / nb;
    // (every wave of every workgroup has 2 or 3 row pairs: the static tile schedule)
    if (tpc < 2 * kWaves || tpc > 3 * kWaves || tpc > kLine - 4 || I % 4 || (int64_t) (nb - 1) * tpc + 2 * kWaves > I) // (the last workgroup too)
        return false;
    const void* kfn = mlp_kernel_of((I + 1023) / 1024);
    if (!kfn)
        return false;
    const int cus = launch_util::device_cus();
    return cus >= nb && launch_util::blocks_per_cu(kfn, 64 * kWaves, 0) * cus >= nb;
}

int launch_mlp_fused(const FusedMlpParams& p, hipStream_t stream)
{
    if (!mlp_fused_serves(p.K, p.I, p.N))
    {
        set_error("fused MLP: shape not served or grid not resident (K %d, I %d, N %d)", p.K, p.I, p.N);
        return -1;
    }
    if (!p.x || !p.x_out || !p.gamma || !p.act_quant || !p.w_fc || !p.w_gate || !p.scale_fc || !p.scale_gate || !p.row_fc || !p.row_gate || !p.row_proj || !p.out_quant
        || !p.w_proj || !p.scale_proj || !p.flags || !p.error || p.ldw % 16 || p.ldw_proj % 16 || p.ldw < p.K || p.ldw_proj < p.I)
    {
        set_error("fused MLP: missing operand");
        return -1;
    }
    const void* kfn = mlp_kernel_of((p.I + 1023) / 1024);
    FusedMlpParams q = p;
    void* args[] = {&q};
    const hipError_t e = hipLaunchKernel(kfn, dim3(256), dim3(64 * kWaves), args, 0, stream);
    if (e != hipSuccess)
    {
        set_error("fused MLP launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}

} // namespace kernels
} // namespace tllm
