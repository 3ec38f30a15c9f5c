// Probe (r06): does a poll through the SCALAR memory path (s_load_dwordx2 glc) come back ahead of the vector loads a CU already has
// in flight?  The fused decode launch (kernels/qkv_attn_fused.hip) looks at its siblings' granules only behind the LAST weight tile
// because a vector poll returns in order behind everything queued on the CU (~25 GB/s per CU).  If the scalar path bypasses that
// queue, the q hand-off and the attention arithmetic can move under the v rows' wait.
//   256 workgroups x 8 waves; every wave requests NL x 16 B per lane (NL = 24: 24 KB per wave, 192 KB per CU) at t = 0;
//   wave 1 lane 0 publishes the workgroup's granule {tag, value} write-through `dp` ticks (10 ns) after the start;
//   wave 0 starts polling the granule of workgroup (b + hop) % 256 at `dq` ticks: mode 0 = vector agent-scope load, 1 = s_load glc.
// Output per mode: when the stream ended, when the tag was seen (relative to the publisher's store), polls needed, time per poll.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(1))) unsigned long long gu64;

__device__ __forceinline__ unsigned long long sld(const void* p)
{
    unsigned long long v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

// 16 lines (stride 64 B) in one batch through the scalar path: how long does a sweep of N distinct lines take?
__device__ __forceinline__ unsigned sld16(const void* g)
{
    unsigned v[16];
    asm volatile("s_load_dword %0, %16, 0x0 glc\n\ts_load_dword %1, %16, 0x40 glc\n\ts_load_dword %2, %16, 0x80 glc\n\t"
                 "s_load_dword %3, %16, 0xc0 glc\n\ts_load_dword %4, %16, 0x100 glc\n\ts_load_dword %5, %16, 0x140 glc\n\t"
                 "s_load_dword %6, %16, 0x180 glc\n\ts_load_dword %7, %16, 0x1c0 glc\n\ts_load_dword %8, %16, 0x200 glc\n\t"
                 "s_load_dword %9, %16, 0x240 glc\n\ts_load_dword %10, %16, 0x280 glc\n\ts_load_dword %11, %16, 0x2c0 glc\n\t"
                 "s_load_dword %12, %16, 0x300 glc\n\ts_load_dword %13, %16, 0x340 glc\n\ts_load_dword %14, %16, 0x380 glc\n\t"
                 "s_load_dword %15, %16, 0x3c0 glc\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(v[0]), "=&s"(v[1]), "=&s"(v[2]), "=&s"(v[3]), "=&s"(v[4]), "=&s"(v[5]), "=&s"(v[6]), "=&s"(v[7]), "=&s"(v[8]),
                 "=&s"(v[9]), "=&s"(v[10]), "=&s"(v[11]), "=&s"(v[12]), "=&s"(v[13]), "=&s"(v[14]), "=&s"(v[15])
                 : "s"(g)
                 : "memory");
    unsigned r = 0;
    for (int i = 0; i < 16; ++i)
        r |= v[i];
    return r;
}
__device__ __forceinline__ unsigned sld4x16(const void* g)
{
    typedef unsigned u16v __attribute__((ext_vector_type(16)));
    u16v a, b, c, d;
    asm volatile("s_load_dwordx16 %0, %4, 0x0 glc\n\ts_load_dwordx16 %1, %4, 0x40 glc\n\ts_load_dwordx16 %2, %4, 0x80 glc\n\t"
                 "s_load_dwordx16 %3, %4, 0xc0 glc\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d)
                 : "s"(g)
                 : "memory");
    return a[0] | b[0] | c[0] | d[0];
}

template <int NL>
__global__ __launch_bounds__(512) void probe(const uint4* w, unsigned long long* flags, unsigned long long* out, uint32_t tag, int mode,
    int dp, int dq, int hop, uint4* sink)
{
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int b = blockIdx.x;
    const unsigned long long t0 = wall_clock64();
    uint4 r[NL];
    const uint4* src = w + ((size_t) (b * 8 + wid) * NL) * 64 + lane;
#pragma unroll
    for (int i = 0; i < NL; ++i)
    {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + i * 64));
        r[i] = make_uint4(v.x, v.y, v.z, v.w);
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long* o = out + (size_t) b * 16;
    if (wid == 1)
    {
        while ((int) (wall_clock64() - t0) < dp)
            __builtin_amdgcn_s_sleep(2);
        if (lane == 0)
        {
            __hip_atomic_store((gu64*) flags + b * 16, ((unsigned long long) tag << 32) | (uint32_t) b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            o[1] = wall_clock64(); // publish time (global constant clock)
        }
    }
    if (wid == 0 && mode >= 0)
    {
        while ((int) (wall_clock64() - t0) < dq)
            __builtin_amdgcn_s_sleep(2);
        const unsigned long long* f = flags + ((b + hop) & 255) * 16;
        const unsigned long long tq = wall_clock64();
        int polls = 0;
        unsigned long long g = 0, first_ret = 0;
        for (;;)
        {
            ++polls;
            if (mode == 0)
                g = __hip_atomic_load((const gu64*) f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (mode == 2) // a sweep of 16 lines (the partner's line first; the others only cost time)
                g = sld(f) | (sld16(flags + ((b * 8) & 127) * 16) & 0);
            else if (mode == 3) // a sweep of 4 lines with 64-byte loads
                g = sld(f) | (sld4x16(flags + ((b * 8) & 127) * 16) & 0);
            else
                g = sld(f);
            if (polls == 1)
            {
                if (mode == 0)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (waits for the stream too: that is the point - in-order return)
                first_ret = wall_clock64();
            }
            if ((uint32_t) (g >> 32) == tag || polls > 100000)
                break;
            __builtin_amdgcn_s_sleep(1);
        }
        const unsigned long long ts = wall_clock64();
        if (lane == 0)
        {
            o[2] = tq;
            o[3] = first_ret;
            o[4] = ts;
            o[5] = (unsigned long long) polls;
            o[6] = g;
        }
    }
    uint4 acc = r[0];
#pragma unroll
    for (int i = 1; i < NL; ++i)
    {
        acc.x ^= r[i].x;
        acc.y ^= r[i].y;
        acc.z ^= r[i].z;
        acc.w ^= r[i].w;
    }
    const unsigned long long te = wall_clock64();
    if (acc.x == 0x12345678u && acc.y == 77u)
        sink[tid] = acc;
    if (lane == 0)
    {
        o[8 + wid] = te;
        if (wid == 2)
            o[0] = t0;
    }
}

static double med(std::vector<double> v)
{
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main(int argc, char** argv)
{
    const int dp = argc > 1 ? atoi(argv[1]) : 100, dq = argc > 2 ? atoi(argv[2]) : 250, hop = argc > 3 ? atoi(argv[3]) : 37;
    constexpr int NL = 24;
    const size_t wbytes = (size_t) 256 * 8 * NL * 64 * 16;
    uint4 *w, *sink;
    unsigned long long *flags, *out;
    hipMalloc(&w, wbytes * 8); // eight different regions: every launch streams cold bytes
    hipMalloc(&sink, 512 * 16);
    hipMalloc(&flags, 256 * 16 * 8 + 4096);
    hipMalloc(&out, 256 * 16 * 8);
    hipMemset(w, 1, wbytes * 8);
    hipMemset(flags, 0, 256 * 16 * 8);
    std::vector<unsigned long long> h(256 * 16);
    uint32_t tag = 1;
    printf("publish at %.2f us, first poll at %.2f us after the workgroup's start, partner %d workgroups away; %d KB per CU in flight\n", dp / 100., dq / 100.,
        hop, NL * 8);
    for (int mode : {-1, 0, 1, 2, 3, 2, 3})
    {
        std::vector<double> v_end, v_first, v_seen, v_polls, v_lat;
        for (int rep = 0; rep < 8; ++rep, ++tag)
        {
            hipMemset(out, 0, 256 * 16 * 8);
            hipLaunchKernelGGL(probe<NL>, dim3(256), dim3(512), 0, 0, w + (size_t) (rep & 7) * (wbytes / 16), flags, out, tag, mode, dp, dq, hop, sink);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), out, 256 * 16 * 8, hipMemcpyDeviceToHost);
            if (rep < 2)
                continue;
            for (int b = 0; b < 256; ++b)
            {
                const unsigned long long* o = &h[b * 16];
                unsigned long long te = 0;
                for (int k = 0; k < 8; ++k)
                    te = std::max(te, o[8 + k]);
                v_end.push_back((double) (te - o[0]) / 100.);
                if (mode >= 0)
                {
                    const unsigned long long pub = h[((b + hop) & 255) * 16 + 1];
                    v_first.push_back((double) (o[3] - o[2]) / 100.);
                    v_seen.push_back(((double) o[4] - (double) std::max(pub, o[2])) / 100.);
                    v_polls.push_back((double) o[5]);
                    v_lat.push_back((double) (o[4] - o[2]) / 100. / (double) o[5]);
                }
            }
        }
        printf("mode %2d (%s): stream ends %.2f us (median over workgroups)", mode, mode < 0 ? "no poll" : mode == 0 ? "vector sc1" : mode == 1 ? "scalar glc" : mode == 2 ? "scalar glc + 16 more lines" : "scalar glc + 4 x 64 B", med(v_end));
        if (mode >= 0)
            printf(" | first poll returns after %.2f us | tag seen %.2f us after max(publish, first poll issue) | polls %.0f | %.2f us per poll",
                med(v_first), med(v_seen), med(v_polls), med(v_lat));
        printf("\n");
    }
    return 0;
}
