// Host-side launch state keyed by DEVICE (a process may hold sessions on devices with different CU counts, and a
// function attribute raised on one device says nothing about another): CU count, "dynamic-LDS attribute raised for this kernel on
// this device", occupancy of an instance.  Header-only; the statics of an inline function are one object per shared library.
#pragma once
#include <hip/hip_runtime.h>
#include <map>
#include <mutex>
#include <set>
#include <tuple>
#include <utility>

namespace tllm
{
namespace kernels
{
namespace launch_util
{
inline std::mutex& mu()
{
    static std::mutex m;
    return m;
}
inline int current_device()
{
    int dev = 0;
    (void) hipGetDevice(&dev);
    return dev;
}
// CUs of the current device (256 when the query fails)
inline int device_cus()
{
    static std::map<int, int> cache;
    const int dev = current_device();
    std::lock_guard<std::mutex> lock(mu());
    auto it = cache.find(dev);
    if (it == cache.end())
    {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        it = cache.emplace(dev, n).first;
    }
    return it->second;
}
// raises hipFuncAttributeMaxDynamicSharedMemorySize once per (device, kernel) when the launch needs more than the default 64 KB
inline void ensure_dynamic_lds(const void* kfn, size_t bytes)
{
    if (bytes <= 64 * 1024)
        return;
    static std::set<std::pair<int, const void*>> done;
    const int dev = current_device();
    std::lock_guard<std::mutex> lock(mu());
    if (done.insert(std::make_pair(dev, kfn)).second)
        (void) hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes);
}
// workgroups of this instance one CU of the current device admits (registers + LDS; 0 when the query fails)
inline int blocks_per_cu(const void* kfn, int threads, size_t dyn_lds)
{
    static std::map<std::tuple<int, const void*, int, size_t>, int> cache;
    const int dev = current_device();
    std::lock_guard<std::mutex> lock(mu());
    const auto key = std::make_tuple(dev, kfn, threads, dyn_lds);
    auto it = cache.find(key);
    if (it == cache.end())
    {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kfn, threads, dyn_lds) != hipSuccess)
            nb = 0;
        it = cache.emplace(key, nb).first;
    }
    return it->second;
}
} // namespace launch_util
} // namespace kernels
} // namespace tllm
