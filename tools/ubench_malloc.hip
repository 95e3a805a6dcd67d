// ubench_malloc.hip -- cost of getting device memory on MI355X (cold one-shot runs are dominated by it):
// hipMalloc / hipFree / re-malloc, hipMallocAsync from a pool, hipMemCreate+hipMemMap (VMM), first-touch memset.
// hipcc --offload-arch=gfx950 -O2 tools/ubench_malloc.hip -o tools/ubench_malloc
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

using Clock = std::chrono::steady_clock;
static double ms(Clock::time_point a) { return std::chrono::duration<double, std::milli>(Clock::now() - a).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  CK(hipSetDevice(0));
  CK(hipFree(0));
  for (size_t gb : {1, 4, 16, 64}) {
    size_t bytes = gb << 30;
    void* p = nullptr;
    auto t = Clock::now();
    CK(hipMalloc(&p, bytes));
    double t_alloc = ms(t);
    t = Clock::now();
    CK(hipMemset(p, 0, bytes));
    CK(hipDeviceSynchronize());
    double t_set1 = ms(t);
    t = Clock::now();
    CK(hipMemset(p, 0, bytes));
    CK(hipDeviceSynchronize());
    double t_set2 = ms(t);
    t = Clock::now();
    CK(hipFree(p));
    double t_free = ms(t);
    t = Clock::now();
    CK(hipMalloc(&p, bytes));
    double t_alloc2 = ms(t);
    CK(hipFree(p));
    printf("hipMalloc %3zu GB: %8.1f ms (%.1f ms/GB)  memset#1 %7.1f  memset#2 %7.1f  hipFree %7.1f  re-malloc %8.1f\n", gb, t_alloc,
           t_alloc / gb, t_set1, t_set2, t_free, t_alloc2);
  }
  {  // many 1 GB blocks vs one big one
    std::vector<void*> ps(32);
    auto t = Clock::now();
    for (auto& p : ps) CK(hipMalloc(&p, (size_t)1 << 30));
    printf("32 x 1 GB hipMalloc: %.1f ms\n", ms(t));
    for (auto& p : ps) CK(hipFree(p));
  }
  {  // stream-ordered pool
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipMemPool_t pool;
    CK(hipDeviceGetDefaultMemPool(&pool, 0));
    uint64_t thr = ~0ull;
    CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
    for (size_t gb : {4, 32}) {
      void* p = nullptr;
      auto t = Clock::now();
      CK(hipMallocAsync(&p, gb << 30, s));
      CK(hipStreamSynchronize(s));
      double a = ms(t);
      t = Clock::now();
      CK(hipFreeAsync(p, s));
      CK(hipStreamSynchronize(s));
      double f = ms(t);
      t = Clock::now();
      CK(hipMallocAsync(&p, gb << 30, s));
      CK(hipStreamSynchronize(s));
      double a2 = ms(t);
      CK(hipFreeAsync(p, s));
      CK(hipStreamSynchronize(s));
      printf("hipMallocAsync %2zu GB: %8.1f ms, free %.1f ms, again %.1f ms\n", gb, a, f, a2);
    }
  }
  {  // VMM
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    size_t bytes = (size_t)32 << 30;
    auto t = Clock::now();
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, bytes, &prop, 0));
    double c = ms(t);
    t = Clock::now();
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, bytes, gran, nullptr, 0));
    CK(hipMemMap(va, bytes, 0, h, 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, bytes, &acc, 1));
    double m = ms(t);
    printf("VMM 32 GB: granularity %zu KB, hipMemCreate %.1f ms, reserve+map+access %.1f ms\n", gran >> 10, c, m);
    CK(hipMemUnmap(va, bytes));
    CK(hipMemRelease(h));
    CK(hipMemAddressFree(va, bytes));
  }
  {  // two threads allocating at once
    void *a = nullptr, *b = nullptr;
    auto t = Clock::now();
    std::thread t1([&] { (void)hipSetDevice(0); (void)hipMalloc(&a, (size_t)16 << 30); });
    std::thread t2([&] { (void)hipSetDevice(0); (void)hipMalloc(&b, (size_t)16 << 30); });
    t1.join(); t2.join();
    printf("2 threads x 16 GB hipMalloc concurrently: %.1f ms\n", ms(t));
    (void)hipFree(a); (void)hipFree(b);
  }
  return 0;
}
