// ubench_malloc2.hip -- which hipMalloc calls are slow?  (clean vs. recently freed VRAM)
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <vector>

using Clock = std::chrono::steady_clock;
static double ms(Clock::time_point a) { return std::chrono::duration<double, std::milli>(Clock::now() - a).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  CK(hipSetDevice(0));
  CK(hipFree(0));
  size_t fr, tot;
  CK(hipMemGetInfo(&fr, &tot));
  printf("free %.1f GB of %.1f GB\n", fr / 1e9, tot / 1e9);
  std::vector<void*> ps;
  printf("A: 8 x 32 GB, no frees:");
  for (int i = 0; i < 8; i++) {
    void* p = nullptr;
    auto t = Clock::now();
    if (hipMalloc(&p, (size_t)32 << 30) != hipSuccess) { printf(" fail"); break; }
    printf(" %.1f", ms(t));
    ps.push_back(p);
  }
  printf(" ms\n");
  auto t = Clock::now();
  for (auto p : ps) CK(hipFree(p));
  printf("free all: %.1f ms\n", ms(t));
  ps.clear();
  printf("B: immediately again 8 x 32 GB:");
  for (int i = 0; i < 8; i++) {
    void* p = nullptr;
    auto t2 = Clock::now();
    if (hipMalloc(&p, (size_t)32 << 30) != hipSuccess) { printf(" fail"); break; }
    printf(" %.1f", ms(t2));
    ps.push_back(p);
  }
  printf(" ms\n");
  for (auto p : ps) CK(hipFree(p));
  ps.clear();
  sleep(20);
  printf("C: after 20 s idle, 8 x 32 GB:");
  for (int i = 0; i < 8; i++) {
    void* p = nullptr;
    auto t2 = Clock::now();
    if (hipMalloc(&p, (size_t)32 << 30) != hipSuccess) { printf(" fail"); break; }
    printf(" %.1f", ms(t2));
    ps.push_back(p);
  }
  printf(" ms\n");
  // D: keep 7 blocks, free one, malloc one
  t = Clock::now();
  CK(hipFree(ps.back()));
  ps.pop_back();
  void* p = nullptr;
  CK(hipMalloc(&p, (size_t)32 << 30));
  printf("D: free 32 GB + malloc 32 GB with 224 GB held: %.1f ms\n", ms(t));
  ps.push_back(p);
  for (auto q : ps) CK(hipFree(q));
  ps.clear();
  // E: small blocks after the big frees
  t = Clock::now();
  for (int i = 0; i < 64; i++) { CK(hipMalloc(&p, (size_t)256 << 20)); ps.push_back(p); }
  printf("E: 64 x 256 MB right after freeing 256 GB: %.1f ms\n", ms(t));
  for (auto q : ps) CK(hipFree(q));
  return 0;
}
