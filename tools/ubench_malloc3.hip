// ubench_malloc3.hip -- hipMalloc time for a given sequence of sizes (GB) in a fresh process: ./ubench_malloc3 25 77 ...
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

using Clock = std::chrono::steady_clock;
int main(int argc, char** argv) {
  (void)hipSetDevice(0);
  (void)hipFree(0);
  for (int i = 1; i < argc; i++) {
    double gb = atof(argv[i]);
    void* p = nullptr;
    auto t = Clock::now();
    hipError_t e = hipMalloc(&p, (size_t)(gb * (double)(1ull << 30)));
    double ms = std::chrono::duration<double, std::milli>(Clock::now() - t).count();
    printf("hipMalloc %6.1f GB: %8.1f ms  (%s)\n", gb, ms, hipGetErrorString(e));
  }
  return 0;
}
