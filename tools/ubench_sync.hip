// ubench_sync.hip -- latency of "launch a tiny kernel, get 3 words back on the host":
// (a) hipStreamSynchronize, (b) host spin on a pinned sequence word the kernel writes last.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench_sync.hip -o /tmp/ubench_sync
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>

__global__ void k_plain(uint32_t* out, uint32_t v) { if (threadIdx.x == 0) { out[0] = v; out[1] = v + 1; } }
__global__ void k_flag(volatile uint32_t* out, uint32_t seq) {
  if (threadIdx.x == 0) { out[0] = seq * 3; out[1] = seq * 5; __threadfence_system(); out[16] = seq; }
}
__global__ void k_two(uint32_t* tmp, uint32_t v) { if (threadIdx.x == 0) tmp[0] = v; }

int main() {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  uint32_t* h; hipHostMalloc((void**)&h, 4096, hipHostMallocDefault);
  uint32_t* d; hipMalloc((void**)&d, 4096);
  for (int i = 0; i < 32; i++) h[i] = 0;
  const int n = 2000;
  using C = std::chrono::steady_clock;
  for (int rep = 0; rep < 2; rep++) {
    auto t0 = C::now();
    for (int i = 0; i < n; i++) { hipLaunchKernelGGL(k_plain, dim3(1), dim3(64), 0, s, h, (uint32_t)i); hipStreamSynchronize(s); }
    double a = std::chrono::duration<double>(C::now() - t0).count() / n * 1e6;
    t0 = C::now();
    for (int i = 0; i < n; i++) {
      hipLaunchKernelGGL(k_two, dim3(1), dim3(64), 0, s, d, (uint32_t)i);
      hipLaunchKernelGGL(k_plain, dim3(1), dim3(64), 0, s, h, (uint32_t)i);
      hipStreamSynchronize(s);
    }
    double a2 = std::chrono::duration<double>(C::now() - t0).count() / n * 1e6;
    t0 = C::now();
    volatile uint32_t* hv = h;
    for (int i = 1; i <= n; i++) {
      hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s, (volatile uint32_t*)h, (uint32_t)i);
      while (hv[16] != (uint32_t)i) { __builtin_ia32_pause(); }
    }
    double b = std::chrono::duration<double>(C::now() - t0).count() / n * 1e6;
    hipStreamSynchronize(s);
    t0 = C::now();
    for (int i = 1; i <= n; i++) {
      hipLaunchKernelGGL(k_two, dim3(1), dim3(64), 0, s, d, (uint32_t)i);
      hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s, (volatile uint32_t*)h, (uint32_t)(i + n));
      while (hv[16] != (uint32_t)(i + n)) { __builtin_ia32_pause(); }
    }
    double b2 = std::chrono::duration<double>(C::now() - t0).count() / n * 1e6;
    hipStreamSynchronize(s);
    printf("rep %d: launch+sync %.2f us | 2 launches+sync %.2f us | launch+spin %.2f us | 2 launches+spin %.2f us\n", rep, a, a2, b, b2);
  }
  return 0;
}
