// ubench_cumask -- which compute units does bit i of a hipExtStreamCreateWithCUMask mask enable on this part?
// For a few masks: launch 4096 short workgroups on the masked stream, each records HW_REG_XCC_ID and HW_REG_HW_ID,
// and print per mask the XCDs and (SE, SH, CU) ids that were seen.  Build: hipcc --offload-arch=gfx950 -O2 -o ubench_cumask ubench_cumask.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <set>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void where_kernel(uint32_t* out, int spin) {
  uint32_t xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));   // HW_REG_XCC_ID, bits 3:0
  uint32_t hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((32 - 1) << 11));    // HW_REG_HW_ID
  // keep the workgroup alive for a while so that the grid spreads over every enabled CU
  uint32_t x = threadIdx.x;
  for (int i = 0; i < spin; i++) x = x * 1664525u + 1013904223u;
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc & 15u; out[2 * blockIdx.x + 1] = hw | (x & 0u); }
}

static int run(const char* name, const std::vector<uint32_t>& mask) {
  hipStream_t s;
  CHECK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
  const int nb = 8192;
  uint32_t* d;
  CHECK(hipMalloc(&d, nb * 2 * sizeof(uint32_t)));
  hipLaunchKernelGGL(where_kernel, dim3(nb), dim3(64), 0, s, d, 20000);
  CHECK(hipStreamSynchronize(s));
  std::vector<uint32_t> h(nb * 2);
  CHECK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
  std::map<uint32_t, std::set<uint32_t>> per_xcc;
  std::map<uint32_t, int> blocks_mod8_to_xcc[8];
  for (int b = 0; b < nb; b++) {
    const uint32_t xcc = h[2 * b], hw = h[2 * b + 1];
    const uint32_t cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
    per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
    blocks_mod8_to_xcc[b & 7][xcc]++;
  }
  int total = 0;
  printf("%s\n", name);
  for (auto& kv : per_xcc) {
    printf("  xcc %u: %zu CUs:", kv.first, kv.second.size());
    for (uint32_t v : kv.second) printf(" %u.%u.%u", v >> 8, (v >> 4) & 15u, v & 15u);
    printf("\n");
    total += (int)kv.second.size();
  }
  printf("  total CUs seen: %d; xcc of blockIdx%%8:", total);
  for (int k = 0; k < 8; k++) {
    printf(" %d->{", k);
    for (auto& kv : blocks_mod8_to_xcc[k]) printf("%u:%d ", kv.first, kv.second);
    printf("}");
  }
  printf("\n");
  CHECK(hipFree(d));
  CHECK(hipStreamDestroy(s));
  return 0;
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  printf("device: %s, %d CUs\n", p.name, p.multiProcessorCount);
  const int ncu = p.multiProcessorCount, nw = (ncu + 31) / 32;
  auto mk = [&](auto pred) { std::vector<uint32_t> m(nw, 0u); for (int i = 0; i < ncu; i++) if (pred(i)) m[i / 32] |= 1u << (i % 32); return m; };
  if (run("all bits", mk([](int) { return true; }))) return 1;
  if (run("bits 0..63", mk([](int i) { return i < 64; }))) return 1;
  if (run("bits 0..7", mk([](int i) { return i < 8; }))) return 1;
  if (run("bits 64..255", mk([](int i) { return i >= 64; }))) return 1;
  if (run("bits with i%8 < 2", mk([](int i) { return i % 8 < 2; }))) return 1;
  if (run("bits with i%32 < 8", mk([](int i) { return i % 32 < 8; }))) return 1;
  if (run("bit 0 only", mk([](int i) { return i == 0; }))) return 1;
  if (run("bit 1 only", mk([](int i) { return i == 1; }))) return 1;
  if (run("bit 8 only", mk([](int i) { return i == 8; }))) return 1;
  return 0;
}
