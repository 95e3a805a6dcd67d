// ubench_valu.hip -- VALU instruction-throughput micro-benchmark for gfx950.
// Answers SURVEY.md section 7 "first task on the GPU box": how fast are the integer
// multiply forms a 255-bit modular multiplier can be built from, relative to plain adds
// and to FP64 FMA?  Prints ops/s chip-wide and cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;  // independent chains per thread

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
  uint32_t a[UNROLL], b = seed * 2654435761u + threadIdx.x;
  uint64_t acc[UNROLL];
  double d[UNROLL];
  for (int i = 0; i < UNROLL; i++) { a[i] = seed + i * 7919u + threadIdx.x; acc[i] = a[i]; d[i] = (double)a[i]; }
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < UNROLL; i++) {
      if (OP == 0) {  // v_mad_u64_u32
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b) : "vcc");
      } else if (OP == 1) {  // v_mul_lo_u32
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      } else if (OP == 2) {  // v_mul_hi_u32
        asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      } else if (OP == 3) {  // v_add_u32
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      } else if (OP == 4) {  // v_lshl_add_u64
        asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) % UNROLL]));
      } else if (OP == 5) {  // v_fma_f64
        asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) % UNROLL]));
      } else if (OP == 6) {  // v_mov_b32
        asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
      } else if (OP == 7) {  // v_addc_co_u32 with carry in/out through vcc
        asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
      } else if (OP == 8) {  // v_mad_u32_u24
        asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
      } else if (OP == 9) {  // v_mul_hi_u32_u24
        asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      } else if (OP == 10) {  // v_mad_u64_u32 with carry out to an SGPR pair + addc consuming it
        uint64_t cy;
        asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc[i]), "=s"(cy) : "v"(a[i]), "v"(b));
        asm volatile("v_addc_co_u32 %0, vcc, %0, 0, %1" : "+v"(a[i]) : "s"(cy) : "vcc");
      } else if (OP == 11) {  // v_fma_f32
        float f; asm volatile("v_fma_f32 %0, %1, %1, %1" : "=v"(f) : "v"(b)); a[i] += (uint32_t)f * 0;
      } else if (OP == 12) {  // v_add_co_u32 + v_addc_co_u32 pair (64-bit add)
        asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, %2, 0, vcc" : "+v"(a[i]), "+v"(b) : "v"(b), "v"(a[(i+1)%UNROLL]) : "vcc");
      }
    }
  }
  uint32_t r = 0;
  for (int i = 0; i < UNROLL; i++) r ^= a[i] ^ (uint32_t)acc[i] ^ (uint32_t)(acc[i] >> 32) ^ (uint32_t)d[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
int run(const char* name, int per_iter_instrs, int cus, double clock_ghz) {
  int blocks = cus * 8, threads = 256;  // 8 waves per SIMD
  uint32_t* out;
  CHECK(hipMalloc(&out, (size_t)blocks * threads * 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 1u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 2u);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double wave_instrs = (double)blocks * (threads / 64) * ITERS * UNROLL * per_iter_instrs;
  double lane_ops = wave_instrs * 64;
  double simd_cycles = ms * 1e-3 * clock_ghz * 1e9;           // cycles elapsed on each SIMD
  double per_simd = wave_instrs / (cus * 4.0);                 // wave-instructions per SIMD
  printf("%-28s %8.3f ms  %8.2f T lane-ops/s  %6.2f cycles/wave-instr/SIMD (at %.2f GHz)\n", name, ms,
         lane_ops / (ms * 1e-3) / 1e12, simd_cycles / per_simd, clock_ghz);
  CHECK(hipFree(out));
  return 0;
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  double ghz = p.clockRate / 1e6;
  printf("device: %s  CUs=%d  clock=%.2f GHz  arch=%s\n", p.name, cus, ghz, p.gcnArchName);
  run<3>("v_add_u32", 1, cus, ghz);
  run<6>("v_mov_b32", 1, cus, ghz);
  run<7>("v_addc_co_u32", 1, cus, ghz);
  run<0>("v_mad_u64_u32", 1, cus, ghz);
  run<10>("v_mad_u64_u32+v_addc(sgpr)", 2, cus, ghz);
  run<1>("v_mul_lo_u32", 1, cus, ghz);
  run<2>("v_mul_hi_u32", 1, cus, ghz);
  run<8>("v_mad_u32_u24", 1, cus, ghz);
  run<9>("v_mul_hi_u32_u24", 1, cus, ghz);
  run<4>("v_lshl_add_u64", 1, cus, ghz);
  run<12>("v_add_co+v_addc_co (64b add)", 2, cus, ghz);
  run<5>("v_fma_f64", 1, cus, ghz);
  run<11>("v_fma_f32", 1, cus, ghz);
  return 0;
}
