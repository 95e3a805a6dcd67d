// ubench_fs.hip -- what a Fiat-Shamir turn-around costs on MI355X, to decide where the transcript of the small
// sum-check rounds should live (DESIGN.md "round protocol"):
//   (a) Keccak-f[1600] on one GPU lane (the whole state in registers),
//   (b) Keccak-f[1600] spread over 25 lanes of one wave (one 64-bit word per lane, ds_bpermute exchanges),
//   (c) a modular multiplication chain on one lane (the transcript's scalar work: from_bytes_wide, unipoly),
//   (d) a mailbox round trip: a resident kernel publishes a word to pinned host memory and polls a pinned
//       host word for the answer; the host thread polls and answers (what a host-resident transcript costs a
//       persistent round kernel).
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench_fs.hip -o /tmp/ubench_fs
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <thread>

__constant__ uint64_t RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

__device__ __forceinline__ uint64_t rol(uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; }

// (a) one lane, fully unrolled round (same statement as host/transcript.h)
__device__ void keccak_lane(uint64_t* a) {
  uint64_t a00 = a[0], a10 = a[1], a20 = a[2], a30 = a[3], a40 = a[4], a01 = a[5], a11 = a[6], a21 = a[7], a31 = a[8],
           a41 = a[9], a02 = a[10], a12 = a[11], a22 = a[12], a32 = a[13], a42 = a[14], a03 = a[15], a13 = a[16],
           a23 = a[17], a33 = a[18], a43 = a[19], a04 = a[20], a14 = a[21], a24 = a[22], a34 = a[23], a44 = a[24];
#pragma unroll 1
  for (int round = 0; round < 24; round++) {
    uint64_t c0 = a00 ^ a01 ^ a02 ^ a03 ^ a04, c1 = a10 ^ a11 ^ a12 ^ a13 ^ a14, c2 = a20 ^ a21 ^ a22 ^ a23 ^ a24,
             c3 = a30 ^ a31 ^ a32 ^ a33 ^ a34, c4 = a40 ^ a41 ^ a42 ^ a43 ^ a44;
    uint64_t d0 = c4 ^ rol(c1, 1), d1 = c0 ^ rol(c2, 1), d2 = c1 ^ rol(c3, 1), d3 = c2 ^ rol(c4, 1), d4 = c3 ^ rol(c0, 1);
    a00 ^= d0; a01 ^= d0; a02 ^= d0; a03 ^= d0; a04 ^= d0;
    a10 ^= d1; a11 ^= d1; a12 ^= d1; a13 ^= d1; a14 ^= d1;
    a20 ^= d2; a21 ^= d2; a22 ^= d2; a23 ^= d2; a24 ^= d2;
    a30 ^= d3; a31 ^= d3; a32 ^= d3; a33 ^= d3; a34 ^= d3;
    a40 ^= d4; a41 ^= d4; a42 ^= d4; a43 ^= d4; a44 ^= d4;
    uint64_t b00 = a00, b13 = rol(a01, 36), b21 = rol(a02, 3), b34 = rol(a03, 41), b42 = rol(a04, 18);
    uint64_t b02 = rol(a10, 1), b10 = rol(a11, 44), b23 = rol(a12, 10), b31 = rol(a13, 45), b44 = rol(a14, 2);
    uint64_t b04 = rol(a20, 62), b12 = rol(a21, 6), b20 = rol(a22, 43), b33 = rol(a23, 15), b41 = rol(a24, 61);
    uint64_t b01 = rol(a30, 28), b14 = rol(a31, 55), b22 = rol(a32, 25), b30 = rol(a33, 21), b43 = rol(a34, 56);
    uint64_t b03 = rol(a40, 27), b11 = rol(a41, 20), b24 = rol(a42, 39), b32 = rol(a43, 8), b40 = rol(a44, 14);
    a00 = b00 ^ (~b10 & b20); a10 = b10 ^ (~b20 & b30); a20 = b20 ^ (~b30 & b40); a30 = b30 ^ (~b40 & b00); a40 = b40 ^ (~b00 & b10);
    a01 = b01 ^ (~b11 & b21); a11 = b11 ^ (~b21 & b31); a21 = b21 ^ (~b31 & b41); a31 = b31 ^ (~b41 & b01); a41 = b41 ^ (~b01 & b11);
    a02 = b02 ^ (~b12 & b22); a12 = b12 ^ (~b22 & b32); a22 = b22 ^ (~b32 & b42); a32 = b32 ^ (~b42 & b02); a42 = b42 ^ (~b02 & b12);
    a03 = b03 ^ (~b13 & b23); a13 = b13 ^ (~b23 & b33); a23 = b23 ^ (~b33 & b43); a33 = b33 ^ (~b43 & b03); a43 = b43 ^ (~b03 & b13);
    a04 = b04 ^ (~b14 & b24); a14 = b14 ^ (~b24 & b34); a24 = b24 ^ (~b34 & b44); a34 = b34 ^ (~b44 & b04); a44 = b44 ^ (~b04 & b14);
    a00 ^= RC[round];
  }
  a[0] = a00; a[1] = a10; a[2] = a20; a[3] = a30; a[4] = a40; a[5] = a01; a[6] = a11; a[7] = a21; a[8] = a31; a[9] = a41;
  a[10] = a02; a[11] = a12; a[12] = a22; a[13] = a32; a[14] = a42; a[15] = a03; a[16] = a13; a[17] = a23; a[18] = a33;
  a[19] = a43; a[20] = a04; a[21] = a14; a[22] = a24; a[23] = a34; a[24] = a44;
}

__global__ void k_keccak_lane(uint64_t* st, int n) {
  if (threadIdx.x) return;
  uint64_t a[25];
  for (int i = 0; i < 25; i++) a[i] = st[i];
  for (int i = 0; i < n; i++) keccak_lane(a);
  for (int i = 0; i < 25; i++) st[i] = a[i];
}

// (b) lane l = x + 5y holds A[x][y]; lanes >= 25 idle along
__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src) {
  uint32_t lo = __shfl((uint32_t)v, src), hi = __shfl((uint32_t)(v >> 32), src);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t rolv(uint64_t v, uint32_t n) {
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  if (n & 32) { uint32_t t = lo; lo = hi; hi = t; }
  uint32_t nh = __funnelshift_l(lo, hi, n & 31), nl = __funnelshift_l(hi, lo, n & 31);
  return ((uint64_t)nh << 32) | nl;
}
__constant__ int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};  // r[x + 5y]

__device__ uint64_t keccak_wave(uint64_t a) {
  const int l = threadIdx.x & 63, x = l % 5, y = (l / 5) % 5;
  const bool live = l < 25;
  // column mates, row neighbours, rho-pi source and its rotation: fixed per lane
  int col[4];
  for (int k = 1; k <= 4; k++) col[k - 1] = x + 5 * ((y + k) % 5);
  const int xm = (x + 4) % 5 + 5 * y, xp = (x + 1) % 5 + 5 * y, xpp = (x + 2) % 5 + 5 * y;
  const int sx = (x + 3 * y) % 5, sy = x, src = sx + 5 * sy;
  const uint32_t rot = (uint32_t)RHO[live ? src : 0];
#pragma unroll 1
  for (int round = 0; round < 24; round++) {
    uint64_t c = a ^ shfl64(a, col[0]) ^ shfl64(a, col[1]) ^ shfl64(a, col[2]) ^ shfl64(a, col[3]);
    uint64_t d = shfl64(c, xm) ^ rol(shfl64(c, xp), 1);
    a ^= d;
    uint64_t b = rolv(shfl64(a, src), rot);
    a = b ^ (~shfl64(b, xp) & shfl64(b, xpp));
    if (l == 0) a ^= RC[round];
  }
  return a;
}

__global__ void k_keccak_wave(uint64_t* st, int n) {
  const int l = threadIdx.x;
  uint64_t a = l < 25 ? st[l] : 0;
  for (int i = 0; i < n; i++) a = keccak_wave(a);
  if (l < 25) st[l] = a;
}

// (c) a dependent chain of 256-bit Montgomery-like products (8x8 32-bit limb MACs + folding), one lane
__global__ void k_mulchain(uint32_t* io, int n) {
  if (threadIdx.x) return;
  uint32_t a[8], b[8];
  for (int i = 0; i < 8; i++) { a[i] = io[i]; b[i] = io[8 + i]; }
  for (int it = 0; it < n; it++) {
    uint32_t t[16] = {0};
    for (int i = 0; i < 8; i++) {
      uint64_t carry = 0;
      for (int j = 0; j < 8; j++) {
        uint64_t p = (uint64_t)a[i] * b[j] + t[i + j] + carry;
        t[i + j] = (uint32_t)p;
        carry = p >> 32;
      }
      t[i + 8] = (uint32_t)carry;
    }
    // cheap stand-in for the reduction: 8 more MAC rows against a constant
    for (int i = 0; i < 8; i++) {
      uint64_t carry = 0;
      uint32_t m = t[i] * 0x12547e1bu;
      for (int j = 0; j < 4; j++) {
        uint64_t p = (uint64_t)m * (0x5cf5d3edu + j) + t[i + j] + carry;
        t[i + j] = (uint32_t)p;
        carry = p >> 32;
      }
      t[i + 4] += (uint32_t)carry;
    }
    for (int i = 0; i < 8; i++) a[i] = t[8 + i];
  }
  for (int i = 0; i < 8; i++) io[i] = a[i];
}

// (d) mailbox: GPU -> host word `up`, host -> GPU word `down`, both in pinned host memory
__global__ void k_mailbox(uint32_t* up, uint32_t* down, int n, uint32_t* bailed) {
  if (threadIdx.x) return;
  for (int i = 1; i <= n; i++) {
    __hip_atomic_store(up, (uint32_t)i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    long spins = 0;
    while (__hip_atomic_load(down, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != (uint32_t)i) {
      if (++spins > 20000000) { *bailed = (uint32_t)i; return; }  // every wave reaches an exit
    }
  }
}

int main() {
  using C = std::chrono::steady_clock;
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  uint64_t* d;
  hipMalloc((void**)&d, 4096);
  hipMemset(d, 1, 4096);
  uint64_t h0[25], h1[25];
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int n = 200;
  {
    // known answer: Keccak-f[1600] of the all-zero state starts with lane 0 = f1258f7940e1dde7 (FIPS 202 test vectors)
    hipMemset(d, 0, 200);
    hipLaunchKernelGGL(k_keccak_wave, dim3(1), dim3(64), 0, s, d, 1);
    hipStreamSynchronize(s);
    hipMemcpy(h0, d, 200, hipMemcpyDeviceToHost);
    printf("keccak-f(0) lane 0 = %016llx (%s)\n", (unsigned long long)h0[0], h0[0] == 0xf1258f7940e1dde7ULL ? "matches FIPS 202" : "WRONG");
  }
  for (int rep = 0; rep < 2; rep++) {
    float ms_a, ms_b, ms_c;
    hipMemset(d, 1, 200);
    hipEventRecord(e0, s); hipLaunchKernelGGL(k_keccak_lane, dim3(1), dim3(64), 0, s, d, n); hipEventRecord(e1, s);
    hipStreamSynchronize(s); hipEventElapsedTime(&ms_a, e0, e1);
    hipMemcpy(h0, d, 200, hipMemcpyDeviceToHost);
    hipMemset(d, 1, 200);
    hipEventRecord(e0, s); hipLaunchKernelGGL(k_keccak_wave, dim3(1), dim3(64), 0, s, d, n); hipEventRecord(e1, s);
    hipStreamSynchronize(s); hipEventElapsedTime(&ms_b, e0, e1);
    hipMemcpy(h1, d, 200, hipMemcpyDeviceToHost);
    bool same = true;
    for (int i = 0; i < 25; i++) same = same && h0[i] == h1[i];
    hipEventRecord(e0, s); hipLaunchKernelGGL(k_mulchain, dim3(1), dim3(64), 0, s, (uint32_t*)d, 1000); hipEventRecord(e1, s);
    hipStreamSynchronize(s); hipEventElapsedTime(&ms_c, e0, e1);
    printf("rep %d: keccak-f one lane %.2f us | 25 lanes %.2f us (states %s) | 256-bit mulmod chain %.3f us per product\n", rep,
           ms_a * 1e3 / n, ms_b * 1e3 / n, same ? "equal" : "DIFFER", ms_c * 1e3 / 1000);
  }
  // mailbox
  uint32_t* hp;
  hipHostMalloc((void**)&hp, 4096, hipHostMallocDefault);
  volatile uint32_t* up = hp;
  volatile uint32_t* down = hp + 64;
  uint32_t* bailed = hp + 128;
  for (int rep = 0; rep < 2; rep++) {
    const int rounds = 5000;
    *up = 0; *down = 0; *bailed = 0;
    auto t0 = C::now();
    hipLaunchKernelGGL(k_mailbox, dim3(1), dim3(64), 0, s, hp, hp + 64, rounds, bailed);
    for (int i = 1; i <= rounds; i++) {
      long spins = 0;
      while (*up != (uint32_t)i && ++spins < 2000000000L) __builtin_ia32_pause();
      *down = (uint32_t)i;
    }
    hipStreamSynchronize(s);
    double us = std::chrono::duration<double>(C::now() - t0).count() / rounds * 1e6;
    printf("rep %d: mailbox round trip (GPU store -> host poll -> host store -> GPU poll) %.2f us%s\n", rep, us,
           *bailed ? "  [kernel bailed out]" : "");
  }
  return 0;
}
