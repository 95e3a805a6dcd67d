// ubench_msm_variants.hip -- VERDICT r3 item 2: the two variants of the row-commitment MSM's inner loop that rounds 2-3 only
// costed, BUILT and measured next to the shipped one, plus the clock / power the chip sustains under each instruction mix.
//
//   ref   acc <- acc + (affine table entry) in extended coordinates, ten limbs of 26/25 bits (fp10_dev.h ge10_add_niels:
//         7 products of 101 v_mad_u64_u32): what msm_rows_kernel runs, on a register-resident chain
//   (a)   BATCHED AFFINE: every lane keeps K independent affine partial sums; a round adds one affine table point to each of
//         them with ONE inversion (Montgomery's trick over the K denominators): 3(K-1) products for the trick, an inversion
//         of 254 squarings + 11 products, and per addition the short-Weierstrass chord formulas (lambda = dy/dx,
//         x3 = lambda^2 - x1 - x2, y3 = lambda (x1 - x3) - y1: 2 products + 1 squaring) -- "5M + 1S instead of 7M" is what
//         the addition itself costs; the inversion is I/K on top.  K = 4, 8, 16.  The arithmetic is checked on the device
//         (every denominator times its inverse is 1).  (curve25519's Edwards form would need two denominators per addition;
//         the Weierstrass model is the variant's best case.)
//   (b)   FP64-FMA product: GF(2^255-19) in five 51-bit limbs held as doubles; a limb product is split into its high and low
//         52 bits by two v_fma_f64 (round toward zero, magic constants 2^104 and 2^104 + 2^52) and the halves are summed in
//         64-bit integer columns -- the scheme of Emmart, Zheng, Weems ("Faster modular exponentiation using double precision
//         floating point arithmetic on the GPU", ARITH 2018).  25 limb products = 50 FMAs + 25 subtractions + 50 64-bit
//         adds; checked against fp_mul.
//   gather  the ref chain with a 96-byte gather from a multi-GB table per addition (what the real kernel adds: HBM power)
//
// While each kernel runs (>= ~0.4 s) a host thread samples the GPU's sclk and socket power from sysfs (hwmon freq1_input /
// power1_average, pp_dpm_sclk) so the report says at which clock each mix really runs: the MSM is VALU-issue bound, and the
// SQ counters of round 3 already hinted that the chip does not hold 2.4 GHz under it.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vpin_amd/csrc tools/ubench_msm_variants.hip -o tools/ubench_msm_variants -lpthread
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <glob.h>
#include <string>
#include <thread>
#include <vector>

#include "fp10_dev.h"

using namespace vpin;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ---------------------------------------------------------------------------------------------------------------------
// ref: the shipped point addition on a chain

__global__ __launch_bounds__(256, 3) void chain_ref(const fp* __restrict__ in, fp* __restrict__ out, int iters) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  ge10 acc = ge10_identity();
  ge_niels q;
  q.ypx = fp_load(in + 2 * t); q.ymx = fp_load(in + 2 * t + 1); q.xy2d = fp_add(q.ypx, q.ymx);
  for (int i = 0; i < iters; i++) acc = ge10_add_niels(acc, q, (i & 1) != 0);
  const ge_ext e = ge10_to_ext(acc);
  fp_store(out + t, fp_freeze(fp_add(fp_add(e.X, e.Y), fp_add(e.Z, e.T))));
}

// the same with one 96-byte gather per addition from `table` (n_entries a power of two), software-pipelined like
// table_mul_acc10 (the entry of step i+1 is requested before the addition of step i)
__global__ __launch_bounds__(256, 3) void chain_gather(const ge_niels* __restrict__ table, uint32_t mask, fp* __restrict__ out, int iters) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  ge10 acc = ge10_identity();
  uint32_t s = (uint32_t)t * 2654435761u + 12345u;
  auto load = [&](uint32_t idx) {
    ge_niels q;
    const ge_niels* p = table + (idx & mask);
    q.ypx = fp_load(&p->ypx); q.ymx = fp_load(&p->ymx); q.xy2d = fp_load(&p->xy2d);
    return q;
  };
  ge_niels cur = load(s);
  for (int i = 0; i < iters; i++) {
    s = s * 1664525u + 1013904223u;
    const ge_niels nxt = load(s >> 4);
    acc = ge10_add_niels(acc, cur, (s & 1) != 0);
    cur = nxt;
  }
  const ge_ext e = ge10_to_ext(acc);
  fp_store(out + t, fp_freeze(fp_add(fp_add(e.X, e.Y), fp_add(e.Z, e.T))));
}

__global__ void fill_kernel(uint32_t* p, size_t n_words) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride) {
    uint64_t z = i * 0x9e3779b97f4a7c15ull + 0x632be59bd9b4e019ull;
    z ^= z >> 29; z *= 0xbf58476d1ce4e5b9ull; z ^= z >> 32;
    p[i] = ((i & 7) == 7) ? ((uint32_t)z & 0x7fffffffu) : (uint32_t)z;  // every 32-byte element < 2^255
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// The table walk of msm_rows_kernel with its real access pattern, over a synthetic table of the real shape: W windows x NB
// generators x E multiples.  A workgroup is a commitment row; its lanes take generators j = 256 s + lane (all rows walk the
// generators in the same order, like the compacted lists of the real rows) and add one entry per window, digits drawn at
// random.  LAYOUT 0: entry (w, j, d) at ((w NB + j) E + d) -- the shipped one: a lane's 22 gathers are 3.2 GB apart;
//         1: ((j W + w) E + d): a lane's walk stays inside 4.3 MB.   SLOT: bytes per entry (96 packed, 128 = one line each).
//         DEPTH: entries requested ahead of the addition that uses them (the shipped loop: 1).
//         2: ROW PER LANE -- a lane is a commitment row and every lane of every workgroup of a "strip" adds the entry of the
//            SAME (window, generator) at the same step, each with its own digit: the 196 KB block of that (w, j) is read
//            by thousands of lanes within microseconds, i.e. out of L2 instead of HBM.  Workgroup b works on strip
//            2 (b mod 8) + (b / 8 mod 2): workgroups are dealt round-robin to the 8 XCDs, so a strip stays on one L2.
template <int LAYOUT, int SLOT, int DEPTH>
__global__ __launch_bounds__(256, 3) void walk_kernel(const uint8_t* __restrict__ table, uint32_t NB, int W, uint32_t E, int scalars,
                                                      fp* __restrict__ out) {
  ge10 acc = ge10_identity();
  const uint32_t row = blockIdx.x;
  auto entry = [&](uint32_t j, int w, uint32_t d) {
    const size_t idx = LAYOUT != 1 ? ((size_t)w * NB + j) * E + d : ((size_t)j * W + w) * E + d;
    const ge_niels* p = reinterpret_cast<const ge_niels*>(table + idx * SLOT);
    ge_niels q;
    q.ypx = fp_load(&p->ypx); q.ymx = fp_load(&p->ymx); q.xy2d = fp_load(&p->xy2d);
    return q;
  };
  uint32_t rng = (row * 256u + threadIdx.x) * 2654435761u + 99u;
  auto next_digit = [&]() { rng = rng * 1664525u + 1013904223u; return (rng >> 9) & (E - 1); };
  const int total = scalars * W;
  ge_niels ring[DEPTH];
  // flat (scalar, window) counter so that the pipeline runs across scalars, as the compiler's loop does not in the shipped
  // kernel (its prefetch restarts with every scalar)
  const uint32_t strip = 2u * (blockIdx.x & 7u) + ((blockIdx.x >> 3) & 1u), strip_len = NB / 16u;
  auto coords = [&](int k, uint32_t* j, int* w) {
    if (LAYOUT == 2) *j = strip * strip_len + ((uint32_t)(k / W) % strip_len);
    else *j = ((uint32_t)(k / W) * 256u + threadIdx.x) & (NB - 1);
    *w = k % W;
  };
#pragma unroll
  for (int k = 0; k < DEPTH; k++) { uint32_t j; int w; coords(k, &j, &w); ring[k] = entry(j, w, next_digit()); }
  for (int k = 0; k < total; k += DEPTH) {
#pragma unroll
    for (int u = 0; u < DEPTH; u++) {
      const ge_niels cur = ring[u];
      if (k + u + DEPTH < total) { uint32_t j; int w; coords(k + u + DEPTH, &j, &w); ring[u] = entry(j, w, next_digit()); }
      acc = ge10_add_niels(acc, cur, (rng & 1) != 0);
    }
  }
  const ge_ext e = ge10_to_ext(acc);
  fp_store(out + (size_t)row * 256 + threadIdx.x, fp_freeze(fp_add(fp_add(e.X, e.Y), fp_add(e.Z, e.T))));
}

// ---------------------------------------------------------------------------------------------------------------------
// (a) batched affine

__device__ __noinline__ fe10 fe10_sqr_n(fe10 a, int n) {
  for (int i = 0; i < n; i++) a = fe10_mul(a, a);
  return a;
}
// a^(p-2), the exponentiation chain of fp_invert in the ten-limb form (254 squarings, 11 products)
__device__ __noinline__ fe10 fe10_invert(const fe10& z) {
  fe10 t0 = fe10_sqr_n(z, 1);
  fe10 t1 = fe10_mul(z, fe10_sqr_n(t0, 2));
  t0 = fe10_mul(t0, t1);
  t0 = fe10_mul(t1, fe10_sqr_n(t0, 1));
  t0 = fe10_mul(fe10_sqr_n(t0, 5), t0);
  t1 = fe10_mul(fe10_sqr_n(t0, 10), t0);
  fe10 t2 = fe10_mul(fe10_sqr_n(t1, 20), t1);
  t1 = fe10_mul(fe10_sqr_n(t2, 10), t0);
  t2 = fe10_mul(fe10_sqr_n(t1, 50), t1);
  fe10 t3 = fe10_mul(fe10_sqr_n(t2, 100), t2);
  t1 = fe10_mul(fe10_sqr_n(t3, 50), t1);
  const fe10 p58 = fe10_mul(fe10_sqr_n(t1, 2), z);      // z^(2^252 - 3)
  return fe10_mul(fe10_sqr_n(p58, 3), fe10_mul(fe10_mul(z, z), z));
}
// one carry pass: any bound <= 4x -> 1x (keeps the value mod p)
__device__ __forceinline__ fe10 fe10_carry(const fe10& a) {
  fe10 r = a;
  uint32_t c;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int bits = (i & 1) ? 25 : 26;
    c = r.v[i] >> bits; r.v[i] &= (1u << bits) - 1u; r.v[i + 1] += c;
  }
  c = r.v[9] >> 25; r.v[9] &= 0x1ffffffu; r.v[0] += 19u * c;
  return r;
}

template <int K>
__global__ __launch_bounds__(256) void chain_batched_affine(const fp* __restrict__ in, fp* __restrict__ out, int rounds, unsigned* __restrict__ bad) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  fe10 ax[K], ay[K];
  const fe10 qx0 = fe10_from_fp(fp_load(in + 2 * t)), qy0 = fe10_from_fp(fp_load(in + 2 * t + 1));
#pragma unroll
  for (int k = 0; k < K; k++) {  // distinct starting sums
    fe10 d = fe10_zero(); d.v[0] = 3u + 2u * k;
    ax[k] = fe10_carry(fe10_add(qx0, d));
    ay[k] = fe10_carry(fe10_add(qy0, d));
  }
  fe10 qx = qx0, qy = qy0;
  unsigned wrong = 0;
  for (int r = 0; r < rounds; r++) {
    // the "table point" of this round (the real kernel gathers it): keep it moving so that nothing is loop invariant
    qx = fe10_carry(fe10_add(qx, qy0)); qy = fe10_carry(fe10_add(qy, qx0));
    // Montgomery's trick over the K denominators dx_k = qx - ax_k
    fe10 dx[K], pre[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
      dx[k] = fe10_carry(fe10_sub(qx, ax[k]));
      pre[k] = k == 0 ? dx[0] : fe10_mul(pre[k - 1], dx[k]);
    }
    fe10 inv = fe10_invert(pre[K - 1]);
#pragma unroll
    for (int k = K - 1; k >= 0; k--) {
      const fe10 idx = k == 0 ? inv : fe10_mul(inv, pre[k - 1]);   // 1 / dx_k
      if (k > 0) inv = fe10_mul(inv, dx[k]);
      if (r == rounds - 1) {  // self-check on the last round: dx_k * (1/dx_k) == 1
        const fp one = fp_freeze(fe10_to_fp(fe10_mul(dx[k], idx)));
        unsigned o = one.v[0] ^ 1u;
        for (int i = 1; i < 8; i++) o |= one.v[i];
        wrong |= o;
      }
      // chord addition: lambda = (qy - ay) / (qx - ax); x3 = lambda^2 - ax - qx; y3 = lambda (ax - x3) - ay
      const fe10 lam = fe10_mul(fe10_sub(qy, ay[k]), idx);
      const fe10 x3 = fe10_carry(fe10_sub(fe10_sub(fe10_mul(lam, lam), ax[k]), qx));
      const fe10 y3 = fe10_carry(fe10_sub(fe10_mul(fe10_sub(ax[k], x3), lam), ay[k]));
      ax[k] = x3; ay[k] = y3;
    }
  }
  fe10 s = fe10_zero();
#pragma unroll
  for (int k = 0; k < K; k++) s = fe10_carry(fe10_add(s, fe10_add(ax[k], ay[k])));
  fp_store(out + t, fp_freeze(fe10_to_fp(s)));
  if (wrong) atomicAdd(bad, 1u);
}

// ---------------------------------------------------------------------------------------------------------------------
// (b) FP64-FMA product, five 51-bit limbs as doubles (each < 2^52 on entry to a product)

struct fe5 { double v[5]; };

__device__ __forceinline__ fe5 fe5_from_fp(const fp& a8) {
  const fp a = fp_freeze(a8);
  uint64_t w[4];
  for (int i = 0; i < 4; i++) w[i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32);
  const uint64_t m = (1ull << 51) - 1;
  uint64_t l[5];
  l[0] = w[0] & m;
  l[1] = ((w[0] >> 51) | (w[1] << 13)) & m;
  l[2] = ((w[1] >> 38) | (w[2] << 26)) & m;
  l[3] = ((w[2] >> 25) | (w[3] << 39)) & m;
  l[4] = (w[3] >> 12) & m;
  fe5 r;
  for (int i = 0; i < 5; i++) r.v[i] = (double)l[i];  // exact: < 2^51
  return r;
}
__device__ __forceinline__ fp fe5_to_fp(const fe5& f) {
  uint64_t l[5];
  for (int i = 0; i < 5; i++) l[i] = (uint64_t)f.v[i];
  // carry to 51 bits, wrap with 19
  for (int pass = 0; pass < 2; pass++) {
    uint64_t c;
    for (int i = 0; i < 4; i++) { c = l[i] >> 51; l[i] &= (1ull << 51) - 1; l[i + 1] += c; }
    c = l[4] >> 51; l[4] &= (1ull << 51) - 1; l[0] += 19 * c;
  }
  uint64_t w[4];
  w[0] = l[0] | (l[1] << 51);
  w[1] = (l[1] >> 13) | (l[2] << 38);
  w[2] = (l[2] >> 26) | (l[3] << 25);
  w[3] = (l[3] >> 39) | (l[4] << 12);
  fp r;
  for (int i = 0; i < 4; i++) { r.v[2 * i] = (uint32_t)w[i]; r.v[2 * i + 1] = (uint32_t)(w[i] >> 32); }
  return r;
}

// f, g: limbs < 2^52  ->  limbs < 2^51 + small.  Must run with the double-precision rounding mode = toward zero.
__device__ __forceinline__ fe5 fe5_mul(const fe5& f, const fe5& g) {
  const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;
  // column sums of the raw bit patterns: lo[k] holds sum (2^52 + lo) patterns, hi[k] sum (2^104 + hi) patterns
  uint64_t lo[9], hi[9];
#pragma unroll
  for (int k = 0; k < 9; k++) { lo[k] = 0; hi[k] = 0; }
#pragma unroll
  for (int i = 0; i < 5; i++) {
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const double ph = __builtin_fma(f.v[i], g.v[j], C1);        // 2^104 + floor(fg / 2^52) 2^52
      const double pl = __builtin_fma(f.v[i], g.v[j], C2 - ph);   // 2^52 + (fg mod 2^52)
      hi[i + j] += (uint64_t)__double_as_longlong(ph);
      lo[i + j] += (uint64_t)__double_as_longlong(pl);
    }
  }
  // strip the exponent patterns: n terms per column, each pattern = (exp << 52) + mantissa field
  //   ph: value 2^104 + H 2^52 with H < 2^52 -> bits = (0x467 << 52) | H        (1023 + 104 = 0x467)
  //   pl: value 2^52 + L                      -> bits = (0x433 << 52) | L        (1023 + 52  = 0x433)
  uint64_t col[10];
#pragma unroll
  for (int k = 0; k < 10; k++) col[k] = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const uint64_t n = (uint64_t)(k < 5 ? k + 1 : 9 - k);
    const uint64_t L = lo[k] - n * (0x433ull << 52), H = hi[k] - n * (0x467ull << 52);
    col[k] += L;            // weight 2^(51 k)
    col[k + 1] += H << 1;   // weight 2^(51 k + 52) = 2 * 2^(51 (k+1))
  }
  // 2^255 = 19: columns 5..9 wrap (col < 2^57, 19 col < 2^62)
#pragma unroll
  for (int k = 0; k < 5; k++) col[k] += 19ull * col[k + 5];
  uint64_t c;
#pragma unroll
  for (int k = 0; k < 4; k++) { c = col[k] >> 51; col[k] &= (1ull << 51) - 1; col[k + 1] += c; }
  c = col[4] >> 51; col[4] &= (1ull << 51) - 1; col[0] += 19ull * c;
  c = col[0] >> 51; col[0] &= (1ull << 51) - 1; col[1] += c;
  fe5 r;
#pragma unroll
  for (int k = 0; k < 5; k++) r.v[k] = __longlong_as_double((long long)(col[k] | (0x433ull << 52))) - 0x1p52;  // exact int -> double
  return r;
}

// Double-precision rounding mode = toward zero (MODE register, FP_ROUND bits [3:2] = 3).  The compiler manages MODE itself
// (SIModeRegister: it switches back to round-to-nearest for its own u64 -> double expansion), so the switch is an asm
// statement placed AFTER the conversions, and the loop's operands pass through it so that no FMA is scheduled above it.
__device__ __forceinline__ void set_f64_round_toward_zero(fe5& x, fe5& y) {
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3"
               : "+v"(x.v[0]), "+v"(x.v[1]), "+v"(x.v[2]), "+v"(x.v[3]), "+v"(x.v[4]), "+v"(y.v[0]), "+v"(y.v[1]), "+v"(y.v[2]),
                 "+v"(y.v[3]), "+v"(y.v[4]));
}
__device__ __forceinline__ void set_f64_round_to_nearest(fe5& x) {
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0" : "+v"(x.v[0]), "+v"(x.v[1]), "+v"(x.v[2]), "+v"(x.v[3]), "+v"(x.v[4]));
}

__global__ __launch_bounds__(256, 3) void chain_fp64(const fp* __restrict__ in, fp* __restrict__ out, int iters) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  fe5 x = fe5_from_fp(fp_load(in + 2 * t));
  fe5 y = fe5_from_fp(fp_load(in + 2 * t + 1));
  set_f64_round_toward_zero(x, y);
  for (int i = 0; i < iters; i++) x = fe5_mul(x, y);
  set_f64_round_to_nearest(x);
  fp_store(out + t, fp_freeze(fe5_to_fp(x)));
}
__global__ __launch_bounds__(256, 3) void chain_mad10(const fp* __restrict__ in, fp* __restrict__ out, int iters) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  fe10 x = fe10_from_fp(fp_load(in + 2 * t));
  const fe10 y = fe10_from_fp(fp_load(in + 2 * t + 1));
  for (int i = 0; i < iters; i++) x = fe10_mul(x, y);
  fp_store(out + t, fp_freeze(fe10_to_fp(x)));
}
// plain VALU adds only: how fast the chip clocks when no multiplier is busy
__global__ __launch_bounds__(256, 3) void chain_adds(const fp* __restrict__ in, fp* __restrict__ out, int iters) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  fe10 x = fe10_from_fp(fp_load(in + 2 * t));
  const fe10 y = fe10_from_fp(fp_load(in + 2 * t + 1));
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 14; r++) {  // ~140 adds: one product's worth of instructions
#pragma unroll
      for (int l = 0; l < 10; l++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x.v[l]) : "v"(y.v[(l + r) % 10]));
    }
  }
  fp_store(out + t, fe10_to_fp(fe10_carry(fe10_carry(x))));
}

// ---------------------------------------------------------------------------------------------------------------------
// sclk / power sampling from sysfs while a kernel runs

struct Sampler {
  std::vector<std::string> freq_files, power_files;
  std::string dpm_file;
  std::atomic<bool> run{false};
  std::thread th;
  std::vector<double> mhz, watts;

  static std::vector<std::string> globv(const char* pat) {
    glob_t g;
    std::vector<std::string> v;
    if (glob(pat, 0, nullptr, &g) == 0) for (size_t i = 0; i < g.gl_pathc; i++) v.push_back(g.gl_pathv[i]);
    globfree(&g);
    return v;
  }
  static bool read_num(const std::string& f, double* out) {
    std::ifstream s(f);
    double v;
    if (!(s >> v)) return false;
    *out = v;
    return true;
  }
  // the hwmon files of THE device this process computes on (the host has eight cards, other tenants on some of them)
  explicit Sampler(const char* pci_bus_id) {
    std::string id(pci_bus_id);
    for (auto& ch : id) ch = (char)tolower(ch);
    const std::string base = "/sys/bus/pci/devices/" + id;
    freq_files = globv((base + "/hwmon/hwmon*/freq1_input").c_str());
    power_files = globv((base + "/hwmon/hwmon*/power1_average").c_str());
    if (power_files.empty()) power_files = globv((base + "/hwmon/hwmon*/power1_input").c_str());
    auto d = globv((base + "/pp_dpm_sclk").c_str());
    if (!d.empty()) dpm_file = d[0];
  }
  double dpm_mhz() const {
    std::ifstream s(dpm_file);
    std::string line;
    while (std::getline(s, line)) {
      if (line.find('*') == std::string::npos) continue;
      const size_t c = line.find(':');
      return c == std::string::npos ? 0.0 : atof(line.c_str() + c + 1);
    }
    return 0.0;
  }
  void start() {
    mhz.clear(); watts.clear();
    run = true;
    th = std::thread([this] {
      while (run) {
        double f = 0, w = 0, v, fmax = 0;
        for (auto& p : freq_files) if (read_num(p, &v) && v > fmax) fmax = v;   // the card under load is the fastest-clocked one
        f = fmax / 1e6;
        if (f == 0 && !dpm_file.empty()) f = dpm_mhz();
        for (auto& p : power_files) if (read_num(p, &v) && v / 1e6 > w) w = v / 1e6;
        mhz.push_back(f); watts.push_back(w);
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
      }
    });
  }
  void stop(double* f_med, double* w_med) {
    run = false;
    th.join();
    auto med = [](std::vector<double> v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    // drop the ramp: second half of the samples
    std::vector<double> a(mhz.begin() + mhz.size() / 2, mhz.end()), b(watts.begin() + watts.size() / 2, watts.end());
    *f_med = med(a); *w_med = med(b);
  }
};

struct Result { const char* name; double ms, units, mhz, watts; };

template <typename F>
static Result timed(const char* name, Sampler& sm, double units, F launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch();  // warm
  CK(hipDeviceSynchronize());
  sm.start();
  CK(hipEventRecord(e0));
  launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  double f, w;
  sm.stop(&f, &w);
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return Result{name, ms, units, f, w};
}

int main(int argc, char** argv) {
  const int scale = argc > 1 ? atoi(argv[1]) : 1;           // multiplies every chain length
  const size_t table_gb = argc > 2 ? (size_t)atoi(argv[2]) : 16;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int blocks = prop.multiProcessorCount * 3, threads = blocks * 256;
  fp *din, *da, *db;
  CK(hipMalloc(&din, (size_t)threads * 64)); CK(hipMalloc(&da, (size_t)threads * 32)); CK(hipMalloc(&db, (size_t)threads * 32));
  hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, (uint32_t*)din, (size_t)threads * 16);
  unsigned* dbad;
  CK(hipMalloc(&dbad, 4)); CK(hipMemset(dbad, 0, 4));
  // gather table: power of two entries of 96 B
  // shape of the walk table: 22 windows x NB generators x 2048 multiples, 128-byte slots (the 96-byte runs use its front)
  const int W = 22;
  const uint32_t E = 2048;
  uint32_t NB = 256;
  while ((size_t)W * (NB * 2) * E * 128 <= table_gb * (size_t)1e9) NB *= 2;
  const size_t table_bytes = (size_t)W * NB * E * 128;
  size_t n_ent = 1;
  while (n_ent * 2 * sizeof(ge_niels) <= table_bytes) n_ent *= 2;
  ge_niels* table;
  CK(hipMalloc(&table, table_bytes));
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (uint32_t*)table, table_bytes / 4);
  CK(hipDeviceSynchronize());
  char bus[64] = {0};
  CK(hipDeviceGetPCIBusId(bus, sizeof bus, 0));
  Sampler sm(bus);
  printf("device %s (%s), %d CUs, nominal %d MHz; %d lanes (3 workgroups of 256 per CU); sysfs: %zu freq, %zu power files, dpm %s\n", prop.name, bus,
         prop.multiProcessorCount, prop.clockRate / 1000, threads, sm.freq_files.size(), sm.power_files.size(), sm.dpm_file.empty() ? "-" : "yes");

  std::vector<Result> res;
  const int n_add = 12000 * scale, n_mul = 80000 * scale;
  res.push_back(timed("adds only (v_add_u32), 140 per unit", sm, (double)threads * n_mul, [&] { hipLaunchKernelGGL(chain_adds, dim3(blocks), dim3(256), 0, 0, din, da, n_mul); }));
  res.push_back(timed("product, ten 25.5-bit limbs (v_mad_u64_u32)", sm, (double)threads * n_mul, [&] { hipLaunchKernelGGL(chain_mad10, dim3(blocks), dim3(256), 0, 0, din, da, n_mul); }));
  res.push_back(timed("product, five 51-bit limbs (v_fma_f64)  [b]", sm, (double)threads * n_mul, [&] { hipLaunchKernelGGL(chain_fp64, dim3(blocks), dim3(256), 0, 0, din, db, n_mul); }));
  // (b) correctness: both chains computed x * y^n
  std::vector<uint32_t> ra((size_t)threads * 8), rb((size_t)threads * 8);
  CK(hipMemcpy(ra.data(), da, ra.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(rb.data(), db, rb.size() * 4, hipMemcpyDeviceToHost));
  size_t bad_b = 0;
  for (size_t i = 0; i < ra.size(); i++) bad_b += ra[i] != rb[i];
  res.push_back(timed("point addition, extended + affine entry (ref)", sm, (double)threads * n_add, [&] { hipLaunchKernelGGL(chain_ref, dim3(blocks), dim3(256), 0, 0, din, da, n_add); }));
  res.push_back(timed("ref + one 96-byte HBM gather per addition", sm, (double)threads * n_add, [&] { hipLaunchKernelGGL(chain_gather, dim3(blocks), dim3(256), 0, 0, table, (uint32_t)(n_ent - 1), da, n_add); }));
  {
    const int scalars = 200 * scale;  // per lane; x 22 windows
    const double units = (double)threads * scalars * W;
    const uint8_t* tb = (const uint8_t*)table;
    printf("walk table: %d windows x %u generators x %u multiples, %.1f GB at 96 B per entry, %.1f GB at 128 B\n", W, NB, E, (double)W * NB * E * 96 / 1e9,
           (double)table_bytes / 1e9);
    res.push_back(timed("walk [w][j][d], 96 B, prefetch 1 (shipped layout)", sm, units, [&] { hipLaunchKernelGGL((walk_kernel<0, 96, 1>), dim3(blocks), dim3(256), 0, 0, tb, NB, W, E, scalars, da); }));
    res.push_back(timed("walk [w][j][d], 96 B, prefetch 2", sm, units, [&] { hipLaunchKernelGGL((walk_kernel<0, 96, 2>), dim3(blocks), dim3(256), 0, 0, tb, NB, W, E, scalars, da); }));
    res.push_back(timed("walk [j][w][d], 96 B, prefetch 1", sm, units, [&] { hipLaunchKernelGGL((walk_kernel<1, 96, 1>), dim3(blocks), dim3(256), 0, 0, tb, NB, W, E, scalars, da); }));
    res.push_back(timed("walk [j][w][d], 96 B, prefetch 2", sm, units, [&] { hipLaunchKernelGGL((walk_kernel<1, 96, 2>), dim3(blocks), dim3(256), 0, 0, tb, NB, W, E, scalars, da); }));
    res.push_back(timed("walk [w][j][d], 128 B slots, prefetch 1", sm, units, [&] { hipLaunchKernelGGL((walk_kernel<0, 128, 1>), dim3(blocks), dim3(256), 0, 0, tb, NB, W, E, scalars, da); }));
    res.push_back(timed("walk [j][w][d], 128 B slots, prefetch 1", sm, units, [&] { hipLaunchKernelGGL((walk_kernel<1, 128, 1>), dim3(blocks), dim3(256), 0, 0, tb, NB, W, E, scalars, da); }));
    res.push_back(timed("walk [j][w][d], 128 B slots, prefetch 2", sm, units, [&] { hipLaunchKernelGGL((walk_kernel<1, 128, 2>), dim3(blocks), dim3(256), 0, 0, tb, NB, W, E, scalars, da); }));
    res.push_back(timed("walk ROW PER LANE (same (w,j) chip-wide), 96 B", sm, units, [&] { hipLaunchKernelGGL((walk_kernel<2, 96, 1>), dim3(blocks), dim3(256), 0, 0, tb, NB, W, E, scalars, da); }));
    res.push_back(timed("walk ROW PER LANE, 128 B slots", sm, units, [&] { hipLaunchKernelGGL((walk_kernel<2, 128, 1>), dim3(blocks), dim3(256), 0, 0, tb, NB, W, E, scalars, da); }));
  }
  const int rounds = 40 * scale;
  res.push_back(timed("batched affine, K = 4 sums per lane      [a]", sm, (double)threads * rounds * 4, [&] { hipLaunchKernelGGL(chain_batched_affine<4>, dim3(blocks), dim3(256), 0, 0, din, da, rounds, dbad); }));
  res.push_back(timed("batched affine, K = 8 sums per lane      [a]", sm, (double)threads * rounds * 8, [&] { hipLaunchKernelGGL(chain_batched_affine<8>, dim3(blocks), dim3(256), 0, 0, din, da, rounds, dbad); }));
  res.push_back(timed("batched affine, K = 16 sums per lane     [a]", sm, (double)threads * rounds * 16, [&] { hipLaunchKernelGGL(chain_batched_affine<16>, dim3(blocks), dim3(256), 0, 0, din, da, rounds, dbad); }));
  unsigned bad_a = 0;
  CK(hipMemcpy(&bad_a, dbad, 4, hipMemcpyDeviceToHost));

  printf("%-50s %10s %14s %10s %9s %12s\n", "kernel", "ms", "G units/s", "sclk MHz", "watts", "units/joule");
  for (auto& r : res)
    printf("%-50s %10.2f %14.2f %10.0f %9.0f %12.3g\n", r.name, r.ms, r.units / r.ms / 1e6, r.mhz, r.watts,
           r.watts > 0 ? r.units / (r.ms / 1e3) / r.watts : 0.0);
  printf("(b) FP64 product equals the integer product mod p on every lane: %s (%zu words differ)\n", bad_b ? "NO" : "yes", bad_b);
  printf("(a) batched-affine inverses check (dx * 1/dx == 1): %s (%u lanes wrong)\n", bad_a ? "NO" : "yes", bad_a);
  printf("JSON {");
  for (size_t i = 0; i < res.size(); i++)
    printf("%s\"%s\": {\"ms\": %.3f, \"G_per_s\": %.3f, \"sclk_mhz\": %.0f, \"watts\": %.0f}", i ? ", " : "", res[i].name, res[i].ms,
           res[i].units / res[i].ms / 1e6, res[i].mhz, res[i].watts);
  printf(", \"fp64_equal\": %s, \"batched_affine_ok\": %s}\n", bad_b ? "false" : "true", bad_a ? "false" : "true");
  return (bad_a || bad_b) ? 2 : 0;
}
