// bullet.hip -- device side of the inner-product (bullet) reduction inside DotProductProofLog::prove
//   Spartan/src/nizk/bullet.rs:32-132   BulletReductionProof::prove
// The reference folds the generator vector G each round; here G is never folded: the folded generator is
// G_k[i] = sum_{j = i mod n} s_j g_j for known coefficients s_j, so every L / R is a fixed-base MSM over the
// original stream with scalars a_i * s_j (host/prover_common.h: dplog_prove).  Those O(R) scalar vectors, the two
// cross inner products <a_L, b_R>, <a_R, b_L> and the folds of a, b, s with the round challenge all live on the
// device (R <= 32768 elements: launch-latency bound); a round moves only the partial points of L and R, the two
// inner products and the challenge across PCIe.
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "ctx.h"
#include "mailbox_dev.h"

namespace vpin {

constexpr int kBB = 256;

struct BulletState {
  fq *av = nullptr, *bv = nullptr, *sj = nullptr;  // [R] each
  fq *av2 = nullptr, *bv2 = nullptr;               // fused rounds: the folded vectors are double-buffered
  void* fparts = nullptr;                          // fused rounds, R > 4096: the MSM workgroups' partial points (2 x R/32 x 128 B)
  int cur = 0;                                     // fused rounds: 0 = the live vectors are in av / bv, 1 = in av2 / bv2
  fq* rows = nullptr;                              // [2][R]: scalars of L and R over g_0..g_{R-1}
  fq* partials = nullptr;                          // [nblk][2] block partials of the two inner products
  void* msm_scratch = nullptr;                     // partial points of the round's two-row MSM (gens_msm_parts_launch)
  hipEvent_t ev_partials = nullptr;                // the inner products have reached the host
  size_t R = 0;
  int nblk = 0;
};

__global__ __launch_bounds__(kBB) void bullet_init_kernel(fq* __restrict__ sj, size_t R) {
  size_t j = (size_t)blockIdx.x * kBB + threadIdx.x;
  if (j < R) fq_store(sj + j, fq_one());
}

// n = half of the live length.  rows[0][j] = a_L . G_R scalars, rows[1][j] = a_R . G_L scalars (bullet.rs:63-83);
// partials[blk] = block sums of a_L[i]*b_R[i] and a_R[i]*b_L[i].
__global__ __launch_bounds__(kBB) void bullet_rows_kernel(const fq* __restrict__ av, const fq* __restrict__ bv,
                                                          const fq* __restrict__ sj, size_t n, size_t R, fq* __restrict__ rows,
                                                          fq* __restrict__ partials, int used) {
  const size_t j = (size_t)blockIdx.x * kBB + threadIdx.x;
  fq pl = fq_zero(), pr = fq_zero();
  if (j < R) {
    const size_t pos = j & (2 * n - 1);
    const fq s = fq_load(sj + j);
    if (pos >= n) {
      fq_store(rows + j, fq_mul(fq_load(av + (pos - n)), s));
      fq_store(rows + R + j, fq_zero());
    } else {
      fq_store(rows + j, fq_zero());
      fq_store(rows + R + j, fq_mul(fq_load(av + (n + pos)), s));
    }
    if (j < n) {
      pl = fq_mul(fq_load(av + j), fq_load(bv + n + j));
      pr = fq_mul(fq_load(av + n + j), fq_load(bv + j));
    }
  }
  __shared__ fq sh[kBB / 64][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  pl = fq_wave_sum(pl);
  pr = fq_wave_sum(pr);
  if (lane == 0) { sh[wave][0] = pl; sh[wave][1] = pr; }
  __syncthreads();
  if (threadIdx.x < 2 && (int)blockIdx.x < used) {  // the blocks past i < n hold zeros nobody reads
    fq s = sh[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < kBB / 64; w++) s = fq_add(s, sh[w][threadIdx.x]);
    fq_store(partials + 2 * (size_t)blockIdx.x + threadIdx.x, s);
  }
}

// a'[i] = a_L[i]*u + u^-1*a_R[i]; b'[i] = b_L[i]*u^-1 + u*b_R[i]; s_j *= u^-1 (j in a left half) or u (bullet.rs:99-109)
__global__ __launch_bounds__(kBB) void bullet_fold_kernel(fq* __restrict__ av, fq* __restrict__ bv, fq* __restrict__ sj, size_t n,
                                                          size_t R, fq u, fq u_inv) {
  const size_t j = (size_t)blockIdx.x * kBB + threadIdx.x;
  if (j >= R) return;
  fq_store(sj + j, fq_mul(fq_load(sj + j), ((j & (2 * n - 1)) < n) ? u_inv : u));
  if (j < n) {
    const fq al = fq_load(av + j), ar = fq_load(av + n + j), bl = fq_load(bv + j), br = fq_load(bv + n + j);
    fq_store(av + j, fq_add(fq_mul(al, u), fq_mul(u_inv, ar)));
    fq_store(bv + j, fq_add(fq_mul(bl, u_inv), fq_mul(u, br)));
  }
}

constexpr size_t kBulletPinned = 64 * 1024;

uint8_t* bullet_pinned(vpin_ctx* c) {
  if (!c->h_bullet) {
    if (hipHostMalloc(&c->h_bullet, kBulletPinned, hipHostMallocDefault) != hipSuccess) c->h_bullet = nullptr;
    else memset(c->h_bullet, 0, kBulletPinned);  // mailbox pieces start with sequence number 0 = never published
  }
  return (uint8_t*)c->h_bullet;
}

constexpr size_t kFusedMaxR = 32768;  // up to 4096: 2 x R/32 partial points of 128 B in the first half of the pinned buffer; beyond: summed per 128 on the device
bool bullet_fused(const BulletState* st) {
  static const bool off = getenv("VPIN_BULLET_CLASSIC") != nullptr;
  return st && !off && st->av2 && st->R % 32 == 0 && st->R <= kFusedMaxR;
}

void bullet_free(vpin_ctx* c, BulletState* st) {
  if (!st) return;
  for (void* p : {(void*)st->av, (void*)st->bv, (void*)st->sj, (void*)st->rows, (void*)st->partials, st->msm_scratch, (void*)st->av2,
                  (void*)st->bv2, st->fparts})
    if (p) dev_free(c, p);
  if (st->ev_partials) (void)hipEventDestroy(st->ev_partials);
  delete st;
}

// x, a: R Montgomery scalars each (the vectors DotProductProofLog::prove reduces); s_j = 1
int bullet_begin(vpin_ctx* c, const uint8_t* x_mont, const uint8_t* a_mont, size_t R, BulletState** out) {
  if (!c || !x_mont || !a_mont || !out || !is_pow2(R)) return VPIN_EINVAL;
  (void)hipSetDevice(c->device);
  BulletState* st = new (std::nothrow) BulletState();
  if (!st) return VPIN_ENOMEM;
  st->R = R;
  st->nblk = (int)((R + kBB - 1) / kBB);
  const bool want_fused = R % 32 == 0 && R <= kFusedMaxR && bullet_pinned(c) != nullptr;
  if ((want_fused && (dev_alloc(c, R * 32, (void**)&st->av2) || dev_alloc(c, R * 32, (void**)&st->bv2) ||
                      (R > 4096 && dev_alloc(c, 2 * (R / 32) * 128, &st->fparts)))) ||
      dev_alloc(c, R * 32, (void**)&st->av) || dev_alloc(c, R * 32, (void**)&st->bv) || dev_alloc(c, R * 32, (void**)&st->sj) ||
      dev_alloc(c, 2 * R * 32, (void**)&st->rows) || dev_alloc(c, (size_t)st->nblk * 64, (void**)&st->partials) ||
      dev_alloc(c, gens_msm_parts_scratch_bytes(2, R), &st->msm_scratch) ||
      hipEventCreateWithFlags(&st->ev_partials, hipEventDisableTiming | hipEventReleaseToSystem) != hipSuccess) {
    bullet_free(c, st);
    return VPIN_ENOMEM;
  }
  hipError_t e = hipMemcpyAsync(st->av, x_mont, R * 32, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(st->bv, a_mont, R * 32, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(bullet_init_kernel, dim3(st->nblk), dim3(kBB), 0, c->stream, st->sj, R);
    e = hipGetLastError();
  }
  // no wait here: x_mont / a_mont must stay valid until the first round's results are back (dplog_prove's vectors do), and
  // the uploads overlap the host's transcript work before the first round
  if (e != hipSuccess) { set_last_error("bullet_begin", e); bullet_free(c, st); return VPIN_EHIP; }
  *out = st;
  return VPIN_OK;
}

// One round at half length n, in two halves so the host's work overlaps the MSM:
//   bullet_round_begin : rows kernel, inner products to the host (event), the two-row MSM and its copy enqueued;
//                        returns the inner products c_L, c_R as soon as THEY are there (the MSM is still running)
//   bullet_round_end   : waits for the partial points of L and R (2 x vpin_gens_msm_parts_count(R) x 128 B, over the R
//                        stream generators only; the c*Q and blind*H terms are the caller's, computed in between)
int bullet_round_begin(vpin_ctx* c, const vpin_gens* g, BulletState* st, size_t n, uint8_t* parts_xyzt, uint8_t cLR[64]) {
  if (!c || !g || !st || !parts_xyzt || !cLR || n == 0 || 2 * n > st->R) return VPIN_EINVAL;
  (void)hipSetDevice(c->device);
  // only the blocks covering i < n carry non-zero partials
  const int used = (int)((n + kBB - 1) / kBB);
  // Pinned staging (second half of the context's buffer; the caller's partial points use the first half).  Pinned host
  // memory is device-addressable: the kernels store the block partials and the partial points straight into it, so a
  // round has no copy command at all (each costs ~10 us of latency here).
  uint8_t* pin = bullet_pinned(c);
  const bool direct = pin && (size_t)used * 64 <= kBulletPinned / 2;
  std::vector<fq> pageable;
  fq* part;
  if (direct) part = (fq*)(pin + kBulletPinned / 2);
  else { pageable.resize((size_t)used * 2); part = pageable.data(); }
  hipLaunchKernelGGL(bullet_rows_kernel, dim3(st->nblk), dim3(kBB), 0, c->stream, (const fq*)st->av, (const fq*)st->bv,
                     (const fq*)st->sj, n, st->R, st->rows, direct ? part : st->partials, used);
  VPIN_HIP_TRY(hipGetLastError());
  if (!direct) VPIN_HIP_TRY(hipMemcpyAsync(part, st->partials, (size_t)used * 64, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipEventRecord(st->ev_partials, c->stream));
  int rc = gens_msm_parts_launch(c, g, st->rows, 2, st->R, st->msm_scratch, parts_xyzt, pin && parts_xyzt == pin);
  if (rc) return rc;
  // spin on the event: a blocking wait costs more than the ~10 us this copy takes
  for (long spins = 0;; spins++) {
    const hipError_t q = hipEventQuery(st->ev_partials);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) { set_last_error("bullet_round_begin: hipEventQuery", q); return VPIN_EHIP; }
    if (spins > 2000000) { VPIN_HIP_TRY(hipEventSynchronize(st->ev_partials)); break; }
  }
  // block partials summed on the host (<= 64 pairs): plain modular additions of Montgomery values
  auto add = [](fq& a, const fq& b) {
    uint64_t cy = 0;
    uint32_t t[8];
    for (int i = 0; i < 8; i++) { cy += (uint64_t)a.v[i] + b.v[i]; t[i] = (uint32_t)cy; cy >>= 32; }
    uint32_t d[8];
    int64_t bw = 0;
    for (int i = 0; i < 8; i++) { bw += (int64_t)t[i] - (int64_t)fq_modulus_limb(i); d[i] = (uint32_t)bw; bw >>= 32; }
    for (int i = 0; i < 8; i++) a.v[i] = bw ? t[i] : d[i];
  };
  fq sums[2] = {part[0], part[1]};
  for (int b = 1; b < used; b++) { add(sums[0], part[2 * (size_t)b]); add(sums[1], part[2 * (size_t)b + 1]); }
  memcpy(cLR, sums, 64);
  return VPIN_OK;
}

int bullet_round_end(vpin_ctx* c) {
  for (long spins = 0;; spins++) {
    const hipError_t q = hipStreamQuery(c->stream);
    if (q == hipSuccess) return VPIN_OK;
    if (q != hipErrorNotReady) { set_last_error("bullet_round_end: hipStreamQuery", q); return VPIN_EHIP; }
    if (spins > 2000000) break;
  }
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  return VPIN_OK;
}

// asynchronous: the next bullet_round / bullet_finish is ordered behind it on the stream
int bullet_fold(vpin_ctx* c, BulletState* st, size_t n, const uint8_t u[32], const uint8_t u_inv[32]) {
  if (!c || !st || !u || !u_inv || n == 0 || 2 * n > st->R) return VPIN_EINVAL;
  fq fu, fi;
  memcpy(fu.v, u, 32);
  memcpy(fi.v, u_inv, 32);
  hipLaunchKernelGGL(bullet_fold_kernel, dim3(st->nblk), dim3(kBB), 0, c->stream, st->av, st->bv, st->sj, n, st->R, fu, fi);
  VPIN_HIP_TRY(hipGetLastError());
  return VPIN_OK;
}


// ---- fused rounds: one launch each (msm.hip bullet_step_kernel) ---------------------------------------------------
static inline uint32_t* fused_up(vpin_ctx* c) { return reinterpret_cast<uint32_t*>((uint8_t*)c->h_bullet + kBulletPinned / 2); }

static int fused_take(vpin_ctx* c, int nslots, uint32_t seq, fq sums[2]) {
  const uint32_t* up = fused_up(c);
  auto add = [](fq& a, const fq& b) {
    uint64_t cy = 0;
    uint32_t t[8];
    for (int i = 0; i < 8; i++) { cy += (uint64_t)a.v[i] + b.v[i]; t[i] = (uint32_t)cy; cy >>= 32; }
    uint32_t d[8];
    int64_t bw = 0;
    for (int i = 0; i < 8; i++) { bw += (int64_t)t[i] - (int64_t)fq_modulus_limb(i); d[i] = (uint32_t)bw; bw >>= 32; }
    for (int i = 0; i < 8; i++) a.v[i] = bw ? t[i] : d[i];
  };
  for (int b = 0; b < nslots; b++)
    for (int k = 0; k < 2; k++) {
      fq v;
      for (long spins = 0; !tail_take(up + (2 * (size_t)b + k) * 12, seq, &v); spins++) {
        __builtin_ia32_pause();
        if ((spins & 0xffff) == 0xffff) {
          const hipError_t q = hipStreamQuery(c->stream);  // a finished stream without the scalars = a failed launch
          if (q != hipErrorNotReady && !tail_take(up + (2 * (size_t)b + k) * 12, seq, &v)) {
            set_last_error("bullet_step: the round kernel ended without publishing its inner products", q);
            return VPIN_EHIP;
          }
        }
      }
      if (b == 0) sums[k] = v; else add(sums[k], v);
    }
  return VPIN_OK;
}

int bullet_step(vpin_ctx* c, const vpin_gens* g, BulletState* st, size_t n, const uint8_t* u_prev, const uint8_t* u_inv_prev,
                uint8_t cLR[64]) {
  if (!c || !g || !st || !cLR || !bullet_fused(st) || n == 0 || 2 * n > st->R) return VPIN_EINVAL;
  uint8_t* pin = bullet_pinned(c);
  if (!pin) return VPIN_ENOMEM;
  const uint32_t seq = ++c->bullet_seq;
  fq* ap = st->cur ? st->av2 : st->av; fq* bp = st->cur ? st->bv2 : st->bv;
  fq* an = st->cur ? st->av : st->av2; fq* bn = st->cur ? st->bv : st->bv2;
  int rc = bullet_step_launch(c, g, ap, bp, an, bn, st->sj, n, st->R, u_prev != nullptr, false, u_prev, u_inv_prev, pin, fused_up(c), seq,
                              st->fparts);
  if (rc) return rc;
  st->cur ^= 1;
  fq sums[2];
  if ((rc = fused_take(c, (int)((n + 255) / 256), seq, sums))) return rc;
  memcpy(cLR, sums, 64);
  return VPIN_OK;
}

size_t bullet_part_ptrs(vpin_ctx* c, const BulletState* st, size_t n, int row, const uint8_t** out) {
  const uint8_t* base = (const uint8_t*)c->h_bullet;
  const size_t nblk = st->R / 32;
  size_t k = 0;
  if (nblk > 128) {  // summed per 128 workgroups on the device: [rows][nblk / 128], every entry valid
    const size_t G = nblk / 128;
    for (size_t g = 0; g < G; g++) out[k++] = base + ((size_t)row * G + g) * 128;
    return k;
  }
  for (size_t b = 0; b < nblk; b++) {  // a workgroup of 32 generators holds one side while n >= 32, both below
    const bool is_L = n == 0 || ((b * 32) & (2 * n - 1)) >= n;
    if (n != 0 && n < 32) { out[k++] = base + ((size_t)row * nblk + b) * 128; continue; }
    if (is_L == (row == 0)) out[k++] = base + ((size_t)row * nblk + b) * 128;
  }
  return k;
}

int bullet_finish_fused(vpin_ctx* c, const vpin_gens* g, BulletState* st, const uint8_t u[32], const uint8_t u_inv[32],
                        uint8_t xhat_ahat[64]) {
  if (!c || !g || !st || !u || !u_inv || !xhat_ahat || !bullet_fused(st)) return VPIN_EINVAL;
  uint8_t* pin = bullet_pinned(c);
  if (!pin) return VPIN_ENOMEM;
  const uint32_t seq = ++c->bullet_seq;
  fq* ap = st->cur ? st->av2 : st->av; fq* bp = st->cur ? st->bv2 : st->bv;
  int rc = bullet_step_launch(c, g, ap, bp, nullptr, nullptr, st->sj, 0, st->R, false, true, u, u_inv, pin, fused_up(c), seq, st->fparts);
  if (rc) return rc;
  fq sums[2];
  if ((rc = fused_take(c, 1, seq, sums))) return rc;
  memcpy(xhat_ahat, sums, 64);
  return bullet_round_end(c);
}

// after the last fold: x_hat = a[0], a_hat = b[0] and the partial points of g_hat = sum_j s_j g_j
int bullet_finish(vpin_ctx* c, const vpin_gens* g, BulletState* st, uint8_t xhat_ahat[64], uint8_t* parts_xyzt) {
  if (!c || !g || !st || !xhat_ahat || !parts_xyzt) return VPIN_EINVAL;
  VPIN_HIP_TRY(hipMemcpyAsync(xhat_ahat, st->av, 32, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipMemcpyAsync(xhat_ahat + 32, st->bv, 32, hipMemcpyDeviceToHost, c->stream));
  return gens_msm_parts_dev(c, g, st->sj, 1, st->R, parts_xyzt);
}

}  // namespace vpin
